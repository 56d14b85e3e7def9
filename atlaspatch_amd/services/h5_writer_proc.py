"""Coordinate H5 files written by helper PROCESSES.

``write_coords`` (reference services/storage.py:106-161) costs 10-20 ms of libhdf5 + Python per 100 000^2 slide (10.6 MB:
59 k rows of coords + S160 passports).  The system's libhdf5 is not thread-safe, so every call of every thread goes through
one lock (utils/h5lite.py, as h5py does): eight coordinate workers write their slides one after the other and hold the
interpreter lock for the Python part, next to a SAM2 forward of 5 ms per slide.  Processes do not share that lock.

``H5WriterPool`` keeps up to ``workers`` children of ``python -m atlaspatch_amd.services.h5_writer_proc`` (numpy + libhdf5
only -- the passports arrive formatted, so the HIP library is not loaded: 0.3 s to start, side by side, in the background).  A job
= the ``H5PatchWriter`` keyword arguments, the output path, the int32 [N, 5] coords and their S160 passports; the child runs the
very same ``H5PatchWriter.write_coords_array`` -> identical bytes on disk (tests/test_host_logic.py).  Any failure to start or
talk to a child makes the caller write in-process instead: a child that does not answer within ``ATLASPATCH_H5_PROC_TIMEOUT``
seconds (default 60, plus one second per 10 000 rows of the job: an NFS stall, a stopped process) is killed by a watchdog, which
turns the blocked pipe I/O into an error -- its half-written ``.<name>.tmp.<uuid>`` file is removed and the loss is logged; a
child that REPORTS a write error stays in the pool and the caller retries the write in-process, where the exception (if it is
real) surfaces with its own traceback; a dead child is replaced in the background, so one failure does not send the rest of the
run in-process.

Protocol (stdin / stdout of the child, binary): 8-byte little-endian length + pickle of {"kwargs", "path", "rows"} followed by
rows * 20 bytes of coords and rows * 160 bytes of passports; reply: 8-byte length + pickle of {"ok": n} or {"error": text}.
"""
from __future__ import annotations

import glob
import logging
import os
import pickle
import queue
import struct
import subprocess
import sys
import threading

import numpy as np


def _read_exact(stream, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = stream.read(n - len(buf))
        if not chunk:
            raise EOFError("h5 writer pipe closed")
        buf += chunk
    return bytes(buf)


def _send(stream, obj, payload: bytes = b"") -> None:
    head = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    stream.write(struct.pack("<Q", len(head)))
    stream.write(head)
    if payload:
        stream.write(payload)
    stream.flush()


def _recv(stream):
    (n,) = struct.unpack("<Q", _read_exact(stream, 8))
    return pickle.loads(_read_exact(stream, n))


log = logging.getLogger("atlaspatch_amd.h5_writer_proc")


def _timeout() -> float:
    try:
        return max(1.0, float(os.environ.get("ATLASPATCH_H5_PROC_TIMEOUT", "60")))
    except ValueError:
        return 60.0


def _job_timeout(rows: int) -> float:
    """The per-job limit grows with the job: the base (ATLASPATCH_H5_PROC_TIMEOUT) plus one second per 10 000 rows (1.8 MB), so a
    large coordinate block on a slow but PROGRESSING file system is not mistaken for a hang."""
    return _timeout() + rows / 10000.0


def _start_timeout() -> float:
    """The hello of a starting child covers an interpreter start, the numpy / libhdf5 imports and a warm-up write: its own,
    longer limit (ATLASPATCH_H5_PROC_START_TIMEOUT, default max(60, the job limit)) -- a short job limit must not kill a
    replacement child on a loaded machine."""
    try:
        return max(1.0, float(os.environ.get("ATLASPATCH_H5_PROC_START_TIMEOUT", "") or max(60.0, _timeout())))
    except ValueError:
        return max(60.0, _timeout())


def _remove_leftovers(path: str) -> None:
    """A child killed mid-write leaves its ``.<name>.tmp.<uuid>`` (utils/h5.py) next to the target: remove them."""
    folder, base = os.path.split(os.path.abspath(path))
    for stale in glob.glob(os.path.join(glob.escape(folder), f".{glob.escape(base)}.tmp.*")):
        try:
            os.unlink(stale)
        except OSError:
            pass


class _Watchdog:
    """Kills ``proc`` when the guarded section takes longer than ``seconds``: the blocked read / write then fails with EOF /
    EPIPE and the caller falls back.  (select() on the pipes would cover reads only; a 10-MB job also blocks in write.)"""

    def __init__(self, proc, seconds: float) -> None:
        self.fired = False
        self._timer = threading.Timer(seconds, self._fire, args=(proc,))
        self._timer.daemon = True

    def _fire(self, proc) -> None:
        self.fired = True
        try:
            proc.kill()
        except Exception:  # noqa: BLE001
            pass

    def __enter__(self):
        self._timer.start()
        return self

    def __exit__(self, *exc) -> None:
        self._timer.cancel()


class ChildWriteError(RuntimeError):
    pass


class _Worker:
    def __init__(self) -> None:
        env = dict(os.environ)
        env.setdefault("OMP_NUM_THREADS", "1")
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        self.proc = subprocess.Popen([sys.executable, "-m", "atlaspatch_amd.services.h5_writer_proc"], stdin=subprocess.PIPE,
                                     stdout=subprocess.PIPE, env=env, close_fds=True)
        try:
            with _Watchdog(self.proc, _start_timeout()):
                hello = _recv(self.proc.stdout)
            if hello.get("ready") is not True:
                raise RuntimeError(f"h5 writer did not start: {hello}")
        except BaseException:
            self.proc.kill()                                  # never leave a half-started child behind
            self.proc.wait()
            raise

    def write(self, kwargs: dict, path: str, coords: np.ndarray, passports: np.ndarray) -> int:
        """-> rows written; raises ``ChildWriteError`` when the child reports a failed write (it is still healthy), an I/O error
        when the child is gone or was killed by the watchdog."""
        with _Watchdog(self.proc, _job_timeout(int(coords.shape[0]))):
            _send(self.proc.stdin, {"kwargs": kwargs, "path": path, "rows": int(coords.shape[0])}, coords.tobytes())
            self.proc.stdin.write(memoryview(passports).cast("B"))
            self.proc.stdin.flush()
            reply = _recv(self.proc.stdout)
        if "error" in reply:
            raise ChildWriteError(reply["error"])
        return int(reply["ok"])

    def close(self) -> None:
        try:
            if self.proc.poll() is None:
                _send(self.proc.stdin, {"quit": True})
                self.proc.stdin.close()
                self.proc.wait(timeout=5)
        except Exception:  # noqa: BLE001
            self.proc.kill()
        try:
            self.proc.wait(timeout=5)
        except Exception:  # noqa: BLE001
            pass


class H5WriterPool:
    """``write(kwargs, path, coords)`` from any thread.  ``prestart()`` launches the ``workers`` children in the background; a
    call that finds no child ready yet writes in-process (returns None) instead of waiting for an interpreter to start, a call
    that finds every child busy waits for one (they are busy for milliseconds)."""

    def __init__(self, workers: int) -> None:
        self.workers = max(1, int(workers))
        self._idle: "queue.Queue[_Worker]" = queue.Queue()
        self._ready = 0                               # children that have said hello
        self._starting = False
        self._lock = threading.Lock()
        self._closed = False
        self.broken = False                           # the LAST start attempt failed (cleared by the next successful one)
        self.jobs = 0
        self.child_errors = 0                         # jobs a child reported as failed (retried in-process by the caller)
        self.restarts = 0

    def _start_one(self, name: str) -> None:
        def run():
            if self._closed:
                return
            try:
                w = _Worker()
            except Exception:  # noqa: BLE001
                self.broken = True
                return
            with self._lock:
                self._ready += 1
                self.broken = False
                closed = self._closed
            if closed:
                w.close()
            else:
                self._idle.put(w)
        threading.Thread(target=run, name=name, daemon=True).start()

    def prestart(self) -> None:
        with self._lock:
            if self._starting or self._closed:
                return
            self._starting = True
        for k in range(self.workers):                 # the interpreters start side by side
            self._start_one(f"h5-writer-start-{k}")

    def ready(self) -> bool:
        """True once at least one child answers (before that a caller should not even format passports for it)."""
        with self._lock:
            return self._ready > 0 and not self._closed

    def _take(self):
        try:
            return self._idle.get_nowait()
        except queue.Empty:
            pass
        with self._lock:
            ready = self._ready
        if ready == 0:
            self.prestart()
            return None                               # nobody has started yet: do not wait for an interpreter
        try:
            return self._idle.get(timeout=2.0)        # every ready child is writing: one frees up within milliseconds
        except queue.Empty:
            return None

    def write(self, kwargs: dict, path: str, coords: np.ndarray, passports: np.ndarray):
        """-> rows written, or None when no child is usable right now (the caller then writes in-process).  ``passports``: the
        S160 strings of the rows, formatted by the caller (natively, outside the interpreter lock): the children load numpy and
        libhdf5 only, not the HIP library, and start in a quarter of a second."""
        if self._closed:
            return None
        w = self._take()
        if w is None:
            return None
        try:
            n = w.write(kwargs, path, np.ascontiguousarray(coords, dtype=np.int32).reshape(-1, 5),
                        np.ascontiguousarray(passports))
        except ChildWriteError as exc:
            with self._lock:
                self.child_errors += 1
            log.warning("h5 writer child could not write %s (%s); writing it in-process", path, exc)
            self._idle.put(w)                         # the child is healthy; the caller repeats the write in-process, where a real
            return None                               # error raises with its own traceback (like every other failure mode here)
        except (EOFError, BrokenPipeError, OSError, ValueError, struct.error, pickle.UnpicklingError) as exc:
            with self._lock:                          # the child died, hung (watchdog) or desynchronised: replace it
                self._ready -= 1
                self.restarts += 1
                n_restart = self.restarts
            log.warning("h5 writer child lost while writing %s (%s: %s); writing it in-process, replacing the child", path,
                        type(exc).__name__, exc)
            w.close()
            _remove_leftovers(path)                   # what the killed child had written so far
            if not self._closed and n_restart <= 4 * self.workers:      # a child that can never start must not respawn forever
                self._start_one(f"h5-writer-restart-{n_restart}")
            return None
        with self._lock:
            self.jobs += 1
        self._idle.put(w)
        return n

    def close(self) -> None:
        self._closed = True
        while True:
            try:
                self._idle.get_nowait().close()
            except queue.Empty:
                break


_POOL = None
_POOL_LOCK = threading.Lock()


def default_workers() -> int:
    raw = os.environ.get("ATLASPATCH_H5_PROCS")
    if raw not in (None, ""):
        return max(0, int(raw))
    return max(0, min(8, (os.cpu_count() or 1) // 4))


def shared_pool(create: bool = True):
    """The process-wide pool (None when disabled: ATLASPATCH_H5_PROCS=0 or fewer than 4 hardware threads)."""
    global _POOL
    with _POOL_LOCK:
        if _POOL is None and create:
            n = default_workers()
            if n > 0:
                import atexit
                _POOL = H5WriterPool(n)
                atexit.register(_POOL.close)
        return _POOL


def _serve() -> None:
    from .storage import H5PatchWriter
    # nothing but protocol frames may reach the pipe: the protocol gets a private duplicate of fd 1, and fd 1 itself -- which
    # C libraries (libhdf5's error stack, a stray printf) write to behind Python's back -- is pointed at stderr
    stdin = sys.stdin.buffer
    sys.stdout.flush()
    stdout = os.fdopen(os.dup(1), "wb")
    os.dup2(2, 1)
    sys.stdout = sys.stderr
    try:                                             # load libhdf5 BEFORE saying hello: the first job would otherwise pay for it
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            H5PatchWriter(chunk_rows=8, patch_size=1, patch_size_level0=1, level0_mag=1, target_mag=1, level0_wh=(1, 1), overlap=0,
                          slide_stem="warm", wsi_path="").write_coords_array(os.path.join(tmp, "warm.h5"),
                                                                            np.zeros((3, 5), dtype=np.int32),
                                                                            passports=np.zeros((3,), dtype="S160"))
    except Exception as exc:  # noqa: BLE001
        _send(stdout, {"ready": False, "error": f"{type(exc).__name__}: {exc}"})
        return
    _send(stdout, {"ready": True})
    while True:
        try:
            job = _recv(stdin)
        except EOFError:
            return
        if job.get("quit"):
            return
        try:
            rows = int(job["rows"])
            coords = np.frombuffer(_read_exact(stdin, rows * 20), dtype=np.int32).reshape(rows, 5)
            passports = np.frombuffer(_read_exact(stdin, rows * 160), dtype="S160")
            n = H5PatchWriter(**job["kwargs"]).write_coords_array(job["path"], coords, passports=passports)
            _send(stdout, {"ok": int(n)})
        except Exception as exc:  # noqa: BLE001
            _send(stdout, {"error": f"{type(exc).__name__}: {exc}"})


if __name__ == "__main__":
    _serve()
