"""Pinned, multi-slot tile staging ring: host decode threads -> pinned host slots -> HBM.

Replaces the reference's serial main-thread tile loop (services/feature_embedding.py:81-96: one
``wsi.extract`` per H5 row, then a fresh DataLoader per 32-patch batch) with a pipeline:

    decode threads (wsi.extract, pinned to the slot's pinned-memory view)
        -> hipMemcpyAsync on a copy stream into one of `slots` device buffers
        -> encoder forward on the compute stream (waits on the copy event)
        -> features copied back asynchronously into a pinned [N, D] matrix

A pinned host slot is refilled as soon as its H2D copy has completed; the device slot is reused only after the
forward that read it has finished (the copy stream waits on that event, the CPU does not).  The ring is sized
in tiles, not bytes: ``slots x batch`` tiles of ``ps x ps x 3`` bytes (2 x 1024 x 196 608 B =
403 MB pinned at the defaults), trivial next to 288 GB of HBM, and deep enough to cover the
~ms-scale decode latency of real slides.
"""
from __future__ import annotations

import concurrent.futures as futures
from typing import Callable, Sequence

import ctypes as C

import numpy as np
import torch

from .. import _lib


def _cpu_list(text: str) -> list:
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def _device_bdf(device) -> str | None:
    """PCI address ``dddd:bb:dd.0`` of a torch device (torch exposes the three fields as integers)."""
    try:
        p = torch.cuda.get_device_properties(device)
        return f"{int(p.pci_domain_id):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}.0"
    except (AttributeError, RuntimeError, AssertionError, TypeError, ValueError):
        return None


def _pin_order(device, *, sysfs: str = "/sys", allowed=None, bdf: str | None = None,
               local_rank: int | None = None, local_world: int | None = None) -> list:
    """CPUs this rank's decode threads are pinned to, in order of preference: the GPU's NUMA node first, one hardware
    thread per physical core before any SMT sibling, all from the process's affinity mask.  Under a multi-rank launch
    (LOCAL_RANK / LOCAL_WORLD_SIZE) every class is dealt round-robin over the local ranks, so the ranks of one node take
    DISJOINT cores (eight ranks with the same mask would otherwise all pin to the same few cores).
    ``sysfs`` / ``allowed`` / ``bdf`` / ``local_*`` are injectable for the host-logic test (fake sysfs tree)."""
    import os
    if allowed is None:
        if not hasattr(os, "sched_getaffinity"):
            return []
        allowed = os.sched_getaffinity(0)
    allowed = sorted(allowed)
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0") or 0)
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
    local_world = max(1, local_world)
    local_rank = min(max(0, local_rank), local_world - 1)
    local = set(allowed)
    if bdf is None:
        bdf = _device_bdf(device)
    if bdf is not None:
        try:
            with open(f"{sysfs}/bus/pci/devices/{bdf}/numa_node") as fh:
                node = int(fh.read())
            if node >= 0:
                with open(f"{sysfs}/devices/system/node/node{node}/cpulist") as fh:
                    local = set(_cpu_list(fh.read())) & set(allowed) or set(allowed)
        except (OSError, ValueError):       # no topology information: every allowed CPU counts as local
            pass
    primary, sibling = [], []
    for cpu in allowed:
        try:
            with open(f"{sysfs}/devices/system/cpu/cpu{cpu}/topology/thread_siblings_list") as fh:
                first = min(_cpu_list(fh.read()))
        except (OSError, ValueError):
            first = cpu
        (primary if first == cpu else sibling).append(cpu)
    classes = ([c for c in primary if c in local], [c for c in primary if c not in local],
               [c for c in sibling if c in local], [c for c in sibling if c not in local])
    order = []
    for cls in classes:
        order += cls[local_rank::local_world]
    return order


class TileRing:
    def __init__(self, *, device: torch.device, batch: int, patch_size: int, slots: int = 3,
                 workers: int = 4, tile_hw: tuple | None = None) -> None:
        """``tile_hw`` = (read_h, read_w) of the tiles that cross the ring; defaults to the patch size.  Slides whose
        level read differs from ``patch_size`` ship tiles at their READ size and are resized on the device
        (the reference's per-tile ``cv2.resize``, feature_embedding.py:94-95), so the host does no resampling."""
        self.device = device
        self.batch = int(batch)
        self.ps = int(patch_size)
        self.th, self.tw = (int(tile_hw[0]), int(tile_hw[1])) if tile_hw is not None else (self.ps, self.ps)
        self.slots = max(2, int(slots))
        self.host = [torch.empty((self.batch, self.th, self.tw, 3), dtype=torch.uint8, pin_memory=True)
                     for _ in range(self.slots)]
        self.dev = [torch.empty((self.batch, self.th, self.tw, 3), dtype=torch.uint8, device=device)
                    for _ in range(self.slots)]
        self._out_host = None
        self.copy_stream = torch.cuda.Stream(device=device)
        self.free_events: list[torch.cuda.Event | None] = [None] * self.slots
        self.workers = max(1, int(workers))
        self._lib = _lib.load()
        # decode threads pinned one per host core (north star: "tile decode on host cores pinned"): cores of the NUMA node
        # the GPU hangs off first (the pinned slots live there and the H2D DMA reads them from there), one hardware thread
        # per physical core before any SMT sibling, all from the process's own affinity mask, dealt over the local ranks of a
        # multi-rank launch (disjoint cores per rank); the first core is left to the main thread that drives the streams.
        # ATLASPATCH_PIN_THREADS=0 disables.
        import itertools
        import os
        cores = _pin_order(device)
        pin = os.environ.get("ATLASPATCH_PIN_THREADS", "1") != "0" and len(cores) > 1
        counter = itertools.count()

        def _pin_worker():
            if pin:
                try:
                    os.sched_setaffinity(0, {cores[1 + next(counter) % (len(cores) - 1)]})      # pid 0 = the calling thread
                except OSError:
                    pass

        self.pool = futures.ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="tile", initializer=_pin_worker)

    def run(self, coords: np.ndarray, read_tile: Callable[[int, int, int, int, int], np.ndarray],
            forward: Callable[[torch.Tensor, torch.Tensor], None], out_dim: int, *,
            read_chunk: Callable | None = None, out_host: torch.Tensor | None = None) -> np.ndarray:
        """coords int32 [N, 5]; ``read_tile(x, y, rw, rh, lv)`` -> uint8 [th, tw, 3];
        ``forward(tiles_dev [n,th,tw,3], out_dev [n,D])`` enqueues on the current stream.
        ``read_chunk(rows, dst_ptr, tile_side) -> bool`` (optional): a backend's native batched decoder; when it
        returns True the chunk's tiles are already in the pinned slot.  Returns float32 [N, D] (host; a fresh array --
        or, when the caller passes its own pinned ``out_host`` [>= N, D], a view of that buffer: no extra copy)."""
        n_total = int(coords.shape[0])
        if n_total == 0:
            return np.empty((0, out_dim), dtype=np.float32)
        # grow-only pinned result buffer: re-pinning [N, D] for every slide costs more than the copy out of it
        own = out_host is None
        if own:
            if self._out_host is None or self._out_host.shape[0] < n_total or self._out_host.shape[1] != out_dim:
                self._out_host = torch.empty((max(n_total, self.batch), out_dim), dtype=torch.float32, pin_memory=True)
            out_host = self._out_host
        assert out_host.shape[0] >= n_total and out_host.shape[1] == out_dim and out_host.is_pinned()
        out_dev = [torch.empty((self.batch, out_dim), dtype=torch.float32, device=self.device)
                   for _ in range(self.slots)]
        compute = torch.cuda.current_stream(self.device)
        # batch cuts: the first batches ramp up (batch / 8, / 4, / 2, then full batches) so that the first forward starts
        # after an eighth of a slot has been decoded instead of a whole one -- on a 10 000-tile slide decoded at 20 k tiles/s
        # that is 13 ms instead of 100 ms of idle GPU in front of 350 ms of work.  Features do not depend on the cuts.
        bounds, lo, size = [], 0, max(1, min(self.batch, max(64, self.batch // 8)))
        while lo < n_total:
            hi = min(n_total, lo + size)
            bounds.append((lo, hi))
            lo, size = hi, min(self.batch, size * 2)
        nb = len(bounds)

        def fill(slot: int, b: int):
            lo, hi = bounds[b]
            base = self.host[slot].data_ptr()
            count = hi - lo
            chunk = max(1, -(-count // (4 * self.workers)))          # a few tasks per worker, not one per tile
            rows = coords[lo:hi].tolist()
            tile_bytes = self.th * self.tw * 3
            want_shape = (self.th, self.tw, 3)

            def some(start):
                # decode the chunk, then ONE ap_host_gather_tiles call copies it into the pinned slot with the
                # interpreter lock released (a NumPy slice assignment per tile would hold it for every 196 KB memcpy)
                stop = min(count, start + chunk)
                if read_chunk is not None and self.th == self.tw and \
                        read_chunk(rows[start:stop], base + start * tile_bytes, self.tw):
                    return
                tiles = []
                for i in range(start, stop):
                    x, y, rw, rh, lv = rows[i]
                    t = np.ascontiguousarray(read_tile(x, y, rw, rh, lv), dtype=np.uint8)
                    if t.shape != want_shape:
                        raise ValueError(f"tile source returned shape {t.shape}, expected {want_shape}")
                    tiles.append(t)
                ptrs = (C.c_void_p * len(tiles))(*[t.ctypes.data for t in tiles])
                _lib.check(self._lib.ap_host_gather_tiles(base + start * tile_bytes, ptrs, len(tiles), tile_bytes),
                           "ap_host_gather_tiles")

            return [self.pool.submit(some, s) for s in range(0, count, chunk)], count

        pending = {}
        try:
            for b in range(min(nb, self.slots)):         # prime the ring
                if self.free_events[b % self.slots] is not None:
                    self.free_events[b % self.slots].synchronize()
                    self.free_events[b % self.slots] = None
                pending[b] = fill(b % self.slots, b)
            for b in range(nb):
                slot = b % self.slots
                tasks, count = pending[b]
                for t in tasks:
                    t.result()
                del pending[b]
                with torch.cuda.stream(self.copy_stream):
                    if self.free_events[slot] is not None:           # device slot: the forward that read it is done
                        self.copy_stream.wait_event(self.free_events[slot])
                    self.dev[slot][:count].copy_(self.host[slot][:count], non_blocking=True)
                    copied = torch.cuda.Event()
                    copied.record(self.copy_stream)
                compute.wait_event(copied)
                forward(self.dev[slot][:count], out_dev[slot][:count])
                lo = bounds[b][0]
                out_host[lo:lo + count].copy_(out_dev[slot][:count], non_blocking=True)
                done = torch.cuda.Event()
                done.record(compute)
                self.free_events[slot] = done
                nxt = b + self.slots
                if nxt < nb:
                    # the pinned host slot may be refilled as soon as its H2D copy has completed; the device slot is
                    # protected by the copy stream waiting on `done` above, so the CPU never waits for a forward
                    copied.synchronize()
                    pending[nxt] = fill(slot, nxt)
        except BaseException:
            # a tile source failed (or the forward did): let the decode tasks that are still filling pinned slots
            # finish before the ring is reused for the next slide, then drain the device
            for tasks, _ in pending.values():
                for t in tasks:
                    try:
                        t.result()
                    except Exception:  # noqa: BLE001
                        pass
            torch.cuda.synchronize(self.device)
            raise
        torch.cuda.synchronize(self.device)
        return out_host[:n_total].numpy().copy() if own else out_host[:n_total].numpy()

    def close(self) -> None:
        self.pool.shutdown(wait=True)
