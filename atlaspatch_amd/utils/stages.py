"""Wall-clock stage accounting for one CLI invocation (the reference has no timers; section 8d asks
for end-to-end next to kernel-only).  ``with stage("name"):`` accumulates seconds and counts per stage in a process-wide
table; ``snapshot(reset=True)`` hands it to whoever reports (bench.py's `rates`, `--verbose`).  Stages may nest and may run
on worker threads; the table is only ever a sum, so concurrent stages can add up to more than the wall time."""
from __future__ import annotations

import threading
import time
from contextlib import contextmanager

_lock = threading.Lock()
_table: dict[str, list] = {}


@contextmanager
def stage(name: str):
    t0 = time.perf_counter()
    try:
        yield
    finally:
        dt = time.perf_counter() - t0
        with _lock:
            rec = _table.setdefault(name, [0.0, 0])
            rec[0] += dt
            rec[1] += 1


def add(name: str, seconds: float, count: int = 1) -> None:
    with _lock:
        rec = _table.setdefault(name, [0.0, 0])
        rec[0] += float(seconds)
        rec[1] += int(count)


def snapshot(reset: bool = False) -> dict:
    with _lock:
        out = {k: {"seconds": round(v[0], 4), "count": v[1]} for k, v in _table.items()}
        if reset:
            _table.clear()
    return out
