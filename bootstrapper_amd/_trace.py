"""BSMI_IO_TRACE=1: where the wall time of the drivers goes (`bs predict`, `bs segment`), printed to stderr.  Off by default."""
import contextlib
import os
import sys
import threading
import time

ON = os.environ.get("BSMI_IO_TRACE", "") not in ("", "0")
_T0 = time.perf_counter()
_acc = {}
_lock = threading.Lock()


@contextlib.contextmanager
def span(label, accumulate=False):
    """time a region; accumulate=True: summed per label (worker threads), reported by `report`"""
    if not ON:
        yield
        return
    t = time.perf_counter()
    try:
        yield
    finally:
        dt = time.perf_counter() - t
        if accumulate:
            with _lock:
                a = _acc.setdefault(label, [0.0, 0])
                a[0] += dt
                a[1] += 1
        else:
            print(f"[bsmi io {time.perf_counter() - _T0:8.3f}] {label}: {dt:.3f} s", file=sys.stderr, flush=True)


def report(prefix=""):
    if not ON:
        return
    with _lock:
        for k, (s, n) in sorted(_acc.items()):
            print(f"[bsmi io {time.perf_counter() - _T0:8.3f}] {prefix}{k}: {s:.3f} s in {n} calls (summed over threads)", file=sys.stderr, flush=True)
        _acc.clear()
