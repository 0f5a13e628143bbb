"""Host-side hygiene around the GPU path: what the CPU must NOT do while it drives the kernels."""


import os
import sys
import time

_DEBUG_TIMING = bool(os.environ.get("BLUEST_DEBUG_TIMING"))


class phase_clock(object):
    """wall-clock split of a host-side section (`tick(label)` after each phase; ~0.1 us each); BLUEST_DEBUG_TIMING=1 prints it"""

    def __init__(self, name):
        self.name, self.t, self.rows = name, time.perf_counter(), []

    def tick(self, label):
        now = time.perf_counter()
        self.rows.append((label, now - self.t))
        self.t = now

    def as_dict(self):
        return {k + "_ms": v * 1e3 for k, v in self.rows}

    def report(self):
        if _DEBUG_TIMING and self.rows:
            sys.stderr.write("[bluest timing] %s: %s\n" % (self.name, "  ".join("%s %.2f ms" % (k, v * 1e3) for k, v in self.rows)))


class host_section(object):
    """Context of the constructors and of solve(): two things on the HOST that cost more than the GPU work they surround.

    * the cyclic collector: these sections create ~1e5 short-lived Python objects (group lists, views); that is enough to
      trigger a FULL collection, which in a process that has torch imported walks ~1e6 objects (30-70 ms, i.e. as long as
      the whole set-up).  The automatic collector is paused and its previous state restored.
    * the BLAS thread pool: numpy hands a dot product of two K_tot-vectors (`samples @ e`, K_tot > 10^4) to OpenBLAS, which
      wakes one thread per visible core and lets each of them SPIN for ~100 ms afterwards.  In a container whose CPU quota
      is smaller than the core count it sees (16 of 256 on the GPU boxes used here) that burns the cgroup's quota and the
      kernel freezes every thread of the process, the one launching kernels included, for the rest of the 100 ms period:
      measured as ONE 50-75 ms hole per solve in the kernel trace (profiles/r02_host_stall.txt).  BLAS is limited to one
      thread inside the section (threadpoolctl, if installed; its previous limits are restored on exit)."""

    _controller = None
    _controller_key = None
    _BLAS_CARRIERS = ("numpy", "scipy.linalg", "scipy.sparse.linalg", "torch", "sklearn", "mkl")

    @classmethod
    def _blas(cls):
        # the scan (a dlopen of every loaded BLAS / OpenMP runtime) is repeated only when a package that brings its own BLAS has
        # been imported since (scipy.linalg is lazy), not whenever sys.modules grew
        key = tuple(name in sys.modules for name in cls._BLAS_CARRIERS)
        if cls._controller_key != key:
            cls._controller_key = key
            try:
                from threadpoolctl import ThreadpoolController
                cls._controller = ThreadpoolController()
            except Exception:                                  # optional dependency
                cls._controller = None
        return cls._controller

    def __enter__(self):
        import gc
        self._was = gc.isenabled()
        gc.disable()
        ctl = self._blas()
        self._limit = ctl.limit(limits=1, user_api="blas") if ctl is not None else None
        return self

    def __exit__(self, *exc):
        if self._limit is not None:
            self._limit.restore_original_limits()
        if self._was:
            import gc
            gc.enable()
        return False


def warm():
    """everything a host_section touches for the first time -- the import of threadpoolctl and its scan of the loaded
    libraries -- done at package import, NOT inside the first timed constructor: on a box whose image is still paging in
    (the first minutes of a fresh GPU lease) one first-touch file read costs up to a second, which the round-2 driver run
    booked as `sap_wallclock.cold.setup_s` = 1.0 s (11.9 ms on a box with a warm page cache)."""
    host_section._blas()


def in_host_section(fn):
    """decorator: run the whole method inside a host_section"""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        with host_section():
            return fn(*args, **kwargs)
    return wrapped
