"""The LAPACK numpy itself links, by address (for rwh_host_dlt4_svd, include/rwh.h).

`numpy.linalg.svd` is `dgesdd` of the OpenBLAS bundled with the numpy wheel.  The settle step of `RANSAC.run` needs
exactly that routine's results (LAPACK's null vector of a rank-deficient 8 x 9 system is arbitrary, and it is what the
reference uses), but calling it through numpy costs ~11 us per 8 x 9 matrix under the interpreter lock.  `dgesdd_address()`
returns the address of the same symbol in the library numpy has already loaded, so that the native loop in librwh_hip.so
can call it from several threads; None if it cannot be found (other numpy builds): callers then stay on numpy.linalg.svd,
which gives the same numbers, only slower."""
import ctypes
import os

_addrs = {}


def routine_address(routine):
    """Address of LAPACK `routine` ('dgesdd', 'dgesv': what numpy.linalg.svd / numpy.linalg.inv call) in the ILP64 OpenBLAS numpy
    has loaded, or None."""
    if routine in _addrs:
        return _addrs[routine]
    _addrs[routine] = None
    try:
        import numpy.linalg._umath_linalg  # noqa: F401
        paths = set()
        with open("/proc/self/maps") as f:
            for line in f:
                p = line.rsplit(" ", 1)[-1].strip()
                if "openblas" in os.path.basename(p) and "numpy" in p:
                    paths.add(p)
        for p in sorted(paths):
            lib = ctypes.CDLL(p)
            for name in ("scipy_%s_64_" % routine, "%s_64_" % routine):
                try:
                    _addrs[routine] = ctypes.cast(getattr(lib, name), ctypes.c_void_p).value
                    return _addrs[routine]
                except AttributeError:
                    continue
    except Exception:
        _addrs[routine] = None
    return _addrs[routine]


def dgesv_address():
    return routine_address("dgesv")


def dgesdd_address():
    return routine_address("dgesdd")
