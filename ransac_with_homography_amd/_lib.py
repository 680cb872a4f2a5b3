"""ctypes binding of librwh_hip.so (C ABI in include/rwh.h).

The product has no CPU fallback: if the library is missing or the process has
no MI355X, every hot-path call raises `RwhUnavailable` instead of silently
computing on the host.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "librwh_hip.so")

ABI_VERSION = 3          # RWH_ABI_VERSION of the include/rwh.h this binding was written against
RWH_U8, RWH_F32, RWH_F64 = 0, 1, 2
RWH_NEAREST, RWH_BILINEAR = 0, 1
RWH_LOSS = {"fwd": 0, "backward": 1, "reproj": 2}
RWH_WARP_ZERO_ORIGIN = 1
RWH_WARP_EXACT = 2
RWH_STITCH_FAST = 4
RWH_HYP_REPEATED, RWH_HYP_SINGULAR, RWH_HYP_ILLCOND, RWH_HYP_DEGENERATE = 1, 2, 4, 8
RWH_BATCH_DEVICE_SAMPLING = 1
RWH_BATCH_EARLY_STOP = 2
RWH_TUNE_WARP_SHAPE, RWH_TUNE_SCORE_HPW, RWH_TUNE_SCORE_EXACT, RWH_TUNE_WARP_FRAMES = 0, 1, 2, 3

# every symbol include/rwh.h declares (tests check the library exports them all)
EXPORTS = ("rwh_abi_version", "rwh_strerror", "rwh_lab_tune", "rwh_lab_clock_probe", "rwh_warp_backward", "rwh_warp_plan", "rwh_sample_points", "rwh_dlt4_batched",
           "rwh_score_count", "rwh_project_points", "rwh_project_points_ex", "rwh_ransac_search", "rwh_ransac_batched", "rwh_stitch_panorama",
           "rwh_host_dlt4_svd", "rwh_ransac_run", "rwh_ransac_run_layout", "rwh_warp_index_check", "rwh_score_count_inv", "rwh_host_inv3", "rwh_stitch_panorama_rows",
           "rwh_host_legacy_randint", "rwh_score_interval")


class RwhUnavailable(RuntimeError):
    """librwh_hip.so (or a GPU to run it on) is not available."""


class RwhError(RuntimeError):
    """A library call returned a negative RWH_E_* status."""


_lib = None


def _bind(lib):
    c = ctypes
    vp, i32, i64, f64, u32 = c.c_void_p, c.c_int, c.c_int64, c.c_double, c.c_uint
    lib.rwh_abi_version.restype = i32
    lib.rwh_abi_version.argtypes = []
    lib.rwh_strerror.restype = c.c_char_p
    lib.rwh_strerror.argtypes = [i32]
    lib.rwh_lab_tune.restype = i32
    lib.rwh_lab_tune.argtypes = [i32, i32]
    lib.rwh_lab_clock_probe.restype = i32
    lib.rwh_lab_clock_probe.argtypes = [vp, f64, vp]
    lib.rwh_warp_backward.restype = i32
    lib.rwh_warp_backward.argtypes = [vp, i32, i32, i32, i32, i64, i32,        # src, h, w, c, dtype, stride, batch
                                      c.POINTER(f64), i32,                    # inv_h, n_h
                                      f64, f64, f64, f64, f64, f64,           # x0 step_x x_last y0 step_y y_last
                                      i32, i32, i32, i32, i32,                # out_h out_w bound_h bound_w interp
                                      vp, i32, i64, i32, i32, u32, vp]        # dst dtype stride row_begin row_end flags stream
    lib.rwh_warp_plan.restype = i32
    lib.rwh_warp_plan.argtypes = [i32, i32, i32, i32, i32, c.POINTER(f64), i32, f64, f64, f64, f64, f64, f64,
                                  i32, i32, i32, i32, i32, i32, i32, i32, u32, c.c_char_p, i32]
    lib.rwh_warp_index_check.restype = i32
    lib.rwh_warp_index_check.argtypes = [i32, i32, c.POINTER(f64), f64, f64, f64, f64, f64, f64, i32, i32, i32, i32, i32, vp, vp]
    lib.rwh_sample_points.restype = i32
    lib.rwh_sample_points.argtypes = [vp, i32, i32, i32, i32, vp, vp, i64, i32, i32, i32, vp, i32, u32, vp]
    lib.rwh_dlt4_batched.restype = i32
    lib.rwh_dlt4_batched.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp]
    lib.rwh_score_count.restype = i32
    lib.rwh_score_count.argtypes = [vp, vp, vp, i32, i32, f64, i32, i32, i64, vp, vp, vp, vp, vp]
    lib.rwh_score_count_inv.restype = i32
    lib.rwh_score_count_inv.argtypes = [vp, vp, vp, vp, i32, i32, f64, i32, i32, i64, vp, vp, vp, vp, vp]
    lib.rwh_host_inv3.restype = i32
    lib.rwh_host_inv3.argtypes = [vp, i32, vp, vp]
    lib.rwh_ransac_search.restype = i32
    lib.rwh_ransac_search.argtypes = [vp, vp, i32, vp, i32, f64, i32, i32, i64, vp, vp, vp, vp, vp, i32, vp]
    lib.rwh_ransac_batched.restype = i32
    lib.rwh_ransac_batched.argtypes = [vp, vp, vp, i32, i32, i32, vp, c.c_uint64, i64, f64, i32, vp, vp, vp, vp, vp, vp, u32, vp]
    lib.rwh_stitch_panorama.restype = i32
    lib.rwh_stitch_panorama.argtypes = [vp, i32, i32, vp, i32, i32, c.POINTER(f64), i32, i32, i32, i32,
                                        i32, i32, i32, i32, i32, i32, i32, f64, vp, u32, vp]
    lib.rwh_stitch_panorama_rows.restype = i32
    lib.rwh_stitch_panorama_rows.argtypes = [vp, i32, i32, vp, i32, i32, c.POINTER(f64), i32, i32, i32, i32,
                                             i32, i32, i32, i32, i32, i32, i32, f64, vp, i32, i32, u32, vp]
    lib.rwh_project_points_ex.restype = i32
    lib.rwh_project_points_ex.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.rwh_project_points.restype = i32
    lib.rwh_project_points.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.rwh_host_dlt4_svd.restype = i32
    lib.rwh_host_dlt4_svd.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp]
    lib.rwh_score_interval.restype = i32
    lib.rwh_score_interval.argtypes = [vp, vp, i32, vp, vp, vp, i32, f64, f64, f64, f64, vp, vp, vp]
    lib.rwh_host_legacy_randint.restype = i32
    lib.rwh_host_legacy_randint.argtypes = [vp, vp, i64, i64, vp, vp]
    lib.rwh_ransac_run_layout.restype = i32
    lib.rwh_ransac_run_layout.argtypes = [i32, i32, vp, i32]
    lib.rwh_ransac_run.restype = i32
    lib.rwh_ransac_run.argtypes = [vp, vp, i32, vp, i32, f64, i32, i32, i32, vp, vp, i32, vp, vp, i64, vp, vp, vp, vp]
    return lib


def load():
    """Load (once) and return the bound library.  torch must be imported first
    so that librwh_hip.so resolves libamdhip64.so.7 to the HIP runtime torch
    already mapped (one runtime per process: streams and pointers are shared)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RwhUnavailable(
                "%s not found: build it with `make -C ransac_with_homography_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`)" % LIB_PATH)
        import torch  # noqa: F401  (maps the HIP runtime)
        try:
            _lib = _bind(ctypes.CDLL(LIB_PATH))
        except OSError as e:
            raise RwhUnavailable("cannot load %s: %s" % (LIB_PATH, e)) from e
        if _lib.rwh_abi_version() != ABI_VERSION:
            got = _lib.rwh_abi_version()
            _lib = None
            raise RwhUnavailable("librwh_hip.so reports ABI version %d, this package binds version %d: rebuild it "
                                 "(`make -C ransac_with_homography_amd/csrc`)" % (got, ABI_VERSION))
    return _lib


def require_gpu():
    """Return the torch device of the GPU the hot path runs on, or raise."""
    import torch
    if not torch.cuda.is_available():
        raise RwhUnavailable("no MI355X visible to this process: the RANSAC/warp hot path has no CPU fallback")
    load()
    return torch.device("cuda", torch.cuda.current_device())


def check(status, what):
    if status != 0:
        raise RwhError("%s failed: %s (%d)" % (what, load().rwh_strerror(status).decode(), status))


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
