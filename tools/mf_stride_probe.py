"""Lab probe: is the multi-frame warp kernel's slow-down with many frames per block a property of WHERE the frames lie in memory?
The same launch with the frames' source and / or destination strides set to 0 (every "frame" = frame 0: all blocks stream
through one frame's bytes, the writes of the frames overwrite each other with identical data).
   MF=1,4,8 python tools/mf_stride_probe.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_with_homography_amd import _lib, kernels
from ransac_with_homography_amd import homography as hg
if os.environ.get("RWH_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["RWH_LIB"])
H_S = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
frames, W, Hh = 32, 3840, 2160
dev = _lib.require_gpu()
src = torch.randint(0, 256, (frames, Hh, W, 3), dtype=torch.uint8).to(dev)
mx, my, ow, oh = hg._bounds(Hh, W, H_S, 0)
g = kernels.Grid(mx, mx + ow - 1, ow, my, my + oh - 1, oh)
inv = np.ascontiguousarray(np.linalg.inv(H_S).reshape(9))
out = torch.empty((frames, oh, ow, 3), dtype=torch.uint8, device=dev)
lib = _lib.load()
def launch(ss, ds):
    st = lib.rwh_warp_backward(ctypes.c_void_p(src.data_ptr()), Hh, W, 3, _lib.RWH_U8, src.stride(0) * ss, frames,
                               inv.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 1, g.x0, g.step_x, g.x_last, g.y0, g.step_y, g.y_last,
                               g.out_h, g.out_w, Hh, W, _lib.RWH_BILINEAR, ctypes.c_void_p(out.data_ptr()), _lib.RWH_U8, oh * ow * 3 * ds,
                               0, oh, 0, _lib.stream_ptr())
    assert st == 0, st
for mf in [int(m) for m in os.environ.get("MF", "1,4,8").split(",")]:
    assert lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_FRAMES, mf) == 0
    for ss, ds in ((1, 1), (0, 1), (1, 0), (0, 0)):
        for _ in range(40): launch(ss, ds)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): launch(ss, ds)
        e1.record(); torch.cuda.synchronize()
        print("mf %2d  src stride %s  dst stride %s  %.4f ms per 32 frames" % (mf, "real" if ss else "0", "real" if ds else "0", e0.elapsed_time(e1) / 100), flush=True)
