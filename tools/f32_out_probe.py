"""Fast kernel with float32 output (wrapPerspective-level result, 15 algorithmic bytes per pixel) vs uint8 output."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ransac_with_homography_amd import _lib, kernels
if os.environ.get("RWH_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["RWH_LIB"])      # a lab build (tools/build_variant.sh)
from ransac_with_homography_amd.homography import _bounds
dev = torch.device("cuda")
H = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
torch.manual_seed(1)
B, SH, SW = int(os.environ.get("FRAMES", "16")), 2160, 3840
src = torch.randint(0, 256, (B, SH, SW, 3), dtype=torch.uint8, device=dev)
mx, my, ow, oh = _bounds(SH, SW, H, 0)
grid = kernels.Grid(mx, mx + ow - 1, ow, my, my + oh - 1, oh)
inv = np.linalg.inv(H)
for dt, esz in ((torch.uint8, 1), (torch.float32, 4)):
    dst = torch.empty((B, oh, ow, 3), dtype=dt, device=dev)
    f = lambda: kernels.warp_backward(src, inv, grid, (SH, SW), "bilinear", dt, zero_origin=False, out=dst)
    for _ in range(200): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    byt = B * (3 * SH * SW + 3 * oh * ow * esz)
    import hashlib
    print("sha1 %s  " % hashlib.sha1(dst.cpu().numpy().tobytes()).hexdigest()[:12], end="")
    print("%s out: %.3f ms per %d frames = %.2f us/frame, %.0f GB/s algorithmic = %.3f of 8 TB/s" % (str(dt), ms, B, ms * 1e3 / B, byt / ms / 1e6, byt / ms / 1e6 / 8000))
