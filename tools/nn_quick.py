"""Developer probe: the nearest-neighbour warp kernel on the bench geometry, 4K / 1080p / 8K.   RWH_LIB=<another build of the library>"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_with_homography_amd import _lib, kernels
from ransac_with_homography_amd import homography as hg
H_S = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
if os.environ.get("RWH_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["RWH_LIB"])
dev = _lib.require_gpu()
for (W, Hh, F) in ((3840, 2160, 16), (1920, 1080, 64), (7680, 4320, 4)):
    g = torch.Generator(device=dev).manual_seed(3)
    src = torch.randint(0, 256, (F, Hh, W, 3), dtype=torch.uint8, device=dev, generator=g)
    mx, my, ow, oh = hg._bounds(Hh, W, H_S, 0)
    grid = kernels.Grid(mx, mx + ow - 1, ow, my, my + oh - 1, oh); inv = np.linalg.inv(H_S)
    out = torch.empty((F, oh, ow, 3), dtype=torch.uint8, device=dev)
    for _ in range(60): kernels.warp_backward(src, inv, grid, (Hh, W), "nn", torch.uint8, zero_origin=False, out=out)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(40): kernels.warp_backward(src, inv, grid, (Hh, W), "nn", torch.uint8, zero_origin=False, out=out)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 40
    by = F * 3 * (Hh * W + oh * ow)
    print("nn %dx%d x%d: %.4f ms  %.0f GB/s = %.3f of 8 TB/s   checksum %d" % (W, Hh, F, ms, by / ms / 1e6, by / ms / 1e6 / 8000, int(out.sum(dtype=torch.int64))), flush=True)
