"""Developer probe: the generic warp kernel (RGBA / float32 sources) on 4K frames, per configuration."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_with_homography_amd import _lib, kernels
from ransac_with_homography_amd import homography as hg
H_S = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
W, Hh, frames = 3840, 2160, 8
dev = _lib.require_gpu()
mx, my, ow, oh = hg._bounds(Hh, W, H_S, 0)
grid = kernels.Grid(mx, mx + ow - 1, ow, my, my + oh - 1, oh)
inv = np.linalg.inv(H_S)
for C, sdt, interp, odt in ((4, torch.uint8, "bilinear", torch.uint8), (4, torch.uint8, "nn", torch.uint8), (4, torch.float32, "bilinear", torch.float32),
                            (3, torch.float32, "bilinear", torch.float32), (4, torch.float32, "nn", torch.float32), (3, torch.uint8, "bilinear", torch.uint8)):
    src = (torch.rand((frames, Hh, W, C), device=dev) * 255).to(sdt)
    out = torch.empty((frames, oh, ow, C), dtype=odt, device=dev)
    run = lambda: kernels.warp_backward(src, inv, grid, (Hh, W), interp, odt, zero_origin=False, out=out)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    by = frames * (Hh * W * C * src.element_size() + oh * ow * C * out.element_size())
    print("%-44s C=%d %-8s -> %-8s %-8s %.3f ms / %d frames = %.1f us/frame  %.0f GB/s = %.3f of 8 TB/s" %
          (kernels.warp_plan((frames, Hh, W, C), sdt, inv, grid, (Hh, W), interp, odt), C, str(sdt)[6:], str(odt)[6:], interp, ms, frames, ms * 1e3 / frames,
           by / ms / 1e6, by / ms / 1e6 / 8000), flush=True)
    del src, out
