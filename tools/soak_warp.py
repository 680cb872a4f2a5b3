"""Developer soak test (not part of the suite): the fast warp kernels against the exact float64 kernel (itself pinned
bit for bit to the reference by the golden tests) on many random homographies, sizes, grids and patch shapes.
   python tools/soak_warp.py [cases] [seed]        CH=4: uint8 RGBA images (the staged RGBA kernel / the generic kernel)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_with_homography_amd import _lib, kernels
dev = _lib.require_gpu()
lib = _lib.load()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
CH = int(os.environ.get("CH", "3"))
F32 = os.environ.get("SRC", "u8") == "f32"      # SRC=f32: float32 source images (the generic kernel, one pixel per lane)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = [0.0, 0, 0.0]
for case in range(cases):
    sh, sw = int(rng.integers(40, 1400)), int(rng.integers(40, 2000))
    img = torch.randint(0, 256, (sh, sw, CH), dtype=torch.uint8, device=dev)
    if F32: img = img.float() * 0.37 + 0.11
    t = rng.uniform(-np.pi, np.pi) if case % 3 == 0 else rng.uniform(-0.08, 0.08)
    sx, sy = rng.uniform(0.5, 2.2, 2) if case % 5 == 0 else rng.uniform(0.85, 1.2, 2)
    if os.environ.get("ZOOM"): sx, sy = 1.0 / rng.uniform(1.2, 2.4) * rng.uniform(0.95, 1.05, 2)      # ZOOM=1: minification (the HALVES kernel)
    A = np.array([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]]) @ np.array([[sx, rng.uniform(-0.15, 0.15)], [0, sy]])
    H = np.eye(3); H[:2, :2] = A
    H[:2, 2] = rng.uniform(-60, 60, 2) + np.array([sw / 2, sh / 2]) - A @ np.array([sw / 2, sh / 2])
    H[2, :2] = rng.uniform(-2e-4, 2e-4, 2) if case % 7 else rng.uniform(-2e-3, 2e-3, 2)      # sometimes a horizon inside the grid
    inv = np.linalg.inv(H)
    ow, oh = int(rng.integers(8, 2300)), int(rng.integers(5, 1500))
    x0, y0 = rng.uniform(-120, 60, 2)
    stepx, stepy = rng.uniform(0.8, 1.25, 2)
    grid = kernels.Grid(x0, x0 + stepx * (ow - 1), ow, y0, y0 + stepy * (oh - 1), oh)
    bound = (sh, sw) if case % 4 else (int(rng.integers(sh // 2, sh + 1)), int(rng.integers(sw // 2, sw + 1)))
    shape = int(rng.choice([0, 0, 5, 6, 7, 13, 14]))      # 13 / 14: staged by half patches (bilinear uint8 RGB; the other kernels then run 128 x 4)
    assert lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_SHAPE, shape) == 0
    ex = kernels.warp_backward(img, inv, grid, bound, "bilinear", torch.float64, zero_origin=False, exact=True)
    f32 = kernels.warp_backward(img, inv, grid, bound, "bilinear", torch.float32, zero_origin=False)
    u8 = kernels.warp_backward(img, inv, grid, bound, "bilinear", torch.uint8, zero_origin=False) if not F32 else ex.clamp(0, 255).to(torch.uint8)
    err = (f32.double() - ex).abs()
    rel = err / ex.abs().clamp(min=1.0)
    bad = int((rel > 1e-4).sum())                      # pixels on a mask edge band may differ (documented): count them
    d = (u8.to(torch.int16) - ex.to(torch.uint8).to(torch.int16)).abs()
    big = int((d > 1).sum())
    nn_e = kernels.warp_backward(img, inv, grid, bound, "nn", img.dtype, zero_origin=False, exact=True)
    nn_f = kernels.warp_backward(img, inv, grid, bound, "nn", img.dtype, zero_origin=False)
    nn_bad = int((nn_e != nn_f).any(dim=2).sum())
    worst = [max(worst[0], float(rel.max())), max(worst[1], big), max(worst[2], float((d != 0).float().mean()))]
    flag = "" if (bad <= 6 and big <= 6 and nn_bad == 0) else "   <-- LOOK"
    if big > 6:      # where are they?
        ys_, xs_ = torch.nonzero((d > 1).any(dim=2), as_tuple=True)
        flag += "  bad u8 pixels: rows %d..%d cols %d..%d (distinct rows %d, cols %d) max|d| %d  bound %s" % (
            int(ys_.min()), int(ys_.max()), int(xs_.min()), int(xs_.max()), len(torch.unique(ys_)), len(torch.unique(xs_)), int(d.max()), bound)
    if flag or case % 20 == 0:
        print("case %3d src %4dx%-4d out %4dx%-4d shape %d rot %+.2f  f32: %d px > 1e-4 (max rel %.2e)  u8: %d px > 1 LSB, %.4f differ  nn: %d differ%s"
              % (case, sw, sh, ow, oh, shape, t, bad, float(rel.max()), big, float((d != 0).float().mean()), nn_bad, flag), flush=True)
lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_SHAPE, 0)
print("done: %d cases; worst f32 rel %.2e, worst u8 >1 LSB count %d, worst u8 differing fraction %.4f" % (cases, *worst))
