"""Developer soak test (not part of the suite): homographies whose denominator W reaches ~0 right next to a patch corner --
the horizon line passes within 1e-5 .. 0.5 px of a corner of the 8 px kernel's patch lattice, on either side -- so that
source coordinates run from valid values to +-1e5 .. 1e9 inside one patch.  Fast kernels (bilinear float32 / uint8, nearest)
against the exact float64 kernel, every patch shape; every fifth case with the nine entries scaled by 1e-280 .. 1e280, every fiftieth
with a NaN / Inf entry.   python tools/soak_horizon.py [cases] [seed]      TRACE=<file>: the case about to run is written there"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_with_homography_amd import _lib, kernels
dev = _lib.require_gpu()
lib = _lib.load()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad_cases = 0
for case in range(cases):
    sh, sw = int(rng.integers(40, 600)), int(rng.integers(40, 900))
    img = torch.randint(1, 256, (sh, sw, 3), dtype=torch.uint8, device=dev)
    ow, oh = int(rng.integers(128, 900)), int(rng.integers(16, 300))
    shape = int(rng.choice([0, 5, 6, 7, 13, 14]))
    pw = {0: 64, 5: 32, 6: 64, 7: 128, 13: 32, 14: 64}[shape]
    ph = 512 // pw
    # a lattice corner (column, row) of the patch grid and a line through a point delta away from it
    cc = float(rng.integers(0, ow // pw + 1) * pw - rng.integers(0, 2))
    rr = float(rng.integers(0, oh // ph + 1) * ph - rng.integers(0, 2))
    delta = 10.0 ** rng.uniform(-5, -0.3) * rng.choice([-1, 1])
    th = rng.uniform(0, 2 * np.pi)
    n = np.array([np.cos(th), np.sin(th)])                     # W = g * (n . (p - p0)),  p0 = corner + delta * n
    g = 10.0 ** rng.uniform(-4, 0)
    p0 = np.array([cc, rr]) + delta * n
    w_row = np.array([g * n[0], g * n[1], -g * (n @ p0)])
    if rng.random() < 0.5: w_row = -w_row
    A = rng.uniform(-1.5, 1.5, (2, 3)); A[:, 2] = rng.uniform(-50, 50, 2) + np.array([sw / 2, sh / 2]) * abs(w_row[2])
    if case % 3 == 0:                                          # Y proportional to W: y stays in range while x explodes
        A[1] = w_row * rng.uniform(0, sh - 1)
    if case % 3 == 1:
        A[0] = w_row * rng.uniform(0, sw - 1)
    ih = np.vstack([A, w_row])
    if case % 5 == 4:                                          # the same map with all nine entries scaled: W far from 1 (2^-300 .. 2^300 is the
        ih = ih * 10.0 ** rng.uniform(-280, 280)               # batch inversion's range; outside it every pixel takes its own reciprocal)
    if case % 50 == 49:                                        # a non-finite entry: everything is masked, as in the reference
        ih = ih.copy(); ih[int(rng.integers(0, 3)), int(rng.integers(0, 3))] = [np.nan, np.inf, -np.inf][int(rng.integers(0, 3))]
        img[0, 0, :] = 0                                       # (a +-Inf denominator maps to texel (0,0), which the reference blanks: rwh.h)
    grid = kernels.Grid(0, ow - 1, ow, 0, oh - 1, oh)
    assert lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_SHAPE, shape) == 0
    if os.environ.get("TRACE"):        # TRACE=<file>: the case about to run (a GPU fault ends the process: the last line names the case)
        with open(os.environ["TRACE"], "w") as fh: fh.write("case %d src %dx%d out %dx%d shape %d ih %r\n" % (case, sw, sh, ow, oh, shape, ih.tolist()))
    ex = kernels.warp_backward(img, ih, grid, (sh, sw), "bilinear", torch.float64, zero_origin=False, exact=True)
    f32 = kernels.warp_backward(img, ih, grid, (sh, sw), "bilinear", torch.float32, zero_origin=False)
    u8 = kernels.warp_backward(img, ih, grid, (sh, sw), "bilinear", torch.uint8, zero_origin=False)
    nn_e = kernels.warp_backward(img, ih, grid, (sh, sw), "nn", torch.uint8, zero_origin=False, exact=True)
    nn_f = kernels.warp_backward(img, ih, grid, (sh, sw), "nn", torch.uint8, zero_origin=False)
    rel = (f32.double() - ex).abs() / ex.abs().clamp(min=1.0)
    nan = int(torch.isnan(f32).sum())
    bad = int((rel > 1e-4).sum())          # pixels on the mask edge band may differ (documented in rwh.h): a handful at most
    big = int(((u8.to(torch.int16) - ex.to(torch.uint8).to(torch.int16)).abs() > 1).sum())
    nn_bad = int((nn_e != nn_f).any(dim=2).sum())
    look = nan or bad > 8 or big > 8 or nn_bad
    bad_cases += bool(look)
    if look or case % 50 == 0:
        print("case %3d src %dx%d out %dx%d shape %2d corner (%g, %g) delta %+.2e g %.1e: f32 %d px > 1e-4 (%d NaN), u8 %d px > 1 LSB, nn %d differ, valid px %d%s"
              % (case, sw, sh, ow, oh, shape, cc, rr, delta, g, bad, nan, big, nn_bad, int((ex != 0).any(dim=2).sum()), "   <-- LOOK" if look else ""), flush=True)
lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_SHAPE, 0)
print("done: %d cases, %d to look at" % (cases, bad_cases))
