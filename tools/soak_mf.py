"""Developer soak test (not part of the suite): the multi-frame kernels (rwh_lab_tune RWH_TUNE_WARP_FRAMES: n = frames per block, 100 + n = one staging
window per block) against the one-frame kernel on random homographies / grids / batches / patch shapes / row shards: BIT-IDENTICAL or the case is printed.
   python tools/soak_mf.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_with_homography_amd import _lib, kernels
dev = _lib.require_gpu()
lib = _lib.load()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = taken = 0
for case in range(cases):
    sh, sw = int(rng.integers(40, 900)), int(rng.integers(140, 1500))
    nb = int(rng.integers(2, 12))
    img = torch.randint(0, 256, (nb, sh, sw, 3), dtype=torch.uint8, device=dev)
    t = rng.uniform(-np.pi, np.pi) if case % 4 == 0 else rng.uniform(-0.08, 0.08)
    sx, sy = rng.uniform(0.6, 1.6, 2) if case % 5 == 0 else rng.uniform(0.9, 1.15, 2)
    A = np.array([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]]) @ np.array([[sx, rng.uniform(-0.1, 0.1)], [0, sy]])
    H = np.eye(3); H[:2, :2] = A
    H[:2, 2] = rng.uniform(-60, 60, 2) + np.array([sw / 2, sh / 2]) - A @ np.array([sw / 2, sh / 2])
    H[2, :2] = rng.uniform(-2e-4, 2e-4, 2) if case % 7 else rng.uniform(-2e-3, 2e-3, 2)      # sometimes a horizon inside the grid
    inv = np.linalg.inv(H)
    ow, oh = int(rng.integers(128, 1900)), int(rng.integers(5, 1100))
    x0, y0 = rng.uniform(-120, 60, 2)
    stepx, stepy = rng.uniform(0.85, 1.2, 2)
    grid = kernels.Grid(x0, x0 + stepx * (ow - 1), ow, y0, y0 + stepy * (oh - 1), oh)
    bound = (sh, sw) if case % 4 else (int(rng.integers(sh // 2, sh + 1)), int(rng.integers(sw // 2, sw + 1)))
    shape = int(rng.choice([0, 0, 5, 6, 7]))
    rows = None if case % 3 else tuple(sorted(int(v) for v in rng.integers(0, oh + 1, 2)))
    if rows is not None and rows[0] == rows[1]: rows = None
    assert lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_SHAPE, shape) == 0
    assert lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_FRAMES, 1) == 0
    ref = kernels.warp_backward(img, inv, grid, bound, "bilinear", torch.uint8, zero_origin=False, rows=rows)
    msgs = []
    for n in (int(rng.integers(2, 6)), 100 + int(rng.integers(2, 6))):
        assert lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_FRAMES, n) == 0
        plan = kernels.warp_plan(tuple(img.shape), torch.uint8, inv, grid, bound, "bilinear", torch.uint8)
        taken += "fast8m" in plan
        got = kernels.warp_backward(img, inv, grid, bound, "bilinear", torch.uint8, zero_origin=False, rows=rows)
        if not torch.equal(got, ref):
            bad += 1
            msgs.append("n %d (%s): %d bytes differ" % (n, plan, int((got != ref).sum())))
    if msgs or case % 50 == 0:
        print("case %4d src %4dx%-4d x %2d out %4dx%-4d shape %d rot %+.2f rows %s  %s" % (case, sw, sh, nb, ow, oh, shape, t, rows, "; ".join(msgs) + ("   <-- LOOK" if msgs else "ok")), flush=True)
lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_SHAPE, 0); lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_FRAMES, 0)
print("done: %d cases x 2 kernels, %d mismatches; the multi-frame kernels were the plan in %d of %d launches" % (cases, bad, taken, 2 * cases))
