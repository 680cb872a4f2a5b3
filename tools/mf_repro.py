"""Developer repro: replay chosen cases of tools/soak_mf.py (same seed stream) and show WHERE the multi-frame kernel's output differs from the one-frame kernel's."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_with_homography_amd import _lib, kernels
dev = _lib.require_gpu()
lib = _lib.load()
seed, want = int(sys.argv[1]), set(int(v) for v in sys.argv[2].split(","))
rng = np.random.default_rng(seed)
for case in range(max(want) + 1):
    sh, sw = int(rng.integers(40, 900)), int(rng.integers(140, 1500))
    nb = int(rng.integers(2, 12))
    img = torch.randint(0, 256, (nb, sh, sw, 3), dtype=torch.uint8, device=dev)
    t = rng.uniform(-np.pi, np.pi) if case % 4 == 0 else rng.uniform(-0.08, 0.08)
    sx, sy = rng.uniform(0.6, 1.6, 2) if case % 5 == 0 else rng.uniform(0.9, 1.15, 2)
    A = np.array([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]]) @ np.array([[sx, rng.uniform(-0.1, 0.1)], [0, sy]])
    H = np.eye(3); H[:2, :2] = A
    H[:2, 2] = rng.uniform(-60, 60, 2) + np.array([sw / 2, sh / 2]) - A @ np.array([sw / 2, sh / 2])
    H[2, :2] = rng.uniform(-2e-4, 2e-4, 2) if case % 7 else rng.uniform(-2e-3, 2e-3, 2)
    inv = np.linalg.inv(H)
    ow, oh = int(rng.integers(128, 1900)), int(rng.integers(5, 1100))
    x0, y0 = rng.uniform(-120, 60, 2)
    stepx, stepy = rng.uniform(0.85, 1.2, 2)
    grid = kernels.Grid(x0, x0 + stepx * (ow - 1), ow, y0, y0 + stepy * (oh - 1), oh)
    bound = (sh, sw) if case % 4 else (int(rng.integers(sh // 2, sh + 1)), int(rng.integers(sw // 2, sw + 1)))
    shape = int(rng.choice([0, 0, 5, 6, 7]))
    rows = None if case % 3 else tuple(sorted(int(v) for v in rng.integers(0, oh + 1, 2)))
    if rows is not None and rows[0] == rows[1]: rows = None
    ns = (int(rng.integers(2, 6)), 100 + int(rng.integers(2, 6)))
    if case not in want: continue
    lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_SHAPE, shape); lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_FRAMES, 1)
    ref = kernels.warp_backward(img, inv, grid, bound, "bilinear", torch.uint8, zero_origin=False, rows=rows)
    n = ns[0]
    lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_FRAMES, n)
    print("case %d: src %dx%d x %d, out %dx%d, shape %d, rows %s, bound %s, n %d, plan %s" % (case, sw, sh, nb, ow, oh, shape, rows, bound, n,
          kernels.warp_plan(tuple(img.shape), torch.uint8, inv, grid, bound, "bilinear", torch.uint8)))
    for rep in range(4):
        got = kernels.warp_backward(img, inv, grid, bound, "bilinear", torch.uint8, zero_origin=False, rows=rows)
        d = (got != ref).any(dim=3)
        f, y, x = torch.nonzero(d, as_tuple=True)
        if len(f) == 0: print("   rep %d: identical" % rep); continue
        print("   rep %d: %d pixels differ; frames %s; rows %d..%d (mod 16: %s); cols %d..%d (tile %s, col in tile %s)" % (
            rep, len(f), sorted(set(f.tolist())), int(y.min()), int(y.max()), sorted(set((y % 16).tolist()))[:16], int(x.min()), int(x.max()),
            sorted(set((x // 128).tolist()))[:8], sorted(set((x % 128).tolist()))[:12]))
        if rep == 0:
            for i in range(min(len(f), 6)):
                print("      frame %d row %d col %d: got %s ref %s" % (int(f[i]), int(y[i]), int(x[i]), got[f[i], y[i], x[i]].tolist(), ref[f[i], y[i], x[i]].tolist()))
