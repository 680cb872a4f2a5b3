"""Developer soak test (not part of the suite): the PRODUCT kernels' batch paths against single-image launches -- a batch with one homography, a batch with one
homography per image (the coefficient-table launches, images grouped by patch shape), row shards of both -- for uint8 / float32 / nearest output: every image of
every batch launch must equal its own single-image launch BIT FOR BIT (the patch shape is a function of the homography and the whole grid only).
   python tools/soak_batch.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_with_homography_amd import _lib, kernels
dev = _lib.require_gpu()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0


def rand_h(sw, sh, case):
    t = rng.uniform(-np.pi, np.pi) if case % 4 == 0 else rng.uniform(-0.08, 0.08)
    sx, sy = rng.uniform(0.45, 1.6, 2) if case % 3 == 0 else rng.uniform(0.9, 1.15, 2)
    A = np.array([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]]) @ np.array([[sx, rng.uniform(-0.1, 0.1)], [0, sy]])
    H = np.eye(3); H[:2, :2] = A
    H[:2, 2] = rng.uniform(-60, 60, 2) + np.array([sw / 2, sh / 2]) - A @ np.array([sw / 2, sh / 2])
    H[2, :2] = rng.uniform(-2e-4, 2e-4, 2) if case % 7 else rng.uniform(-2e-3, 2e-3, 2)
    return np.linalg.inv(H)


for case in range(cases):
    sh, sw = int(rng.integers(40, 700)), int(rng.integers(140, 1200))
    nb = int(rng.integers(2, 13))
    img = torch.randint(0, 256, (nb, sh, sw, 3), dtype=torch.uint8, device=dev)
    ow, oh = int(rng.integers(100, 1700)), int(rng.integers(5, 900))
    x0, y0 = rng.uniform(-120, 60, 2)
    stepx, stepy = rng.uniform(0.85, 1.2, 2)
    grid = kernels.Grid(x0, x0 + stepx * (ow - 1), ow, y0, y0 + stepy * (oh - 1), oh)
    bound = (sh, sw) if case % 4 else (int(rng.integers(sh // 2, sh + 1)), int(rng.integers(sw // 2, sw + 1)))
    rows = None if case % 3 else tuple(sorted(int(v) for v in rng.integers(0, oh + 1, 2)))
    if rows is not None and rows[0] == rows[1]: rows = None
    one = rand_h(sw, sh, case)
    per = np.stack([rand_h(sw, sh, case + i) if i % 2 else one for i in range(nb)])      # a mix: equal and different homographies, several patch shapes
    msgs = []
    for interp, dt in (("bilinear", torch.uint8), ("bilinear", torch.float32), ("nn", torch.uint8)):
        for name, ih in (("one H", one), ("H per image", per)):
            got = kernels.warp_backward(img, ih, grid, bound, interp, dt, zero_origin=False, rows=rows)
            for i in range(nb):
                ref = kernels.warp_backward(img[i], ih if ih.ndim == 2 else ih[i], grid, bound, interp, dt, zero_origin=False, rows=rows)
                if not torch.equal(got[i], ref):
                    bad += 1
                    msgs.append("%s %s %s image %d: %d values differ" % (interp, str(dt).replace("torch.", ""), name, i, int((got[i] != ref).sum())))
                    break
    if msgs or case % 50 == 0:
        print("case %4d src %4dx%-4d x %2d out %4dx%-4d rows %s  %s" % (case, sw, sh, nb, ow, oh, rows, "; ".join(msgs) + ("   <-- LOOK" if msgs else "ok")), flush=True)
print("done: %d cases x 6 batch launches, %d mismatches" % (cases, bad))
