"""Developer soak test (not part of the suite): the RANSAC scorer's filter kernels against the all-exact kernel
(rwh_lab_tune RWH_TUNE_SCORE_EXACT) on random problems -- counts, masks and packed keys must be identical.
   python tools/soak_ransac.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_with_homography_amd import _lib, kernels
dev = _lib.require_gpu()
lib = _lib.load()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(cases):
    M = int(rng.choice([5, 64, 65, 185, 256, 257, 700, 3000]))
    scale = float(rng.choice([50.0, 1200.0, 8000.0, 1e5]))
    Hs = np.array([[rng.uniform(0.7, 1.3), rng.uniform(-0.2, 0.2), rng.uniform(-50, 50)], [rng.uniform(-0.2, 0.2), rng.uniform(0.7, 1.3), rng.uniform(-50, 50)],
                   [rng.uniform(-1e-5, 1e-5), rng.uniform(-1e-5, 1e-5), 1.0]])
    A = rng.uniform(0, scale, (M, 2))
    P = np.c_[A, np.ones(M)] @ Hs.T
    B = P[:, :2] / P[:, 2:] + rng.normal(0, rng.uniform(0.1, 3.0), (M, 2))
    out = rng.random(M) < rng.uniform(0.1, 0.7)
    B[out] = rng.uniform(0, scale, (int(out.sum()), 2))
    A, B = A.astype(np.float32), B.astype(np.float32)
    pa, pb = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    K = int(rng.choice([7, 500, 3000]))
    idx = torch.from_numpy(rng.integers(0, M, (K, 4)).astype(np.int32)).to(dev)
    th = float(rng.choice([0.05, 1.0, 3.0, 5.0, 40.0]))
    need = kernels.need_count(M, int(rng.integers(30, 95)), 4)
    for method in ("fwd", "backward", "reproj"):
        res = {}
        for exact in (1, 0):
            assert lib.rwh_lab_tune(_lib.RWH_TUNE_SCORE_EXACT, exact) == 0
            ws = kernels.SearchWorkspace(K, M, dev)
            kernels.ransac_search(pa, pb, idx, th, method, need, ws)
            bws = kernels.BatchWorkspace(2, K, M, dev)
            kernels.ransac_batched(torch.cat([pa, pa]), torch.cat([pb, pb]), torch.tensor([0, M, 2 * M], dtype=torch.int32, device=dev),
                                   torch.tensor([need, need], dtype=torch.int32, device=dev), th, method, bws, idx=torch.stack([idx, idx]))
            res[exact] = [ws.counts.clone(), ws.masks.clone(), ws.best.clone(), bws.counts.clone(), bws.masks.clone(), bws.best.clone()]
        same = all(torch.equal(a, b) for a, b in zip(res[0], res[1])) and torch.equal(res[0][3][0], res[0][0]) and torch.equal(res[0][3][1], res[0][0])
        if not same:
            bad += 1
            print("case %d M=%d K=%d th=%g scale=%g %s: MISMATCH" % (case, M, K, th, scale, method), flush=True)
lib.rwh_lab_tune(_lib.RWH_TUNE_SCORE_EXACT, 0)
print("done: %d cases x 3 losses, %d mismatches" % (cases, bad))
