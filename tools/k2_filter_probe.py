"""Developer probe: K2 with and without the reciprocal filter (rwh_lab_tune RWH_TUNE_SCORE_EXACT) -- identical counts /
masks / keys, and the time of each, on matchespoints (K = 100 000) and the batched P = 64 x K = 10 000 search."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_with_homography_amd import _lib, kernels
dev = _lib.require_gpu()
lib = _lib.load()
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "matchespoints.npz"))
A, B = z["ptsA"].astype(np.float32), z["ptsB"].astype(np.float32)
pa, pb = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
M = A.shape[0]
for method in ("fwd", "backward", "reproj"):
    for K in (10000, 100000):
        np.random.seed(0)
        idx = torch.from_numpy(np.random.randint(0, M, (K, 4)).astype(np.int32)).to(dev)
        need = kernels.need_count(M, 70, 4)
        res = {}
        for exact in (1, 0):
            assert lib.rwh_lab_tune(_lib.RWH_TUNE_SCORE_EXACT, exact) == 0
            ws = kernels.SearchWorkspace(K, M, dev)
            for _ in range(5): kernels.ransac_search(pa, pb, idx, 5.0, method, need, ws)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20): kernels.ransac_search(pa, pb, idx, 5.0, method, need, ws)
            torch.cuda.synchronize()
            res[exact] = (ws.counts.clone(), ws.masks.clone(), ws.best.clone(), (time.perf_counter() - t0) / 20 * 1e6)
        same = all(torch.equal(res[0][i], res[1][i]) for i in range(3))
        print("%-8s K=%6d  exact %.1f us  filter %.1f us  identical counts/masks/keys: %s" % (method, K, res[1][3], res[0][3], same), flush=True)
# batched
P, K = 64, 10000
offs = torch.arange(0, M * (P + 1), M, dtype=torch.int32, device=dev)
pa_b, pb_b = pa.repeat(P, 1), pb.repeat(P, 1)
needs = torch.full((P,), kernels.need_count(M, 70, 4), dtype=torch.int32, device=dev)
res = {}
for exact in (1, 0):
    assert lib.rwh_lab_tune(_lib.RWH_TUNE_SCORE_EXACT, exact) == 0
    bws = kernels.BatchWorkspace(P, K, M, dev, want_masks=False)
    for _ in range(3): kernels.ransac_batched(pa_b, pb_b, offs, needs, 5.0, "fwd", bws, seed=2024)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): kernels.ransac_batched(pa_b, pb_b, offs, needs, 5.0, "fwd", bws, seed=2024)
    torch.cuda.synchronize()
    res[exact] = (bws.counts.clone(), bws.best.clone(), (time.perf_counter() - t0) / 10 * 1e6)
print("batched P=%d K=%d  exact %.1f us  filter %.1f us  identical: %s" % (P, K, res[1][2], res[0][2],
      torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])))
lib.rwh_lab_tune(_lib.RWH_TUNE_SCORE_EXACT, 0)
