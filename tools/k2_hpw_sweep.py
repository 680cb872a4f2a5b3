"""Developer probe: hypotheses per wave (rwh_lab_tune RWH_TUNE_SCORE_HPW) vs search time, K = 100 000 single and 64 x 10 000 batched."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_with_homography_amd import _lib, kernels
dev = _lib.require_gpu()
lib = _lib.load()
z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "matchespoints.npz"))
A, B = z["ptsA"].astype(np.float32), z["ptsB"].astype(np.float32)
pa, pb = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
M = A.shape[0]
need = kernels.need_count(M, 70, 4)
P = 64
offs = torch.arange(0, M * (P + 1), M, dtype=torch.int32, device=dev)
pa_b, pb_b = pa.repeat(P, 1), pb.repeat(P, 1)
needs = torch.full((P,), need, dtype=torch.int32, device=dev)
for hpw in (0, 1, 2, 3, 4, 7, 14, 21, 28):
    assert lib.rwh_lab_tune(_lib.RWH_TUNE_SCORE_HPW, hpw) == 0
    out = []
    for K in (10000, 100000):
        np.random.seed(0)
        idx = torch.from_numpy(np.random.randint(0, M, (K, 4)).astype(np.int32)).to(dev)
        ws = kernels.SearchWorkspace(K, M, dev, want_masks=False)
        for _ in range(5): kernels.ransac_search(pa, pb, idx, 5.0, "fwd", need, ws)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): kernels.ransac_search(pa, pb, idx, 5.0, "fwd", need, ws)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 30 * 1e6)
    bws = kernels.BatchWorkspace(P, 10000, M, dev, want_masks=False)
    for _ in range(3): kernels.ransac_batched(pa_b, pb_b, offs, needs, 5.0, "fwd", bws, seed=2024)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): kernels.ransac_batched(pa_b, pb_b, offs, needs, 5.0, "fwd", bws, seed=2024)
    torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 10 * 1e6)
    print("hpw %2d   K=10000 %.1f us   K=100000 %.1f us   batched 64x10000 %.1f us" % (hpw, *out), flush=True)
lib.rwh_lab_tune(_lib.RWH_TUNE_SCORE_HPW, 0)
