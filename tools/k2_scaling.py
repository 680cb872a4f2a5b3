"""Scorer scaling with the number of correspondences (SURVEY 8d scaling set: M = 1 024 and 16 384)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ransac_with_homography_amd import kernels
dev = torch.device("cuda")
Hs = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
for M in (185, 256, 1024, 16384):
    rng = np.random.default_rng(42)
    A = rng.uniform(0, 4000, (M, 2)); P = np.c_[A, np.ones(M)] @ Hs.T
    B = P[:, :2] / P[:, 2:] + rng.normal(0, 1.0, (M, 2)); out = rng.random(M) < 0.4
    B[out] = rng.uniform(0, 4000, (int(out.sum()), 2))
    pa, pb = torch.from_numpy(A.astype(np.float32)).to(dev), torch.from_numpy(B.astype(np.float32)).to(dev)
    for K in (10000, 100000):
        idx = torch.from_numpy(rng.integers(0, M, (K, 4)).astype(np.int32)).to(dev)
        ws = kernels.SearchWorkspace(K, M, dev, want_masks=False)
        f = lambda: kernels.ransac_search(pa, pb, idx, 3.0, "fwd", kernels.need_count(M, 70, 4), ws)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        w = kernels.decode_best(ws.best.cpu().numpy(), K)
        print("M=%5d K=%6d: %8.1f us per search  %.2e hyp/s  %.2e pairs/s  winner count %s" % (M, K, us, K / us * 1e6, K * M / us * 1e6, w[1]))
