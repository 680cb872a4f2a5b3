"""Where does the time of one search go?  K2 alone vs K1+K2, single vs batched entry point (events, 20 iterations)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ransac_with_homography_amd import kernels
dev = torch.device("cuda")
z = np.load(os.path.join(ROOT, "tests", "golden", "matchespoints.npz"))
pa, pb = torch.from_numpy(z["ptsA"]).to(dev), torch.from_numpy(z["ptsB"]).to(dev)
K = 100000
np.random.seed(0)
idx = torch.from_numpy(np.random.randint(0, 185, (K, 4)).astype(np.int32)).to(dev)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ws = kernels.SearchWorkspace(K, 185, dev, want_masks=False)
H, _ = kernels.dlt4_batched(pa, pb, idx)
best = kernels.new_best(dev)
print("K2 only, single entry: %.1f us" % timeit(lambda: kernels.score_count(H, pa, pb, 5.0, "fwd", 134, best, want_masks=False)))
print("K1 only: %.1f us" % timeit(lambda: kernels.dlt4_batched(pa, pb, idx)))
print("K1+K2 single entry (rwh_ransac_search): %.1f us" % timeit(lambda: kernels.ransac_search(pa, pb, idx, 5.0, "fwd", 134, ws)))
offs = torch.tensor([0, 185], dtype=torch.int32, device=dev)
needs = torch.tensor([134], dtype=torch.int32, device=dev)
bws = kernels.BatchWorkspace(1, K, 185, dev, want_masks=False)
idx3 = idx.view(1, K, 4)
print("batched entry, P=1, caller idx: %.1f us" % timeit(lambda: kernels.ransac_batched(pa, pb, offs, needs, 5.0, "fwd", bws, idx=idx3)))
print("batched entry, P=1, device sampling: %.1f us" % timeit(lambda: kernels.ransac_batched(pa, pb, offs, needs, 5.0, "fwd", bws, seed=5)))
H2 = H.clone()
def rewrite_then_score():
    H.copy_(H2)                      # H rewritten by another kernel right before the scorer, like K1 does
    kernels.score_count(H, pa, pb, 5.0, "fwd", 134, best, want_masks=False)
print("copy(H) + K2: %.1f us;  copy(H) alone: %.1f us" % (timeit(rewrite_then_score), timeit(lambda: H.copy_(H2))))
cnt = torch.empty(K, dtype=torch.int32, device=dev)
def unrelated_then_score():
    cnt.zero_()                      # an unrelated small kernel in between
    kernels.score_count(H, pa, pb, 5.0, "fwd", 134, best, want_masks=False)
print("zero(counts) + K2: %.1f us" % timeit(unrelated_then_score))
def reset_then_score():
    best.zero_()
    kernels.score_count(H, pa, pb, 5.0, "fwd", 134, best, want_masks=False)
print("zero(best) + K2: %.1f us" % timeit(reset_then_score))
