"""Developer probe for PMC passes: the batched search (P = 64 x K = 10 000, fwd) a few times.  argv[1]: 1 = exact-only scorer."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_with_homography_amd import _lib, kernels
dev = _lib.require_gpu()
lib = _lib.load()
z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "matchespoints.npz"))
A, B = z["ptsA"].astype(np.float32), z["ptsB"].astype(np.float32)
pa, pb = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
M = A.shape[0]
assert lib.rwh_lab_tune(_lib.RWH_TUNE_SCORE_EXACT, int(sys.argv[1]) if len(sys.argv) > 1 else 0) == 0
P, K = 64, 10000
offs = torch.arange(0, M * (P + 1), M, dtype=torch.int32, device=dev)
pa_b, pb_b = pa.repeat(P, 1), pb.repeat(P, 1)
needs = torch.full((P,), kernels.need_count(M, 70, 4), dtype=torch.int32, device=dev)
bws = kernels.BatchWorkspace(P, K, M, dev, want_masks=False)
for _ in range(4): kernels.ransac_batched(pa_b, pb_b, offs, needs, 5.0, "fwd", bws, seed=2024)
torch.cuda.synchronize()
print("done", [kernels.decode_best(bws.best[p].cpu().numpy(), K)[1] for p in range(3)])
