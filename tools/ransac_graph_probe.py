"""Probe: RANSAC search latency per run -- eager launches vs a captured hipGraph (torch.cuda.CUDAGraph)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from ransac_with_homography_amd import kernels
dev = torch.device("cuda")
z = np.load("tests/golden/matchespoints.npz")
pa, pb = torch.from_numpy(z["ptsA"]).to(dev), torch.from_numpy(z["ptsB"]).to(dev)
for K in (10000, 100000):
    np.random.seed(0)
    idx = torch.from_numpy(np.random.randint(0, 185, (K, 4)).astype(np.int32)).to(dev)
    ws = kernels.SearchWorkspace(K, 185, dev, want_masks=False)
    host = torch.empty(2, dtype=torch.int64).pin_memory()
    def eager():
        kernels.ransac_search(pa, pb, idx, 5.0, "fwd", 134, ws)
        return ws.best.cpu()
    for _ in range(5): eager()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): b = eager()
    t_eager = (time.perf_counter() - t0) / 50
    # graph: search + async copy of the 16-byte result into pinned memory
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        kernels.ransac_search(pa, pb, idx, 5.0, "fwd", 134, ws)
        host.copy_(ws.best, non_blocking=True)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        kernels.ransac_search(pa, pb, idx, 5.0, "fwd", 134, ws)
        host.copy_(ws.best, non_blocking=True)
    for _ in range(5): g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay(); torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / 50
    print(K, "eager %.1f us/run (%.3g hyp/s)  graph %.1f us/run (%.3g hyp/s)" % (t_eager * 1e6, K / t_eager, t_graph * 1e6, K / t_graph),
          kernels.decode_best(b.numpy(), K), kernels.decode_best(host.numpy(), K))
