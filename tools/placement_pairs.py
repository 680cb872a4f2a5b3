"""Developer probe: the float32-output warp is 8-10 % faster on some allocations than on others (profiles/r04_lab_notes.txt section 9).  Which buffer carries
the mode?  NS source batches x ND output buffers, every pair timed in one process.   python tools/placement_pairs.py"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ransac_with_homography_amd import _lib, kernels
from ransac_with_homography_amd.homography import _bounds
dev = _lib.require_gpu()
torch.manual_seed(1)
H = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
B, SH, SW = 16, 2160, 3840
mx, my, ow, oh = _bounds(SH, SW, H, 0)
grid = kernels.Grid(mx, mx + ow - 1, ow, my, my + oh - 1, oh)
inv = np.linalg.inv(H)
NS, ND = int(os.environ.get("NS", "4")), int(os.environ.get("ND", "4"))
junk = []
srcs, dsts = [], []
for i in range(NS):
    srcs.append(torch.randint(0, 256, (B, SH, SW, 3), dtype=torch.uint8, device=dev))
    junk.append(torch.empty((37 + 61 * i) << 20, dtype=torch.uint8, device=dev))       # shift the next allocation
for i in range(ND):
    dsts.append(torch.empty((B, oh, ow, 3), dtype=torch.float32, device=dev))
    junk.append(torch.empty((53 + 29 * i) << 20, dtype=torch.uint8, device=dev))


def time_it(src, dst, n=40):
    f = lambda: kernels.warp_backward(src, inv, grid, (SH, SW), "bilinear", torch.float32, zero_origin=False, out=dst)
    for _ in range(25): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for _ in range(30): time_it(srcs[0], dsts[0], 10)        # clock ramp
print("dst:            " + "  ".join("0x%x" % d.data_ptr() for d in dsts))
for s in srcs:
    print("src 0x%x: " % s.data_ptr() + "  ".join("%14.3f" % time_it(s, d) for d in dsts), flush=True)
