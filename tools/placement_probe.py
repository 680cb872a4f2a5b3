"""Developer probe: does the warp's speed depend on WHERE its buffers lie?  One process, one source batch; the output is placed at a sweep of byte
offsets inside one pool (and the pool itself is re-allocated a few times), float32 and uint8 output, ms per launch per placement.
   python tools/placement_probe.py            FRAMES=16"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ransac_with_homography_amd import _lib, kernels
from ransac_with_homography_amd.homography import _bounds
dev = _lib.require_gpu()
torch.manual_seed(1)
H = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
B, SH, SW = int(os.environ.get("FRAMES", "16")), 2160, 3840
mx, my, ow, oh = _bounds(SH, SW, H, 0)
grid = kernels.Grid(mx, mx + ow - 1, ow, my, my + oh - 1, oh)
inv = np.linalg.inv(H)
OFFS = [0, 256, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 8 << 20, 32 << 20, 100 << 20, (1 << 30) + 4096]
keep = []


def time_it(src, dst, dt, n=60):
    f = lambda: kernels.warp_backward(src, inv, grid, (SH, SW), "bilinear", dt, zero_origin=False, out=dst)
    for _ in range(40): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for trial in range(int(os.environ.get("TRIALS", "3"))):
    src = torch.randint(0, 256, (B, SH, SW, 3), dtype=torch.uint8, device=dev)
    for dt, esz in ((torch.float32, 4), (torch.uint8, 1)):
        nbytes = B * oh * ow * 3 * esz
        pool = torch.empty(nbytes + max(OFFS) + 4096, dtype=torch.uint8, device=dev)
        print("trial %d %s: src at 0x%x, pool at 0x%x (pool - src = %d MB + %d)" % (trial, str(dt), src.data_ptr(), pool.data_ptr(),
              (pool.data_ptr() - src.data_ptr()) >> 20, (pool.data_ptr() - src.data_ptr()) & ((1 << 20) - 1)))
        row = []
        for off in OFFS:
            dst = pool[off:off + nbytes].view(dt).reshape(B, oh, ow, 3)
            row.append("%d:%.3f" % (off, time_it(src, dst, dt)))
        print("   ms per %d frames by output offset:  " % B + "  ".join(row), flush=True)
        keep.append(pool)          # (kept: the next trial's pool lands somewhere else)
    keep.append(src)
    if len(keep) > 4: del keep[:3]
