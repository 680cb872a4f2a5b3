"""Developer probe: a 3-4x cliff was seen when the output buffer lay 257 x 16 MB behind the source (tools/placement_pairs.py).  One pool; the source batch
at its start, the output at source + DELTA for a sweep of DELTA = k x 16 MB + d; float32 and uint8 output, 16 x 4K.   python tools/alias_probe.py"""
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
MB = 1 << 20
src_bytes = B * SH * SW * 3
pool = torch.empty(src_bytes + (300 * 16 + 64) * MB + B * oh * ow * 12, dtype=torch.uint8, device=dev)
base = pool.data_ptr()
pad = (-base) % (16 * MB)                       # the source starts on a multiple of 16 MB
src = pool[pad:pad + src_bytes].view(B, SH, SW, 3)
src.copy_(torch.randint(0, 256, (B, SH, SW, 3), dtype=torch.uint8, device=dev))
k0 = (src_bytes + 16 * MB - 1) // (16 * MB)     # first multiple of 16 MB past the source


def time_it(dst, dt, n=30):
    f = lambda: kernels.warp_backward(src, inv, grid, (SH, SW), "bilinear", dt, zero_origin=False, out=dst)
    for _ in range(15): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for dt, esz in ((torch.float32, 4), (torch.uint8, 1)):
    nb = B * oh * ow * 3 * esz
    for _ in range(20): time_it(pool[pad + (k0 + 1) * 16 * MB + 4096:][:nb].view(dt).reshape(B, oh, ow, 3), dt, 10)
    print("%s output, ms per %d frames; source at 0x%x (a multiple of 16 MB)" % (str(dt), B, src.data_ptr()))
    for k in (k0, k0 + 1, k0 + 7, 64, 257, 300):
        if k < k0: continue
        row = []
        for d in (-2 * MB, -65536, -4096, -256, 0, 256, 4096, 65536, MB, 2 * MB, 4 * MB, 8 * MB):
            off = pad + k * 16 * MB + d
            if off < pad + src_bytes: row.append("%9s" % "-"); continue
            dst = pool[off:off + nb].view(dt).reshape(B, oh, ow, 3)
            row.append("%9.3f" % time_it(dst, dt))
        print("  k = %3d x 16 MB, d = -2M -64K -4K -256 0 +256 +4K +64K +1M +2M +4M +8M: %s" % (k, " ".join(row)), flush=True)
