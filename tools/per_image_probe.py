"""Batch of 1080p frames: one homography for all vs one per image (coefficient tables, 8 images per launch)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ransac_with_homography_amd import kernels
dev = torch.device("cuda")
B = 64
src = torch.randint(0, 256, (B, 1080, 1920, 3), dtype=torch.uint8, device=dev)
dst = torch.empty_like(src)
grid = kernels.Grid(0, 1919, 1920, 0, 1079, 1080)
H = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
inv = np.linalg.inv(H)
rng = np.random.default_rng(0)
invs = np.stack([np.linalg.inv(H @ np.array([[1, 0, rng.uniform(-5, 5)], [0, 1, rng.uniform(-5, 5)], [0, 0, 1.0]])) for _ in range(B)])
def t(f, n=30):
    for _ in range(100): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
a = t(lambda: kernels.warp_backward(src, inv, grid, (1080, 1920), "bilinear", torch.uint8, zero_origin=False, out=dst))
b = t(lambda: kernels.warp_backward(src, invs, grid, (1080, 1920), "bilinear", torch.uint8, zero_origin=False, out=dst))
def loop():
    for i in range(B):
        kernels.warp_backward(src[i], invs[i], grid, (1080, 1920), "bilinear", torch.uint8, zero_origin=False, out=dst[i])
c = t(loop, 5)
print("64 x 1080p: one H %.3f ms | one H per image (tables) %.3f ms | 64 single-image calls %.3f ms" % (a, b, c))
