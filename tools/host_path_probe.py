"""Drop-in (numpy in / numpy out) cost of one 4K transformImageH: where the milliseconds go."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import homography as hg
from ransac_with_homography_amd import homography as impl
H = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
img = np.random.default_rng(0).integers(0, 256, (2160, 3840, 3), dtype=np.uint8)
def t(f, n=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("transformImageH numpy->numpy (exact kernel): %.2f ms" % t(lambda: hg.transformImageH(img.copy(), H)))
impl.EXACT = False
print("transformImageH numpy->numpy (fast kernel):  %.2f ms" % t(lambda: hg.transformImageH(img.copy(), H)))
print("img.copy() alone: %.2f ms" % t(lambda: img.copy()))
d = torch.from_numpy(img).cuda()
print("upload pageable: %.2f ms" % t(lambda: torch.from_numpy(img).cuda()))
pin = torch.from_numpy(img).pin_memory()
print("upload pinned:   %.2f ms" % t(lambda: pin.cuda(non_blocking=True)))
print("download to pageable: %.2f ms" % t(lambda: d.cpu()))
hp = torch.empty_like(d, device="cpu").pin_memory()
print("download to pinned:   %.2f ms" % t(lambda: hp.copy_(d, non_blocking=True)))
print("tensor in -> tensor out (fast kernel): %.3f ms" % t(lambda: hg.transformImageH(d, H), 20))
