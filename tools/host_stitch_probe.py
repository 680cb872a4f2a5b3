"""Developer probe: stitchPanorama / transformImageH from HOST numpy arrays (the drop-in API), wall clock per call."""
import os, sys, time, io, contextlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import homography as hg
from ransac_with_homography_amd import _xfer
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = np.load(os.path.join(ROOT, "tests", "golden", "img_foto1.npz"))
A8 = np.ascontiguousarray(np.repeat(np.repeat(f["A"], 8, axis=0), 8, axis=1))
B8 = np.ascontiguousarray(np.repeat(np.repeat(f["B"], 8, axis=0), 8, axis=1))
H8 = np.load(os.path.join(ROOT, "tests", "golden", "g13_config4_x8.npz"))["H"]
dev = torch.device("cuda", 0)
def t(fn, n=5):
    fn(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    return (time.perf_counter() - t0) / n * 1e3, r
ms, d = t(lambda: _xfer.to_device(A8, dev)); torch.cuda.synchronize(); print("upload 134 MB: %.2f ms = %.1f GB/s" % (ms, A8.nbytes / ms / 1e6))
ms2, _ = t(lambda: (torch.from_numpy(A8).to(dev), torch.cuda.synchronize())); print("   torch .to(): %.2f ms" % ms2)
big = torch.empty((6313, 13181, 3), dtype=torch.uint8, device=dev).random_(0, 255)
ms, h = t(lambda: _xfer.to_host(big)); print("download 250 MB: %.2f ms = %.1f GB/s" % (ms, big.numel() / ms / 1e6)); assert np.array_equal(h, big.cpu().numpy())
ms2, _ = t(lambda: big.cpu().numpy()); print("   torch .cpu(): %.2f ms" % ms2)
from ransac_with_homography_amd import kernels
from ransac_with_homography_amd import homography as pk
def staged():
    t0 = time.perf_counter(); td = _xfer.to_device(A8, dev); qd = _xfer.to_device(B8, dev); torch.cuda.synchronize(); t1 = time.perf_counter()
    h, w, _ = A8.shape; mx, my, wt, ht = pk._bounds(h, w, H8, 0)
    (tsx, tsy, tex, tey), (qsx, qsy, qex, qey), (fw, fh) = pk._stitch_geometry(wt, ht, B8.shape[1], B8.shape[0], mx, my)
    out = kernels.stitch_panorama(td, qd, np.linalg.inv(H8), (mx, my), (wt, ht), (tsx, tsy), (qsx, qsy), (fh, fw), 0, 0.2, zero_origin=True, fast=False)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    r = _xfer.to_host(out); t3 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3
staged(); staged()
print("stages (ms): upload 2 x 134 MB %.2f, kernel + canvas allocation %.2f, download 250 MB %.2f" % staged())
with contextlib.redirect_stdout(io.StringIO()):
    ms_nc, out = t(lambda: hg.stitchPanorama(B8, A8, H8), 5)
    ms, out = t(lambda: hg.stitchPanorama(B8, A8.copy(), H8), 3)
print("stitchPanorama from host arrays without the caller's copy: %.1f ms per call" % ms_nc)
print("stitchPanorama from host arrays (exact kernel, canvas %s): %.1f ms per call (incl. one 134 MB A8.copy() = ~%.0f ms)" % (out.shape, ms, t(lambda: A8.copy(), 3)[0]))
img = np.random.default_rng(1).integers(0, 256, (2160, 3840, 3), dtype=np.uint8)
Hs = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
ms, r = t(lambda: hg.transformImageH(img, Hs), 10); print("transformImageH 4K numpy -> numpy (exact kernel): %.2f ms" % ms)
from ransac_with_homography_amd import homography as pk; pk.EXACT = False
ms, r = t(lambda: hg.transformImageH(img, Hs), 10); print("transformImageH 4K numpy -> numpy (fast kernel): %.2f ms" % ms)
