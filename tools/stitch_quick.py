"""Developer probe: time rwh_stitch_panorama (fast / exact, paste / 'Rate') on the x8 foto1 canvases with HIP events."""
import os, sys, io, contextlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_with_homography_amd import _lib, kernels
from ransac_with_homography_amd import homography as hg
if os.environ.get("RWH_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["RWH_LIB"])      # a lab build
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = np.load(os.path.join(ROOT, "tests", "golden", "img_foto1.npz")); z = np.load(os.path.join(ROOT, "tests", "golden", "g8_stitch.npz"))
dev = _lib.require_gpu()
A8 = torch.from_numpy(np.ascontiguousarray(np.repeat(np.repeat(f["A"], 8, axis=0), 8, axis=1))).to(dev)
B8 = torch.from_numpy(np.ascontiguousarray(np.repeat(np.repeat(f["B"], 8, axis=0), 8, axis=1))).to(dev)
S = np.diag([8.0, 8.0, 1.0]); H8 = S @ z["H_g5"] @ np.linalg.inv(S)
h, w, _ = A8.shape
mx, my, wt, ht = hg._bounds(h, w, H8, 0)
(tsx, tsy, tex, tey), (qsx, qsy, qex, qey), (fw, fh) = hg._stitch_geometry(wt, ht, B8.shape[1], B8.shape[0], mx, my)
inv = np.linalg.inv(H8)
by = 3 * (A8.shape[0] * A8.shape[1] + B8.shape[0] * B8.shape[1] + fh * fw)
for fast in (True, False):
    for blend in (False, True):
        run = lambda: kernels.stitch_panorama(A8, B8, inv, (mx, my), (wt, ht), (tsx, tsy), (qsx, qsy), (fh, fw), blend, 0.2, zero_origin=False, fast=fast)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%s %-5s %.4f ms  canvas %dx%d  %.0f GB/s = %.3f of 8 TB/s (imgT + imgQ read once, canvas written once)" %
              ("fast " if fast else "exact", "rate" if blend else "paste", ms, fw, fh, by / ms / 1e6, by / ms / 1e6 / 8000), flush=True)
