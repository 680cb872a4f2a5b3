"""Developer soak test (not part of the suite): stitchPanorama's fast compositor against the exact float64 compositor on
random image sizes / homographies / blend modes, and the one-homography-per-image launches against single-image launches.
   python tools/soak_stitch.py [cases] [seed]"""
import contextlib, io, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_with_homography_amd import _lib, kernels
from ransac_with_homography_amd import homography as hg
dev = _lib.require_gpu()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0
for case in range(cases):
    th, tw = int(rng.integers(140, 900)), int(rng.integers(140, 1300))
    qh, qw = int(rng.integers(140, 900)), int(rng.integers(140, 1300))
    T = torch.randint(0, 256, (th, tw, 3), dtype=torch.uint8, device=dev)
    Q = torch.randint(0, 256, (qh, qw, 3), dtype=torch.uint8, device=dev)
    t = rng.uniform(-0.5, 0.5)
    s = rng.uniform(0.8, 1.25)
    H = np.array([[s * np.cos(t), -s * np.sin(t), rng.uniform(-300, 600)], [s * np.sin(t), s * np.cos(t), rng.uniform(-300, 400)],
                  [rng.uniform(-1e-4, 1e-4), rng.uniform(-1e-4, 1e-4), 1.0]])
    blending = [False, "Rate"][case % 2]
    res = {}
    try:
        for exact in (True, False):
            hg.EXACT = exact
            with contextlib.redirect_stdout(io.StringIO()):
                res[exact] = hg.stitchPanorama(Q, T.clone(), H, blending=blending, blendrate=float(rng.uniform(0.05, 0.9)) if exact else res["rate"])
            if exact: res["rate"] = 0.2
    except Exception as e:      # same rate for both modes: redo simply
        pass
    rate = float(rng.uniform(0.05, 0.9))
    out = {}
    for exact in (True, False):
        hg.EXACT = exact
        with contextlib.redirect_stdout(io.StringIO()):
            out[exact] = hg.stitchPanorama(Q, T.clone(), H, blending=blending, blendrate=rate)
    hg.EXACT = None
    d = (out[True].to(torch.int16) - out[False].to(torch.int16)).abs()
    big = int((d > 1).sum())
    worst = max(worst, big)
    if big or case % 25 == 0:
        print("case %3d T %dx%d Q %dx%d canvas %s blend %s: %d px > 1 LSB, %.4f differ%s" % (case, tw, th, qw, qh, tuple(out[True].shape[:2]), blending, big,
              float((d != 0).float().mean()), "   <-- LOOK" if big else ""), flush=True)
    # one homography per image vs single launches (bit-identical)
    if case % 5 == 0:
        B = 3
        imgs = torch.randint(0, 256, (B, th, tw, 3), dtype=torch.uint8, device=dev)
        invs = np.stack([np.linalg.inv(H @ np.array([[1, 0, 10.0 * i], [0, 1, -7.0 * i], [0, 0, 1]])) for i in range(B)])
        grid = kernels.Grid(-50.0, -50.0 + 899, 900, -40.0, -40.0 + 499, 500)
        for interp, dt in (("bilinear", torch.uint8), ("nn", torch.uint8), ("bilinear", torch.float32)):
            per = kernels.warp_backward(imgs, invs, grid, (th, tw), interp, dt, zero_origin=False)
            for i in range(B):
                one = kernels.warp_backward(imgs[i].contiguous(), invs[i], grid, (th, tw), interp, dt, zero_origin=False)
                if not torch.equal(per[i], one):
                    print("case %d: per-image launch differs from single launch (%s %s image %d)   <-- LOOK" % (case, interp, dt, i), flush=True)
print("done: %d cases; worst count of compositor pixels beyond 1 LSB: %d" % (cases, worst))
