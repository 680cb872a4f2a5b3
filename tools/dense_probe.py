"""Developer probe: RANSAC.run on the dense-cloud recipe (sigma 3 px, M = 1400, K = 10 000) per host-thread count: every run's time and the settle statistics."""
import os, sys, time, io, contextlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ransac as rs
from ransac_with_homography_amd import ransac as impl
H_S = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
rng = np.random.default_rng(7)
M, sigma = 1400, float(os.environ.get("SIGMA", "3"))
G = rng.normal(500, sigma, (M, 2))
P = np.c_[G, np.ones(M)] @ H_S.T
Bp = P[:, :2] / P[:, 2:3] + rng.normal(0, 1.0, (M, 2))
o = rng.random(M) < 0.3
Bp[o] = rng.uniform(Bp.min(), Bp.max(), (int(o.sum()), 2))
Xd, Yd = G.astype(np.float32).T.copy(), Bp.astype(np.float32).T.copy()
for th in (16, 32, 16, 32, 8, 64):
    impl.HOST_THREADS = th
    ts = []
    for i in range(12):
        np.random.seed(0)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            r = rs.RANSAC(rs.HomoModel(th=5, d=95, n=4), k=10000)
            r.run([Xd, Yd], method="fwd")
        ts.append((time.perf_counter() - t0) * 1e3)
    lr = r.last_run
    print("threads %2d: ms %s | host-solved %d rounds %d flagged %d intervals %d" % (th, " ".join("%.2f" % t for t in ts), lr["host_settled"], lr["host_rounds"], lr["flagged"], lr["intervals"]))
