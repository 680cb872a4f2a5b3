"""cProfile of RANSAC.run (host overhead of one search).   K=<hypotheses> (default 1500)   python tools/run_profile.py"""
import os, sys, time, numpy as np, cProfile, pstats, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ransac as rs
from ransac_with_homography_amd import ransac as _impl
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "matchespoints.npz"))
X, Y = z["ptsA"].T.copy(), z["ptsB"].T.copy()
K = int(os.environ.get("K", "1500"))
r = rs.RANSAC(rs.HomoModel(th=5 if K > 1500 else 4, d=70 if K > 1500 else 95, n=4), k=K)
def run():
    np.random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        return r.run([X, Y], method="fwd")
for _ in range(5): run()
t = time.perf_counter()
for _ in range(10): run()
print("K = %d: %.3f ms per run; host-solved %d, flagged by K1 %d, rounds %d; cpu_count %s, HOST_THREADS %d" %
      (K, (time.perf_counter() - t) * 100, r.last_run["host_settled"], r.last_run["flagged"], r.last_run["host_rounds"], os.cpu_count(), _impl.HOST_THREADS))
np.random.seed(0); t = time.perf_counter()
for _ in range(10): np.random.randint(0, X.shape[1], (K, 4))
print("np.random.randint(0, M, (K, 4)) alone: %.3f ms" % ((time.perf_counter() - t) * 100))
pr = cProfile.Profile(); pr.enable()
for _ in range(20 if K > 20000 else 50): run()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12); print(s.getvalue())
