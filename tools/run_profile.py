import os, sys, numpy as np, cProfile, pstats, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ransac as rs
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "matchespoints.npz"))
X, Y = z["ptsA"].T.copy(), z["ptsB"].T.copy()
def run():
    np.random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        return rs.RANSAC(rs.HomoModel(th=4, d=95, n=4), k=1500).run([X, Y], method="fwd")
for _ in range(5): run()
pr = cProfile.Profile(); pr.enable()
for _ in range(50): run()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
