"""Soak test (lives under tests/ because it calls the oracle; not collected by pytest): RANSAC.run against the CPU oracle (the reference's loop, same numpy seed) on random
problems from families that stress the settle step -- lattices (collinear triples, equal coordinates at different indices),
tight clusters, heavy contamination (the winner is often a repeated-index sample), tiny problems (M = 5..12: most samples repeat
an index), large coordinates -- with random th / d / k / n / loss.  Every run must return the oracle's winner iteration,
count and inlier list and leave numpy's generator where the oracle leaves it.
   python tests/soak_settle.py [cases] [seed]"""
import contextlib, io, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ransac as rs
from oracle import rwh_oracle as orc
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
HS = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "matchespoints.npz"))


def project(G, noise):
    P = np.concatenate([G, np.ones((len(G), 1))], 1) @ HS.T
    return P[:, :2] / P[:, 2:3] + rng.normal(0, noise, (len(G), 2))


def problem(kind):
    if kind == 0:      # lattice
        M = int(rng.integers(30, 900)); nx, ny = int(rng.integers(2, 20)), int(rng.integers(2, 12))
        G = np.stack([rng.integers(0, nx, M) * rng.uniform(5, 100), rng.integers(0, ny, M) * rng.uniform(5, 100)], 1)
        B = project(G, rng.uniform(0.1, 1.5))
    elif kind == 1:    # clusters
        M = int(rng.integers(30, 600)); c = rng.uniform(0, 2000, (int(rng.integers(3, 9)), 2))
        G = c[rng.integers(0, len(c), M)] + rng.normal(0, rng.choice([0.0, 0.01, 1.0]), (M, 2))
        B = project(G, 0.5)
    elif kind == 2:    # contaminated real matches
        G = z["ptsA"].astype(np.float64) * rng.choice([1.0, 8.0, 100.0]); B = z["ptsB"].astype(np.float64) * (G[0, 0] / z["ptsA"][0, 0])
    elif kind == 3:    # tiny
        M = int(rng.integers(5, 13)); G = rng.uniform(0, 500, (M, 2)); B = project(G, 0.5)
    else:              # uniform, large coordinates
        M = int(rng.integers(50, 1500)); G = rng.uniform(0, rng.choice([1e3, 1e4, 1e5]), (M, 2)); B = project(G, 1.0)
    out = rng.random(len(G)) < rng.choice([0.0, 0.2, 0.5, 0.8])
    B = B.copy(); B[out] = rng.uniform(0, max(1.0, float(np.abs(B).max())), (int(out.sum()), 2))
    return G.astype(np.float32), B.astype(np.float32)


bad = 0
settled = []
for case in range(cases):
    A, B = problem(case % 5)
    scale = max(1.0, float(np.abs(A).max()) / 1000.0)
    th = float(rng.choice([1, 3, 5])) * scale
    d = int(rng.choice([20, 40, 70, 95])); k = int(rng.integers(50, 400)); n = int(rng.choice([4, 4, 4, 6]))
    m = str(rng.choice(["fwd", "backward", "reproj"]))
    seed = int(rng.integers(0, 1 << 30))
    np.random.seed(seed)
    with np.errstate(all="ignore"), contextlib.redirect_stdout(io.StringIO()):
        try:
            Ho, inlo, cnto, ito = orc.ransac_run(A.T, B.T, th=th, d=d, n=n, k=k, method=m)
            nxt_o = np.random.randint(0, 1 << 30); err_o = None
        except Exception as e:      # e.g. np.where(None): nothing ever scored > 0; too few inliers for the refit
            err_o = type(e).__name__
    np.random.seed(seed)
    with np.errstate(all="ignore"), contextlib.redirect_stdout(io.StringIO()):
        r = rs.RANSAC(rs.HomoModel(th=th, d=d, n=n), k=k)
        try:
            Hg, inlg, cntg = r.run([A.T, B.T], method=m)
            nxt_g = np.random.randint(0, 1 << 30); err_g = None
        except Exception as e:
            err_g = type(e).__name__
    if err_o or err_g:
        ok = (err_o is not None) == (err_g is not None)
        if not ok:
            bad += 1; print("case %d: oracle %s, GPU path %s" % (case, err_o, err_g))
        continue
    ok = int(cntg) == int(cnto) and r.last_run["winner"] == ito and np.array_equal(inlg[0], inlo[0]) and nxt_g == nxt_o
    settled.append(r.last_run["host_settled"] / k)
    if not ok:
        bad += 1
        print("case %d kind %d M %d th %g d %d k %d n %d %s seed %d: oracle (it %d, count %d) GPU path (it %s, count %d) rng %s"
              % (case, case % 5, len(A), th, d, k, n, m, seed, ito, int(cnto), r.last_run["winner"], int(cntg), nxt_g == nxt_o))
print("%d cases, %d mismatches; host-solved share of the hypotheses: median %.3f, max %.3f" % (cases, bad, float(np.median(settled)), float(np.max(settled))))
sys.exit(1 if bad else 0)
