"""Developer probe: where the time of one bit-exact RANSAC.run goes at K = 100 000 (K=<n> to change): the sample table, the native
driver rwh_ransac_run alone for several host-thread counts, the host SVD loop alone, the whole call.   python tools/run_phases.py"""
import os, sys, time, io, contextlib, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ransac as rs
from ransac_with_homography_amd import ransac as impl, kernels, _lib, _lapack

if os.environ.get("RWH_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["RWH_LIB"])      # e.g. a -DRWH_RUN_STAMPS build (phase times on stderr)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
z = np.load(os.path.join(root, "tests", "golden", "matchespoints.npz"))
X, Y = z["ptsA"].T.copy(), z["ptsB"].T.copy()
K = int(os.environ.get("K", "100000"))
dev = _lib.require_gpu()
pa = np.ascontiguousarray(X.T, dtype=np.float32); pb = np.ascontiguousarray(Y.T, dtype=np.float32)
M = pa.shape[0]


def best_of(f, n=15, warm=3):
    for _ in range(warm): f()
    v = []
    for _ in range(n):
        t = time.perf_counter(); f(); v.append((time.perf_counter() - t) * 1e3)
    v.sort()
    return v[0], v[len(v) // 2]


np.random.seed(0)
print("sample table (rwh_host_legacy_randint, int32 only): min %.3f median %.3f ms" % best_of(lambda: impl.legacy_randint_table(M, K, 4, want64=False)))
print("sample table (int64 too):                            min %.3f median %.3f ms" % best_of(lambda: impl.legacy_randint_table(M, K, 4, want64=True)))
np.random.seed(0)
_, idx32 = impl.legacy_randint_table(M, K, 4, want64=False)
addr, gesv = _lapack.dgesdd_address(), _lapack.routine_address("dgesv")
ws = kernels.RunWorkspace(M, K, dev)
need, thr = kernels.need_count(M, int(os.environ.get("D", "70")), 4), impl._weak_threshold(5)
for th in [int(v) for v in os.environ.get("THREADS", "8,16,24,32,48,64").split(",")]:
    r = [None]
    def f():
        r[0] = kernels.ransac_run(pa, pb, idx32, thr, "fwd", need, 8, ws, addr, th, dgesv=gesv, want_keys=True)
    print("rwh_ransac_run alone, %2d host threads: min %.3f median %.3f ms   (winner %s, host-solved %d, rounds %d, intervals %d)" %
          ((th,) + best_of(f) + (r[0][0], r[0][3], r[0][4], r[0][8])))
if os.environ.get("RWH_LIB"): sys.exit(0)
rep = impl.repeated_rows(idx32)
rows = np.ascontiguousarray(idx32[rep])
for th in (8, 16, 32, 64):
    print("host SVD loop alone, %d samples, %2d threads: min %.3f median %.3f ms" % ((len(rows), th) + best_of(lambda: impl.svd_hypotheses(pa, pb, rows, threads=th))))
r = rs.RANSAC(rs.HomoModel(th=5, d=int(os.environ.get("D", "70")), n=4), k=K)
def whole():
    np.random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        r.run([X, Y], method="fwd")
print("RANSAC.run whole: min %.3f median %.3f ms; os.cpu_count %s, sched_getaffinity %d, HOST_THREADS %d" %
      (best_of(whole) + (os.cpu_count(), len(os.sched_getaffinity(0)), impl.HOST_THREADS)))
