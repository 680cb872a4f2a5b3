"""Soak test of the numpy-facing module surface against the oracle (lives under tests/ because it calls the oracle; not collected by
pytest): random small problems through stitchPanorama (exact mode: canvases bit for bit, every blending value), wrapPerspective /
wrapPerspectiveScan / transformImageH (bit for bit, or the same exception type), run_batch with the caller's index tables
(several problems per call) against one oracle run per problem.   python tests/soak_api.py [cases] [seed]"""
import contextlib, io, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import homography as hg
import ransac as rs
from oracle import rwh_oracle as orc
from ransac_with_homography_amd import ransac as rmod
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
HS = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
bad = 0


def outcome(fn, *a, **k):
    try:
        with np.errstate(all="ignore"), contextlib.redirect_stdout(io.StringIO()):
            return "ok", fn(*a, **k)
    except Exception as e:      # noqa: BLE001
        return type(e).__name__, None


def rand_h(w, h, strong):
    t = rng.uniform(-0.3, 0.3) if not strong else rng.uniform(-np.pi, np.pi)
    s = rng.uniform(0.6, 1.6)
    A = np.array([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]]) * s
    H = np.eye(3); H[:2, :2] = A
    H[:2, 2] = rng.uniform(-0.8, 0.8, 2) * np.array([w, h])
    H[2, :2] = rng.uniform(-1e-3, 1e-3, 2)
    if rng.random() < 0.15:
        H = np.round(H, 0)          # integer maps: coordinates exactly on texels, the reference's IndexError
        H[2] = [0, 0, 1]
    return H


for case in range(cases):
    # ---- warps
    h, w = int(rng.integers(3, 60)), int(rng.integers(3, 80))
    c = 3 if rng.random() < 0.8 else 4
    img = rng.integers(0, 256, (h, w, c), dtype=np.uint8) if c == 3 else rng.uniform(0, 255, (h, w, c)).astype(np.float32)
    H = rand_h(w, h, rng.random() < 0.3)
    conv = str(rng.choice(["nn", "bilinear"]))
    for name, f_ref, f_got, args in (("wrapPerspective", orc.wrap_perspective, hg.wrapPerspective, (H,)),
                                     ("wrapPerspectiveScan", orc.wrap_perspective_scan, hg.wrapPerspectiveScan, (H, (int(rng.integers(2, 70)), int(rng.integers(2, 90))))),
                                     ("transformImageH", orc.transform_image_h, hg.transformImageH, (H,))):
        kw = {"convert": conv} if name != "transformImageH" else {"method": conv}
        o1, r1 = outcome(f_ref, img.copy(), *args, **kw)
        o2, r2 = outcome(f_got, img.copy(), *args, **kw)
        same = o1 == o2 and (o1 != "ok" or (r1[0].dtype == r2[0].dtype and np.array_equal(r1[0], r2[0], equal_nan=True) and (int(r1[1]), int(r1[2])) == (int(r2[1]), int(r2[2]))))
        if not same:
            bad += 1
            print("case %d %s %s img %s: oracle %s, product %s%s" % (case, name, conv, img.shape, o1, o2,
                  "" if o1 != "ok" or o2 != "ok" else " differ in %d values" % int((r1[0] != r2[0]).sum()) if r1[0].shape == r2[0].shape else " shapes %s %s" % (r1[0].shape, r2[0].shape)), flush=True)
    # ---- stitch
    Q = rng.integers(0, 256, (int(rng.integers(8, 50)), int(rng.integers(8, 70)), 3), dtype=np.uint8)
    T = rng.integers(0, 256, (int(rng.integers(8, 50)), int(rng.integers(8, 70)), 3), dtype=np.uint8)
    Hs = rand_h(T.shape[1], T.shape[0], False)
    blending = [False, "Rate", "Gradient", True][int(rng.integers(0, 4))]
    rate = float(rng.uniform(0.05, 0.9))
    o1, r1 = outcome(orc.stitch_panorama, Q.copy(), T.copy(), Hs, blending=blending, blendrate=rate)
    o2, r2 = outcome(hg.stitchPanorama, Q.copy(), T.copy(), Hs, blending=blending, blendrate=rate)
    if not (o1 == o2 and (o1 != "ok" or (r1.shape == r2.shape and r1.dtype == r2.dtype and np.array_equal(r1, r2)))):
        bad += 1
        print("case %d stitch blending %r Q %s T %s: oracle %s, product %s%s" % (case, blending, Q.shape, T.shape, o1, o2,
              "" if o1 != "ok" or o2 != "ok" else (" differ in %d values" % int((r1 != r2).sum()) if r1.shape == r2.shape else " shapes %s %s" % (r1.shape, r2.shape))), flush=True)
    # ---- run_batch with the caller's tables: P problems of different sizes in one call
    if case % 4 == 0:
        P = int(rng.integers(2, 5)); k = int(rng.integers(40, 200)); n = 4
        th = float(rng.choice([1, 3, 5])); d = int(rng.choice([20, 50, 90])); m = str(rng.choice(["fwd", "backward", "reproj"]))
        probs, tables, want = [], [], []
        for p_ in range(P):
            M = int(rng.integers(6, 300))
            G = rng.uniform(0, 800, (M, 2)) if p_ % 2 else rng.uniform(0, 2000, (int(rng.integers(3, 6)), 2))[rng.integers(0, 3, M)] + rng.normal(0, 0.01, (M, 2))
            Pp = np.concatenate([G, np.ones((M, 1))], 1) @ HS.T
            B = Pp[:, :2] / Pp[:, 2:3] + rng.normal(0, 0.6, (M, 2))
            o = rng.random(M) < rng.choice([0.0, 0.4]); B[o] = rng.uniform(0, 1000, (int(o.sum()), 2))
            A32, B32 = G.astype(np.float32), B.astype(np.float32)
            seed = int(rng.integers(0, 1 << 30))
            np.random.seed(seed)
            tables.append(np.random.randint(0, M, (k, n)))
            np.random.seed(seed)
            want.append(outcome(orc.ransac_run, A32.T, B32.T, th=th, d=d, n=n, k=k, method=m))
            probs.append([A32.T, B32.T])
        o2, got = outcome(rmod.run_batch, probs, th=th, d=d, n=n, k=k, method=m, idx=tables)
        # a problem whose winner has too few inliers for the refit: the reference raises AssertionError (ransac.py:38), run_batch returns
        # finalModel None for it (its docstring) and goes on with the others
        ok = o2 == "ok" and all((g_[0] is None) if w_[0] == "AssertionError" else
                                (w_[0] == "ok" and int(g_[2]) == int(w_[1][2]) and np.array_equal(g_[1][0], w_[1][1][0])) for g_, w_ in zip(got, want))
        if not ok:
            bad += 1
            print("case %d run_batch P %d k %d %s th %g d %d: oracle %s, product %s" % (case, P, k, m, th, d, [w_[0] for w_ in want], o2), flush=True)
    if case % 50 == 0:
        print("case %d ok so far (%d to look at)" % (case, bad), flush=True)
print("done: %d cases, %d to look at" % (cases, bad))
sys.exit(1 if bad else 0)
