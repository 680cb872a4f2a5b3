#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own functions.

Runs only in the build container, where the read-only reference checkout is
mounted at /root/reference.  It imports the reference's unmodified
`homography.py` and `ransac.py` (their one missing dependency, OpenCV, is
satisfied by an empty stub module: no hot-path function touches cv2) and
records inputs + outputs as plain, pickle-free numpy arrays.  No reference
source text is written anywhere; the fixtures are data only.

Usage:  python tests/golden/make_golden.py            (writes next to this file)

Fixture IDs follow SURVEY.md section 8(c): G1..G8.
"""
import hashlib
import os
import sys
import types

import numpy as np

REF = os.environ.get("RWH_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, REF)
import homography as ref_h  # noqa: E402  (the reference)
import ransac as ref_r      # noqa: E402  (the reference)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def decode(jpg):
    from PIL import Image
    return np.asarray(Image.open(os.path.join(REF, jpg)).convert("RGB"), dtype=np.uint8).copy()


def digest(prefix, img, rng, nsample=20000):
    """Summary of a big output: shape, dtype-independent sha256 of the raw
    bytes, per-channel sums, and `nsample` (flat index, value) pairs."""
    flat = img.reshape(-1)
    pick = np.sort(rng.choice(flat.size, size=min(nsample, flat.size), replace=False))
    return {
        prefix + "_shape": np.array(img.shape, dtype=np.int64),
        prefix + "_sha256": np.array(sha(img)),
        prefix + "_chansum": img.reshape(-1, img.shape[2]).sum(axis=0, dtype=np.float64),
        prefix + "_pick": pick.astype(np.int64),
        prefix + "_vals": flat[pick].copy(),
    }


# ---------------------------------------------------------------- inputs ----
U0 = np.array([[50, 470, 600., 90], [50, 40, 300., 360], [1, 1, 1, 1.]])
V0 = np.array([[0, 400, 400., 0], [0, 0, 780., 780], [1, 1, 1, 1]])
H_BENCH = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
H_NOTEBOOK = np.array([[7.92272362e-01, 8.22117652e-02, 4.34671442e+02],
                       [-1.33501977e-01, 9.58413827e-01, 6.55962640e+01],
                       [-2.09248580e-04, 1.60937718e-05, 1.00000000e+00]])
# scanner target for an A4 page, app-style: box=(h,w)=(1188,840)
V_A4 = np.array([[0, 840, 840., 0], [0, 0, 1188., 1188], [1, 1, 1, 1]])


def g1():
    u, v = U0.T[:, :2], V0.T[:, :2]
    A, b = ref_h.calc_correspLinear(u, v)
    mat = ref_h.calc_corresp(u, v)
    save("g1_fourpoint", u=U0, v=V0, A=A, b=b, mat=mat,
         H_linear=ref_h.calcHomographyLinear(u, v),
         H_dlt=ref_h.calcHomography(u, v),
         # float32 inputs exercise the float32-product path of the builders
         mat_f32in=ref_h.calc_corresp(u.astype(np.float32), v.astype(np.float32)),
         H_dlt_f32in=ref_h.calcHomography(u.astype(np.float32), v.astype(np.float32)))


def load_matches():
    a = np.load(os.path.join(REF, "matchespoints.npy"), allow_pickle=True).tolist()
    return np.ascontiguousarray(a["ptsA"], dtype=np.float32), np.ascontiguousarray(a["ptsB"], dtype=np.float32)


def per_hypothesis(ptsA, ptsB, seed, K, th):
    X, Y = ptsA.T, ptsB.T
    np.random.seed(seed)
    idx = np.random.randint(0, X.shape[1], (K, 4))
    model = ref_r.HomoModel(th=th, d=70, n=4)
    Hs = np.empty((K, 9), np.float32)
    counts = {m: np.empty(K, np.int16) for m in ("fwd", "backward", "reproj")}
    for i in range(K):
        val = model.fit(X[:, idx[i]], Y[:, idx[i]])
        assert val.dtype == np.float32
        Hs[i] = val.reshape(9)
        for m in counts:
            counts[m][i] = np.sum(model.computeLoss(X, Y, m) < th)
    return idx.astype(np.int32), Hs, counts


def g2_g3(ptsA, ptsB):
    X, Y = ptsA.T, ptsB.T
    for name, seed in (("g2_hyp_seed0", 0), ("g3_hyp_seed7", 7)):
        idx, Hs, counts = per_hypothesis(ptsA, ptsB, seed, 10000, 5)
        c = counts["fwd"].astype(np.int64)
        win = int(np.argmax(c))
        model = ref_r.HomoModel(th=5, d=70, n=4)
        model.val = Hs[win].reshape(3, 3)
        err = model.computeLoss(X, Y, "fwd")
        save(name, seed=np.int64(seed), idx=idx, H=Hs,
             counts_fwd=counts["fwd"], counts_backward=counts["backward"], counts_reproj=counts["reproj"],
             degenerate=np.array([len(set(r)) < 4 for r in idx.tolist()]),
             winner=np.int64(win), winner_count=np.int64(c[win]),
             winner_ties=np.int64(np.sum(c == c[win])),
             winner_inliers=np.where(err < 5)[0].astype(np.int64),
             winner_err=err.astype(np.float32))


def run_ref_ransac(ptsA, ptsB, seed, th, d, k, method):
    np.random.seed(seed)
    model = ref_r.HomoModel(th=th, d=d, n=4)
    H, inl, cnt = ref_r.RANSAC(model, k=k).run([ptsA.T, ptsB.T], method=method)
    return np.asarray(H, np.float64), inl[0].astype(np.int64), np.int64(cnt)


def g4_g5(ptsA, ptsB):
    out = {}
    # G4: ransac.example0 parameters; G5: app.py panorama parameters; GE: early-exit cases
    cases = [("g4", s, 5, 70, 1000, m) for s in (0, 1, 2) for m in ("fwd", "backward", "reproj")]
    cases += [("g5", 0, 4, 95, 1500, "fwd")]
    cases += [("ge", s, 5, 50, 1000, "fwd") for s in (0, 3)] + [("ge", 1, 5, 40, 1000, "reproj")]
    names = []
    for tag, seed, th, d, k, m in cases:
        H, inl, cnt = run_ref_ransac(ptsA, ptsB, seed, th, d, k, m)
        key = "%s_s%d_th%d_d%d_k%d_%s" % (tag, seed, th, d, k, m)
        names.append(key)
        out[key + "_H"] = H
        out[key + "_inliers"] = inl
        out[key + "_count"] = cnt
        print(key, int(cnt))
    out["cases"] = np.array(names)
    save("g4_ransac_runs", **out)


def contaminate(ptsB, seed, fracs):
    """Replace a random subset of the matches' second points by uniform noise (the last fraction decides)."""
    rng = np.random.default_rng(seed)
    for frac in fracs:
        Bc = ptsB.copy()
        m = rng.random(ptsB.shape[0]) < frac
        Bc[m] = rng.uniform(0, 1000, (int(m.sum()), 2)).astype(np.float32)
    return Bc


def g9(ptsA, ptsB):
    """Low-inlier problems on which the REFERENCE's winner is a sample with a repeated index: np.random.randint draws
    with replacement (ransac.py:177), the rank-deficient 8x9 system still yields an H (LAPACK's arbitrary null vector,
    homography.py:81-87) and that H competes for the best (ransac.py:199-202).  Per case: the run's outputs and the
    reference's per-iteration H / counts (model.fit + computeLoss, exactly the loop body)."""
    out = {}
    names = []
    cases = [("a", 0, (0.0, 0.3, 0.5), 2, 5, 70, 1000, "fwd"),
             ("a", 0, (0.0, 0.3, 0.5), 2, 5, 70, 1000, "backward"),
             ("a", 0, (0.0, 0.3, 0.5), 2, 5, 70, 1000, "reproj"),
             ("a", 0, (0.0, 0.3, 0.5), 2, 5, 20, 1000, "fwd"),      # need = 41: the repeated sample 151 triggers `break`
             ("b", 9, (0.7,), 2, 5, 70, 1000, "fwd"),
             ("c", 11, (0.6,), 0, 5, 70, 1000, "fwd")]
    for tag, cseed, fracs, seed, th, d, k, m in cases:
        Bc = contaminate(ptsB, cseed, fracs)
        out["ptsB_" + tag] = Bc
        H, inl, cnt = run_ref_ransac(ptsA, Bc, seed, th, d, k, m)
        key = "%s_s%d_th%d_d%d_k%d_%s" % (tag, seed, th, d, k, m)
        names.append(key)
        X, Y = ptsA.T, Bc.T
        np.random.seed(seed)
        idx = np.random.randint(0, X.shape[1], (k, 4))
        model = ref_r.HomoModel(th=th, d=d, n=4)
        Hs = np.empty((k, 9), np.float32)
        counts = np.empty(k, np.int32)
        with np.errstate(all="ignore"):
            for i in range(k):
                Hs[i] = model.fit(X[:, idx[i]], Y[:, idx[i]]).reshape(9)
                counts[i] = np.sum(model.computeLoss(X, Y, m) < th)
        need = X.shape[1] * d / 100 + 4
        hit = np.nonzero(counts >= need)[0]
        win = int(hit[0]) if hit.size else int(np.argmax(counts))
        assert counts[win] == cnt, (key, counts[win], cnt)
        out[key + "_H"] = H
        out[key + "_inliers"] = inl
        out[key + "_count"] = cnt
        out[key + "_idx"] = idx.astype(np.int32)
        out[key + "_hyp_H"] = Hs
        out[key + "_hyp_counts"] = counts
        out[key + "_winner"] = np.int64(win)
        out[key + "_early"] = np.bool_(hit.size > 0)
        print(key, "count", int(cnt), "winner", win, idx[win].tolist(), "early" if hit.size else "")
    out["cases"] = np.array(names)
    save("g9_low_inlier", **out)


def g10(ptsA, ptsB):
    """BASELINE config 5's search: 100 000 hypotheses (seed 0) on matchespoints, th = 5, 'fwd' -- the reference's loop
    body per hypothesis.  Winner / count / inlier list, and the counts themselves (int16) + their SHA-256."""
    X, Y = ptsA.T, ptsB.T
    K = 100000
    np.random.seed(0)
    idx = np.random.randint(0, X.shape[1], (K, 4))
    model = ref_r.HomoModel(th=5, d=70, n=4)
    counts = np.empty(K, np.int32)
    win_mask = None
    best = 0
    with np.errstate(all="ignore"):
        for i in range(K):
            model.fit(X[:, idx[i]], Y[:, idx[i]])
            inl = model.computeLoss(X, Y, "fwd") < 5
            counts[i] = np.sum(inl)
            if counts[i] > best:
                best, win_mask = counts[i], inl
    win = int(np.argmax(counts))
    save("g10_config5_search", K=np.int64(K), seed=np.int64(0), counts=counts.astype(np.int16),
         counts_sha256=np.array(sha(counts)), winner=np.int64(win), winner_count=np.int64(counts[win]),
         winner_ties=np.int64(np.sum(counts == counts[win])), winner_sample=idx[win].astype(np.int32),
         winner_inliers=np.where(win_mask)[0].astype(np.int64),
         degenerate=np.int64(sum(len(set(r)) < 4 for r in idx.tolist())))
    print("g10 winner", win, int(counts[win]))


def small_images():
    rng = np.random.default_rng(1234)
    noise = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:60, 0:80]
    ramp = np.stack([(xx * 255 // 79), (yy * 255 // 59), ((xx + yy) * 255 // 138)], axis=2).astype(np.uint8)
    return {"noise": noise, "ramp": ramp}


def g6():
    u, v = U0.T[:, :2], V0.T[:, :2]
    Hs = {"bench": H_BENCH, "g1lin": ref_h.calcHomographyLinear(u, v), "notebook": H_NOTEBOOK,
          # a rotation + zoom: source footprint of an output tile is a tilted quad
          "rot": np.array([[1.2 * np.cos(0.5), -1.2 * np.sin(0.5), 40.0],
                           [1.2 * np.sin(0.5), 1.2 * np.cos(0.5), -10.0],
                           [3e-4, -2e-4, 1.0]])}
    out = {"H_names": np.array(list(Hs))}
    for hn, H in Hs.items():
        out["H_" + hn] = H
    for iname, img in small_images().items():
        out["img_" + iname] = img
        for hn, H in Hs.items():
            for conv in ("nn", "bilinear"):
                o, mx, my = ref_h.wrapPerspective(img.copy(), H, convert=conv)
                k = "wp_%s_%s_%s" % (iname, hn, conv)
                out[k] = o
                out[k + "_org"] = np.array([mx, my], np.int64)
            o, mx, my = ref_h.transformImageH(img.copy(), H)
            out["tih_%s_%s" % (iname, hn)] = o
        # boundary=1 clamps the origin at 0
        o, mx, my = ref_h.wrapPerspective(img.copy(), Hs["rot"], convert="bilinear", boundary=1)
        out["wpb_%s_rot_bilinear" % iname] = o
        out["wpb_%s_rot_bilinear_org" % iname] = np.array([mx, my], np.int64)
        # fixed-resolution scan warps (bounds use res, which must not exceed the source)
        hs, ws, _ = img.shape
        for conv in ("nn", "bilinear"):
            for hn in ("bench", "rot"):
                res = (hs - 8, ws - 16)
                o, _, _ = ref_h.wrapPerspectiveScan(img.copy(), Hs[hn], res, convert=conv)
                out["scan_%s_%s_%s" % (iname, hn, conv)] = o
        # 4-channel float32 image (constant-rate alpha), both interpolators
        rgba = ref_h.addAlpha(img.copy(), method="Rate", rate=0.2)
        out["rgba_" + iname] = rgba
        for conv in ("nn", "bilinear"):
            o, mx, my = ref_h.wrapPerspective(rgba.copy(), Hs["notebook"], convert=conv)
            out["wp4_%s_notebook_%s" % (iname, conv)] = o
    save("g6_small_warps", **out)


def g7():
    rng = np.random.default_rng(7)
    img = decode("notebook.jpg")
    u, v = U0.T[:, :2], V0.T[:, :2]
    H = ref_h.calcHomographyLinear(u, v)
    out = {"H": H, "u": U0, "v": V0, "v_a4": V_A4}
    o, mx, my = ref_h.wrapPerspective(img.copy(), H, convert="bilinear")
    out.update(digest("wp_bilinear", o, rng)); out["wp_bilinear_org"] = np.array([mx, my], np.int64)
    o, mx, my = ref_h.wrapPerspective(img.copy(), H, convert="nn")
    out.update(digest("wp_nn", o, rng))
    o = ref_h.transformImage(img.copy(), U0, V0)
    out.update(digest("ti", o, rng))
    o = ref_h.transformImage(img.copy(), U0, V0, method="nn")
    out.update(digest("ti_nn", o, rng))
    # scanner mode, app.py:350-359 with the A4 preset: box=(h,w)=(1188,840).  The
    # inverse map sends the box back inside the clicked quad, so the source being
    # smaller than `res` does not trip the reference's IndexError here.
    o = ref_h.transformImage(img.copy(), U0, V_A4, box=[1188, 840])
    out.update(digest("scan_a4", o, rng))
    o = ref_h.transformImage(img.copy(), U0, V_A4, box=[1188, 840], method="nn")
    out.update(digest("scan_a4_nn", o, rng))
    save("g7_notebook", **out)
    save("img_notebook", img=img)


def g8():
    rng = np.random.default_rng(8)
    A = decode("foto1A.jpg")
    B = decode("foto1B.jpg")
    out = {"H_notebook": H_NOTEBOOK}
    o, mx, my = ref_h.transformImageH(A.copy(), H_NOTEBOOK)
    out.update(digest("tih", o, rng)); out["tih_org"] = np.array([mx, my], np.int64)
    o = ref_h.stitchPanorama(B.copy(), A.copy(), H_NOTEBOOK)
    out.update(digest("stitch_paste", o, rng))
    o = ref_h.stitchPanorama(B.copy(), A.copy(), H_NOTEBOOK, blending="Rate", blendrate=0.2)
    out.update(digest("stitch_rate", o, rng))
    # with the G5 RANSAC homography (app.py parameters, seed 0)
    ptsA, ptsB = load_matches()
    H5, _, _ = run_ref_ransac(ptsA, ptsB, 0, 4, 95, 1500, "fwd")
    out["H_g5"] = H5
    o = ref_h.stitchPanorama(B.copy(), A.copy(), H5, blending="Rate", blendrate=0.2)
    out.update(digest("stitch_g5_rate", o, rng))
    save("g8_stitch", **out)
    save("img_foto1", A=A, B=B)


def g11():
    """'Gradient' blending (homography.py:259-266, 327-334): the alpha ramp the reference leaves half finished but runs."""
    rng = np.random.default_rng(11)
    A = decode("foto1A.jpg")
    B = decode("foto1B.jpg")
    H5 = np.load(os.path.join(OUT, "g8_stitch.npz"))["H_g5"]
    out = {}
    o = ref_h.stitchPanorama(B.copy(), A.copy(), H_NOTEBOOK, blending="Gradient")
    out.update(digest("stitch_gradient", o, rng))
    o = ref_h.stitchPanorama(B.copy(), A.copy(), H5, blending="Gradient")
    out.update(digest("stitch_g5_gradient", o, rng))
    save("g11_stitch_gradient", **out)


H_S = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])


def lattice_problem(seed, M, nx=13, ny=5, sx=80., sy=90., noise=.7, outl=.2):
    """Source points on a coarse lattice (many collinear triples, many equal coordinates at different indices), targets =
    H_S-projected + noise, a fraction replaced by uniform outliers: the samples K1's elimination and LAPACK's SVD disagree
    on WITHOUT a repeated index (round-2 verdict, "What's weak" 1b)."""
    rng = np.random.default_rng(seed)
    G = np.stack([rng.integers(0, nx, M) * sx, rng.integers(0, ny, M) * sy], 1)
    P = np.concatenate([G, np.ones((M, 1))], 1) @ H_S.T
    P = P[:, :2] / P[:, 2:3]
    B = P + rng.normal(0, noise, (M, 2))
    out = rng.random(M) < outl
    B[out] = rng.uniform(0, 1000, (int(out.sum()), 2))
    return G.astype(np.float32), B.astype(np.float32)


def cluster_problem(seed, M):
    """Six tight clusters of source points: most samples take two points of one cluster (nearly equal coordinates)."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(0, 2000, (6, 2))
    G = c[rng.integers(0, 6, M)] + rng.normal(0, 0.01, (M, 2))
    P = np.concatenate([G, np.ones((M, 1))], 1) @ H_S.T
    P = P[:, :2] / P[:, 2:3]
    B = P + rng.normal(0, .5, (M, 2))
    return G.astype(np.float32), B.astype(np.float32)


def g12():
    """Ill-conditioned samples WITHOUT a repeated index (collinear triples, equal coordinates at distinct indices): the
    reference's RANSAC.run on lattice / cluster problems, plus its loop body per iteration (H, count) so that the GPU
    path's conditioning flag and settle step are pinned hypothesis by hypothesis."""
    out = {}
    names = []
    cases = [("lat2024", lattice_problem(2024, 3000), 41, 3, 70, 1500, "fwd"),
             ("lat2024", lattice_problem(2024, 3000), 41, 3, 60, 1500, "fwd"),       # need = 1804: early exit on the way
             ("lat2024", lattice_problem(2024, 3000), 41, 3, 95, 1500, "fwd"),       # need = 2854 > best: running best over all
             ("lat5", lattice_problem(5, 1500), 3, 3, 70, 1000, "reproj"),
             ("clus1", cluster_problem(1, 600), 3, 3, 90, 1000, "fwd"),
             ("clus1", cluster_problem(1, 600), 3, 3, 90, 1000, "backward"),
             ("clus1", cluster_problem(1, 600), 3, 1, 101, 1000, "fwd")]             # need > M: ties at the top, first index wins
    for tag, (A, B), seed, th, d, k, m in cases:
        out["ptsA_" + tag] = A
        out["ptsB_" + tag] = B
        with np.errstate(all="ignore"):
            H, inl, cnt = run_ref_ransac(A, B, seed, th, d, k, m)
        key = "%s_s%d_th%d_d%d_k%d_%s" % (tag, seed, th, d, k, m)
        names.append(key)
        X, Y = A.T, B.T
        np.random.seed(seed)
        idx = np.random.randint(0, X.shape[1], (k, 4))
        model = ref_r.HomoModel(th=th, d=d, n=4)
        Hs = np.empty((k, 9), np.float32)
        counts = np.empty(k, np.int32)
        with np.errstate(all="ignore"):
            for i in range(k):
                Hs[i] = model.fit(X[:, idx[i]], Y[:, idx[i]]).reshape(9)
                counts[i] = np.sum(model.computeLoss(X, Y, m) < th)
        need = X.shape[1] * d / 100 + 4
        hit = np.nonzero(counts >= need)[0]
        win = int(hit[0]) if hit.size else int(np.argmax(counts))
        assert counts[win] == cnt, (key, counts[win], cnt)
        out[key + "_H"] = H
        out[key + "_inliers"] = inl
        out[key + "_count"] = cnt
        out[key + "_idx"] = idx.astype(np.int32)
        out[key + "_hyp_H"] = Hs
        out[key + "_hyp_counts"] = counts
        out[key + "_winner"] = np.int64(win)
        out[key + "_early"] = np.bool_(hit.size > 0)
        print(key, "count", int(cnt), "winner", win, idx[win].tolist(), "early" if hit.size else "",
              "repeated", sum(len(set(r)) < 4 for r in idx.tolist()))
    out["cases"] = np.array(names)
    save("g12_illcond", **out)


def g13(ptsA, ptsB):
    """BASELINE config 4's RANSAC at the x8 scale: foto1A/foto1B upsampled x8 means matches x 8 and th x 8 (app.py
    parameters th = 4 -> 32, d = 95, k = 1500, 'fwd', seed 0): the reference run itself on the scaled points."""
    A8, B8 = (ptsA * 8).astype(np.float32), (ptsB * 8).astype(np.float32)
    H, inl, cnt = run_ref_ransac(A8, B8, 0, 32, 95, 1500, "fwd")
    X, Y = A8.T, B8.T
    np.random.seed(0)
    idx = np.random.randint(0, X.shape[1], (1500, 4))
    model = ref_r.HomoModel(th=32, d=95, n=4)
    counts = np.empty(1500, np.int32)
    with np.errstate(all="ignore"):
        for i in range(1500):
            model.fit(X[:, idx[i]], Y[:, idx[i]])
            counts[i] = np.sum(model.computeLoss(X, Y, "fwd") < 32)
    win = int(np.argmax(counts))
    assert counts[win] == cnt
    out = dict(H=H, inliers=inl, count=cnt, winner=np.int64(win), hyp_counts=counts)
    # HomoModel(n = 6): six indices per iteration, fit on the first four, exit at d + 6 (ransac.py:177-190; homography.py:4-14)
    for seed, d in ((0, 70), (3, 50)):
        np.random.seed(seed)
        model = ref_r.HomoModel(th=5, d=d, n=6)
        H6, inl6, cnt6 = ref_r.RANSAC(model, k=1000).run([ptsA.T, ptsB.T], method="fwd")
        after = np.random.randint(0, 1 << 30)                        # where the run left numpy's generator
        key = "n6_s%d_d%d" % (seed, d)
        out[key + "_H"] = np.asarray(H6, np.float64); out[key + "_inliers"] = inl6[0].astype(np.int64)
        out[key + "_count"] = np.int64(cnt6); out[key + "_next_draw"] = np.int64(after)
        print(key, int(cnt6))
    save("g13_config4_x8", **out)
    print("g13 count", int(cnt), "winner", win)


def edge_cases():
    """Inputs of g14: the corners of RANSAC.run's input space (name, ptsA [M,2] float32, ptsB, th, d, n, k)."""
    rng = np.random.default_rng(77)
    Hs = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])

    def pair(M, noise=0.5):
        G = rng.uniform(0, 800, (M, 2))
        P = np.concatenate([G, np.ones((M, 1))], 1) @ Hs.T
        return G.astype(np.float32), (P[:, :2] / P[:, 2:3] + rng.normal(0, noise, (M, 2))).astype(np.float32)
    cases = []
    for M in (0, 1, 2, 3, 4, 5):
        A, B = pair(M)
        cases.append(("M%d" % M, A, B, 5, 1, 4, 30))
    A, B = pair(60)
    for i, (th, d, n, k) in enumerate(((5, 20, 4, 0), (5, 20, 4, 1), (0, 20, 4, 50), (-1, 20, 4, 50), (1e30, 20, 4, 50), (5, 0, 4, 50),
                                       (5, 1000, 4, 50), (5, 20, 60, 20), (5, 20, 61, 20), (2.5, 20.5, 4, 50), (5, 20, 3, 20))):
        cases.append(("par%d" % i, A, B, th, d, n, k))
    same = np.tile(np.array([[10.0, 20.0]], np.float32), (40, 1))
    cases.append(("allequal", same, same.copy(), 5, 20, 4, 40))
    line = np.stack([np.arange(50, dtype=np.float32) * 7, np.arange(50, dtype=np.float32) * 3], 1)
    cases.append(("collinear", line, line + 1.0, 5, 20, 4, 40))
    for name, bad in (("nan", np.nan), ("pinf", np.inf), ("ninf", -np.inf)):
        A2, B2 = A.copy(), B.copy()
        A2[7, 0] = bad; B2[13, 1] = bad
        cases.append((name, A2, B2, 5, 20, 4, 60))
    cases.append(("ragged", A, B[:-1], 5, 20, 4, 10))
    return cases


def g14():
    """What the reference itself does at the corners of RANSAC.run's input space (seed 4242, 'fwd' and 'reproj'): the result
    (count, inlier list, where the run left numpy's generator) or the TYPE of the exception it raises -- a k of 0 reads a
    variable the loop never assigned, no hypothesis with an inlier makes np.where(None) the index (an error from numpy 2.1
    on), a NaN coordinate makes LAPACK's SVD fail at the iteration that samples it, ..."""
    import contextlib
    import io
    out = {"numpy_version": np.array(np.__version__)}
    names = []
    for name, A, B, th, d, n, k in edge_cases():
        out[name + "_A"] = A; out[name + "_B"] = B
        out[name + "_par"] = np.array([th, d, n, k], dtype=np.float64)
        names.append(name)
        for m in ("fwd", "reproj"):
            np.random.seed(4242)
            key = "%s_%s" % (name, m)
            with np.errstate(all="ignore"), contextlib.redirect_stdout(io.StringIO()):
                try:
                    model = ref_r.HomoModel(th=th, d=d, n=n)
                    H, inl, cnt = ref_r.RANSAC(model, k=k).run([A.T, B.T], method=m)
                    out[key + "_outcome"] = np.array("ok")
                    out[key + "_count"] = np.int64(cnt); out[key + "_inliers"] = np.asarray(inl[0]).astype(np.int64)
                    out[key + "_H"] = np.asarray(H, np.float64)
                except Exception as e:      # noqa: BLE001 -- the type is the datum
                    out[key + "_outcome"] = np.array(type(e).__name__)
            out[key + "_next_draw"] = np.int64(np.random.randint(0, 1 << 30))
            print(key, str(out[key + "_outcome"]), int(out.get(key + "_count", -1)))
    out["names"] = np.array(names)
    save("g14_edge_cases", **out)


def warp_edge_cases():
    """Inputs of g15: (name, function name, image, args) at the corners of the warp entry points' input space."""
    rng = np.random.default_rng(15)
    img = rng.integers(1, 256, (9, 11, 3), dtype=np.uint8)
    tiny = rng.integers(1, 256, (2, 2, 3), dtype=np.uint8)
    three = rng.integers(1, 256, (3, 3, 3), dtype=np.uint8)
    Hs = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
    t = np.pi / 2
    Hm = {"identity": np.eye(3), "shift": np.array([[1, 0, 2.0], [0, 1, -3.0], [0, 0, 1]]), "scale2": np.diag([2.0, 2.0, 1.0]),
          "mild": Hs, "rot90": np.array([[np.cos(t), -np.sin(t), 8.0], [np.sin(t), np.cos(t), 0.0], [0, 0, 1.0]]),
          "mirror": np.array([[-1.0, 0, 10.0], [0, 1.0, 0], [0, 0, 1]]), "halfshift": np.array([[1, 0, 0.5], [0, 1, 0.25], [0, 0, 1.0]]),
          "singular": np.array([[1.0, 2, 3], [2, 4, 6], [0, 0, 1]]), "nan": np.array([[1.0, 0, np.nan], [0, 1, 0], [0, 0, 1]]),
          "persp": np.array([[0.9, 0.2, 1.0], [-0.15, 1.1, 3.0], [3e-3, -2e-3, 1.0]])}
    cases = []
    for hn, H in Hm.items():
        for conv in ("nn", "bilinear"):
            cases.append(("wp_%s_%s" % (hn, conv), "wrapPerspective", img, dict(H=H, convert=conv)))
        cases.append(("tih_%s" % hn, "transformImageH", img, dict(H=H)))
    for hn in ("identity", "mild", "halfshift"):
        for conv in ("nn", "bilinear"):
            for res in ((9, 11), (5, 7), (12, 14)):
                cases.append(("scan_%s_%s_%dx%d" % (hn, conv, res[0], res[1]), "wrapPerspectiveScan", img, dict(H=Hm[hn], res=res, convert=conv)))
    for conv in ("nn", "bilinear"):
        cases.append(("tiny_mild_%s" % conv, "wrapPerspective", tiny, dict(H=Hs, convert=conv)))
        cases.append(("tiny_half_%s" % conv, "wrapPerspective", tiny, dict(H=Hm["halfshift"], convert=conv)))
        cases.append(("three_mild_%s" % conv, "wrapPerspective", three, dict(H=Hs, convert=conv)))
    # transformImage(img, u, v, box): u / v = 3 x 4 corner lists (app.py:351-356), box = the scan resolution or None
    u = np.array([[1.0, 9.0, 8.5, 1.5], [1.0, 1.5, 7.0, 6.5], [1, 1, 1, 1]])
    v = np.array([[0.0, 7, 7, 0], [0.0, 0, 5, 5], [1, 1, 1, 1]])
    corners = np.array([[0.0, 10, 10, 0], [0.0, 0, 8, 8], [1, 1, 1, 1]])
    cases.append(("ti_quad_scan", "transformImage", img, dict(u=u, v=v, box=(6, 8))))
    cases.append(("ti_quad_auto", "transformImage", img, dict(u=u, v=v)))
    cases.append(("ti_quad_nn", "transformImage", img, dict(u=u, v=v, box=(6, 8), method="nn")))
    cases.append(("ti_corners_scan", "transformImage", img, dict(u=corners, v=corners, box=(9, 11))))
    cases.append(("ti_corners_auto", "transformImage", img, dict(u=corners, v=corners)))
    return cases


def g15():
    """What the reference's warp entry points do at the corners of their input space: the output array, or the TYPE of the
    exception -- bilinear indexes texel x + 1, so a coordinate exactly on the last column / row (identity, integer shifts, pure
    scales, rot90) raises IndexError; a scan `res` beyond the image lets coordinates past it through; a singular H fails in
    np.linalg.inv; a NaN entry fails in int(); 2 x 2 images work."""
    out = {"numpy_version": np.array(np.__version__)}
    names = []
    for name, fn, img, kw in warp_edge_cases():
        names.append(name)
        out[name + "_fn"] = np.array(fn); out[name + "_img"] = img
        for k_, v in kw.items():
            out[name + "_arg_" + k_] = np.asarray(v) if not isinstance(v, str) else np.array(v)
        try:
            with np.errstate(all="ignore"):
                r = getattr(ref_h, fn)(img.copy(), **kw)
            arr = r[0] if isinstance(r, tuple) else r
            out[name + "_outcome"] = np.array("ok")
            out[name + "_out"] = np.asarray(arr)
            if isinstance(r, tuple):
                out[name + "_origin"] = np.array([r[1], r[2]], dtype=np.int64)
        except Exception as e:      # noqa: BLE001 -- the type is the datum
            out[name + "_outcome"] = np.array(type(e).__name__)
        print(name, str(out[name + "_outcome"]), out.get(name + "_out", np.zeros(0)).shape)
    out["names"] = np.array(names)
    save("g15_warp_edge_cases", **out)


def g16():
    """stitchPanorama on small images for every branch of its canvas geometry (homography.py:303-321: the warped image to the
    left / right of and above / below the query image, inside it, covering it) and every `blending` value the reference's code
    distinguishes (False, 'Rate', 'Gradient', another truthy value), plus the identity (IndexError from the bilinear warp)."""
    import contextlib
    import io
    rng = np.random.default_rng(16)
    Q = rng.integers(1, 256, (20, 30, 3), dtype=np.uint8)
    T = rng.integers(1, 256, (18, 26, 3), dtype=np.uint8)
    big = rng.integers(1, 256, (40, 52, 3), dtype=np.uint8)
    P = np.array([[1.0, 0.01, 0], [0.012, 0.99, 0], [2e-4, 1e-4, 1.0]])

    def shift(tx, ty):
        S = np.eye(3); S[0, 2] = tx; S[1, 2] = ty
        return S @ P
    Hm = {"left_up": shift(-7.3, -5.6), "left_down": shift(-9.2, 6.4), "right_up": shift(11.5, -4.3), "right_down": shift(13.7, 8.2),
          "inside": shift(2.4, 1.3), "far_right": shift(45.2, 3.1), "far_up": shift(3.3, -31.7), "identity": np.eye(3)}
    out = {"numpy_version": np.array(np.__version__), "Q": Q, "T": T, "big": big}
    names = []
    for hn, H in Hm.items():
        for bn, blending in (("paste", False), ("rate", "Rate"), ("grad", "Gradient"), ("true", True)):
            for tn, timg in (("T", T), ("big", big)):
                if tn == "big" and hn not in ("left_up", "inside"):
                    continue
                name = "%s_%s_%s" % (hn, bn, tn)
                names.append(name)
                out[name + "_H"] = H
                try:
                    with np.errstate(all="ignore"), contextlib.redirect_stdout(io.StringIO()):
                        r = ref_h.stitchPanorama(Q.copy(), timg.copy(), H, blending=blending, blendrate=0.35)
                    out[name + "_outcome"] = np.array("ok"); out[name + "_out"] = np.asarray(r)
                except Exception as e:      # noqa: BLE001 -- the type is the datum
                    out[name + "_outcome"] = np.array(type(e).__name__)
                print(name, str(out[name + "_outcome"]), out.get(name + "_out", np.zeros(0)).shape)
    out["names"] = np.array(names)
    save("g16_stitch_geometry", **out)


def model_cases(ptsA, ptsB):
    """Inputs of g17: (name, val or None, op, args) for HomoModel's helpers at the corners of their input space."""
    rng = np.random.default_rng(17)
    A, B = ptsA[:40], ptsB[:40]
    m = ref_r.HomoModel(th=5, d=50, n=4)
    v32 = np.asarray(m.fit(A[:4].T, B[:4].T), dtype=np.float32)
    m6 = ref_r.HomoModel(th=5, d=50, n=4)
    v64 = np.asarray(m6.fit(A[:9].T, B[:9].T, collective=True))
    vals = {"v32": v32, "v64": v64, "zerorow": np.array([[1, 0, 2], [0, 1, 3], [0, 0, 0]], np.float32),
            "singular": np.array([[1, 2, 3], [2, 4, 6], [0, 0, 1]], np.float32), "nanval": np.array([[1, 0, np.nan], [0, 1, 0], [0, 0, 1]], np.float32)}
    X2, Y2 = A.T.copy(), B.T.copy()                                            # 2 x 40 float32
    X3 = np.vstack([X2, rng.uniform(0.5, 2.0, (1, 40)).astype(np.float32)])    # third row is NOT 1
    Y3 = np.vstack([Y2, np.ones((1, 40), np.float32)])
    inputs = {"x2": X2, "x3": X3, "x2f64": X2.astype(np.float64), "x3f64": X3.astype(np.float64), "x2int": X2.astype(np.int32),
              "empty2": np.zeros((2, 0), np.float32), "one2": X2[:, :1].copy(), "x4rows": np.vstack([X3, X3[:1]]), "x1row": X2[:1].copy()}
    cases = []
    for vn, val in vals.items():
        for xn in (("x2", "x3", "x2f64", "x3f64", "x2int", "empty2", "one2", "x4rows", "x1row") if vn == "v32" else ("x2", "x3", "x3f64")):
            cases.append(("fwd_%s_%s" % (vn, xn), val, "fwd", (inputs[xn],)))
            cases.append(("reproj_%s_%s" % (vn, xn), val, "reproj", (inputs[xn],)))
        for meth in ("fwd", "backward", "reproj", "nonsense"):
            cases.append(("loss_%s_%s" % (vn, meth), val, "computeLoss", (X2, Y2, meth)))
    cases.append(("loss_v32_x3y3", v32, "computeLoss", (X3, Y3, "fwd")))
    cases.append(("loss_v32_x3y2_reproj", v32, "computeLoss", (X3, Y2, "reproj")))
    cases.append(("loss_v32_f64", v32, "computeLoss", (X2.astype(np.float64), Y2.astype(np.float64), "reproj")))
    cases.append(("dist_basic", v32, "dist", (X2, Y2)))
    cases.append(("dist_shape", v32, "dist", (X3, Y2)))
    for name, xa, ya, coll in (("fit_2x4", X2[:, :4], Y2[:, :4], False), ("fit_3x4", X3[:, :4], Y3[:, :4], False), ("fit_2x5", X2[:, :5], Y2[:, :5], False),
                               ("fit_2x9_coll", X2[:, :9], Y2[:, :9], True), ("fit_3x9_coll", X3[:, :9], Y3[:, :9], True), ("fit_2x4_coll", X2[:, :4], Y2[:, :4], True),
                               ("fit_2x3", X2[:, :3], Y2[:, :3], False), ("fit_mixed_rows", X3[:, :4], Y2[:, :4], False), ("fit_1row", X2[:1, :4], Y2[:1, :4], False),
                               ("fit_repeated", X2[:, [0, 1, 1, 2]], Y2[:, [0, 1, 1, 2]], False), ("fit_f64", X2[:, :4].astype(np.float64), Y2[:, :4].astype(np.float64), False)):
        cases.append((name, None, "fit", (xa, ya, coll)))
    return cases


def g17(ptsA, ptsB):
    """HomoModel's helpers (ransac.py:30-98) at the corners of their input space: float32 / float64 `val`, a zero bottom row, a
    singular and a NaN `val`; 2- and 3-row inputs whose third row is kept, float64 and integer inputs, no point, one point,
    wrong row counts; every `method` of computeLoss and an unknown one (SystemExit); fit on 2- / 3-row samples, collective
    refits, wrong sizes, a repeated point.  The returned array (values AND dtype) or the exception type."""
    import contextlib
    import io
    out = {"numpy_version": np.array(np.__version__)}
    names = []
    for name, val, op, args in model_cases(ptsA, ptsB):
        names.append(name)
        out[name + "_op"] = np.array(op)
        if val is not None: out[name + "_val"] = val
        for i, a in enumerate(args):
            out[name + "_a%d" % i] = np.array(a) if isinstance(a, (str, bool)) else a
        m = ref_r.HomoModel(th=5, d=50, n=4)
        if val is not None: m.val = val.copy()
        try:
            with np.errstate(all="ignore"), contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                r = getattr(m, op)(*[a.copy() if isinstance(a, np.ndarray) else a for a in args])
            out[name + "_outcome"] = np.array("ok"); out[name + "_out"] = np.asarray(r)
        except BaseException as e:      # noqa: BLE001 -- SystemExit included: the type is the datum
            out[name + "_outcome"] = np.array(type(e).__name__)
        print(name, str(out[name + "_outcome"]), getattr(out.get(name + "_out"), "dtype", ""), getattr(out.get(name + "_out"), "shape", ""))
    out["names"] = np.array(names)
    save("g17_model_helpers", **out)


def host_cases(ptsA, ptsB):
    """Inputs of g18: (name, function, args, kwargs) for the host-side builders / solvers, addAlpha and the two interpolators on
    caller-computed coordinates."""
    rng = np.random.default_rng(18)
    A, B = ptsA[:12].copy(), ptsB[:12].copy()
    cases = []
    for tag, u, v in (("f32", A, B), ("f64", A.astype(np.float64), B.astype(np.float64)), ("int", A.astype(np.int32), B.astype(np.int32)),
                      ("big", A * 1e4, B * 1e4), ("same", np.tile(A[:1], (12, 1)), np.tile(B[:1], (12, 1))),
                      ("collinear", np.stack([np.arange(12.0), 2 * np.arange(12.0)], 1).astype(np.float32), B)):
        for fn in ("calc_corresp", "calc_correspLinear"):
            cases.append(("%s_%s" % (fn, tag), fn, (u[:4], v[:4]), {}))
        for fn in ("calc_correspCollective", "calc_correspLinearCollective"):
            cases.append(("%s_%s" % (fn, tag), fn, (u, v), {}))
        for fn in ("calcHomography", "calcHomographyLinear"):
            cases.append(("%s_%s" % (fn, tag), fn, (u[:4], v[:4]), {}))
            cases.append(("%s_%s_coll" % (fn, tag), fn, (u, v), {"collective": True}))
    cases.append(("calcHomography_3pts", "calcHomography", (A[:3], B[:3]), {}))
    cases.append(("calcHomography_5pts_noncoll", "calcHomography", (A[:5], B[:5]), {}))
    cases.append(("calcHomographyLinear_coll_4", "calcHomographyLinear", (A[:4], B[:4]), {"collective": True}))
    cases.append(("calcHomographyLinear_coll_3", "calcHomographyLinear", (A[:3], B[:3]), {"collective": True}))
    cases.append(("calcHomography_nan", "calcHomography", (np.where(np.arange(8).reshape(4, 2) == 3, np.nan, A[:4]).astype(np.float32), B[:4]), {}))
    img3 = rng.integers(0, 256, (7, 9, 3), dtype=np.uint8)
    img4 = rng.uniform(0, 255, (7, 9, 4)).astype(np.float32)
    for meth in ("Rate", "Gradient", "Other"):
        for ao in (False, True):
            cases.append(("addAlpha_%s_%d" % (meth, ao), "addAlpha", (img3,), {"method": meth, "rate": 0.3, "alphaOnly": ao}))
    cases.append(("addAlpha_grad_right", "addAlpha", (img3,), {"method": "Gradient", "direction": ref_h.BLENDDIR.RIGHT, "alphaOnly": True}))
    cases.append(("addAlpha_grad_top", "addAlpha", (img3,), {"method": "Gradient", "direction": ref_h.BLENDDIR.TOP, "alphaOnly": True}))
    cases.append(("addAlpha_4ch", "addAlpha", (img4,), {"method": "Rate"}))
    # interpolators on caller-computed coordinates (3 x N, dehomogenised), mh x mw = 4 x 6
    base = np.stack([rng.uniform(-1.5, 9.5, 24), rng.uniform(-1.5, 7.5, 24), np.ones(24)])
    inside = np.stack([rng.uniform(0, 7.9, 24), rng.uniform(0, 5.9, 24), np.ones(24)])
    edge = inside.copy(); edge[0, 3] = 8.0; edge[1, 5] = 3.0
    edge_y = inside.copy(); edge_y[1, 7] = 6.0
    halves = inside.copy(); halves[0, :6] = [0.5, 1.5, 2.5, 7.5, 7.49999, 0.49999]; halves[1, :6] = [0.5, 5.5, 2.5, 3.5, 5.49999, 0.0]
    nanc = inside.copy(); nanc[0, 2] = np.nan
    infc = inside.copy(); infc[0, 2] = np.inf; infc[1, 4] = -np.inf
    neg = inside.copy(); neg[0, 1] = -0.3; neg[1, 2] = -0.7; neg[0, 3] = -1.2
    for cn, z in (("mixed", base), ("inside", inside), ("edge_x", edge), ("edge_y", edge_y), ("halves", halves), ("nan", nanc), ("inf", infc), ("neg", neg)):
        for fn in ("nearestNeighbor", "bilinear"):
            for tn, im in (("u8", img3), ("f32x4", img4)):
                cases.append(("%s_%s_%s" % (fn, cn, tn), fn, (z, im, 7, 9, 4, 6), {}))
    for fn in ("nearestNeighbor", "bilinear"):      # a bound (h, w) smaller and larger than the image
        cases.append(("%s_smallbound" % fn, fn, (inside, img3, 5, 6, 4, 6), {}))
        cases.append(("%s_bigbound" % fn, fn, (base + 0.0, img3, 12, 14, 4, 6), {}))
    return cases


def g18(ptsA, ptsB):
    """The host-side builders and solvers (homography.py:4-105), addAlpha (250-286) and the two interpolators on coordinates the
    caller computed (108-138) at the corners of their input space: float32 / float64 / integer points, huge coordinates, twelve
    equal points, collinear points, too few points, a NaN coordinate; every addAlpha method, direction and channel count;
    coordinates inside, outside, exactly on the last column / row, on .5, NaN and Inf, bounds smaller and larger than the image.
    Returned arrays (values and dtype), what the call did to its arguments (the interpolators blank texel (0,0) and bilinear
    writes zeros into the caller's coordinates), or the exception type."""
    import contextlib
    import io
    out = {"numpy_version": np.array(np.__version__)}
    names = []
    for name, fn, args, kw in host_cases(ptsA, ptsB):
        names.append(name)
        out[name + "_fn"] = np.array(fn)
        for i, a in enumerate(args):
            out[name + "_a%d" % i] = np.asarray(a)
        for k_, v in kw.items():
            out[name + "_kw_" + k_] = np.asarray(v)
        call = [a.copy() if isinstance(a, np.ndarray) else a for a in args]
        try:
            with np.errstate(all="ignore"), contextlib.redirect_stdout(io.StringIO()):
                r = getattr(ref_h, fn)(*call, **kw)
            out[name + "_outcome"] = np.array("ok")
            for i, part in enumerate(r if isinstance(r, tuple) else (r,)):
                out[name + "_out%d" % i] = np.asarray(part)
            for i, (a, c_) in enumerate(zip(args, call)):      # side effects on the arguments
                if isinstance(a, np.ndarray) and not np.array_equal(a, c_, equal_nan=True):
                    out[name + "_after%d" % i] = c_
        except Exception as e:      # noqa: BLE001 -- the type is the datum
            out[name + "_outcome"] = np.array(type(e).__name__)
        print(name, str(out[name + "_outcome"]), getattr(out.get(name + "_out0"), "dtype", ""), getattr(out.get(name + "_out0"), "shape", ""),
              [k_ for k_ in out if k_.startswith(name + "_after")])
    out["names"] = np.array(names)
    save("g18_host_helpers", **out)


def g19(ptsA, ptsB):
    """Nearly singular hypotheses under 'backward' / 'reproj' (ransac.py:74 inverts every hypothesis with numpy.linalg.inv): cluster
    problems -- samples drawn from two or three tight clusters give H that numpy's float64 LAPACK inverse and any other float64
    elimination round apart, and with them the losses.  Case 0 is case 266 of `tests/soak_settle.py 2000 109` (the run that found
    it: same winner and count, a different inlier list); the others are more of its kind.  The reference's own runs."""
    rng = np.random.default_rng(109)
    HS = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])

    def project(G, noise):
        P = np.concatenate([G, np.ones((len(G), 1))], 1) @ HS.T
        return P[:, :2] / P[:, 2:3] + rng.normal(0, noise, (len(G), 2))

    def problem(kind):          # tests/soak_settle.py's generator, draw for draw
        if kind == 0:
            M = int(rng.integers(30, 900)); nx, ny = int(rng.integers(2, 20)), int(rng.integers(2, 12))
            G = np.stack([rng.integers(0, nx, M) * rng.uniform(5, 100), rng.integers(0, ny, M) * rng.uniform(5, 100)], 1)
            B = project(G, rng.uniform(0.1, 1.5))
        elif kind == 1:
            M = int(rng.integers(30, 600)); c = rng.uniform(0, 2000, (int(rng.integers(3, 9)), 2))
            G = c[rng.integers(0, len(c), M)] + rng.normal(0, rng.choice([0.0, 0.01, 1.0]), (M, 2))
            B = project(G, 0.5)
        elif kind == 2:
            G = ptsA.astype(np.float64) * rng.choice([1.0, 8.0, 100.0]); B = ptsB.astype(np.float64) * (G[0, 0] / ptsA[0, 0])
        elif kind == 3:
            M = int(rng.integers(5, 13)); G = rng.uniform(0, 500, (M, 2)); B = project(G, 0.5)
        else:
            M = int(rng.integers(50, 1500)); G = rng.uniform(0, rng.choice([1e3, 1e4, 1e5]), (M, 2)); B = project(G, 1.0)
        out = rng.random(len(G)) < rng.choice([0.0, 0.2, 0.5, 0.8])
        B = B.copy(); B[out] = rng.uniform(0, max(1.0, float(np.abs(B).max())), (int(out.sum()), 2))
        return G.astype(np.float32), B.astype(np.float32)
    for case in range(267):
        A, B = problem(case % 5)
        scale = max(1.0, float(np.abs(A).max()) / 1000.0)
        th = float(rng.choice([1, 3, 5])) * scale
        d = int(rng.choice([20, 40, 70, 95])); k = int(rng.integers(50, 400)); n = int(rng.choice([4, 4, 4, 6]))
        m = str(rng.choice(["fwd", "backward", "reproj"]))
        seed = int(rng.integers(0, 1 << 30))
    runs = [(A, B, th, d, k, n, m, seed)]
    rng2 = np.random.default_rng(1909)
    for t in range(11):        # more cluster problems, 'backward' and 'reproj', early exits and running bests
        M = int(rng2.integers(60, 400)); c = rng2.uniform(0, 2000, (int(rng2.integers(3, 6)), 2))
        G = c[rng2.integers(0, len(c), M)] + rng2.normal(0, [0.0, 0.01, 1.0][t % 3], (M, 2))
        P = np.concatenate([G, np.ones((M, 1))], 1) @ HS.T
        Bq = P[:, :2] / P[:, 2:3] + rng2.normal(0, 0.5, (M, 2))
        o = rng2.random(M) < [0.0, 0.3][t % 2]
        Bq[o] = rng2.uniform(0, 2000, (int(o.sum()), 2))
        runs.append((G.astype(np.float32), Bq.astype(np.float32), float(max(1.0, np.abs(G).max() / 1000.0) * [1, 3, 5][t % 3]), [20, 40, 95][t % 3],
                     int(rng2.integers(100, 500)), 4, ["backward", "reproj"][t % 2], int(rng2.integers(0, 1 << 30))))
    out = {}
    import contextlib
    import io
    for i, (A_, B_, th_, d_, k_, n_, m_, seed_) in enumerate(runs):
        np.random.seed(seed_)
        key = "c%d" % i
        out[key + "_A"] = A_; out[key + "_B"] = B_
        out[key + "_par"] = np.array([th_, d_, k_, n_, seed_], dtype=np.float64); out[key + "_method"] = np.array(m_)
        try:
            with np.errstate(all="ignore"), contextlib.redirect_stdout(io.StringIO()):
                model = ref_r.HomoModel(th=th_, d=d_, n=n_)
                H, inl, cnt = ref_r.RANSAC(model, k=k_).run([A_.T, B_.T], method=m_)
            out[key + "_outcome"] = np.array("ok")
            out[key + "_H"] = np.asarray(H, np.float64); out[key + "_inliers"] = inl[0].astype(np.int64); out[key + "_count"] = np.int64(cnt)
        except Exception as e:      # noqa: BLE001
            out[key + "_outcome"] = np.array(type(e).__name__)
        out[key + "_next_draw"] = np.int64(np.random.randint(0, 1 << 30))
        print(key, len(A_), th_, d_, k_, m_, str(out[key + "_outcome"]), int(out.get(key + "_count", -1)))
    out["n_cases"] = np.int64(len(runs))
    save("g19_near_singular_inverse", **out)


def main():
    which = set(sys.argv[1:]) or {"g1", "g2", "g4", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18", "g19"}
    ptsA, ptsB = load_matches()
    save("matchespoints", ptsA=ptsA, ptsB=ptsB)
    if "g1" in which: g1()
    if "g2" in which: g2_g3(ptsA, ptsB)
    if "g4" in which: g4_g5(ptsA, ptsB)
    if "g6" in which: g6()
    if "g7" in which: g7()
    if "g8" in which: g8()
    if "g9" in which: g9(ptsA, ptsB)
    if "g10" in which: g10(ptsA, ptsB)
    if "g11" in which: g11()
    if "g12" in which: g12()
    if "g13" in which: g13(ptsA, ptsB)
    if "g14" in which: g14()
    if "g15" in which: g15()
    if "g16" in which: g16()
    if "g17" in which: g17(ptsA, ptsB)
    if "g18" in which: g18(ptsA, ptsB)
    if "g19" in which: g19(ptsA, ptsB)


if __name__ == "__main__":
    main()
