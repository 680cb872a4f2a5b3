"""Pin the CPU oracle (oracle/rwh_oracle.py) bit-for-bit against fixtures the
REFERENCE produced (tests/golden/make_golden.py).  CPU only.

Every comparison here is exact (array_equal / sha256): the oracle performs the
same numpy operations as the reference, so on the numpy/OpenBLAS build the
fixtures were made with (numpy 2.2.6, OpenBLAS 0.3.29) there is no tolerance.
The final N-point refit H goes through float32 normal equations whose last bits
depend on the BLAS kernel; it is compared exactly first and, only if the host
BLAS differs from the fixture's, within 1e-3 relative.
"""
import hashlib

import numpy as np
import pytest

from conftest import load_golden
from oracle import rwh_oracle as orc


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def same(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b)


def check_digest(z, prefix, img):
    assert tuple(z[prefix + "_shape"]) == img.shape
    assert np.array_equal(img.reshape(-1)[z[prefix + "_pick"]], z[prefix + "_vals"])
    assert str(z[prefix + "_sha256"]) == sha(img)


def test_g1_fourpoint_solvers():
    z = load_golden("g1_fourpoint")
    u, v = z["u"].T[:, :2], z["v"].T[:, :2]
    A, b = orc.linear_system(u, v)
    assert same(A, z["A"]) and same(b, z["b"])
    assert same(orc.dlt_matrix(u, v), z["mat"])
    assert same(orc.calc_homography_linear(u, v), z["H_linear"])
    assert same(orc.calc_homography(u, v), z["H_dlt"])
    u32, v32 = u.astype(np.float32), v.astype(np.float32)
    assert same(orc.dlt_matrix(u32, v32), z["mat_f32in"])
    assert same(orc.calc_homography(u32, v32), z["H_dlt_f32in"])
    # SURVEY 8(c) G1 literal values
    np.testing.assert_allclose(z["H_linear"][0], [0.8576602936, -0.1106677055, -37.34912109], rtol=1e-9)


@pytest.mark.parametrize("name", ["g2_hyp_seed0", "g3_hyp_seed7"])
def test_g2_g3_per_hypothesis(name, matches):
    z = load_golden(name)
    ptsA, ptsB = matches
    X, Y = ptsA.T, ptsB.T
    np.random.seed(int(z["seed"]))
    idx = np.random.randint(0, 185, (10000, 4))
    assert np.array_equal(idx, z["idx"])          # RNG stream parity (ransac.py:177)
    sub = slice(0, 1500)                            # full table takes ~15 s; 1500 rows pin the arithmetic
    for method in ("fwd", "backward", "reproj"):
        Hs, counts = orc.ransac_table(X, Y, idx[sub], th=5, method=method)
        assert np.array_equal(Hs.view(np.uint32), z["H"][sub].view(np.uint32))
        assert np.array_equal(counts, z["counts_" + method][sub])
    c = z["counts_fwd"].astype(np.int64)
    w, early = orc.select_winner(c, 185 * 70 / 100 + 4)
    assert not early and w == int(z["winner"]) and c[w] == int(z["winner_count"])
    err = orc.compute_loss(z["H"][w].reshape(3, 3), X, Y, "fwd")
    assert np.array_equal(np.where(err < 5)[0], z["winner_inliers"])


def test_g2_g3_known_answers():
    z0, z7 = load_golden("g2_hyp_seed0"), load_golden("g3_hyp_seed7")
    assert (int(z0["winner"]), int(z0["winner_count"]), int(z0["winner_ties"])) == (6354, 121, 1)
    assert (int(z7["winner"]), int(z7["winner_count"]), int(z7["winner_ties"])) == (564, 120, 17)
    assert int(z0["degenerate"].sum()) == 373


def test_g4_g5_ransac_runs(matches):
    z = load_golden("g4_ransac_runs")
    ptsA, ptsB = matches
    for key in [str(k) for k in z["cases"]]:
        _, s, th, d, k, method = key.split("_")
        np.random.seed(int(s[1:]))
        H, inl, cnt, it = orc.ransac_run(ptsA.T, ptsB.T, th=int(th[2:]), d=int(d[1:]), n=4, k=int(k[1:]), method=method)
        assert int(cnt) == int(z[key + "_count"]), key
        assert np.array_equal(inl[0], z[key + "_inliers"]), key
        if not np.array_equal(H, z[key + "_H"]):
            np.testing.assert_allclose(H, z[key + "_H"], rtol=1e-3, atol=1e-6, err_msg=key)
        if key.startswith("ge_"):
            assert it < int(k[1:]) - 1   # early exit actually fired


def test_g6_small_warps():
    z = load_golden("g6_small_warps")
    for iname in ("noise", "ramp"):
        img = z["img_" + iname]
        for hn in [str(h) for h in z["H_names"]]:
            H = z["H_" + hn]
            for conv in ("nn", "bilinear"):
                o, mx, my = orc.wrap_perspective(img.copy(), H, convert=conv)
                k = "wp_%s_%s_%s" % (iname, hn, conv)
                assert same(o, z[k]), k
                assert (mx, my) == tuple(z[k + "_org"])
            o, _, _ = orc.transform_image_h(img.copy(), H)
            assert same(o, z["tih_%s_%s" % (iname, hn)])
        o, mx, my = orc.wrap_perspective(img.copy(), z["H_rot"], convert="bilinear", boundary=1)
        assert same(o, z["wpb_%s_rot_bilinear" % iname]) and (mx, my) == tuple(z["wpb_%s_rot_bilinear_org" % iname])
        hs, ws, _ = img.shape
        for conv in ("nn", "bilinear"):
            for hn in ("bench", "rot"):
                o, _, _ = orc.wrap_perspective_scan(img.copy(), z["H_" + hn], (hs - 8, ws - 16), convert=conv)
                assert same(o, z["scan_%s_%s_%s" % (iname, hn, conv)])
        rgba = orc.add_alpha_rate(img.copy(), 0.2)
        assert same(rgba, z["rgba_" + iname])
        for conv in ("nn", "bilinear"):
            o, _, _ = orc.wrap_perspective(rgba.copy(), z["H_notebook"], convert=conv)
            assert same(o, z["wp4_%s_notebook_%s" % (iname, conv)])


def test_warp_mutates_callers_image():
    """homography.py:112-116 / 126-130: texel (0,0) of the caller's array is zeroed."""
    z = load_golden("g6_small_warps")
    img = z["img_noise"].copy()
    assert img[0, 0].any()
    orc.wrap_perspective(img, z["H_bench"], convert="bilinear")
    assert not img[0, 0].any()


def test_g7_notebook():
    z = load_golden("g7_notebook")
    img = load_golden("img_notebook")["img"]
    assert img.shape == (571, 1023, 3)
    o, mx, my = orc.wrap_perspective(img.copy(), z["H"], convert="bilinear")
    assert o.shape == (1607, 1251, 3) and (mx, my) == (-65, -168) == tuple(z["wp_bilinear_org"])
    check_digest(z, "wp_bilinear", o)
    o, _, _ = orc.wrap_perspective(img.copy(), z["H"], convert="nn")
    check_digest(z, "wp_nn", o)
    o = orc.transform_image(img.copy(), z["u"], z["v"])
    assert o.shape == (781, 401, 3)
    check_digest(z, "ti", o)
    check_digest(z, "ti_nn", orc.transform_image(img.copy(), z["u"], z["v"], method="nn"))
    o = orc.transform_image(img.copy(), z["u"], z["v_a4"], box=[1188, 840])
    assert o.shape == (1188, 840, 3)
    check_digest(z, "scan_a4", o)
    check_digest(z, "scan_a4_nn", orc.transform_image(img.copy(), z["u"], z["v_a4"], box=[1188, 840], method="nn"))


def test_g8_stitch():
    z = load_golden("g8_stitch")
    f = load_golden("img_foto1")
    A, B = f["A"], f["B"]
    o, mx, my = orc.transform_image_h(A.copy(), z["H_notebook"])
    assert o.shape == (822, 1199, 3) and (mx, my) == (434, -90) == tuple(z["tih_org"])
    check_digest(z, "tih", o)
    o = orc.stitch_panorama(B.copy(), A.copy(), z["H_notebook"])
    assert o.shape == (822, 1633, 3)
    check_digest(z, "stitch_paste", o)
    check_digest(z, "stitch_rate", orc.stitch_panorama(B.copy(), A.copy(), z["H_notebook"], blending="Rate", blendrate=0.2))
    o = orc.stitch_panorama(B.copy(), A.copy(), z["H_g5"], blending="Rate", blendrate=0.2)
    assert o.shape == (788, 1647, 3)
    check_digest(z, "stitch_g5_rate", o)
    z = load_golden("g11_stitch_gradient")      # the alpha ramp of homography.py:259-266
    check_digest(z, "stitch_gradient", orc.stitch_panorama(B.copy(), A.copy(), load_golden("g8_stitch")["H_notebook"], blending="Gradient"))
    check_digest(z, "stitch_g5_gradient", orc.stitch_panorama(B.copy(), A.copy(), load_golden("g8_stitch")["H_g5"], blending="Gradient"))


def test_oracle_ransac_illcond_config4x8_and_n6(matches):
    """Round-3 fixtures, written by the unmodified reference: g12 (lattice / cluster problems: ill-conditioned samples without
    a repeated index), g13 (BASELINE config 4's RANSAC at the x8 scale; HomoModel(n = 6): six indices per iteration, fit on
    the first four, exit at d + 6) -- the oracle reproduces winner, count, inlier list and the generator's position."""
    z = load_golden("g12_illcond")
    for key in [str(c) for c in z["cases"]]:
        tag, s, th, d, k, m = key.split("_")
        A, B = z["ptsA_" + tag], z["ptsB_" + tag]
        np.random.seed(int(s[1:]))
        with np.errstate(all="ignore"):
            H, inl, cnt, it = orc.ransac_run(A.T, B.T, th=int(th[2:]), d=int(d[1:]), n=4, k=int(k[1:]), method=m)
        assert int(cnt) == int(z[key + "_count"]) and it == int(z[key + "_winner"]), key
        assert np.array_equal(inl[0], z[key + "_inliers"]), key
        assert np.allclose(H, z[key + "_H"], rtol=0, atol=0), key
    ptsA, ptsB = matches
    g = load_golden("g13_config4_x8")
    np.random.seed(0)
    H, inl, cnt, it = orc.ransac_run((ptsA * 8).T, (ptsB * 8).T, th=32, d=95, n=4, k=1500, method="fwd")
    assert int(cnt) == int(g["count"]) == 114 and it == int(g["winner"]) and np.array_equal(inl[0], g["inliers"])
    assert np.array_equal(H, g["H"])
    for seed, d in ((0, 70), (3, 50)):
        key = "n6_s%d_d%d" % (seed, d)
        np.random.seed(seed)
        H, inl, cnt, it = orc.ransac_run(ptsA.T, ptsB.T, th=5, d=d, n=6, k=1000, method="fwd")
        assert int(cnt) == int(g[key + "_count"]) and np.array_equal(inl[0], g[key + "_inliers"]) and np.array_equal(H, g[key + "_H"])
        assert np.random.randint(0, 1 << 30) == int(g[key + "_next_draw"])
    with pytest.raises(IndexError):
        orc.calc_homography(ptsA[:3], ptsB[:3])


def test_oracle_ransac_edge_cases_g14():
    """g14: what the unmodified reference does at the corners of RANSAC.run's input space -- k = 0 (UnboundLocalError), no
    hypothesis with an inlier (np.where(None): ValueError from numpy 2.1 on), fewer than four correspondences, n < 4, a NaN
    coordinate (LinAlgError at the iteration that samples it), Inf coordinates (no error), ragged inputs ...: the oracle gives
    the same count, inlier list and generator position, or raises the same exception type."""
    import contextlib
    import io
    g = load_golden("g14_edge_cases")
    same_numpy = str(g["numpy_version"]).split(".")[:2] == np.__version__.split(".")[:2]
    for name in [str(n) for n in g["names"]]:
        A, B = g[name + "_A"], g[name + "_B"]
        th, d, n, k = g[name + "_par"]
        for m in ("fwd", "reproj"):
            key = "%s_%s" % (name, m)
            want = str(g[key + "_outcome"])
            if want == "ValueError" and not same_numpy and A.shape[0] > 0:
                continue                      # np.where(None) changed its mind between numpy versions
            np.random.seed(4242)
            with np.errstate(all="ignore"), contextlib.redirect_stdout(io.StringIO()):
                try:
                    H, inl, cnt, _ = orc.ransac_run(A.T, B.T, th=th, d=d, n=int(n), k=int(k), method=m)
                    got = "ok"
                except Exception as e:      # noqa: BLE001 -- the type is what is compared
                    got = type(e).__name__
            assert got == want, (key, got, want)
            assert int(np.random.randint(0, 1 << 30)) == int(g[key + "_next_draw"]), key
            if want == "ok":
                assert int(cnt) == int(g[key + "_count"]) and np.array_equal(inl[0], g[key + "_inliers"]), key
                assert np.allclose(H, g[key + "_H"], rtol=1e-6, atol=1e-9, equal_nan=True), key


def _g15_call(mod_fns, g, name):
    """Run case `name` of g15 through a module's (wrapPerspective, wrapPerspectiveScan, transformImage, transformImageH)."""
    fn = str(g[name + "_fn"])
    img = g[name + "_img"].copy()
    kw = {}
    for key in g.files:
        if key.startswith(name + "_arg_"):
            v = g[key]
            k_ = key[len(name) + 5:]
            kw[k_] = str(v) if v.dtype.kind in "US" else (tuple(int(t) for t in v) if k_ in ("res", "box") else v)
    return mod_fns[fn](img, **kw)


def test_oracle_warp_edge_cases_g15():
    """g15: what the unmodified reference's warp entry points do at the corners of their input space -- bilinear on a coordinate
    exactly ON the last column / row (identity, integer shifts, pure scales, rot90, mirrors: IndexError), a scan `res` beyond the
    image (IndexError), a singular H (LinAlgError), a NaN entry (ValueError), 2 x 2 and 3 x 3 images, transformImage with and
    without a box: the oracle returns the same array bit for bit or raises the same exception type."""
    g = load_golden("g15_warp_edge_cases")
    fns = {"wrapPerspective": orc.wrap_perspective, "wrapPerspectiveScan": orc.wrap_perspective_scan,
           "transformImage": orc.transform_image, "transformImageH": orc.transform_image_h}
    for name in [str(n) for n in g["names"]]:
        want = str(g[name + "_outcome"])
        try:
            with np.errstate(all="ignore"):
                r = _g15_call(fns, g, name)
            got = "ok"
        except Exception as e:      # noqa: BLE001 -- the type is what is compared
            got = type(e).__name__
        assert got == want, (name, got, want)
        if want == "ok":
            arr = r[0] if isinstance(r, tuple) else r
            assert arr.dtype == g[name + "_out"].dtype and np.array_equal(arr, g[name + "_out"]), name
            if isinstance(r, tuple):
                assert [int(r[1]), int(r[2])] == g[name + "_origin"].tolist(), name


def _g16_cases(g):
    for name in [str(n) for n in g["names"]]:
        hn, bn, tn = name.rsplit("_", 2)
        blending = {"paste": False, "rate": "Rate", "grad": "Gradient", "true": True}[bn]
        yield name, g["Q"].copy(), g[tn].copy(), g[name + "_H"], blending, str(g[name + "_outcome"])


def test_oracle_stitch_geometry_g16():
    """g16: stitchPanorama of the unmodified reference on small images for every branch of its canvas geometry (the warped image
    left / right of, above / below, inside and around the query image), every `blending` value its code distinguishes, and the
    identity (IndexError from the bilinear warp): the oracle's canvas is the same array, or it raises the same exception type."""
    import contextlib
    import io
    g = load_golden("g16_stitch_geometry")
    for name, Q, T, H, blending, want in _g16_cases(g):
        try:
            with np.errstate(all="ignore"), contextlib.redirect_stdout(io.StringIO()):
                r = orc.stitch_panorama(Q, T, H, blending=blending, blendrate=0.35)
            got = "ok"
        except Exception as e:      # noqa: BLE001 -- the type is what is compared
            got = type(e).__name__
        assert got == want, (name, got, want)
        if want == "ok":
            assert r.dtype == g[name + "_out"].dtype and np.array_equal(r, g[name + "_out"]), name


def _g17_cases(g):
    for name in [str(n) for n in g["names"]]:
        args = []
        i = 0
        while name + "_a%d" % i in g.files:
            a = g[name + "_a%d" % i]
            args.append(str(a) if a.dtype.kind in "US" else bool(a) if a.dtype == np.bool_ and a.ndim == 0 else a.copy())
            i += 1
        yield name, str(g[name + "_op"]), (g[name + "_val"].copy() if name + "_val" in g.files else None), args, str(g[name + "_outcome"])


def _same_array(got, want):
    got = np.asarray(got)
    return got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want, equal_nan=True)


def test_oracle_model_helpers_g17():
    """g17: HomoModel.fwd / reproj / dist / computeLoss / fit of the unmodified reference at the corners of their input space
    (float32 and float64 `val`, zero bottom row, singular, NaN; 2- and 3-row inputs, float64 and integer inputs, no point, one
    point; every method and an unknown one; collective refits, a repeated point): the oracle's functions return the same array,
    values and dtype, or fail the same way.  (The reference's input assertions live in its class, not in the oracle's free
    functions: those cases are the product's to reproduce, tests/test_gpu_parity.py.)"""
    g = load_golden("g17_model_helpers")
    for name, op, val, args, want in _g17_cases(g):
        if want == "AssertionError":
            continue
        try:
            with np.errstate(all="ignore"):
                if op == "fwd": r = orc.project_fwd(val, *args)
                elif op == "reproj": r = orc.project_back(val, *args)
                elif op == "dist": r = orc.l2_dist(*args)
                elif op == "computeLoss": r = orc.compute_loss(val, *args)
                else: r = orc.fit_all(args[0], args[1]) if args[2] else orc.fit_minimal(args[0], args[1])
            got = "ok"
        except BaseException as e:      # noqa: BLE001 -- SystemExit included
            got = type(e).__name__
        assert got == want, (name, got, want)
        if want == "ok":
            assert _same_array(r, g[name + "_out"]), (name, np.asarray(r).dtype, g[name + "_out"].dtype)


def _g18_cases(g):
    for name in [str(n) for n in g["names"]]:
        args, i = [], 0
        while name + "_a%d" % i in g.files:
            a = g[name + "_a%d" % i]
            args.append(a.copy() if a.ndim else a.item())
            i += 1
        kw = {}
        for key in g.files:
            if key.startswith(name + "_kw_"):
                v = g[key]
                kw[key[len(name) + 4:]] = str(v) if v.dtype.kind in "US" else v.item()
        outs, i = [], 0
        while name + "_out%d" % i in g.files:
            outs.append(g[name + "_out%d" % i]); i += 1
        after = {j: g[name + "_after%d" % j] for j in range(len(args)) if name + "_after%d" % j in g.files}
        yield name, str(g[name + "_fn"]), args, kw, str(g[name + "_outcome"]), outs, after


def _g18_check(name, r, args, outs, after, wrong):
    parts = r if isinstance(r, tuple) else (r,)
    if len(parts) != len(outs) or not all(_same_array(p_, o) for p_, o in zip(parts, outs)):
        wrong.append((name, "result", [str(np.asarray(p_).dtype) for p_ in parts], [str(o.dtype) for o in outs]))
    for j, a in enumerate(args):            # what the call did to its arguments
        if isinstance(a, np.ndarray):
            want = after.get(j)
            if want is not None and not np.array_equal(a, want, equal_nan=True):
                wrong.append((name, "argument %d after the call" % j))


def test_oracle_host_helpers_and_interpolators_g18(matches):
    """g18: the reference's builders / solvers (homography.py:4-105) and its two interpolators on caller-computed coordinates
    (108-138) at the corners of their input space: the oracle returns the same arrays (values and dtype), leaves its arguments
    as the reference leaves them (texel (0,0) blanked; bilinear zeroes the masked coordinates in the caller's array), or raises
    the same exception type.  (addAlpha's variants and the fixed 4-row builders' IndexError are the product's to reproduce.)"""
    g = load_golden("g18_host_helpers")
    table = {"calc_corresp": lambda u, v: orc.dlt_matrix(u, v), "calc_correspCollective": lambda u, v: orc.dlt_matrix(u, v),
             "calc_correspLinear": lambda u, v: orc.linear_system(u, v), "calc_correspLinearCollective": lambda u, v: orc.linear_system(u, v),
             "calcHomography": orc.calc_homography, "calcHomographyLinear": orc.calc_homography_linear,
             "nearestNeighbor": orc.nearest_neighbor, "bilinear": orc.bilinear}
    wrong = []
    for name, fn, args, kw, want, outs, after in _g18_cases(g):
        if fn not in table or name in ("calcHomography_3pts",):
            continue
        try:
            with np.errstate(all="ignore"):
                r = table[fn](*args, **kw)
            got = "ok"
        except Exception as e:      # noqa: BLE001 -- the type is what is compared
            got = type(e).__name__
        if got != want:
            wrong.append((name, got, want))
        elif want == "ok":
            _g18_check(name, r, args, outs, after, wrong)
    assert not wrong, wrong


def _g19_cases(g):
    for i in range(int(g["n_cases"])):
        key = "c%d" % i
        th, d, k, n, seed = g[key + "_par"]
        yield key, g[key + "_A"], g[key + "_B"], float(th), int(d), int(k), int(n), int(seed), str(g[key + "_method"])


def test_oracle_ransac_near_singular_inverse_g19():
    """g19: cluster problems under 'backward' / 'reproj' -- hypotheses whose matrix is nearly singular, where the loss depends on
    HOW numpy.linalg.inv rounds (ransac.py:74).  The oracle calls the same numpy routine: count, inlier list, generator."""
    import contextlib
    import io
    g = load_golden("g19_near_singular_inverse")
    for key, A, B, th, d, k, n, seed, m in _g19_cases(g):
        np.random.seed(seed)
        with np.errstate(all="ignore"), contextlib.redirect_stdout(io.StringIO()):
            H, inl, cnt, _ = orc.ransac_run(A.T, B.T, th=th, d=d, n=n, k=k, method=m)
        assert int(cnt) == int(g[key + "_count"]) and np.array_equal(inl[0], g[key + "_inliers"]), key
        assert int(np.random.randint(0, 1 << 30)) == int(g[key + "_next_draw"]), key
