"""CPU-only checks of the C ABI boundary: the library builds, loads, and exports every symbol
include/rwh.h declares; argument validation works without a GPU; the product refuses to compute
without one (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from ransac_with_homography_amd import _lib
    return _lib.load()


def test_exports_match_header(lib):
    from ransac_with_homography_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "rwh.h")).read()
    declared = set(re.findall(r"RWH_API\s+[\w\s\*]+?\b(rwh_\w+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.rwh_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define RWH_ABI_VERSION (\d+)", hdr).group(1))
    assert lib.rwh_strerror(-1) == b"invalid argument"


def test_argument_validation_without_gpu(lib):
    ih = (ctypes.c_double * 9)(1, 0, 0, 0, 1, 0, 0, 0, 1)
    null = ctypes.c_void_p(0)
    # NULL pointers -> RWH_E_INVALID before anything touches a device
    st = lib.rwh_warp_backward(null, 10, 10, 3, 0, 300, 1, ih, 1, 0., 1., 9., 0., 1., 9., 10, 10, 10, 10, 1,
                               null, 0, 300, 0, 10, 0, null)
    assert st == -1
    assert lib.rwh_dlt4_batched(null, null, 10, null, 4, null, null, null) == -1
    assert lib.rwh_score_count(null, null, null, 10, 4, 5.0, 0, 3, 0, null, null, null, null, null) == -1
    assert lib.rwh_project_points(null, null, 10, 0, null, null) == -1
    assert lib.rwh_ransac_search(null, null, 10, null, 4, 5.0, 0, 3, 0, null, null, null, null, null, 1, null) == -1
    assert lib.rwh_ransac_batched(null, null, null, 2, 10, 4, null, 0, 0, 5.0, 0, null, null, null, null, null, null, 0, null) == -1
    assert lib.rwh_stitch_panorama(null, 8, 8, null, 8, 8, ih, 0, 0, 8, 8, 0, 0, 0, 0, 8, 8, 0, 0.2, null, 0, null) == -1
    # n_h must be 1 or the batch size; bad row ranges; an empty row shard is a no-op even with NULL buffers
    one = ctypes.c_void_p(1)            # non-NULL, never dereferenced: validation comes first
    st = lib.rwh_warp_backward(one, 10, 10, 3, 0, 300, 3, ih, 2, 0., 1., 9., 0., 1., 9., 10, 10, 10, 10, 1, one, 0, 300, 0, 10, 0, null)
    assert st == -1
    st = lib.rwh_warp_backward(one, 10, 10, 3, 0, 300, 1, ih, 1, 0., 1., 9., 0., 1., 9., 10, 10, 10, 10, 1, one, 0, 300, 5, 4, 0, null)
    assert st == -1
    st = lib.rwh_warp_backward(null, 10, 10, 3, 0, 300, 1, ih, 1, 0., 1., 9., 0., 1., 9., 10, 10, 10, 10, 1, null, 0, 300, 4, 4, 0, null)
    assert st == 0


def test_round3_host_entry_points_without_gpu(lib):
    """rwh_ransac_run_layout (pure arithmetic), argument validation of rwh_ransac_run / rwh_host_dlt4_svd / rwh_lab_clock_probe,
    and the host solver itself -- rwh_host_dlt4_svd needs no GPU: it must reproduce the reference's H on golden samples."""
    from ransac_with_homography_amd import _lapack
    off = (ctypes.c_longlong * 27)()
    assert lib.rwh_ransac_run_layout(185, 1500, off, 27) == 27 and lib.rwh_ransac_run_layout(185, 1500, off, 22) == -1
    o = list(off)
    assert o[0] == 0 and all(b >= a for a, b in zip(o[:11], o[1:11])) and all(b >= a for a, b in zip(o[12:18], o[13:19]))
    assert o[5] == o[4] + 4 * 1500 and o[14] == o[13] + 4 * 1500            # counts and flags adjacent: one readback
    assert o[11] > 36 * 1500 * 3 and o[19] > 16 * 1500 + 36 * 1500
    assert o[10] < o[20] < o[11] and o[18] < o[21] < o[19]                  # the inverses of the settled rows: inside both workspaces
    assert o[20] < o[22] < o[23] < o[24] < o[11] and o[21] < o[25] < o[19] and o[26] == o[25] + 4 * 1500   # round 4: candidate rows, count intervals
    assert lib.rwh_ransac_run_layout(185, 1500, off, 21) == -1 and lib.rwh_ransac_run_layout(0, 10, off, 22) == -1
    null = ctypes.c_void_p(0)
    assert lib.rwh_ransac_run(null, null, 185, null, 10, 5.0, 0, 100, 8, null, null, 1, null, null, 0, null, null, null, null) == -1
    assert lib.rwh_score_interval(null, null, 4, null, null, null, 185, 5.0, 1000.0, 1e-6, 1e-5, null, null, null) == -1
    assert lib.rwh_host_dlt4_svd(null, null, 185, null, 4, null, 1, null) == -1
    assert lib.rwh_host_inv3(null, 4, null, null) == -1 and lib.rwh_score_count_inv(null, null, null, null, 4, 4, 1.0, 0, 1, 0, null, null, null, null, null) == -1
    assert lib.rwh_lab_clock_probe(null, 1.0, null) == -1
    addr = _lapack.dgesdd_address()
    if addr is None:
        pytest.skip("numpy's LAPACK symbol not found in this environment")
    g = np.load(os.path.join(ROOT, "tests", "golden", "g2_hyp_seed0.npz"))
    z = np.load(os.path.join(ROOT, "tests", "golden", "matchespoints.npz"))
    pa, pb = np.ascontiguousarray(z["ptsA"]), np.ascontiguousarray(z["ptsB"])
    idx = np.ascontiguousarray(g["idx"][:500], dtype=np.int32)
    out = np.empty((500, 9), np.float32)
    assert lib.rwh_host_dlt4_svd(pa.ctypes.data, pb.ctypes.data, 185, idx.ctypes.data, 500, ctypes.c_void_p(addr), 3, out.ctypes.data) == 0
    assert np.array_equal(out.view(np.uint32), g["H"][:500].view(np.uint32))
    bad = idx.copy(); bad[7, 2] = 185                                              # an index past the table: refused, nothing read
    assert lib.rwh_host_dlt4_svd(pa.ctypes.data, pb.ctypes.data, 185, bad.ctypes.data, 500, ctypes.c_void_p(addr), 3, out.ctypes.data) == -1
    # rwh_host_inv3 == numpy.linalg.inv on float32 3 x 3 matrices, bit for bit -- well-conditioned, pixel-scaled and nearly singular
    # ones (rank 2 plus 1e-9 .. 1e-3 of noise: where the kernels' own elimination and LAPACK's round apart)
    gesv = _lapack.dgesv_address()
    assert gesv is not None
    rng = np.random.default_rng(0)
    n = 5000
    well = (np.eye(3) + rng.normal(0, 0.3, (n, 3, 3))).astype(np.float32)
    u, v, w, x = (rng.normal(0, 1, sh) for sh in ((n, 3, 1), (n, 1, 3), (n, 3, 1), (n, 1, 3)))
    ill = (u @ v + w @ x + 10.0 ** rng.uniform(-9, -3, (n, 1, 1)) * rng.normal(0, 1, (n, 3, 3))).astype(np.float32)
    pix = well.copy(); pix[:, :2, 2] *= 300; pix[:, 2, :2] *= 1e-5
    for mats in (well, ill, pix, g["H"][:2000].reshape(-1, 3, 3).copy()):
        mats = np.ascontiguousarray(mats[np.isfinite(mats).all(axis=(1, 2))])
        got = np.empty_like(mats)
        assert lib.rwh_host_inv3(mats.ctypes.data, len(mats), ctypes.c_void_p(gesv), got.ctypes.data) == 0
        with np.errstate(all="ignore"):
            want = np.linalg.inv(mats)
        assert want.dtype == np.float32 and np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_new_entry_points_validate_and_plan(lib):
    """Round-2 entry points: NULL / bad arguments are refused before any device access, and rwh_warp_plan (the dispatch of
    rwh_warp_backward without the launch) names the kernel each configuration gets -- no GPU needed."""
    import torch
    from ransac_with_homography_amd import _lib, kernels
    null = ctypes.c_void_p(0)
    assert lib.rwh_sample_points(null, 8, 8, 3, 0, null, null, 4, 8, 8, 1, null, 2, 0, null) == -1
    assert lib.rwh_project_points_ex(null, null, 4, 2, null, null) == -1
    assert lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_SHAPE, 9) == -1 and lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_SHAPE, 0) == 0
    assert lib.rwh_lab_tune(_lib.RWH_TUNE_SCORE_HPW, 65) == -1 and lib.rwh_lab_tune(7, 0) == -1
    Hs = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
    inv = np.linalg.inv(Hs)
    g = kernels.Grid(5, 3775, 3771, 7, 2034, 2028)
    plan = lambda shape, dt, *a, **k: kernels.warp_plan(shape, dt, inv, g, (2160, 3840), *a, **k)
    # a batch with one homography: the lab knob RWH_TUNE_WARP_FRAMES selects the multi-frame kernel (round 4; off by default)
    assert plan((32, 2160, 3840, 3), torch.uint8, "bilinear", torch.uint8) == "rwh::warp_rgb8_fast8<unsigned char, 6>"
    assert lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_FRAMES, 65) == -1 and lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_FRAMES, 4) == 0
    assert plan((32, 2160, 3840, 3), torch.uint8, "bilinear", torch.uint8) == "rwh::warp_rgb8_fast8m<6>"
    assert lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_FRAMES, 0) == 0
    assert plan((2160, 3840, 3), torch.uint8, "bilinear", torch.uint8) == "rwh::warp_rgb8_fast8<unsigned char, 6>"
    assert plan((2160, 3840, 3), torch.uint8, "bilinear", torch.float32) == "rwh::warp_rgb8_fast8<float, 6>"
    assert plan((2160, 3840, 3), torch.uint8, "nn", torch.uint8) == "rwh::warp_rgb8_nn<6>"
    assert plan((2160, 3840, 3), torch.uint8, "bilinear", torch.float64, exact=True) == "rwh::warp_exact<unsigned char, 3, double, 1>"
    assert plan((2160, 3840, 4), torch.float32, "bilinear", torch.float32) == "rwh::warp_generic<float, 4, float, 1>"
    assert kernels.warp_plan((4, 2160, 3840, 3), torch.uint8, np.stack([inv] * 4), g, (2160, 3840), "bilinear",
                             torch.uint8) == "rwh::warp_rgb8_fast8_tab<unsigned char, 6>"
    # the shape is a function of the homography and the whole grid, never of the row shard
    assert plan((2160, 3840, 3), torch.uint8, "bilinear", torch.uint8, rows=(500, 700)) == "rwh::warp_rgb8_fast8<unsigned char, 6>"
    # a 30-degree rotation leaves the 64 x 8 window: 32 x 16 patches; an output narrower than 128 px: the 4 px kernel
    t = np.deg2rad(30)
    R = np.array([[np.cos(t), -np.sin(t), 100.0], [np.sin(t), np.cos(t), -50.0], [0, 0, 1.0]])
    assert kernels.warp_plan((2160, 3840, 3), torch.uint8, np.linalg.inv(R), g, (2160, 3840), "bilinear",
                             torch.uint8) == "rwh::warp_rgb8_fast8<unsigned char, 5>"
    assert kernels.warp_plan((100, 100, 3), torch.uint8, inv, kernels.Grid(0, 99, 100, 0, 99, 100), (100, 100), "bilinear",
                             torch.uint8) == "rwh::warp_rgb8_fast<unsigned char>"


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import homography as hg
    import ransac as rs
    from ransac_with_homography_amd import RwhUnavailable
    with pytest.raises(RwhUnavailable):
        hg.wrapPerspective(np.zeros((8, 8, 3), np.uint8), np.eye(3), "bilinear")
    with pytest.raises(RwhUnavailable):
        rs.RANSAC(rs.HomoModel(), k=10).run([np.zeros((2, 8), np.float32), np.zeros((2, 8), np.float32)], "fwd")


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing shipped may reference it."""
    shipped = [os.path.join(ROOT, "homography.py"), os.path.join(ROOT, "ransac.py")]
    pkg = os.path.join(ROOT, "ransac_with_homography_amd")
    for d, _, files in os.walk(pkg):
        shipped += [os.path.join(d, f) for f in files if f.endswith((".py", ".hip", ".h"))]
    for path in shipped:
        text = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), path


def test_host_solvers_match_goldens():
    """Host-side O(1) solvers of the product against the reference's G1 outputs."""
    import homography as hg
    from conftest import load_golden
    z = load_golden("g1_fourpoint")
    u, v = z["u"].T[:, :2], z["v"].T[:, :2]
    A, b = hg.calc_correspLinear(u, v)
    assert np.array_equal(A, z["A"]) and np.array_equal(b, z["b"])
    assert np.array_equal(hg.calc_corresp(u, v), z["mat"])
    assert np.array_equal(hg.calcHomographyLinear(u, v), z["H_linear"])
    assert np.array_equal(hg.calcHomography(u, v), z["H_dlt"])          # float64 inputs: host SVD path
    assert hg.calcH is hg.calcHomographyLinear and hg.perspectiveTransform is hg.wrapPerspective


def test_plain_c_consumer(lib, tmp_path):
    """include/rwh.h compiles as C and a gcc-built program links the library and gets the documented status codes."""
    import subprocess
    from ransac_with_homography_amd import _lib
    exe = str(tmp_path / "cabi_smoke")
    libdir = os.path.dirname(_lib.LIB_PATH)
    torch_lib = os.path.join(os.path.dirname(__import__("torch").__file__), "lib")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cabi", "cabi_smoke.c"), "-o", exe, "-L", libdir, "-lrwh_hip",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath," + torch_lib,
                    "-Wl,--allow-shlib-undefined"], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "cabi ok" in out
