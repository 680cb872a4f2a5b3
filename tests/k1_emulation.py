"""float64 numpy emulation of K1 (`dlt4_kernel`, ransac_with_homography_amd/csrc/rwh_ransac.hip), flags included -- TEST
INFRASTRUCTURE for the CPU suite: it lets `_settle_on_host` be exercised on ill-conditioned problems without a GPU.  Same
elimination order, pivoting and thresholds as the kernel; reciprocal-multiplies are plain divisions here (the float32
rounded H agrees with the kernel's on ~all samples, which is all these tests need: the flags and the order of magnitude of
the count differences)."""
import numpy as np

RWH_HYP_REPEATED, RWH_HYP_SINGULAR, RWH_HYP_ILLCOND, RWH_HYP_DEGENERATE = 1, 2, 4, 8


def dlt4(pa, pb, idx, near_singular=False):
    """pa, pb: float32 [M, 2]; idx: int [K, 4] -> (H float32 [K, 9], flags uint8 [K])."""
    idx = np.asarray(idx)[:, :4]
    K = idx.shape[0]
    A, B = pa[idx], pb[idx]
    x, y, xp, yp = A[..., 0], A[..., 1], B[..., 0], B[..., 1]
    M = np.zeros((K, 4, 9))
    M[..., 0] = -x.astype(np.float64); M[..., 1] = -y.astype(np.float64); M[..., 2] = -1.0
    M[..., 3] = (x * xp).astype(np.float64); M[..., 4] = (y * xp).astype(np.float64); M[..., 5] = -xp.astype(np.float64)
    M[..., 6] = (x * yp).astype(np.float64); M[..., 7] = (y * yp).astype(np.float64); M[..., 8] = -yp.astype(np.float64)
    colscale = np.abs(M[:, :, :3]).max(axis=1)
    qscale = np.abs(M[:, :, [3, 4, 6, 7]]).max(axis=(1, 2))
    ratios = np.zeros((K, 5))
    ar = np.arange(K)
    with np.errstate(all="ignore"):
        for c in range(3):
            p = c + np.argmax(np.abs(M[:, c:, c]), axis=1)
            tmp = M[ar, c].copy(); M[ar, c] = M[ar, p]; M[ar, p] = tmp
            piv = M[:, c, c]
            ratios[:, c] = np.abs(piv) / colscale[:, c]
            for i in range(c + 1, 4):
                f = M[:, i, c] / piv
                M[:, i, c + 1:] = M[:, i, c + 1:] - f[:, None] * M[:, c, c + 1:]
        a11, a12, b1 = M[:, 3, 3].copy(), M[:, 3, 4].copy(), M[:, 3, 5].copy()
        a21, a22, b2 = M[:, 3, 6].copy(), M[:, 3, 7].copy(), M[:, 3, 8].copy()
        sw = np.abs(a21) > np.abs(a11)
        a11, a21 = np.where(sw, a21, a11), np.where(sw, a11, a21)
        a12, a22 = np.where(sw, a22, a12), np.where(sw, a12, a22)
        b1, b2 = np.where(sw, b2, b1), np.where(sw, b1, b2)
        f2 = a21 / a11
        d2 = a22 - f2 * a12
        ratios[:, 3] = np.abs(a11) / qscale
        ratios[:, 4] = np.abs(d2) / (np.abs(a22) + np.abs(f2 * a12))
        h8 = (b2 - f2 * b1) / d2
        h7 = (b1 - a12 * h8) / a11
        h = np.zeros((K, 9))
        for blk in range(2):
            q = 3 + 3 * blk
            r2 = (M[:, 2, q + 2] - M[:, 2, q] * h7 - M[:, 2, q + 1] * h8) / M[:, 2, 2]
            r1 = (M[:, 1, q + 2] - M[:, 1, q] * h7 - M[:, 1, q + 1] * h8 - M[:, 1, 2] * r2) / M[:, 1, 1]
            r0 = (M[:, 0, q + 2] - M[:, 0, q] * h7 - M[:, 0, q + 1] * h8 - M[:, 0, 1] * r1 - M[:, 0, 2] * r2) / M[:, 0, 0]
            h[:, 3 * blk] = r0; h[:, 3 * blk + 1] = r1; h[:, 3 * blk + 2] = r2
        h[:, 6] = h7; h[:, 7] = h8; h[:, 8] = 1.0
        ss = (h * h).sum(1)
        n = (h / np.sqrt(ss)[:, None]).astype(np.float32)
        H = n / n[:, 8:9]
        degenerate = ~(ss <= 1e14) | ~(ratios >= 1e-7).all(axis=1)
        illcond = ~(ss <= 1e14) | ~(ratios >= 1e-3).all(axis=1)
        finite = np.isfinite(H).all(axis=1)
        # a nearly singular H: |det| against the sum of the six products' magnitudes (rwh_ransac.hip, dlt4_kernel)
        q = H.astype(np.float64)
        t = [q[:, 0] * q[:, 4] * q[:, 8], q[:, 1] * q[:, 5] * q[:, 6], q[:, 2] * q[:, 3] * q[:, 7],
             q[:, 2] * q[:, 4] * q[:, 6], q[:, 1] * q[:, 3] * q[:, 8], q[:, 0] * q[:, 5] * q[:, 7]]
        det = (t[0] + t[1] + t[2]) - (t[3] + t[4] + t[5])
        if near_singular:
            illcond |= ~(np.abs(det) > 1e-6 * sum(np.abs(v) for v in t))
            degenerate |= ~(np.abs(det) > 1e-6 * sum(np.abs(v) for v in t))
    a, b, c, d = (idx[:, i] for i in range(4))
    rep = (a == b) | (a == c) | (a == d) | (b == c) | (b == d) | (c == d)
    flags = (rep * RWH_HYP_REPEATED + (~finite) * RWH_HYP_SINGULAR + illcond * RWH_HYP_ILLCOND + degenerate * RWH_HYP_DEGENERATE).astype(np.uint8)
    return np.ascontiguousarray(H, dtype=np.float32), flags
