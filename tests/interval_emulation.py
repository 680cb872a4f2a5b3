"""numpy emulation of K2i (`score_interval_kernel`, ransac_with_homography_amd/csrc/rwh_ransac.hip) -- TEST INFRASTRUCTURE for the CPU
suite: the count of a hypothesis under 'fwd' as an interval [lo, hi] over every H within a perturbation budget of delta x each
entry's natural scale.  Same formulas as the kernel (float64 margins around the reference's float32 projection)."""
import numpy as np

RWH_HYP_ILLCOND = 4


def score_interval(H, rows, flags, pa, pb, th, coord_scale, delta0, delta1):
    """H float32 [K, 9], rows int [n], flags uint8 [K] or None, pa / pb float32 [M, 2] -> (lo, hi) int64 [n]."""
    rows = np.asarray(rows, dtype=np.int64)
    lo, hi = np.zeros(len(rows), np.int64), np.zeros(len(rows), np.int64)
    x, y = pa[:, 0].astype(np.float32), pa[:, 1].astype(np.float32)
    tx, ty = pb[:, 0].astype(np.float32), pb[:, 1].astype(np.float32)
    ax, ay = np.abs(x).astype(np.float64), np.abs(y).astype(np.float64)
    C = float(coord_scale)
    for o, r in enumerate(rows):
        h = H[r].astype(np.float32)
        delta = delta1 if (flags is not None and (int(flags[r]) & RWH_HYP_ILLCOND)) else delta0
        a = np.abs(h.astype(np.float64))
        s = max(a[0], a[1], a[3], a[4], a[8], max(a[2], a[5]) / C, max(a[6], a[7]) * C)
        N = np.array([s, s, s * C, s, s, s * C, s / C, s / C, s])
        D = delta * np.maximum(a, N)
        with np.errstate(all="ignore"):
            # the reference's (OpenBLAS sgemm) order: rounded multiply, ONE fused multiply-add, add; the float64 product
            # of two float32 values is exact, so rounding its sum once reproduces fmaf up to rare double roundings
            fmaf = lambda a, b, c: (np.float64(a) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
            a0 = fmaf(h[1], y, h[0] * x) + h[2]
            a1 = fmaf(h[4], y, h[3] * x) + h[5]
            a2 = fmaf(h[7], y, h[6] * x) + h[8]
            den = a2 + np.float32(1e-10)
            px, py = a0 / den, a1 / den
            dx, dy = px - tx, py - ty
            err = np.sqrt(dx * dx + dy * dy)
            d0, d1, d2 = D[0] * ax + D[1] * ay + D[2], D[3] * ax + D[4] * ay + D[5], D[6] * ax + D[7] * ay + D[8]
            room = np.abs(den.astype(np.float64)) - d2
            mg = ((d0 + np.abs(px.astype(np.float64)) * d2) + (d1 + np.abs(py.astype(np.float64)) * d2)) / room
            mg = mg + 2.0 ** -21 * (np.abs(px) + np.abs(py) + np.abs(tx) + np.abs(ty)).astype(np.float64)
            ok = (room > 0) & np.isfinite(mg) & np.isfinite(err)
            e = err.astype(np.float64)
            lo[o] = int((ok & (e + mg < th)).sum())
            hi[o] = len(x) - int((ok & (e - mg >= th)).sum())
    return lo, hi
