"""Pure-numpy Philox4x32-10 and the 4-distinct-indices mapping of `sample4_kernel`
(ransac_with_homography_amd/csrc/rwh_ransac.hip): test infrastructure for the device sampler of
rwh_ransac_batched.  The generator is checked against the Random123 known-answer vectors in
tests/test_batched_cpu.py."""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    """ctr: [...,4] uint32-valued, key: (k0, k1) -> [...,4] uint64 array holding 32-bit outputs."""
    c = np.asarray(ctr, dtype=np.uint64).copy()
    k0, k1 = int(key[0]) & MASK, int(key[1]) & MASK
    for _ in range(10):
        p0 = np.uint64(M0) * c[..., 0]
        p1 = np.uint64(M1) * c[..., 2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & np.uint64(MASK)
        hi1, lo1 = p1 >> np.uint64(32), p1 & np.uint64(MASK)
        n0 = hi1 ^ c[..., 1] ^ np.uint64(k0)
        n2 = hi0 ^ c[..., 3] ^ np.uint64(k1)
        c = np.stack([n0, lo1, n2, lo0], axis=-1)
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c


def sample4(seed, problem, k, m):
    """The [k,4] int32 index table the device draws for `problem` (size m) under `seed`."""
    ctr = np.zeros((k, 4), dtype=np.uint64)
    ctr[:, 0] = np.arange(k, dtype=np.uint64)
    ctr[:, 1] = problem
    r = philox4x32_10(ctr, (seed & MASK, (seed >> 32) & MASK))
    if m < 4:
        return np.zeros((k, 4), dtype=np.int32)
    mulhi = lambda a, n: ((a * np.uint64(n)) >> np.uint64(32)).astype(np.int64)
    i0 = mulhi(r[:, 0], m)
    j = mulhi(r[:, 1], m - 1)
    i1 = j + (j >= i0)
    a, b = np.minimum(i0, i1), np.maximum(i0, i1)
    j = mulhi(r[:, 2], m - 2)
    j = j + (j >= a)
    j = j + (j >= b)
    i2 = j
    lo, hi = np.minimum(a, i2), np.maximum(b, i2)
    mid = a + b + i2 - lo - hi
    j = mulhi(r[:, 3], m - 3)
    j = j + (j >= lo)
    j = j + (j >= mid)
    j = j + (j >= hi)
    return np.stack([i0, i1, i2, j], axis=1).astype(np.int32)
