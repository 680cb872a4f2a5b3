// K1 (batched 4-point DLT) and K2 (hypothesis x correspondence inlier scorer) for gfx950.
//
// K1 replaces K calls of HomoModel.fit -> calcHomography -> calc_corresp + numpy SVD
// (ransac.py:178-180, 52; homography.py:71-88, 4-14); K2 replaces computeLoss + `err < th` +
// np.sum and the accept rules (ransac.py:182-202).  Contract in include/rwh.h.
//
// Numerical recipes (re-verified against tests/golden):
//  K1  the 8x9 matrix entries are float32 (products x*x' etc. rounded to float32 like the
//      reference's float32 inputs give); its null vector is computed in float64, scaled to unit
//      2-norm, rounded to float32 and divided by its 9th element in float32 -- the same
//      roundings numpy applies to LAPACK's float64 singular vector.  The float64 solve uses the
//      block structure of the DLT matrix (both row families share the 4x3 block [-x -y -1]):
//      one pivoted elimination of that block applied to both right-hand blocks, a 2x2 system
//      for (h31, h32), two back substitutions.  ~100 float64 operations instead of an 8x8 LU.
//  K2  per output row j of val @ [x;y;1] (OpenBLAS sgemm k-order):  acc = h[j,0]*x (rounded),
//      acc = fmaf(h[j,1], y, acc), acc = acc + h[j,2];  den = acc2 + 1e-10f;  IEEE divides;
//      s = dx*dx + dy*dy with separately rounded products;  err = sqrtf(s);  inlier = err < th.
//      'backward' does the same through inv(val) (float64 inverse rounded to float32).
// No FMA contraction anywhere (rwh_common.h), no fast-math, IEEE divide / sqrt.
#include "rwh_common.h"

namespace rwh {

// ------------------------------------------------------------------------------------------------
// K1: one lane = one hypothesis.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void swap_if(bool c, double& a, double& b) {
    const double t = a;
    a = c ? b : a;
    b = c ? t : b;
}

// Batched mode (rwh_ransac_batched): `offsets` != NULL; hypothesis t belongs to problem t / k_per, whose correspondences
// are rows offsets[p] .. offsets[p+1]-1 of pa / pb, and idx holds indices local to the problem.
// 1/v to float64 round-off (v_rcp_f64 + two Newton steps: ~1.1e-16 relative), ~10 instructions where an IEEE divide
// takes ~35.  K1's float64 solve is an elimination of our own choosing, not LAPACK's sequence: any float64-accurate
// result rounds to the same float32 H (except on float32 rounding boundaries, where LAPACK's own round-off decides too).
__device__ __forceinline__ double recip(double v) {
    double r = __builtin_amdgcn_rcp(v);
    r = fma(fma(-v, r, 1.0), r, r);
    r = fma(fma(-v, r, 1.0), r, r);
    return r;
}

__global__ __launch_bounds__(64) void dlt4_kernel(const float* __restrict__ pa, const float* __restrict__ pb, int m,
                                                  const int32_t* __restrict__ idx, int k,
                                                  float* __restrict__ hout, uint8_t* __restrict__ flags,
                                                  const int32_t* __restrict__ offsets, int k_per,
                                                  unsigned long long* __restrict__ reset_keys, int n_reset, int flag_near_singular) {
    __shared__ float hstage[64 * 9];
    const int t = blockIdx.x * 64 + threadIdx.x;
    // a search's packed argmax keys are cleared here instead of by a memset launch of their own (the grid always
    // covers n_reset threads, see launch_dlt4)
    if (reset_keys && t < n_reset) reset_keys[t] = 0ull;
    if (blockIdx.x * 64 >= k) return;             // whole wave past the end
    const bool live = t < k;
    if (!live) idx = nullptr;                     // lanes past the end compute on point 0 of problem 0 and store nothing
    if (offsets) {
        const int p = live ? t / k_per : 0;       // (a dead lane's own t / k_per can lie past the offsets table)
        const int base = offsets[p];
        m = offsets[p + 1] - base;
        pa += 2 * (size_t)base; pb += 2 * (size_t)base;
    }
    int id[4];
    bool bad_index = false;
    const int4 raw = idx ? reinterpret_cast<const int4*>(idx)[t] : int4{0, 0, 0, 0};   // one 16-byte load per lane
    id[0] = raw.x; id[1] = raw.y; id[2] = raw.z; id[3] = raw.w;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (id[i] < 0 || id[i] >= m) { bad_index = true; id[i] = 0; }
    const bool repeated = (id[0] == id[1]) | (id[0] == id[2]) | (id[0] == id[3]) | (id[1] == id[2]) |
                          (id[1] == id[3]) | (id[2] == id[3]);

    // rows i = 0..3 of the shared elimination:  [ -x -y -1 | x*x' y*x' -x' | x*y' y*y' -y' ]
    // i.e.  P h(1..3) + Q (h7,h8) = rhs  for the even DLT rows (cols 3..5) and the odd ones (6..8)
    double M[4][9];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float2 A = reinterpret_cast<const float2*>(pa)[id[i]], Bp = reinterpret_cast<const float2*>(pb)[id[i]];
        const float x = A.x, y = A.y, xp = Bp.x, yp = Bp.y;
        M[i][0] = -(double)x;  M[i][1] = -(double)y;  M[i][2] = -1.0;
        M[i][3] = (double)(x * xp);  M[i][4] = (double)(y * xp);  M[i][5] = -(double)xp;  // float32 products
        M[i][6] = (double)(x * yp);  M[i][7] = (double)(y * yp);  M[i][8] = -(double)yp;
    }

    // Conditioning flag (RWH_HYP_ILLCOND): the smallest pivot of the elimination against the scale of its column.  Samples with
    // three collinear source points, or equal coordinates at different indices, are finite here but arbitrary -- K1's
    // elimination and LAPACK's SVD then return DIFFERENT float32 H (inlier counts apart by hundreds on lattice-like data) --
    // and so is any sample whose unit null vector has a 9th element below 1e-7 (h / h[8] amplifies round-off).  The host
    // settles flagged samples with the reference's own solver (ransac._settle_on_host).  Thresholds: tools/README (k1 calibration).
    double cs0 = 0.0, cs1 = 0.0, qs = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        cs0 = fmax(cs0, fabs(M[i][0])); cs1 = fmax(cs1, fabs(M[i][1]));
        qs = fmax(fmax(qs, fmax(fabs(M[i][3]), fabs(M[i][4]))), fmax(fabs(M[i][6]), fabs(M[i][7])));
    }
    double piv_ratio[5];
    // eliminate the shared 4x3 block with partial pivoting (first maximum wins)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int p = c;
        double best = fabs(M[c][c]);
#pragma unroll
        for (int i = c + 1; i < 4; ++i) {
            const double v = fabs(M[i][c]);
            if (v > best) { best = v; p = i; }
        }
#pragma unroll
        for (int i = c + 1; i < 4; ++i) {
            const bool sw = (p == i);
#pragma unroll
            for (int j = c; j < 9; ++j) swap_if(sw, M[c][j], M[i][j]);
        }
        piv_ratio[c] = fabs(M[c][c]) / (c == 0 ? cs0 : c == 1 ? cs1 : 1.0);
        const double rpiv = recip(M[c][c]);
#pragma unroll
        for (int i = c + 1; i < 4; ++i) {
            const double f = M[i][c] * rpiv;
#pragma unroll
            for (int j = c + 1; j < 9; ++j) M[i][j] = M[i][j] - f * M[c][j];
        }
    }
    // row 3 is now  e1*h7 + e2*h8 = e3  and  o1*h7 + o2*h8 = o3
    double a11 = M[3][3], a12 = M[3][4], b1 = M[3][5];
    double a21 = M[3][6], a22 = M[3][7], b2 = M[3][8];
    const bool sw2 = fabs(a21) > fabs(a11);
    swap_if(sw2, a11, a21); swap_if(sw2, a12, a22); swap_if(sw2, b1, b2);
    const double ra11 = recip(a11);
    const double f2 = a21 * ra11;
    const double d2 = a22 - f2 * a12;
    piv_ratio[3] = fabs(a11) / qs;
    piv_ratio[4] = fabs(d2) / (fabs(a22) + fabs(f2 * a12));
    const double h8 = (b2 - f2 * b1) * recip(d2);
    const double h7 = (b1 - a12 * h8) * ra11;

    double h[9];
    const double rm22 = recip(M[2][2]), rm11 = recip(M[1][1]), rm00 = recip(M[0][0]);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {  // blk 0: h1..h3 from cols 3..5, blk 1: h4..h6 from cols 6..8
        const int q = 3 + 3 * blk;
        const double t2 = M[2][q + 2] - M[2][q] * h7 - M[2][q + 1] * h8;
        const double r2 = t2 * rm22;
        const double t1 = M[1][q + 2] - M[1][q] * h7 - M[1][q + 1] * h8 - M[1][2] * r2;
        const double r1 = t1 * rm11;
        const double t0 = M[0][q + 2] - M[0][q] * h7 - M[0][q + 1] * h8 - M[0][1] * r1 - M[0][2] * r2;
        const double r0 = t0 * rm00;
        h[3 * blk] = r0; h[3 * blk + 1] = r1; h[3 * blk + 2] = r2;
    }
    h[6] = h7; h[7] = h8; h[8] = 1.0;

    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) ss = ss + h[i] * h[i];
    // !(x >= t) also catches NaN (0 / 0 scales, inf - inf)
    bool illcond = !(ss <= 1e14);                             // |n[8]| = 1 / sqrt(ss) < 1e-7
    // RWH_HYP_DEGENERATE (round 4): the part of the ill-conditioned samples whose H says NOTHING about the reference's -- a pivot
    // below 1e-7 of its column's scale (LAPACK's own null vector is round-off there), |n[8]| < 1e-7, and (below) a nearly
    // singular H in searches that invert.  The rest of RWH_HYP_ILLCOND (pivot ratios 1e-7 .. 1e-3) is accurate here -- this
    // elimination is ~1000 x closer to the exact null vector than LAPACK's SVD, profiles/r04_lab_notes.txt -- and under 'fwd' goes
    // through rwh_score_interval with a wider perturbation budget instead of straight to the host.
    bool degenerate = illcond;
#pragma unroll
    for (int i = 0; i < 5; ++i) { illcond |= !(piv_ratio[i] >= 1e-3); degenerate |= !(piv_ratio[i] >= 1e-7); }
    const double rnrm = recip(sqrt(ss));
    float n[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) n[i] = (float)(h[i] * rnrm);
    bool finite = true;
    double q[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float v = n[i] / n[8];
        finite &= (fabsf(v) <= 3.4028234664e38f);  // false for NaN / inf
        hstage[9 * threadIdx.x + i] = v;
        q[i] = (double)v;
    }
    // ... and, for searches that will INVERT the hypotheses ('backward' / 'reproj': numpy.linalg.inv of every one, ransac.py:74),
    // a nearly singular H -- a sample drawn from two or three tight clusters, or simply a bad one: where the determinant is what
    // is left of six products that cancel below 1e-6 of their size, LAPACK's float64 elimination and any other round apart
    // after the cast to float32 (measured against numpy on 17 555 cluster-problem hypotheses: 286 of the 287 differing inverses
    // have a ratio below 1e-7, the last one 4e-7; none above 1e-5 in 27 000).  The settle step gives the flagged ones numpy's
    // own inverse (rwh_score_count_inv).  5-12 % of the samples of a real, contaminated match set trip it.
    if (flag_near_singular) {
        const double t0 = q[0] * q[4] * q[8], t1 = q[1] * q[5] * q[6], t2 = q[2] * q[3] * q[7];
        const double t3 = q[2] * q[4] * q[6], t4 = q[1] * q[3] * q[8], t5 = q[0] * q[5] * q[7];
        const double det = (t0 + t1 + t2) - (t3 + t4 + t5);
        const double perm = fabs(t0) + fabs(t1) + fabs(t2) + fabs(t3) + fabs(t4) + fabs(t5);
        illcond |= !(fabs(det) > 1e-6 * perm);
        degenerate |= !(fabs(det) > 1e-6 * perm);
    }
    if (live) flags[t] = (uint8_t)((repeated || bad_index ? RWH_HYP_REPEATED : 0u) | (finite ? 0u : RWH_HYP_SINGULAR) |
                                   (illcond ? RWH_HYP_ILLCOND : 0u) | (degenerate ? RWH_HYP_DEGENERATE : 0u));
    // the wave's 64 x 9 floats leave as 9 coalesced 256-byte stores (lane-strided 36-byte records would be 9 scattered ones)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nfl = 9 * min(64, k - (int)blockIdx.x * 64);
    float* wout = hout + 9 * (size_t)blockIdx.x * 64;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int f = i * 64 + threadIdx.x;
        if (f < nfl) wout[f] = hstage[f];
    }
}

// ------------------------------------------------------------------------------------------------
// K0 (batched mode only): device sampling.  Philox4x32-10 (Salmon et al., SC'11), counter = (hypothesis index inside
// the problem, problem index, 0, 0), key = the 64-bit seed: the samples of a problem depend on (seed, problem, hypothesis)
// only.  The four 32-bit outputs become four DISTINCT indices in [0, m): j_i = mulhi(r_i, m - i) picks among the indices
// not taken yet (multiply-shift range reduction: bias < m / 2^32).  NOT the reference's numpy.random stream
// (ransac.py:177 samples with replacement from the legacy generator): a documented non-parity mode.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__global__ __launch_bounds__(256) void sample4_kernel(const int32_t* __restrict__ offsets, int n_problems, int k_per,
                                                      unsigned long long seed, unsigned problem_base,
                                                      int32_t* __restrict__ idx) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_problems * k_per) return;
    const int p = t / k_per, hyp = t - p * k_per;
    const int m = offsets[p + 1] - offsets[p];
    uint32_t c[4] = {(uint32_t)hyp, problem_base + (uint32_t)p, 0u, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    int id[4] = {0, 0, 0, 0};
    if (m >= 4) {
        id[0] = (int)__umulhi(c[0], (uint32_t)m);
        int j = (int)__umulhi(c[1], (uint32_t)(m - 1));
        id[1] = j + (j >= id[0]);
        int a = min(id[0], id[1]), b = max(id[0], id[1]);          // taken so far, ascending
        j = (int)__umulhi(c[2], (uint32_t)(m - 2));
        j += (j >= a); j += (j >= b);
        id[2] = j;
        int lo = min(a, j), hi = max(b, j), mid = a + b + j - lo - hi;
        j = (int)__umulhi(c[3], (uint32_t)(m - 3));
        j += (j >= lo); j += (j >= mid); j += (j >= hi);
        id[3] = j;
    }                                                               // m < 4: (0,0,0,0), flagged "repeated" by K1
    reinterpret_cast<int4*>(idx)[t] = int4{id[0], id[1], id[2], id[3]};
}

// ------------------------------------------------------------------------------------------------
// K2: one wavefront = one hypothesis, lanes stride over correspondences, ballot + popcount.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void proj(const float (&h)[9], float x, float y, float& px, float& py, float& pw) {
    float a0 = h[0] * x; a0 = fmaf(h[1], y, a0); a0 = a0 + h[2];
    float a1 = h[3] * x; a1 = fmaf(h[4], y, a1); a1 = a1 + h[5];
    float a2 = h[6] * x; a2 = fmaf(h[7], y, a2); a2 = a2 + h[8];
    const float den = a2 + 1e-10f;
    px = a0 / den; py = a1 / den; pw = a2 / den;  // IEEE: -fhip-fp32-correctly-rounded-divide-sqrt
}

__device__ __forceinline__ float proj_sq(const float (&h)[9], float x, float y, float xp, float yp) {
    float px, py, pw;
    proj(h, x, y, px, py, pw);
    const float dx = px - xp, dy = py - yp;
    return dx * dx + dy * dy;
}

__device__ __forceinline__ float proj_err(const float (&h)[9], float x, float y, float xp, float yp) {
    return sqrtf(proj_sq(h, x, y, xp, yp));  // correctly rounded (__fsqrt_rn would lower to the approximate native sqrt)
}

// ---- K2's filter: decide most pairs without the two IEEE divisions ---------------------------------------------------
// The reference's decision per pair is dist < th with dist computed in float32 through two correctly rounded divisions
// (22 of the ~35 instructions of a pair).  proj_fast replaces them with q~ = a * rcp(den): the numerators and the
// denominator are the SAME float32 values as in proj(); v_rcp_f32 is accurate to 1 ulp, so |q~ - q^| <= 2^-22 |q^|
// against the correctly rounded quotient q^, as long as den is a normal number whose reciprocal is normal too
// (`den_ok`; NaNs fail it).  A pair whose two answers could differ has q^ or q~ within th of the target, hence
// |q| <= |target| + th, and its float32 distance moves by less than 2^-21 (|tx| + |ty| + 2 th) + 2^-21 th.  The band
// below is four times that: a pair is decided here only when its approximate distance clears th by the band on either
// side; everything else (and every NaN) makes the whole 64-pair word take the exact path.  Counts and masks therefore
// stay bit-identical to the reference's, and on real data fewer than 1 % of the words take the exact path.
struct Band { float lo, hi; };      // fwd / backward: limits on the SQUARED distance; reproj: on the sum of distances
__device__ __forceinline__ float band_margin(float tx, float ty, float thf) {
    return (fabsf(tx) + fabsf(ty) + 2.f * thf + 2.f) * 0x1p-19f;
}
__device__ __forceinline__ Band band_sq(float tx, float ty, float thf) {
    const float mgn = band_margin(tx, ty, thf);
    const float lo = fmaxf(thf - mgn, 0.f), hi = thf + mgn;
    return Band{lo * lo * (1.f - 0x1p-20f), hi * hi * (1.f + 0x1p-20f)};
}
__device__ __forceinline__ float proj_sq_fast(const float (&h)[9], float x, float y, float xp, float yp, float& den) {
    float a0 = h[0] * x; a0 = fmaf(h[1], y, a0); a0 = a0 + h[2];
    float a1 = h[3] * x; a1 = fmaf(h[4], y, a1); a1 = a1 + h[5];
    float a2 = h[6] * x; a2 = fmaf(h[7], y, a2); a2 = a2 + h[8];
    den = a2 + 1e-10f;                                  // usable when 2^-100 < |den| < 2^100 (the caller tests it)
    const float r = __builtin_amdgcn_rcpf(den);
    const float dx = a0 * r - xp, dy = a1 * r - yp;
    return dx * dx + dy * dy;
}
__device__ __forceinline__ float proj_sq_fast(const float (&h)[9], float x, float y, float xp, float yp, bool& den_ok) {
    float den;
    const float sq = proj_sq_fast(h, x, y, xp, yp, den);
    den_ok = (int)(fabsf(den) > 0x1p-100f) & (int)(fabsf(den) < 0x1p100f);
    return sq;
}

// float64 inverse of a float32 3x3 rounded back to float32 (numpy.linalg.inv on a float32 array,
// ransac.py:74).  Plain LU with partial pivoting and reciprocal scaling; agrees with LAPACK's
// result after the float32 rounding (checked on the golden hypotheses).
__device__ __forceinline__ void inverse3(const float (&hf)[9], float (&inv)[9]) {
    double A[3][3], B[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { A[i][j] = (double)hf[3 * i + j]; B[i][j] = (i == j) ? 1.0 : 0.0; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int p = c;
        double best = fabs(A[c][c]);
#pragma unroll
        for (int i = c + 1; i < 3; ++i) {
            const double v = fabs(A[i][c]);
            if (v > best) { best = v; p = i; }
        }
#pragma unroll
        for (int i = c + 1; i < 3; ++i) {
            const bool sw = (p == i);
#pragma unroll
            for (int j = 0; j < 3; ++j) { swap_if(sw, A[c][j], A[i][j]); swap_if(sw, B[c][j], B[i][j]); }
        }
        const double r = 1.0 / A[c][c];
#pragma unroll
        for (int i = c + 1; i < 3; ++i) {
            A[i][c] = A[i][c] * r;
#pragma unroll
            for (int j = c + 1; j < 3; ++j) A[i][j] = A[i][j] - A[i][c] * A[c][j];
        }
    }
#pragma unroll
    for (int col = 0; col < 3; ++col) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int q = 0; q < i; ++q) B[i][col] = B[i][col] - A[i][q] * B[q][col];
#pragma unroll
        for (int i = 2; i >= 0; --i) {
#pragma unroll
            for (int q = i + 1; q < 3; ++q) B[i][col] = B[i][col] - A[i][q] * B[q][col];
            B[i][col] = B[i][col] * (1.0 / A[i][i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) inv[3 * i + j] = (float)B[i][j];
}

// One wavefront scores HPW consecutive hypotheses.  WORDS > 0: the correspondences (M <= 64*WORDS) live in
// registers for the whole wave -- lane l holds points l, l+64, ... -- so the loop over hypotheses is pure VALU +
// scalar loads of H; WORDS == 0: any M, points streamed from L1/L2 per hypothesis.  Writes counts (and masks, losses);
// the winner is picked from the counts by argmax_kernel.
template <int LOSS, int WORDS>
__global__ __launch_bounds__(256) void score_kernel(const float* __restrict__ hs, const float* __restrict__ pa,
                                                    const float* __restrict__ pb, int m, int k, int hpw, double th,
                                                    float sq_limit, int32_t* __restrict__ counts, uint64_t* __restrict__ masks,
                                                    float* __restrict__ errs, const int32_t* __restrict__ offsets,
                                                    int k_per, int mask_stride, const int32_t* __restrict__ stop_needs,
                                                    unsigned long long* stop_keys, float thf, int filter,
                                                    const float* __restrict__ hinvs) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    int h_begin, h_end;
    if (offsets) {   // batched mode: a wave stays inside one problem
        const int wpp = (k_per + hpw - 1) / hpw;                     // waves per problem
        const int p = wid / wpp, chunk = wid - p * wpp;
        if ((long long)p * k_per >= k) return;
        const int base = offsets[p];
        m = offsets[p + 1] - base;
        pa += 2 * (size_t)base; pb += 2 * (size_t)base;
        h_begin = p * k_per + chunk * hpw;
        h_end = min((p + 1) * k_per, h_begin + hpw);
    } else {
        h_begin = wid * hpw;
        if (h_begin >= k) return;
        h_end = min(k, h_begin + hpw);
    }
    const int words = (m + 63) >> 6;
    // early stop (batched mode, RWH_BATCH_EARLY_STOP): word 1 of the problem's packed keys doubles as its "done" word --
    // 0xFFFFFFFF - (lowest hypothesis index that reached `need` so far).  The reference stops at that hypothesis
    // (`break`, ransac.py:186-190): later ones are never evaluated, so a wave that finds an earlier exit on record skips
    // its hypothesis (count -1).  Waves are dispatched in index order, so most of the work after the exit disappears.
    const int stop_p = stop_keys ? h_begin / k_per : 0;
    const int stop_need = stop_keys ? stop_needs[stop_p] : 0;
    unsigned long long* stop_word = stop_keys ? stop_keys + 2 * (size_t)stop_p + 1 : nullptr;

    float2 ra[WORDS > 0 ? WORDS : 1], rb[WORDS > 0 ? WORDS : 1];
    Band band[WORDS > 0 ? WORDS : 1];
    const bool fast = filter && !errs;                           // uniform; the loss values themselves need the exact path
    auto make_band = [&](float2 a, float2 b) -> Band {
        if constexpr (LOSS == RWH_LOSS_FWD) return band_sq(b.x, b.y, thf);
        else if constexpr (LOSS == RWH_LOSS_BACKWARD) return band_sq(a.x, a.y, thf);
        else {
            const float mgn = band_margin(a.x, a.y, thf) + band_margin(b.x, b.y, thf);
            return Band{thf - mgn, thf + mgn};
        }
    };
    if constexpr (WORDS > 0) {
#pragma unroll
        for (int w = 0; w < WORDS; ++w) {
            const int j = w * 64 + lane;
            ra[w] = j < m ? reinterpret_cast<const float2*>(pa)[j] : float2{0.f, 0.f};
            rb[w] = j < m ? reinterpret_cast<const float2*>(pb)[j] : float2{0.f, 0.f};
            band[w] = make_band(ra[w], rb[w]);
        }
    }
    // The matrices of HB consecutive hypotheses are contiguous in `hs`: ONE coalesced vector load fetches them (lane l =
    // float l of the block), the next block's load is in flight while this block is scored, and a hypothesis takes its
    // nine floats with v_readlane (uniform lane index) -- straight into the aligned SGPR pairs the packed multiplies want.
    constexpr int HB = 7;                                          // 63 floats per block
    auto fetch = [&](int first) -> float {
        const int n = 9 * min(HB, h_end - first);
        return (first < h_end && lane < n) ? hs[9 * (size_t)first + lane] : 0.f;
    };
    float hv = fetch(h_begin);
    for (int hyp0 = h_begin; hyp0 < h_end; hyp0 += HB) {
    const float hv_next = fetch(hyp0 + HB);
    const int nb = min(HB, h_end - hyp0);
#pragma unroll 1
    for (int q = 0; q < nb; ++q) {
        const int hyp = hyp0 + q;
        if (stop_word) {
            const unsigned long long done = __hip_atomic_load(stop_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int local = hyp - stop_p * k_per;
            if (done != 0ull && (long long)(0xFFFFFFFFull - done) < (long long)local) {      // an earlier hypothesis already exits
                if (lane == 0) counts[hyp] = -1;
                if (masks)
                    for (int w = lane; w < mask_stride; w += 64) masks[(size_t)hyp * mask_stride + w] = 0;   // every word (M > 4096: more than 64)
                continue;
            }
        }
        float h[9], hi[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) h[i] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(hv), 9 * q + i));
        if constexpr (LOSS != RWH_LOSS_FWD) {
            if (hinvs) {                       // the caller's inverses (numpy.linalg.inv's own, for the settle step): uniform loads
#pragma unroll
                for (int i = 0; i < 9; ++i) hi[i] = hinvs[9 * (size_t)hyp + i];
            } else {
                inverse3(h, hi);
            }
        }
        int count = 0;
        auto score = [&](int w, float2 a, float2 b, const Band& bd) {
            const int j = w * 64 + lane;
            if (fast) {
                const int rem = m - w * 64;                                   // pairs in this word (uniform)
                const unsigned long long valid = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
                bool in, sure;
                if constexpr (LOSS == RWH_LOSS_REPROJ) {
                    bool ok1, ok2;
                    const float e = __builtin_amdgcn_sqrtf(proj_sq_fast(h, a.x, a.y, b.x, b.y, ok1)) +
                                    __builtin_amdgcn_sqrtf(proj_sq_fast(hi, b.x, b.y, a.x, a.y, ok2));
                    in = e < bd.lo;
                    sure = (in | (e > bd.hi)) & ok1 & ok2;
                } else {
                    bool ok;
                    const float sq = LOSS == RWH_LOSS_FWD ? proj_sq_fast(h, a.x, a.y, b.x, b.y, ok) : proj_sq_fast(hi, b.x, b.y, a.x, a.y, ok);
                    in = sq < bd.lo;
                    sure = (in | (sq > bd.hi)) & ok;
                }
                if ((~__ballot(sure) & valid) == 0ull) {                      // every pair of the word is decided
                    const unsigned long long bal = __ballot(in) & valid;
                    count += __popcll(bal);
                    if (masks && lane == 0) masks[(size_t)hyp * mask_stride + w] = bal;
                    return;
                }
            }
            bool inl = false;
            if (j < m) {
                if constexpr (LOSS == RWH_LOSS_REPROJ) {
                    float e = proj_err(h, a.x, a.y, b.x, b.y);
                    e = e + proj_err(hi, b.x, b.y, a.x, a.y);
                    inl = (double)e < th;
                    if (errs) errs[(size_t)hyp * m + j] = e;
                } else {
                    // one square root per pair: (double)sqrtf(s) < th  <=>  s < sq_limit, the smallest float whose
                    // correctly rounded root reaches th (sqrtf is monotonic; found by the host, score_sq_limit) --
                    // the same decision bit for bit without computing the root
                    const float sq = LOSS == RWH_LOSS_FWD ? proj_sq(h, a.x, a.y, b.x, b.y) : proj_sq(hi, b.x, b.y, a.x, a.y);
                    inl = sq < sq_limit;
                    if (errs) errs[(size_t)hyp * m + j] = sqrtf(sq);
                }
            }
            const unsigned long long bal = __ballot(inl);
            count += __popcll(bal);
            if (masks && lane == 0) masks[(size_t)hyp * mask_stride + w] = bal;
        };
        if constexpr (WORDS > 0) {
#pragma unroll
            for (int w = 0; w < WORDS; ++w)
                if (w < words) score(w, ra[w], rb[w], band[w]);
                else if (masks && w < mask_stride && lane == 0) masks[(size_t)hyp * mask_stride + w] = 0;   // batched: shorter problem
        } else {
            for (int w = 0; w < words; ++w) {
                const int j = w * 64 + lane;
                const float2 a = j < m ? reinterpret_cast<const float2*>(pa)[j] : float2{0.f, 0.f};
                const float2 b = j < m ? reinterpret_cast<const float2*>(pb)[j] : float2{0.f, 0.f};
                score(w, a, b, make_band(a, b));
            }
            if (masks && lane == 0)
                for (int w = words; w < mask_stride; ++w) masks[(size_t)hyp * mask_stride + w] = 0;
        }
        if (lane == 0) {
            counts[hyp] = count;
            if (stop_word && count >= stop_need) atomicMax(stop_word, 0xFFFFFFFFull - (unsigned long long)(hyp - stop_p * k_per));
        }
    }
    hv = hv_next;
    }
}

// K2, filter form (counts + optional masks, M <= 256: the points and their bands stay in registers).  PMC on the
// general kernel above (batched search, 640 000 hypotheses): VALU 89 % busy with the divisions, 59 % with the filter in
// it -- and no faster, because its ~147 SALU instructions per hypothesis (mask bookkeeping and uniform branches per
// 64-pair word) then fill the CU's one scalar issue slot per clock.  This kernel decides all words of a hypothesis with
// straight-line code and branches ONCE, to the exact arithmetic, if any pair of the hypothesis is inside its band.
template <int LOSS, int WORDS, bool MASKS, bool CHUNKED = false>
__global__ __launch_bounds__(256) void score_filter_kernel(const float* __restrict__ hs, const float* __restrict__ pa,
                                                           const float* __restrict__ pb, int m, int k, int hpw, double th,
                                                           float sq_limit, int32_t* __restrict__ counts, uint64_t* __restrict__ masks,
                                                           const int32_t* __restrict__ offsets, int k_per, int mask_stride,
                                                           const int32_t* __restrict__ stop_needs, unsigned long long* stop_keys,
                                                           float thf, const float* __restrict__ hinvs) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    int h_begin, h_end;
    if (offsets) {   // batched mode: a wave stays inside one problem
        const int wpp = (k_per + hpw - 1) / hpw;
        const int p = wid / wpp, chunk = wid - p * wpp;
        if ((long long)p * k_per >= k) return;
        const int base = offsets[p];
        m = offsets[p + 1] - base;
        pa += 2 * (size_t)base; pb += 2 * (size_t)base;
        h_begin = p * k_per + chunk * hpw;
        h_end = min((p + 1) * k_per, h_begin + hpw);
    } else {
        h_begin = wid * hpw;
        if (h_begin >= k) return;
        h_end = min(k, h_begin + hpw);
    }
    // CHUNKED (M > 256, one problem): blockIdx.y picks 256 of the correspondences; every chunk's wave adds its part of a
    // hypothesis' count to counts[] (zeroed by the launcher) and writes its own four mask words -- the points stay in
    // registers and nothing is re-read per hypothesis, whatever M is
    const int chunk = CHUNKED ? (int)blockIdx.y : 0;
    if constexpr (CHUNKED) { pa += 2 * (size_t)(256 * chunk); pb += 2 * (size_t)(256 * chunk); m = min(256, m - 256 * chunk); }
    const int stop_p = stop_keys ? h_begin / k_per : 0;
    const int stop_need = stop_keys ? stop_needs[stop_p] : 0;
    unsigned long long* stop_word = stop_keys ? stop_keys + 2 * (size_t)stop_p + 1 : nullptr;

    float2 ra[WORDS], rb[WORDS];
    Band band[WORDS];
    unsigned long long valid[WORDS];                              // the pairs that exist in each word (uniform)
#pragma unroll
    for (int w = 0; w < WORDS; ++w) {
        const int j = w * 64 + lane;
        ra[w] = j < m ? reinterpret_cast<const float2*>(pa)[j] : float2{0.f, 0.f};
        rb[w] = j < m ? reinterpret_cast<const float2*>(pb)[j] : float2{0.f, 0.f};
        if constexpr (LOSS == RWH_LOSS_FWD) band[w] = band_sq(rb[w].x, rb[w].y, thf);
        else if constexpr (LOSS == RWH_LOSS_BACKWARD) band[w] = band_sq(ra[w].x, ra[w].y, thf);
        else {
            const float mgn = band_margin(ra[w].x, ra[w].y, thf) + band_margin(rb[w].x, rb[w].y, thf);
            band[w] = Band{thf - mgn, thf + mgn};
        }
        const int rem = m - w * 64;
        valid[w] = rem >= 64 ? ~0ull : rem <= 0 ? 0ull : ((1ull << rem) - 1ull);
    }
    constexpr int HB = 7;
    auto fetch = [&](int first) -> float {
        const int n = 9 * min(HB, h_end - first);
        return (first < h_end && lane < n) ? hs[9 * (size_t)first + lane] : 0.f;
    };
    float hv = fetch(h_begin);
    int32_t* cnt_out = counts + h_begin;                          // running output positions: no 64-bit multiplies in the loop
    uint64_t* mask_out = MASKS ? masks + (size_t)h_begin * mask_stride + 4 * chunk : nullptr;
    const int my_words = CHUNKED ? min(4, mask_stride - 4 * chunk) : mask_stride;    // mask words this wave owns per hypothesis
    for (int hyp0 = h_begin; hyp0 < h_end; hyp0 += HB) {
        const float hv_next = fetch(hyp0 + HB);
        const int nb = min(HB, h_end - hyp0);
        // 'backward' / 'reproj' need inv(H) (ransac.py:74): lane q inverts hypothesis q of the block -- once per block
        // instead of once per hypothesis in every lane
        float hiv[9];
        if constexpr (LOSS != RWH_LOSS_FWD) {
            float mine[9];
            const int src = 9 * min(lane, HB - 1);
#pragma unroll
            for (int i = 0; i < 9; ++i) mine[i] = __shfl(hv, src + i);
            if (hinvs) {                       // the caller's inverses: lane q takes hypothesis q's
                const size_t hq = (size_t)min(hyp0 + min(lane, HB - 1), h_end - 1);
#pragma unroll
                for (int i = 0; i < 9; ++i) hiv[i] = hinvs[9 * hq + i];
            } else {
                inverse3(mine, hiv);
            }
        }
#pragma unroll 1
        for (int q = 0; q < nb; ++q, ++cnt_out, mask_out += (MASKS ? mask_stride : 0)) {
            const int hyp = hyp0 + q;
            if (stop_word) {
                const unsigned long long done = __hip_atomic_load(stop_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int local = hyp - stop_p * k_per;
                if (done != 0ull && (long long)(0xFFFFFFFFull - done) < (long long)local) {      // an earlier hypothesis already exits
                    if (lane == 0) *cnt_out = -1;
                    if (MASKS && lane < my_words) mask_out[lane] = 0;
                    continue;
                }
            }
            float h[9], hi[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) h[i] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(hv), 9 * q + i));
            if constexpr (LOSS != RWH_LOSS_FWD) {
#pragma unroll
                for (int i = 0; i < 9; ++i) hi[i] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(hiv[i]), q));
            }
            unsigned long long inb[WORDS], undecided = 0ull;
#pragma unroll
            for (int w = 0; w < WORDS; ++w) {
                // one ballot per comparison (each is the comparison's own SGPR mask), combined on the scalar unit: a ballot
                // of a combined predicate costs two more VALU instructions, and VALU is what bounds this kernel
                unsigned long long b_in, b_sure;
                if constexpr (LOSS == RWH_LOSS_REPROJ) {
                    float d1, d2;
                    const float e = __builtin_amdgcn_sqrtf(proj_sq_fast(h, ra[w].x, ra[w].y, rb[w].x, rb[w].y, d1)) +
                                    __builtin_amdgcn_sqrtf(proj_sq_fast(hi, rb[w].x, rb[w].y, ra[w].x, ra[w].y, d2));
                    b_in = __ballot(e < band[w].lo);
                    b_sure = (b_in | __ballot(e > band[w].hi)) & __ballot(fabsf(d1) > 0x1p-100f) & __ballot(fabsf(d1) < 0x1p100f) &
                             __ballot(fabsf(d2) > 0x1p-100f) & __ballot(fabsf(d2) < 0x1p100f);
                } else {
                    float d;
                    const float sq = LOSS == RWH_LOSS_FWD ? proj_sq_fast(h, ra[w].x, ra[w].y, rb[w].x, rb[w].y, d)
                                                          : proj_sq_fast(hi, rb[w].x, rb[w].y, ra[w].x, ra[w].y, d);
                    b_in = __ballot(sq < band[w].lo);
                    b_sure = (b_in | __ballot(sq > band[w].hi)) & __ballot(fabsf(d) > 0x1p-100f) & __ballot(fabsf(d) < 0x1p100f);
                }
                inb[w] = b_in & valid[w];
                undecided |= ~b_sure & valid[w];
            }
            if (undecided != 0ull) {                              // rare: some pair is inside its band (or not a number)
#pragma unroll
                for (int w = 0; w < WORDS; ++w) {
                    bool inl;
                    if constexpr (LOSS == RWH_LOSS_REPROJ) {
                        float e = proj_err(h, ra[w].x, ra[w].y, rb[w].x, rb[w].y);
                        e = e + proj_err(hi, rb[w].x, rb[w].y, ra[w].x, ra[w].y);
                        inl = (double)e < th;
                    } else {
                        const float sq = LOSS == RWH_LOSS_FWD ? proj_sq(h, ra[w].x, ra[w].y, rb[w].x, rb[w].y)
                                                              : proj_sq(hi, rb[w].x, rb[w].y, ra[w].x, ra[w].y);
                        inl = sq < sq_limit;
                    }
                    inb[w] = __ballot(inl) & valid[w];
                }
            }
            int count = 0;
#pragma unroll
            for (int w = 0; w < WORDS; ++w) count += __popcll(inb[w]);
            if constexpr (MASKS) {
                unsigned long long mine = 0ull;                   // lane w stores word w: one store instruction per hypothesis
#pragma unroll
                for (int w = 0; w < WORDS; ++w) mine = lane == w ? inb[w] : mine;
                if (lane < my_words) mask_out[lane] = mine;
            }
            if (lane == 0) {
                if constexpr (CHUNKED) atomicAdd(cnt_out, count);
                else *cnt_out = count;
                if (stop_word && count >= stop_need) atomicMax(stop_word, 0xFFFFFFFFull - (unsigned long long)(hyp - stop_p * k_per));
            }
        }
        hv = hv_next;
    }
}

// K2b: the accept rules of ransac.py:186-202 in their order-independent form, over the counts K2 wrote.
//   word 0 = max over hypotheses of (count << 32) | (0xFFFFFFFF - index)      -> max count, lowest index on ties
//   word 1 = max of 0xFFFFFFFF - index over hypotheses with count >= need     -> first index reaching `need`
// One block reduces `chunk` hypotheses of one problem (blockIdx.y; k_per == 0: a single problem, indices offset by
// hyp_base) and publishes with at most one atomic per word.  A separate pass because the alternative -- every wave of
// K2 racing atomicMax on the same two words -- costs more than K2's arithmetic once the keys start at zero (+16 us at
// K = 100 000: tools/batched_probe2.py).
__global__ __launch_bounds__(256) void argmax_kernel(const int32_t* __restrict__ counts, int k, int k_per, int chunk, int need,
                                                     const int32_t* __restrict__ needs, long long hyp_base,
                                                     unsigned long long* best) {
    __shared__ unsigned long long red[2][4];
    const int p = blockIdx.y;
    if (k_per) { counts += (size_t)p * k_per; k = k_per; need = needs[p]; best += 2 * p; hyp_base = 0; }
    const int c0 = blockIdx.x * chunk, c1 = min(k, c0 + chunk);
    unsigned long long key0 = 0, key1 = 0;
    for (int i = c0 + (int)threadIdx.x; i < c1; i += 256) {
        const int c = max(counts[i], 0);                      // -1 = skipped after an early exit (RWH_BATCH_EARLY_STOP)
        const unsigned long long inv_idx = 0xFFFFFFFFull - (unsigned long long)(hyp_base + i);
        const unsigned long long key = ((unsigned long long)(unsigned)c << 32) | inv_idx;
        key0 = key > key0 ? key : key0;
        if (c >= need) key1 = inv_idx > key1 ? inv_idx : key1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long a = __shfl_xor(key0, o), b = __shfl_xor(key1, o);
        key0 = a > key0 ? a : key0;
        key1 = b > key1 ? b : key1;
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = key0; red[1][wave] = key1; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            key0 = red[0][w] > key0 ? red[0][w] : key0;
            key1 = red[1][w] > key1 ? red[1][w] : key1;
        }
        if (key0 > __hip_atomic_load(&best[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&best[0], key0);
        if (key1 > __hip_atomic_load(&best[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&best[1], key1);
    }
}

__global__ __launch_bounds__(256) void project_kernel(const float* __restrict__ hs, const float* __restrict__ pts, int m,
                                                      int inverse, float* __restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    float h[9], hi[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) h[i] = hs[i];
    if (inverse) {
        inverse3(h, hi);
#pragma unroll
        for (int i = 0; i < 9; ++i) h[i] = hi[i];
    }
    if (j >= m) return;
    float px, py, pw;
    proj(h, pts[2 * j], pts[2 * j + 1], px, py, pw);
    out[j] = px; out[m + j] = py; out[2 * (size_t)m + j] = pw;
}

// General form of HomoModel.fwd / reproj (ransac.py:55-76): val @ x for a 3 x M x whose third row is the caller's (not
// necessarily 1), in float32 or float64 -- numpy promotes val @ x to float64 when either operand is (model.val is the
// float64 refit after RANSAC.run).  OpenBLAS k-order for both types: rounded multiply, FMA, FMA (checked against numpy
// with exact rational arithmetic); then y / (y[2] + 1e-10) with IEEE divides.
template <typename T>
__global__ __launch_bounds__(256) void project3_kernel(const T* __restrict__ h, const T* __restrict__ pts3, int m, T* __restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const T x = pts3[j], y = pts3[m + j], w = pts3[2 * (size_t)m + j];
    T a[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        T acc = h[3 * r] * x;
        acc = fma(h[3 * r + 1], y, acc);
        a[r] = fma(h[3 * r + 2], w, acc);
    }
    const T den = a[2] + (T)1e-10;
    out[j] = a[0] / den; out[m + j] = a[1] / den; out[2 * (size_t)m + j] = a[2] / den;
}

}  // namespace rwh

extern "C" int rwh_project_points_ex(const void* d_h, const void* d_pts3, int m, int dtype, void* d_out, void* stream) {
    using namespace rwh;
    if (!d_h || !d_pts3 || !d_out || m <= 0) return RWH_E_INVALID;
    if (dtype != RWH_F32 && dtype != RWH_F64) return RWH_E_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == RWH_F32)
        hipLaunchKernelGGL(project3_kernel<float>, dim3((m + 255) / 256), dim3(256), 0, s, static_cast<const float*>(d_h),
                           static_cast<const float*>(d_pts3), m, static_cast<float*>(d_out));
    else
        hipLaunchKernelGGL(project3_kernel<double>, dim3((m + 255) / 256), dim3(256), 0, s, static_cast<const double*>(d_h),
                           static_cast<const double*>(d_pts3), m, static_cast<double*>(d_out));
    return check_launch();
}

extern "C" int rwh_project_points(const float* d_h, const float* d_pts, int m, int inverse, float* d_out, void* stream) {
    using namespace rwh;
    if (!d_h || !d_pts || !d_out || m <= 0) return RWH_E_INVALID;
    hipLaunchKernelGGL(project_kernel, dim3((m + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), d_h, d_pts,
                       m, inverse, d_out);
    return check_launch();
}

namespace rwh {
static void launch_dlt4(hipStream_t s, const float* d_pts_a, const float* d_pts_b, int m, const int32_t* d_idx, int k,
                        float* d_h, uint8_t* d_flags, const int32_t* offsets, int k_per, unsigned long long* reset_keys,
                        int n_reset, int flag_near_singular = 0) {
    const int threads = k > n_reset ? k : n_reset;
    hipLaunchKernelGGL(dlt4_kernel, dim3((threads + 63) / 64), dim3(64), 0, s, d_pts_a, d_pts_b, m, d_idx, k, d_h, d_flags,
                       offsets, k_per, reset_keys, n_reset, flag_near_singular);
}

static void launch_argmax(hipStream_t s, const int32_t* d_counts, int k, int n_problems, int k_per, int need,
                          const int32_t* needs, long long hyp_base, unsigned long long* best) {
    const int per = k_per ? k_per : k;
    const int chunk = per <= 2048 ? per : 2048;   // 8 counts per thread: the pass is latency-bound, not bandwidth-bound
    hipLaunchKernelGGL(argmax_kernel, dim3((per + chunk - 1) / chunk, n_problems), dim3(256), 0, s, d_counts, k, k_per, chunk,
                       need, needs, hyp_base, best);
}
}  // namespace rwh

extern "C" int rwh_dlt4_batched(const float* d_pts_a, const float* d_pts_b, int m, const int32_t* d_idx, int k,
                                float* d_h, uint8_t* d_flags, void* stream) {
    using namespace rwh;
    if (!d_pts_a || !d_pts_b || !d_idx || !d_h || !d_flags || m <= 0 || k < 0) return RWH_E_INVALID;
    if (k == 0) return RWH_OK;
    launch_dlt4(static_cast<hipStream_t>(stream), d_pts_a, d_pts_b, m, d_idx, k, d_h, d_flags, nullptr, 0, nullptr, 0);
    return check_launch();
}

namespace rwh {
// Smallest non-negative float s with (double)sqrtf(s) >= th, so that "sqrtf(s) < th" == "s < limit" for every float s
// (NaN compares false on both sides).  Bisection over the bit patterns of the non-negative floats, on which sqrtf --
// correctly rounded here (glibc) and on the device (-fhip-fp32-correctly-rounded-divide-sqrt) -- is monotonic.
static float score_sq_limit(double th) {
    if (!(th > 0.0)) return 0.0f;                      // th <= 0 or NaN: nothing is ever < th
    uint32_t lo = 0u, hi = 0x7f800000u;                // sqrtf(0) = 0 < th; sqrtf(inf) = inf is not < th (even for th = inf)
    while (hi - lo > 1u) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        float f;
        __builtin_memcpy(&f, &mid, 4);
        if ((double)__builtin_sqrtf(f) < th) lo = mid; else hi = mid;
    }
    float f;
    __builtin_memcpy(&f, &hi, 4);
    return f;
}

template <int LOSS>
void launch_score(int words, dim3 grid, hipStream_t s, const float* d_h, const float* d_pts_a, const float* d_pts_b, int m, int k,
                  int hpw, double th, int32_t* d_counts, uint64_t* d_masks, float* d_err,
                  const int32_t* offsets = nullptr, int k_per = 0, int mask_stride = -1,
                  const int32_t* stop_needs = nullptr, unsigned long long* stop_keys = nullptr, const float* d_hinv = nullptr) {
    const dim3 block(256);
    if (mask_stride < 0) mask_stride = words;
    const float sq_limit = score_sq_limit(th);
    // the filter's band arithmetic wants a positive threshold of ordinary size; anything else: exact path only
    const int filter = (th > 1e-6 && th < 1e6 && !g_score_exact_only) ? 1 : 0;
    const float thf = (float)th;
    if (filter && !d_err && words <= 4) {      // the common case: counts (+ masks) of M <= 256 correspondences
#define RWH_FILTER(W) do { if (d_masks) hipLaunchKernelGGL((score_filter_kernel<LOSS, W, true>), grid, block, 0, s, d_h, d_pts_a, d_pts_b, m, k, \
                                                          hpw, th, sq_limit, d_counts, d_masks, offsets, k_per, mask_stride, stop_needs, stop_keys, thf, d_hinv); \
                           else hipLaunchKernelGGL((score_filter_kernel<LOSS, W, false>), grid, block, 0, s, d_h, d_pts_a, d_pts_b, m, k, \
                                                   hpw, th, sq_limit, d_counts, d_masks, offsets, k_per, mask_stride, stop_needs, stop_keys, thf, d_hinv); } while (0)
        switch (words) {
            case 1: RWH_FILTER(1); break;
            case 2: RWH_FILTER(2); break;
            case 3: RWH_FILTER(3); break;
            default: RWH_FILTER(4); break;
        }
#undef RWH_FILTER
        return;
    }
    if (filter && !d_err && !offsets && !stop_keys) {     // M > 256, one problem: the same kernel per chunk of 256 correspondences
        const int chunks = (m + 255) / 256;
        int hc = (int)((long long)k * chunks / 14000);
        hc = hc < 1 ? 1 : (hc > 14 ? 14 : hc);
        if (g_force_score_hpw) hc = g_force_score_hpw;
        const dim3 cgrid((unsigned)(((k + hc - 1) / hc + 3) / 4), (unsigned)chunks);
        if (chunks <= 65535 && hipMemsetAsync(d_counts, 0, sizeof(int32_t) * (size_t)k, s) == hipSuccess) {
            if (d_masks) hipLaunchKernelGGL((score_filter_kernel<LOSS, 4, true, true>), cgrid, block, 0, s, d_h, d_pts_a, d_pts_b, m, k, hc, th,
                                            sq_limit, d_counts, d_masks, nullptr, 0, mask_stride, nullptr, nullptr, thf, d_hinv);
            else hipLaunchKernelGGL((score_filter_kernel<LOSS, 4, false, true>), cgrid, block, 0, s, d_h, d_pts_a, d_pts_b, m, k, hc, th,
                                    sq_limit, d_counts, d_masks, nullptr, 0, mask_stride, nullptr, nullptr, thf, d_hinv);
            return;
        }
    }
#define RWH_SCORE(W) hipLaunchKernelGGL((score_kernel<LOSS, W>), grid, block, 0, s, d_h, d_pts_a, d_pts_b, m, k, hpw, th, \
                                        sq_limit, d_counts, d_masks, d_err, offsets, k_per, mask_stride, stop_needs, stop_keys, thf, filter, d_hinv)
    switch (words <= 4 ? words : 0) {
        case 1: RWH_SCORE(1); break;
        case 2: RWH_SCORE(2); break;
        case 3: RWH_SCORE(3); break;
        case 4: RWH_SCORE(4); break;
        default: RWH_SCORE(0); break;
    }
#undef RWH_SCORE
}
}  // namespace rwh


namespace rwh {
// ------------------------------------------------------------------------------------------------
// K2i: the inlier count of a hypothesis as an INTERVAL (round 4; 'fwd' loss).  The reference scores H_L -- LAPACK's null
// vector rounded to float32 --, the search scored K1's H_K.  The two differ by a few float32 ulps of each entry's NATURAL
// scale (rows 0 / 1: s, s, s C; row 2: s / C, s / C, s; C = the coordinates' magnitude, s = the largest scale-free entry):
// measured on 13 problem families, 20 000 samples each, 99.9 % within 13 ulps (profiles/r04_lab_notes.txt).  For every pair
// this kernel bounds how far its projected point -- hence its error -- can move when every entry of H moves by delta x its
// natural scale, and counts the pairs that are inliers for EVERY such H (lo) and for SOME such H (hi).  lo == hi: the
// reference's count and inlier mask are K2's, whatever LAPACK rounded to.  A hypothesis whose hi cannot reach the decision
// needs no host solve either.  What this replaces is the flat "+- 8 counts" margin of rounds 2-3, which a wild hypothesis
// (its horizon through the data: one ulp of H moves nine pairs across the threshold) could exceed and a tame one never needs.
// One wavefront per listed hypothesis; delta0: unflagged rows, delta1: RWH_HYP_ILLCOND rows (K1 accurate, LAPACK noisier).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void score_interval_kernel(const float* __restrict__ hs, const int32_t* __restrict__ rows, int n_rows,
                                                             const uint8_t* __restrict__ flags, const float* __restrict__ pa,
                                                             const float* __restrict__ pb, int m, double th, double cscale,
                                                             double delta0, double delta1, int32_t* __restrict__ lo_out,
                                                             int32_t* __restrict__ hi_out) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= n_rows) return;
    const int row = rows ? rows[w] : w;
    float h[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) h[i] = hs[9 * (size_t)row + i];
    const double delta = (flags && (flags[row] & RWH_HYP_ILLCOND)) ? delta1 : delta0;
    const double C = cscale, rC = 1.0 / cscale;
    double a[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) a[i] = fabs((double)h[i]);
    double s = fmax(fmax(fmax(a[0], a[1]), fmax(a[3], a[4])), a[8]);
    s = fmax(s, fmax(fmax(a[2], a[5]) * rC, fmax(a[6], a[7]) * C));
    double D[9];
    D[0] = delta * fmax(a[0], s); D[1] = delta * fmax(a[1], s); D[2] = delta * fmax(a[2], s * C);
    D[3] = delta * fmax(a[3], s); D[4] = delta * fmax(a[4], s); D[5] = delta * fmax(a[5], s * C);
    D[6] = delta * fmax(a[6], s * rC); D[7] = delta * fmax(a[7], s * rC); D[8] = delta * fmax(a[8], s);
    int n_in = 0, n_out = 0;
    for (int j = lane; j < m; j += 64) {
        const float2 A = reinterpret_cast<const float2*>(pa)[j], B = reinterpret_cast<const float2*>(pb)[j];
        float px, py, pw;
        proj(h, A.x, A.y, px, py, pw);                          // the reference's arithmetic (K2's)
        const float dx = px - B.x, dy = py - B.y;
        const float err = sqrtf(dx * dx + dy * dy);
        float a2 = h[6] * A.x; a2 = fmaf(h[7], A.y, a2); a2 = a2 + h[8];
        const double den = fabs((double)(a2 + 1e-10f));
        const double ax = fabs((double)A.x), ay = fabs((double)A.y);
        const double d0 = D[0] * ax + D[1] * ay + D[2], d1 = D[3] * ax + D[4] * ay + D[5], d2 = D[6] * ax + D[7] * ay + D[8];
        const double room = den - d2;
        double mg = ((d0 + fabs((double)px) * d2) + (d1 + fabs((double)py) * d2)) / room;
        // + the float32 evaluation's own round-off at a neighbouring H (a few ulps of the quantities subtracted)
        mg += 0x1p-21 * (fabs((double)px) + fabs((double)py) + fabs((double)B.x) + fabs((double)B.y));
        const bool ok = (room > 0.0) & (mg == mg) & (mg < 1e300) & (err == err) & (fabsf(err) < 3.0e38f);
        const double e = (double)err;
        n_in += (int)(ok & (e + mg < th));
        n_out += (int)(ok & (e - mg >= th));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { n_in += __shfl_xor(n_in, o); n_out += __shfl_xor(n_out, o); }
    if (lane == 0) { lo_out[w] = n_in; hi_out[w] = m - n_out; }
}
}  // namespace rwh

extern "C" int rwh_score_interval(const float* d_h, const int32_t* d_rows, int n_rows, const uint8_t* d_flags, const float* d_pts_a,
                                  const float* d_pts_b, int m, double th, double coord_scale, double delta0, double delta1,
                                  int32_t* d_lo, int32_t* d_hi, void* stream) {
    using namespace rwh;
    if (!d_h || !d_pts_a || !d_pts_b || !d_lo || !d_hi || m <= 0 || n_rows < 0) return RWH_E_INVALID;
    if (!(coord_scale >= 1.0) || !(delta0 >= 0.0) || !(delta1 >= 0.0)) return RWH_E_INVALID;
    if (n_rows == 0) return RWH_OK;
    hipLaunchKernelGGL(score_interval_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), d_h, d_rows, n_rows,
                       d_flags, d_pts_a, d_pts_b, m, th, coord_scale, delta0, delta1, d_lo, d_hi);
    return check_launch();
}

extern "C" int rwh_score_count_inv(const float* d_h, const float* d_hinv, const float* d_pts_a, const float* d_pts_b, int m, int k,
                                   double th, int loss, int need, int64_t hyp_base, int32_t* d_counts, uint64_t* d_masks,
                                   uint64_t* d_best, float* d_err, void* stream);

extern "C" int rwh_score_count(const float* d_h, const float* d_pts_a, const float* d_pts_b, int m, int k, double th,
                               int loss, int need, int64_t hyp_base, int32_t* d_counts, uint64_t* d_masks,
                               uint64_t* d_best, float* d_err, void* stream) {
    return rwh_score_count_inv(d_h, nullptr, d_pts_a, d_pts_b, m, k, th, loss, need, hyp_base, d_counts, d_masks, d_best, d_err, stream);
}

extern "C" int rwh_score_count_inv(const float* d_h, const float* d_hinv, const float* d_pts_a, const float* d_pts_b, int m, int k,
                                   double th, int loss, int need, int64_t hyp_base, int32_t* d_counts, uint64_t* d_masks,
                                   uint64_t* d_best, float* d_err, void* stream) {
    using namespace rwh;
    if (!d_h || !d_pts_a || !d_pts_b || !d_counts || !d_best || m <= 0 || k < 0 || hyp_base < 0) return RWH_E_INVALID;
    if (hyp_base + k > 0xFFFFFFFFll) return RWH_E_INVALID;
    if (loss < RWH_LOSS_FWD || loss > RWH_LOSS_REPROJ) return RWH_E_INVALID;
    if (k == 0) return RWH_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // hypotheses per wave: enough waves to fill the chip (>= ~8 per SIMD) before a wave gets more than one
    int hpw = k / 14000;            // measured (tools/k2_hpw_sweep.py): 1 up to ~20 k hypotheses, 7 at 100 k
    hpw = hpw < 1 ? 1 : (hpw > 14 ? 14 : hpw);   // (the kernel fetches H in blocks of 7 hypotheses)
    if (g_force_score_hpw) hpw = g_force_score_hpw;   // lab override (rwh_lab_tune)
    const int waves = (k + hpw - 1) / hpw;
    const dim3 grid((waves + 3) / 4);
    const int words = (m + 63) / 64;
    if (loss == RWH_LOSS_FWD)
        launch_score<RWH_LOSS_FWD>(words, grid, s, d_h, d_pts_a, d_pts_b, m, k, hpw, th, d_counts, d_masks, d_err);
    else if (loss == RWH_LOSS_BACKWARD)
        launch_score<RWH_LOSS_BACKWARD>(words, grid, s, d_h, d_pts_a, d_pts_b, m, k, hpw, th, d_counts, d_masks, d_err, nullptr, 0, -1, nullptr, nullptr, d_hinv);
    else
        launch_score<RWH_LOSS_REPROJ>(words, grid, s, d_h, d_pts_a, d_pts_b, m, k, hpw, th, d_counts, d_masks, d_err, nullptr, 0, -1, nullptr, nullptr, d_hinv);
    launch_argmax(s, d_counts, k, 1, 0, need, nullptr, (long long)hyp_base, reinterpret_cast<unsigned long long*>(d_best));
    return check_launch();
}

extern "C" int rwh_ransac_search(const float* d_pts_a, const float* d_pts_b, int m, const int32_t* d_idx, int k, double th,
                                 int loss, int need, int64_t hyp_base, float* d_h, uint8_t* d_flags, int32_t* d_counts,
                                 uint64_t* d_masks, uint64_t* d_best, int reset_best, void* stream) {
    using namespace rwh;
    if (!d_best || !d_pts_a || !d_pts_b || !d_idx || !d_h || !d_flags || m <= 0 || k < 0) return RWH_E_INVALID;
    // the DLT launch also clears the two keys (saves a memset launch per search)
    if (k > 0 || reset_best)
        launch_dlt4(static_cast<hipStream_t>(stream), d_pts_a, d_pts_b, m, d_idx, k, d_h, d_flags, nullptr, 0,
                    reset_best ? reinterpret_cast<unsigned long long*>(d_best) : nullptr, reset_best ? 2 : 0, loss != RWH_LOSS_FWD);
    const int st = check_launch();
    if (st != RWH_OK) return st;
    return rwh_score_count(d_h, d_pts_a, d_pts_b, m, k, th, loss, need, hyp_base, d_counts, d_masks, d_best, nullptr, stream);
}

extern "C" int rwh_ransac_batched(const float* d_pts_a, const float* d_pts_b, const int32_t* d_offsets, int n_problems,
                                  int m_max, int k, int32_t* d_idx, uint64_t seed, int64_t problem_base, double th,
                                  int loss, const int32_t* d_need, float* d_h, uint8_t* d_flags, int32_t* d_counts,
                                  uint64_t* d_masks, uint64_t* d_best, unsigned flags, void* stream) {
    using namespace rwh;
    if (!d_pts_a || !d_pts_b || !d_offsets || !d_idx || !d_need || !d_h || !d_flags || !d_counts || !d_best) return RWH_E_INVALID;
    if (n_problems < 0 || k < 0 || m_max <= 0) return RWH_E_INVALID;
    if (problem_base < 0 || problem_base + n_problems > 0xFFFFFFFFll) return RWH_E_INVALID;
    if (loss < RWH_LOSS_FWD || loss > RWH_LOSS_REPROJ) return RWH_E_INVALID;
    if (n_problems == 0 || k == 0) return RWH_OK;
    const long long total = (long long)n_problems * k;
    if (total >= (1ll << 31)) return RWH_E_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (flags & RWH_BATCH_DEVICE_SAMPLING)
        hipLaunchKernelGGL(sample4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_offsets, n_problems, k,
                           (unsigned long long)seed, (unsigned)problem_base, d_idx);
    launch_dlt4(s, d_pts_a, d_pts_b, m_max, d_idx, (int)total, d_h, d_flags, d_offsets, k,
                reinterpret_cast<unsigned long long*>(d_best), 2 * n_problems, loss != RWH_LOSS_FWD);   // also clears the P x 2 keys
    int hpw = (int)(total / 14000);
    hpw = hpw < 1 ? 1 : (hpw > 14 ? 14 : hpw);
    if (g_force_score_hpw) hpw = g_force_score_hpw;   // lab override (rwh_lab_tune)
    if (hpw > k) hpw = k;
    const long long waves = (long long)n_problems * ((k + hpw - 1) / hpw);
    const dim3 grid((unsigned)((waves + 3) / 4));
    const int words = (m_max + 63) / 64;
    const int32_t* sn = (flags & RWH_BATCH_EARLY_STOP) ? d_need : nullptr;
    unsigned long long* sk = (flags & RWH_BATCH_EARLY_STOP) ? reinterpret_cast<unsigned long long*>(d_best) : nullptr;
    if (loss == RWH_LOSS_FWD)
        launch_score<RWH_LOSS_FWD>(words, grid, s, d_h, d_pts_a, d_pts_b, m_max, (int)total, hpw, th, d_counts, d_masks, nullptr, d_offsets, k, words, sn, sk);
    else if (loss == RWH_LOSS_BACKWARD)
        launch_score<RWH_LOSS_BACKWARD>(words, grid, s, d_h, d_pts_a, d_pts_b, m_max, (int)total, hpw, th, d_counts, d_masks, nullptr, d_offsets, k, words, sn, sk);
    else
        launch_score<RWH_LOSS_REPROJ>(words, grid, s, d_h, d_pts_a, d_pts_b, m_max, (int)total, hpw, th, d_counts, d_masks, nullptr, d_offsets, k, words, sn, sk);
    launch_argmax(s, d_counts, (int)total, n_problems, k, 0, d_need, 0, reinterpret_cast<unsigned long long*>(d_best));
    return check_launch();
}
