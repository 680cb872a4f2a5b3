// K3: per-output-pixel inverse-homography backward warp (nearest / bilinear) for gfx950.
//
// Replaces the numpy pipeline of homography.py:166-179 / 197-208 + 108-138 of the
// reference (grid -> inv(H) @ z -> divide -> mask -> gather -> lerp).  See include/rwh.h
// for the contract.  Design notes (DESIGN.md has the long form):
//   * one lane owns 4 consecutive output pixels of one row -> 12 B (RGB u8) contiguous
//     per lane, 768 B per wave store instruction;
//   * source coordinates are float64 per pixel (3 FMA + v_rcp_f64 + Newton + 2 MUL): fp32
//     coordinates at x ~ 4000-8000 are off by 2-5e-4 px, which breaks the 1e-4 budget;
//   * bilinear weights w and 1-w are both rounded from float64, the blend is float32;
//   * RGB u8 taps of one source row are ONE unaligned 8-byte load (6 useful bytes);
//     the rare wave that touches the last source rows takes a byte-exact guarded path,
//     so nothing is ever read past the image;
//   * blockIdx is remapped so that each XCD (blocks b, b+8, ...) walks a contiguous band
//     of output tile rows: vertically adjacent tiles share source rows in one L2;
//   * no MFMA: there is no dense contraction in this path.
#include "rwh_common.h"

namespace rwh {

struct WarpArgs {
    const unsigned char* src;
    unsigned char* dst;
    long long src_img_stride, dst_img_stride;  // bytes
    double ih[9];
    double x0, step_x, x_last, y0, step_y, y_last;
    int src_h, src_w;      // addressing
    int bound_h, bound_w;  // bounds test (<= src size)
    int out_h, out_w;
    int row_begin, rows;   // produce rows [row_begin, row_begin+rows)
    unsigned tiles_x, tiles_y, nblocks, cpx;
    double dxs[3][3];      // dxs[j-1] = j*step_x * (ih[0], ih[3], ih[6]): X,Y,W increments of pixel j of a lane
    unsigned long long xmax_bits, ymax_bits;  // bit patterns of MAGIC + (bound_w-1), MAGIC + (bound_h-1)
};

constexpr int PX = 4;        // pixels per lane
constexpr int TILE_ROWS = 4; // waves per block, one output row each

template <typename T> struct elem;
template <> struct elem<unsigned char> { static constexpr int dtype = RWH_U8; };
template <> struct elem<float> { static constexpr int dtype = RWH_F32; };

// ---- block -> tile decode with XCD-contiguous remap -------------------------------------------
__device__ __forceinline__ bool decode_tile(const WarpArgs& a, unsigned& tx, unsigned& ty, unsigned& img) {
    const unsigned b = blockIdx.x;
    const unsigned logical = (b & 7u) * a.cpx + (b >> 3);
    if (logical >= a.nblocks) return false;
    tx = logical % a.tiles_x;
    const unsigned t = logical / a.tiles_x;
    ty = t % a.tiles_y;
    img = t / a.tiles_y;
    return true;
}

__device__ __forceinline__ double grid_coord(int i, int n, double c0, double step, double last) {
    // numpy.linspace: arange(n)*step + start, endpoint forced to `stop`
    return (i == n - 1) ? last : (double)i * step + c0;
}

// Source coordinate of output (x, y-row terms) in float64.  v_rcp_f64 + two Newton steps
// (relative error ~1e-16 after the second; the first alone leaves ~2^-46).
__device__ __forceinline__ void project(const WarpArgs& a, double x, double rx, double ry, double rw,
                                        double& sx, double& sy) {
    const double X = fma(a.ih[0], x, rx);
    const double Y = fma(a.ih[3], x, ry);
    const double W = fma(a.ih[6], x, rw);
    double r = __builtin_amdgcn_rcp(W);
    r = fma(fma(-W, r, 1.0), r, r);
    r = fma(fma(-W, r, 1.0), r, r);
    sx = X * r;
    sy = Y * r;
}

__device__ __forceinline__ float ub(uint32_t v, int byte) { return (float)((v >> (8 * byte)) & 0xffu); }

// ---- texel access for the generic path ---------------------------------------------------------
template <typename SrcT, int C>
__device__ __forceinline__ void load_texel(const unsigned char* img, int src_w, int iy, int ix, float (&t)[C]) {
    const size_t off = ((size_t)iy * (size_t)src_w + (size_t)ix) * (size_t)(C * sizeof(SrcT));
    if constexpr (sizeof(SrcT) == 1) {
        if constexpr (C == 4) {
            const uint32_t v = ld4(img + off);
#pragma unroll
            for (int c = 0; c < 4; ++c) t[c] = ub(v, c);
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) t[c] = (float)img[off + c];
        }
    } else {
        const float* p = reinterpret_cast<const float*>(img + off);
#pragma unroll
        for (int c = 0; c < C; ++c) t[c] = p[c];
    }
}

template <typename DstT>
__device__ __forceinline__ DstT to_dst(float v) {
    if constexpr (sizeof(DstT) == 1) return (unsigned char)(unsigned)v;  // truncation == astype(uint8)
    else return v;
}

// ================================================================================================
// Generic kernel: any of {u8,f32} x {3,4} channels, nearest or bilinear.  Texel-exact loads with
// clamped +1 taps (their weight is 0 whenever the clamp acts), so no read ever leaves the image.
// ================================================================================================
template <typename SrcT, int C, typename DstT, int INTERP>
__global__ __launch_bounds__(256) void warp_generic(const WarpArgs a) {
    unsigned tx, ty, img;
    if (!decode_tile(a, tx, ty, img)) return;
    const int lane = threadIdx.x & 63, wrow = threadIdx.x >> 6;
    const int rr = (int)ty * TILE_ROWS + wrow;
    if (rr >= a.rows) return;
    const int r = a.row_begin + rr;
    const int c0 = ((int)tx * RWH_WAVE + lane) * PX;
    if (c0 >= a.out_w) return;

    const unsigned char* simg = a.src + (long long)img * a.src_img_stride;
    DstT* drow = reinterpret_cast<DstT*>(a.dst + (long long)img * a.dst_img_stride) +
                 ((size_t)rr * (size_t)a.out_w + (size_t)c0) * C;

    const double y = grid_coord(r, a.out_h, a.y0, a.step_y, a.y_last);
    const double rx = fma(a.ih[1], y, a.ih[2]);
    const double ry = fma(a.ih[4], y, a.ih[5]);
    const double rw = fma(a.ih[7], y, a.ih[8]);
    const double bw1 = (double)(a.bound_w - 1), bh1 = (double)(a.bound_h - 1);

#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int c = c0 + j;
        if (c >= a.out_w) break;
        const double x = grid_coord(c, a.out_w, a.x0, a.step_x, a.x_last);
        double sx, sy;
        project(a, x, rx, ry, rw, sx, sy);
        float o[C];
        if constexpr (INTERP == RWH_NEAREST) {
            // homography.py:110,117: trunc(coord + 0.5) as int32, mask on the integers
            const int xi = (int)(sx + 0.5), yi = (int)(sy + 0.5);
            const bool valid = (xi >= 0) & (xi <= a.bound_w - 1) & (yi >= 0) & (yi <= a.bound_h - 1);
            if (valid) {
                load_texel<SrcT, C>(simg, a.src_w, yi, xi, o);
            } else {
#pragma unroll
                for (int k = 0; k < C; ++k) o[k] = 0.f;
            }
        } else {
            // homography.py:131-137: mask on the float coords, truncate, lerp x then y
            const bool valid = (sx >= 0.0) & (sx <= bw1) & (sy >= 0.0) & (sy <= bh1);
            if (valid) {
                const int ix = (int)sx, iy = (int)sy;
                const double fx = sx - (double)ix, fy = sy - (double)iy;
                const float wx1 = (float)fx, wx0 = (float)(1.0 - fx);
                const float wy1 = (float)fy, wy0 = (float)(1.0 - fy);
                const int ix1 = min(ix + 1, a.src_w - 1), iy1 = min(iy + 1, a.src_h - 1);
                float p00[C], p01[C], p10[C], p11[C];
                load_texel<SrcT, C>(simg, a.src_w, iy, ix, p00);
                load_texel<SrcT, C>(simg, a.src_w, iy, ix1, p01);
                load_texel<SrcT, C>(simg, a.src_w, iy1, ix, p10);
                load_texel<SrcT, C>(simg, a.src_w, iy1, ix1, p11);
#pragma unroll
                for (int k = 0; k < C; ++k) {
                    const float top = fmaf(p01[k], wx1, p00[k] * wx0);
                    const float bot = fmaf(p11[k], wx1, p10[k] * wx0);
                    o[k] = fmaf(bot, wy1, top * wy0);
                }
            } else {
#pragma unroll
                for (int k = 0; k < C; ++k) o[k] = 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < C; ++k) drow[j * C + k] = to_dst<DstT>(o[k]);
    }
}

// ================================================================================================
// Fast kernel: RGB u8 source, bilinear, u8 (truncated) or f32 output -- the BASELINE configuration.
// ================================================================================================
struct Tap {       // per-pixel state kept in registers between the coordinate and the load phase
    uint32_t off;  // byte offset of texel (iy, ix) in the image
    float wx0, wx1, wy0, wy1;
};

template <typename DstT>
__device__ __forceinline__ void store4_rgb(DstT* d, const float (&o)[PX][3], int npx) {
    if constexpr (sizeof(DstT) == 1) {
        uint32_t q[PX][3];
#pragma unroll
        for (int j = 0; j < PX; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) q[j][k] = (uint32_t)o[j][k];  // v_cvt_u32_f32 truncates
        if (npx == PX) {
            pk3 w;
            w.a = q[0][0] | (q[0][1] << 8) | (q[0][2] << 16) | (q[1][0] << 24);
            w.b = q[1][1] | (q[1][2] << 8) | (q[2][0] << 16) | (q[2][1] << 24);
            w.c = q[2][2] | (q[3][0] << 8) | (q[3][1] << 16) | (q[3][2] << 24);
            __builtin_memcpy(d, &w, 12);
        } else {
#pragma unroll
            for (int j = 0; j < PX; ++j)
                if (j < npx) {
                    d[3 * j + 0] = (unsigned char)q[j][0];
                    d[3 * j + 1] = (unsigned char)q[j][1];
                    d[3 * j + 2] = (unsigned char)q[j][2];
                }
        }
    } else {
        if (npx == PX) {
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                pk4 w;
                const float* f = &o[0][0] + 4 * v;
                w.a = __float_as_uint(f[0]); w.b = __float_as_uint(f[1]);
                w.c = __float_as_uint(f[2]); w.d = __float_as_uint(f[3]);
                __builtin_memcpy(reinterpret_cast<unsigned char*>(d) + 16 * v, &w, 16);
            }
        } else {
#pragma unroll
            for (int j = 0; j < PX; ++j)
                if (j < npx) { d[3 * j] = o[j][0]; d[3 * j + 1] = o[j][1]; d[3 * j + 2] = o[j][2]; }
        }
    }
}

__device__ __forceinline__ void blend_rgb(const Tap& t, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1,
                                          float (&o)[3]) {
    // row bytes: a = [R0 G0 B0 R1], b = [G1 B1 . .]
    const float p00[3] = {ub(a0, 0), ub(a0, 1), ub(a0, 2)};
    const float p01[3] = {ub(a0, 3), ub(b0, 0), ub(b0, 1)};
    const float p10[3] = {ub(a1, 0), ub(a1, 1), ub(a1, 2)};
    const float p11[3] = {ub(a1, 3), ub(b1, 0), ub(b1, 1)};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float top = fmaf(p01[k], t.wx1, p00[k] * t.wx0);
        const float bot = fmaf(p11[k], t.wx1, p10[k] * t.wx0);
        o[k] = fmaf(bot, t.wy1, top * t.wy0);
    }
}

template <typename DstT>
__global__ __launch_bounds__(256) void warp_rgb8_bilinear(const WarpArgs a) {
    unsigned tx, ty, img;
    if (!decode_tile(a, tx, ty, img)) return;
    const int lane = threadIdx.x & 63, wrow = threadIdx.x >> 6;
    const int rr = (int)ty * TILE_ROWS + wrow;
    if (rr >= a.rows) return;
    const int r = a.row_begin + rr;
    const int c0 = ((int)tx * RWH_WAVE + lane) * PX;
    if (c0 >= a.out_w) return;
    const int npx = min(PX, a.out_w - c0);

    const unsigned char* simg = a.src + (long long)img * a.src_img_stride;
    DstT* drow = reinterpret_cast<DstT*>(a.dst + (long long)img * a.dst_img_stride) +
                 ((size_t)rr * (size_t)a.out_w + (size_t)c0) * 3;

    const double y = grid_coord(r, a.out_h, a.y0, a.step_y, a.y_last);
    const double rx = fma(a.ih[1], y, a.ih[2]);
    const double ry = fma(a.ih[4], y, a.ih[5]);
    const double rw = fma(a.ih[7], y, a.ih[8]);
    const double bw1 = (double)(a.bound_w - 1), bh1 = (double)(a.bound_h - 1);
    const uint32_t pitch = (uint32_t)a.src_w * 3u;

    Tap t[PX];
    bool near_end = false;  // some tap row of this lane is one of the last two source rows
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int c = min(c0 + j, a.out_w - 1);  // lanes past the row end recompute the last pixel
        const double x = grid_coord(c, a.out_w, a.x0, a.step_x, a.x_last);
        double sx, sy;
        project(a, x, rx, ry, rw, sx, sy);
        const bool valid = (sx >= 0.0) & (sx <= bw1) & (sy >= 0.0) & (sy <= bh1);
        const int ix = valid ? (int)sx : 0;
        const int iy = valid ? (int)sy : 0;
        const double fx = __builtin_amdgcn_fract(sx), fy = __builtin_amdgcn_fract(sy);
        t[j].wx1 = (float)fx;
        t[j].wx0 = (float)(1.0 - fx);
        t[j].wy1 = valid ? (float)fy : 0.f;
        t[j].wy0 = valid ? (float)(1.0 - fy) : 0.f;
        t[j].off = ((uint32_t)iy * (uint32_t)a.src_w + (uint32_t)ix) * 3u;
        near_end |= (iy > a.src_h - 3);
    }

    float o[PX][3];
    if (!__any(near_end)) {
        // fast path: rows iy and iy+1 are both above the last source row, so the 8-byte
        // loads (6 bytes used) stay inside the image even at the right edge.
        pk2 r0[PX], r1[PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            r0[j] = ld8(simg + t[j].off);
            r1[j] = ld8(simg + t[j].off + pitch);
        }
#pragma unroll
        for (int j = 0; j < PX; ++j) blend_rgb(t[j], r0[j].a, r0[j].b, r1[j].a, r1[j].b, o[j]);
    } else {
        // guarded path: byte-exact loads, +1 taps clamped to the image (weight 0 when clamped)
        const uint32_t last = (uint32_t)a.src_h * pitch - 3u;  // offset of the last texel
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const uint32_t o00 = t[j].off;
            const uint32_t o01 = min(o00 + 3u, last);
            const uint32_t o10 = min(o00 + pitch, last);
            const uint32_t o11 = min(o00 + pitch + 3u, last);
            const uint32_t a0 = simg[o00] | (simg[o00 + 1] << 8) | (simg[o00 + 2] << 16) | (simg[o01] << 24);
            const uint32_t b0 = simg[o01 + 1] | (simg[o01 + 2] << 8);
            const uint32_t a1 = simg[o10] | (simg[o10 + 1] << 8) | (simg[o10 + 2] << 16) | (simg[o11] << 24);
            const uint32_t b1 = simg[o11 + 1] | (simg[o11 + 2] << 8);
            blend_rgb(t[j], a0, b0, a1, b1, o[j]);
        }
    }
    store4_rgb<DstT>(drow, o, npx);
}


// ================================================================================================
// Fast kernel v2.  The v1 kernel above is VALU-issue bound (every VALU instruction, f32 or f64,
// costs ~4.4 cycles per wave on gfx950; v1 spends ~87 of them per pixel).  v2 removes instructions:
//   * X, Y, W advance by host-precomputed increments (1 add each) instead of cvt + mul + fma;
//   * ONE v_rcp_f64 per lane: the four W of a lane are inverted together (Montgomery batch
//     inversion: 9 multiplies + 1 rcp + 1 Newton step for 4 reciprocals);
//   * floor / fraction / weights come from the "magic number" trick: u = s + 1.5*2^20 has
//     ulp 2^-32, so hi(u) - 0x41380000 = floor(s) and lo(u) = frac(s) * 2^32 (rounded to nearest,
//     error 1.2e-10 px); w = cvt(lo) and 1-w = cvt(~lo) are both correct to float32 rounding, and
//     the 2^-64 scale of (x weight) * (y weight) rides on the y weights;
//   * a wave whose first and last pixel are strictly inside the image (coordinates are monotone
//     along an output row while W keeps its sign) needs NO per-pixel bounds logic at all; only waves
//     that touch the border pay for masks and the end-of-buffer guard;
//   * the blend uses the 4-weight form on packed float32 pairs (v_pk_mul/fma_f32 do two lanes' worth
//     for ~1.15x the cost of one), pairing pixel j with pixel j+1 so both halves are always useful;
//   * v_cvt_pk_u8_f32 converts and packs an output byte in one instruction.
// ================================================================================================
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr double MAGIC = 1572864.0;               // 1.5 * 2^20
constexpr uint32_t MAGIC_HI = 0x41380000u;        // high dword of MAGIC
constexpr unsigned long long MAGIC_BITS = 0x4138000000000000ull;
constexpr float W_SCALE = 5.42101086242752217e-20f;  // 2^-64

__device__ __forceinline__ uint32_t hi32(double v) { return (uint32_t)(__double_as_longlong(v) >> 32); }
__device__ __forceinline__ uint32_t lo32(double v) { return (uint32_t)__double_as_longlong(v); }

template <typename DstT, bool PKU8>
__global__ __launch_bounds__(256) void warp_rgb8_bilinear2(const WarpArgs a) {
    unsigned tx, ty, img;
    if (!decode_tile(a, tx, ty, img)) return;
    const int lane = threadIdx.x & 63, wrow = threadIdx.x >> 6;
    const int rr = (int)ty * TILE_ROWS + wrow;
    if (rr >= a.rows) return;                       // wave-uniform
    const int r = a.row_begin + rr;
    const int c0 = ((int)tx * RWH_WAVE + lane) * PX;
    // Lanes at / past the row end recompute the last four pixels of the row (keeps every lane's
    // coordinates inside the grid and monotone along the wave); they store only what is theirs.
    const int c0p = min(c0, a.out_w - PX);
    const int shift = c0 - c0p;                     // 0 for full lanes

    const unsigned char* simg = a.src + (long long)img * a.src_img_stride;
    DstT* drow = reinterpret_cast<DstT*>(a.dst + (long long)img * a.dst_img_stride) +
                 ((size_t)rr * (size_t)a.out_w + (size_t)c0p) * 3;

    const double y = fma((double)r, a.step_y, a.y0);
    const double xb = fma((double)c0p, a.step_x, a.x0);
    double X[PX], Y[PX], W[PX];
    X[0] = fma(a.ih[0], xb, fma(a.ih[1], y, a.ih[2]));
    Y[0] = fma(a.ih[3], xb, fma(a.ih[4], y, a.ih[5]));
    W[0] = fma(a.ih[6], xb, fma(a.ih[7], y, a.ih[8]));
#pragma unroll
    for (int j = 1; j < PX; ++j) {
        X[j] = X[0] + a.dxs[j - 1][0];
        Y[j] = Y[0] + a.dxs[j - 1][1];
        W[j] = W[0] + a.dxs[j - 1][2];
    }
    // four reciprocals from one v_rcp_f64
    double rc[PX];
    {
        const double p01 = W[0] * W[1], p23 = W[2] * W[3], P = p01 * p23;
        // normal, finite product (either sign): class mask = -normal | +normal
        if (__all(__builtin_amdgcn_class(P, 0x008 | 0x100))) {
            double rp = __builtin_amdgcn_rcp(P);
            rp = fma(fma(-P, rp, 1.0), rp, rp);
            const double r01 = rp * p23, r23 = rp * p01;
            rc[0] = r01 * W[1]; rc[1] = r01 * W[0]; rc[2] = r23 * W[3]; rc[3] = r23 * W[2];
        } else {  // a W at / across zero (the horizon) in this wave: invert pixel by pixel
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                double q = __builtin_amdgcn_rcp(W[j]);
                rc[j] = fma(fma(-W[j], q, 1.0), q, q);
            }
        }
    }
    uint32_t lx[PX], ly[PX];
    int ix[PX], iy[PX];
    unsigned long long ubx[PX], uby[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const double ux = X[j] * rc[j] + MAGIC;
        const double uy = Y[j] * rc[j] + MAGIC;
        ubx[j] = (unsigned long long)__double_as_longlong(ux);
        uby[j] = (unsigned long long)__double_as_longlong(uy);
        lx[j] = lo32(ux); ly[j] = lo32(uy);
        ix[j] = (int)(hi32(ux) - MAGIC_HI);
        iy[j] = (int)(hi32(uy) - MAGIC_HI);
    }
    const uint32_t pitch = (uint32_t)a.src_w * 3u;

    // ---- wave-uniform interior test on the two end pixels of the wave's row segment -------------
    const unsigned lim_x = (unsigned)(a.bound_w - 1);
    const unsigned lim_y = (unsigned)min(a.bound_h - 1, a.src_h - 2);
    const unsigned ixF = (unsigned)__builtin_amdgcn_readlane(ix[0], 0), ixL = (unsigned)__builtin_amdgcn_readlane(ix[PX - 1], 63);
    const unsigned iyF = (unsigned)__builtin_amdgcn_readlane(iy[0], 0), iyL = (unsigned)__builtin_amdgcn_readlane(iy[PX - 1], 63);
    const int wF = __builtin_amdgcn_readlane((int)hi32(W[0]), 0), wL = __builtin_amdgcn_readlane((int)hi32(W[PX - 1]), 63);
    const bool interior = (ixF < lim_x) & (ixL < lim_x) & (iyF < lim_y) & (iyL < lim_y) & (wF > 0) & (wL > 0);

    float wx0[PX], wx1[PX], wy0[PX], wy1[PX];
    uint32_t off[PX];
    bool guard = false;
    if (interior) {
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            wx1[j] = (float)lx[j]; wx0[j] = (float)(~lx[j]);
            wy1[j] = (float)ly[j] * W_SCALE; wy0[j] = (float)(~ly[j]) * W_SCALE;
            off[j] = (uint32_t)iy[j] * pitch + (uint32_t)ix[j] * 3u;
        }
    } else {
        bool near_end = false;
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            // 0 <= s <= bound-1 on the bit patterns: positive doubles order like unsigned integers,
            // negative values and NaNs have patterns above xmax_bits or below MAGIC_BITS
            const bool valid = (ubx[j] >= MAGIC_BITS) & (ubx[j] <= a.xmax_bits) & (uby[j] >= MAGIC_BITS) & (uby[j] <= a.ymax_bits);
            const float sc = valid ? W_SCALE : 0.f;
            wx1[j] = (float)lx[j]; wx0[j] = (float)(~lx[j]);
            wy1[j] = (float)ly[j] * sc; wy0[j] = (float)(~ly[j]) * sc;
            off[j] = valid ? (uint32_t)iy[j] * pitch + (uint32_t)ix[j] * 3u : 0u;
            near_end |= valid & (iy[j] > a.src_h - 3);
        }
        guard = __any(near_end);
    }

    uint32_t a0[PX], b0[PX], a1[PX], b1[PX];   // row bytes: a = [R0 G0 B0 R1], b = [G1 B1 . .]
    if (!guard) {
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const pk2 r0 = ld8(simg + off[j]);
            const pk2 r1 = ld8(simg + off[j] + pitch);
            a0[j] = r0.a; b0[j] = r0.b; a1[j] = r1.a; b1[j] = r1.b;
        }
    } else {  // byte-exact loads, +1 taps clamped to the image (their weight is 0 when clamped)
        const uint32_t last = (uint32_t)a.src_h * pitch - 3u;
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const uint32_t o00 = off[j];
            const uint32_t o01 = min(o00 + 3u, last), o10 = min(o00 + pitch, last), o11 = min(o00 + pitch + 3u, last);
            a0[j] = simg[o00] | (simg[o00 + 1] << 8) | (simg[o00 + 2] << 16) | (simg[o01] << 24);
            b0[j] = simg[o01 + 1] | (simg[o01 + 2] << 8);
            a1[j] = simg[o10] | (simg[o10 + 1] << 8) | (simg[o10 + 2] << 16) | (simg[o11] << 24);
            b1[j] = simg[o11 + 1] | (simg[o11 + 2] << 8);
        }
    }

    // ---- 4-weight blend on packed pairs (pixel j | pixel j+1) ------------------------------------
    float o[PX][3];
#pragma unroll
    for (int j = 0; j < PX; j += 2) {
        const f2 WX0 = {wx0[j], wx0[j + 1]}, WX1 = {wx1[j], wx1[j + 1]};
        const f2 WY0 = {wy0[j], wy0[j + 1]}, WY1 = {wy1[j], wy1[j + 1]};
        const f2 W00 = WX0 * WY0, W01 = WX1 * WY0, W10 = WX0 * WY1, W11 = WX1 * WY1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            f2 P00, P01, P10, P11;
            if (k == 0) {
                P00 = f2{ub(a0[j], 0), ub(a0[j + 1], 0)}; P01 = f2{ub(a0[j], 3), ub(a0[j + 1], 3)};
                P10 = f2{ub(a1[j], 0), ub(a1[j + 1], 0)}; P11 = f2{ub(a1[j], 3), ub(a1[j + 1], 3)};
            } else if (k == 1) {
                P00 = f2{ub(a0[j], 1), ub(a0[j + 1], 1)}; P01 = f2{ub(b0[j], 0), ub(b0[j + 1], 0)};
                P10 = f2{ub(a1[j], 1), ub(a1[j + 1], 1)}; P11 = f2{ub(b1[j], 0), ub(b1[j + 1], 0)};
            } else {
                P00 = f2{ub(a0[j], 2), ub(a0[j + 1], 2)}; P01 = f2{ub(b0[j], 1), ub(b0[j + 1], 1)};
                P10 = f2{ub(a1[j], 2), ub(a1[j + 1], 2)}; P11 = f2{ub(b1[j], 1), ub(b1[j + 1], 1)};
            }
            f2 acc = P00 * W00;
            acc = __builtin_elementwise_fma(P01, W01, acc);
            acc = __builtin_elementwise_fma(P10, W10, acc);
            acc = __builtin_elementwise_fma(P11, W11, acc);
            o[j][k] = acc.x; o[j + 1][k] = acc.y;
        }
    }

    // ---- store -------------------------------------------------------------------------------------
    if (shift == 0) {
        if constexpr (sizeof(DstT) == 1 && PKU8) {
            pk3 w;
            uint32_t q = 0;
            q = __builtin_amdgcn_cvt_pk_u8_f32(o[0][0], 0, q); q = __builtin_amdgcn_cvt_pk_u8_f32(o[0][1], 1, q);
            q = __builtin_amdgcn_cvt_pk_u8_f32(o[0][2], 2, q); q = __builtin_amdgcn_cvt_pk_u8_f32(o[1][0], 3, q);
            w.a = q; q = 0;
            q = __builtin_amdgcn_cvt_pk_u8_f32(o[1][1], 0, q); q = __builtin_amdgcn_cvt_pk_u8_f32(o[1][2], 1, q);
            q = __builtin_amdgcn_cvt_pk_u8_f32(o[2][0], 2, q); q = __builtin_amdgcn_cvt_pk_u8_f32(o[2][1], 3, q);
            w.b = q; q = 0;
            q = __builtin_amdgcn_cvt_pk_u8_f32(o[2][2], 0, q); q = __builtin_amdgcn_cvt_pk_u8_f32(o[3][0], 1, q);
            q = __builtin_amdgcn_cvt_pk_u8_f32(o[3][1], 2, q); q = __builtin_amdgcn_cvt_pk_u8_f32(o[3][2], 3, q);
            w.c = q;
            __builtin_memcpy(drow, &w, 12);
        } else {
            store4_rgb<DstT>(drow, o, PX);
        }
    } else if (c0 < a.out_w) {  // the one straddling lane of a ragged row: its pixels are local j >= shift
#pragma unroll
        for (int j = 1; j < PX; ++j)
            if (j >= shift) {
#pragma unroll
                for (int k = 0; k < 3; ++k) drow[3 * j + k] = to_dst<DstT>(o[j][k]);
            }
    }
}


// ================================================================================================
// Fast kernel v3 = v2's arithmetic + a 2-D patch per wave + LDS-staged source tiles.
// Measured on MI355X: v1/v2 are bound by the texture-address path, not by HBM or the VALU: eight
// 12-byte-stride gather loads per lane cost ~31 TA cycles each, and the stores queue behind them.
// v3 gives every wave a 64 x 4 output patch (lane = 4 consecutive pixels of one of 4 rows).  Its
// source footprint (bounding box of the four mapped corners: extremes of a projective map over a
// rectangle sit at its corners while W keeps its sign) is a few rows of <= 256 bytes, which the wave
// copies into a private LDS slab with coalesced 16-byte loads (16 lanes per source row, 4 rows per
// instruction); the bilinear taps are then 8-byte LDS reads.  No block barrier: the slab is
// wave-private.  Waves whose footprint touches the image border, is too large (strong zoom-out /
// rotation) or crosses the horizon fall back to masked gathers straight from global memory.
// ================================================================================================
constexpr int V3_ROWS = 9;           // source rows a wave can stage (3 per staging instruction)
constexpr int V3_LANES = 21;         // staging lanes per source row, 4 texels (12 B) each
constexpr int V3_TEXELS = 4 * V3_LANES;          // 84 texels per staged row
constexpr int V3_PITCH = 4 * V3_TEXELS + 16;     // LDS bytes per staged row (RGBX texels) + bank stagger

template <typename DstT, bool PKU8>
__global__ __launch_bounds__(256) void warp_rgb8_bilinear3(const WarpArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char slab[4][V3_ROWS * V3_PITCH];
    unsigned tx, ty, img;
    if (!decode_tile(a, tx, ty, img)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int prow = lane >> 4, pq = lane & 15;              // patch row 0..3, 4-pixel column group 0..15
    const int rr_raw = (int)ty * TILE_ROWS + prow;
    const int rr = min(rr_raw, a.rows - 1);                  // rows past the shard recompute its last row
    const int r = a.row_begin + rr;
    const int c0 = ((int)tx * 4 + wave) * 64 + pq * PX;
    const int c0p = min(c0, a.out_w - PX);                   // columns past the row end recompute its last 4 px
    const int shift = c0 - c0p;

    const unsigned char* simg = a.src + (long long)img * a.src_img_stride;
    DstT* drow = reinterpret_cast<DstT*>(a.dst + (long long)img * a.dst_img_stride) +
                 ((size_t)rr * (size_t)a.out_w + (size_t)c0p) * 3;

    const double y = fma((double)r, a.step_y, a.y0);
    const double xb = fma((double)c0p, a.step_x, a.x0);
    double X[PX], Y[PX], W[PX];
    X[0] = fma(a.ih[0], xb, fma(a.ih[1], y, a.ih[2]));
    Y[0] = fma(a.ih[3], xb, fma(a.ih[4], y, a.ih[5]));
    W[0] = fma(a.ih[6], xb, fma(a.ih[7], y, a.ih[8]));
#pragma unroll
    for (int j = 1; j < PX; ++j) {
        X[j] = X[0] + a.dxs[j - 1][0];
        Y[j] = Y[0] + a.dxs[j - 1][1];
        W[j] = W[0] + a.dxs[j - 1][2];
    }
    double rc[PX];
    {
        const double p01 = W[0] * W[1], p23 = W[2] * W[3], P = p01 * p23;
        if (__all(__builtin_amdgcn_class(P, 0x008 | 0x100))) {   // -normal | +normal
            double rp = __builtin_amdgcn_rcp(P);
            rp = fma(fma(-P, rp, 1.0), rp, rp);
            const double r01 = rp * p23, r23 = rp * p01;
            rc[0] = r01 * W[1]; rc[1] = r01 * W[0]; rc[2] = r23 * W[3]; rc[3] = r23 * W[2];
        } else {
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                double q = __builtin_amdgcn_rcp(W[j]);
                rc[j] = fma(fma(-W[j], q, 1.0), q, q);
            }
        }
    }
    uint32_t lx[PX], ly[PX], hx[PX], hy[PX];
    unsigned long long ubx[PX], uby[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const double ux = X[j] * rc[j] + MAGIC;
        const double uy = Y[j] * rc[j] + MAGIC;
        ubx[j] = (unsigned long long)__double_as_longlong(ux);
        uby[j] = (unsigned long long)__double_as_longlong(uy);
        lx[j] = lo32(ux); ly[j] = lo32(uy);
        hx[j] = hi32(ux); hy[j] = hi32(uy);
    }
    const uint32_t pitch = (uint32_t)a.src_w * 3u;

    // ---- wave-uniform footprint from the four patch corners (lanes 0, 15, 48, 63) ----------------
    int xmn, xmx, ymn, ymx;
    bool wpos;
    {
        const int x0 = (int)(__builtin_amdgcn_readlane(hx[0], 0) - MAGIC_HI), x1 = (int)(__builtin_amdgcn_readlane(hx[PX - 1], 15) - MAGIC_HI);
        const int x2 = (int)(__builtin_amdgcn_readlane(hx[0], 48) - MAGIC_HI), x3 = (int)(__builtin_amdgcn_readlane(hx[PX - 1], 63) - MAGIC_HI);
        const int y0 = (int)(__builtin_amdgcn_readlane(hy[0], 0) - MAGIC_HI), y1 = (int)(__builtin_amdgcn_readlane(hy[PX - 1], 15) - MAGIC_HI);
        const int y2 = (int)(__builtin_amdgcn_readlane(hy[0], 48) - MAGIC_HI), y3 = (int)(__builtin_amdgcn_readlane(hy[PX - 1], 63) - MAGIC_HI);
        xmn = min(min(x0, x1), min(x2, x3)); xmx = max(max(x0, x1), max(x2, x3));
        ymn = min(min(y0, y1), min(y2, y3)); ymx = max(max(y0, y1), max(y2, y3));
        const int w0 = __builtin_amdgcn_readlane((int)hi32(W[0]), 0), w1 = __builtin_amdgcn_readlane((int)hi32(W[PX - 1]), 15);
        const int w2 = __builtin_amdgcn_readlane((int)hi32(W[0]), 48), w3 = __builtin_amdgcn_readlane((int)hi32(W[PX - 1]), 63);
        wpos = (w0 > 0) & (w1 > 0) & (w2 > 0) & (w3 > 0);
    }
    // strictly inside: 0 <= floor(s) <= bound-2 on both axes, and the tap rows stay above the last source row
    const bool interior = wpos & (xmn >= 0) & (xmx < a.bound_w - 1) & (ymn >= 0) & (ymx < min(a.bound_h - 1, a.src_h - 2));
    const int nrows = ymx - ymn + 2;                         // rows ymn .. ymx+1
    const int ntex = xmx - xmn + 2;                          // texels xmn .. xmx+1
    const bool staged = interior & (nrows <= V3_ROWS) & (ntex <= V3_TEXELS);

    uint32_t a0[PX], b0[PX], a1[PX], b1[PX];   // per source row: a = texel ix (R,G,B in bytes 0..2), b = texel ix+1
    float wx0[PX], wx1[PX], wy0[PX], wy1[PX];
    if (staged) {
        unsigned char* my = slab[wave];
        // staging: lane -> (source row srow of 3, texel group scol of 21); 12 packed bytes in, 4 RGBX texels out
        const int srow = (lane * 49) >> 10;                  // lane / 21 for lane < 64
        const int scol = lane - V3_LANES * srow;
        const unsigned char* g = simg + (uint32_t)(ymn + srow) * pitch + (uint32_t)(xmn + 4 * scol) * 3u;
        unsigned char* w = my + srow * V3_PITCH + scol * 16;
        const bool mine = (srow < 3) & (4 * scol < ntex);
        // all staging loads are issued before the first LDS write so that their latencies overlap
        pk3 v[V3_ROWS / 3];
        bool on[V3_ROWS / 3];
#pragma unroll
        for (int k = 0; k < V3_ROWS / 3; ++k) {
            on[k] = mine & (3 * k + srow < nrows);
            v[k] = pk3{0u, 0u, 0u};
            if (on[k]) __builtin_memcpy(&v[k], g + (uint32_t)(3 * k) * pitch, 12);
        }
        const uint32_t hx_base = MAGIC_HI + (uint32_t)xmn, hy_base = MAGIC_HI + (uint32_t)ymn;
        uint32_t lo[PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            wx1[j] = (float)lx[j]; wx0[j] = (float)(~lx[j]);
            wy1[j] = (float)ly[j] * W_SCALE; wy0[j] = (float)(~ly[j]) * W_SCALE;
            lo[j] = (hy[j] - hy_base) * (uint32_t)V3_PITCH + (hx[j] - hx_base) * 4u;
        }
#pragma unroll
        for (int k = 0; k < V3_ROWS / 3; ++k) {
            if (on[k]) {
                uint4 t;
                t.x = v[k].a;
                t.y = __builtin_amdgcn_alignbyte(v[k].b, v[k].a, 3);
                t.z = __builtin_amdgcn_alignbyte(v[k].c, v[k].b, 2);
                t.w = v[k].c >> 8;
                *reinterpret_cast<uint4*>(w + 3 * k * V3_PITCH) = t;
            }
        }
        // the slab is wave-private: order this wave's LDS writes before its LDS reads, no block barrier
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            // two dwords at a 4-byte aligned address: ds_read2_b32 (a ds_read_b64 here would be "unaligned" and stall)
            const uint32_t* t0 = reinterpret_cast<const uint32_t*>(my + lo[j]);
            const uint32_t* t1 = reinterpret_cast<const uint32_t*>(my + lo[j] + V3_PITCH);
            a0[j] = t0[0]; b0[j] = t0[1]; a1[j] = t1[0]; b1[j] = t1[1];
        }
    } else {
        uint32_t off[PX];
        bool near_end = false;
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const bool valid = (ubx[j] >= MAGIC_BITS) & (ubx[j] <= a.xmax_bits) & (uby[j] >= MAGIC_BITS) & (uby[j] <= a.ymax_bits);
            const float sc = valid ? W_SCALE : 0.f;
            const int ix = (int)(hx[j] - MAGIC_HI), iy = (int)(hy[j] - MAGIC_HI);
            wx1[j] = (float)lx[j]; wx0[j] = (float)(~lx[j]);
            wy1[j] = (float)ly[j] * sc; wy0[j] = (float)(~ly[j]) * sc;
            off[j] = valid ? (uint32_t)iy * pitch + (uint32_t)ix * 3u : 0u;
            near_end |= valid & (iy > a.src_h - 3);
        }
        if (!__any(near_end)) {
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                const pk2 r0 = ld8(simg + off[j]);
                const pk2 r1 = ld8(simg + off[j] + pitch);
                a0[j] = r0.a; b0[j] = __builtin_amdgcn_alignbyte(r0.b, r0.a, 3);
                a1[j] = r1.a; b1[j] = __builtin_amdgcn_alignbyte(r1.b, r1.a, 3);
            }
        } else {  // byte-exact loads, +1 taps clamped to the image (their weight is 0 when clamped)
            const uint32_t last = (uint32_t)a.src_h * pitch - 3u;
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                const uint32_t o00 = off[j];
                const uint32_t o01 = min(o00 + 3u, last), o10 = min(o00 + pitch, last), o11 = min(o00 + pitch + 3u, last);
                a0[j] = simg[o00] | (simg[o00 + 1] << 8) | (simg[o00 + 2] << 16);
                b0[j] = simg[o01] | (simg[o01 + 1] << 8) | (simg[o01 + 2] << 16);
                a1[j] = simg[o10] | (simg[o10 + 1] << 8) | (simg[o10 + 2] << 16);
                b1[j] = simg[o11] | (simg[o11 + 1] << 8) | (simg[o11 + 2] << 16);
            }
        }
    }

    // ---- 4-weight blend on packed pairs (pixel j | pixel j+1) ------------------------------------
    // PKU8: v_cvt_pk_u8_f32 rounds to nearest, so the accumulator starts at -0.5 + 2^-15: exact integers
    // (weights 0/1) still land on themselves and anything else is floor(v + 3e-5), inside the blend's own
    // float32 noise; it saves the separate convert + shift/or packing.
    constexpr float BIAS = (sizeof(DstT) == 1 && PKU8) ? (-0.5f + 3.0517578125e-05f) : 0.f;
    float o[PX][3];
#pragma unroll
    for (int j = 0; j < PX; j += 2) {
        const f2 WX0 = {wx0[j], wx0[j + 1]}, WX1 = {wx1[j], wx1[j + 1]};
        const f2 WY0 = {wy0[j], wy0[j + 1]}, WY1 = {wy1[j], wy1[j + 1]};
        const f2 W00 = WX0 * WY0, W01 = WX1 * WY0, W10 = WX0 * WY1, W11 = WX1 * WY1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const f2 P00 = {ub(a0[j], k), ub(a0[j + 1], k)}, P01 = {ub(b0[j], k), ub(b0[j + 1], k)};
            const f2 P10 = {ub(a1[j], k), ub(a1[j + 1], k)}, P11 = {ub(b1[j], k), ub(b1[j + 1], k)};
            f2 acc = __builtin_elementwise_fma(P00, W00, f2{BIAS, BIAS});
            acc = __builtin_elementwise_fma(P01, W01, acc);
            acc = __builtin_elementwise_fma(P10, W10, acc);
            acc = __builtin_elementwise_fma(P11, W11, acc);
            o[j][k] = acc.x; o[j + 1][k] = acc.y;
        }
    }

    // ---- store -------------------------------------------------------------------------------------
    if (rr_raw < a.rows) {
        if (shift == 0) {
            if constexpr (sizeof(DstT) == 1 && PKU8) {
                pk3 w;
                uint32_t q = 0;
                q = __builtin_amdgcn_cvt_pk_u8_f32(o[0][0], 0, q); q = __builtin_amdgcn_cvt_pk_u8_f32(o[0][1], 1, q);
                q = __builtin_amdgcn_cvt_pk_u8_f32(o[0][2], 2, q); q = __builtin_amdgcn_cvt_pk_u8_f32(o[1][0], 3, q);
                w.a = q; q = 0;
                q = __builtin_amdgcn_cvt_pk_u8_f32(o[1][1], 0, q); q = __builtin_amdgcn_cvt_pk_u8_f32(o[1][2], 1, q);
                q = __builtin_amdgcn_cvt_pk_u8_f32(o[2][0], 2, q); q = __builtin_amdgcn_cvt_pk_u8_f32(o[2][1], 3, q);
                w.b = q; q = 0;
                q = __builtin_amdgcn_cvt_pk_u8_f32(o[2][2], 0, q); q = __builtin_amdgcn_cvt_pk_u8_f32(o[3][0], 1, q);
                q = __builtin_amdgcn_cvt_pk_u8_f32(o[3][1], 2, q); q = __builtin_amdgcn_cvt_pk_u8_f32(o[3][2], 3, q);
                w.c = q;
                __builtin_memcpy(drow, &w, 12);
            } else {
                store4_rgb<DstT>(drow, o, PX);
            }
        } else if (c0 < a.out_w) {  // the one straddling lane of a ragged row: its pixels are local j >= shift
#pragma unroll
            for (int j = 1; j < PX; ++j)
                if (j >= shift) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        if constexpr (sizeof(DstT) == 1 && PKU8) drow[3 * j + k] = (unsigned char)__builtin_amdgcn_cvt_pk_u8_f32(o[j][k], 0, 0);
                        else drow[3 * j + k] = to_dst<DstT>(o[j][k]);
                    }
                }
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------
template <typename K>
int launch(K kernel, const WarpArgs& a, hipStream_t s) {
    const unsigned grid = 8u * a.cpx;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, s, a);
    return check_launch();
}

template <typename SrcT, int C>
int dispatch(const WarpArgs& a, int interp, int dst_dtype, hipStream_t s) {
    if (interp == RWH_NEAREST) {
        if (dst_dtype != elem<SrcT>::dtype) return RWH_E_UNSUPPORTED;
        return launch(warp_generic<SrcT, C, SrcT, RWH_NEAREST>, a, s);
    }
    if (dst_dtype == RWH_F32) {
        if constexpr (sizeof(SrcT) == 1 && C == 3) return launch(warp_rgb8_bilinear<float>, a, s);
        else return launch(warp_generic<SrcT, C, float, RWH_BILINEAR>, a, s);
    }
    if (dst_dtype == RWH_U8) {
        if constexpr (sizeof(SrcT) == 1 && C == 3) return launch(warp_rgb8_bilinear<unsigned char>, a, s);
        else return launch(warp_generic<SrcT, C, unsigned char, RWH_BILINEAR>, a, s);
    }
    return RWH_E_UNSUPPORTED;
}

}  // namespace rwh

extern "C" int rwh_warp_backward(const void* d_src, int src_h, int src_w, int channels, int src_dtype,
                                 int64_t src_image_stride, int batch, const double* inv_h, int n_h,
                                 double x0, double step_x, double x_last, double y0, double step_y, double y_last,
                                 int out_h, int out_w, int bound_h, int bound_w, int interp,
                                 void* d_dst, int dst_dtype, int64_t dst_image_stride,
                                 int row_begin, int row_end, unsigned flags, void* stream) {
    using namespace rwh;
    if (batch <= 0 || out_h <= 0 || out_w <= 0) return RWH_E_INVALID;
    if (row_begin < 0 || row_end > out_h || row_begin > row_end) return RWH_E_INVALID;
    if (row_begin == row_end) return RWH_OK;  // empty row tile (a rank with no rows): nothing to do
    if (!d_src || !d_dst || !inv_h) return RWH_E_INVALID;
    if (src_h < 3 || src_w < 3) return RWH_E_UNSUPPORTED;
    if (bound_h <= 0 || bound_w <= 0) return RWH_E_INVALID;
    if (interp != RWH_NEAREST && interp != RWH_BILINEAR) return RWH_E_INVALID;
    if (n_h != 1) return RWH_E_UNSUPPORTED;
    if (channels != 3 && channels != 4) return RWH_E_UNSUPPORTED;
    if (src_dtype != RWH_U8 && src_dtype != RWH_F32) return RWH_E_UNSUPPORTED;
    const size_t esz = src_dtype == RWH_U8 ? 1 : 4;
    if ((size_t)src_h * (size_t)src_w * channels * esz >= (1ull << 32)) return RWH_E_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);

    if (flags & RWH_WARP_ZERO_ORIGIN) {
        for (int b = 0; b < batch; ++b)
            if (hipMemsetAsync(const_cast<unsigned char*>(static_cast<const unsigned char*>(d_src)) +
                                   (long long)b * src_image_stride, 0, channels * esz, s) != hipSuccess)
                return RWH_E_LAUNCH;
    }

    WarpArgs a;
    a.src = static_cast<const unsigned char*>(d_src);
    a.dst = static_cast<unsigned char*>(d_dst);
    a.src_img_stride = src_image_stride;
    a.dst_img_stride = dst_image_stride;
    for (int i = 0; i < 9; ++i) a.ih[i] = inv_h[i];
    a.x0 = x0; a.step_x = step_x; a.x_last = x_last;
    a.y0 = y0; a.step_y = step_y; a.y_last = y_last;
    a.src_h = src_h; a.src_w = src_w;
    a.bound_h = bound_h < src_h ? bound_h : src_h;
    a.bound_w = bound_w < src_w ? bound_w : src_w;
    a.out_h = out_h; a.out_w = out_w;
    a.row_begin = row_begin; a.rows = row_end - row_begin;
    a.tiles_x = (unsigned)((out_w + RWH_WAVE * PX - 1) / (RWH_WAVE * PX));
    a.tiles_y = (unsigned)((a.rows + TILE_ROWS - 1) / TILE_ROWS);
    const unsigned long long nb = (unsigned long long)a.tiles_x * a.tiles_y * (unsigned)batch;
    if (nb >= (1ull << 31) / 8) return RWH_E_UNSUPPORTED;
    a.nblocks = (unsigned)nb;
    a.cpx = (a.nblocks + 7u) / 8u;
    for (int j = 1; j <= 3; ++j) {
        a.dxs[j - 1][0] = (double)j * step_x * inv_h[0];
        a.dxs[j - 1][1] = (double)j * step_x * inv_h[3];
        a.dxs[j - 1][2] = (double)j * step_x * inv_h[6];
    }
    {
        const double xm = MAGIC + (double)(a.bound_w - 1), ym = MAGIC + (double)(a.bound_h - 1);
        __builtin_memcpy(&a.xmax_bits, &xm, 8);
        __builtin_memcpy(&a.ymax_bits, &ym, 8);
    }

    if (src_dtype == RWH_U8) return channels == 3 ? dispatch<unsigned char, 3>(a, interp, dst_dtype, s)
                                                  : dispatch<unsigned char, 4>(a, interp, dst_dtype, s);
    return channels == 3 ? dispatch<float, 3>(a, interp, dst_dtype, s) : dispatch<float, 4>(a, interp, dst_dtype, s);
}
