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

    if (src_dtype == RWH_U8) return channels == 3 ? dispatch<unsigned char, 3>(a, interp, dst_dtype, s)
                                                  : dispatch<unsigned char, 4>(a, interp, dst_dtype, s);
    return channels == 3 ? dispatch<float, 3>(a, interp, dst_dtype, s) : dispatch<float, 4>(a, interp, dst_dtype, s);
}
