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
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "rwh_warp_rgb8.h"

namespace rwh {

// rwh_warp_plan: the dispatch below runs as usual but, instead of launching, every launch site records the kernel it
// would have launched (the names are the demangled kernel names rocprofv3 prints).
static thread_local char* g_plan_buf = nullptr;
static thread_local int g_plan_len = 0;
static bool plan_only(const char* fmt, const char* a = "", int b = 0, const char* c = "", int d = 0) {
    if (!g_plan_buf) return false;
    if (g_plan_buf[0] == 0) snprintf(g_plan_buf, (size_t)g_plan_len, fmt, a, b, c, d);   // the first launch = the dominant kernel
    return true;
}
template <typename T> constexpr const char* tname() { return sizeof(T) == 1 ? "unsigned char" : sizeof(T) == 4 ? "float" : "double"; }

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

constexpr int PX = 4;        // pixels per lane (uint8 output: 12 / 16 bytes per lane and store)
// float32 output is 12 / 16 bytes per PIXEL already: one pixel per lane, so that a wave's taps go out together, a wave
// stores 64 consecutive pixels with one instruction, and four times as many waves are there to hide the gathers
template <typename DstT, int C> constexpr int generic_px() { return sizeof(DstT) == 1 ? PX : 1; }   // (uint8 RGBA at 1 px per lane: 23 % slower)
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
__device__ __forceinline__ void load_texel(const unsigned char* img, size_t img_bytes, int src_w, int iy, int ix, float (&t)[C]) {
    const size_t off = ((size_t)iy * (size_t)src_w + (size_t)ix) * (size_t)(C * sizeof(SrcT));
    if constexpr (sizeof(SrcT) == 1) {
        if constexpr (C == 4) {
            const uint32_t v = ld4(img + off);
#pragma unroll
            for (int c = 0; c < 4; ++c) t[c] = ub(v, c);
        } else {
            // RGB: one unaligned 4-byte load (3 bytes used) unless that would step past the last texel of the image
            if (off + 4 <= img_bytes) {
                const uint32_t v = ld4(img + off);
#pragma unroll
                for (int c = 0; c < C; ++c) t[c] = ub(v, c);
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) t[c] = (float)img[off + c];
            }
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
    constexpr int PX = generic_px<DstT, C>();
    const int c0 = ((int)tx * RWH_WAVE + lane) * PX;
    if (c0 >= a.out_w) return;

    const unsigned char* simg = a.src + (long long)img * a.src_img_stride;
    const size_t img_bytes = (size_t)a.src_h * (size_t)a.src_w * (size_t)(C * sizeof(SrcT));
    DstT* drow = reinterpret_cast<DstT*>(a.dst + (long long)img * a.dst_img_stride) +
                 ((size_t)rr * (size_t)a.out_w + (size_t)c0) * C;

    const double y = grid_coord(r, a.out_h, a.y0, a.step_y, a.y_last);
    const double rx = fma(a.ih[1], y, a.ih[2]);
    const double ry = fma(a.ih[4], y, a.ih[5]);
    const double rw = fma(a.ih[7], y, a.ih[8]);
    const double bw1 = (double)(a.bound_w - 1), bh1 = (double)(a.bound_h - 1);

    const bool whole = c0 + PX <= a.out_w;   // the lane's PX pixels are all inside the row: one vector store at the end
    DstT packed[PX * C];
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
                load_texel<SrcT, C>(simg, img_bytes, a.src_w, yi, xi, o);
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
                // (one 8-byte load per tap PAIR of a uint8 source was tried: 20 % slower -- the kernel is bound by its
                //  per-pixel float64 arithmetic, ~200 Gpix/s whatever the format, not by the gathers)
                load_texel<SrcT, C>(simg, img_bytes, a.src_w, iy, ix, p00);
                load_texel<SrcT, C>(simg, img_bytes, a.src_w, iy, ix1, p01);
                load_texel<SrcT, C>(simg, img_bytes, a.src_w, iy1, ix, p10);
                load_texel<SrcT, C>(simg, img_bytes, a.src_w, iy1, ix1, p11);
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
        if (whole) {
#pragma unroll
            for (int k = 0; k < C; ++k) packed[j * C + k] = to_dst<DstT>(o[k]);
        } else {
#pragma unroll
            for (int k = 0; k < C; ++k) drow[j * C + k] = to_dst<DstT>(o[k]);
        }
    }
    // PX*C elements = 12 / 16 bytes (u8) or 48 / 64 bytes (f32) per lane in one go instead of PX*C scalar stores
    // (a 64-lane byte store costs the texture-address unit as much as a 64-lane dword store)
    if (whole) __builtin_memcpy(drow, packed, sizeof(packed));
}

template <typename K> int launch(K kernel, const WarpArgs& a, hipStream_t s, const char* family, const char* src, int c, const char* dst, int interp);

// ================================================================================================
// Exact kernel (flag RWH_WARP_EXACT): the reference's float64 arithmetic, operation by operation, so that
// results are bit-identical to numpy's (fixtures made with numpy 2.2.6 / OpenBLAS 0.3.29):
//   * grid coordinate = arange*step + start with the endpoint forced to `stop` (numpy.linspace);
//   * inv(H) @ z in OpenBLAS dgemm k-order: acc = a0*x (rounded), acc = fma(a1, y, acc), acc = acc + a2
//     (checked against numpy on 23 M elements: 0 mismatches; any other order mismatches 10-30 %);
//   * z_t /= z_t[2]: IEEE float64 divides;
//   * bilinear (homography.py:131-137): float mask, astype(int32) truncation, f = z - trunc(z),
//     p00*(1-fx) + p01*fx, p10*(1-fx) + p11*fx, top*(1-fy) + bot*fy with every product and sum rounded
//     separately in float64; uint8 output = truncation of that float64 (transformImage's astype(uint8));
//   * nearest (homography.py:110-121): (z + 0.5) truncated to int32, mask on the integers.
// ~4x slower than the fast kernel: it exists for parity, not for throughput.
// ================================================================================================
template <typename SrcT, int C>
__device__ __forceinline__ void load_texel_f64(const unsigned char* img, size_t img_bytes, int src_w, int iy, int ix, double (&t)[C]) {
    const size_t off = ((size_t)iy * (size_t)src_w + (size_t)ix) * (size_t)(C * sizeof(SrcT));
    if constexpr (sizeof(SrcT) == 1) {
        if (off + 4 <= img_bytes) {     // one unaligned 4-byte load covers an RGB / RGBA texel
            const uint32_t v = ld4(img + off);
#pragma unroll
            for (int c = 0; c < C; ++c) t[c] = (double)((v >> (8 * c)) & 0xffu);
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) t[c] = (double)img[off + c];
        }
    } else {
        const float* p = reinterpret_cast<const float*>(img + off);
#pragma unroll
        for (int c = 0; c < C; ++c) t[c] = (double)p[c];
    }
}

// One sample at the float64 source coordinate (sx, sy), the reference's arithmetic operation by operation (see above).
template <typename SrcT, int C, typename DstT, int INTERP>
__device__ __forceinline__ void sample_exact(const unsigned char* simg, size_t img_bytes, int src_h, int src_w, int bound_h, int bound_w,
                                             double sx, double sy, DstT* out) {
    if constexpr (INTERP == RWH_NEAREST) {
        const int xi = (int)(sx + 0.5), yi = (int)(sy + 0.5);
        // numpy's astype(int32) turns NaN into INT_MIN, which the reference then masks (homography.py:117); the GPU's
        // conversion gives 0 for NaN, so a NaN coordinate is masked explicitly (+-inf saturate to a masked value either way)
        const bool valid = (xi >= 0) & (xi <= bound_w - 1) & (yi >= 0) & (yi <= bound_h - 1) & (sx == sx) & (sy == sy);
        if (valid) {
            const size_t off = ((size_t)yi * (size_t)src_w + (size_t)xi) * (size_t)(C * sizeof(SrcT));
            const SrcT* p = reinterpret_cast<const SrcT*>(simg + off);
#pragma unroll
            for (int k = 0; k < C; ++k) out[k] = (DstT)p[k];
        } else {
#pragma unroll
            for (int k = 0; k < C; ++k) out[k] = (DstT)0;
        }
    } else {
        double o[C];
        const double bw1 = (double)(bound_w - 1), bh1 = (double)(bound_h - 1);
        const bool valid = (sx >= 0.0) & (sx <= bw1) & (sy >= 0.0) & (sy <= bh1);
        if (valid) {
            const int ix = (int)sx, iy = (int)sy;
            const double fx = sx - (double)ix, fy = sy - (double)iy;
            const double gx = 1.0 - fx, gy = 1.0 - fy;
            const int ix1 = min(ix + 1, src_w - 1), iy1 = min(iy + 1, src_h - 1);
            double p00[C], p01[C], p10[C], p11[C];
            load_texel_f64<SrcT, C>(simg, img_bytes, src_w, iy, ix, p00);
            load_texel_f64<SrcT, C>(simg, img_bytes, src_w, iy, ix1, p01);
            load_texel_f64<SrcT, C>(simg, img_bytes, src_w, iy1, ix, p10);
            load_texel_f64<SrcT, C>(simg, img_bytes, src_w, iy1, ix1, p11);
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const double top = p00[k] * gx + p01[k] * fx;   // contraction is off: three roundings, like numpy
                const double bot = p10[k] * gx + p11[k] * fx;
                o[k] = top * gy + bot * fy;
            }
        } else {
#pragma unroll
            for (int k = 0; k < C; ++k) o[k] = 0.0;
        }
#pragma unroll
        for (int k = 0; k < C; ++k) {
            if constexpr (sizeof(DstT) == 1) out[k] = (unsigned char)(int)o[k];
            else out[k] = (DstT)o[k];
        }
    }
}

// convertfunc[...](z_t, img, h, w, mh, mw) on coordinates the caller computed (homography.py:108-138): one thread per point.
template <typename SrcT, int C, typename DstT, int INTERP>
__global__ __launch_bounds__(256) void sample_points_kernel(const unsigned char* img, int src_h, int src_w, int bound_h, int bound_w,
                                                            const double* __restrict__ xs, const double* __restrict__ ys, long long n,
                                                            DstT* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t img_bytes = (size_t)src_h * (size_t)src_w * (size_t)(C * sizeof(SrcT));
    sample_exact<SrcT, C, DstT, INTERP>(img, img_bytes, src_h, src_w, bound_h, bound_w, xs[i], ys[i], out + i * C);
}

template <typename SrcT, int C, typename DstT, int INTERP>
__global__ __launch_bounds__(256) void warp_exact(const WarpArgs a) {
    unsigned tx, ty, img;
    if (!decode_tile(a, tx, ty, img)) return;
    const int lane = threadIdx.x & 63, wrow = threadIdx.x >> 6;
    const int rr = (int)ty * TILE_ROWS + wrow;
    if (rr >= a.rows) return;
    const int r = a.row_begin + rr;
    const int c0 = ((int)tx * RWH_WAVE + lane) * PX;
    if (c0 >= a.out_w) return;
    const unsigned char* simg = a.src + (long long)img * a.src_img_stride;
    const size_t img_bytes = (size_t)a.src_h * (size_t)a.src_w * (size_t)(C * sizeof(SrcT));
    DstT* drow = reinterpret_cast<DstT*>(a.dst + (long long)img * a.dst_img_stride) +
                 ((size_t)rr * (size_t)a.out_w + (size_t)c0) * C;
    const double y = grid_coord(r, a.out_h, a.y0, a.step_y, a.y_last);
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int c = c0 + j;
        if (c >= a.out_w) break;
        const double x = grid_coord(c, a.out_w, a.x0, a.step_x, a.x_last);
        const double X = fma(a.ih[1], y, a.ih[0] * x) + a.ih[2];
        const double Y = fma(a.ih[4], y, a.ih[3] * x) + a.ih[5];
        const double W = fma(a.ih[7], y, a.ih[6] * x) + a.ih[8];
        const double sx = X / W, sy = Y / W;
        sample_exact<SrcT, C, DstT, INTERP>(simg, img_bytes, a.src_h, a.src_w, a.bound_h, a.bound_w, sx, sy, drow + j * C);
    }
}

template <typename SrcT, int C>
int dispatch_exact(const WarpArgs& a, int interp, int dst_dtype, hipStream_t s) {
    if (interp == RWH_NEAREST) {
        if (dst_dtype != elem<SrcT>::dtype) return RWH_E_UNSUPPORTED;
        return launch(warp_exact<SrcT, C, SrcT, RWH_NEAREST>, a, s, "warp_exact", tname<SrcT>(), C, tname<SrcT>(), RWH_NEAREST);
    }
    if (dst_dtype == RWH_F64) return launch(warp_exact<SrcT, C, double, RWH_BILINEAR>, a, s, "warp_exact", tname<SrcT>(), C, "double", RWH_BILINEAR);
    if (dst_dtype == RWH_U8) return launch(warp_exact<SrcT, C, unsigned char, RWH_BILINEAR>, a, s, "warp_exact", tname<SrcT>(), C, "unsigned char", RWH_BILINEAR);
    return RWH_E_UNSUPPORTED;
}

// ---- host side ---------------------------------------------------------------------------------
template <typename K>
int launch(K kernel, const WarpArgs& a, hipStream_t s, const char* family, const char* src, int c, const char* dst, int interp) {
    char fmt[96];
    snprintf(fmt, sizeof fmt, "rwh::%s<%%s, %%d, %%s, %%d>", family);
    if (plan_only(fmt, src, c, dst, interp)) return RWH_OK;
    const unsigned grid = 8u * a.cpx;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, s, a);
    return check_launch();
}

// the block grid for `px` pixels per lane (the entry point sets it up for PX); false: too many blocks for one launch
static bool retile(WarpArgs& a, int px) {
    const unsigned long long per_img = (unsigned long long)a.tiles_x * a.tiles_y;
    const unsigned long long batch = per_img ? a.nblocks / per_img : 0;
    a.tiles_x = (unsigned)((a.out_w + RWH_WAVE * px - 1) / (RWH_WAVE * px));
    const unsigned long long nb = (unsigned long long)a.tiles_x * a.tiles_y * batch;
    if (nb >= (1ull << 31) / 8) return false;
    a.nblocks = (unsigned)nb;
    a.cpx = (a.nblocks + 7u) / 8u;
    return true;
}

template <typename SrcT, int C>
int dispatch(const WarpArgs& a, int interp, int dst_dtype, hipStream_t s) {
    if (interp == RWH_NEAREST) {
        if (dst_dtype != elem<SrcT>::dtype) return RWH_E_UNSUPPORTED;
        if constexpr (generic_px<SrcT, C>() != PX) {
            WarpArgs b = a;
            if (!retile(b, generic_px<SrcT, C>())) return RWH_E_UNSUPPORTED;
            return launch(warp_generic<SrcT, C, SrcT, RWH_NEAREST>, b, s, "warp_generic", tname<SrcT>(), C, tname<SrcT>(), RWH_NEAREST);
        }
        return launch(warp_generic<SrcT, C, SrcT, RWH_NEAREST>, a, s, "warp_generic", tname<SrcT>(), C, tname<SrcT>(), RWH_NEAREST);
    }
    if (dst_dtype == RWH_F32) {
        WarpArgs b = a;
        if (!retile(b, generic_px<float, C>())) return RWH_E_UNSUPPORTED;
        return launch(warp_generic<SrcT, C, float, RWH_BILINEAR>, b, s, "warp_generic", tname<SrcT>(), C, "float", RWH_BILINEAR);
    }
    if (dst_dtype == RWH_U8) {
        WarpArgs b = a;
        if (!retile(b, generic_px<unsigned char, C>())) return RWH_E_UNSUPPORTED;
        return launch(warp_generic<SrcT, C, unsigned char, RWH_BILINEAR>, b, s, "warp_generic", tname<SrcT>(), C, "unsigned char", RWH_BILINEAR);
    }
    return RWH_E_UNSUPPORTED;
}

__global__ void zero_origin_kernel(unsigned char* src, long long stride, int batch, int nbytes) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch)
        for (int i = 0; i < nbytes; ++i) src[(long long)b * stride + i] = 0;
}

// Patch shape of the 8 px kernel (log2 of the patch width: 7, 6 or 5).  Over a 5 x 5 sample of patch positions, a shape
// qualifies if its source footprints fit the LDS slab (nearly) everywhere.  64 x 8 is the default (measured 2-3 % ahead
// of 128 x 4 on axis-aligned warps: fewer staged chunks per pixel); another qualifying shape replaces it only if its
// staging loads touch clearly fewer 128-byte lines (a 64 x 8 patch rotated by 90 degrees "fits", but as 66 rows of 10
// texels).  The choice is a function of the homography and the WHOLE output grid only -- never of the row shard or the
// batch -- so that shards, batches and single launches of the same warp run the same arithmetic and agree bit for bit.
// rwh_lab_tune(RWH_TUNE_WARP_SHAPE, 5|6|7) overrides (tests, lab).
// (r0, c0, nr, nc): the part of the output grid the samples are taken from (the compositor's canvas extends past the warped
// image's rectangle, where the map means nothing); default: the whole grid.
// allow_halves (uint8 bilinear RGB, one homography): when no whole patch fits -- minification beyond ~1.28x -- the HALVES
// form of the kernel is tried, 64 x 8 first (halves of 32 x 8: up to ~1.6x), then 32 x 16 (halves of 16 x 16: windows up
// to ~2.2x); the return value is then the shape + 8.  Staging pays while a half's window holds at most HALVES_MAX_TEXELS
// texels per output pixel: the gathers cost the same per output pixel whatever the minification s, the staged texels grow
// as s^2 -- measured on 4K frames (profiles/r03_lab_notes.txt section 11): 1.4x 0.45 of the roofline staged vs 0.37
// gathered, 1.5x 0.45 vs 0.38, 1.7x 0.44 vs 0.44, 2x 0.46 vs 0.59.
constexpr double HALVES_MAX_TEXELS = 4.0;
static int choose_shape(const FastArgs& a, int r0 = 0, int c0 = 0, int nr = -1, int nc = -1, bool allow_halves = false) {
    if (g_force_warp_shape) return (g_force_warp_shape & 8) && !allow_halves ? 7 : g_force_warp_shape;
    if (nr < 0) { nr = a.out_h; nc = a.out_w; }
    int best = 0;
    double best_lines = 1e300;
    const int order[3] = {6, 7, 5};
    for (int lp : order) {
        const int pw = 1 << lp, ph = 512 >> lp;
        int seen = 0, fit = 0;
        double lines_sum = 0;
        for (int i = 0; i < 5; ++i)
            for (int j = 0; j < 5; ++j) {
                const double r = r0 + (nr > ph ? (nr - ph) * (i / 4.0) : 0.0), c = c0 + (nc > pw ? (nc - pw) * (j / 4.0) : 0.0);
                long long fr, ft; double ln;
                if (!patch_footprint(a, __builtin_floor(r), __builtin_floor(c), pw, ph, &fr, &ft, &ln)) continue;   // horizon: gathers anyway
                ++seen; fit += f8_window_fits(lp, fr, ft); lines_sum += ln;
            }
        if (seen == 0) return 6;
        if (10 * fit < 9 * seen) continue;
        if (!best || lines_sum < 0.85 * best_lines) { best = lp; best_lines = lines_sum; }
    }
    if (best) return best;
    if (allow_halves) {
        const int horder[2] = {6, 5};
        for (int lp : horder) {
            const int pw = 1 << lp, ph = 512 >> lp;
            int seen = 0, fit = 0;
            double staged = 0;
            for (int i = 0; i < 5; ++i)
                for (int j = 0; j < 5; ++j)
                    for (int h = 0; h < 2; ++h) {
                        const double r = r0 + (nr > ph ? (nr - ph) * (i / 4.0) : 0.0), c = c0 + (nc > pw ? (nc - pw) * (j / 4.0) : 0.0);
                        long long fr, ft; double ln;
                        if (!patch_footprint(a, __builtin_floor(r), __builtin_floor(c) + h * (pw / 2), pw / 2, ph, &fr, &ft, &ln)) continue;
                        ++seen; fit += f8_window_fits(lp, fr, ft);          // (ft counts from the window's start, a multiple of 4 texels)
                        staged += (double)fr * (double)(((ft + 3) >> 2) << 2);
                    }
            if (seen && 10 * fit >= 9 * seen && staged <= HALVES_MAX_TEXELS * 256.0 * seen) return lp + 8;
        }
    }
    return 7;   // nothing fits (strong zoom-out): every wave gathers; 128 x 4 has the longest stores
}

// Coefficients of one homography on one output grid: X = cx[0] + row*cx[1] + col*cx[2] etc.
static void fill_coef(Coef& c, const double* ih, double x0, double step_x, double y0, double step_y) {
    c.cx[0] = ih[0] * x0 + ih[1] * y0 + ih[2]; c.cx[1] = ih[1] * step_y; c.cx[2] = ih[0] * step_x;
    c.cy[0] = ih[3] * x0 + ih[4] * y0 + ih[5]; c.cy[1] = ih[4] * step_y; c.cy[2] = ih[3] * step_x;
    c.cw[0] = ih[6] * x0 + ih[7] * y0 + ih[8]; c.cw[1] = ih[7] * step_y; c.cw[2] = ih[6] * step_x;
    for (int i = 0; i < 9; ++i) c.ih[i] = ih[i];
    c.image = 0;
}

// Column offsets of a lane's pixels in the 8 px kernel: two runs PW/2 apart, pixels of a run 1 column apart (uint8
// output) or PW/8 columns apart (float32 output, rwh_warp_rgb8.h).
static void fill_offsets(Coef& c, int shape, int pstr) {
    for (int j = 1; j <= 7; ++j) {
        const double o8 = (double)(j < 4 ? j * pstr : (1 << shape) / 2 + (j - 4) * pstr);
        c.dxs8[j - 1][0] = o8 * c.cx[2]; c.dxs8[j - 1][1] = o8 * c.cy[2]; c.dxs8[j - 1][2] = o8 * c.cw[2];
    }
}

// Fast-path launch (RGB u8; bilinear u8 / float32 output, or nearest): returns RWH_E_UNSUPPORTED when the shape needs
// the generic kernel.
// variant: 0 = 4 px per lane (256 x 4 block tiles), 1 = 8 px per lane (128 x 16 block tiles, patch shape chosen here),
// 3 = nearest neighbour on the 8 px kernel's tiling.  n_h = 1: one homography for the batch; n_h = batch (variants 1
// and 3): one per image, launched TAB_N images at a time with their coefficients as a second kernel argument.
// (variant 2 / `group` / `custom` serve tools/warp_lab.hip: an experimental kernel on the 8 px kernel's 128 x 16 block tiles.)
int launch_fast(const WarpArgs& w, const double* ih, double x0, double step_x, double y0, double step_y,
                int dst_dtype, int batch, hipStream_t s, int variant, int group = 1,
                void (*custom)(const FastArgs) = nullptr, int n_h = 1, const CompArgs* comp = nullptr, int channels = 3) {
    const bool px8 = variant >= 1, nn = variant == 3;
    if (w.out_w < (px8 ? 128 : FP_PX) || w.bound_w > (1 << 19) || w.bound_h > (1 << 19)) return RWH_E_UNSUPPORTED;
    if (n_h != 1 && (!px8 || custom)) return RWH_E_UNSUPPORTED;
    // 4 channels: the uint8 RGBA form of the 8 px bilinear kernel only (one homography, no compositor)
    if (channels == 4 && (!px8 || nn || dst_dtype != RWH_U8 || n_h != 1 || comp || custom)) return RWH_E_UNSUPPORTED;
    const size_t dst_esz = dst_dtype == RWH_U8 ? 1 : 4;
    if ((size_t)w.rows * (size_t)w.out_w * (size_t)channels * dst_esz >= (1ull << 32)) return RWH_E_UNSUPPORTED;  // 32-bit lane offsets
    FastArgs a;
    a.src = w.src; a.dst = w.dst; a.src_img_stride = w.src_img_stride; a.dst_img_stride = w.dst_img_stride;
    a.src_h = w.src_h; a.src_w = w.src_w; a.bound_h = w.bound_h; a.bound_w = w.bound_w; a.out_w = w.out_w; a.pitch_w = w.out_w;
    // a thin ragged right edge (1 .. STRIP_MAX columns past a multiple of 128) is left to warp_rgb8_strip (rwh_warp_rgb8.h): the
    // bilinear 8 px kernels then cover whole tiles only.  A function of out_w alone: shards, batches, tables and shapes agree.
    const int strip = (px8 && !nn && !custom && !comp && channels == 3 && w.out_w >= 256 && w.out_w % 128 >= 1 && w.out_w % 128 <= STRIP_MAX)
                          ? w.out_w % 128 : 0;
    a.out_w = w.out_w - strip;
    auto launch_strip = [&](const CoefTab* tab, int count) -> int {
        if (!strip || g_plan_buf) return RWH_OK;
        const dim3 sgrid((unsigned)((w.rows + 255) / 256), (unsigned)count);
        if (tab) {
            if (dst_dtype == RWH_U8) hipLaunchKernelGGL(warp_rgb8_strip_tab<unsigned char>, sgrid, dim3(256), 0, s, a, *tab);
            else hipLaunchKernelGGL(warp_rgb8_strip_tab<float>, sgrid, dim3(256), 0, s, a, *tab);
        } else {
            if (dst_dtype == RWH_U8) hipLaunchKernelGGL(warp_rgb8_strip<unsigned char>, sgrid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL(warp_rgb8_strip<float>, sgrid, dim3(256), 0, s, a);
        }
        return check_launch();
    };
    a.row_begin = w.row_begin; a.rows = w.rows;
    a.gx0 = w.x0; a.gstep_x = w.step_x; a.gx_last = w.x_last; a.gy0 = w.y0; a.gstep_y = w.step_y; a.gy_last = w.y_last;
    a.out_h = w.out_h;
    // patch shape: the host's choice per homography; with one homography per image the images are grouped by shape
    int shape = 0, shapes[16] = {};      // indexed by the shape code: 5..7, + 8 for the HALVES form
    std::vector<int> shape_of(n_h > 1 ? n_h : 0);
    if (px8) {
        for (int i = 0; i < n_h; ++i) {
            fill_coef(a.c, ih + 9 * i, x0, step_x, y0, step_y);
            // uint8 bilinear RGB only.  (RGBA was tried: its gathers are aligned 8-byte loads of exactly the two texels a tap
            //  pair needs, and beat the halves at every minification -- 4K x 16, 1.4x: 0.57 of the roofline gathered vs 0.48 staged)
            const bool halves_ok = !comp && !nn && !custom && channels == 3 && dst_dtype == RWH_U8;
            shape = comp ? choose_shape(a, comp->tsy, comp->tsx, comp->ht, comp->wt) : choose_shape(a, 0, 0, -1, -1, halves_ok);
            if (n_h > 1) { shape_of[i] = shape; shapes[shape] = 1; }
        }
    }
    fill_coef(a.c, ih, x0, step_x, y0, step_y);
    const bool halves = px8 && (shape & 8);
    shape &= 7;
    const int pstr = (px8 && dst_dtype != RWH_U8) ? (1 << shape) / 8 : 1;
    if (px8) fill_offsets(a.c, shape, pstr);
    for (int j = 1; j <= 3; ++j) { a.dxs[j - 1][0] = j * a.c.cx[2]; a.dxs[j - 1][1] = j * a.c.cy[2]; a.dxs[j - 1][2] = j * a.c.cw[2]; }
    const double xm = MAGIC + (double)(w.bound_w - 1), ym = MAGIC + (double)(w.bound_h - 1);
    __builtin_memcpy(&a.xmax_bits, &xm, 8);
    __builtin_memcpy(&a.ymax_bits, &ym, 8);
    a.tiles_x = (unsigned)(px8 ? (a.out_w + 127) / 128 : (a.out_w + 255) / 256);
    a.group = group;   // free parameter of a tools/warp_lab custom kernel
    a.tiles_y = (unsigned)(px8 ? (w.rows + 15) / 16 : (w.rows + 3) / 4);
    const bool u8 = dst_dtype == RWH_U8;
    const dim3 block(256);
    auto geometry = [&](int count) {
        const unsigned long long nb = (unsigned long long)a.tiles_x * a.tiles_y * (unsigned)count;
        if (nb >= (1ull << 31) / 8) return false;
        a.nblocks = (unsigned)nb;
        a.cpx = (a.nblocks + 7u) / 8u;
#ifdef RWH_XCD_CHUNK_LOG   // lab builds (rwh_warp_rgb8.h): the grid is a whole number of 8-chunk groups
        a.cpx = (a.cpx + (1u << RWH_XCD_CHUNK_LOG) - 1u) >> RWH_XCD_CHUNK_LOG << RWH_XCD_CHUNK_LOG;
#endif
        a.tiles_x_magic = div_magic(a.tiles_x, nb);
        a.tiles_y_magic = div_magic(a.tiles_y, nb / a.tiles_x + 1);
        return !((a.tiles_x > 1 && !a.tiles_x_magic) || (a.tiles_y > 1 && !a.tiles_y_magic));
    };
    if (n_h == 1) {
        if (!geometry(batch)) return RWH_E_UNSUPPORTED;
        void (*kern)(const FastArgs) = custom;
        if (!kern) {
            if (nn) kern = shape == 7 ? warp_rgb8_nn<7> : shape == 6 ? warp_rgb8_nn<6> : warp_rgb8_nn<5>;
            else if (!px8) kern = u8 ? warp_rgb8_fast<unsigned char> : warp_rgb8_fast<float>;
            else if (halves) kern = shape == 6 ? warp_rgb8_fast8h<6> : warp_rgb8_fast8h<5>;
            else if (shape == 7) kern = u8 ? warp_rgb8_fast8<unsigned char, 7> : warp_rgb8_fast8<float, 7>;
            else if (shape == 6) kern = u8 ? warp_rgb8_fast8<unsigned char, 6> : warp_rgb8_fast8<float, 6>;
            else kern = u8 ? warp_rgb8_fast8<unsigned char, 5> : warp_rgb8_fast8<float, 5>;
        }
        if (channels == 4) {
            if (plan_only("rwh::warp_rgba8_fast8<%.0s%d>", "", shape)) return RWH_OK;
            kern = shape == 7 ? warp_rgba8_fast8<7> : shape == 6 ? warp_rgba8_fast8<6> : warp_rgba8_fast8<5>;
            hipLaunchKernelGGL(kern, dim3(8u * a.cpx), block, 0, s, a);
            return check_launch();
        }
        if (comp) {      // canvas compositor: the 8 px kernel with the paste / 'Rate' epilogue (uint8, one image)
            if (!px8 || nn || !u8 || batch != 1) return RWH_E_UNSUPPORTED;
            if (plan_only("rwh::warp_rgb8_comp<%.0s%d>", "", shape)) return RWH_OK;
            void (*ck)(const FastArgs, const CompArgs) = shape == 7 ? warp_rgb8_comp<7> : shape == 6 ? warp_rgb8_comp<6> : warp_rgb8_comp<5>;
            hipLaunchKernelGGL(ck, dim3(8u * a.cpx), block, 0, s, a, *comp);
            return check_launch();
        }
        // one homography, several frames, interior geometry shared by the frames of a block (warp_rgb8_fast8m): uint8 RGB bilinear
        // whole-patch kernel only.  OFF unless rwh_lab_tune(RWH_TUNE_WARP_FRAMES, n >= 2) asks for n frames per block: the kernel
        // halves the VALU work per frame (366 -> ~185 instructions per wave and frame) and is bit-identical, but same-box A/B
        // (profiles/r04_lab_notes.txt) gives +3 % on 4K x 32, +2 % on 1080p x 512 and -3 % on 8K x 8 at its best setting (3-4
        // frames): this warp is bound by the memory system's throughput on its access pattern, not by its arithmetic.
        if (px8 && !nn && !custom && !halves && u8 && batch >= 2 && g_force_warp_frames >= 2) {
            const int want = g_force_warp_frames % 100;
            const bool block_window = g_force_warp_frames >= 100;      // 100 + n: one staging window per block (warp_rgb8_fast8mb)
            const int F = want < batch ? want : batch;
            const unsigned long long ntiles = (unsigned long long)a.tiles_x * a.tiles_y, groups = ((unsigned)batch + F - 1) / F;
            const unsigned long long mnb = ntiles * groups;
            const unsigned magic = div_magic((unsigned)ntiles, mnb);
            if (mnb < (1ull << 31) / 8 && (ntiles == 1 || magic)) {
                a.mf_frames = F; a.mf_batch = batch; a.ntiles = (unsigned)ntiles; a.ntiles_magic = magic;
                a.mf_nblocks = (unsigned)mnb; a.mf_cpx = (a.mf_nblocks + 7u) / 8u;
                if (plan_only(block_window ? "rwh::warp_rgb8_fast8mb<%.0s%d>" : "rwh::warp_rgb8_fast8m<%.0s%d>", "", shape)) return RWH_OK;
                void (*mk)(const FastArgs) = block_window ? (shape == 7 ? warp_rgb8_fast8mb<7> : shape == 6 ? warp_rgb8_fast8mb<6> : warp_rgb8_fast8mb<5>)
                                                          : (shape == 7 ? warp_rgb8_fast8m<7> : shape == 6 ? warp_rgb8_fast8m<6> : warp_rgb8_fast8m<5>);
                hipLaunchKernelGGL(mk, dim3(8u * a.mf_cpx), block, 0, s, a);
                if (check_launch() != RWH_OK) return RWH_E_LAUNCH;
                return launch_strip(nullptr, batch);
            }
        }
        if (custom ? false : nn ? plan_only("rwh::warp_rgb8_nn<%.0s%d>", "", shape)
                          : !px8 ? plan_only("rwh::warp_rgb8_fast<%s>", u8 ? "unsigned char" : "float")
                          : halves ? plan_only("rwh::warp_rgb8_fast8h<%.0s%d>", "", shape)
                                 : plan_only("rwh::warp_rgb8_fast8<%s, %d>", u8 ? "unsigned char" : "float", shape))
            return RWH_OK;
        hipLaunchKernelGGL(kern, dim3(8u * a.cpx), block, 0, s, a);
        if (check_launch() != RWH_OK) return RWH_E_LAUNCH;
        return launch_strip(nullptr, batch);
    }
    // one homography per image: per shape, TAB_N images per launch with their coefficients as a second kernel argument
    for (int code = 15; code >= 5; --code) {
        if (!shapes[code]) continue;
        const int sh = code & 7;
        const bool hv = code & 8;            // images whose homography minifies: patches staged by halves
        const int ps = u8 ? 1 : (1 << sh) / 8;
        void (*kern)(const FastArgs, const CoefTab);
        if (hv) kern = sh == 6 ? warp_rgb8_fast8h_tab<6> : warp_rgb8_fast8h_tab<5>;
        else if (nn) kern = sh == 7 ? warp_rgb8_nn_tab<7> : sh == 6 ? warp_rgb8_nn_tab<6> : warp_rgb8_nn_tab<5>;
        else if (sh == 7) kern = u8 ? warp_rgb8_fast8_tab<unsigned char, 7> : warp_rgb8_fast8_tab<float, 7>;
        else if (sh == 6) kern = u8 ? warp_rgb8_fast8_tab<unsigned char, 6> : warp_rgb8_fast8_tab<float, 6>;
        else kern = u8 ? warp_rgb8_fast8_tab<unsigned char, 5> : warp_rgb8_fast8_tab<float, 5>;
        CoefTab tab;
        int count = 0;
        for (int i = 0; i <= n_h; ++i) {
            if (i < n_h && shape_of[i] == code) {
                fill_coef(tab.e[count], ih + 9 * i, x0, step_x, y0, step_y);
                fill_offsets(tab.e[count], sh, ps);
                tab.e[count++].image = i;
            }
            if (count == TAB_N || (i == n_h && count > 0)) {
                for (int k = count; k < TAB_N; ++k) tab.e[k] = tab.e[0];
                if (!geometry(count)) return RWH_E_UNSUPPORTED;
                if (hv ? plan_only("rwh::warp_rgb8_fast8h_tab<%.0s%d>", "", sh) : nn ? plan_only("rwh::warp_rgb8_nn_tab<%.0s%d>", "", sh) : plan_only("rwh::warp_rgb8_fast8_tab<%s, %d>", u8 ? "unsigned char" : "float", sh)) { count = 0; continue; }
                hipLaunchKernelGGL(kern, dim3(8u * a.cpx), block, 0, s, a, tab);
                if (check_launch() != RWH_OK) return RWH_E_LAUNCH;
                if (!nn && launch_strip(&tab, count) != RWH_OK) return RWH_E_LAUNCH;
                count = 0;
            }
        }
    }
    return RWH_OK;
}

// Fast form of rwh_stitch_panorama (rwh_stitch.hip): imgT warped onto the canvas grid by the staged 8 px kernel, imgQ
// composited in its epilogue.  (x0, y0) = the warp-grid coordinate of canvas pixel (0, 0).
int warp_composite(const unsigned char* d_img_t, int t_h, int t_w, const double* inv_h, double x0, double y0, int canvas_h,
                   int canvas_w, unsigned char* d_canvas, const CompArgs& comp, hipStream_t s) {
    if (canvas_w < 128 || (size_t)t_h * t_w * 3 >= (1ull << 32)) return RWH_E_UNSUPPORTED;
    WarpArgs a;
    a.src = d_img_t; a.dst = d_canvas;
    a.src_img_stride = 0; a.dst_img_stride = 0;
    for (int i = 0; i < 9; ++i) a.ih[i] = inv_h[i];
    a.x0 = x0; a.step_x = 1.0; a.x_last = x0 + (double)(canvas_w - 1);
    a.y0 = y0; a.step_y = 1.0; a.y_last = y0 + (double)(canvas_h - 1);
    a.src_h = t_h; a.src_w = t_w; a.bound_h = t_h; a.bound_w = t_w;
    a.out_h = canvas_h; a.out_w = canvas_w; a.row_begin = 0; a.rows = canvas_h;
    return launch_fast(a, inv_h, x0, 1.0, y0, 1.0, RWH_U8, 1, s, /*variant=*/1, 1, nullptr, 1, &comp);
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
    if (n_h != 1 && n_h != batch) return RWH_E_INVALID;
    if (channels != 3 && channels != 4) return RWH_E_UNSUPPORTED;
    if (src_dtype != RWH_U8 && src_dtype != RWH_F32) return RWH_E_UNSUPPORTED;
    const size_t esz = src_dtype == RWH_U8 ? 1 : 4;
    if ((size_t)src_h * (size_t)src_w * channels * esz >= (1ull << 32)) return RWH_E_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);

    if ((flags & RWH_WARP_ZERO_ORIGIN) && !g_plan_buf) {  // one tiny launch for the whole batch (memsets cost ~4 us each)
        hipLaunchKernelGGL(zero_origin_kernel, dim3((batch + 255) / 256), dim3(256), 0, s,
                           const_cast<unsigned char*>(static_cast<const unsigned char*>(d_src)), (long long)src_image_stride,
                           batch, (int)(channels * esz));
        if (check_launch() != RWH_OK) return RWH_E_LAUNCH;
    }

    if (n_h == batch && batch > 1) {
        // one homography per image.  The staged RGB u8 kernels take a table of TAB_N coefficient sets per launch ...
        const bool fast_bil = src_dtype == RWH_U8 && channels == 3 && interp == RWH_BILINEAR && !(flags & RWH_WARP_EXACT) &&
                              (dst_dtype == RWH_U8 || dst_dtype == RWH_F32);
        const bool fast_nn = src_dtype == RWH_U8 && channels == 3 && interp == RWH_NEAREST && dst_dtype == RWH_U8 &&
                             (size_t)src_h * src_w * 3 < (1ull << 32) - 4;
        if ((fast_bil || fast_nn) && out_w >= 128) {
            WarpArgs a;
            a.src = static_cast<const unsigned char*>(d_src);
            a.dst = static_cast<unsigned char*>(d_dst);
            a.src_img_stride = src_image_stride; a.dst_img_stride = dst_image_stride;
            a.x0 = x0; a.step_x = step_x; a.x_last = x_last; a.y0 = y0; a.step_y = step_y; a.y_last = y_last;
            a.src_h = src_h; a.src_w = src_w;
            a.bound_h = bound_h < src_h ? bound_h : src_h;
            a.bound_w = bound_w < src_w ? bound_w : src_w;
            a.out_h = out_h; a.out_w = out_w; a.row_begin = row_begin; a.rows = row_end - row_begin;
            const int st = launch_fast(a, inv_h, x0, step_x, y0, step_y, dst_dtype, batch, s, fast_nn ? 3 : 1, 1, nullptr, batch);
            if (st != RWH_E_UNSUPPORTED) return st;
        }
        // ... every other configuration is one launch per image on the same stream
        for (int i = 0; i < batch; ++i) {
            const int st = rwh_warp_backward(static_cast<const unsigned char*>(d_src) + (int64_t)i * src_image_stride, src_h, src_w,
                                             channels, src_dtype, src_image_stride, 1, inv_h + 9 * i, 1, x0, step_x, x_last, y0,
                                             step_y, y_last, out_h, out_w, bound_h, bound_w, interp,
                                             static_cast<unsigned char*>(d_dst) + (int64_t)i * dst_image_stride, dst_dtype,
                                             dst_image_stride, row_begin, row_end, flags & ~RWH_WARP_ZERO_ORIGIN, stream);
            if (st != RWH_OK) return st;
        }
        return RWH_OK;
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

    // nearest neighbour, RGB u8: one kernel for the exact and the default mode -- it is bit-exact by construction
    if (src_dtype == RWH_U8 && channels == 3 && interp == RWH_NEAREST && dst_dtype == RWH_U8 && a.out_w >= 128 &&
        (size_t)src_h * src_w * 3 < (1ull << 32) - 4) {
        const int st = launch_fast(a, inv_h, x0, step_x, y0, step_y, dst_dtype, batch, s, /*variant=*/3);
        if (st != RWH_E_UNSUPPORTED) return st;
    }
    if (flags & RWH_WARP_EXACT) {
        if (src_dtype == RWH_U8) return channels == 3 ? dispatch_exact<unsigned char, 3>(a, interp, dst_dtype, s)
                                                      : dispatch_exact<unsigned char, 4>(a, interp, dst_dtype, s);
        return channels == 3 ? dispatch_exact<float, 3>(a, interp, dst_dtype, s) : dispatch_exact<float, 4>(a, interp, dst_dtype, s);
    }
    if (src_dtype == RWH_U8 && channels == 3 && interp == RWH_BILINEAR && (dst_dtype == RWH_U8 || dst_dtype == RWH_F32)) {
        const int st = launch_fast(a, inv_h, x0, step_x, y0, step_y, dst_dtype, batch, s, /*variant=*/a.out_w >= 128 ? 1 : 0);
        if (st != RWH_E_UNSUPPORTED) return st;  // else: shape outside the fast kernel's limits -> generic kernel
    }

    if (src_dtype == RWH_U8 && channels == 4 && interp == RWH_BILINEAR && dst_dtype == RWH_U8 && a.out_w >= 128) {
        const int st = launch_fast(a, inv_h, x0, step_x, y0, step_y, dst_dtype, batch, s, /*variant=*/1, 1, nullptr, 1, nullptr, 4);
        if (st != RWH_E_UNSUPPORTED) return st;
    }

    if (src_dtype == RWH_U8) return channels == 3 ? dispatch<unsigned char, 3>(a, interp, dst_dtype, s)
                                                  : dispatch<unsigned char, 4>(a, interp, dst_dtype, s);
    return channels == 3 ? dispatch<float, 3>(a, interp, dst_dtype, s) : dispatch<float, 4>(a, interp, dst_dtype, s);
}

extern "C" int rwh_warp_plan(int src_h, int src_w, int channels, int src_dtype, int batch, const double* inv_h, int n_h,
                             double x0, double step_x, double x_last, double y0, double step_y, double y_last,
                             int out_h, int out_w, int bound_h, int bound_w, int interp, int dst_dtype,
                             int row_begin, int row_end, unsigned flags, char* kernel_name, int name_len) {
    if (!kernel_name || name_len <= 0) return RWH_E_INVALID;
    kernel_name[0] = 0;
    rwh::g_plan_buf = kernel_name; rwh::g_plan_len = name_len;
    static unsigned char dummy[16];   // never dereferenced: every launch site returns before touching the device
    const int64_t src_stride = (int64_t)src_h * src_w * channels * (src_dtype == RWH_U8 ? 1 : 4);
    const int64_t dst_stride = (int64_t)(row_end - row_begin) * out_w * channels * (dst_dtype == RWH_U8 ? 1 : dst_dtype == RWH_F32 ? 4 : 8);
    const int st = rwh_warp_backward(dummy, src_h, src_w, channels, src_dtype, src_stride, batch, inv_h, n_h, x0, step_x, x_last, y0,
                                     step_y, y_last, out_h, out_w, bound_h, bound_w, interp, dummy, dst_dtype, dst_stride, row_begin,
                                     row_end, flags, nullptr);
    rwh::g_plan_buf = nullptr; rwh::g_plan_len = 0;
    return st;
}

extern "C" int rwh_warp_index_check(int src_h, int src_w, const double* inv_h, double x0, double step_x, double x_last,
                                    double y0, double step_y, double y_last, int out_h, int out_w, int bound_h, int bound_w,
                                    int interp, int* d_flag, void* stream);

namespace rwh {
// Would the reference raise IndexError on this warp?  Coordinates by the exact kernels' arithmetic (warp_exact); no image access.
// bits of *flag: 1 = an index past the last column (axis 1), 2 = past the last row (axis 0), 4 = a NaN coordinate (bilinear).
__global__ __launch_bounds__(256) void index_check_kernel(const WarpArgs a, int interp, int* flag) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)a.out_h * a.out_w) return;
    const int r = (int)(i / a.out_w), c = (int)(i - (long long)r * a.out_w);
    const double y = grid_coord(r, a.out_h, a.y0, a.step_y, a.y_last), x = grid_coord(c, a.out_w, a.x0, a.step_x, a.x_last);
    const double X = fma(a.ih[1], y, a.ih[0] * x) + a.ih[2];
    const double Y = fma(a.ih[4], y, a.ih[3] * x) + a.ih[5];
    const double W = fma(a.ih[7], y, a.ih[6] * x) + a.ih[8];
    const double sx = X / W, sy = Y / W;
    int bits = 0;
    if (interp == RWH_NEAREST) {                       // homography.py:110-119: the mask is on the integers, NaN -> INT_MIN is masked
        const int xi = (int)(sx + 0.5), yi = (int)(sy + 0.5);
        const bool unmasked = (xi >= 0) & (xi <= a.bound_w - 1) & (yi >= 0) & (yi <= a.bound_h - 1) & (sx == sx) & (sy == sy);
        if (unmasked) bits = (xi > a.src_w - 1 ? 1 : 0) | (yi > a.src_h - 1 ? 2 : 0);
    } else {                                           // homography.py:131-135: NaN passes the float mask and indexes with INT_MIN
        const bool masked = (sx > (double)(a.bound_w - 1)) | (sx < 0.0) | (sy > (double)(a.bound_h - 1)) | (sy < 0.0);
        if (!masked) {
            if (!(sx == sx) || !(sy == sy)) bits = 4;
            else bits = ((int)sx + 1 > a.src_w - 1 ? 1 : 0) | ((int)sy + 1 > a.src_h - 1 ? 2 : 0);
        }
    }
    if (bits) atomicOr(flag, bits);
}

template <typename SrcT, int C>
static int sample_dispatch(const unsigned char* img, int src_h, int src_w, int bound_h, int bound_w, const double* xs, const double* ys,
                           long long n, int interp, void* out, int dst_dtype, hipStream_t s) {
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (interp == RWH_NEAREST) {
        if (dst_dtype != elem<SrcT>::dtype) return RWH_E_UNSUPPORTED;
        hipLaunchKernelGGL((sample_points_kernel<SrcT, C, SrcT, RWH_NEAREST>), grid, block, 0, s, img, src_h, src_w, bound_h, bound_w, xs, ys, n,
                           static_cast<SrcT*>(out));
    } else if (dst_dtype == RWH_F64) {
        hipLaunchKernelGGL((sample_points_kernel<SrcT, C, double, RWH_BILINEAR>), grid, block, 0, s, img, src_h, src_w, bound_h, bound_w, xs, ys, n,
                           static_cast<double*>(out));
    } else {
        return RWH_E_UNSUPPORTED;
    }
    return check_launch();
}
}  // namespace rwh

extern "C" int rwh_sample_points(const void* d_img, int src_h, int src_w, int channels, int src_dtype, const double* d_x, const double* d_y,
                                 int64_t n, int bound_h, int bound_w, int interp, void* d_out, int dst_dtype, unsigned flags, void* stream) {
    using namespace rwh;
    if (!d_img || !d_x || !d_y || !d_out || n < 0 || src_h < 1 || src_w < 1 || bound_h <= 0 || bound_w <= 0) return RWH_E_INVALID;
    if (interp != RWH_NEAREST && interp != RWH_BILINEAR) return RWH_E_INVALID;
    if (channels != 3 && channels != 4) return RWH_E_UNSUPPORTED;
    if (src_dtype != RWH_U8 && src_dtype != RWH_F32) return RWH_E_UNSUPPORTED;
    if (n == 0) return RWH_OK;
    if (n > (1ll << 31) * 255) return RWH_E_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t esz = src_dtype == RWH_U8 ? 1 : 4;
    if (flags & RWH_WARP_ZERO_ORIGIN) {
        hipLaunchKernelGGL(zero_origin_kernel, dim3(1), dim3(64), 0, s, const_cast<unsigned char*>(static_cast<const unsigned char*>(d_img)),
                           0ll, 1, (int)(channels * esz));
        if (check_launch() != RWH_OK) return RWH_E_LAUNCH;
    }
    const unsigned char* img = static_cast<const unsigned char*>(d_img);
    const int bh = bound_h < src_h ? bound_h : src_h, bw = bound_w < src_w ? bound_w : src_w;
    if (src_dtype == RWH_U8)
        return channels == 3 ? sample_dispatch<unsigned char, 3>(img, src_h, src_w, bh, bw, d_x, d_y, n, interp, d_out, dst_dtype, s)
                             : sample_dispatch<unsigned char, 4>(img, src_h, src_w, bh, bw, d_x, d_y, n, interp, d_out, dst_dtype, s);
    return channels == 3 ? sample_dispatch<float, 3>(img, src_h, src_w, bh, bw, d_x, d_y, n, interp, d_out, dst_dtype, s)
                         : sample_dispatch<float, 4>(img, src_h, src_w, bh, bw, d_x, d_y, n, interp, d_out, dst_dtype, s);
}

extern "C" int rwh_warp_index_check(int src_h, int src_w, const double* inv_h, double x0, double step_x, double x_last,
                                    double y0, double step_y, double y_last, int out_h, int out_w, int bound_h, int bound_w,
                                    int interp, int* d_flag, void* stream) {
    using namespace rwh;
    if (!inv_h || !d_flag || src_h <= 0 || src_w <= 0 || out_h <= 0 || out_w <= 0 || bound_h <= 0 || bound_w <= 0) return RWH_E_INVALID;
    if (interp != RWH_NEAREST && interp != RWH_BILINEAR) return RWH_E_INVALID;
    WarpArgs a = {};
    for (int i = 0; i < 9; ++i) a.ih[i] = inv_h[i];
    a.x0 = x0; a.step_x = step_x; a.x_last = x_last; a.y0 = y0; a.step_y = step_y; a.y_last = y_last;
    a.src_h = src_h; a.src_w = src_w; a.bound_h = bound_h; a.bound_w = bound_w;      // the bound is NOT clipped here: it is the reference's mask
    a.out_h = out_h; a.out_w = out_w;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(d_flag, 0, sizeof(int), s) != hipSuccess) return RWH_E_LAUNCH;
    const long long n = (long long)out_h * out_w;
    if ((n + 255) / 256 >= (1ll << 31)) return RWH_E_UNSUPPORTED;
    hipLaunchKernelGGL(index_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, interp, d_flag);
    return check_launch();
}
