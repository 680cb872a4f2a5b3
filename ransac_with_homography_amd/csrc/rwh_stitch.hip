// Fused panorama compositor: replaces, in ONE pass over the canvas,
//   addAlpha(imgT, 'Rate' | 'Gradient', rate)                    homography.py:250-266
//   transformImageH(imgT[, alpha]) -> wrapPerspective + bilinear  homography.py:230-242, 142-184, 123-138
//   the canvas paste / alpha blend of stitchPanorama             homography.py:322-338
// Nothing is materialised: no H x W x 4 float32 alpha image, no float64 warped RGBA, no float32 canvas.
// Every canvas pixel evaluates the reference's float64 arithmetic in the reference's order (the warp part is
// warp_exact's recipe, rwh_warp.hip), so the uint8 canvas is bit-identical to stitchPanorama's.
#include "rwh_common.h"
#include "rwh_warp_rgb8.h"

namespace rwh {

int warp_composite(const unsigned char* d_img_t, int t_h, int t_w, const double* inv_h, double x0, double y0, int canvas_h,
                   int canvas_w, unsigned char* d_canvas, const CompArgs& comp, hipStream_t s);   // rwh_warp.hip

struct StitchArgs {
    const unsigned char* src_t;   // imgT, ht_src x wt_src x 3 uint8 (texel (0,0) already blanked)
    const unsigned char* src_q;   // imgQ, hq x wq x 3 uint8
    unsigned char* dst;           // canvas fh x fw x 3 uint8
    double ih[9];                 // inv(H)
    int t_h, t_w;                 // imgT size
    int q_h, q_w;                 // imgQ size
    int fh, fw;                   // canvas size
    int tsx, tsy, wt, ht;         // warped-T rectangle on the canvas and its size (= the warp's output grid)
    int gx0, gy0;                 // warp grid origin (min_x, min_y): output pixel (c, r) of the warp is at (gx0 + c, gy0 + r)
    int qsx, qsy;                 // imgQ rectangle origin on the canvas
    int blend;                    // 0 = paste (imgQ over warped imgT), 1 = 'Rate' alpha blend, 2 = 'Gradient' alpha blend,
                                  // 3 = any other truthy `blending`: addAlpha leaves imgT's alpha plane at 0 (homography.py:250-266)
    float alpha_t;                // 'Rate': float32(rate + 1e-10), the constant alpha plane of imgT
    double ramp_den;              // 'Gradient': w + h of imgT; alpha(x, y) = float32((x + y) / (w + h) * 0.5)
    float alpha_q_in, alpha_q_out;  // canvas alpha inside / outside the imgQ rectangle (float32)
    int row_begin, row_end;       // canvas rows this launch produces (rwh_stitch_panorama_rows)
};

// One RGB texel as a dword: an unaligned 4-byte load (3 bytes used) unless that would step past the image's last byte.
__device__ __forceinline__ uint32_t rgb_at(const unsigned char* base, size_t off, size_t img_bytes) {
    if (off + 4 <= img_bytes) return ld4(base + off);
    return (uint32_t)base[off] | ((uint32_t)base[off + 1] << 8) | ((uint32_t)base[off + 2] << 16);
}
__device__ __forceinline__ double chan(uint32_t texel, int k) { return (double)((texel >> (8 * k)) & 0xffu); }

// One canvas pixel, the reference's arithmetic operation by operation; returns the pixel as 0x00BBGGRR.
template <bool BLEND>
__device__ __forceinline__ uint32_t stitch_pixel(const StitchArgs& a, int cx, int cy, size_t t_bytes, size_t q_bytes) {
    const int qx = cx - a.qsx, qy = cy - a.qsy;
    const bool in_q = (qx >= 0) & (qx < a.q_w) & (qy >= 0) & (qy < a.q_h);
    const uint32_t q = in_q ? rgb_at(a.src_q, ((size_t)qy * a.q_w + qx) * 3, q_bytes) & 0xFFFFFFu : 0u;
    const int tx = cx - a.tsx, ty = cy - a.tsy;
    const bool in_t = (tx >= 0) & (tx < a.wt) & (ty >= 0) & (ty < a.ht);
    if (!BLEND && in_q) return q;    // paste: imgQ is written last (homography.py:337-338)

    double t_rgb[3] = {0.0, 0.0, 0.0}, t_a = 0.0;
    if (in_t) {
        // warp_exact's coordinate recipe: dgemm k-order, IEEE divides
        const double x = (double)(a.gx0 + tx), y = (double)(a.gy0 + ty);
        const double X = fma(a.ih[1], y, a.ih[0] * x) + a.ih[2];
        const double Y = fma(a.ih[4], y, a.ih[3] * x) + a.ih[5];
        const double W = fma(a.ih[7], y, a.ih[6] * x) + a.ih[8];
        const double sx = X / W, sy = Y / W;
        const bool valid = (sx >= 0.0) & (sx <= (double)(a.t_w - 1)) & (sy >= 0.0) & (sy <= (double)(a.t_h - 1));
        if (valid) {
            const int ix = (int)sx, iy = (int)sy;
            const double fx = sx - (double)ix, fy = sy - (double)iy;
            const double gx = 1.0 - fx, gy = 1.0 - fy;
            const int ix1 = min(ix + 1, a.t_w - 1), iy1 = min(iy + 1, a.t_h - 1);
            const uint32_t p00 = rgb_at(a.src_t, ((size_t)iy * a.t_w + ix) * 3, t_bytes);
            const uint32_t p01 = rgb_at(a.src_t, ((size_t)iy * a.t_w + ix1) * 3, t_bytes);
            const uint32_t p10 = rgb_at(a.src_t, ((size_t)iy1 * a.t_w + ix) * 3, t_bytes);
            const uint32_t p11 = rgb_at(a.src_t, ((size_t)iy1 * a.t_w + ix1) * 3, t_bytes);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double top = chan(p00, k) * gx + chan(p01, k) * fx;
                const double bot = chan(p10, k) * gx + chan(p11, k) * fx;
                t_rgb[k] = top * gy + bot * fy;
            }
            if constexpr (BLEND) {
                // the alpha plane is never read from memory: 'Rate' is a constant, 'Gradient' the float32 ramp of
                // homography.py:260-265 (linspace gives exact integers; float64 divide, * 0.5, stored as float32);
                // texel (0,0) is the one bilinear() blanks (the ramp is 0 there anyway)
                const double A = (double)a.alpha_t, den = a.ramp_den;
                const bool ramp = a.blend == 2;
                auto tap = [&](int x, int y) -> double {
                    if ((x | y) == 0) return 0.0;
                    return ramp ? (double)(float)(((double)x + (double)y) / den * 0.5) : A;
                };
                const double top = tap(ix, iy) * gx + tap(ix1, iy) * fx, bot = tap(ix, iy1) * gx + tap(ix1, iy1) * fx;
                t_a = top * gy + bot * fy;
            }
        }
    }
    uint32_t out = 0u;
    if constexpr (!BLEND) {          // paste, outside imgQ: truncated warp (transformImageH's astype(uint8)) or 0
#pragma unroll
        for (int k = 0; k < 3; ++k) out |= (uint32_t)(unsigned char)(int)t_rgb[k] << (8 * k);
        return out;
    }
    // float32 canvas: rgb = imgQ (or 0), alpha = alpha_q_in / alpha_q_out; blended inside the warped rectangle
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float cq = (float)((q >> (8 * k)) & 0xffu);       // 0 outside imgQ
        float v = cq;
        if (in_t) {
            const double qa = (double)(in_q ? a.alpha_q_in : a.alpha_q_out);
            const double base = qa + t_a;
            const double r = (qa / base) * (double)cq + (t_a / base) * t_rgb[k];
            v = (float)r;            // assignment into the float32 canvas
        }
        out |= (uint32_t)(unsigned char)(int)v << (8 * k);
    }
    return out;
}

// 4 consecutive canvas pixels per lane: taps and imgQ pixels come in as dwords, the 12 output bytes leave in one store
// (the texture-address path charges per lane and instruction, not per byte: 12 byte loads + 3 byte stores per pixel
//  made the one-pixel-per-thread form of round 1 2.4x slower)
constexpr int ST_PX = 4;
template <bool BLEND>
__global__ __launch_bounds__(256) void stitch_kernel(const StitchArgs a) {
    const int cx0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * ST_PX;
    const int cy = a.row_begin + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (cx0 >= a.fw || cy >= a.row_end) return;
    const size_t t_bytes = (size_t)a.t_h * a.t_w * 3, q_bytes = (size_t)a.q_h * a.q_w * 3;
    unsigned char* out = a.dst + ((size_t)cy * a.fw + cx0) * 3;
    uint32_t px[ST_PX];
#pragma unroll
    for (int j = 0; j < ST_PX; ++j) px[j] = cx0 + j < a.fw ? stitch_pixel<BLEND>(a, cx0 + j, cy, t_bytes, q_bytes) : 0u;
    if (cx0 + ST_PX <= a.fw) {
        pk3 w;
        w.a = px[0] | (px[1] << 24);
        w.b = (px[1] >> 8) | (px[2] << 16);
        w.c = (px[2] >> 16) | (px[3] << 8);
        __builtin_memcpy(out, &w, 12);
    } else {
        for (int j = 0; cx0 + j < a.fw; ++j) { out[3 * j] = (unsigned char)px[j]; out[3 * j + 1] = (unsigned char)(px[j] >> 8); out[3 * j + 2] = (unsigned char)(px[j] >> 16); }
    }
}

}  // namespace rwh

extern "C" int rwh_stitch_panorama_rows(const void* d_img_t, int t_h, int t_w, const void* d_img_q, int q_h, int q_w,
                                        const double* inv_h, int grid_x0, int grid_y0, int warp_w, int warp_h,
                                        int tsx, int tsy, int qsx, int qsy, int canvas_h, int canvas_w,
                                        int blend, double rate, void* d_canvas, int row_begin, int row_end, unsigned flags, void* stream);

extern "C" int rwh_stitch_panorama(const void* d_img_t, int t_h, int t_w, const void* d_img_q, int q_h, int q_w,
                                   const double* inv_h, int grid_x0, int grid_y0, int warp_w, int warp_h,
                                   int tsx, int tsy, int qsx, int qsy, int canvas_h, int canvas_w,
                                   int blend, double rate, void* d_canvas, unsigned flags, void* stream) {
    return rwh_stitch_panorama_rows(d_img_t, t_h, t_w, d_img_q, q_h, q_w, inv_h, grid_x0, grid_y0, warp_w, warp_h, tsx, tsy, qsx, qsy,
                                    canvas_h, canvas_w, blend, rate, d_canvas, 0, canvas_h, flags, stream);
}

extern "C" int rwh_stitch_panorama_rows(const void* d_img_t, int t_h, int t_w, const void* d_img_q, int q_h, int q_w,
                                        const double* inv_h, int grid_x0, int grid_y0, int warp_w, int warp_h,
                                        int tsx, int tsy, int qsx, int qsy, int canvas_h, int canvas_w,
                                        int blend, double rate, void* d_canvas, int row_begin, int row_end, unsigned flags, void* stream) {
    using namespace rwh;
    if (!d_img_t || !d_img_q || !d_canvas || !inv_h) return RWH_E_INVALID;
    if (row_begin < 0 || row_end > canvas_h || row_begin > row_end) return RWH_E_INVALID;
    const bool whole = row_begin == 0 && row_end == canvas_h;
    if (t_h < 2 || t_w < 2 || q_h <= 0 || q_w <= 0 || warp_w <= 0 || warp_h <= 0 || canvas_h <= 0 || canvas_w <= 0) return RWH_E_INVALID;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (flags & RWH_WARP_ZERO_ORIGIN) {
        if (hipMemsetAsync(const_cast<void*>(d_img_t), 0, 3, s) != hipSuccess) return RWH_E_LAUNCH;
    }
    if (blend < 0 || blend > 3) return RWH_E_INVALID;
    if (row_begin == row_end) return RWH_OK;
    if ((flags & RWH_STITCH_FAST) && blend < 2 && whole) {   // the staged compositor carries constant weights: no ramp (whole canvases only)
        // the reference's float32 alphas (see below), then the two weight pairs of the mean in float64 -> float32
        const double ta = (double)(float)(rate + 1e-10), qa_in = (double)(float)(1 + 1e-10 - rate), qa_out = (double)(float)1e-10;
        CompArgs c;
        c.q = static_cast<const unsigned char*>(d_img_q);
        c.q_w = q_w; c.q_h = q_h; c.qsx = qsx; c.qsy = qsy; c.tsx = tsx; c.tsy = tsy; c.wt = warp_w; c.ht = warp_h;
        c.mode = blend ? 2 : 1;
        c.wq_in = (float)(qa_in / (qa_in + ta)); c.wt_in = (float)(ta / (qa_in + ta));
        c.wq_out = (float)(qa_out / (qa_out + ta)); c.wt_out = (float)(ta / (qa_out + ta));
        const int st = warp_composite(static_cast<const unsigned char*>(d_img_t), t_h, t_w, inv_h, (double)(grid_x0 - tsx),
                                      (double)(grid_y0 - tsy), canvas_h, canvas_w, static_cast<unsigned char*>(d_canvas), c, s);
        if (st != RWH_E_UNSUPPORTED) return st;     // else: a shape the staged kernel does not take -> the exact kernel
    }
    StitchArgs a;
    a.src_t = static_cast<const unsigned char*>(d_img_t);
    a.src_q = static_cast<const unsigned char*>(d_img_q);
    a.dst = static_cast<unsigned char*>(d_canvas);
    for (int i = 0; i < 9; ++i) a.ih[i] = inv_h[i];
    a.t_h = t_h; a.t_w = t_w; a.q_h = q_h; a.q_w = q_w; a.fh = canvas_h; a.fw = canvas_w;
    a.tsx = tsx; a.tsy = tsy; a.wt = warp_w; a.ht = warp_h; a.gx0 = grid_x0; a.gy0 = grid_y0; a.qsx = qsx; a.qsy = qsy;
    a.blend = blend;
    a.row_begin = row_begin; a.row_end = row_end;
    a.ramp_den = (double)(t_w + t_h);
    // the reference's Python-float arithmetic, then the float32 storage of its arrays
    a.alpha_t = blend == 3 ? 0.0f                      // a `blending` that is neither 'Rate' nor 'Gradient': the alpha plane stays 0
                           : (float)(rate + 1e-10);    // addAlpha: rate += 1e-10; imgn[:, :, c] = rate   (float32 array)
    a.alpha_q_in = blend >= 2 ? 1.0f                   // imgn[q-rect, 3] = 1                                 (homography.py:329)
                              : (float)(1 + 1e-10 - rate);   // imgn[q-rect, 3] = 1 + 1e-10 - blendrate       (float32 array)
    a.alpha_q_out = (float)1e-10;                      // imgn[:, :, 3] += 1e-10 on a float32 zero
    const dim3 grid((canvas_w + 64 * ST_PX - 1) / (64 * ST_PX), (row_end - row_begin + 3) / 4), block(256);
    if (blend) hipLaunchKernelGGL(stitch_kernel<true>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(stitch_kernel<false>, grid, block, 0, s, a);
    return check_launch();
}
