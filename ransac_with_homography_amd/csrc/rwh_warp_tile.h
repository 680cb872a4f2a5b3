// K3 tile kernel (round 2): RGB u8 source, bilinear, u8 output -- the BASELINE configuration -- for maps whose source
// footprint of a 128 x 16 output tile is at most TL_ROWS rows x TL_TEX texels (scale ~1, a few degrees of rotation, any
// mild perspective).  Everything else stays on warp_rgb8_fast8 / the gather paths (rwh_warp_rgb8.h).
//
// Why (profiles/r02_isa_histogram_fast8.txt): warp_rgb8_fast8 issues 422 VALU instructions per 512-pixel wave and is
// VALU-issue-bound.  72 of them stage the footprint (chunk -> row / column arithmetic, RGB -> RGBX expansion), 96 are
// byte -> float converts in front of the packed FMAs.  This kernel removes both groups:
//
//   * staging is pure DMA: one block = one 128 x 16 output tile = ONE shared footprint (bounding box of the tile's four
//     mapped corners), copied row by row with global_load_lds_dwordx4 -- global memory straight into LDS, 16 bytes per
//     lane, no VGPRs, no VALU, raw RGB bytes (3 B / texel: a 19 x 133 footprint is 8 KB).  Source row = scalar base +
//     lane * 16, LDS row = M0: nothing per lane is computed at all;
//   * a bilinear tap is read from LDS byte by byte (ds_read_u8 zero-extends: the register then holds the float16
//     DENORMAL b * 2^-24) and blended with v_fma_mix_f32 (f16 x f32 + f32, honours f16 denormals -- checked on the
//     device, tools/isa_probe.hip), the 2^24 riding on the weights: no convert instructions;
//   * the blend's products and their order are those of blend_store (rwh_warp_rgb8.h): (b * 2^-24) * (w * 2^24) is the
//     same real number as float(b) * w and each FMA rounds once, so this kernel, warp_rgb8_fast8 and the gather path
//     produce identical bits; which kernel serves a warp is a function of the homography and the whole output grid only
//     (launch_fast), never of the row shard or the batch;
//   * coordinates as before: float64, affine numerators, one v_rcp_f64 per run of 4 pixels (batch inversion), magic-
//     number floor / fraction.  The footprint comes from the four tile corners evaluated on lanes 0..3 of every wave
//     (block-uniform by construction: same inputs, same instructions).  A pixel whose rounded coordinate differs from the
//     corner's by one 2^-32 grid step can address one texel outside the window: the tap that falls outside has weight
//     <= 2^-32 and reads a finite byte (the slab holds nothing but bytes; out-of-range LDS addresses read 0), so the
//     result moves by < 6e-8;
//   * tiles that touch the image border (floor < 1 or > bound - 3), cross the horizon or exceed the window take the
//     masked gather path, where the bounds test is made on the unrounded coordinate.
#pragma once
#include "rwh_warp_rgb8.h"

namespace rwh {

constexpr int TL_ROWS = 36;                              // source rows a tile may stage
#ifndef RWH_TL_PITCH
#define RWH_TL_PITCH 480
#endif
constexpr int TL_PITCH = RWH_TL_PITCH;                   // LDS bytes per staged row (a multiple of 16: DMA lanes)
constexpr int TL_TEX = (TL_PITCH - 16) / 3;              // 144 texels (+ up to 15 bytes of row-start slack)
constexpr float W_SCALE24 = 9.094947017729282e-13f;      // 2^-40: y weights carry 2^-64 (x weights are * 2^32) and 2^24 (f16 denormal)
constexpr float W_ONE24 = 0.00390625f;                   // 2^-8 = the y weights' "1.0" at that scale

__device__ __forceinline__ float den16(uint32_t byte_in_low_half) {      // low 16 bits as float16 (a byte: the denormal b * 2^-24)
    return (float)__builtin_bit_cast(_Float16, (unsigned short)byte_in_low_half);
}

// 4 taps x 3 channels of zero-extended bytes -> one pixel.  Same products, same order as blend_store.
template <bool U8>
__device__ __forceinline__ void blend_px(const uint32_t (&t)[4][3], float w00, float w01, float w10, float w11, float (&o)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float acc = __builtin_fmaf(den16(t[0][k]), w00, U8 ? U8_BIAS : 0.f);
        acc = __builtin_fmaf(den16(t[1][k]), w01, acc);
        acc = __builtin_fmaf(den16(t[2][k]), w10, acc);
        acc = __builtin_fmaf(den16(t[3][k]), w11, acc);
        o[k] = acc;
    }
}

// The 12 bytes of a pixel's four taps (texels (iy, ix), (iy, ix+1) and the two below) straight from the raw-RGB slab:
// ds_read_u8 zero-extends, so each register's low half is the float16 denormal b * 2^-24.  Spelled in assembly because
// hipcc merges adjacent byte loads into ds_read_b32 + ds_read_u16 at byte alignment -- misaligned LDS reads run 25x
// slower on gfx950 (tools/isa_probe.hip) -- and then spends ~12 VALU per pixel taking the bytes apart again.  The
// compiler does not count these loads in lgkmcnt, hence tap_wait: LDS returns in order, so "at most N outstanding"
// after issuing N more means this pixel's twelve have landed; the registers pass through the wait as in/out operands,
// which orders every use behind it.
__device__ __forceinline__ void tap_issue(uint32_t lds_addr, uint32_t (&t)[4][3]) {
    asm volatile(
        "ds_read_u8 %0, %12\n ds_read_u8 %1, %12 offset:1\n ds_read_u8 %2, %12 offset:2\n"
        "ds_read_u8 %3, %12 offset:3\n ds_read_u8 %4, %12 offset:4\n ds_read_u8 %5, %12 offset:5\n"
        "ds_read_u8 %6, %12 offset:%13\n ds_read_u8 %7, %12 offset:%13+1\n ds_read_u8 %8, %12 offset:%13+2\n"
        "ds_read_u8 %9, %12 offset:%13+3\n ds_read_u8 %10, %12 offset:%13+4\n ds_read_u8 %11, %12 offset:%13+5"
        : "=&v"(t[0][0]), "=&v"(t[0][1]), "=&v"(t[0][2]), "=&v"(t[1][0]), "=&v"(t[1][1]), "=&v"(t[1][2]),
          "=&v"(t[2][0]), "=&v"(t[2][1]), "=&v"(t[2][2]), "=&v"(t[3][0]), "=&v"(t[3][1]), "=&v"(t[3][2])
        : "v"(lds_addr), "n"(TL_PITCH));
}
template <int N>
__device__ __forceinline__ void tap_wait(uint32_t (&t)[4][3]) {
    asm volatile("s_waitcnt lgkmcnt(%12)"
                 : "+v"(t[0][0]), "+v"(t[0][1]), "+v"(t[0][2]), "+v"(t[1][0]), "+v"(t[1][1]), "+v"(t[1][2]),
                   "+v"(t[2][0]), "+v"(t[2][1]), "+v"(t[2][2]), "+v"(t[3][0]), "+v"(t[3][1]), "+v"(t[3][2])
                 : "n"(N));
}

__device__ __forceinline__ void store_run_u8(const float (&o)[FP_PX][3], unsigned char* drow, bool store_any, int shift) {
    if (!store_any) return;
    if (shift <= 0) {
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
    } else {  // the one straddling lane of a ragged row
#pragma unroll
        for (int j = 1; j < FP_PX; ++j)
            if (j >= shift) {
#pragma unroll
                for (int k = 0; k < 3; ++k) drow[3 * j + k] = (unsigned char)__builtin_amdgcn_cvt_pk_u8_f32(o[j][k], 0, 0);
            }
    }
}

template <int LOG_PW>
__device__ __forceinline__ void tile_body(const FastArgs& a, const Coef* tab) {
    constexpr int PW = 1 << LOG_PW, PH = 512 / PW;          // one wave's patch inside the 128 x 16 tile
    constexpr int LPR = PW / 8;                             // lanes per patch row
    constexpr int WX = 128 / PW;                            // waves side by side
    __shared__ __attribute__((aligned(16))) unsigned char slab[TL_ROWS * TL_PITCH];

    // ---- block / wave -> patch (all scalar) ---------------------------------------------------------
    const unsigned b = blockIdx.x;
    const unsigned logical = (b & 7u) * a.cpx + (b >> 3);   // XCD k walks logical blocks [k*cpx, (k+1)*cpx)
    if (logical >= a.nblocks) return;
    const unsigned t = a.tiles_x_magic ? __umulhi(logical, a.tiles_x_magic) : logical;
    const unsigned tx = logical - t * a.tiles_x;
    const unsigned img = a.tiles_y_magic ? __umulhi(t, a.tiles_y_magic) : t;
    const unsigned ty = t - img * a.tiles_y;
    const Coef& co = tab ? tab[img] : a.c;
    const unsigned img_mem = tab ? (unsigned)co.image : img;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int prow = lane / LPR, pq = lane % LPR;
    const int wave_x = (wave % WX) * PW, wave_y = (wave / WX) * PH;

    const int rr_raw = (int)ty * 16 + wave_y + prow;
    const int rr = min(rr_raw, a.rows - 1);                   // rows past the shard recompute its last row
    const int tcol0 = (int)tx * 128;
    const int tcol = min(tcol0, a.out_w - 128);               // a tile that sticks out of the row is moved left as a whole
    const int tshift = tcol0 - tcol;
    const int lcol = wave_x + pq * 4;
    const int c0p = tcol + lcol;
    const bool store_any = rr_raw < a.rows;
    const unsigned char* simg = a.src + (long long)img_mem * a.src_img_stride;
    unsigned char* dimg = a.dst + (long long)img_mem * a.dst_img_stride;
    unsigned char* drow = dimg + ((uint32_t)rr * (uint32_t)a.out_w + (uint32_t)c0p) * 3u;
    const uint32_t pitch = (uint32_t)a.src_w * 3u;

    // ---- the tile's four corners, one per lane & 3 (every wave of the block: same inputs -> same footprint) --------
    int xmn, xmx, ymn, ymx;
    bool wpos;
    {
        const int k = lane & 3;
        const int crow = min((int)ty * 16 + ((k & 2) ? 15 : 0), a.rows - 1);
        const double fr = (double)(a.row_begin + crow), fc = (double)(tcol + ((k & 1) ? 127 : 0));
        const double X = fma(fc, co.cx[2], fma(fr, co.cx[1], co.cx[0]));
        const double Y = fma(fc, co.cy[2], fma(fr, co.cy[1], co.cy[0]));
        const double W = fma(fc, co.cw[2], fma(fr, co.cw[1], co.cw[0]));
        double r = __builtin_amdgcn_rcp(W); r = fma(fma(-W, r, 1.0), r, r);
        const uint32_t chx = hi32(fma(X, r, MAGIC)), chy = hi32(fma(Y, r, MAGIC));
        // W is affine: positive at the four corners <=> positive on the tile; magnitudes inside [2^-250, 2^250] so that
        // the product of the four W of a run is a finite normal number
        const int hw = (int)hi32(W);
        wpos = __all((int)(hw > 0x30500000) & (int)(hw < 0x4F900000));
        const int x0 = (int)__builtin_amdgcn_readlane(chx, 0), x1 = (int)__builtin_amdgcn_readlane(chx, 1);
        const int x2 = (int)__builtin_amdgcn_readlane(chx, 2), x3 = (int)__builtin_amdgcn_readlane(chx, 3);
        const int y0 = (int)__builtin_amdgcn_readlane(chy, 0), y1 = (int)__builtin_amdgcn_readlane(chy, 1);
        const int y2 = (int)__builtin_amdgcn_readlane(chy, 2), y3 = (int)__builtin_amdgcn_readlane(chy, 3);
        // hi dwords compare like the integers they encode (same exponent); out-of-range / NaN corners end up as the min
        // or the max and fail the range test below
        xmn = (int)((uint32_t)smin(smin(x0, x1), smin(x2, x3)) - MAGIC_HI); xmx = (int)((uint32_t)smax(smax(x0, x1), smax(x2, x3)) - MAGIC_HI);
        ymn = (int)((uint32_t)smin(smin(y0, y1), smin(y2, y3)) - MAGIC_HI); ymx = (int)((uint32_t)smax(smax(y0, y1), smax(y2, y3)) - MAGIC_HI);
    }
    // window: rows ymn .. ymx+1, texels xmn .. xmx+1; strictly inside the image with one texel to spare on every side
    const int nrows = ymx - ymn + 2, ntex = xmx - xmn + 2;
    const bool staged = wpos & (xmn >= 1) & (xmx <= a.bound_w - 3) & (ymn >= 1) & (ymx <= min(a.bound_h, a.src_h) - 3) &
                        (nrows <= TL_ROWS) & (ntex <= TL_TEX);

    // ---- DMA: source rows -> LDS, 16 bytes per lane, row r of the window by wave r & 3 ---------------------------------
    const uint32_t g0 = (uint32_t)ymn * pitch + (uint32_t)xmn * 3u;      // window origin in the image (host: image < 4 GB)
    // rows are copied from their 16-byte boundary (host: image base, image stride and row pitch are multiples of 16 --
    // otherwise launch_fast does not pick this kernel -- so every row of the window has the same skew)
    const uint32_t skew = g0 & 15u;
    if (staged) {
        const int nl = (int)((skew + 3u * (uint32_t)ntex + 15u) >> 4);     // <= 28
        if (lane < nl) {
            const unsigned char* gl = simg + (g0 - skew) + 16u * (uint32_t)lane;
            for (int r = wave; r < nrows; r += 4)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gl + (size_t)r * pitch),
                                                 (void __attribute__((address_space(3)))*)(slab + r * TL_PITCH), 16, 0, 0);
        }
    }

    // ---- the lane's 8 pixels: two runs of 4 consecutive pixels, PW/2 columns apart ------------------------------------
    const double fr = (double)(a.row_begin + rr), fc = (double)c0p;
    const double X0 = fma(fc, co.cx[2], fma(fr, co.cx[1], co.cx[0]));
    const double Y0 = fma(fc, co.cy[2], fma(fr, co.cy[1], co.cy[0]));
    const double W0 = fma(fc, co.cw[2], fma(fr, co.cw[1], co.cw[0]));
    uint32_t lx[FP_PX], ly[FP_PX], hx[FP_PX], hy[FP_PX];
    double sx[FP_PX], sy[FP_PX];                                   // unrounded coordinates (gather path's bounds test)
    auto run_coords = [&](const int h, const bool want_s) {
        double X[FP_PX], Y[FP_PX], W[FP_PX], rc[FP_PX];
#pragma unroll
        for (int j = 0; j < FP_PX; ++j) {
            if (h == 0 && j == 0) { X[j] = X0; Y[j] = Y0; W[j] = W0; }
            else { X[j] = X0 + co.dxs8[4 * h + j - 1][0]; Y[j] = Y0 + co.dxs8[4 * h + j - 1][1]; W[j] = W0 + co.dxs8[4 * h + j - 1][2]; }
        }
        if (wpos) {                                                // one reciprocal for the run (Montgomery batch inversion)
            const double p01 = W[0] * W[1], p23 = W[2] * W[3], P = p01 * p23;
            double rp = __builtin_amdgcn_rcp(P);
            rp = fma(fma(-P, rp, 1.0), rp, rp);
            const double r01 = rp * p23, r23 = rp * p01;
            rc[0] = r01 * W[1]; rc[1] = r01 * W[0]; rc[2] = r23 * W[3]; rc[3] = r23 * W[2];
        } else {                                                   // a W at / across zero (the horizon): pixel by pixel
#pragma unroll
            for (int j = 0; j < FP_PX; ++j) { double q = __builtin_amdgcn_rcp(W[j]); rc[j] = fma(fma(-W[j], q, 1.0), q, q); }
        }
#pragma unroll
        for (int j = 0; j < FP_PX; ++j) {
            const double ux = fma(X[j], rc[j], MAGIC), uy = fma(Y[j], rc[j], MAGIC);   // one rounding, onto the 2^-32 grid
            hx[j] = hi32(ux); lx[j] = lo32(ux); hy[j] = hi32(uy); ly[j] = lo32(uy);
            if (want_s) { sx[j] = X[j] * rc[j]; sy[j] = Y[j] * rc[j]; }
        }
    };

    if (staged) {
        // tap (iy, ix) lives at slab byte (iy - ymn) * TL_PITCH + 3 * (ix - xmn) + skew; straight from the hi dwords:
        // the 24-bit multiplies see 0x380000 + i, the constants go into one uniform
        const uint32_t tap_c = (0x380000u + (uint32_t)ymn) * (uint32_t)TL_PITCH + 3u * (0x380000u + (uint32_t)xmn) - skew -
                               (uint32_t)(uintptr_t)(unsigned char __attribute__((address_space(3)))*)slab;
        float o[FP_PX][3];
        run_coords(0, false);
        // the DMA has had the coordinate arithmetic to land
        __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0): this wave's rows are in LDS
        __syncthreads();
        uint32_t tp[2][4][3];                                      // taps of two pixels in flight
        uint32_t lo[FP_PX];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h) run_coords(h, false);
#pragma unroll
            for (int j = 0; j < FP_PX; ++j) {
                asm("v_mad_u32_u24 %0, %1, 3, %2" : "=v"(lo[j]) : "v"(hx[j]), "s"(0u - tap_c));
                lo[j] = mad24_s(hy[j], (uint32_t)TL_PITCH, lo[j]);
            }
            tap_issue(lo[0], tp[0]);
#pragma unroll
            for (int j = 0; j < FP_PX; ++j) {
                if (j + 1 < FP_PX) tap_issue(lo[j + 1], tp[(j + 1) & 1]);
                float wx0, wx1, wy0, wy1;
                weights(lx[j], ly[j], W_SCALE24, W_ONE24, wx0, wx1, wy0, wy1);
                const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
                if (j + 1 < FP_PX) tap_wait<12>(tp[j & 1]); else tap_wait<0>(tp[j & 1]);
                blend_px<true>(tp[j & 1], w00, w01, w10, w11, o[j]);
            }
            const int first = tshift - (lcol + (PW / 2) * h);    // local pixels at columns >= first are this tile's
            store_run_u8(o, drow + 3 * (PW / 2) * h, store_any & (first <= 3), max(first, 0));
        }
        return;
    }

    // ---- border / horizon / oversize tiles: masked gathers from global memory (bounds on the unrounded coordinate) -----
    const double xlim = (double)(a.bound_w - 1), ylim = (double)(a.bound_h - 1);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float o[FP_PX][3];
        run_coords(h, true);
        uint32_t off[FP_PX];
        bool valid[FP_PX], near_end = false;
#pragma unroll
        for (int j = 0; j < FP_PX; ++j) {
            // homography.py:131 masks on the float64 coordinate itself; the magic-number form must agree with it
            const unsigned long long ubx = ((unsigned long long)hx[j] << 32) | lx[j], uby = ((unsigned long long)hy[j] << 32) | ly[j];
            valid[j] = (sx[j] >= 0.0) & (sx[j] <= xlim) & (sy[j] >= 0.0) & (sy[j] <= ylim) &
                       (ubx >= MAGIC_BITS) & (ubx <= a.xmax_bits) & (uby >= MAGIC_BITS) & (uby <= a.ymax_bits);
            const int ix = (int)(hx[j] - MAGIC_HI), iy = (int)(hy[j] - MAGIC_HI);
            off[j] = valid[j] ? (uint32_t)iy * pitch + (uint32_t)ix * 3u : 0u;
            near_end |= valid[j] & (iy > a.src_h - 3);
        }
        const bool guarded = __any(near_end);
        const uint32_t last = (uint32_t)a.src_h * pitch - 3u;
#pragma unroll
        for (int j = 0; j < FP_PX; ++j) {
            float wx0, wx1, wy0, wy1;
            weights(lx[j], ly[j], valid[j] ? W_SCALE24 : 0.f, valid[j] ? W_ONE24 : 0.f, wx0, wx1, wy0, wy1);
            uint32_t a0, b0, a1, b1;
            if (!guarded) {
                const pk2 r0 = ld8(simg + off[j]);
                const pk2 r1 = ld8(simg + off[j] + pitch);
                a0 = r0.a; b0 = __builtin_amdgcn_alignbyte(r0.b, r0.a, 3);
                a1 = r1.a; b1 = __builtin_amdgcn_alignbyte(r1.b, r1.a, 3);
            } else {      // byte-exact loads, +1 taps clamped to the image (their weight is 0 when clamped)
                const uint32_t o00 = off[j];
                const uint32_t o01 = min(o00 + 3u, last), o10 = min(o00 + pitch, last), o11 = min(o00 + pitch + 3u, last);
                a0 = simg[o00] | (simg[o00 + 1] << 8) | (simg[o00 + 2] << 16);
                b0 = simg[o01] | (simg[o01 + 1] << 8) | (simg[o01 + 2] << 16);
                a1 = simg[o10] | (simg[o10 + 1] << 8) | (simg[o10 + 2] << 16);
                b1 = simg[o11] | (simg[o11 + 1] << 8) | (simg[o11 + 2] << 16);
            }
            uint32_t tp[4][3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                tp[0][k] = (a0 >> (8 * k)) & 0xFFu; tp[1][k] = (b0 >> (8 * k)) & 0xFFu;
                tp[2][k] = (a1 >> (8 * k)) & 0xFFu; tp[3][k] = (b1 >> (8 * k)) & 0xFFu;
            }
            blend_px<true>(tp, wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1, o[j]);
        }
        const int first = tshift - (lcol + (PW / 2) * h);
        store_run_u8(o, drow + 3 * (PW / 2) * h, store_any & (first <= 3), max(first, 0));
    }
}

template <int LOG_PW>
__global__ __launch_bounds__(256) void warp_rgb8_tile(const FastArgs a) { tile_body<LOG_PW>(a, nullptr); }
template <int LOG_PW>
__global__ __launch_bounds__(256) void warp_rgb8_tile_tab(const FastArgs a, const CoefTab t) { tile_body<LOG_PW>(a, t.e); }

// Host-side twin of the tile footprint test: does the 128 x 16 tile whose top-left pixel is (row r, column c) fit the window?
inline bool tile_fits(const FastArgs& a, double r, double c) {
    long long chunks; double lines;
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    for (int k = 0; k < 4; ++k) {
        const double rr = r + (k & 2 ? 15 : 0), cc = c + (k & 1 ? 127 : 0);
        const double X = a.c.cx[0] + rr * a.c.cx[1] + cc * a.c.cx[2], Y = a.c.cy[0] + rr * a.c.cy[1] + cc * a.c.cy[2];
        const double W = a.c.cw[0] + rr * a.c.cw[1] + cc * a.c.cw[2];
        if (!(W > 0)) return false;
        const double x = __builtin_floor(X / W), y = __builtin_floor(Y / W);
        xmin = x < xmin ? x : xmin; xmax = x > xmax ? x : xmax; ymin = y < ymin ? y : ymin; ymax = y > ymax ? y : ymax;
    }
    (void)chunks; (void)lines;
    return (ymax - ymin + 2 <= TL_ROWS) && (xmax - xmin + 2 <= TL_TEX);
}

}  // namespace rwh
