// EXPERIMENT (not part of the product; included only by tools/warp_lab.hip).
// Round-1 result: correct (same output as warp_rgb8_fast8) but NOT faster (17.7-18.3 us vs 17.5 us per 4K frame):
// hipcc guards the re-used prefetch registers with `s_waitcnt vmcnt(0)` at the loop back-edge, and vmcnt is in-order,
// so every iteration also waits for the previous patch's stores.  Kept as the starting point for a version with
// hand-counted waits / LDS-DMA staging.
#pragma once
#include "../ransac_with_homography_amd/csrc/rwh_warp_rgb8.h"

namespace rwh {

// ================================================================================================
// Software-pipelined variant: one wave walks G vertically adjacent 128 x 4 patches and keeps the NEXT
// patch's source tile in flight (registers) while it blends the current one.
// Why: with one patch per wave the chip holds too few bytes in flight to cover the beyond-L2 latency
// (Little's law: ~4.5 KB per wave during ~20 % of its life ~ 18 KB per CU; measured: confining the
// loads/stores to an L2-resident window takes the kernel from 17.3 to 13.4 us per 4K frame although HBM
// runs at only 2.7 TB/s).  Here every wave has a 4.5 KB tile in flight all the time.
// Per patch:  [A] expand + LDS-write the tile that was prefetched,  [B] corner pixels of the next patch
// (per-pixel reciprocal) -> footprint -> issue its loads,  [C] LDS taps + blend + store of this patch,
// [D] the other six pixels of the next patch (batch inversion).  Pixels 0 and 7 are computed ONCE, in [B],
// so the footprint and the taps can never disagree about a floor().
// ================================================================================================
struct Foot {  // wave-uniform footprint of a patch, in hi-dword (MAGIC_HI-biased) units
    int hxmn, hxmx, hymn, hymx;
    bool staged;
};

template <typename DstT>
__global__ __launch_bounds__(256) void warp_rgb8_pipe(const FastArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char slab[4][FP_ROWS * F8_PITCH];

    const unsigned b = blockIdx.x;
    const unsigned logical = (b & 7u) * a.cpx + (b >> 3);
    if (logical >= a.nblocks) return;
    const unsigned t = a.tiles_x_magic ? __umulhi(logical, a.tiles_x_magic) : logical;
    const unsigned tx = logical - t * a.tiles_x;
    const unsigned img = a.tiles_y_magic ? __umulhi(t, a.tiles_y_magic) : t;
    const unsigned ty = t - img * a.tiles_y;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int prow = lane >> 4, pq = lane & 15;
    const int G = a.group;
    const int row0 = ((int)ty * 4 + wave) * 4 * G;             // first output row of this wave's strip (uniform)
    if (row0 >= a.rows) return;
    const int npatch = min(G, (a.rows - row0 + 3) >> 2);       // uniform

    const int c0 = (int)tx * 128 + pq * F8_PX;
    const int c0p = min(c0, a.out_w - F8_PX);
    const int shift = c0 - c0p;
    const bool col_ok = c0 < a.out_w;
    const unsigned char* simg = a.src + (long long)img * a.src_img_stride;
    unsigned char* dimg = a.dst + (long long)img * a.dst_img_stride;
    const uint32_t pitch = (uint32_t)a.src_w * 3u;
    unsigned char* my = slab[wave];

    // lane constants: column part of X, Y, W; staging role
    const double fc = (double)c0p;
    const double Xc = fma(fc, a.cx[2], a.cx[0]), Yc = fma(fc, a.cy[2], a.cy[0]), Wc = fma(fc, a.cw[2], a.cw[0]);
    const int srow = (lane * 49) >> 10;
    const int scol = lane - F8_LANES * srow;
    const uint32_t goff = (uint32_t)srow * pitch + (uint32_t)scol * 24u;
    unsigned char* wlds = my + srow * F8_PITCH + scol * 32;

    // per-patch state
    uint32_t lx[F8_PX], ly[F8_PX], hx[F8_PX], hy[F8_PX];        // current patch
    uint32_t nlx0, nly0, nhx0, nhy0, nlx7, nly7, nhx7, nhy7;    // pixels 0 and 7 of the next patch
    double nX0, nY0, nW0;                                        // X, Y, W of pixel 0 of the next patch
    pk4 va[FP_ROWS / 3]; pk2 vb[FP_ROWS / 3]; bool on[FP_ROWS / 3];
    Foot cur, nxt;

    // pixels 0 and 7 of patch g (per-pixel reciprocal), footprint, and the staging loads of its tile
    auto corners_and_prefetch = [&](int g) {
        const int rr = min(row0 + 4 * g + prow, a.rows - 1);
        const double fr = (double)(a.row_begin + rr);
        nX0 = fma(fr, a.cx[1], Xc); nY0 = fma(fr, a.cy[1], Yc); nW0 = fma(fr, a.cw[1], Wc);
        const double X7 = nX0 + a.dxs8[6][0], Y7 = nY0 + a.dxs8[6][1], W7 = nW0 + a.dxs8[6][2];
        double r0 = __builtin_amdgcn_rcp(nW0); r0 = fma(fma(-nW0, r0, 1.0), r0, r0);
        double r7 = __builtin_amdgcn_rcp(W7);  r7 = fma(fma(-W7, r7, 1.0), r7, r7);
        const double ux0 = nX0 * r0 + MAGIC, uy0 = nY0 * r0 + MAGIC, ux7 = X7 * r7 + MAGIC, uy7 = Y7 * r7 + MAGIC;
        nhx0 = hi32(ux0); nlx0 = lo32(ux0); nhy0 = hi32(uy0); nly0 = lo32(uy0);
        nhx7 = hi32(ux7); nlx7 = lo32(ux7); nhy7 = hi32(uy7); nly7 = lo32(uy7);
        const bool wpos = __all((int)((int)hi32(nW0) > 0) & (int)((int)hi32(W7) > 0));
        const int x0 = (int)__builtin_amdgcn_readlane(nhx0, 0), x1 = (int)__builtin_amdgcn_readlane(nhx7, 15);
        const int x2 = (int)__builtin_amdgcn_readlane(nhx0, 48), x3 = (int)__builtin_amdgcn_readlane(nhx7, 63);
        const int y0 = (int)__builtin_amdgcn_readlane(nhy0, 0), y1 = (int)__builtin_amdgcn_readlane(nhy7, 15);
        const int y2 = (int)__builtin_amdgcn_readlane(nhy0, 48), y3 = (int)__builtin_amdgcn_readlane(nhy7, 63);
        nxt.hxmn = __builtin_amdgcn_readfirstlane(min(min(x0, x1), min(x2, x3)));
        nxt.hxmx = __builtin_amdgcn_readfirstlane(max(max(x0, x1), max(x2, x3)));
        nxt.hymn = __builtin_amdgcn_readfirstlane(min(min(y0, y1), min(y2, y3)));
        nxt.hymx = __builtin_amdgcn_readfirstlane(max(max(y0, y1), max(y2, y3)));
        const int xmn = (int)((uint32_t)nxt.hxmn - MAGIC_HI), xmx = (int)((uint32_t)nxt.hxmx - MAGIC_HI);
        const int ymn = (int)((uint32_t)nxt.hymn - MAGIC_HI), ymx = (int)((uint32_t)nxt.hymx - MAGIC_HI);
        nxt.staged = wpos & (xmn >= 0) & (xmx < a.bound_w - 1) & (ymn >= 0) & (ymx < min(a.bound_h - 1, a.src_h - 2)) &
                     (ymx - ymn + 2 <= FP_ROWS) & (xmx - xmn + 2 <= F8_TEXELS);
        if (nxt.staged) {
            const int nrows = ymx - ymn + 2, ntex = xmx - xmn + 2;
            const unsigned char* gbase = simg + (size_t)((uint32_t)ymn * pitch + (uint32_t)xmn * 3u);
            const bool mine = (srow < 3) & (8 * scol < ntex);
#pragma unroll
            for (int k = 0; k < FP_ROWS / 3; ++k) {
                on[k] = mine & (3 * k + srow < nrows);
                va[k] = pk4{0u, 0u, 0u, 0u}; vb[k] = pk2{0u, 0u};
                if (on[k]) {
                    __builtin_memcpy(&va[k], gbase + (size_t)(3 * k) * pitch + goff, 16);
                    __builtin_memcpy(&vb[k], gbase + (size_t)(3 * k) * pitch + goff + 16, 8);
                }
            }
        }
    };

    // pixels 1..6 of the patch whose pixel-0 terms are in nX0/nY0/nW0 (batch inversion), then pixels 0 / 7 copied in
    auto finish_coords = [&]() {
        double X[6], Y[6], W[6], rc[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) { X[j] = nX0 + a.dxs8[j][0]; Y[j] = nY0 + a.dxs8[j][1]; W[j] = nW0 + a.dxs8[j][2]; }
        const double p12 = W[0] * W[1], p34 = W[2] * W[3], p56 = W[4] * W[5], pa = p12 * p34, P = pa * p56;
        if (__all((int)__builtin_amdgcn_class(P, 0x100) & (int)((int)hi32(W[0]) > 0) & (int)((int)hi32(W[5]) > 0))) {
            double rp = __builtin_amdgcn_rcp(P);
            rp = fma(fma(-P, rp, 1.0), rp, rp);
            const double r56 = rp * pa, rpa = rp * p56, r12 = rpa * p34, r34 = rpa * p12;
            rc[0] = r12 * W[1]; rc[1] = r12 * W[0]; rc[2] = r34 * W[3]; rc[3] = r34 * W[2]; rc[4] = r56 * W[5]; rc[5] = r56 * W[4];
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) { double q = __builtin_amdgcn_rcp(W[j]); rc[j] = fma(fma(-W[j], q, 1.0), q, q); }
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const double ux = X[j] * rc[j] + MAGIC, uy = Y[j] * rc[j] + MAGIC;
            hx[j + 1] = hi32(ux); lx[j + 1] = lo32(ux); hy[j + 1] = hi32(uy); ly[j + 1] = lo32(uy);
        }
        hx[0] = nhx0; lx[0] = nlx0; hy[0] = nhy0; ly[0] = nly0;
        hx[7] = nhx7; lx[7] = nlx7; hy[7] = nhy7; ly[7] = nly7;
    };

    corners_and_prefetch(0);
    finish_coords();
    cur = nxt;

#pragma unroll 1
    for (int g = 0; g < npatch; ++g) {
        // ---- [A] the prefetched tile of this patch goes to LDS ----------------------------------------------------
        if (cur.staged) {
#pragma unroll
            for (int k = 0; k < FP_ROWS / 3; ++k) {
                if (on[k]) {
                    uint4 t4, t5;
                    t4.x = va[k].a;
                    t4.y = __builtin_amdgcn_alignbyte(va[k].b, va[k].a, 3);
                    t4.z = __builtin_amdgcn_alignbyte(va[k].c, va[k].b, 2);
                    t4.w = va[k].c >> 8;
                    t5.x = va[k].d;
                    t5.y = __builtin_amdgcn_alignbyte(vb[k].a, va[k].d, 3);
                    t5.z = __builtin_amdgcn_alignbyte(vb[k].b, vb[k].a, 2);
                    t5.w = vb[k].b >> 8;
                    *reinterpret_cast<uint4*>(wlds + 3 * k * F8_PITCH) = t4;
                    *reinterpret_cast<uint4*>(wlds + 3 * k * F8_PITCH + 16) = t5;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- [B] next patch: corner pixels, footprint, loads in flight ----------------------------------------------
        const bool more = g + 1 < npatch;                           // uniform
        if (more) corners_and_prefetch(g + 1);

        // ---- [C] this patch: taps, blend, store ----------------------------------------------------------------------
        const int rr_raw = row0 + 4 * g + prow;
        const int rr = min(rr_raw, a.rows - 1);
        const bool store_any = (rr_raw < a.rows) & col_ok;
        DstT* drow = reinterpret_cast<DstT*>(dimg + ((uint32_t)rr * (uint32_t)a.out_w + (uint32_t)c0p) * (uint32_t)(3 * sizeof(DstT)));
        uint32_t a0[FP_PX], b0[FP_PX], a1[FP_PX], b1[FP_PX];
        float wx0[FP_PX], wx1[FP_PX], wy0[FP_PX], wy1[FP_PX];
        if (cur.staged) {
            const uint32_t lds_c = (uint32_t)cur.hymn * (uint32_t)F8_PITCH + (uint32_t)cur.hxmn * 4u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int j = 0; j < FP_PX; ++j) {
                    const int q = 4 * h + j;
                    wx1[j] = (float)lx[q]; wx0[j] = (float)(~lx[q]);
                    wy1[j] = (float)ly[q] * W_SCALE; wy0[j] = (float)(~ly[q]) * W_SCALE;
                    const uint32_t lo = hy[q] * (uint32_t)F8_PITCH + hx[q] * 4u - lds_c;
                    const uint32_t* t0 = reinterpret_cast<const uint32_t*>(my + lo);
                    a0[j] = t0[0]; b0[j] = t0[1]; a1[j] = t0[F8_PITCH / 4]; b1[j] = t0[F8_PITCH / 4 + 1];
                }
                blend_store<DstT>(a0, b0, a1, b1, wx0, wx1, wy0, wy1, drow + 12 * h, store_any & (shift < 4 * h + 4), max(shift - 4 * h, 0));
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t off[FP_PX];
                bool near_end = false;
#pragma unroll
                for (int j = 0; j < FP_PX; ++j) {
                    const int q = 4 * h + j;
                    const unsigned long long ubx = ((unsigned long long)hx[q] << 32) | lx[q], uby = ((unsigned long long)hy[q] << 32) | ly[q];
                    const bool valid = (ubx >= MAGIC_BITS) & (ubx <= a.xmax_bits) & (uby >= MAGIC_BITS) & (uby <= a.ymax_bits);
                    const float sc = valid ? W_SCALE : 0.f;
                    const int ix = (int)(hx[q] - MAGIC_HI), iy = (int)(hy[q] - MAGIC_HI);
                    wx1[j] = (float)lx[q]; wx0[j] = (float)(~lx[q]);
                    wy1[j] = (float)ly[q] * sc; wy0[j] = (float)(~ly[q]) * sc;
                    off[j] = valid ? (uint32_t)iy * pitch + (uint32_t)ix * 3u : 0u;
                    near_end |= valid & (iy > a.src_h - 3);
                }
                if (!__any(near_end)) {
#pragma unroll
                    for (int j = 0; j < FP_PX; ++j) {
                        const pk2 r0 = ld8(simg + off[j]);
                        const pk2 r1 = ld8(simg + off[j] + pitch);
                        a0[j] = r0.a; b0[j] = __builtin_amdgcn_alignbyte(r0.b, r0.a, 3);
                        a1[j] = r1.a; b1[j] = __builtin_amdgcn_alignbyte(r1.b, r1.a, 3);
                    }
                } else {
                    const uint32_t last = (uint32_t)a.src_h * pitch - 3u;
#pragma unroll
                    for (int j = 0; j < FP_PX; ++j) {
                        const uint32_t o00 = off[j];
                        const uint32_t o01 = min(o00 + 3u, last), o10 = min(o00 + pitch, last), o11 = min(o00 + pitch + 3u, last);
                        a0[j] = simg[o00] | (simg[o00 + 1] << 8) | (simg[o00 + 2] << 16);
                        b0[j] = simg[o01] | (simg[o01 + 1] << 8) | (simg[o01 + 2] << 16);
                        a1[j] = simg[o10] | (simg[o10 + 1] << 8) | (simg[o10 + 2] << 16);
                        b1[j] = simg[o11] | (simg[o11 + 1] << 8) | (simg[o11 + 2] << 16);
                    }
                }
                blend_store<DstT>(a0, b0, a1, b1, wx0, wx1, wy0, wy1, drow + 12 * h, store_any & (shift < 4 * h + 4), max(shift - 4 * h, 0));
            }
        }

        // ---- [D] next patch: the other six pixels ---------------------------------------------------------------------
        if (more) { finish_coords(); cur = nxt; }
    }
}


}  // namespace rwh
