// K3 fast path: RGB u8 source, bilinear, u8 (truncated) or f32 output -- the BASELINE configuration.
//
// What bounds this kernel on MI355X (measured): not HBM.  A direct gather
// version (one lane = 4 output pixels, two unaligned 8-byte loads per pixel) is bound first by the
// texture-address unit (a 64-lane gather costs ~16 cycles per dword per lane-quad, i.e. one TA cycle
// per pixel per CU) and, once that is fixed, by VALU issue: every VALU instruction, float32 or
// float64, costs ~4.2 cycles per wave on gfx950 and only v_pk_*_f32 does two lanes' worth per issue.
// The final 8-pixel-per-lane kernel issues 440 VALU instructions per wave (55 per pixel) and keeps the
// VALU ~85-90 % busy.  So the design goal is "few instructions per pixel", and loads that are wide and coalesced:
//
//   * X, Y, W of a pixel are affine in (row, column): lane setup is two float64 FMAs per coordinate
//     from host-precomputed coefficients, every further pixel is one add each;
//   * few reciprocals: a run of pixels shares ONE v_rcp_f64 (Montgomery batch inversion);
//   * floor / fraction / weights come from the float64 "magic number" u = s + 1.5*2^20 (ulp 2^-32):
//     hi(u) - 0x41380000 = floor(s), lo(u) = frac(s) * 2^32; w = cvt(lo) and 2^32 - w are both
//     correct to float32 rounding; the 2^-64 scale of (x weight)*(y weight) rides on the y weights;
//   * the patch's source footprint is the bounding box of its four mapped corners (extremes of a
//     projective map over a rectangle sit at the corners while W keeps its sign); if it lies strictly
//     inside the image the wave needs NO per-pixel bounds logic, and it copies the footprint into a
//     wave-private LDS slab with coalesced 12-byte loads, expanding RGB to 4-byte RGBX texels so that the
//     bilinear taps are 4-byte aligned ds_read2_b32 pairs (8-byte LDS reads at 4-byte alignment stall as
//     "unaligned");
//   * waves that touch the border, cross the horizon, or have a footprint that does not fit the slab
//     (strong zoom-out) take a masked gather path straight from global memory, with a byte-exact guard for
//     the last source rows so nothing is read past the image;
//   * the blend is the 4-weight form on packed float32 pairs (pixel j | pixel j+1);
//   * u8 output: v_cvt_pk_u8_f32 converts + packs a byte per instruction; it rounds to nearest, so the
//     accumulator starts at -0.5 + 2^-15: exact integers (weights 0/1) land on themselves, anything else
//     is floor(v + 3e-5), inside the float32 blend's own noise.
// Two kernels: warp_rgb8_fast (4 px per lane, 64 x 4 patches, fixed 9 x 84-texel slab; outputs narrower than
// 128 px) and warp_rgb8_fast8 (8 px per lane, three patch shapes, slab capacity as an area; everything else).
// No MFMA: there is no dense contraction on this path.
#pragma once
#include "rwh_common.h"
#include <type_traits>

namespace rwh {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr double MAGIC = 1572864.0;                   // 1.5 * 2^20: ulp(MAGIC + s) = 2^-32 for |s| < 2^19
constexpr uint32_t MAGIC_HI = 0x41380000u;            // high dword of MAGIC
constexpr unsigned long long MAGIC_BITS = 0x4138000000000000ull;
constexpr float W_SCALE = 5.42101086242752217e-20f;   // 2^-64
constexpr float U8_BIAS = -0.5f + 3.0517578125e-05f;  // -0.5 + 2^-15, see above

constexpr int FP_PX = 4;                               // pixels per lane
constexpr int FP_ROWS = 9;                             // source rows a wave can stage (3 per staging step)
constexpr int FP_LANES = 21;                           // staging lanes per source row, 4 texels (12 B) each
constexpr int FP_TEXELS = 4 * FP_LANES;                // 84 texels per staged row
constexpr int FP_PITCH = 4 * FP_TEXELS + 16;           // 352 B per staged row of RGBX texels (+16: bank stagger)

// Per-homography coefficients.  One set travels inside FastArgs; launches with one homography per image carry a table
// of them as a second kernel argument (CoefTab) and index it with the image number.
struct Coef {
    // X = cx[0] + row*cx[1] + col*cx[2] for output (row, col); likewise Y, W   (float64, host-built)
    double cx[3], cy[3], cw[3];
    double dxs8[7][3];                                   // 8 px kernel: column offsets of a lane's pixels x (cx[2], cy[2], cw[2])
    double ih[9];                                        // inv(H), row major (nearest-neighbour kernel's exact formula)
    long long image;                                     // table launches: index of this entry's image in the batch
};
constexpr int TAB_N = 8;                                 // images per launch when every image has its own homography
struct CoefTab { Coef e[TAB_N]; };

struct FastArgs {
    const unsigned char* src;
    unsigned char* dst;
    long long src_img_stride, dst_img_stride;            // bytes
    Coef c;
    double dxs[3][3];                                    // 4 px kernel: dxs[j-1] = j * (cx[2], cy[2], cw[2])
    unsigned long long xmax_bits, ymax_bits;             // bit patterns of MAGIC + (bound_w-1), MAGIC + (bound_h-1)
    int src_h, src_w, bound_h, bound_w, out_w;          // out_w: the columns this launch's tiles cover (its first columns of the row)
    int pitch_w;                                         // pixels per output row in memory (round 4: a thin ragged right edge is left to
                                                         //  warp_rgb8_strip, so the tiled launch may cover fewer columns than the row has)
    int row_begin, rows;                                 // produce output rows [row_begin, row_begin+rows)
    unsigned tiles_x, tiles_y, nblocks, cpx;
    unsigned tiles_x_magic, tiles_y_magic;               // floor(n/d) = umulhi(n, magic) for n < nblocks
    int group;                                           // free parameter of a tools/warp_lab custom kernel
    double gx0, gstep_x, gx_last, gy0, gstep_y, gy_last; // numpy.linspace output grid (nearest-neighbour kernel's exact formula)
    int out_h;
    // multi-frame kernels (warp_rgb8_fast8m, round 4): one block = one 128 x 16 tile of mf_frames consecutive frames
    int mf_frames, mf_batch;                             // frames per block, frames in the launch
    unsigned mf_nblocks, mf_cpx, ntiles, ntiles_magic;   // blocks of the multi-frame grid (tile fastest), tiles per frame
};

__device__ __forceinline__ uint32_t hi32(double v) { return (uint32_t)(__double_as_longlong(v) >> 32); }
__device__ __forceinline__ uint32_t lo32(double v) { return (uint32_t)__double_as_longlong(v); }
__device__ __forceinline__ float ubyte(uint32_t v, int byte) { return (float)((v >> (8 * byte)) & 0xffu); }

// 24-bit multiplies spelled in assembly: LLVM rewrites __umul24 of provably small operands into the quarter-rate
// 32-bit v_mul_lo_u32.
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t mul24_s(uint32_t a, uint32_t b_uniform) {      // b in an SGPR
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b_uniform));
    return r;
}
__device__ __forceinline__ uint32_t mad24_s(uint32_t a, uint32_t b_uniform, uint32_t c) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t shl2_add_s(uint32_t a, uint32_t c_uniform) {    // (a << 2) + c, one instruction
    uint32_t r;                                                                      // (LLVM spells hi32(double) << 2 as
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(r) : "v"(a), "s"(c_uniform));          //  alignbit + and + add otherwise)
    return r;
}
// min / max of wave-uniform integers on the scalar unit (left to itself, hipcc reduces the four footprint corners with
// v_min3 / v_max3 plus the v_mov copies their operands need: a dozen VALU issues per wave)
__device__ __forceinline__ int smin(int a, int b) { int r; asm("s_min_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; }
__device__ __forceinline__ int smax(int a, int b) { int r; asm("s_max_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; }
__device__ __forceinline__ uint32_t mul24_12(uint32_t a) {                          // inline constant
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, 12" : "=v"(r) : "v"(a));
    return r;
}

constexpr float W_ONE = 2.3283064365386963e-10f;      // 2^-32 = the y weights' "1.0" after the 2^-64 scale

// Bilinear weights of one pixel from the low dwords of its magic-number coordinates (fraction * 2^32):
// wx1 = frac_x * 2^32, wx0 = 2^32 - wx1 (Sterbenz-exact or rounded once, like a convert of the complement would be);
// wy1 = frac_y * 2^-32, wy0 = 2^-32 - wy1 in one fused step.  sc / one = (2^-64, 2^-32), or (0, 0) for a masked pixel.
__device__ __forceinline__ void weights(uint32_t lx, uint32_t ly, float sc, float one, float& wx0, float& wx1, float& wy0, float& wy1) {
    wx1 = (float)lx;
    wx0 = 4294967296.0f - wx1;
    const float fy = (float)ly;
    wy1 = fy * sc;
    wy0 = __builtin_fmaf(-fy, sc, one);
}

// The 8 px kernels' form (blend4<..., FOLDED = true>): only wx1 = frac_x * 2^32 and wy1 = frac_y * 2^-32 are computed
// per pixel; the slots wx0 / wy0 carry the pixel's "one" (2^-32, or 0 for a masked pixel) and "one * 2^32" (1 or 0), from
// which blend4 derives the four tap weights with one multiply, three fused multiply-adds and one subtraction per pixel
// pair: w11 = fx fy, w01 = fx - w11, w10 = fy - w11, w00 = (1 - fx) - w10  (5.5 instead of 7 VALU instructions per pixel).
__device__ __forceinline__ void weights2(uint32_t lx, uint32_t ly, float sc, float one, float c1, float& wx0, float& wx1, float& wy0, float& wy1) {
    wx1 = (float)lx;
    wy1 = (float)ly * sc;
    wx0 = one;
    wy0 = c1;
}

// Blend 4 pixels (taps a0 = texel(iy,ix), b0 = texel(iy,ix+1), a1/b1 = row iy+1; R,G,B in bytes 0..2)
// and store them.  `full`: the lane owns all 4 pixels; otherwise it owns local pixels j >= shift.
// PSTR > 1 (float32 output of the 8 px kernel): the four pixels are PSTR columns apart, pixel j is stored (12 bytes)
// at drow + 3*j*PSTR if j*PSTR >= shift -- neighbouring lanes then write neighbouring pixels in every store
// instruction (four consecutive float32 pixels per lane = 48-byte lane stride cost 2.7x the time).
// `xpose` (float32 output, PSTR > 1 only; wave-uniform, or nullptr): 48*PSTR bytes of LDS per lane group through which
// the group's 4*PSTR pixels are re-dealt so that lane l stores the 16-byte pieces l, PSTR + l, 2*PSTR + l of the
// group's contiguous 48*PSTR-byte row segment: three fully coalesced dwordx4 stores instead of four 12-byte ones
// (the texture-address path, 87 % busy on this variant, charges a dwordx3 like a dwordx4).
// ---- blend without byte -> float converts (round 3) -------------------------------------------------------------------
// A byte zero-extended to 16 bits IS the float16 denormal b * 2^-24, and v_fma_mix_f32 takes a float16 operand from either
// half of a VGPR (denormals honoured: tools/energy_probe.hip's check).  With the four tap weights carrying the 2^24
// (MIX_S) the products are the same real numbers as cvt(b) * w, the fused multiply-adds round the same sums in the same
// order: results are bit-identical to the convert + v_pk_fma_f32 form, for 16 instead of 18 VALU instructions per pixel
// and 18 instead of 22 nJ (one v_perm per tap puts R and G into the halves of a dword; B sits in the upper half of the
// RGBX texel itself, whose X byte the staging code therefore leaves zero).
constexpr float MIX_S = 16777216.0f;                  // 2^24
#ifdef RWH_ABL_NOMIX   // tools/ A/B hook (never defined in the product build): the convert + v_pk_fma_f32 blend of rounds 1-2
constexpr bool MIX_RGB = false;
#else
constexpr bool MIX_RGB = true;
#endif
__device__ __forceinline__ float fmix_lo(uint32_t h, float w, float acc) {
    float d; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(w), "v"(acc)); return d;
}
__device__ __forceinline__ float fmix_hi(uint32_t h, float w, float acc) {
    float d; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(w), "v"(acc)); return d;
}
__device__ __forceinline__ float fmix_lo_c(uint32_t h, float w, float acc_uniform) {     // first term: accumulator = a constant
    float d; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(w), "s"(acc_uniform)); return d;
}
__device__ __forceinline__ float fmix_hi_c(uint32_t h, float w, float acc_uniform) {
    float d; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(w), "s"(acc_uniform)); return d;
}
// RGBX texel (X == 0) -> R | G << 16
__device__ __forceinline__ uint32_t rg_halves(uint32_t t) { return __builtin_amdgcn_perm(0u, t, 0x0C010C00u); }

// float32 output, whole-run stores (uniform: no lane of the wave straddles a ragged row): a run's 4 * PSTR pixels per lane group --
// pixel j of lane l is column l + j * PSTR of the group's row segment -- are re-dealt through `region` (48 * PSTR bytes of LDS per group)
// so that lane l stores the 16-byte pieces l, PSTR + l, 2 * PSTR + l of the contiguous 48 * PSTR-byte segment: three fully coalesced
// dwordx4 stores instead of four 12-byte ones (the texture-address path charges a dwordx3 like a dwordx4 and dislikes strided pieces).
template <int PSTR>
__device__ __forceinline__ void redeal_store_f32(const float (&o)[FP_PX][3], float* drow, bool store_any, unsigned char* region, int l) {
#pragma unroll
    for (int j = 0; j < FP_PX; ++j) {
        const pk3 w = {__float_as_uint(o[j][0]), __float_as_uint(o[j][1]), __float_as_uint(o[j][2])};
        __builtin_memcpy(region + 12 * (l + j * PSTR), &w, 12);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    unsigned char* seg = reinterpret_cast<unsigned char*>(drow) - 12 * l;   // the group's row segment
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        const uint4 piece = *reinterpret_cast<const uint4*>(region + 16 * (v * PSTR + l));
        if (store_any) __builtin_memcpy(seg + 16 * (v * PSTR + l), &piece, 16);
    }
    // the next run's pieces go to the same region: its LDS writes queue behind these reads (in-order LDS)
    __builtin_amdgcn_wave_barrier();
}

template <bool U8, int CH = 3, bool FOLDED = false>
__device__ __forceinline__ void blend4(const uint32_t (&a0)[FP_PX], const uint32_t (&b0)[FP_PX],
                                       const uint32_t (&a1)[FP_PX], const uint32_t (&b1)[FP_PX],
                                       const float (&wx0)[FP_PX], const float (&wx1)[FP_PX],
                                       const float (&wy0)[FP_PX], const float (&wy1)[FP_PX], float (&o)[FP_PX][CH]) {
    constexpr float BIAS = U8 ? U8_BIAS : 0.f;
#pragma unroll
    for (int j = 0; j < FP_PX; j += 2) {
        const f2 WX0 = {wx0[j], wx0[j + 1]}, WX1 = {wx1[j], wx1[j + 1]};
        const f2 WY0 = {wy0[j], wy0[j + 1]}, WY1 = {wy1[j], wy1[j + 1]};
        f2 W00, W01, W10, W11;
        if constexpr (FOLDED) {              // WX0 = "one", WY0 = "one * 2^32" (see weights2)
            W11 = WX1 * WY1;
            W01 = __builtin_elementwise_fma(WX1, WX0, -W11);
            W10 = __builtin_elementwise_fma(WY1, f2{4294967296.0f, 4294967296.0f}, -W11);
            W00 = __builtin_elementwise_fma(-WX1, WX0, WY0) - W10;
            if constexpr (CH == 3 && MIX_RGB) {         // the weights carry MIX_S (the callers' constants): taps enter as float16 halves
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int p = j + q;
                    const float w00 = q ? W00.y : W00.x, w01 = q ? W01.y : W01.x, w10 = q ? W10.y : W10.x, w11 = q ? W11.y : W11.x;
                    const uint32_t g00 = rg_halves(a0[p]), g01 = rg_halves(b0[p]), g10 = rg_halves(a1[p]), g11 = rg_halves(b1[p]);
                    o[p][0] = fmix_lo(g11, w11, fmix_lo(g10, w10, fmix_lo(g01, w01, fmix_lo_c(g00, w00, BIAS))));
                    o[p][1] = fmix_hi(g11, w11, fmix_hi(g10, w10, fmix_hi(g01, w01, fmix_hi_c(g00, w00, BIAS))));
                    o[p][2] = fmix_hi(b1[p], w11, fmix_hi(a1[p], w10, fmix_hi(b0[p], w01, fmix_hi_c(a0[p], w00, BIAS))));
                }
                continue;
            }
        } else {
            W00 = WX0 * WY0; W01 = WX1 * WY0; W10 = WX0 * WY1; W11 = WX1 * WY1;
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const f2 P00 = {ubyte(a0[j], k), ubyte(a0[j + 1], k)}, P01 = {ubyte(b0[j], k), ubyte(b0[j + 1], k)};
            const f2 P10 = {ubyte(a1[j], k), ubyte(a1[j + 1], k)}, P11 = {ubyte(b1[j], k), ubyte(b1[j + 1], k)};
            f2 acc = __builtin_elementwise_fma(P00, W00, f2{BIAS, BIAS});
            acc = __builtin_elementwise_fma(P01, W01, acc);
            acc = __builtin_elementwise_fma(P10, W10, acc);
            acc = __builtin_elementwise_fma(P11, W11, acc);
            o[j][k] = acc.x; o[j + 1][k] = acc.y;
        }
    }
}

template <typename DstT, int PSTR = 1, int CH = 3, bool FOLDED = false>
__device__ __forceinline__ void blend_store(const uint32_t (&a0)[FP_PX], const uint32_t (&b0)[FP_PX],
                                            const uint32_t (&a1)[FP_PX], const uint32_t (&b1)[FP_PX],
                                            const float (&wx0)[FP_PX], const float (&wx1)[FP_PX],
                                            const float (&wy0)[FP_PX], const float (&wy1)[FP_PX],
                                            DstT* drow, bool store_any, int shift) {
    constexpr bool U8 = sizeof(DstT) == 1;
    if constexpr (CH == 4) {                                  // RGBA uint8: a pixel is one dword, a run 16 bytes
        static_assert(U8 && PSTR == 1, "the 4-channel form is uint8 in, uint8 out");
        float o4[FP_PX][4];
        blend4<true, 4, FOLDED>(a0, b0, a1, b1, wx0, wx1, wy0, wy1, o4);
        if (!store_any) return;
        uint32_t px[FP_PX];
#pragma unroll
        for (int j = 0; j < FP_PX; ++j) {
            uint32_t q = 0;
            q = __builtin_amdgcn_cvt_pk_u8_f32(o4[j][0], 0, q); q = __builtin_amdgcn_cvt_pk_u8_f32(o4[j][1], 1, q);
            q = __builtin_amdgcn_cvt_pk_u8_f32(o4[j][2], 2, q); q = __builtin_amdgcn_cvt_pk_u8_f32(o4[j][3], 3, q);
            px[j] = q;
        }
        // (an RGBA pixel is 4-byte aligned: said to the compiler, which splits a 16-byte copy to a byte pointer into four
        //  dword stores -- 6.9 store instructions per wave instead of 2)
        uint32_t* d32 = reinterpret_cast<uint32_t*>(__builtin_assume_aligned(drow, 4));
        if (!__any(shift != 0)) {
            // the whole wave stores whole runs (every tile but a ragged row's last): ONE dwordx4 store, spelled in assembly --
            // left to itself the compiler folds this branch into the per-pixel predicated stores below (6.9 store
            // instructions per wave instead of 2: the kernel's time follows them)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 w = {px[0], px[1], px[2], px[3]};
            // s_nop: a VMEM store of more than 8 bytes reads its data registers over two cycles, and the next VALU
            // instruction must not overwrite them in that window -- the compiler pads its own stores, not an asm statement's
            // (without it, the first two pixels of a run came out as whatever the next instruction wrote: tools/soak_warp.py CH=4)
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 2" : : "v"(d32), "v"(w) : "memory");
        } else {
#pragma unroll
            for (int j = 0; j < FP_PX; ++j)
                if (j >= shift) d32[j] = px[j];
        }
        return;
    }
    float o[FP_PX][3];
    blend4<U8, 3, FOLDED>(a0, b0, a1, b1, wx0, wx1, wy0, wy1, o);
    if (!store_any) return;
#ifdef RWH_ABL_NOSTORE  // tools/warp_lab ablation hook (never defined in the product build)
    if (o[0][0] + o[1][1] + o[2][2] + o[3][0] != -12345.f) return;
#endif
    if constexpr (PSTR > 1) {
        static_assert(!U8, "strided runs are the float32 layout");
#pragma unroll
        for (int j = 0; j < FP_PX; ++j)
            if (j * PSTR >= shift) {
                const pk3 w = {__float_as_uint(o[j][0]), __float_as_uint(o[j][1]), __float_as_uint(o[j][2])};
                __builtin_memcpy(drow + 3 * j * PSTR, &w, 12);
            }
        return;
    }
    if (shift == 0) {
        if constexpr (U8) {
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
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const float* f = &o[0][0] + 4 * v;
                pk4 w = {__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
                __builtin_memcpy(reinterpret_cast<unsigned char*>(drow) + 16 * v, &w, 16);
            }
        }
    } else {  // the one straddling lane of a ragged row
#pragma unroll
        for (int j = 1; j < FP_PX; ++j)
            if (j >= shift) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if constexpr (U8) drow[3 * j + k] = (unsigned char)__builtin_amdgcn_cvt_pk_u8_f32(o[j][k], 0, 0);
                    else drow[3 * j + k] = o[j][k];
                }
            }
    }
}

// A run of 4 pixels that all map outside the source: zeros, in blend_store's store layout (`shift` as there).
template <typename DstT, int PSTR, int CH>
__device__ __forceinline__ void zero_store(DstT* drow, bool store_any, int shift) {
    if (!store_any) return;
    if constexpr (PSTR > 1) {                                 // float32 output of the 8 px kernel: pixels PSTR columns apart
#pragma unroll
        for (int j = 0; j < FP_PX; ++j)
            if (j * PSTR >= shift) { const pk3 z = {0u, 0u, 0u}; __builtin_memcpy(drow + 3 * j * PSTR, &z, 12); }
    } else if (shift == 0) {
        if constexpr (CH * sizeof(DstT) == 3) { const pk3 z = {0u, 0u, 0u}; __builtin_memcpy(drow, &z, 12); }
        else if constexpr (CH * sizeof(DstT) == 4) {          // RGBA uint8: one aligned 16-byte store
            uint32_t* d32 = reinterpret_cast<uint32_t*>(__builtin_assume_aligned(drow, 4));
#pragma unroll
            for (int j = 0; j < FP_PX; ++j) d32[j] = 0u;
        } else {
#pragma unroll
            for (int v = 0; v < 3; ++v) { const pk4 z = {0u, 0u, 0u, 0u}; __builtin_memcpy(reinterpret_cast<unsigned char*>(drow) + 16 * v, &z, 16); }
        }
    } else {
#pragma unroll
        for (int j = 1; j < FP_PX; ++j)
            if (j >= shift) {
#pragma unroll
                for (int k = 0; k < CH; ++k) drow[CH * j + k] = (DstT)0;
            }
    }
}

#ifndef RWH_F4_WAVES
#define RWH_F4_WAVES 8   // fits in 62 VGPRs without spilling: 8 waves per SIMD
#endif
template <typename DstT>
__global__ __launch_bounds__(256, RWH_F4_WAVES) void warp_rgb8_fast(const FastArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char slab[4][FP_ROWS * FP_PITCH];

    // ---- block / wave -> patch (all scalar) ---------------------------------------------------------
    const unsigned b = blockIdx.x;
    const unsigned logical = (b & 7u) * a.cpx + (b >> 3);   // XCD k walks logical blocks [k*cpx, (k+1)*cpx)
    if (logical >= a.nblocks) return;
    const unsigned t = a.tiles_x_magic ? __umulhi(logical, a.tiles_x_magic) : logical;   // magic 0 <=> divisor 1
    const unsigned tx = logical - t * a.tiles_x;
    const unsigned img = a.tiles_y_magic ? __umulhi(t, a.tiles_y_magic) : t;
    const unsigned ty = t - img * a.tiles_y;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int prow = lane >> 4, pq = lane & 15;               // patch row 0..3, 4-pixel column group 0..15

    const int rr_raw = (int)ty * 4 + prow;
    const int rr = min(rr_raw, a.rows - 1);                   // rows past the shard recompute its last row
    const int c0 = ((int)tx * 4 + wave) * 64 + pq * FP_PX;
    const int c0p = min(c0, a.out_w - FP_PX);                 // columns past the row end recompute its last 4 px
    const int shift = c0 - c0p;
    const bool store_any = (rr_raw < a.rows) & (c0 < a.out_w);

    const unsigned char* simg = a.src + (long long)img * a.src_img_stride;       // uniform
    unsigned char* dimg = a.dst + (long long)img * a.dst_img_stride;              // uniform
    // 32-bit lane offset from a uniform base (host guarantees rows*out_w*3*sizeof(DstT) < 2^32)
    DstT* drow = reinterpret_cast<DstT*>(dimg + ((uint32_t)rr * (uint32_t)a.pitch_w + (uint32_t)c0p) * (uint32_t)(3 * sizeof(DstT)));

    // ---- source coordinates of the lane's 4 pixels -----------------------------------------------
    const double fr = (double)(a.row_begin + rr), fc = (double)c0p;
    double X[FP_PX], Y[FP_PX], W[FP_PX];
    X[0] = fma(fc, a.c.cx[2], fma(fr, a.c.cx[1], a.c.cx[0]));
    Y[0] = fma(fc, a.c.cy[2], fma(fr, a.c.cy[1], a.c.cy[0]));
    W[0] = fma(fc, a.c.cw[2], fma(fr, a.c.cw[1], a.c.cw[0]));
#pragma unroll
    for (int j = 1; j < FP_PX; ++j) {
        X[j] = X[0] + a.dxs[j - 1][0];
        Y[j] = Y[0] + a.dxs[j - 1][1];
        W[j] = W[0] + a.dxs[j - 1][2];
    }
    double rc[FP_PX];
    // W is affine along a row, so W[0] > 0 and W[3] > 0 make all four positive; with a finite, normal product
    // the four reciprocals come from ONE v_rcp_f64 (Montgomery batch inversion).  `wpos` is wave-uniform.
    const double p01 = W[0] * W[1], p23 = W[2] * W[3], P = p01 * p23;
    const bool wpos = __all((int)__builtin_amdgcn_class(P, 0x100) & (int)((int)hi32(W[0]) > 0) & (int)((int)hi32(W[FP_PX - 1]) > 0));
    if (wpos) {
        double rp = __builtin_amdgcn_rcp(P);
        rp = fma(fma(-P, rp, 1.0), rp, rp);
        const double r01 = rp * p23, r23 = rp * p01;
        rc[0] = r01 * W[1]; rc[1] = r01 * W[0]; rc[2] = r23 * W[3]; rc[3] = r23 * W[2];
    } else {                                                      // a W at / across zero (the horizon): pixel by pixel
#pragma unroll
        for (int j = 0; j < FP_PX; ++j) {
            double q = __builtin_amdgcn_rcp(W[j]);
            rc[j] = fma(fma(-W[j], q, 1.0), q, q);
        }
    }
    uint32_t lx[FP_PX], ly[FP_PX], hx[FP_PX], hy[FP_PX];
    unsigned long long ubx[FP_PX], uby[FP_PX];
#pragma unroll
    for (int j = 0; j < FP_PX; ++j) {
        const double ux = fma(X[j], rc[j], MAGIC);   // one rounding, straight onto the 2^-32 grid
        const double uy = fma(Y[j], rc[j], MAGIC);
        ubx[j] = (unsigned long long)__double_as_longlong(ux);
        uby[j] = (unsigned long long)__double_as_longlong(uy);
        lx[j] = lo32(ux); ly[j] = lo32(uy);
        hx[j] = hi32(ux); hy[j] = hi32(uy);
    }
    const uint32_t pitch = (uint32_t)a.src_w * 3u;

    // ---- wave-uniform footprint from the four patch corners (lanes 0, 15, 48, 63), on the scalar unit ----
    // hi dwords compare like the integers they encode (same exponent); out-of-range / NaN corners end up
    // as the min or the max and fail the range test below.
    const int x0 = (int)__builtin_amdgcn_readlane(hx[0], 0), x1 = (int)__builtin_amdgcn_readlane(hx[FP_PX - 1], 15);
    const int x2 = (int)__builtin_amdgcn_readlane(hx[0], 48), x3 = (int)__builtin_amdgcn_readlane(hx[FP_PX - 1], 63);
    const int y0 = (int)__builtin_amdgcn_readlane(hy[0], 0), y1 = (int)__builtin_amdgcn_readlane(hy[FP_PX - 1], 15);
    const int y2 = (int)__builtin_amdgcn_readlane(hy[0], 48), y3 = (int)__builtin_amdgcn_readlane(hy[FP_PX - 1], 63);
    const int hxmn = smin(smin(x0, x1), smin(x2, x3)), hxmx = smax(smax(x0, x1), smax(x2, x3));
    const int hymn = smin(smin(y0, y1), smin(y2, y3)), hymx = smax(smax(y0, y1), smax(y2, y3));
    const int xmn = (int)((uint32_t)smax(hxmn, 0) - MAGIC_HI), xmx = (int)((uint32_t)smax(hxmx, 0) - MAGIC_HI);   // (negative hi dwords:
    const int ymn = (int)((uint32_t)smax(hymn, 0) - MAGIC_HI), ymx = (int)((uint32_t)smax(hymx, 0) - MAGIC_HI);   //  see fast8_body)
    // strictly inside: 0 <= floor(s) <= bound-2 on both axes, tap rows above the last source row,
    // and the footprint (rows ymn..ymx+1, texels xmn..xmx+1) fits the slab
    const bool staged = wpos & (xmn >= 0) & (xmx < a.bound_w - 1) & (ymn >= 0) &
                        (ymx < min(a.bound_h - 1, a.src_h - 2)) & (ymx - ymn + 2 <= FP_ROWS) & (xmx - xmn + 2 <= FP_TEXELS);

    uint32_t a0[FP_PX], b0[FP_PX], a1[FP_PX], b1[FP_PX];
    float wx0[FP_PX], wx1[FP_PX], wy0[FP_PX], wy1[FP_PX];
    if (staged) {
        const int nrows = ymx - ymn + 2, ntex = xmx - xmn + 2;
        unsigned char* my = slab[wave];
        // staging lane -> (source row srow of 3, texel group scol of 21): 12 packed bytes in, 4 RGBX texels out
        const int srow = (lane * 49) >> 10;                     // lane / 21 for lane < 64
        const int scol = lane - FP_LANES * srow;
        const unsigned char* gbase = simg + (size_t)((uint32_t)ymn * pitch + (uint32_t)xmn * 3u);   // uniform
        const uint32_t goff = (uint32_t)srow * pitch + (uint32_t)scol * 12u;
        unsigned char* wlds = my + srow * FP_PITCH + scol * 16;
        const bool mine = (srow < 3) & (4 * scol < ntex);
        // all staging loads are issued before the first LDS write so that their latencies overlap
        pk3 v[FP_ROWS / 3];
        bool on[FP_ROWS / 3];
#pragma unroll
        for (int k = 0; k < FP_ROWS / 3; ++k) {
            on[k] = mine & (3 * k + srow < nrows);
            v[k] = pk3{0u, 0u, 0u};
            if (on[k]) __builtin_memcpy(&v[k], gbase + (size_t)(3 * k) * pitch + goff, 12);
        }
        // weights and LDS addresses while the loads fly
        const uint32_t lds_c = (uint32_t)hymn * (uint32_t)FP_PITCH + (uint32_t)hxmn * 4u;  // uniform
        uint32_t lo[FP_PX];
#pragma unroll
        for (int j = 0; j < FP_PX; ++j) {
            wx1[j] = (float)lx[j]; wx0[j] = (float)(~lx[j]);
            wy1[j] = (float)ly[j] * W_SCALE; wy0[j] = (float)(~ly[j]) * W_SCALE;
            lo[j] = hy[j] * (uint32_t)FP_PITCH + hx[j] * 4u - lds_c;
        }
#pragma unroll
        for (int k = 0; k < FP_ROWS / 3; ++k) {
            if (on[k]) {
                uint4 t4;
                t4.x = v[k].a;
                t4.y = __builtin_amdgcn_alignbyte(v[k].b, v[k].a, 3);
                t4.z = __builtin_amdgcn_alignbyte(v[k].c, v[k].b, 2);
                t4.w = v[k].c >> 8;
                *reinterpret_cast<uint4*>(wlds + 3 * k * FP_PITCH) = t4;
            }
        }
        // the slab is wave-private: order this wave's LDS writes before its LDS reads, no block barrier
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < FP_PX; ++j) {
            // two dwords at a 4-byte aligned address: ds_read2_b32 (a ds_read_b64 here is "unaligned" and stalls)
            const uint32_t* t0 = reinterpret_cast<const uint32_t*>(my + lo[j]);
            a0[j] = t0[0]; b0[j] = t0[1]; a1[j] = t0[FP_PITCH / 4]; b1[j] = t0[FP_PITCH / 4 + 1];
        }
        blend_store<DstT>(a0, b0, a1, b1, wx0, wx1, wy0, wy1, drow, store_any, shift);
        return;
    }

    // ---- border / oversize waves: masked gathers from global memory -----------------------------------
    uint32_t off[FP_PX];
    bool near_end = false;
#pragma unroll
    for (int j = 0; j < FP_PX; ++j) {
        // 0 <= s <= bound-1 on the bit patterns: positive doubles order like unsigned integers; negative
        // values and NaNs have patterns outside [MAGIC_BITS, xmax_bits]
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
        for (int j = 0; j < FP_PX; ++j) {
            const pk2 r0 = ld8(simg + off[j]);
            const pk2 r1 = ld8(simg + off[j] + pitch);
            a0[j] = r0.a; b0[j] = __builtin_amdgcn_alignbyte(r0.b, r0.a, 3);
            a1[j] = r1.a; b1[j] = __builtin_amdgcn_alignbyte(r1.b, r1.a, 3);
        }
    } else {  // byte-exact loads, +1 taps clamped to the image (their weight is 0 when clamped)
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
    blend_store<DstT>(a0, b0, a1, b1, wx0, wx1, wy0, wy1, drow, store_any, shift);
}


// ---- canvas compositor (stitchPanorama's paste / 'Rate' blend as the warp's epilogue) ---------------
// The launch's output grid IS the canvas (homography.py:303-321): the warped image imgT occupies the rectangle T, imgQ the
// rectangle Q.  Paste (homography.py:335-338): Q over warped T over 0.  'Rate' (322-334): inside T the alpha-weighted mean
// (qa * Q + ta * warped) / (qa + ta) with qa = 1 + 1e-10 - rate inside Q, 1e-10 outside (Q's colour = 0 there) and ta = the
// warped alpha plane = rate + 1e-10 wherever the source coordinate is inside imgT, 0 where it is masked (then the result is
// Q's pixel itself); outside T the canvas keeps Q / 0.  The fast form takes ta = its nominal value (the reference's float64
// lerp of a constant plane is that value to 1e-16; the one output pixel whose taps include the blanked alpha texel (0,0)
// differs) and blends in float32: within 1 LSB of the reference's uint8 canvas.
struct CompArgs {
    const unsigned char* q;              // imgQ, q_h x q_w x 3 uint8
    int q_w, q_h, qsx, qsy;              // Q's rectangle on the canvas
    int tsx, tsy, wt, ht;                // T's rectangle on the canvas (the bounding box wrapPerspective warps into)
    int mode;                            // 1 = paste, 2 = 'Rate' blend
    float wq_in, wt_in, wq_out, wt_out;  // qa / (qa + ta), ta / (qa + ta) inside / outside Q
};

// imgQ's pixels under a run of 4 canvas pixels (row cy, columns cx .. cx+3): 24-bit RGB each, and which of them lie in Q
__device__ __forceinline__ unsigned comp_load_q(const CompArgs& c, int cy, int cx, uint32_t (&q)[FP_PX]) {
    const int qy = cy - c.qsy, qx = cx - c.qsx;
    unsigned in = 0;
#pragma unroll
    for (int j = 0; j < FP_PX; ++j) { q[j] = 0u; in |= (unsigned)((qy >= 0) & (qy < c.q_h) & (qx + j >= 0) & (qx + j < c.q_w)) << j; }
    if (in == 15u) {
        const unsigned char* p = c.q + ((size_t)qy * (size_t)c.q_w + (size_t)qx) * 3u;
        pk3 w;
        __builtin_memcpy(&w, p, 12);
        q[0] = w.a & 0xFFFFFFu; q[1] = __builtin_amdgcn_alignbyte(w.b, w.a, 3) & 0xFFFFFFu;
        q[2] = __builtin_amdgcn_alignbyte(w.c, w.b, 2) & 0xFFFFFFu; q[3] = w.c >> 8;
    } else if (in) {
#pragma unroll
        for (int j = 0; j < FP_PX; ++j)
            if (in & (1u << j)) {
                const unsigned char* p = c.q + ((size_t)qy * (size_t)c.q_w + (size_t)(qx + j)) * 3u;
                q[j] = p[0] | (p[1] << 8) | (p[2] << 16);
            }
    }
    return in;
}

// Composite a run of 4 canvas pixels and store it.  o: the warp's float32 blends (U8_BIAS included); tin: bit j = pixel j
// lies in T and its source coordinate is inside imgT; qin / q: from comp_load_q; `first`: ragged-row start as in blend_store.
__device__ __forceinline__ void comp_store(const CompArgs& c, const float (&o)[FP_PX][3], unsigned tin, unsigned qin,
                                           const uint32_t (&q)[FP_PX], unsigned char* drow, bool store_any, int first) {
    uint32_t px[FP_PX];
#pragma unroll
    for (int j = 0; j < FP_PX; ++j) {
        const bool t = tin & (1u << j), in_q = qin & (1u << j);
        uint32_t v = 0u;
        if (c.mode == 1) {                                     // paste: Q over the truncated warp over 0
            v = __builtin_amdgcn_cvt_pk_u8_f32(o[j][0], 0, v); v = __builtin_amdgcn_cvt_pk_u8_f32(o[j][1], 1, v);
            v = __builtin_amdgcn_cvt_pk_u8_f32(o[j][2], 2, v);
            v = in_q ? q[j] : (t ? v : 0u);
        } else {                                               // v = wq * Q + wt * warp, truncated (bias trick as in blend_store)
            const float wq = t ? (in_q ? c.wq_in : c.wq_out) : 1.f, wt = t ? (in_q ? c.wt_in : c.wt_out) : 0.f;
            const float kb = U8_BIAS - U8_BIAS * wt;           // o carries U8_BIAS once: wt * (o - BIAS) + BIAS + wq * Q
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float cq = (float)((q[j] >> (8 * k)) & 0xFFu);
                v = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(wt, o[j][k], __builtin_fmaf(wq, cq, kb)), k, v);
            }
        }
        px[j] = v;
    }
    if (!store_any) return;
    if (first <= 0) {
        pk3 w;
        w.a = (px[0] & 0xFFFFFFu) | (px[1] << 24);
        w.b = ((px[1] >> 8) & 0xFFFFu) | (px[2] << 16);
        w.c = ((px[2] >> 16) & 0xFFu) | (px[3] << 8);
        __builtin_memcpy(drow, &w, 12);
    } else {
#pragma unroll
        for (int j = 1; j < FP_PX; ++j)
            if (j >= first) { drow[3 * j] = (unsigned char)px[j]; drow[3 * j + 1] = (unsigned char)(px[j] >> 8); drow[3 * j + 2] = (unsigned char)(px[j] >> 16); }
    }
}

// ---- 8 pixels per lane: one wave = a 512-pixel output patch, one block = 4 waves = a 128 x 16 output tile ----------
// Same arithmetic as warp_rgb8_fast, but the per-lane overhead (patch decode, coordinate setup, footprint,
// staging set-up) is shared by twice as many pixels, and the patch SHAPE is a template parameter:
//
//   LOG_PW = 7   128 x 4  patches, 4 stacked vertically      -- widest stores; footprint fits up to ~2 degrees of rotation
//   LOG_PW = 6    64 x 8  patches, 2 x 2                     -- up to ~10 degrees
//   LOG_PW = 5    32 x 16 patches, 4 side by side            -- any rotation at scale ~1
//
// Staging (round 2): the footprint window of a patch shape is FIXED -- F8_LPRW<LOG_PW> chunks (12 bytes = 4 texels)
// wide, F8Window::ROWS rows tall -- and so is the staging lane map: lane l serves chunk l % LPRW of row l / LPRW of
// each pass, RPP = 64 / LPRW rows per pass.  Source address = scalar row-group base + a per-lane offset computed once,
// slab address = a per-lane offset computed once + a scalar: a pass is one 12-byte load, the RGB -> RGBX expansion
// (4 VALU) and one 16-byte LDS write.  (Round 1 let the footprint be any `nrows x C` area and derived row / column of
// chunk lane + 64*pass by a multiply-shift division per pass: 72 of the wave's 422 VALU instructions were staging
// arithmetic, now 28 -- profiles/r02_isa_histogram.txt.)  The host picks the shape whose window fits (launch_fast); a
// wave whose footprint does not fit takes the gather path, so the choice only affects speed, never the result.
//
//   LOG_PW   patch     window (texels x rows)   rows per pass     slab per wave   waves per SIMD (uint8 out)
//     7     128 x 4        168 x 7                  1  (42 lanes)    4704 B          7
//     6      64 x 8         84 x 15                 3  (63 lanes)    5040 B          8   (20 160 B per block: 8 blocks per CU)
//     5      32 x 16        40 x 38                 6  (60 lanes)    6080 B          6
// 32 x 16 is the shape for rotations: a patch turned by any angle has a footprint of at most 38 x 38 texels
// (sqrt(32^2 + 16^2) + 2), so its window is 38 rows tall -- at the price of two waves of occupancy.
constexpr int F8_PX = 8;
template <int LOG_PW> struct F8Window {
    static constexpr int LPRW = LOG_PW == 7 ? 42 : LOG_PW == 6 ? 21 : 10;   // chunks (staging lanes) per window row
    static constexpr int RPP = 64 / LPRW;                                   // window rows per staging pass
    static constexpr int LPITCH = 16 * LPRW;                                // slab bytes per window row
    static constexpr int ROWS = LOG_PW == 7 ? 7 : LOG_PW == 6 ? 15 : 38;    // window rows
    static constexpr int SLAB = ROWS * LPITCH;                              // bytes of LDS per wave
    static constexpr int PASSES = (ROWS + RPP - 1) / RPP;
};
inline bool f8_window_fits(int log_pw, long long nrows, long long ntex) {   // host + device twin of the `staged` test
    const int lprw = log_pw == 7 ? 42 : log_pw == 6 ? 21 : 10, rows = log_pw == 7 ? 7 : log_pw == 6 ? 15 : 38;
    return nrows >= 64 / lprw && nrows <= rows && ((ntex + 3) >> 2) <= lprw;
}

// Waves per SIMD the register allocator must leave room for.  uint8 output: 7 (a budget of 73 VGPRs).  Round 2's blend needed 72
// of them (its 64-register / 8-wave build fitted only by spilling a dozen scalars into VGPR lanes and was 0.8 % slower); round
// 3's v_fma_mix_f32 blend keeps no converted taps and comes to 60 VGPRs under the same bound, so 8 waves per SIMD are
// resident after all (LDS: 8 blocks x 20 160 B per CU) -- no spills of either kind.
// 32 x 16 patches 6 (their windows need the LDS).  The compositor: 6.  float32 output: 5 -- round 4 freed its extra 3 KB of LDS per wave
// (the re-deal region now lies inside the staging window), so its occupancy is a choice: same-box sweeps at 4 / 5 / 6 / 7 resident waves per
// SIMD (tools/occ_sweep_f32.sh): 0.386-0.392 ms per 16 x 4K at 4, 5 and 6, 0.406 at 7 (and 0.43 vs 0.47 in the slow allocation mode: this
// write-heavy kernel is bound by the memory system, more streams in flight make it slower, profiles/r04_lab_notes.txt section 9).
#ifndef RWH_LDS_PAD
#define RWH_LDS_PAD 0
#endif
#ifndef RWH_F32_WAVES
#define RWH_F32_WAVES 5
#endif
template <typename DstT, int LOG_PW> constexpr int f8_waves() {
    return sizeof(DstT) == 1 ? (LOG_PW == 7 ? 7 : LOG_PW == 5 ? 6 : 7) : (LOG_PW == 5 ? (RWH_F32_WAVES < 6 ? RWH_F32_WAVES : 6) : RWH_F32_WAVES);
}
// Order of work inside a wave: the two END pixels of every lane (own reciprocals) give the footprint and the staging
// loads go out at once; the other six pixels are computed run by run (3 + 3, one batch inversion each) right before
// their taps are read, so that only 4 pixels' coordinates, weights and taps are live at a time -- the kernel is
// occupancy-sensitive (time ~ 9.5 + 35/n us per 4K frame for n resident waves per SIMD, n <= 5 measured) and this
// keeps it at 6 waves (LDS-limited).  The end pixels are computed once, so the footprint and the taps can never
// disagree about a floor().
//
// HALVES (round 3, minification): a patch whose footprint does not fit the window is staged and blended ONE RUN AT A TIME --
// the left half (every lane's run 0: PW/2 x PH pixels), then the right half, each with a window of its own in the same
// slab.  A 16 x 16 half of a 32 x 16 patch fits the 40 x 38 window up to ~2.2x minification, where the whole patch stops
// fitting at ~1.09x (64 x 8: 1.28x) and every wave used to gather.  The halves' corners are the run's own end pixels, so
// footprint and taps agree about every floor() as they do for whole patches; a half that does not fit gathers.
// the four waves' LDS slabs: one function-local array, so that two bodies inlined into one kernel (the multi-frame kernel
// and its per-frame fallback) share it
template <int SLAB_BYTES>
__device__ __forceinline__ unsigned char* wave_slabs() {
    __shared__ __attribute__((aligned(16))) unsigned char s[4 * SLAB_BYTES];
    return s;
}
// ovr_logical >= 0: the (tile, image) this block works on, as the logical block index of a one-frame-per-block launch
// (the multi-frame kernel hands its non-interior patches to this body frame by frame)
template <typename DstT, int LOG_PW, bool COMP = false, int CH = 3, bool HALVES = false, bool EXT_SLAB = false>
__device__ __forceinline__ void fast8_body(const FastArgs& a, const Coef* tab, const CompArgs* cp = nullptr, const int ovr_logical = -1,
                                           unsigned char* const slab_ext = nullptr, const int ovr_wave = -1) {
    static_assert(CH == 3 || (CH == 4 && !COMP && sizeof(DstT) == 1), "4 channels: uint8 RGBA in and out, no compositor");
    static_assert(!HALVES || (!COMP && sizeof(DstT) == 1), "halves: uint8 output, no compositor");
    constexpr int PW = 1 << LOG_PW, PH = 512 / PW;          // patch width / height in pixels
    constexpr int LPR = PW / 8;                             // lanes per patch row
    constexpr int WX = 128 / PW;                            // waves side by side in the block tile
    // (float32 output re-deals a run's pixels into coalesced 16-byte stores through the window itself once its taps are read:
    //  redeal_store_f32; every window is larger than the 3 KB that takes)
    constexpr int SLAB = F8Window<LOG_PW>::SLAB + RWH_LDS_PAD;     // (RWH_LDS_PAD: lab builds that lower the occupancy through the LDS budget; 0 in the product)
    static_assert(sizeof(DstT) == 1 || SLAB >= 3072, "the re-deal region of the float32 output lies inside the staging window");
    // weight constants: RGB taps enter the blend as float16 halves (b * 2^-24), so the weights carry 2^24 (blend4)
    constexpr bool MX = CH == 3 && MIX_RGB;
    constexpr float WS = MX ? W_SCALE * MIX_S : W_SCALE, WO = MX ? W_ONE * MIX_S : W_ONE, WC = MX ? MIX_S : 1.0f;
    unsigned char* slab0;
    // slab_ext: THIS WAVE's slab inside the calling kernel's own array (at least SLAB bytes; the caller's per-wave stride may be larger than SLAB --
    // round 4: the multi-frame kernel's is 5 168 B against 5 040 here, and indexing its array with this body's stride let a border wave's staging
    // overwrite the end of its interior neighbour's window: tools/soak_mf.py, 3 % of random launches a few pixels off, never the same twice)
    if constexpr (EXT_SLAB) slab0 = slab_ext; else slab0 = wave_slabs<SLAB>();

    // ---- block / wave -> patch (all scalar) ---------------------------------------------------------
    const unsigned b = blockIdx.x;
#ifdef RWH_XCD_CHUNK_LOG   // lab builds: XCD k takes chunks of 2^LOG consecutive logical blocks round-robin (the product: ONE chunk of cpx blocks per XCD)
    const unsigned logical = ovr_logical >= 0 ? (unsigned)ovr_logical
                                              : (((b >> 3) >> RWH_XCD_CHUNK_LOG) << (RWH_XCD_CHUNK_LOG + 3)) + ((b & 7u) << RWH_XCD_CHUNK_LOG) + ((b >> 3) & ((1u << RWH_XCD_CHUNK_LOG) - 1u));
#else
    const unsigned logical = ovr_logical >= 0 ? (unsigned)ovr_logical : (b & 7u) * a.cpx + (b >> 3);   // XCD k walks logical blocks [k*cpx, (k+1)*cpx)
#endif
    const unsigned t = a.tiles_x_magic ? __umulhi(logical, a.tiles_x_magic) : logical;   // magic 0 <=> divisor 1
    const unsigned tx = logical - t * a.tiles_x;
    {   // a patch that owns nothing does nothing: a block past the grid, or a patch of the moved last tile of a ragged row
        // whose columns all belong to the tile on its left (out_w = 1921 leaves that tile ONE column).  (Folded into the one
        // early return on purpose: a second `return` further down -- e.g. for patches below the last row -- costs the
        // 64-VGPR uint8 kernel three spilled registers.)
        const int w0 = ovr_wave >= 0 ? ovr_wave : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const int own_from = (int)tx * 128 - min((int)tx * 128, a.out_w - 128);
        if ((logical >= a.nblocks) | ((w0 % (128 / (1 << LOG_PW)) + 1) * (1 << LOG_PW) <= own_from)) return;
    }
    const unsigned img = a.tiles_y_magic ? __umulhi(t, a.tiles_y_magic) : t;
    const unsigned ty = t - img * a.tiles_y;
    const Coef& co = tab ? tab[img] : a.c;                  // uniform: scalar loads either way
    const unsigned img_mem = tab ? (unsigned)co.image : img;   // where the image lives in the batch
    const int pwave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // physical wave: owns LDS slab pwave
    const int wave = ovr_wave >= 0 ? ovr_wave : pwave;                                   // which patch of the tile
    const int lane = threadIdx.x & 63;
    const int prow = lane / LPR, pq = lane % LPR;             // patch row, 4-pixel column group within a half row
    const int wave_x = (wave % WX) * PW, wave_y = (wave / WX) * PH;   // patch origin inside the 128 x 16 tile

    const int rr_raw = (int)ty * 16 + wave_y + prow;
    const int rr = min(rr_raw, a.rows - 1);                   // rows past the shard recompute its last row
    // a lane owns two runs of 4 pixels, PW/2 pixels apart: every store instruction then writes LPR lanes x 12 B of
    // contiguous bytes per patch row (8 contiguous pixels per lane would leave 12-byte holes in every store).
    // uint8 output: a run = 4 consecutive pixels (12 bytes); float32 output: 4 pixels LPR columns apart (12 bytes each).
    // A tile that sticks out of the row is moved left as a whole and owns only the columns >= its nominal start.
    const int tcol0 = (int)tx * 128;
    const int tcol = min(tcol0, a.out_w - 128);
    const int tshift = tcol0 - tcol;                          // uniform
    constexpr int PSTR = sizeof(DstT) == 1 ? 1 : LPR;         // column stride inside a run (float32 output: interleaved)
    const int lcol = wave_x + pq * (PSTR == 1 ? 4 : 1);       // column inside the tile
    const int c0p = tcol + lcol;
    const bool store_any = rr_raw < a.rows;

    const unsigned char* simg = a.src + (long long)img_mem * a.src_img_stride;       // uniform
    unsigned char* dimg = a.dst + (long long)img_mem * a.dst_img_stride;              // uniform
    // 32-bit lane offset from a uniform base (host guarantees rows*out_w*3*sizeof(DstT) < 2^32)
    DstT* drow = reinterpret_cast<DstT*>(dimg + ((uint32_t)rr * (uint32_t)a.pitch_w + (uint32_t)c0p) * (uint32_t)(CH * sizeof(DstT)));
    const uint32_t pitch = (uint32_t)a.src_w * (uint32_t)CH;

    // ---- compositor: where does this wave's patch lie relative to the rectangles T and Q? (uniform) ---------------------
    bool all_t = true;
    const int cy = a.row_begin + rr;                          // the lane's canvas row
    if constexpr (COMP) {
        const CompArgs& c = *cp;
        const int px0 = tcol + wave_x, px1 = px0 + PW - 1;                         // patch columns (inclusive)
        const int py0 = a.row_begin + min((int)ty * 16 + wave_y, a.rows - 1), py1 = a.row_begin + min((int)ty * 16 + wave_y + PH - 1, a.rows - 1);
        const bool none_t = (px1 < c.tsx) | (px0 >= c.tsx + c.wt) | (py1 < c.tsy) | (py0 >= c.tsy + c.ht);
        all_t = (px0 >= c.tsx) & (px1 < c.tsx + c.wt) & (py0 >= c.tsy) & (py1 < c.tsy + c.ht);
        const bool all_q = (px0 >= c.qsx) & (px1 < c.qsx + c.q_w) & (py0 >= c.qsy) & (py1 < c.qsy + c.q_h);
        if (none_t | ((c.mode == 1) & all_q)) {               // nothing of the warp shows: Q's pixels or 0
            if (all_q & (tshift == 0)) {                          // a patch inside Q: a plain 12-byte copy per run
                const unsigned char* qrow = c.q + ((size_t)(cy - c.qsy) * (size_t)c.q_w + (size_t)(c0p - c.qsx)) * 3u;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    pk3 w;
                    __builtin_memcpy(&w, qrow + 3 * (PW / 2) * h, 12);
                    if (store_any) __builtin_memcpy(reinterpret_cast<unsigned char*>(drow) + 3 * (PW / 2) * h, &w, 12);
                }
                return;
            }
            const float zero[FP_PX][3] = {};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t q[FP_PX];
                const unsigned qin = comp_load_q(c, cy, c0p + (PW / 2) * h, q);
                const int first = tshift - (lcol + (PW / 2) * h);
                CompArgs cc = c; cc.mode = 1;                 // (paste rule: Q, else 0 -- also what 'Rate' leaves outside T)
                comp_store(cc, zero, 0u, qin, q, reinterpret_cast<unsigned char*>(drow) + 3 * (PW / 2) * h, store_any & (first <= 3), max(first, 0));
            }
            return;
        }
    }

    // ---- pixels 0 and 7 of the lane: own reciprocal ------------------------------------------------------------
    const double fr = (double)(a.row_begin + rr), fc = (double)c0p;
    const double X0 = fma(fc, co.cx[2], fma(fr, co.cx[1], co.cx[0]));
    const double Y0 = fma(fc, co.cy[2], fma(fr, co.cy[1], co.cy[0]));
    const double W0 = fma(fc, co.cw[2], fma(fr, co.cw[1], co.cw[0]));
    uint32_t ehx[2], ehy[2], elx[2], ely[2];                  // the lane's first and last pixel
    bool wpos;
    {
        const double X7 = X0 + co.dxs8[6][0], Y7 = Y0 + co.dxs8[6][1], W7 = W0 + co.dxs8[6][2];
        double r0 = __builtin_amdgcn_rcp(W0); r0 = fma(fma(-W0, r0, 1.0), r0, r0);
        double r7 = __builtin_amdgcn_rcp(W7); r7 = fma(fma(-W7, r7, 1.0), r7, r7);
        const double ux0 = fma(X0, r0, MAGIC), uy0 = fma(Y0, r0, MAGIC), ux7 = fma(X7, r7, MAGIC), uy7 = fma(Y7, r7, MAGIC);
        ehx[0] = hi32(ux0); elx[0] = lo32(ux0); ehy[0] = hi32(uy0); ely[0] = lo32(uy0);
        ehx[1] = hi32(ux7); elx[1] = lo32(ux7); ehy[1] = hi32(uy7); ely[1] = lo32(uy7);
        // W is affine along a row: positive at both ends of every lane's span <=> positive on the whole patch.
        // Magnitudes inside [2^-300, 2^300] at both ends as well: then the product of three W of a run is a finite
        // normal number, and the batch inversion below needs no per-run check.
        const int h0 = (int)hi32(W0), h7 = (int)hi32(W7);
        wpos = __all((int)(h0 > 0x2D300000) & (int)(h0 < 0x52B00000) & (int)(h7 > 0x2D300000) & (int)(h7 < 0x52B00000));
    }
    // run h (0: pixels 0..3, 1: pixels 4..7 = columns +PW/2 ..): the three pixels that are not an end pixel share
    // one reciprocal (batch inversion); registers are reused between the runs
    uint32_t lx[FP_PX], ly[FP_PX], hx[FP_PX], hy[FP_PX];
    auto run_coords = [&](const int h) {
        double X[3], Y[3], W[3], rc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { X[j] = X0 + co.dxs8[3 * h + j][0]; Y[j] = Y0 + co.dxs8[3 * h + j][1]; W[j] = W0 + co.dxs8[3 * h + j][2]; }
        const double p12 = W[0] * W[1], P = p12 * W[2];
        if (wpos) {                                                        // finite, normal product of positive W
            double rp = __builtin_amdgcn_rcp(P);
            rp = fma(fma(-P, rp, 1.0), rp, rp);
            const double r12 = rp * W[2];
            rc[2] = rp * p12; rc[0] = r12 * W[1]; rc[1] = r12 * W[0];
        } else {                                                           // a W at / across zero (the horizon)
#pragma unroll
            for (int j = 0; j < 3; ++j) { double q = __builtin_amdgcn_rcp(W[j]); rc[j] = fma(fma(-W[j], q, 1.0), q, q); }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double ux = fma(X[j], rc[j], MAGIC), uy = fma(Y[j], rc[j], MAGIC);   // one rounding, onto the 2^-32 grid
            const int q = j + 1 - h;
            hx[q] = hi32(ux); lx[q] = lo32(ux); hy[q] = hi32(uy); ly[q] = lo32(uy);
        }
        hx[3 * h] = ehx[h]; lx[3 * h] = elx[h]; hy[3 * h] = ehy[h]; ly[3 * h] = ely[h];
    };

    // ---- wave-uniform footprint from the four patch corners, on the scalar unit ----
    // hi dwords compare like the integers they encode (same exponent); out-of-range / NaN corners end up
    // as the min or the max and fail the range test below.
    const int x0 = (int)__builtin_amdgcn_readlane(ehx[0], 0), x1 = (int)__builtin_amdgcn_readlane(ehx[1], LPR - 1);
    const int x2 = (int)__builtin_amdgcn_readlane(ehx[0], 64 - LPR), x3 = (int)__builtin_amdgcn_readlane(ehx[1], 63);
    const int y0 = (int)__builtin_amdgcn_readlane(ehy[0], 0), y1 = (int)__builtin_amdgcn_readlane(ehy[1], LPR - 1);
    const int y2 = (int)__builtin_amdgcn_readlane(ehy[0], 64 - LPR), y3 = (int)__builtin_amdgcn_readlane(ehy[1], 63);
    // (the window starts on a multiple of 4 texels = 12 / 16 bytes, so that the staging loads are dword / 16-byte aligned
    //  whenever the source rows are)
    const int hxmn = smin(smin(x0, x1), smin(x2, x3)) & ~3, hxmx = smax(smax(x0, x1), smax(x2, x3));
    const int hymn = smin(smin(y0, y1), smin(y2, y3)), hymx = smax(smax(y0, y1), smax(y2, y3));
    // (a coordinate below -1.5 * 2^20 makes s + MAGIC negative: its hi dword is a negative int, the min -- clamped to 0 before
    //  the subtraction, which would otherwise wrap around to a large POSITIVE texel index)
    const int xmn = (int)((uint32_t)smax(hxmn, 0) - MAGIC_HI), xmx = (int)((uint32_t)smax(hxmx, 0) - MAGIC_HI);
    const int ymn = (int)((uint32_t)smax(hymn, 0) - MAGIC_HI), ymx = (int)((uint32_t)smax(hymx, 0) - MAGIC_HI);
    // footprint: rows ymn..ymx+1, texels xmn..xmx+1, as nrows x C chunks of 4 texels
    using Win = F8Window<LOG_PW>;
    // ---- the whole patch maps outside the source (2.6 % of a 4K frame's patches with the bench homography, 4.6 % of a 1080p
    // frame's, 6.4 % of an 8K frame's: the corners of the warped quad's bounding box): W > 0 on the patch, so every pixel's
    // coordinate lies in the corners' box, and a box wholly left / right / above / below the source means every pixel is
    // masked -- zeros, with no staging, no coordinates for the other six pixels, no blend
    if constexpr (!COMP) {
        if (wpos & ((xmx < 0) | (xmn >= a.bound_w) | (ymx < 0) | (ymn >= a.bound_h))) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int first = tshift - (lcol + (PW / 2) * h);
                zero_store<DstT, PSTR, CH>(drow + CH * (PW / 2) * h, store_any & (first <= 3 * PSTR), max(first, 0));
            }
            return;
        }
    }
    const int nrows = ymx - ymn + 2, C = (xmx - xmn + 5) >> 2;
    // strictly inside: 0 <= floor(s) <= bound-2 on both axes, tap rows above the last source row (a chunk may read
    // up to 9 bytes past the footprint's last texel: never past the row below), and the footprint fits the window
    // (the HALVES kernel stages per half patch, below: no whole-patch window, no border window)
    const bool staged = !HALVES & wpos & (xmn >= 0) & (xmx < a.bound_w - 1) & (ymn >= 0) & (ymx < min(a.bound_h - 1, a.src_h - 2)) &
                        (nrows >= Win::RPP) & (nrows <= Win::ROWS) & (C <= Win::LPRW);

    // Border patches (4.8 % of a 4K frame's patches, 10.5 % of a 1080p frame's: some pixels map outside the source): the
    // window is the corners' box clamped into the image -- by convexity it still holds every VALID pixel's taps -- and is
    // staged like any other; the pixels then take their taps from the slab under their validity mask (below).  Not for
    // windows that reach the last two source rows (a staging chunk may read 9 bytes past its last texel).
    int wxmn = xmn, wymn = ymn, wnrows = nrows, wC = C;
    bool border = false;
    if (!HALVES & !staged & wpos) {
        const int cx0 = smin(smax(xmn, 0), a.bound_w - 1) & ~3, cx1 = smin(smax(xmx, 0), a.bound_w - 1);
        const int cy0 = smin(smax(ymn, 0), a.bound_h - 1), cy1 = smin(smax(ymx, 0), a.bound_h - 1);
        const int nr = smax(cy1 - cy0 + 2, Win::RPP), nc = (cx1 - cx0 + 5) >> 2;
        border = (nr <= Win::ROWS) & (nc <= Win::LPRW) & (cy0 + nr - 1 <= a.src_h - 2);
        if (border) { wxmn = cx0; wymn = cy0; wnrows = nr; wC = nc; }
    }

    // ---- staging loads go out now: lane -> (row srow of the pass, chunk scol), fixed for the kernel ---------------------
    const uint32_t slab_off_w = EXT_SLAB ? 0u : (uint32_t)pwave * (uint32_t)SLAB;      // this wave's slab inside slab0
    unsigned char* my = slab0 + slab_off_w;
    using chunk_t = typename std::conditional<CH == 4, pk4, pk3>::type;   // 4 texels as they lie in memory: 12 or 16 bytes
    chunk_t v[Win::PASSES];
    // lanes outside the footprint (chunk >= C, or past the last full row of a pass) load nothing: the texture-address
    // path charges per lane (~14 B / clk / CU), and at 79 % busy it was the kernel's second bottleneck
    const int srow = lane / Win::LPRW, scol = lane - srow * Win::LPRW;
    const bool sactive = (srow < Win::RPP) & (scol < C);
    const uint32_t wl = (uint32_t)(srow * Win::LPITCH + scol * 16);               // the lane's slab byte inside a pass
    const uint32_t goff = mad24_s((uint32_t)srow, pitch, CH == 4 ? (uint32_t)scol << 4 : mul24_12((uint32_t)scol));
    if (staged) {
        const unsigned char* gbase = simg + (size_t)((uint32_t)ymn * pitch + (uint32_t)xmn * (uint32_t)CH);   // uniform
#pragma unroll
        for (int p = 0; p < Win::PASSES; ++p) {
            if (p * Win::RPP < nrows && sactive) {                      // first half uniform: unused passes cost nothing
                // the last pass is pulled back so that it ends on the footprint's last row (it re-stages rows the pass
                // before it already wrote: same bytes to the same place)
                const int r0 = min(p * Win::RPP, nrows - Win::RPP);
#ifdef RWH_ABL_NOLOAD   // tools/warp_lab ablation hook (never defined in the product build)
                v[p] = chunk_t{goff, goff * 3u, goff * 5u};
#else
                __builtin_memcpy(&v[p], gbase + (size_t)((uint32_t)r0 * pitch) + goff, sizeof(chunk_t));
#endif
            }
        }
    }

    uint32_t a0[FP_PX], b0[FP_PX], a1[FP_PX], b1[FP_PX];
    float wx0[FP_PX], wx1[FP_PX], wy0[FP_PX], wy1[FP_PX];
    if (staged) {
        constexpr uint32_t lpitch = Win::LPITCH;                                            // slab pitch
        // slab byte of tap (iy, ix) = (iy - ymn) * lpitch + (ix - xmn) * 4, straight from the hi dwords: the 24-bit
        // multiply sees hy & 0xFFFFFF = 0x380000 + iy, the shift drops the exponent bits of hx; both constants, the
        // footprint origin and the slab's own LDS offset go into one uniform
        // (the multiply-add is spelled in assembly: given __umul24, LLVM distributes the subtraction and emits a
        //  quarter-rate v_mul_lo_u32 per tap)
        const uint32_t slab_off = slab_off_w;
        const uint32_t tap_c = ((uint32_t)hymn & 0xFFFFFFu) * lpitch + ((uint32_t)hxmn << 2) - slab_off;   // uniform
#pragma unroll
        for (int p = 0; p < Win::PASSES; ++p) {
            if (p * Win::RPP < nrows && sactive) {                      // 12 packed bytes -> 4 RGBX texels
                const int r0 = min(p * Win::RPP, nrows - Win::RPP);
                uint4 t4;
                if constexpr (CH == 4) {                      // RGBA texels are slab texels already
                    t4.x = v[p].a; t4.y = v[p].b; t4.z = v[p].c; t4.w = v[p].d;
                } else {                                      // RGBX with X = 0 (the blend reads B as the texel's upper half)
                    t4.x = v[p].a & 0xFFFFFFu;
                    t4.y = __builtin_amdgcn_perm(v[p].b, v[p].a, 0x0C050403u);
                    t4.z = __builtin_amdgcn_perm(v[p].c, v[p].b, 0x0C040302u);
                    t4.w = v[p].c >> 8;
                }
                *reinterpret_cast<uint4*>(my + (uint32_t)r0 * lpitch + wl) = t4;
            }
        }
        // the slab is wave-private: order this wave's LDS writes before its LDS reads, no block barrier
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        auto taps_of_run = [&](const int h) __attribute__((always_inline)) {      // coordinates, weights and the 16 taps of run h
            run_coords(h);
#pragma unroll
            for (int j = 0; j < FP_PX; ++j) {
                weights2(lx[j], ly[j], WS, WO, WC, wx0[j], wx1[j], wy0[j], wy1[j]);
                const uint32_t lo = mad24_s(hy[j], lpitch, shl2_add_s(hx[j], 0u - tap_c));
#ifdef RWH_ABL_NOLDS    // tools/warp_lab ablation hook (never defined in the product build)
                a0[j] = lo; b0[j] = lo * 3u; a1[j] = lo * 5u; b1[j] = lo * 7u;
#else
                const uint32_t* t0 = reinterpret_cast<const uint32_t*>(slab0 + lo);
                const uint32_t* t1 = reinterpret_cast<const uint32_t*>(slab0 + lo + lpitch);
                a0[j] = t0[0]; b0[j] = t0[1]; a1[j] = t1[0]; b1[j] = t1[1];
#endif
            }
        };
        if constexpr (PSTR > 1) {
            // float32 output, no ragged row in this tile (uniform): both runs are blended BEFORE anything is stored -- run 1 first, its
            // 12 floats kept --, so that every tap has been read and the staging window itself can serve as the re-deal region
            // (round 4: the 3 KB the region used to take beside the window cost three of the eight resident waves per SIMD)
            if (tshift == 0) {
                float o1[FP_PX][3], o0[FP_PX][3];
                taps_of_run(1);
                blend4<false, 3, true>(a0, b0, a1, b1, wx0, wx1, wy0, wy1, o1);
                taps_of_run(0);
                blend4<false, 3, true>(a0, b0, a1, b1, wx0, wx1, wy0, wy1, o0);
                unsigned char* region = my + prow * (48 * PSTR);           // (the wave's LDS operations execute in order: these writes follow every tap read)
                redeal_store_f32<PSTR>(o0, drow, store_any, region, pq);
                redeal_store_f32<PSTR>(o1, drow + CH * (PW / 2), store_any, region, pq);
                return;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {                           // the two runs of 4 pixels: registers are reused
            taps_of_run(h);
            const int first = tshift - (lcol + (PW / 2) * h);   // local pixels at columns >= first are this tile's
            if constexpr (COMP) {
                float o[FP_PX][3];
                blend4<true, 3, true>(a0, b0, a1, b1, wx0, wx1, wy0, wy1, o);
                uint32_t q[FP_PX];
                const int cx = c0p + (PW / 2) * h;
                const unsigned qin = comp_load_q(*cp, cy, cx, q);
                unsigned tin = 15u;                           // staged: every source coordinate is inside imgT
                if (!all_t) {
                    tin = 0u;
#pragma unroll
                    for (int j = 0; j < FP_PX; ++j)
                        tin |= (unsigned)((cy >= cp->tsy) & (cy < cp->tsy + cp->ht) & (cx + j >= cp->tsx) & (cx + j < cp->tsx + cp->wt)) << j;
                }
                comp_store(*cp, o, tin, qin, q, reinterpret_cast<unsigned char*>(drow) + 3 * (PW / 2) * h, store_any & (first <= 3), max(first, 0));
            } else {
                blend_store<DstT, PSTR, CH, true>(a0, b0, a1, b1, wx0, wx1, wy0, wy1, drow + CH * (PW / 2) * h,
                                        store_any & (first <= 3 * PSTR), max(first, 0));
            }
        }
        return;
    }

    // ---- masked runs (border / oversize waves): validity per pixel, zero weights for the pixels outside the source ------
    auto finish_masked_run = [&](const int h, const unsigned vbits) __attribute__((always_inline)) {
        const int first = tshift - (lcol + (PW / 2) * h);
        if constexpr (COMP) {
            float o[FP_PX][3];
            blend4<true, 3, true>(a0, b0, a1, b1, wx0, wx1, wy0, wy1, o);
            uint32_t q[FP_PX];
            const int cx = c0p + (PW / 2) * h;
            const unsigned qin = comp_load_q(*cp, cy, cx, q);
            unsigned tin = 0u;
#pragma unroll
            for (int j = 0; j < FP_PX; ++j)
                tin |= (unsigned)((cy >= cp->tsy) & (cy < cp->tsy + cp->ht) & (cx + j >= cp->tsx) & (cx + j < cp->tsx + cp->wt)) << j;
            // a pixel outside T never shows the warp, whatever its coordinate maps to (T is the reference's int()-truncated bbox)
            float oz[FP_PX][3];
#pragma unroll
            for (int j = 0; j < FP_PX; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) oz[j][c] = (tin & (1u << j)) ? o[j][c] : U8_BIAS;
            comp_store(*cp, oz, tin & vbits, qin, q, reinterpret_cast<unsigned char*>(drow) + 3 * (PW / 2) * h, store_any & (first <= 3), max(first, 0));
        } else {
            blend_store<DstT, PSTR, CH, true>(a0, b0, a1, b1, wx0, wx1, wy0, wy1, drow + CH * (PW / 2) * h,
                                    store_any & (first <= 3 * PSTR), max(first, 0));
        }
    };
    // 0 <= s <= bound-1 on the bit patterns: positive doubles order like unsigned integers; negative values and NaNs have
    // patterns outside [MAGIC_BITS, xmax_bits]
    auto pixel_valid = [&](int j) __attribute__((always_inline)) -> bool {
        const unsigned long long ubx = ((unsigned long long)hx[j] << 32) | lx[j], uby = ((unsigned long long)hy[j] << 32) | ly[j];
        return (ubx >= MAGIC_BITS) & (ubx <= a.xmax_bits) & (uby >= MAGIC_BITS) & (uby <= a.ymax_bits);
    };

    // ---- border waves: taps from the clamped window's slab -----------------------------------------------------------
    if (border) {
        {   // stage the clamped window with registers of its own (sharing the main path's `v` stretched its live range over
            // both paths and made the compiler spill the staging loads)
            const unsigned char* gbase = simg + (size_t)((uint32_t)wymn * pitch + (uint32_t)wxmn * (uint32_t)CH);   // uniform
            const bool bactive = (srow < Win::RPP) & (scol < wC);
            chunk_t vb[Win::PASSES];
#pragma unroll
            for (int p = 0; p < Win::PASSES; ++p)
                if (p * Win::RPP < wnrows && bactive)
                    __builtin_memcpy(&vb[p], gbase + (size_t)((uint32_t)min(p * Win::RPP, wnrows - Win::RPP) * pitch) + goff, sizeof(chunk_t));
#pragma unroll
            for (int p = 0; p < Win::PASSES; ++p)
                if (p * Win::RPP < wnrows && bactive) {
                    uint4 t4;
                    if constexpr (CH == 4) {
                        t4.x = vb[p].a; t4.y = vb[p].b; t4.z = vb[p].c; t4.w = vb[p].d;
                    } else {
                        t4.x = vb[p].a & 0xFFFFFFu;
                        t4.y = __builtin_amdgcn_perm(vb[p].b, vb[p].a, 0x0C050403u);
                        t4.z = __builtin_amdgcn_perm(vb[p].c, vb[p].b, 0x0C040302u);
                        t4.w = vb[p].c >> 8;
                    }
                    *reinterpret_cast<uint4*>(my + (uint32_t)min(p * Win::RPP, wnrows - Win::RPP) * (uint32_t)Win::LPITCH + wl) = t4;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const uint32_t bx = MAGIC_HI + (uint32_t)wxmn, by = MAGIC_HI + (uint32_t)wymn;          // uniform
        const uint32_t rx_max = (uint32_t)(4 * wC - 2), ry_max = (uint32_t)(wnrows - 2);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned vbits = 0u;
            run_coords(h);
#pragma unroll
            for (int j = 0; j < FP_PX; ++j) {
                const bool valid = pixel_valid(j);
                weights2(lx[j], ly[j], valid ? WS : 0.f, valid ? WO : 0.f, valid ? WC : 0.f, wx0[j], wx1[j], wy0[j], wy1[j]);
                vbits |= (unsigned)valid << j;
                // window-relative texel, kept inside the window whatever the coordinate is: a pixel outside the source has
                // zero weights, and a valid one lies in the window by convexity (up to a floor() that rounding moved across
                // an integer, where the tap that could fall outside has a weight < 2^-32)
                const uint32_t rx = min(hx[j] - bx, rx_max), ry = min(hy[j] - by, ry_max);
                const uint32_t lo = mad24_s(ry, (uint32_t)Win::LPITCH, rx << 2);
                const uint32_t* t0 = reinterpret_cast<const uint32_t*>(my + lo);
                const uint32_t* t1 = reinterpret_cast<const uint32_t*>(my + lo + Win::LPITCH);
                a0[j] = t0[0]; b0[j] = t0[1]; a1[j] = t1[0]; b1[j] = t1[1];
            }
            finish_masked_run(h, vbits);
        }
        return;
    }

    // ---- HALVES: one run at a time, each half patch with a window of its own; halves that do not fit are left to the gathers --
    // Round 4: BOTH halves' staging loads go out before either is waited for (round 3 staged, waited and blended half 0, then staged
    // half 1: two exposed load latencies per wave in a kernel whose units are all below 50 % busy).  The second half's 15 chunk
    // registers stay live across the first half's blend; its coordinates are recomputed when its turn comes (40 VALU instructions,
    // cheaper than 16 more live registers).
    unsigned todo = 3u;                                         // uniform: runs still to do
    if constexpr (HALVES) {
        todo = wpos ? 0u : 3u;
        if (wpos) {
            constexpr uint32_t lpitch = Win::LPITCH;
            chunk_t vh[2][Win::PASSES];
            int hrows[2];                                       // uniform, per half: rows of its window (0: does not fit)
            uint32_t tapc[2];                                   // uniform, per half: the tap-address constant of its window
            bool hact[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                run_coords(h);
                const uint32_t lhx = h == 0 ? ehx[0] : hx[0], lhy = h == 0 ? ehy[0] : hy[0];      // the run's first pixel
                const uint32_t rhx = h == 0 ? hx[3] : ehx[1], rhy = h == 0 ? hy[3] : ehy[1];      // ... and its last
                const int hx0 = (int)__builtin_amdgcn_readlane(lhx, 0), hx1 = (int)__builtin_amdgcn_readlane(rhx, LPR - 1);
                const int hx2 = (int)__builtin_amdgcn_readlane(lhx, 64 - LPR), hx3 = (int)__builtin_amdgcn_readlane(rhx, 63);
                const int hy0 = (int)__builtin_amdgcn_readlane(lhy, 0), hy1 = (int)__builtin_amdgcn_readlane(rhy, LPR - 1);
                const int hy2 = (int)__builtin_amdgcn_readlane(lhy, 64 - LPR), hy3 = (int)__builtin_amdgcn_readlane(rhy, 63);
                const int qxmn = smin(smin(hx0, hx1), smin(hx2, hx3)) & ~3, qxmx = smax(smax(hx0, hx1), smax(hx2, hx3));
                const int qymn = smin(smin(hy0, hy1), smin(hy2, hy3)), qymx = smax(smax(hy0, hy1), smax(hy2, hy3));
                const int sxmn = (int)((uint32_t)smax(qxmn, 0) - MAGIC_HI), sxmx = (int)((uint32_t)smax(qxmx, 0) - MAGIC_HI);
                const int symn = (int)((uint32_t)smax(qymn, 0) - MAGIC_HI), symx = (int)((uint32_t)smax(qymx, 0) - MAGIC_HI);
                const int nr = symx - symn + 2, hC = (sxmx - sxmn + 5) >> 2;
                const bool hfits = (sxmn >= 0) & (sxmx < a.bound_w - 1) & (symn >= 0) & (symx < min(a.bound_h - 1, a.src_h - 2)) &
                                   (nr >= Win::RPP) & (nr <= Win::ROWS) & (hC <= Win::LPRW);
                hrows[h] = hfits ? nr : 0;
                hact[h] = (srow < Win::RPP) & (scol < hC);
                tapc[h] = ((uint32_t)qymn & 0xFFFFFFu) * lpitch + ((uint32_t)qxmn << 2) - slab_off_w;
                if (hfits) {
                    const unsigned char* gbase = simg + (size_t)((uint32_t)symn * pitch + (uint32_t)sxmn * (uint32_t)CH);   // uniform
#pragma unroll
                    for (int p = 0; p < Win::PASSES; ++p)
                        if (p * Win::RPP < nr && hact[h])
                            __builtin_memcpy(&vh[h][p], gbase + (size_t)((uint32_t)min(p * Win::RPP, nr - Win::RPP) * pitch) + goff, sizeof(chunk_t));
                } else {
                    todo |= 1u << h;
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (hrows[h] == 0) continue;                    // uniform
#pragma unroll
                for (int p = 0; p < Win::PASSES; ++p)
                    if (p * Win::RPP < hrows[h] && hact[h]) {
                        uint4 t4;
                        if constexpr (CH == 4) {
                            t4.x = vh[h][p].a; t4.y = vh[h][p].b; t4.z = vh[h][p].c; t4.w = vh[h][p].d;
                        } else {
                            t4.x = vh[h][p].a & 0xFFFFFFu;
                            t4.y = __builtin_amdgcn_perm(vh[h][p].b, vh[h][p].a, 0x0C050403u);
                            t4.z = __builtin_amdgcn_perm(vh[h][p].c, vh[h][p].b, 0x0C040302u);
                            t4.w = vh[h][p].c >> 8;
                        }
                        *reinterpret_cast<uint4*>(my + (uint32_t)min(p * Win::RPP, hrows[h] - Win::RPP) * lpitch + wl) = t4;
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (h == 0) run_coords(0);                      // (run 1's coordinates are the ones in the registers right now)
#pragma unroll
                for (int j = 0; j < FP_PX; ++j) {
                    weights2(lx[j], ly[j], WS, WO, WC, wx0[j], wx1[j], wy0[j], wy1[j]);
                    const uint32_t lo = mad24_s(hy[j], lpitch, shl2_add_s(hx[j], 0u - tapc[h]));
                    const uint32_t* t0 = reinterpret_cast<const uint32_t*>(slab0 + lo);
                    const uint32_t* t1 = reinterpret_cast<const uint32_t*>(slab0 + lo + lpitch);
                    a0[j] = t0[0]; b0[j] = t0[1]; a1[j] = t1[0]; b1[j] = t1[1];
                }
                const int first = tshift - (lcol + (PW / 2) * h);
                blend_store<DstT, PSTR, CH, true>(a0, b0, a1, b1, wx0, wx1, wy0, wy1, drow + CH * (PW / 2) * h,
                                                  store_any & (first <= 3 * PSTR), max(first, 0));
                // the slab is reused by the other half: this half's reads are complete (their values were consumed above)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (h == 0 && hrows[1] != 0) run_coords(1);
            }
        }
        if (todo == 0u) return;
    }

    // ---- everything else (windows that do not fit, the last two source rows, W <= 0): masked gathers from global memory --
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t off[FP_PX];
        bool near_end = false;
        unsigned vbits = 0u;
        if (HALVES && !(todo & (1u << h))) continue;
        run_coords(h);
#pragma unroll
        for (int j = 0; j < FP_PX; ++j) {
            const bool valid = pixel_valid(j);
            const int ix = (int)(hx[j] - MAGIC_HI), iy = (int)(hy[j] - MAGIC_HI);
            weights2(lx[j], ly[j], valid ? WS : 0.f, valid ? WO : 0.f, valid ? WC : 0.f, wx0[j], wx1[j], wy0[j], wy1[j]);
            off[j] = valid ? (uint32_t)iy * pitch + (uint32_t)ix * (uint32_t)CH : 0u;
            near_end |= valid & (iy > a.src_h - 3);
            vbits |= (unsigned)valid << j;
        }
        if (!__any(vbits != 0u)) {           // the whole run maps outside the source (the corners of a warped quad's bounding
#pragma unroll                           // box): every weight is 0 already, nothing to load
            for (int j = 0; j < FP_PX; ++j) { a0[j] = 0u; b0[j] = 0u; a1[j] = 0u; b1[j] = 0u; }
        } else if (!__any(near_end)) {
#pragma unroll
            for (int j = 0; j < FP_PX; ++j) {
                const pk2 r0 = ld8(simg + off[j]);
                const pk2 r1 = ld8(simg + off[j] + pitch);
                if constexpr (CH == 4) { a0[j] = r0.a; b0[j] = r0.b; a1[j] = r1.a; b1[j] = r1.b; }
                else {                                        // 24-bit texels, upper byte 0 (see blend4)
                    a0[j] = r0.a & 0xFFFFFFu; b0[j] = __builtin_amdgcn_perm(r0.b, r0.a, 0x0C050403u);
                    a1[j] = r1.a & 0xFFFFFFu; b1[j] = __builtin_amdgcn_perm(r1.b, r1.a, 0x0C050403u);
                }
            }
        } else {  // byte-exact loads, +1 taps clamped to the image (their weight is 0 when clamped)
            const uint32_t last = (uint32_t)a.src_h * pitch - (uint32_t)CH;
#pragma unroll
            for (int j = 0; j < FP_PX; ++j) {
                const uint32_t o00 = off[j];
                const uint32_t o01 = min(o00 + (uint32_t)CH, last), o10 = min(o00 + pitch, last), o11 = min(o00 + pitch + (uint32_t)CH, last);
                if constexpr (CH == 4) {
                    a0[j] = ld4(simg + o00); b0[j] = ld4(simg + o01); a1[j] = ld4(simg + o10); b1[j] = ld4(simg + o11);
                    continue;
                }
                a0[j] = simg[o00] | (simg[o00 + 1] << 8) | (simg[o00 + 2] << 16);
                b0[j] = simg[o01] | (simg[o01 + 1] << 8) | (simg[o01 + 2] << 16);
                a1[j] = simg[o10] | (simg[o10 + 1] << 8) | (simg[o10 + 2] << 16);
                b1[j] = simg[o11] | (simg[o11 + 1] << 8) | (simg[o11 + 2] << 16);
            }
        }
        finish_masked_run(h, vbits);
    }
}

template <typename DstT, int LOG_PW>
__global__ __launch_bounds__(256, (f8_waves<DstT, LOG_PW>())) void warp_rgb8_fast8(const FastArgs a) { fast8_body<DstT, LOG_PW>(a, nullptr); }
// minification (see HALVES above): the same kernel, windows per half patch
// (round 4, both halves' staging loads in flight at once: 86 VGPRs for 64 x 8 patches -> 5 waves per SIMD, 102 for 32 x 16 -> 4; round 3's
//  one-half-at-a-time form ran 6 with two exposed load latencies per wave)
#ifndef RWH_F8H_WAVES
#define RWH_F8H_WAVES(log_pw) ((log_pw) == 5 ? 4 : 5)
#endif
template <int LOG_PW>
__global__ __launch_bounds__(256, RWH_F8H_WAVES(LOG_PW)) void warp_rgb8_fast8h(const FastArgs a) { fast8_body<unsigned char, LOG_PW, false, 3, true>(a, nullptr); }
template <int LOG_PW>
__global__ __launch_bounds__(256, RWH_F8H_WAVES(LOG_PW)) void warp_rgb8_fast8h_tab(const FastArgs a, const CoefTab t) { fast8_body<unsigned char, LOG_PW, false, 3, true>(a, t.e); }
// RGBA uint8 in, RGBA uint8 out: 4-byte texels are slab texels as they lie in memory (no RGB -> RGBX expansion), a pixel is
// one dword and a run one 16-byte store.  (The reference's own 4-channel images are float32: the generic kernel.)
template <int LOG_PW>
__global__ __launch_bounds__(256, 6) void warp_rgba8_fast8(const FastArgs a) { fast8_body<unsigned char, LOG_PW, false, 4>(a, nullptr); }
// canvas compositor form (uint8): the output grid is the canvas, imgQ is composited in the epilogue (CompArgs)
template <int LOG_PW>
__global__ __launch_bounds__(256, 6) void warp_rgb8_comp(const FastArgs a, const CompArgs c) { fast8_body<unsigned char, LOG_PW, true>(a, nullptr, &c); }
// one homography per image: image i of the launch uses t.e[i]
template <typename DstT, int LOG_PW>
__global__ __launch_bounds__(256, (f8_waves<DstT, LOG_PW>())) void warp_rgb8_fast8_tab(const FastArgs a, const CoefTab t) { fast8_body<DstT, LOG_PW>(a, t.e); }

// ---- several frames per block (round 4): one homography, a batch of frames ---------------------------------------------
// With one homography for the whole batch (BASELINE configs 2 and 5: n_h == 1) everything a patch computes BEFORE it touches
// pixel data -- the 8 float64 source coordinates per lane, the footprint, the staging lane map, the 32 tap weights and the 8
// slab addresses -- is the same in every frame.  The one-frame kernel recomputed it per frame: 186 of its 366 VALU
// instructions per wave (profiles/r03_isa_histogram.txt: float64 coordinates 69, weights 40, addresses ~45, decode / footprint
// ~30).  Here a block owns one 128 x 16 tile in mf_frames CONSECUTIVE frames: an interior (staged) wave computes the geometry
// once, keeps weights and tap addresses in registers (40 VGPRs) and then loops over the frames doing only the data work --
// stage, 16 tap reads, 32 v_perm + 96 v_fma_mix, pack, store; the next frame's staging loads are in flight while this frame
// is blended.  Same arithmetic in the same order as fast8_body's staged path (weights as blend4<FOLDED> derives them): the
// output is bit-identical to the one-frame kernel's (test_warp_multi_frame_equals_single).  Every other kind of wave --
// border, outside, gather -- is handed to fast8_body frame by frame.
// 4 blended pixels (U8_BIAS included) -> 12 packed bytes; the same 12 v_cvt_pk_u8_f32 as blend_store's
__device__ __forceinline__ pk3 pack_run_u8(const float (&o)[FP_PX][3]) {
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
    return w;
}
// ... stored whole, or (the one straddling lane of a ragged row) from local pixel `shift` on, byte by byte
__device__ __forceinline__ void store_run_pk(const pk3& w, unsigned char* drow, bool store_any, int shift) {
    if (!store_any) return;
#ifdef RWH_F8M_STORE_MOD     // lab: cache-policy bits on the multi-frame kernel's stores ("nt", "sc1", "sc0 sc1")
    if (shift == 0) {
        typedef uint32_t u3s __attribute__((ext_vector_type(3)));
        const u3s d = {w.a, w.b, w.c};
        asm volatile("global_store_dwordx3 %0, %1, off " RWH_F8M_STORE_MOD "\n\ts_nop 2" : : "v"(drow), "v"(d) : "memory");
        return;
    }
#else
    if (shift == 0) { __builtin_memcpy(drow, &w, 12); return; }
#endif
    const unsigned long long lo8 = ((unsigned long long)w.b << 32) | w.a;
#pragma unroll
    for (int i = 3; i < 12; ++i)
        if (i >= 3 * shift) drow[i] = (unsigned char)(i < 8 ? lo8 >> (8 * i) : w.c >> (8 * (i - 8)));
}
#ifndef RWH_F8M_SKEW
#define RWH_F8M_SKEW 8
#endif
__device__ __forceinline__ void store_run_u8(const float (&o)[FP_PX][3], unsigned char* drow, bool store_any, int shift) {
    if (!store_any) return;
    if (shift == 0) {
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

// (the four tap weights of every pixel stay in registers across the frames: 32 VGPRs.  Keeping only (wx1, wy1) and re-deriving
//  the four per frame was tried: hipcc hoists the derivation out of the frame loop again -- same code)
template <int LOG_PW>
__device__ __forceinline__ void fast8m_body(const FastArgs& a) {
    constexpr int PW = 1 << LOG_PW, PH = 512 / PW, LPR = PW / 8, WX = 128 / PW;
    using Win = F8Window<LOG_PW>;
    // slab pitch = 2 dwords more than a multiple of 4: the tap reads of a half-wave (four patch rows, lanes 4 texels = 4 dwords
    // apart) then fall on two bank residues instead of one -- 2-way instead of 4-way conflicts.  With the per-frame arithmetic
    // halved, the LDS tap reads (16 ds_read2_b32 per wave and frame at 64 cycles each, tools/energy_probe) became the limiter.
    constexpr uint32_t lpitch = Win::LPITCH + RWH_F8M_SKEW;
    constexpr int SLAB = ((Win::ROWS * (int)lpitch + 15) / 16) * 16;
    static_assert(SLAB >= Win::SLAB, "the per-frame fallback body shares this slab");
    constexpr float WS = W_SCALE * MIX_S, WO = W_ONE * MIX_S, WC = MIX_S;
    static_assert(MIX_RGB, "the multi-frame kernel is written for the float16-halves blend");
    unsigned char* const slab0 = wave_slabs<SLAB>();

    // ---- block -> (128 x 16 tile, group of frames); tile fastest, so that neighbouring blocks work on neighbouring tiles of the
    // same frames.  (Blocks of four patches side by side -- 256 x 8 pixels for 64 x 8 patches -- were tried for DRAM page
    // locality of the stores: 8K frames 10 % slower, the waves of a 2 x 2 block share their vertical halo in L1 / L2.)
    const unsigned b = blockIdx.x;
    const unsigned logical = (b & 7u) * a.mf_cpx + (b >> 3);
    if (logical >= a.mf_nblocks) return;
    const unsigned g = a.ntiles_magic ? __umulhi(logical, a.ntiles_magic) : (a.ntiles == 1u ? logical : 0u);
    const unsigned ti = logical - g * a.ntiles;
    const int f0 = (int)g * a.mf_frames, f1 = min(f0 + a.mf_frames, a.mf_batch);
    const unsigned ty = a.tiles_x_magic ? __umulhi(ti, a.tiles_x_magic) : ti;     // ti < ntiles <= nblocks: the magic holds
    const unsigned tx = ti - ty * a.tiles_x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pwave = wave;
    {
        const int own_from = (int)tx * 128 - min((int)tx * 128, a.out_w - 128);
        if ((wave % WX + 1) * PW <= own_from) return;        // a patch of the moved last tile that owns no column (fast8_body)
    }
    const Coef& co = a.c;
    const int lane = threadIdx.x & 63;
    const int prow = lane / LPR, pq = lane % LPR;
    const int wave_x = (wave % WX) * PW, wave_y = (wave / WX) * PH;
    const int rr_raw = (int)ty * 16 + wave_y + prow;
    const int rr = min(rr_raw, a.rows - 1);
    const int tcol0 = (int)tx * 128;
    const int tcol = min(tcol0, a.out_w - 128);
    const int tshift = tcol0 - tcol;
    const int lcol = wave_x + pq * 4;
    const int c0p = tcol + lcol;
    const bool store_any = rr_raw < a.rows;
    const uint32_t pitch = (uint32_t)a.src_w * 3u;

    // ---- the lane's end pixels and the footprint: fast8_body's arithmetic, operation by operation ------------------------
    const double fr = (double)(a.row_begin + rr), fc = (double)c0p;
    const double X0 = fma(fc, co.cx[2], fma(fr, co.cx[1], co.cx[0]));
    const double Y0 = fma(fc, co.cy[2], fma(fr, co.cy[1], co.cy[0]));
    const double W0 = fma(fc, co.cw[2], fma(fr, co.cw[1], co.cw[0]));
    uint32_t ehx[2], ehy[2], elx[2], ely[2];
    bool wpos;
    {
        const double X7 = X0 + co.dxs8[6][0], Y7 = Y0 + co.dxs8[6][1], W7 = W0 + co.dxs8[6][2];
        double r0 = __builtin_amdgcn_rcp(W0); r0 = fma(fma(-W0, r0, 1.0), r0, r0);
        double r7 = __builtin_amdgcn_rcp(W7); r7 = fma(fma(-W7, r7, 1.0), r7, r7);
        const double ux0 = fma(X0, r0, MAGIC), uy0 = fma(Y0, r0, MAGIC), ux7 = fma(X7, r7, MAGIC), uy7 = fma(Y7, r7, MAGIC);
        ehx[0] = hi32(ux0); elx[0] = lo32(ux0); ehy[0] = hi32(uy0); ely[0] = lo32(uy0);
        ehx[1] = hi32(ux7); elx[1] = lo32(ux7); ehy[1] = hi32(uy7); ely[1] = lo32(uy7);
        const int h0 = (int)hi32(W0), h7 = (int)hi32(W7);
        wpos = __all((int)(h0 > 0x2D300000) & (int)(h0 < 0x52B00000) & (int)(h7 > 0x2D300000) & (int)(h7 < 0x52B00000));
    }
    const int x0 = (int)__builtin_amdgcn_readlane(ehx[0], 0), x1 = (int)__builtin_amdgcn_readlane(ehx[1], LPR - 1);
    const int x2 = (int)__builtin_amdgcn_readlane(ehx[0], 64 - LPR), x3 = (int)__builtin_amdgcn_readlane(ehx[1], 63);
    const int y0 = (int)__builtin_amdgcn_readlane(ehy[0], 0), y1 = (int)__builtin_amdgcn_readlane(ehy[1], LPR - 1);
    const int y2 = (int)__builtin_amdgcn_readlane(ehy[0], 64 - LPR), y3 = (int)__builtin_amdgcn_readlane(ehy[1], 63);
    const int hxmn = smin(smin(x0, x1), smin(x2, x3)) & ~3, hxmx = smax(smax(x0, x1), smax(x2, x3));
    const int hymn = smin(smin(y0, y1), smin(y2, y3)), hymx = smax(smax(y0, y1), smax(y2, y3));
    const int xmn = (int)((uint32_t)smax(hxmn, 0) - MAGIC_HI), xmx = (int)((uint32_t)smax(hxmx, 0) - MAGIC_HI);
    const int ymn = (int)((uint32_t)smax(hymn, 0) - MAGIC_HI), ymx = (int)((uint32_t)smax(hymx, 0) - MAGIC_HI);
    const int nrows = ymx - ymn + 2, C = (xmx - xmn + 5) >> 2;
    const bool staged = wpos & (xmn >= 0) & (xmx < a.bound_w - 1) & (ymn >= 0) & (ymx < min(a.bound_h - 1, a.src_h - 2)) &
                        (nrows >= Win::RPP) & (nrows <= Win::ROWS) & (C <= Win::LPRW);
    if (!staged) {      // outside / border / gather patches: the one-frame body, frame by frame
        for (int f = f0; f < f1; ++f)
            fast8_body<unsigned char, LOG_PW, false, 3, false, true>(a, nullptr, nullptr, (int)((unsigned)f * a.ntiles + ti), slab0 + pwave * SLAB);
        return;
    }

    // ---- staging lane map (fast8_body) ---------------------------------------------------------------------------------
    unsigned char* my = slab0 + pwave * SLAB;
    const int srow = lane / Win::LPRW, scol = lane - srow * Win::LPRW;
    const bool sactive = (srow < Win::RPP) & (scol < C);
    const uint32_t wl = (uint32_t)srow * lpitch + (uint32_t)scol * 16u;
    const uint32_t goff = mad24_s((uint32_t)srow, pitch, mul24_12((uint32_t)scol));
    const unsigned char* gsrc = a.src + (long long)f0 * a.src_img_stride + (size_t)((uint32_t)ymn * pitch + (uint32_t)xmn * 3u);   // uniform
    unsigned char* gdst = a.dst + (long long)f0 * a.dst_img_stride;                                                            // uniform
    const uint32_t doff = ((uint32_t)rr * (uint32_t)a.pitch_w + (uint32_t)c0p) * 3u;
    // Staging loads spelled in assembly: scalar row base + per-lane offset (no address VALU), the lane / pass predicate as an
    // exec mask INSIDE the statement.  Left to hipcc, five conditionally loaded 12-byte tuples that stay live across the blend
    // all land in ONE register triple and are copied out one by one behind an s_waitcnt vmcnt(0) each: no overlap at all.
    // The compiler does not count an asm statement's memory operations: `landed` below is the wait.
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    const unsigned long long smask = __ballot(sactive);     // uniform
    auto issue = [&](const unsigned char* gb, u3 (&v)[Win::PASSES]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < Win::PASSES; ++p) {
            const int r0 = min(p * Win::RPP, nrows - Win::RPP);
            const unsigned long long m = p * Win::RPP < nrows ? smask : 0ull;          // uniform
            const unsigned char* rowbase = gb + (size_t)((uint32_t)r0 * pitch);         // uniform
            unsigned long long keep;
#ifdef RWH_ABL_M_NOLOAD      // lab ablation hooks (never defined in the product build)
            v[p] = u3{goff + (uint32_t)(size_t)rowbase, goff * 3u, goff * 5u}; (void)keep; (void)m;
#else
#ifdef RWH_ABL_M_NOREPEAT    // lanes whose row the pass before already staged load nothing
            const unsigned long long mm = (p > 0 && r0 < p * Win::RPP) ? m & ~((1ull << ((p * Win::RPP - r0) * Win::LPRW)) - 1ull) : m;
            asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, %4\n\tglobal_load_dwordx3 %0, %2, %3\n\ts_mov_b64 exec, %1"
                         : "=&v"(v[p]), "=&s"(keep) : "v"(goff), "s"(rowbase), "s"(mm) : "memory");
#else
            asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, %4\n\tglobal_load_dwordx3 %0, %2, %3\n\ts_mov_b64 exec, %1"
                         : "=&v"(v[p]), "=&s"(keep) : "v"(goff), "s"(rowbase), "s"(m) : "memory");
#endif
#endif
        }
    };
    auto landed = [&](u3 (&v)[Win::PASSES]) __attribute__((always_inline)) {
        // every load issued so far has written its registers (the operands tie the wait to the values)
        if constexpr (Win::PASSES == 5) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]) : : "memory");
        else {
#pragma unroll
            for (int p = 0; p < Win::PASSES; ++p) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[p]) : : "memory");
        }
    };
    auto expand = [&](const u3 (&v)[Win::PASSES]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < Win::PASSES; ++p)
            if (p * Win::RPP < nrows && sactive) {
                const int r0 = min(p * Win::RPP, nrows - Win::RPP);
                uint4 t4;
                t4.x = v[p].x & 0xFFFFFFu;
                t4.y = __builtin_amdgcn_perm(v[p].y, v[p].x, 0x0C050403u);
                t4.z = __builtin_amdgcn_perm(v[p].z, v[p].y, 0x0C040302u);
                t4.w = v[p].z >> 8;
                // (8-byte aligned when the pitch is skewed: ds_write2_b64)
                __builtin_memcpy(__builtin_assume_aligned(my + (uint32_t)r0 * lpitch + wl, RWH_F8M_SKEW % 16 ? 8 : 16), &t4, 16);
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // Order at the start of a block: geometry first, then the first frame's chunks.  (Requesting them BEFORE the ~1500 cycles
    // of coordinate arithmetic -- RWH_F8M_EARLY1, with or without the second frame's right behind, RWH_F8M_EARLY2 -- was 2-4 %
    // SLOWER on 4K frames and 10 % on 8K frames: what this kernel waits for is the memory system's throughput, and a burst of
    // requests from every starting block makes that worse, not better.)
    u3 v[Win::PASSES], v1[Win::PASSES];
#ifdef RWH_F8M_EARLY1
    issue(gsrc, v1);
#ifdef RWH_F8M_EARLY2
    if (f0 + 1 < f1) issue(gsrc + a.src_img_stride, v);
#endif
#endif

    // ---- geometry of the lane's 8 pixels, once: tap weights and slab addresses ---------------------------------------------
    const uint32_t slab_off = (uint32_t)pwave * (uint32_t)SLAB;
    const uint32_t tap_c = ((uint32_t)hymn & 0xFFFFFFu) * lpitch + ((uint32_t)hxmn << 2) - slab_off;   // uniform
    float wa[F8_PX], wb[F8_PX], wc[F8_PX], wd[F8_PX];      // w00, w01, w10, w11
    uint32_t lo[F8_PX];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        double X[3], Y[3], W[3], rc[3];
        uint32_t lx[FP_PX], ly[FP_PX], hx[FP_PX], hy[FP_PX];
#pragma unroll
        for (int j = 0; j < 3; ++j) { X[j] = X0 + co.dxs8[3 * h + j][0]; Y[j] = Y0 + co.dxs8[3 * h + j][1]; W[j] = W0 + co.dxs8[3 * h + j][2]; }
        const double p12 = W[0] * W[1], P = p12 * W[2];
        double rp = __builtin_amdgcn_rcp(P);                  // staged => wpos: a finite, normal product of positive W
        rp = fma(fma(-P, rp, 1.0), rp, rp);
        const double r12 = rp * W[2];
        rc[2] = rp * p12; rc[0] = r12 * W[1]; rc[1] = r12 * W[0];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double ux = fma(X[j], rc[j], MAGIC), uy = fma(Y[j], rc[j], MAGIC);
            const int q = j + 1 - h;
            hx[q] = hi32(ux); lx[q] = lo32(ux); hy[q] = hi32(uy); ly[q] = lo32(uy);
        }
        hx[3 * h] = ehx[h]; lx[3 * h] = elx[h]; hy[3 * h] = ehy[h]; ly[3 * h] = ely[h];
#pragma unroll
        for (int j = 0; j < FP_PX; ++j) {
            const int p = 4 * h + j;
            const float wx1 = (float)lx[j], wy1 = (float)ly[j] * WS;
            const float w11 = wx1 * wy1;                     // blend4<FOLDED>'s derivation
            const float w01 = __builtin_fmaf(wx1, WO, -w11);
            const float w10 = __builtin_fmaf(wy1, 4294967296.0f, -w11);
            const float w00 = __builtin_fmaf(-wx1, WO, WC) - w10;
            wa[p] = w00; wb[p] = w01; wc[p] = w10; wd[p] = w11;
            lo[p] = mad24_s(hy[j], lpitch, shl2_add_s(hx[j], 0u - tap_c));
        }
    }

#ifdef RWH_F8M_EARLY1
    // (volatile statements keep their order: these pin the arithmetic above IN FRONT of the wait below -- left alone, hipcc
    //  sinks it behind the s_waitcnt, where it overlaps nothing)
#pragma unroll
    for (int p = 0; p < F8_PX; ++p) asm volatile("" : : "v"(wa[p]), "v"(wb[p]), "v"(wc[p]), "v"(wd[p]), "v"(lo[p]));
#else
    issue(gsrc, v1);
#endif
    landed(v1);
    expand(v1);
#ifndef RWH_F8M_EARLY2
    if (f0 + 1 < f1) issue(gsrc + a.src_img_stride, v);
#endif
#ifdef RWH_LAB_STAMPS       // lab build only: cycles per phase, summed over the frames, written over the first bytes of the output
    unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, t0 = __builtin_amdgcn_s_memtime(), t1;
#define RWH_STAMP(acc) { t1 = __builtin_amdgcn_s_memtime(); acc += t1 - t0; t0 = t1; }
#else
#define RWH_STAMP(acc)
#endif
    for (int f = f0; f < f1; ++f) {
        gsrc += a.src_img_stride;
        const bool more = f + 1 < f1;                         // uniform
        pk3 w[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float o[FP_PX][3];
#pragma unroll
            for (int j = 0; j < FP_PX; ++j) {
                const int p = 4 * h + j;
                const uint32_t* t0 = reinterpret_cast<const uint32_t*>(slab0 + lo[p]);
                const uint32_t* t1 = reinterpret_cast<const uint32_t*>(slab0 + lo[p] + lpitch);
                const uint32_t a0 = t0[0], b0 = t0[1], a1 = t1[0], b1 = t1[1];
                const float w00 = wa[p], w01 = wb[p], w10 = wc[p], w11 = wd[p];
                const uint32_t g00 = rg_halves(a0), g01 = rg_halves(b0), g10 = rg_halves(a1), g11 = rg_halves(b1);
                o[j][0] = fmix_lo(g11, w11, fmix_lo(g10, w10, fmix_lo(g01, w01, fmix_lo_c(g00, w00, U8_BIAS))));
                o[j][1] = fmix_hi(g11, w11, fmix_hi(g10, w10, fmix_hi(g01, w01, fmix_hi_c(g00, w00, U8_BIAS))));
                o[j][2] = fmix_hi(b1, w11, fmix_hi(a1, w10, fmix_hi(b0, w01, fmix_hi_c(a0, w00, U8_BIAS))));
            }
            w[h] = pack_run_u8(o);
        }
        // this frame's tap reads are complete (their values were consumed above; LDS is in order): the slab is free
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        RWH_STAMP(tA)
        if (more) { landed(v); RWH_STAMP(tB) expand(v); }
        RWH_STAMP(tC)
        unsigned char* drow = gdst + doff;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int first = tshift - (lcol + (PW / 2) * h);
#ifdef RWH_ABL_M_NOSTORE
            if (w[h].a + w[h].b == 0x12345u)
#endif
            store_run_pk(w[h], drow + 3 * (PW / 2) * h, store_any & (first <= 3), max(first, 0));
        }
        gdst += a.dst_img_stride;
        if (f + 2 < f1) issue(gsrc + a.src_img_stride, v);    // two frames ahead: in flight during the whole next blend
        RWH_STAMP(tD)
    }
#ifdef RWH_LAB_STAMPS
    if (lane == 0 && logical < 8192u) {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(a.dst) + ((size_t)logical * 4 + pwave) * 4;
        st[0] = tA; st[1] = tB; st[2] = tC; st[3] = tD;
    }
#endif
}
// ---- the same with ONE staging window per BLOCK (lab kernel, rwh_lab_tune(RWH_TUNE_WARP_FRAMES, 100 + n)) ---------------------
// tools/pattern_probe.hip (the warp's memory patterns without its arithmetic, 32 x 4K frames): loads of four wave-private
// windows + the 96-byte-segment stores 0.3147 ms -- a plain copy of the same bytes: 0.3022 --; ONE window per 128 x 16 tile
// (132 x 18 texels for 2048 pixels instead of 4 x 66 x 10) + the same stores 0.2793 ms.  The halo is what the load path pays for.
// Here the block's 256 threads stage the union of the four patches' footprints into one RGBX slab (fixed lane map: slot c = thread
// + 256 p -> row c / BCH, chunk c % BCH), the waves take their taps from it; two block barriers per frame (slab free / slab ready),
// which the multi-frame loop can afford: its per-frame arithmetic is half the one-frame kernel's.  Blocks with a patch that is not
// strictly interior, or whose union does not fit BROWS x BCH chunks, run the one-frame body frame by frame (all four waves).
constexpr int BCH = 36, BROWS = 22;                     // block window: 144 texels x 22 rows
// block barrier that orders LDS traffic only (__syncthreads' fence also waits for every outstanding global load and store:
// the prefetched chunks of the next frame, the stores of the previous one)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory"); }
template <int LOG_PW>
__device__ __forceinline__ void fast8mb_body(const FastArgs& a) {
    constexpr int PW = 1 << LOG_PW, PH = 512 / PW, LPR = PW / 8, WX = 128 / PW;
    using Win = F8Window<LOG_PW>;
    constexpr uint32_t lpitch = BCH * 16 + RWH_F8M_SKEW;
    constexpr int SLABW = ((Win::ROWS * (int)(Win::LPITCH + RWH_F8M_SKEW) + 15) / 16) * 16;      // per wave, as fast8m_body (the fallback's)
    static_assert(4 * SLABW >= BROWS * (int)lpitch + 128, "the block window lives in the four waves' slabs");
    constexpr float WS = W_SCALE * MIX_S, WO = W_ONE * MIX_S, WC = MIX_S;
    unsigned char* const slab0 = wave_slabs<SLABW>();

    const unsigned b = blockIdx.x;
    const unsigned logical = (b & 7u) * a.mf_cpx + (b >> 3);
    if (logical >= a.mf_nblocks) return;                                             // (block-uniform)
    const unsigned g = a.ntiles_magic ? __umulhi(logical, a.ntiles_magic) : (a.ntiles == 1u ? logical : 0u);
    const unsigned ti = logical - g * a.ntiles;
    const int f0 = (int)g * a.mf_frames, f1 = min(f0 + a.mf_frames, a.mf_batch);
    const unsigned ty = a.tiles_x_magic ? __umulhi(ti, a.tiles_x_magic) : ti;
    const unsigned tx = ti - ty * a.tiles_x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const Coef& co = a.c;
    const int lane = threadIdx.x & 63;
    const int prow = lane / LPR, pq = lane % LPR;
    const int wave_x = (wave % WX) * PW, wave_y = (wave / WX) * PH;
    const int rr_raw = (int)ty * 16 + wave_y + prow;
    const int rr = min(rr_raw, a.rows - 1);
    const int tcol0 = (int)tx * 128;
    const int tcol = min(tcol0, a.out_w - 128);
    const int tshift = tcol0 - tcol;
    const int lcol = wave_x + pq * 4;
    const int c0p = tcol + lcol;
    const bool store_any = rr_raw < a.rows;
    const uint32_t pitch = (uint32_t)a.src_w * 3u;

    // ---- the lane's end pixels and the wave's footprint: fast8_body's arithmetic ---------------------------------------------
    const double fr = (double)(a.row_begin + rr), fc = (double)c0p;
    const double X0 = fma(fc, co.cx[2], fma(fr, co.cx[1], co.cx[0]));
    const double Y0 = fma(fc, co.cy[2], fma(fr, co.cy[1], co.cy[0]));
    const double W0 = fma(fc, co.cw[2], fma(fr, co.cw[1], co.cw[0]));
    uint32_t ehx[2], ehy[2], elx[2], ely[2];
    bool wpos;
    {
        const double X7 = X0 + co.dxs8[6][0], Y7 = Y0 + co.dxs8[6][1], W7 = W0 + co.dxs8[6][2];
        double r0 = __builtin_amdgcn_rcp(W0); r0 = fma(fma(-W0, r0, 1.0), r0, r0);
        double r7 = __builtin_amdgcn_rcp(W7); r7 = fma(fma(-W7, r7, 1.0), r7, r7);
        const double ux0 = fma(X0, r0, MAGIC), uy0 = fma(Y0, r0, MAGIC), ux7 = fma(X7, r7, MAGIC), uy7 = fma(Y7, r7, MAGIC);
        ehx[0] = hi32(ux0); elx[0] = lo32(ux0); ehy[0] = hi32(uy0); ely[0] = lo32(uy0);
        ehx[1] = hi32(ux7); elx[1] = lo32(ux7); ehy[1] = hi32(uy7); ely[1] = lo32(uy7);
        const int h0 = (int)hi32(W0), h7 = (int)hi32(W7);
        wpos = __all((int)(h0 > 0x2D300000) & (int)(h0 < 0x52B00000) & (int)(h7 > 0x2D300000) & (int)(h7 < 0x52B00000));
    }
    const int x0 = (int)__builtin_amdgcn_readlane(ehx[0], 0), x1 = (int)__builtin_amdgcn_readlane(ehx[1], LPR - 1);
    const int x2 = (int)__builtin_amdgcn_readlane(ehx[0], 64 - LPR), x3 = (int)__builtin_amdgcn_readlane(ehx[1], 63);
    const int y0 = (int)__builtin_amdgcn_readlane(ehy[0], 0), y1 = (int)__builtin_amdgcn_readlane(ehy[1], LPR - 1);
    const int y2 = (int)__builtin_amdgcn_readlane(ehy[0], 64 - LPR), y3 = (int)__builtin_amdgcn_readlane(ehy[1], 63);
    const int whxmn = smin(smin(x0, x1), smin(x2, x3)) & ~3, whxmx = smax(smax(x0, x1), smax(x2, x3));
    const int whymn = smin(smin(y0, y1), smin(y2, y3)), whymx = smax(smax(y0, y1), smax(y2, y3));
    {
        const int xmn = (int)((uint32_t)smax(whxmn, 0) - MAGIC_HI), xmx = (int)((uint32_t)smax(whxmx, 0) - MAGIC_HI);
        const int ymn = (int)((uint32_t)smax(whymn, 0) - MAGIC_HI), ymx = (int)((uint32_t)smax(whymx, 0) - MAGIC_HI);
        const bool interior = wpos & (xmn >= 0) & (xmx < a.bound_w - 1) & (ymn >= 0) & (ymx < min(a.bound_h - 1, a.src_h - 2));
        // ---- the block's window = the union of its four patches' footprints (through LDS; block-uniform afterwards) -------------
        int* red = reinterpret_cast<int*>(slab0);
        if (lane == 0) { red[5 * wave] = whxmn; red[5 * wave + 1] = whxmx; red[5 * wave + 2] = whymn; red[5 * wave + 3] = whymx; red[5 * wave + 4] = interior ? 1 : 0; }
    }
    __syncthreads();
    int hxmn, hxmx, hymn, hymx, all_in;
    {
        const int* red = reinterpret_cast<const int*>(slab0);
        hxmn = min(min(red[0], red[5]), min(red[10], red[15])); hxmx = max(max(red[1], red[6]), max(red[11], red[16]));
        hymn = min(min(red[2], red[7]), min(red[12], red[17])); hymx = max(max(red[3], red[8]), max(red[13], red[18]));
        all_in = red[4] & red[9] & red[14] & red[19];
        hxmn = __builtin_amdgcn_readfirstlane(hxmn); hxmx = __builtin_amdgcn_readfirstlane(hxmx);
        hymn = __builtin_amdgcn_readfirstlane(hymn); hymx = __builtin_amdgcn_readfirstlane(hymx);
        all_in = __builtin_amdgcn_readfirstlane(all_in);
    }
    __syncthreads();                                           // (the scratch words are slab bytes: read by all before anyone stages)
    const int xmn = (int)((uint32_t)hxmn - MAGIC_HI), xmx = (int)((uint32_t)hxmx - MAGIC_HI);          // all_in: the hi dwords are >= MAGIC_HI
    const int ymn = (int)((uint32_t)hymn - MAGIC_HI), ymx = (int)((uint32_t)hymx - MAGIC_HI);
    const int nrows = ymx - ymn + 2, C = (xmx - xmn + 5) >> 2;
    if (!(all_in && nrows <= BROWS && C <= BCH)) {            // block-uniform: the one-frame body, frame by frame, every wave its patch
        const int own_from = (int)tx * 128 - min((int)tx * 128, a.out_w - 128);
        if ((wave % WX + 1) * PW <= own_from) return;
        for (int f = f0; f < f1; ++f)
            fast8_body<unsigned char, LOG_PW, false, 3, false, true>(a, nullptr, nullptr, (int)((unsigned)f * a.ntiles + ti), slab0 + wave * SLABW);
        return;
    }

    // ---- staging slots of this thread: c = thread + 256 p -> (row c / BCH, chunk c % BCH) of the block window ---------------
    constexpr int SLOTS = (BROWS * BCH + 255) / 256;
    uint32_t goff[SLOTS], wl[SLOTS];
    unsigned long long smask[SLOTS];
    bool sact[SLOTS];
#pragma unroll
    for (int p = 0; p < SLOTS; ++p) {
        const int c = (int)threadIdx.x + 256 * p, r = c / BCH, k = c - r * BCH;
        sact[p] = (r < nrows) & (k < C);
        goff[p] = mad24_s((uint32_t)r, pitch, mul24_12((uint32_t)k));
        wl[p] = (uint32_t)r * lpitch + (uint32_t)k * 16u;
        smask[p] = __ballot(sact[p]);
    }
    const unsigned char* gsrc = a.src + (long long)f0 * a.src_img_stride + (size_t)((uint32_t)ymn * pitch + (uint32_t)xmn * 3u);   // uniform
    unsigned char* gdst = a.dst + (long long)f0 * a.dst_img_stride;
    const uint32_t doff = ((uint32_t)rr * (uint32_t)a.pitch_w + (uint32_t)c0p) * 3u;
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    auto issue = [&](const unsigned char* gb, u3 (&v)[SLOTS]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < SLOTS; ++p) {
            unsigned long long keep;
            asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, %4\n\tglobal_load_dwordx3 %0, %2, %3\n\ts_mov_b64 exec, %1"
                         : "=&v"(v[p]), "=&s"(keep) : "v"(goff[p]), "s"(gb), "s"(smask[p]) : "memory");
        }
    };
    auto landed = [&](u3 (&v)[SLOTS]) __attribute__((always_inline)) {
        static_assert(SLOTS == 4, "operand list below");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : : "memory");
    };
    auto expand = [&](const u3 (&v)[SLOTS]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < SLOTS; ++p)
            if (sact[p]) {
                uint4 t4;
                t4.x = v[p].x & 0xFFFFFFu;
                t4.y = __builtin_amdgcn_perm(v[p].y, v[p].x, 0x0C050403u);
                t4.z = __builtin_amdgcn_perm(v[p].z, v[p].y, 0x0C040302u);
                t4.w = v[p].z >> 8;
                __builtin_memcpy(__builtin_assume_aligned(slab0 + wl[p], RWH_F8M_SKEW % 16 ? 8 : 16), &t4, 16);
            }
    };

    // ---- geometry of the lane's 8 pixels, once (fast8m_body) ---------------------------------------------------------------
    const uint32_t tap_c = ((uint32_t)hymn & 0xFFFFFFu) * lpitch + ((uint32_t)hxmn << 2);   // uniform; the block slab starts at LDS offset of slab0
    float wa[F8_PX], wb[F8_PX], wc[F8_PX], wd[F8_PX];
    uint32_t lo[F8_PX];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        double X[3], Y[3], W[3], rc[3];
        uint32_t lx[FP_PX], ly[FP_PX], hx[FP_PX], hy[FP_PX];
#pragma unroll
        for (int j = 0; j < 3; ++j) { X[j] = X0 + co.dxs8[3 * h + j][0]; Y[j] = Y0 + co.dxs8[3 * h + j][1]; W[j] = W0 + co.dxs8[3 * h + j][2]; }
        const double p12 = W[0] * W[1], P = p12 * W[2];
        double rp = __builtin_amdgcn_rcp(P);
        rp = fma(fma(-P, rp, 1.0), rp, rp);
        const double r12 = rp * W[2];
        rc[2] = rp * p12; rc[0] = r12 * W[1]; rc[1] = r12 * W[0];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double ux = fma(X[j], rc[j], MAGIC), uy = fma(Y[j], rc[j], MAGIC);
            const int q = j + 1 - h;
            hx[q] = hi32(ux); lx[q] = lo32(ux); hy[q] = hi32(uy); ly[q] = lo32(uy);
        }
        hx[3 * h] = ehx[h]; lx[3 * h] = elx[h]; hy[3 * h] = ehy[h]; ly[3 * h] = ely[h];
#pragma unroll
        for (int j = 0; j < FP_PX; ++j) {
            const int p = 4 * h + j;
            const float wx1 = (float)lx[j], wy1 = (float)ly[j] * WS;
            const float w11 = wx1 * wy1;
            const float w01 = __builtin_fmaf(wx1, WO, -w11);
            const float w10 = __builtin_fmaf(wy1, 4294967296.0f, -w11);
            const float w00 = __builtin_fmaf(-wx1, WO, WC) - w10;
            wa[p] = w00; wb[p] = w01; wc[p] = w10; wd[p] = w11;
            lo[p] = mad24_s(hy[j], lpitch, shl2_add_s(hx[j], 0u - tap_c));
        }
    }
    u3 v[SLOTS];
    issue(gsrc, v);
    landed(v);
    expand(v);
    lds_barrier();                                             // the slab holds frame f0
    if (f0 + 1 < f1) issue(gsrc + a.src_img_stride, v);
    for (int f = f0; f < f1; ++f) {
        gsrc += a.src_img_stride;
        const bool more = f + 1 < f1;                         // block-uniform
        pk3 w[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float o[FP_PX][3];
#pragma unroll
            for (int j = 0; j < FP_PX; ++j) {
                const int p = 4 * h + j;
                const uint32_t* t0 = reinterpret_cast<const uint32_t*>(slab0 + lo[p]);
                const uint32_t* t1 = reinterpret_cast<const uint32_t*>(slab0 + lo[p] + lpitch);
                const uint32_t a0 = t0[0], b0 = t0[1], a1 = t1[0], b1 = t1[1];
                const float w00 = wa[p], w01 = wb[p], w10 = wc[p], w11 = wd[p];
                const uint32_t g00 = rg_halves(a0), g01 = rg_halves(b0), g10 = rg_halves(a1), g11 = rg_halves(b1);
                o[j][0] = fmix_lo(g11, w11, fmix_lo(g10, w10, fmix_lo(g01, w01, fmix_lo_c(g00, w00, U8_BIAS))));
                o[j][1] = fmix_hi(g11, w11, fmix_hi(g10, w10, fmix_hi(g01, w01, fmix_hi_c(g00, w00, U8_BIAS))));
                o[j][2] = fmix_hi(b1, w11, fmix_hi(a1, w10, fmix_hi(b0, w01, fmix_hi_c(a0, w00, U8_BIAS))));
            }
            w[h] = pack_run_u8(o);
        }
        if (more) {
            lds_barrier();                                     // every wave has taken its taps of frame f: the slab is free
            landed(v);
            expand(v);
            lds_barrier();                                     // the slab holds frame f + 1
        }
        unsigned char* drow = gdst + doff;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int first = tshift - (lcol + (PW / 2) * h);
            store_run_pk(w[h], drow + 3 * (PW / 2) * h, store_any & (first <= 3), max(first, 0));
        }
        gdst += a.dst_img_stride;
        if (f + 2 < f1) issue(gsrc + a.src_img_stride, v);
    }
}
template <int LOG_PW>
__global__ __launch_bounds__(256, 4) void warp_rgb8_fast8mb(const FastArgs a) { fast8mb_body<LOG_PW>(a); }

#ifndef RWH_F8M_WAVES
#define RWH_F8M_WAVES 4
#endif
template <int LOG_PW>
__global__ __launch_bounds__(256, RWH_F8M_WAVES) void warp_rgb8_fast8m(const FastArgs a) { fast8m_body<LOG_PW>(a); }

// ---- a thin ragged right edge (round 4) -----------------------------------------------------------------------------------
// An output row of 1921 pixels (a 1080p frame's auto-bounds grid) is 15 tiles of 128 columns and ONE column: the tile that is
// moved left to cover it recomputes 127 columns it does not own -- 4.6 % of the 512 x 1080p batch (profiles/r03_lab_notes.txt
// section 13).  When 1 <= out_w mod 128 <= STRIP_MAX the tiled launch covers the multiple of 128 and this kernel the rest: one
// lane = one output ROW, all its strip pixels; coordinates, validity, weights and the fused multiply-add chain of the blend as in the
// tiled kernels (per pixel, not per run: the strip is 0.05-0.8 % of a frame), byte-exact clamped taps.  Which columns are strip
// columns depends on out_w only, so whole launches, row shards, batches and per-image tables still agree bit for bit.
#ifndef RWH_STRIP_MAX
#define RWH_STRIP_MAX 16      // (lab builds: 0 switches the strip off)
#endif
constexpr int STRIP_MAX = RWH_STRIP_MAX;
template <typename DstT>
__device__ __forceinline__ void strip_body(const FastArgs& a, const Coef* tab) {
    constexpr bool U8 = sizeof(DstT) == 1;
    constexpr float WS = W_SCALE * MIX_S, WO = W_ONE * MIX_S, WC = MIX_S, BIAS = U8 ? U8_BIAS : 0.f;
    const int rr = (int)(blockIdx.x * 256 + threadIdx.x);
    const unsigned img = blockIdx.y;
    if (rr >= a.rows) return;
    const Coef& co = tab ? tab[img] : a.c;
    const unsigned img_mem = tab ? (unsigned)co.image : img;
    const unsigned char* simg = a.src + (long long)img_mem * a.src_img_stride;
    DstT* drow = reinterpret_cast<DstT*>(a.dst + (long long)img_mem * a.dst_img_stride) + ((size_t)rr * (size_t)a.pitch_w) * 3;
    const uint32_t pitch = (uint32_t)a.src_w * 3u, last = (uint32_t)a.src_h * pitch - 3u;
    const double fr = (double)(a.row_begin + rr);
    for (int c = a.out_w; c < a.pitch_w; ++c) {
        const double fc = (double)c;
        const double X = fma(fc, co.cx[2], fma(fr, co.cx[1], co.cx[0]));
        const double Y = fma(fc, co.cy[2], fma(fr, co.cy[1], co.cy[0]));
        const double W = fma(fc, co.cw[2], fma(fr, co.cw[1], co.cw[0]));
        double r = __builtin_amdgcn_rcp(W);
        r = fma(fma(-W, r, 1.0), r, r);
        const double ux = fma(X, r, MAGIC), uy = fma(Y, r, MAGIC);
        const unsigned long long ubx = (unsigned long long)__double_as_longlong(ux), uby = (unsigned long long)__double_as_longlong(uy);
        const bool valid = (ubx >= MAGIC_BITS) & (ubx <= a.xmax_bits) & (uby >= MAGIC_BITS) & (uby <= a.ymax_bits);
        float o[3] = {BIAS, BIAS, BIAS};
        if (valid) {
            const int ix = (int)(hi32(ux) - MAGIC_HI), iy = (int)(hi32(uy) - MAGIC_HI);
            const float wx1 = (float)lo32(ux), wy1 = (float)lo32(uy) * WS;
            const float w11 = wx1 * wy1;                                   // blend4<FOLDED>'s derivation
            const float w01 = __builtin_fmaf(wx1, WO, -w11);
            const float w10 = __builtin_fmaf(wy1, 4294967296.0f, -w11);
            const float w00 = __builtin_fmaf(-wx1, WO, WC) - w10;
            const uint32_t o00 = (uint32_t)iy * pitch + (uint32_t)ix * 3u;
            const uint32_t o01 = min(o00 + 3u, last), o10 = min(o00 + pitch, last), o11 = min(o00 + pitch + 3u, last);   // (weight 0 when clamped)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                // a byte as the float16 denormal b * 2^-24 times a weight that carries the 2^24: the tiled kernels' v_fma_mix_f32 terms
                const float p00 = (float)simg[o00 + k] * 0x1p-24f, p01 = (float)simg[o01 + k] * 0x1p-24f;
                const float p10 = (float)simg[o10 + k] * 0x1p-24f, p11 = (float)simg[o11 + k] * 0x1p-24f;
                o[k] = __builtin_fmaf(p11, w11, __builtin_fmaf(p10, w10, __builtin_fmaf(p01, w01, __builtin_fmaf(p00, w00, BIAS))));
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if constexpr (U8) drow[3 * c + k] = (unsigned char)__builtin_amdgcn_cvt_pk_u8_f32(o[k], 0, 0);
            else drow[3 * c + k] = o[k];
        }
    }
}
template <typename DstT>
__global__ __launch_bounds__(256) void warp_rgb8_strip(const FastArgs a) { strip_body<DstT>(a, nullptr); }
template <typename DstT>
__global__ __launch_bounds__(256) void warp_rgb8_strip_tab(const FastArgs a, const CoefTab t) { strip_body<DstT>(a, t.e); }

// ---- nearest neighbour, RGB u8 (homography.py:108-121), bit-exact ---------------------------------------------------
// Same tiling, patch shapes, footprint staging and store layout as warp_rgb8_fast8; one LDS read per pixel, no blend.
// The pixel index trunc(s + 0.5) is an INTEGER result, so "within 2^-32 px" is not good enough near a tie:
//   * interior waves take floor(s + 0.5) from the magic number (coordinates good to ~3e-10 px) and flag every pixel whose
//     fraction lies within 2^-20 px of a rounding boundary (2e-6 of all pixels); flagged pixels are recomputed with the
//     reference's own formula -- numpy.linspace grid, dgemm-order products, IEEE float64 divide, (int)(s + 0.5), the
//     one warp_exact uses -- and fetched from global memory;
//   * border / horizon / oversize waves use that formula for every pixel (negative coordinates truncate towards zero
//     there, which floor() does not reproduce).
constexpr uint32_t NN_TIE = 1u << 12;                   // 2^-20 px in units of 2^-32

__device__ __forceinline__ uint32_t nn_exact_texel(const FastArgs& a, const double (&ih)[9], const unsigned char* simg, int r, int c) {
    const double x = (c == a.out_w - 1) ? a.gx_last : (double)c * a.gstep_x + a.gx0;
    const double y = (r == a.out_h - 1) ? a.gy_last : (double)r * a.gstep_y + a.gy0;
    const double X = fma(ih[1], y, ih[0] * x) + ih[2];
    const double Y = fma(ih[4], y, ih[3] * x) + ih[5];
    const double W = fma(ih[7], y, ih[6] * x) + ih[8];
    const double sx = X / W, sy = Y / W;
    const int xi = (int)(sx + 0.5), yi = (int)(sy + 0.5);
    // (numpy's astype(int32) turns a NaN into INT_MIN, which the reference masks; the GPU's conversion gives 0: masked explicitly)
    if (!((xi >= 0) & (xi <= a.bound_w - 1) & (yi >= 0) & (yi <= a.bound_h - 1) & (sx == sx) & (sy == sy))) return 0u;   // -> texel (0,0), blanked
    const uint32_t off = ((uint32_t)yi * (uint32_t)a.src_w + (uint32_t)xi) * 3u;
    const uint32_t img_bytes = (uint32_t)a.src_h * (uint32_t)a.src_w * 3u;
    if (off + 4u <= img_bytes) return ld4(simg + off) & 0xFFFFFFu;
    return simg[off] | (simg[off + 1] << 8) | (simg[off + 2] << 16);
}

// 4 RGBX texels -> 12 packed bytes, stored whole or (ragged row) from local pixel `first` on
__device__ __forceinline__ void nn_store(const uint32_t (&t)[FP_PX], unsigned char* drow, bool store_any, int first) {
    if (!store_any) return;
    if (first <= 0) {
        pk3 w;
        w.a = (t[0] & 0xFFFFFFu) | (t[1] << 24);
        w.b = ((t[1] >> 8) & 0xFFFFu) | (t[2] << 16);
        w.c = ((t[2] >> 16) & 0xFFu) | (t[3] << 8);
        __builtin_memcpy(drow, &w, 12);
    } else {
#pragma unroll
        for (int j = 1; j < FP_PX; ++j)
            if (j >= first) { drow[3 * j] = (unsigned char)t[j]; drow[3 * j + 1] = (unsigned char)(t[j] >> 8); drow[3 * j + 2] = (unsigned char)(t[j] >> 16); }
    }
}

template <int LOG_PW>
__device__ __forceinline__ void nn_body(const FastArgs& a, const Coef* tab) {
    constexpr int PW = 1 << LOG_PW, PH = 512 / PW, LPR = PW / 8, WX = 128 / PW;
    constexpr double MAGIC_R = MAGIC + 0.5;                 // hi(s + MAGIC_R) - MAGIC_HI = floor(s + 0.5)
    __shared__ __attribute__((aligned(16))) unsigned char slab[4][F8Window<LOG_PW>::SLAB];

    const unsigned b = blockIdx.x;
    const unsigned logical = (b & 7u) * a.cpx + (b >> 3);
    if (logical >= a.nblocks) return;
    const unsigned t = a.tiles_x_magic ? __umulhi(logical, a.tiles_x_magic) : logical;
    const unsigned tx = logical - t * a.tiles_x;
    const unsigned img = a.tiles_y_magic ? __umulhi(t, a.tiles_y_magic) : t;
    const unsigned ty = t - img * a.tiles_y;
    const Coef& co = tab ? tab[img] : a.c;
    const unsigned img_mem = tab ? (unsigned)co.image : img;
    double ih9[9];                                          // by value: a reference into the indexed table makes hipcc
#pragma unroll                                              // copy the whole table to scratch
    for (int i = 0; i < 9; ++i) ih9[i] = co.ih[i];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int prow = lane / LPR, pq = lane % LPR;
    const int wave_x = (wave % WX) * PW, wave_y = (wave / WX) * PH;
    const int rr_raw = (int)ty * 16 + wave_y + prow;
    const int rr = min(rr_raw, a.rows - 1);
    const int tcol0 = (int)tx * 128, tcol = min(tcol0, a.out_w - 128), tshift = tcol0 - tcol;
    if ((wave_x + PW <= tshift) | ((int)ty * 16 + wave_y >= a.rows)) return;    // the patch owns no pixel (see fast8_body)
    const int lcol = wave_x + pq * 4, c0p = tcol + lcol;
    const bool store_any = rr_raw < a.rows;
    const unsigned char* simg = a.src + (long long)img_mem * a.src_img_stride;
    unsigned char* drow = a.dst + (long long)img_mem * a.dst_img_stride + ((uint32_t)rr * (uint32_t)a.pitch_w + (uint32_t)c0p) * 3u;
    const uint32_t pitch = (uint32_t)a.src_w * 3u;
    const int out_row = a.row_begin + rr;

    const double fr = (double)out_row, fc = (double)c0p;
    const double X0 = fma(fc, co.cx[2], fma(fr, co.cx[1], co.cx[0]));
    const double Y0 = fma(fc, co.cy[2], fma(fr, co.cy[1], co.cy[0]));
    const double W0 = fma(fc, co.cw[2], fma(fr, co.cw[1], co.cw[0]));
    uint32_t ehx[2], ehy[2], elx[2], ely[2];
    bool wpos;
    {
        const double X7 = X0 + co.dxs8[6][0], Y7 = Y0 + co.dxs8[6][1], W7 = W0 + co.dxs8[6][2];
        double r0 = __builtin_amdgcn_rcp(W0); r0 = fma(fma(-W0, r0, 1.0), r0, r0);
        double r7 = __builtin_amdgcn_rcp(W7); r7 = fma(fma(-W7, r7, 1.0), r7, r7);
        const double ux0 = fma(X0, r0, MAGIC_R), uy0 = fma(Y0, r0, MAGIC_R), ux7 = fma(X7, r7, MAGIC_R), uy7 = fma(Y7, r7, MAGIC_R);
        ehx[0] = hi32(ux0); elx[0] = lo32(ux0); ehy[0] = hi32(uy0); ely[0] = lo32(uy0);
        ehx[1] = hi32(ux7); elx[1] = lo32(ux7); ehy[1] = hi32(uy7); ely[1] = lo32(uy7);
        const int h0 = (int)hi32(W0), h7 = (int)hi32(W7);
        wpos = __all((int)(h0 > 0x2D300000) & (int)(h0 < 0x52B00000) & (int)(h7 > 0x2D300000) & (int)(h7 < 0x52B00000));
    }
    uint32_t lx[FP_PX], ly[FP_PX], hx[FP_PX], hy[FP_PX];
    auto run_coords = [&](const int h) {                    // only called when wpos holds (staged waves)
        double X[3], Y[3], W[3], rc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { X[j] = X0 + co.dxs8[3 * h + j][0]; Y[j] = Y0 + co.dxs8[3 * h + j][1]; W[j] = W0 + co.dxs8[3 * h + j][2]; }
        const double p12 = W[0] * W[1], P = p12 * W[2];
        double rp = __builtin_amdgcn_rcp(P);
        rp = fma(fma(-P, rp, 1.0), rp, rp);
        const double r12 = rp * W[2];
        rc[2] = rp * p12; rc[0] = r12 * W[1]; rc[1] = r12 * W[0];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double ux = fma(X[j], rc[j], MAGIC_R), uy = fma(Y[j], rc[j], MAGIC_R);
            const int q = j + 1 - h;
            hx[q] = hi32(ux); lx[q] = lo32(ux); hy[q] = hi32(uy); ly[q] = lo32(uy);
        }
        hx[3 * h] = ehx[h]; lx[3 * h] = elx[h]; hy[3 * h] = ehy[h]; ly[3 * h] = ely[h];
    };

    const int x0 = (int)__builtin_amdgcn_readlane(ehx[0], 0), x1 = (int)__builtin_amdgcn_readlane(ehx[1], LPR - 1);
    const int x2 = (int)__builtin_amdgcn_readlane(ehx[0], 64 - LPR), x3 = (int)__builtin_amdgcn_readlane(ehx[1], 63);
    const int y0 = (int)__builtin_amdgcn_readlane(ehy[0], 0), y1 = (int)__builtin_amdgcn_readlane(ehy[1], LPR - 1);
    const int y2 = (int)__builtin_amdgcn_readlane(ehy[0], 64 - LPR), y3 = (int)__builtin_amdgcn_readlane(ehy[1], 63);
    const int hxmn = smin(smin(x0, x1), smin(x2, x3)) & ~3, hxmx = smax(smax(x0, x1), smax(x2, x3));   // window from a multiple of 4 texels
    const int hymn = smin(smin(y0, y1), smin(y2, y3)), hymx = smax(smax(y0, y1), smax(y2, y3));
    const int xmn = (int)((uint32_t)smax(hxmn, 0) - MAGIC_HI), xmx = (int)((uint32_t)smax(hxmx, 0) - MAGIC_HI);   // (negative hi dwords:
    const int ymn = (int)((uint32_t)smax(hymn, 0) - MAGIC_HI), ymx = (int)((uint32_t)smax(hymx, 0) - MAGIC_HI);   //  see fast8_body)
    // footprint rows ymn..ymx, texels xmn..xmx (a corner within NN_TIE of a boundary may really round one further:
    // such pixels never use the slab); one row of slack below for the 9 bytes a chunk may read past its last texel
    using Win = F8Window<LOG_PW>;
    const int nrows = ymx - ymn + 1, C = (xmx - xmn + 4) >> 2;
    const bool staged = wpos & (xmn >= 0) & (xmx <= a.bound_w - 1) & (ymn >= 0) & (ymx <= min(a.bound_h - 1, a.src_h - 2)) &
                        (nrows >= Win::RPP) & (nrows <= Win::ROWS) & (C <= Win::LPRW);

    // the whole patch rounds to indices a full texel or more outside the source (W > 0: every pixel lies in the corners' box;
    // the margin of one texel covers the 3e-10 px between the magic-number rounding and the reference's own formula): zeros,
    // without the per-pixel IEEE divisions of the fallback below (6.4 % of an 8K frame's patches with the bench homography)
    if (wpos & ((xmx <= -3) | (xmn >= a.bound_w + 1) | (ymx <= -3) | (ymn >= a.bound_h + 1))) {
        const uint32_t zero[FP_PX] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int first = tshift - (lcol + (PW / 2) * h);
            nn_store(zero, drow + 3 * (PW / 2) * h, store_any & (first < 4), first);
        }
        return;
    }
    if (!staged) {                                          // every pixel by the reference's formula
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t tex[FP_PX];
#pragma unroll
            for (int j = 0; j < FP_PX; ++j) tex[j] = nn_exact_texel(a, ih9, simg, out_row, c0p + (PW / 2) * h + j);
            const int first = tshift - (lcol + (PW / 2) * h);
            nn_store(tex, drow + 3 * (PW / 2) * h, store_any & (first < 4), first);
        }
        return;
    }

    // staging: the fixed lane map of warp_rgb8_fast8 (see F8Window)
    pk3 v[Win::PASSES];
    const int sl = min(lane, Win::RPP * Win::LPRW - 1);
    const int srow = sl / Win::LPRW, scol = sl - srow * Win::LPRW;
    const uint32_t wl = (uint32_t)(srow * Win::LPITCH + scol * 16);
    {
        const unsigned char* gbase = simg + (size_t)((uint32_t)ymn * pitch + (uint32_t)xmn * 3u);
        const uint32_t goff = mad24_s((uint32_t)srow, pitch, mul24_12((uint32_t)min(scol, C - 1)));
#pragma unroll
        for (int p = 0; p < Win::PASSES; ++p) {
            if (p * Win::RPP < nrows) {
                const int r0 = min(p * Win::RPP, nrows - Win::RPP);
                __builtin_memcpy(&v[p], gbase + (size_t)((uint32_t)r0 * pitch) + goff, 12);
            }
        }
    }
    unsigned char* my = slab[wave];
    constexpr uint32_t lpitch = Win::LPITCH;
#pragma unroll
    for (int p = 0; p < Win::PASSES; ++p) {
        if (p * Win::RPP < nrows) {
            const int r0 = min(p * Win::RPP, nrows - Win::RPP);
            uint4 t4;
            t4.x = v[p].a;
            t4.y = __builtin_amdgcn_alignbyte(v[p].b, v[p].a, 3);
            t4.z = __builtin_amdgcn_alignbyte(v[p].c, v[p].b, 2);
            t4.w = v[p].c >> 8;
            *reinterpret_cast<uint4*>(my + (uint32_t)r0 * lpitch + wl) = t4;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t slab_off = (uint32_t)wave * (uint32_t)Win::SLAB;
    const uint32_t tap_c = ((uint32_t)hymn & 0xFFFFFFu) * lpitch + ((uint32_t)hxmn << 2) - slab_off;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        run_coords(h);
        uint32_t tex[FP_PX];
        bool tie[FP_PX], any_tie = false;
#pragma unroll
        for (int j = 0; j < FP_PX; ++j) {
            // fraction within NN_TIE of 0 (mod 1): lo + NN_TIE wraps below 2*NN_TIE
            tie[j] = (lx[j] + NN_TIE < 2u * NN_TIE) | (ly[j] + NN_TIE < 2u * NN_TIE);
            any_tie |= tie[j];
            // a tie pixel's rounded index may lie one outside the footprint: keep its (unused) slab read in range
            const uint32_t lo = tie[j] ? slab_off : mad24_s(hy[j], lpitch, shl2_add_s(hx[j], 0u - tap_c));
            tex[j] = *reinterpret_cast<const uint32_t*>(&slab[0][0] + lo);
        }
        if (__any(any_tie)) {
#pragma unroll
            for (int j = 0; j < FP_PX; ++j)
                if (tie[j]) tex[j] = nn_exact_texel(a, ih9, simg, out_row, c0p + (PW / 2) * h + j);
        }
        const int first = tshift - (lcol + (PW / 2) * h);
        nn_store(tex, drow + 3 * (PW / 2) * h, store_any & (first < 4), first);
    }
}

template <int LOG_PW>
__global__ __launch_bounds__(256) void warp_rgb8_nn(const FastArgs a) { nn_body<LOG_PW>(a, nullptr); }
template <int LOG_PW>
__global__ __launch_bounds__(256) void warp_rgb8_nn_tab(const FastArgs a, const CoefTab t) { nn_body<LOG_PW>(a, t.e); }

// Source footprint of a pw x ph output patch whose top-left pixel is (row r, column c) of the output grid: its rows and
// texels (what the shape's staging window has to hold, f8_window_fits) and an estimate of the 128-byte lines its staging
// loads touch.  Host-side twin of the kernel's footprint arithmetic, used only to choose the patch shape.
// false: W <= 0 at a corner.
inline bool patch_footprint(const FastArgs& a, double r, double c, int pw, int ph, long long* rows, long long* texels, double* lines) {
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    for (int k = 0; k < 4; ++k) {
        const double rr = r + (k & 2 ? ph - 1 : 0), cc = c + (k & 1 ? pw - 1 : 0);   // rows of the whole output grid
        const double X = a.c.cx[0] + rr * a.c.cx[1] + cc * a.c.cx[2], Y = a.c.cy[0] + rr * a.c.cy[1] + cc * a.c.cy[2];
        const double W = a.c.cw[0] + rr * a.c.cw[1] + cc * a.c.cw[2];
        if (!(W > 0)) return false;
        const double x = __builtin_floor(X / W), y = __builtin_floor(Y / W);
        xmin = x < xmin ? x : xmin; xmax = x > xmax ? x : xmax; ymin = y < ymin ? y : ymin; ymax = y > ymax ? y : ymax;
    }
    if (!(xmax - xmin < 1e6 && ymax - ymin < 1e6)) return false;
    // (the kernels' windows start on a multiple of 4 texels: up to 3 more columns.  Round 4 -- the whole-patch test used to ignore them, and at
    //  1.3x minification the host chose whole 64 x 8 patches of which most then failed the device's test and gathered: 0.35 instead of 0.44)
    const double xal = xmin >= 0 ? xmin - __builtin_fmod(xmin, 4.0) : xmin;
    *rows = (long long)(ymax - ymin) + 2; *texels = (long long)(xmax - xal) + 2;
    *lines = (double)*rows * (3.0 * (double)*texels / 128.0 + 1.0);
    return true;
}

// floor(n / d) == umulhi(n, magic) for every n < n_max, or 0 if no such 32-bit magic exists
inline unsigned div_magic(unsigned d, unsigned long long n_max) {
    if (d == 1) return 0;  // caller special-cases d == 1 (umulhi cannot express the identity)
    const unsigned long long m = (1ull << 32) / d + 1;
    // error term: n * (m*d - 2^32) < 2^32 must hold for all n < n_max
    const unsigned long long e = m * d - (1ull << 32);
    if (m >= (1ull << 32) || e * n_max >= (1ull << 32)) return 0;
    return (unsigned)m;
}

}  // namespace rwh
