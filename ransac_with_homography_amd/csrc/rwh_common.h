// Shared device/host helpers for librwh_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rwh.h"

#define RWH_WAVE 64

// hipcc contracts a*b+c into FMA by default; the parity recipes in this library
// state every rounding explicitly, so contraction is switched off file-wide and
// fused operations are written as fma()/fmaf() where they are wanted.
#pragma clang fp contract(off)

namespace rwh {

// Packed (alignment 1) views for byte-addressed texel traffic.  gfx950 runs in
// unaligned-access mode: these lower to single global_load_dwordx2 /
// global_store_dwordx3 instructions at any byte address.
struct __attribute__((packed)) pk2 { uint32_t a, b; };
struct __attribute__((packed)) pk3 { uint32_t a, b, c; };
struct __attribute__((packed)) pk4 { uint32_t a, b, c, d; };

__device__ __forceinline__ pk2 ld8(const unsigned char* p) { pk2 v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ uint32_t ld4(const unsigned char* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

extern int g_force_warp_shape, g_force_score_hpw, g_score_exact_only, g_force_warp_frames;   // rwh_api.hip (rwh_lab_tune)

inline int check_launch() {
    return hipGetLastError() == hipSuccess ? RWH_OK : RWH_E_LAUNCH;
}

}  // namespace rwh
