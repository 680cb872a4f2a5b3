// Developer microbenchmark (NOT part of the product): issue cost of the VALU instructions the warp
// kernel is made of, on gfx950.  Every CU runs 8 waves per SIMD; each wave executes N copies of one
// instruction on independent registers in a loop; reported = SIMD cycles per wave-instruction at an
// assumed 2.4 GHz (read ratios, not absolutes: the clock under load is not 2.4 GHz).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rates.hip -o tools/valu_rates && gpurun -- ./tools/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 512;

#define KERNEL(NAME, DECL, BODY, SINK)                                              \
    __global__ __launch_bounds__(256) void NAME(float* out, float seed) {          \
        DECL;                                                                       \
        for (int it = 0; it < ITERS; ++it) { BODY; }                                \
        if (seed == -7.f) out[threadIdx.x] = (float)(SINK);                        \
    }

// 8 independent chains per wave so that dependent-issue latency does not bound the loop
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define D_F32 float a0=seed,a1=seed+1,a2=seed+2,a3=seed+3,a4=seed+4,a5=seed+5,a6=seed+6,a7=seed+7; float b=seed*0.5f, c=seed*0.25f
#define D_F64 double a0=seed,a1=seed+1,a2=seed+2,a3=seed+3,a4=seed+4,a5=seed+5,a6=seed+6,a7=seed+7; double b=seed*0.5, c=seed*0.25
#define D_U32 unsigned a0=(unsigned)seed,a1=a0+1,a2=a0+2,a3=a0+3,a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7; unsigned b=a0*3u, c=a0*5u
#define SUM (a0+a1+a2+a3+a4+a5+a6+a7)

#define X_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL(k_fma_f32, D_F32, R8(X_FMA32), SUM)
#define X_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL(k_fma_f64, D_F64, R8(X_FMA64), SUM)
#define X_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a##i) : "v"(b));
KERNEL(k_add_f64, D_F64, R8(X_ADD64), SUM)
#define X_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a##i) : "v"(b));
KERNEL(k_mul_f64, D_F64, R8(X_MUL64), SUM)
#define X_RCP64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a##i));
KERNEL(k_rcp_f64, D_F64, R8(X_RCP64), SUM)
#define X_FRACT64(i) asm volatile("v_fract_f64 %0, %0" : "+v"(a##i));
KERNEL(k_fract_f64, D_F64, R8(X_FRACT64), SUM)
#define X_RCP32(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##i));
KERNEL(k_rcp_f32, D_F32, R8(X_RCP32), SUM)
#define X_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b) : );
KERNEL(k_cndmask, D_U32, R8(X_CNDMASK), SUM)
#define X_CVTUB(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a##i) : "v"(b));
KERNEL(k_cvt_f32_ubyte, D_F32; unsigned b2 = (unsigned)seed; (void)b2, R8(X_CVTUB), SUM)
#define X_CVTU32(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a##i));
KERNEL(k_cvt_f32_u32, D_F32, R8(X_CVTU32), SUM)
#define X_CVTU32F(i) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a##i));
KERNEL(k_cvt_u32_f32, D_F32, R8(X_CVTU32F), SUM)
#define X_PKU8(i) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(a##i) : "v"(b));
KERNEL(k_cvt_pk_u8_f32, D_F32, R8(X_PKU8), SUM)
#define X_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL(k_mad_u32_u24, D_U32, R8(X_MAD24), SUM)
#define X_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
KERNEL(k_mul_lo_u32, D_U32, R8(X_MULLO), SUM)
#define X_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(a##i) : "v"(b));
KERNEL(k_lshl_or, D_U32, R8(X_LSHLOR), SUM)
#define X_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL(k_perm_b32, D_U32, R8(X_PERM), SUM)

// two-register-class instructions: destination type differs from the source type
__global__ __launch_bounds__(256) void k_cvt_f32_f64(float* out, float seed) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3; float r0, r1, r2, r3; float s = 0;
    for (int it = 0; it < ITERS; ++it) {
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r0) : "v"(a0)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r1) : "v"(a1));
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r2) : "v"(a2)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r3) : "v"(a3));
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r0) : "v"(a1)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r1) : "v"(a2));
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r2) : "v"(a3)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r3) : "v"(a0));
    }
    s = r0 + r1 + r2 + r3;
    if (seed == -7.f) out[threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_cvt_i32_f64(float* out, float seed) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3; int r0, r1, r2, r3;
    for (int it = 0; it < ITERS; ++it) {
        asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r0) : "v"(a0)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r1) : "v"(a1));
        asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r2) : "v"(a2)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r3) : "v"(a3));
        asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r0) : "v"(a1)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r1) : "v"(a2));
        asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r2) : "v"(a3)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r3) : "v"(a0));
    }
    if (seed == -7.f) out[threadIdx.x] = (float)(r0 + r1 + r2 + r3);
}
__global__ __launch_bounds__(256) void k_cvt_f64_i32(float* out, float seed) {
    int a0 = (int)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3; double r0, r1, r2, r3;
    for (int it = 0; it < ITERS; ++it) {
        asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r0) : "v"(a0)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r1) : "v"(a1));
        asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r2) : "v"(a2)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r3) : "v"(a3));
        asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r0) : "v"(a1)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r1) : "v"(a2));
        asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r2) : "v"(a3)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r3) : "v"(a0));
    }
    if (seed == -7.f) out[threadIdx.x] = (float)(r0 + r1 + r2 + r3);
}
__global__ __launch_bounds__(256) void k_cmp_f64(float* out, float seed) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
    for (int it = 0; it < ITERS; ++it) {
        asm volatile("v_cmp_le_f64 vcc, %0, %1\n v_cmp_le_f64 vcc, %1, %2\n v_cmp_le_f64 vcc, %2, %3\n v_cmp_le_f64 vcc, %3, %0\n"
                     "v_cmp_le_f64 vcc, %0, %2\n v_cmp_le_f64 vcc, %1, %3\n v_cmp_le_f64 vcc, %2, %0\n v_cmp_le_f64 vcc, %3, %1"
                     :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");
    }
    if (seed == -7.f) out[threadIdx.x] = (float)a0;
}
__global__ __launch_bounds__(256) void k_cmp_u64(float* out, float seed) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
    for (int it = 0; it < ITERS; ++it) {
        asm volatile("v_cmp_le_u64 vcc, %0, %1\n v_cmp_le_u64 vcc, %1, %2\n v_cmp_le_u64 vcc, %2, %3\n v_cmp_le_u64 vcc, %3, %0\n"
                     "v_cmp_le_u64 vcc, %0, %2\n v_cmp_le_u64 vcc, %1, %3\n v_cmp_le_u64 vcc, %2, %0\n v_cmp_le_u64 vcc, %3, %1"
                     :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");
    }
    if (seed == -7.f) out[threadIdx.x] = (float)a0;
}
__global__ __launch_bounds__(256) void k_cmp_u32(float* out, float seed) {
    unsigned a0 = (unsigned)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    for (int it = 0; it < ITERS; ++it) {
        asm volatile("v_cmp_le_u32 vcc, %0, %1\n v_cmp_le_u32 vcc, %1, %2\n v_cmp_le_u32 vcc, %2, %3\n v_cmp_le_u32 vcc, %3, %0\n"
                     "v_cmp_le_u32 vcc, %0, %2\n v_cmp_le_u32 vcc, %1, %3\n v_cmp_le_u32 vcc, %2, %0\n v_cmp_le_u32 vcc, %3, %1"
                     :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");
    }
    if (seed == -7.f) out[threadIdx.x] = (float)a0;
}
__global__ __launch_bounds__(256) void k_pk_fma_f32(float* out, float seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {seed, seed + 1}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 b = a0 * 0.5f, c = a0 * 0.25f;
    for (int it = 0; it < ITERS; ++it) {
#define X_PK(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
        R8(X_PK)
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (seed == -7.f) out[threadIdx.x] = s.x + s.y;
}

struct Case { const char* name; void (*k)(float*, float); int per_iter; };

int main() {
    float* out; CK(hipMalloc(&out, 4096));
    Case cases[] = {
        {"v_fma_f32", k_fma_f32, 8}, {"v_pk_fma_f32", k_pk_fma_f32, 8}, {"v_fma_f64", k_fma_f64, 8}, {"v_add_f64", k_add_f64, 8},
        {"v_mul_f64", k_mul_f64, 8}, {"v_rcp_f64", k_rcp_f64, 8}, {"v_fract_f64", k_fract_f64, 8}, {"v_rcp_f32", k_rcp_f32, 8},
        {"v_cvt_f32_f64", k_cvt_f32_f64, 8}, {"v_cvt_i32_f64", k_cvt_i32_f64, 8}, {"v_cvt_f64_i32", k_cvt_f64_i32, 8},
        {"v_cmp_le_f64", k_cmp_f64, 8}, {"v_cmp_le_u64", k_cmp_u64, 8}, {"v_cmp_le_u32", k_cmp_u32, 8},
        {"v_cndmask_b32", k_cndmask, 8}, {"v_cvt_f32_ubyte1", k_cvt_f32_ubyte, 8}, {"v_cvt_f32_u32", k_cvt_f32_u32, 8},
        {"v_cvt_u32_f32", k_cvt_u32_f32, 8}, {"v_cvt_pk_u8_f32", k_cvt_pk_u8_f32, 8}, {"v_mad_u32_u24", k_mad_u32_u24, 8},
        {"v_mul_lo_u32", k_mul_lo_u32, 8}, {"v_lshl_or_b32", k_lshl_or, 8}, {"v_perm_b32", k_perm_b32, 8},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
    for (auto& c : cases) {
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
        CK(hipEventRecord(e0));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        // per SIMD: 8 waves x ITERS x per_iter instructions
        const double instr_per_simd = 8.0 * ITERS * c.per_iter;
        printf("%-18s %8.4f ms   %6.2f cycles/wave-instr @2.4GHz\n", c.name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    }
    return 0;
}
