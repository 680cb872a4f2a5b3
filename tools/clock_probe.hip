// Developer microbenchmark (NOT part of the product): which shader clock does MI355X hold under which instruction mix?
// Every CU runs 8 waves per SIMD of one loop body; block 0's lane 0 stamps s_memtime / s_memrealtime around its loop
// (after a 1 s warm-up of the same kernel) and the host prints clock, time and cycles per wave-instruction per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/clock_probe.hip -o tools/clock_probe && gpurun -- ./tools/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <atomic>
#include <string>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int ITERS = 4096;
typedef float f2 __attribute__((ext_vector_type(2)));

#define STAMP0 const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define STAMP1 if (threadIdx.x == 0 && blockIdx.x < 2048) { stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0; stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

__global__ __launch_bounds__(256) void k_fma32(unsigned long long* stamps, float* out, float seed, const float*, float*) {
    float a0=seed,a1=seed+1,a2=seed+2,a3=seed+3,a4=seed+4,a5=seed+5,a6=seed+6,a7=seed+7; float b=seed*0.5f, c=seed*0.25f;
    STAMP0
    for (int it = 0; it < ITERS; ++it) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
        R8(X)
#undef X
    }
    STAMP1
    if (seed == -7.f) out[threadIdx.x] = a0+a1+a2+a3+a4+a5+a6+a7;
}
__global__ __launch_bounds__(256) void k_fma64(unsigned long long* stamps, float* out, float seed, const float*, float*) {
    double a0=seed,a1=seed+1,a2=seed+2,a3=seed+3,a4=seed+4,a5=seed+5,a6=seed+6,a7=seed+7; double b=seed*0.5, c=seed*0.25;
    STAMP0
    for (int it = 0; it < ITERS; ++it) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
        R8(X)
#undef X
    }
    STAMP1
    if (seed == -7.f) out[threadIdx.x] = (float)(a0+a1+a2+a3+a4+a5+a6+a7);
}
__global__ __launch_bounds__(256) void k_pkfma(unsigned long long* stamps, float* out, float seed, const float*, float*) {
    f2 a0 = {seed, seed + 1}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 b = a0 * 0.5f, c = a0 * 0.25f;
    STAMP0
    for (int it = 0; it < ITERS; ++it) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
        R8(X)
#undef X
    }
    STAMP1
    f2 s = a0+a1+a2+a3+a4+a5+a6+a7;
    if (seed == -7.f) out[threadIdx.x] = s.x + s.y;
}
__global__ __launch_bounds__(256) void k_cvtub(unsigned long long* stamps, float* out, float seed, const float*, float*) {
    float a0=seed,a1=seed+1,a2=seed+2,a3=seed+3,a4=seed+4,a5=seed+5,a6=seed+6,a7=seed+7; unsigned b = (unsigned)(seed * 1234567.f);
    STAMP0
    for (int it = 0; it < ITERS; ++it) {
#define X(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a##i) : "v"(b));
        R8(X)
#undef X
    }
    STAMP1
    if (seed == -7.f) out[threadIdx.x] = a0+a1+a2+a3+a4+a5+a6+a7;
}
__global__ __launch_bounds__(256) void k_fmamix(unsigned long long* stamps, float* out, float seed, const float*, float*) {
    float a0=seed,a1=seed+1,a2=seed+2,a3=seed+3,a4=seed+4,a5=seed+5,a6=seed+6,a7=seed+7; float c=seed*0.25f; unsigned b = 0x00370012u + (unsigned)seed;
    STAMP0
    for (int it = 0; it < ITERS; ++it) {
#define X(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a##i) : "v"(b), "v"(c));
        R8(X)
#undef X
    }
    STAMP1
    if (seed == -7.f) out[threadIdx.x] = a0+a1+a2+a3+a4+a5+a6+a7;
}
__global__ __launch_bounds__(256) void k_iadd(unsigned long long* stamps, float* out, float seed, const float*, float*) {
    unsigned a0=(unsigned)seed,a1=a0+1,a2=a0+2,a3=a0+3,a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7; unsigned b=a0*3u, c=a0*5u;
    STAMP0
    for (int it = 0; it < ITERS; ++it) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
        R8(X)
#undef X
    }
    STAMP1
    if (seed == -7.f) out[threadIdx.x] = (float)(a0+a1+a2+a3+a4+a5+a6+a7);
}
// the warp kernel's mix, roughly: per 47 instructions 9 f64, 12 cvt_ubyte, 7 pk_fma, 19 plain 32-bit
__global__ __launch_bounds__(256) void k_mix(unsigned long long* stamps, float* out, float seed, const float*, float*) {
    double d0=seed,d1=seed+1,d2=seed+2; double db=seed*0.5, dc=seed*0.25;
    float a0=seed,a1=seed+1,a2=seed+2,a3=seed+3; unsigned ub = (unsigned)(seed * 1234567.f), u0 = ub, u1 = ub + 1;
    f2 p0 = {seed, seed + 1}, p1 = p0 + 1.f, pb = p0 * 0.5f, pc = p0 * 0.25f;
    STAMP0
    for (int it = 0; it < ITERS; ++it) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(db), "v"(dc));
        asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a0) : "v"(ub));
        asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(a1) : "v"(ub));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pb), "v"(pc));
        asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u0) : "v"(ub), "v"(u1));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d1) : "v"(db), "v"(dc));
        asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a2) : "v"(ub));
        asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a3) : "v"(ub));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(pb), "v"(pc));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(u1) : "v"(ub));
    }
    STAMP1
    if (seed == -7.f) out[threadIdx.x] = (float)(d0+d1+d2) + a0+a1+a2+a3 + p0.x + p1.y + (float)(u0 + u1);
}
// streaming copy, 16 B per lane (what HBM traffic alone does to the clock)
__global__ __launch_bounds__(256) void k_copy(unsigned long long* stamps, float* out, float seed, const float* src, float* dst) {
    STAMP0
    const size_t n4 = (size_t)1 << 26;   // 2^26 float4 = 1 GiB
    const float4* s4 = reinterpret_cast<const float4*>(src); float4* d4 = reinterpret_cast<float4*>(dst);
    for (int rep = 0; rep < 4; ++rep)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) d4[i] = s4[i];
    STAMP1
    if (seed == -7.f) out[threadIdx.x] = 0;
}

struct Case { const char* name; void (*k)(unsigned long long*, float*, float, const float*, float*); double inst_per_wave; };

static std::atomic<bool> g_stop{false};
static void sampler(std::string* log) {   // rocm-smi power / clock while the kernels run (best effort)
    while (!g_stop) {
        FILE* f = popen("rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk|mclk' | tr '\\n' ' '", "r");
        if (f) { char buf[512]; if (fgets(buf, sizeof buf, f)) { *log = buf; } pclose(f); }
        std::this_thread::sleep_for(std::chrono::milliseconds(200));
    }
}

int main() {
    unsigned long long* stamps; float* out; float *src, *dst;
    CK(hipMalloc(&stamps, 2048 * 16)); CK(hipMalloc(&out, 4096));
    CK(hipMalloc(&src, (size_t)1 << 30)); CK(hipMalloc(&dst, (size_t)1 << 30));
    CK(hipMemset(src, 1, (size_t)1 << 30));
    Case cases[] = {{"v_fma_f32", k_fma32, 8.0 * ITERS}, {"v_fma_f64", k_fma64, 8.0 * ITERS}, {"v_pk_fma_f32", k_pkfma, 8.0 * ITERS},
                    {"v_cvt_f32_ubyte", k_cvtub, 8.0 * ITERS}, {"v_fma_mix_f32", k_fmamix, 8.0 * ITERS}, {"v_mad_u32_u24", k_iadd, 8.0 * ITERS},
                    {"warp-like mix", k_mix, 10.0 * ITERS}, {"copy 16B/lane", k_copy, 0}};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 8;
    for (auto& c : cases) {
        std::string smi; g_stop = false;
        std::thread th(sampler, &smi);
        // ~1.5 s of back-to-back launches so that the clock settles, then the measured launches
        float ms = 0; int reps = 0;
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, stamps, out, 1.0f, src, dst);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, stamps, out, 1.0f, src, dst); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const int warm = (int)(1500.0 / ms) + 1;
        for (int i = 0; i < warm; ++i) hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, stamps, out, 1.0f, src, dst);
        CK(hipEventRecord(e0));
        reps = (int)(500.0 / ms) + 1;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, stamps, out, 1.0f, src, dst);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        g_stop = true; th.join();
        unsigned long long h[4096]; CK(hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost));
        double cyc = 0, tick = 0; for (int b = 0; b < 2048; ++b) { cyc += (double)h[2 * b]; tick += (double)h[2 * b + 1]; }
        const double mhz = 100.0 * cyc / tick;
        printf("%-18s %8.4f ms  clock %6.0f MHz", c.name, ms, mhz);
        if (c.inst_per_wave > 0) printf("  %5.2f cycles per wave-instruction per SIMD (8 waves/SIMD)", ms * 1e-3 * mhz * 1e6 / (8.0 * c.inst_per_wave));
        else printf("  %6.0f GB/s", 8.0 * (double)((size_t)1 << 30) / ms / 1e6);
        printf("   | %s\n", smi.c_str());
        fflush(stdout);
    }
    return 0;
}
