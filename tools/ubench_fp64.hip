// ubench_fp64.hip - issue-rate probe for the gfx950 vector instructions the pair loop uses.
//
// Each kernel runs a long unrolled stream of ONE instruction kind on independent
// registers and reports cycles per wave-instruction per SIMD, for 1, 2 and 4 waves per
// SIMD.  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_fp64.hip -o tools/ubench_fp64
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

constexpr int ITERS = 2000;
constexpr int UNROLL = 16;  // instructions per iteration, 8 independent chains x 2

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// Eight independent double chains a0..a7; OP(i) is an asm statement acting on a##i.
#define DEFINE_F64_KERNEL(NAME, ASMSTR)                                              \
    __global__ void NAME(double* out, double seed, unsigned long long* cyc) {        \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3,       \
               a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                   \
        double b = seed * 0.5 + 1.0, c = seed * 0.25 + 0.125;                        \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                        \
        for (int it = 0; it < ITERS; ++it) {                                         \
            _Pragma("unroll") for (int u = 0; u < UNROLL / 8; ++u) {                 \
                asm volatile(ASMSTR : "+v"(a0) : "v"(b), "v"(c) : "vcc", "s20");                    \
                asm volatile(ASMSTR : "+v"(a1) : "v"(b), "v"(c) : "vcc", "s20");                    \
                asm volatile(ASMSTR : "+v"(a2) : "v"(b), "v"(c) : "vcc", "s20");                    \
                asm volatile(ASMSTR : "+v"(a3) : "v"(b), "v"(c) : "vcc", "s20");                    \
                asm volatile(ASMSTR : "+v"(a4) : "v"(b), "v"(c) : "vcc", "s20");                    \
                asm volatile(ASMSTR : "+v"(a5) : "v"(b), "v"(c) : "vcc", "s20");                    \
                asm volatile(ASMSTR : "+v"(a6) : "v"(b), "v"(c) : "vcc", "s20");                    \
                asm volatile(ASMSTR : "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20");                    \
            }                                                                        \
        }                                                                            \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                             \
    }

DEFINE_F64_KERNEL(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
DEFINE_F64_KERNEL(k_mul_f64, "v_mul_f64 %0, %0, %1")
DEFINE_F64_KERNEL(k_add_f64, "v_add_f64 %0, %0, %1")
DEFINE_F64_KERNEL(k_rsq_f64, "v_rsq_f64 %0, %0")
DEFINE_F64_KERNEL(k_rcp_f64, "v_rcp_f64 %0, %0")
DEFINE_F64_KERNEL(k_sqrt_f64, "v_sqrt_f64 %0, %0")
DEFINE_F64_KERNEL(k_rndne_f64, "v_rndne_f64 %0, %0")
DEFINE_F64_KERNEL(k_min_f64, "v_min_f64 %0, %0, %1")
DEFINE_F64_KERNEL(k_cmp_f64, "v_cmp_gt_f64 vcc, %0, %1")
DEFINE_F64_KERNEL(k_mov_b64, "v_mov_b64 %0, %1")

// f32-side instructions on float chains
#define DEFINE_F32_KERNEL(NAME, ASMSTR)                                              \
    __global__ void NAME(double* out, double seed, unsigned long long* cyc) {        \
        float a0 = (float)seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, \
              a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                    \
        float b = (float)seed * 0.5f + 1.0f, c = (float)seed * 0.25f + 0.125f;       \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                        \
        for (int it = 0; it < ITERS; ++it) {                                         \
            _Pragma("unroll") for (int u = 0; u < UNROLL / 8; ++u) {                 \
                asm volatile(ASMSTR : "+v"(a0) : "v"(b), "v"(c) : "vcc", "s20");                    \
                asm volatile(ASMSTR : "+v"(a1) : "v"(b), "v"(c) : "vcc", "s20");                    \
                asm volatile(ASMSTR : "+v"(a2) : "v"(b), "v"(c) : "vcc", "s20");                    \
                asm volatile(ASMSTR : "+v"(a3) : "v"(b), "v"(c) : "vcc", "s20");                    \
                asm volatile(ASMSTR : "+v"(a4) : "v"(b), "v"(c) : "vcc", "s20");                    \
                asm volatile(ASMSTR : "+v"(a5) : "v"(b), "v"(c) : "vcc", "s20");                    \
                asm volatile(ASMSTR : "+v"(a6) : "v"(b), "v"(c) : "vcc", "s20");                    \
                asm volatile(ASMSTR : "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20");                    \
            }                                                                        \
        }                                                                            \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                             \
    }

DEFINE_F32_KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
DEFINE_F32_KERNEL(k_rsq_f32, "v_rsq_f32 %0, %0")
DEFINE_F32_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEFINE_F32_KERNEL(k_mov_b32, "v_mov_b32 %0, %1")
DEFINE_F32_KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3")
DEFINE_F32_KERNEL(k_dpp_mov, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")

// conversions: f64 -> f32 -> f64 round trips counted as two instructions
__global__ void k_cvt_pair(double* out, double seed, unsigned long long* cyc) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4,
           a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float f0, f1, f2, f3, f4, f5, f6, f7;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#define CV(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f##i) : "v"(a##i));
        REP8(CV)
#undef CV
#define CV(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a##i) : "v"(f##i));
        REP8(CV)
#undef CV
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// LDS broadcast read of one 32-byte source record (2 x ds_read_b128), 8 per iteration x2
__global__ void k_lds_bcast(double* out, double seed, unsigned long long* cyc) {
    __shared__ double4 tile[256];
    tile[threadIdx.x] = make_double4(seed, seed + 1, seed + 2, threadIdx.x);
    __syncthreads();
    double acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            double4 p = tile[(it * 8 + u) & 255];
            asm volatile("" : "+v"(p.x), "+v"(p.y), "+v"(p.z), "+v"(p.w));
            acc += p.x;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef void (*kern_t)(double*, double, unsigned long long*);

// Occupancy is pinned with dynamic LDS: 160 KiB / wps per 256-thread block leaves room for
// exactly `wps` blocks per CU, i.e. `wps` waves per SIMD.  The grid is 8 rounds of that.
static void run(const char* name, kern_t k, int insts_per_iter, double* out,
                unsigned long long* cyc) {
    printf("%-14s", name);
    for (int wps : {1, 2, 4, 8}) {
        const int threads = 256;
        const int rounds = 8;
        const int blocks = 256 * wps * rounds;
        size_t lds = (160 * 1024) / wps - 9 * 1024;
        CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, out, 1.0, cyc);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, out, 1.0, cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        double per_simd = (double)ITERS * insts_per_iter * wps * rounds;  // wave-insts per SIMD
        double ns = ms * 1e6;
        printf("  wps=%d: %6.2f ns/1000inst (%6.2f cyc@2.4)", wps, ns / per_simd * 1000.0,
               ns * 2.4 / per_simd);
    }
    printf("\n");
}

// Shader clock under a dense fp64 load: s_memtime (shader clock) over s_memrealtime (100 MHz).
__global__ void k_clock(double* out, double seed, unsigned long long* cyc) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    double b = seed * 0.5 + 1.0, c = seed * 0.25 + 0.125;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < 40 * ITERS; ++it) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}

// Accuracy of the reciprocal-square-root seeds over q in [1e-8, 1e2).
__global__ void k_seed_error(double* out) {
    double worst64 = 0, worst32 = 0;
    unsigned long long s = 88172645463325252ull + threadIdx.x * 2654435761ull + blockIdx.x * 40503ull;
    for (int it = 0; it < 20000; ++it) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        double u = (double)(s >> 11) * (1.0 / 9007199254740992.0);
        double q = exp(log(1e-8) + u * (log(1e2) - log(1e-8)));
        double exact = 1.0 / sqrt(q);
        double y64 = __builtin_amdgcn_rsq(q);
        double y32 = (double)__builtin_amdgcn_rsqf((float)q);
        worst64 = fmax(worst64, fabs(y64 - exact) / exact);
        worst32 = fmax(worst32, fabs(y32 - exact) / exact);
    }
    out[2 * (blockIdx.x * blockDim.x + threadIdx.x)] = worst64;
    out[2 * (blockIdx.x * blockDim.x + threadIdx.x) + 1] = worst32;
}

int main() {
    double* out;
    unsigned long long* cyc;
    CHECK(hipMalloc(&out, 4 * 1024 * 1024 * sizeof(double)));
    CHECK(hipMalloc(&cyc, 65536 * sizeof(unsigned long long)));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, CUs %d, clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount,
           prop.clockRate);
    {
        hipLaunchKernelGGL(k_clock, dim3(1024), dim3(256), 0, 0, out, 1.0, cyc);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_clock, dim3(1024), dim3(256), 0, 0, out, 1.0, cyc);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(2048);
        CHECK(hipMemcpy(h.data(), cyc, 2048 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double ratio = 0; for (int b = 0; b < 1024; ++b) ratio += (double)h[2*b] / (double)h[2*b+1];
        ratio /= 1024;
        // 1024 blocks x 4 waves = 16 waves per CU-SIMD set -> 4 waves/SIMD, each 160000*... insts
        double insts_per_simd = 40.0 * ITERS * 4 * 4;  // 4 waves per SIMD
        printf("dense v_fma_f64, 4 waves/SIMD: %.3f ms, memtime/memrealtime = %.3f (x100 MHz => %.0f MHz), "
               "%.2f shader cycles per wave-inst per SIMD\n", ms, ratio, ratio * 100.0,
               ms * 1e-3 * ratio * 1e8 / insts_per_simd);
    }
    {
        hipLaunchKernelGGL(k_seed_error, dim3(64), dim3(256), 0, 0, out);
        CHECK(hipDeviceSynchronize());
        std::vector<double> h(2 * 64 * 256);
        CHECK(hipMemcpy(h.data(), out, h.size() * sizeof(double), hipMemcpyDeviceToHost));
        double w64 = 0, w32 = 0;
        for (size_t i = 0; i < h.size(); i += 2) { w64 = fmax(w64, h[i]); w32 = fmax(w32, h[i+1]); }
        printf("max rel error of seeds: v_rsq_f64 %.3e (2^%.1f), cvt+v_rsq_f32 %.3e (2^%.1f)\n", w64,
               log2(w64), w32, log2(w32));
    }
    run("v_fma_f64", k_fma_f64, UNROLL, out, cyc);
    run("v_mul_f64", k_mul_f64, UNROLL, out, cyc);
    run("v_add_f64", k_add_f64, UNROLL, out, cyc);
    run("v_rsq_f64", k_rsq_f64, UNROLL, out, cyc);
    run("v_rcp_f64", k_rcp_f64, UNROLL, out, cyc);
    run("v_rndne_f64", k_rndne_f64, UNROLL, out, cyc);
    run("v_cmp_gt_f64", k_cmp_f64, UNROLL, out, cyc);
    run("v_mov_b64", k_mov_b64, UNROLL, out, cyc);
    run("v_fma_f32", k_fma_f32, UNROLL, out, cyc);
    run("v_rsq_f32", k_rsq_f32, UNROLL, out, cyc);
    run("v_cndmask_b32", k_cndmask, UNROLL, out, cyc);
    run("v_mov_b32", k_mov_b32, UNROLL, out, cyc);
    run("v_readlane", k_readlane, UNROLL, out, cyc);
    run("cvt f64<->f32", k_cvt_pair, 16, out, cyc);
    run("lds 32B bcast", k_lds_bcast, 8, out, cyc);
    return 0;
}
