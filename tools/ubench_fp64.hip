// ubench_fp64.hip - issue-rate probe for the gfx950 vector instructions the pair loop uses.
//
// Each kernel runs a long unrolled stream of ONE instruction kind on independent
// registers and reports cycles per wave-instruction per SIMD, for 1, 2 and 4 waves per
// SIMD.  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_fp64.hip -o tools/ubench_fp64
#include <hip/hip_runtime.h>

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

static void run(const char* name, kern_t k, int insts_per_iter, double* out,
                unsigned long long* cyc) {
    // one block per CU-ish (1024 blocks), waves per SIMD = threads/256
    printf("%-14s", name);
    for (int wps : {1, 2, 4}) {
        int threads = 256 * wps;
        int blocks = 256;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 1.0, cyc);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 1.0, cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks);
        CHECK(hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double mean = 0;
        for (auto v : h) mean += (double)v;
        mean /= blocks;
        // s_memtime ticks at a fixed 100 MHz on gfx9; report wall-based cycles at the
        // nominal 2.4 GHz as well as the tick count
        double total_wave_insts = (double)ITERS * insts_per_iter * wps;  // per SIMD
        double ns = ms * 1e6;
        printf("  wps=%d: %6.2f cyc/inst/SIMD@2.4GHz (%.3f ms, ticks %.0f)", wps,
               ns * 2.4 / total_wave_insts, ms, mean);
    }
    printf("\n");
}

int main() {
    double* out;
    unsigned long long* cyc;
    CHECK(hipMalloc(&out, 1024 * 1024 * sizeof(double)));
    CHECK(hipMalloc(&cyc, 4096 * sizeof(unsigned long long)));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, CUs %d, clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount,
           prop.clockRate);
    run("v_fma_f64", k_fma_f64, UNROLL, out, cyc);
    run("v_mul_f64", k_mul_f64, UNROLL, out, cyc);
    run("v_add_f64", k_add_f64, UNROLL, out, cyc);
    run("v_rsq_f64", k_rsq_f64, UNROLL, out, cyc);
    run("v_rcp_f64", k_rcp_f64, UNROLL, out, cyc);
    run("v_sqrt_f64", k_sqrt_f64, UNROLL, out, cyc);
    run("v_rndne_f64", k_rndne_f64, UNROLL, out, cyc);
    run("v_min_f64", k_min_f64, UNROLL, out, cyc);
    run("v_cmp_gt_f64", k_cmp_f64, UNROLL, out, cyc);
    run("v_mov_b64", k_mov_b64, UNROLL, out, cyc);
    run("v_fma_f32", k_fma_f32, UNROLL, out, cyc);
    run("v_rsq_f32", k_rsq_f32, UNROLL, out, cyc);
    run("v_cndmask_b32", k_cndmask, UNROLL, out, cyc);
    run("v_mov_b32", k_mov_b32, UNROLL, out, cyc);
    run("v_readlane", k_readlane, UNROLL, out, cyc);
    run("v_mov_dpp", k_dpp_mov, UNROLL, out, cyc);
    run("cvt f64<->f32", k_cvt_pair, 16, out, cyc);
    run("lds 32B bcast", k_lds_bcast, 8, out, cyc);
    return 0;
}
