// Does the FP64 matrix pipe run beside the FP64 vector pipe on gfx950?  Times a loop of NV
// independent v_fma_f64 per iteration with 0, 1 or 2 v_mfma_f64_4x4x4 (and the MFMAs alone, and the
// 16x16x4 form) at 4 and 8 wavefronts per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 -o mfma_f64_overlap mfma_f64_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

template <int NV, int NM, int FORM>
__global__ void __launch_bounds__(256) k_loop(int iters, double* out, double seed) {
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3 + i;
    double m0 = 0.0, m1 = 0.0;
    v4d w0 = {0, 0, 0, 0}, w1 = {0, 0, 0, 0};
    const double x = seed * 0.5, y = 1.0 + seed * 1e-9;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int v = 0; v < NV; ++v) a[v & 7] = __builtin_fma(a[v & 7], y, x);
        if constexpr (FORM == 0) {
            if constexpr (NM >= 1) m0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], x, m0, 0, 0, 0);
            if constexpr (NM >= 2) m1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[1], y, m1, 0, 0, 0);
        } else {
            if constexpr (NM >= 1) w0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], x, w0, 0, 0, 0);
            if constexpr (NM >= 2) w1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], y, w1, 0, 0, 0);
        }
    }
    double s = m0 + m1 + w0[0] + w0[1] + w0[2] + w0[3] + w1[0] + w1[1] + w1[2] + w1[3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int NM, int FORM>
void run(const char* name, int waves_per_simd) {
    const int iters = 200000;
    const int blocks = 256 * waves_per_simd;   // 256 CUs, 4 waves (256 threads) per block: one wave per SIMD per block
    double* out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_loop<NV, NM, FORM><<<blocks, 256>>>(1000, out, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_loop<NV, NM, FORM><<<blocks, 256>>>(iters, out, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: waves_per_simd waves, each `iters` iterations
    const double ns_per_iter_per_simd = ms * 1e6 / (double(iters) * waves_per_simd);
    std::printf("%-34s waves/SIMD %d: %8.3f ms, %7.2f ns per iteration per SIMD (%.1f cycles at 2.4 GHz)\n", name,
                waves_per_simd, ms, ns_per_iter_per_simd, ns_per_iter_per_simd * 2.4);
    hipFree(out);
}

int main() {
    for (int w : {4, 8}) {
        run<16, 0, 0>("16 fma", w);
        run<16, 1, 0>("16 fma + 1 mfma 4x4x4", w);
        run<16, 2, 0>("16 fma + 2 mfma 4x4x4", w);
        run<0, 2, 0>("2 mfma 4x4x4", w);
        run<16, 1, 1>("16 fma + 1 mfma 16x16x4", w);
        run<0, 1, 1>("1 mfma 16x16x4", w);
        run<24, 0, 0>("24 fma", w);
    }
    return 0;
}
