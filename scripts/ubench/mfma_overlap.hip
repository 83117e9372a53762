// Micro-benchmark: do f32 MFMAs of one wave overlap with VALU work of the other waves of its SIMD?
// Each wave repeats { MF mfma_f32_32x32x2f32 ; VA dependent-free v_fma_f32 } and we time three mixes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ITER = 2000;

template <int MF, int VA>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
    f32x16 acc[2] = {};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    float ma = threadIdx.x * 0.01f, mb = 1.0f + threadIdx.x * 0.001f;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < MF; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc[i & 1], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < VA; ++i) x[i & 7] = __builtin_fmaf(x[i & 7], a, b);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K> float run(K kernel, int blocks, float* d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 8192 * sizeof(float));
    for (int blocks : {1024, 2048, 3072}) {   // 4, 8, 12 waves per SIMD requested (VGPR-limited)
        float v = run(k<0, 512>, blocks, d), m = run(k<16, 0>, blocks, d), both = run(k<16, 512>, blocks, d), half = run(k<16, 256>, blocks, d);
        double waves_per_simd = blocks * 4.0 / 1024.0;
        printf("blocks %d (%.0f waves/SIMD): VALU-only(512 fma) %.3f ms | MFMA-only(16) %.3f ms | 16 mfma + 512 fma %.3f ms | 16 mfma + 256 fma %.3f ms\n",
               blocks, waves_per_simd, v, m, both, half);
        printf("    cycles per iteration per SIMD at 2.4 GHz: VALU %.0f, MFMA %.0f, both %.0f, 16+256: %.0f\n",
               v * 1e-3 * 2.4e9 / ITER / waves_per_simd, m * 1e-3 * 2.4e9 / ITER / waves_per_simd,
               both * 1e-3 * 2.4e9 / ITER / waves_per_simd, half * 1e-3 * 2.4e9 / ITER / waves_per_simd);
    }
    return 0;
}
