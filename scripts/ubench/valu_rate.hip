// Micro-benchmark: issue rate of scalar and packed FP32 VALU instructions on one chip.
// Each thread runs ITER iterations over 8 independent accumulator chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float F2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

__global__ __launch_bounds__(256) void k_fma(float* out, float a, float b) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_pkfma(float* out, float a, float b) {
    F2 x[8]; F2 a2 = {a, a * 1.0001f}, b2 = {b, b * 0.999f};
    for (int i = 0; i < 8; ++i) { x[i].x = threadIdx.x * 0.001f + i; x[i].y = x[i].x + 0.5f; }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], a2, b2);
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_cmp(float* out, float a, float b) {
    float x[8]; float best = 1e30f; int id = 0;
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { x[i] = __builtin_fmaf(x[i], a, b); if (x[i] < best) { best = x[i]; id = it + i; } }
    }
    out[blockIdx.x * 256 + threadIdx.x] = best + id;
}
__global__ __launch_bounds__(256) void k_rcp(float* out, float a, float b) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i + 1.0f;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_rcpf(x[i]) + a;
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K> void run(const char* name, K kernel, int blocks, double insts_per_thread, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f); hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_insts = insts_per_thread * blocks * 4.0;   // 4 waves per block
    double per_simd_per_s = wave_insts / 1024.0 / (ms * 1e-3);
    printf("%-8s blocks %5d: %.3f ms, %.2f G wave-inst/s/SIMD -> %.2f cycles per wave64 instruction at 2.4 GHz\n", name, blocks, ms,
           per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8192 * sizeof(float));
    for (int blocks : {256, 1024, 2048, 4096}) {   // 1, 4, 8, 16 waves per SIMD requested
        run("fma", k_fma, blocks, 8.0 * ITER, d);
        run("pk_fma", k_pkfma, blocks, 8.0 * ITER, d);
        run("fma+cmp", k_cmp, blocks, 8.0 * ITER * 4, d);
        run("rcp+add", k_rcp, blocks, 8.0 * ITER * 2, d);
    }
    return 0;
}
