// Micro-benchmark: cost of the closest-hit update "if (|hx| <= 1 && |hy| <= 1 && t < best && t > min_hit)
// { best = t; id = k; }" in two encodings: compares into SGPR masks + v_cndmask (what the compiler emits)
// against v_cmpx narrowing of EXEC + plain moves.  Both are fed by one FMA per test value so that the
// values change every iteration; the FMA-only kernel gives the baseline to subtract.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 2048;
constexpr int UNROLL = 8;

template <int MODE>
__global__ __launch_bounds__(256) void k_select(float* out, float a, float b, float min_hit) {
    float hx = threadIdx.x * 0.001f - 0.1f, hy = threadIdx.x * 0.0007f - 0.05f, t = 3.0f + threadIdx.x * 0.01f;
    float best = 3.0e38f;
    unsigned id = 0;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            hx = __builtin_fmaf(hx, a, b);
            hy = __builtin_fmaf(hy, a, -b);
            t = __builtin_fmaf(t, a, b);
            const unsigned k = (unsigned)(it * UNROLL + u);
            if (MODE == 1) {
                if (__builtin_fabsf(hx) <= 1.0f && __builtin_fabsf(hy) <= 1.0f && t < best && t > min_hit) { best = t; id = k; }
            } else if (MODE == 2) {
                unsigned long long save;
                float m;
                asm volatile(
                    "s_mov_b64 %[save], exec\n\t"
                    "v_cmpx_lt_f32_e32 %[t], %[best]\n\t"
                    "v_max_f32_e64 %[m], |%[hx]|, |%[hy]|\n\t"
                    "v_cmpx_gt_f32_e32 %[t], %[minhit]\n\t"
                    "v_cmpx_ge_f32_e32 1.0, %[m]\n\t"
                    "v_mov_b32_e32 %[best], %[t]\n\t"
                    "v_mov_b32_e32 %[id], %[k]\n\t"
                    "s_mov_b64 exec, %[save]\n\t"
                    : [save] "=&s"(save), [m] "=&v"(m), [best] "+v"(best), [id] "+v"(id)
                    : [t] "v"(t), [hx] "v"(hx), [hy] "v"(hy), [minhit] "v"(min_hit), [k] "s"(k)
                    : "vcc");
            } else if (MODE == 3) {
                unsigned long long save;
                asm volatile(
                    "s_mov_b64 %[save], exec\n\t"
                    "v_cmpx_lt_f32_e32 %[t], %[best]\n\t"
                    "v_cmpx_ge_f32_e64 exec, 1.0, |%[hx]|\n\t"
                    "v_cmpx_ge_f32_e64 exec, 1.0, |%[hy]|\n\t"
                    "v_cmpx_lt_f32_e32 %[minhit], %[t]\n\t"
                    "v_mov_b32_e32 %[best], %[t]\n\t"
                    "v_mov_b32_e32 %[id], %[k]\n\t"
                    "s_mov_b64 exec, %[save]\n\t"
                    : [save] "=&s"(save), [best] "+v"(best), [id] "+v"(id)
                    : [t] "v"(t), [hx] "v"(hx), [hy] "v"(hy), [minhit] "s"(min_hit), [k] "s"(k)
                    : "vcc");
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = hx + hy + t + best + (float)id;
}

template <typename K> float run(const char* name, K kernel, int blocks, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 0.9999f, 0.0001f, 0.001f);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 0.9999f, 0.0001f, 0.001f); hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double seqs = (double)ITER * UNROLL * blocks * 4.0 / 1024.0;   // sequences per SIMD
    printf("%-22s blocks %5d: %.3f ms, %.2f ns per wave-sequence per SIMD (%.1f cycles at 2.4 GHz)\n", name, blocks, ms,
           ms * 1e6 / seqs, ms * 1e-3 / seqs * 2.4e9);
    return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 8192 * sizeof(float));
    float* h = new float[256 * 4096];
    for (int blocks : {1024, 2048, 4096}) {
        run("3 fma", k_select<0>, blocks, d);
        run("3 fma + cndmask form", k_select<1>, blocks, d);
        hipMemcpy(h, d, 256 * 4096 * sizeof(float), hipMemcpyDeviceToHost);
        double s1 = 0; for (int i = 0; i < 256 * blocks; ++i) s1 += h[i];
        run("3 fma + cmpx form", k_select<2>, blocks, d);
        hipMemcpy(h, d, 256 * 4096 * sizeof(float), hipMemcpyDeviceToHost);
        double s2 = 0; for (int i = 0; i < 256 * blocks; ++i) s2 += h[i];
        run("3 fma + 4 cmpx form", k_select<3>, blocks, d);
        hipMemcpy(h, d, 256 * 4096 * sizeof(float), hipMemcpyDeviceToHost);
        double s3 = 0; for (int i = 0; i < 256 * blocks; ++i) s3 += h[i];
        printf("  checksums %s (%.9g vs %.9g vs %.9g)\n", s1 == s2 && s1 == s3 ? "equal" : "DIFFER", s1, s2, s3);
    }
    return 0;
}
