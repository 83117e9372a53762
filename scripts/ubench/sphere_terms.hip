// Micro-benchmark: the per-sphere part of the sphere test (b, c, discriminant of main.cpp:66-72 in its
// fma form, four spheres at a time, with the "any discriminant positive" test) on register-resident
// data - no LDS, no memory - to find the VALU floor of that loop on this chip, in SIMD cycles per
// sphere and wave, at 1, 2, 4 and 8 waves per SIMD.  Built twice by the caller: as is (the compiler
// packs pairs of spheres into v_pk_* instructions) and with -fno-slp-vectorize (scalar v_fma/v_sub/v_mul).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off [-fno-slp-vectorize] -o build/sphere_terms scripts/ubench/sphere_terms.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITER = 4096;   // groups of four spheres per wave
struct Stamp { unsigned long long cycles, real; };

__device__ __forceinline__ float dot_fma(float ax, float ay, float az, float bx, float by, float bz) {
    return __builtin_fmaf(ax, bx, __builtin_fmaf(ay, by, az * bz));
}

__global__ __launch_bounds__(256) void k_terms(float* out, float4 s0, float4 s1, float4 s2, float4 s3, float step, Stamp* stamps) {
    const float ox = threadIdx.x * 0.01f, oy = 1.0f + blockIdx.x * 0.001f, oz = -3.0f;
    const float dx = 0.6f, dy = 0.0f + threadIdx.x * 1e-4f, dz = 0.8f;
    const float d2x = 2 * dx, d2y = 2 * dy, d2z = 2 * dz;
    const float minus_four_a = -4.0f * dot_fma(dx, dy, dz, dx, dy, dz);
    // the records change from group to group (as they would coming from LDS), all in vector registers
    float4 r0 = s0, r1 = s1, r2 = s2, r3 = s3;
    r0.x += threadIdx.x * 0.f; r1.x += threadIdx.x * 0.f; r2.x += threadIdx.x * 0.f; r3.x += threadIdx.x * 0.f;
    r0.y += threadIdx.x * 0.f; r1.y += threadIdx.x * 0.f; r2.y += threadIdx.x * 0.f; r3.y += threadIdx.x * 0.f;
    r0.z += threadIdx.x * 0.f; r1.z += threadIdx.x * 0.f; r2.z += threadIdx.x * 0.f; r3.z += threadIdx.x * 0.f;
    r0.w += threadIdx.x * 0.f; r1.w += threadIdx.x * 0.f; r2.w += threadIdx.x * 0.f; r3.w += threadIdx.x * 0.f;
    float acc = 0.0f;
    uint32_t hits = 0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITER; ++it) {
        float e[4];
        const float4 rec[4] = {r0, r1, r2, r3};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float cx = ox - rec[j].x, cy = oy - rec[j].y, cz = oz - rec[j].z;
            const float b = dot_fma(d2x, d2y, d2z, cx, cy, cz);
            const float c = dot_fma(cx, cy, cz, cx, cy, cz) - rec[j].w;
            e[j] = __builtin_fmaf(b, b, minus_four_a * c);
        }
        if ((int32_t)(__float_as_uint(e[0]) & __float_as_uint(e[1]) & __float_as_uint(e[2]) & __float_as_uint(e[3])) >= 0) {
            acc += e[0] + e[1] + e[2] + e[3];   // never taken with the host's data (all discriminants negative)
            ++hits;
        }
        // opaque to the compiler: the next group's records "arrive" (keeps the loop from being hoisted)
        asm volatile("" : "+v"(r0.x), "+v"(r1.x), "+v"(r2.x), "+v"(r3.x));
        asm volatile("" : "+v"(r0.y), "+v"(r1.y), "+v"(r2.y), "+v"(r3.y));
        asm volatile("" : "+v"(r0.z), "+v"(r1.z), "+v"(r2.z), "+v"(r3.z));
        asm volatile("" : "+v"(r0.w), "+v"(r1.w), "+v"(r2.w), "+v"(r3.w));
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) { Stamp st; st.cycles = c1 - c0; st.real = t1 - t0; stamps[blockIdx.x * 4 + threadIdx.x / 64] = st; }
    out[blockIdx.x * 256 + threadIdx.x] = acc + hits + step;
}

int main() {
    float* d; Stamp* d_stamps;
    (void)hipMalloc(&d, 256 * 4096 * sizeof(float));
    (void)hipMalloc(&d_stamps, sizeof(Stamp) * 4 * 4096);
    // spheres far off the rays: every discriminant is negative
    const float4 s0 = make_float4(50, 60, 70, 0.25f), s1 = make_float4(-50, 60, 70, 0.25f), s2 = make_float4(50, -60, 70, 0.25f),
                 s3 = make_float4(50, 60, -70, 0.25f);
    for (int blocks : {256, 512, 1024, 2048}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k_terms, dim3(blocks), dim3(256), 0, 0, d, s0, s1, s2, s3, 0.0f, d_stamps);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_terms, dim3(blocks), dim3(256), 0, 0, d, s0, s1, s2, s3, 0.0f, d_stamps);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<Stamp> stamps(4 * blocks);
        (void)hipMemcpy(stamps.data(), d_stamps, sizeof(Stamp) * stamps.size(), hipMemcpyDeviceToHost);
        double cycles = 0, real = 0;
        for (const Stamp& st : stamps) { cycles += (double)st.cycles; real += (double)st.real; }
        const double clock_ghz = cycles / real * 0.1;
        const int waves = blocks / 256;
        // wall time of the launch x clock held = SIMD cycles; each SIMD ran `waves` waves of ITER groups of 4 spheres
        const double simd_cycles = ms * 1e-3 * clock_ghz * 1e9;
        printf("%d waves/SIMD: %.2f SIMD cycles per sphere and wave (wall), %.2f by wave stamps / waves, clock %.2f GHz\n", waves,
               simd_cycles / (waves * ITER * 4.0), cycles / stamps.size() / (ITER * 4.0) / waves, clock_ghz);
    }
    return 0;
}
