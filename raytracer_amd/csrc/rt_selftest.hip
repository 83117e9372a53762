// Device self-tests of arithmetic building blocks of rt_kernels.hpp, exported from their own small
// library (raytracer_amd/librt_selftest.so) so the product library's ABI stays the renderer only.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "rt_kernels.hpp"

namespace {

__device__ __forceinline__ uint64_t splitmix(uint64_t& s) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Random float with a uniformly random mantissa and sign, exponent uniform in [exp_lo, exp_hi] (biased).
__device__ __forceinline__ float random_float(uint64_t& s, uint32_t exp_lo, uint32_t exp_hi) {
    uint64_t r = splitmix(s);
    uint32_t mant = (uint32_t)r & 0x007FFFFFu;
    uint32_t sign = (uint32_t)(r >> 32) & 0x80000000u;
    uint32_t e = exp_lo + (uint32_t)((r >> 40) % (exp_hi - exp_lo + 1));
    return __uint_as_float(sign | (e << 23) | mant);
}

// counts[0] = pairs whose packed quotient differs from the IEEE quotient, counts[1] = pairs tested.
__global__ void div2_kernel(uint64_t seed, uint32_t per_thread, uint32_t exp_lo, uint32_t exp_hi,
                            unsigned long long* counts, float* first_bad) {
    uint64_t s = seed + 0x632BE59BD9B4E019ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned long long bad = 0;
    for (uint32_t i = 0; i < per_thread; ++i) {
        rt::F2 a = rt::f2(random_float(s, exp_lo, exp_hi), random_float(s, exp_lo, exp_hi));
        rt::F2 b = rt::f2(random_float(s, exp_lo, exp_hi), random_float(s, exp_lo, exp_hi));
        rt::F2 q = rt::div2(a, b);
        float rx = a.x / b.x, ry = a.y / b.y;   // compiler's correctly rounded division
        bool mx = __float_as_uint(q.x) != __float_as_uint(rx), my = __float_as_uint(q.y) != __float_as_uint(ry);
        if (mx || my) {
            if (bad == 0 && atomicAdd(&counts[2], 1ull) == 0) {
                first_bad[0] = mx ? a.x : a.y; first_bad[1] = mx ? b.x : b.y;
                first_bad[2] = mx ? q.x : q.y; first_bad[3] = mx ? rx : ry;
            }
            bad += (mx ? 1 : 0) + (my ? 1 : 0);
        }
    }
    atomicAdd(&counts[0], bad);
    atomicAdd(&counts[1], 2ull * per_thread);
}

// Every float whose bit pattern lies in [first, first + count): inv_sqrt_exact and sqrt_exact against the
// library operations they stand for.  NaN results count as equal to each other.
__global__ void inv_sqrt_kernel(uint32_t first, uint64_t count, unsigned long long* counts, float* first_bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const float x = __uint_as_float(first + (uint32_t)i);
        const float fast = rt::inv_sqrt_exact(x);
        const float slow = 1.0f / __builtin_sqrtf(x);
        const float root_fast = rt::sqrt_exact(x), root_slow = __builtin_sqrtf(x);
        const bool same = (__float_as_uint(fast) == __float_as_uint(slow) || (fast != fast && slow != slow)) &&
                          (__float_as_uint(root_fast) == __float_as_uint(root_slow) || (root_fast != root_fast && root_slow != root_slow));
        if (!same) {
            if (bad == 0 && atomicAdd(&counts[2], 1ull) == 0) { first_bad[0] = x; first_bad[1] = fast; first_bad[2] = slow; }
            ++bad;
        }
    }
    atomicAdd(&counts[0], bad);
}

}  // namespace

extern "C" {

// Compares rt::div2 with IEEE division on `blocks * 256 * per_thread * 2` random quotients whose
// operands have biased exponents in [exp_lo, exp_hi].  Returns 0 on success (HIP error code otherwise);
// *mismatches receives the number of differing quotients, first_bad[4] = {a, b, div2, ieee} of one of them.
int rt_selftest_div2(uint64_t seed, uint32_t blocks, uint32_t per_thread, uint32_t exp_lo, uint32_t exp_hi,
                     uint64_t* mismatches, uint64_t* tested, float* first_bad) {
    unsigned long long* d_counts = nullptr;
    float* d_bad = nullptr;
    hipError_t e = hipMalloc(&d_counts, 3 * sizeof(unsigned long long));
    if (e != hipSuccess) return (int)e;
    e = hipMalloc(&d_bad, 4 * sizeof(float));
    if (e != hipSuccess) return (int)e;
    (void)hipMemset(d_counts, 0, 3 * sizeof(unsigned long long));
    (void)hipMemset(d_bad, 0, 4 * sizeof(float));
    hipLaunchKernelGGL(div2_kernel, dim3(blocks), dim3(256), 0, 0, seed, per_thread, exp_lo, exp_hi, d_counts, d_bad);
    e = hipDeviceSynchronize();
    unsigned long long h[3] = {0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, d_counts, sizeof h, hipMemcpyDeviceToHost);
    if (e == hipSuccess && first_bad) e = hipMemcpy(first_bad, d_bad, 4 * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_counts);
    (void)hipFree(d_bad);
    if (mismatches) *mismatches = h[0];
    if (tested) *tested = h[1];
    return (int)e;
}

// Compares rt::inv_sqrt_exact with 1.0f / sqrtf and rt::sqrt_exact with sqrtf on the `count` floats whose bit patterns start at
// `first_bits` (count = 2^32 from 0 covers every float).  Returns 0 on success; *mismatches receives the
// number of differing results, first_bad[3] = {x, inv_sqrt_exact, library} of one of them.
int rt_selftest_inv_sqrt(uint32_t first_bits, uint64_t count, uint64_t* mismatches, float* first_bad) {
    unsigned long long* d_counts = nullptr;
    float* d_bad = nullptr;
    hipError_t e = hipMalloc(&d_counts, 3 * sizeof(unsigned long long));
    if (e != hipSuccess) return (int)e;
    e = hipMalloc(&d_bad, 4 * sizeof(float));
    if (e != hipSuccess) { (void)hipFree(d_counts); return (int)e; }
    (void)hipMemset(d_counts, 0, 3 * sizeof(unsigned long long));
    (void)hipMemset(d_bad, 0, 4 * sizeof(float));
    hipLaunchKernelGGL(inv_sqrt_kernel, dim3(8192), dim3(256), 0, 0, first_bits, count, d_counts, d_bad);
    e = hipDeviceSynchronize();
    unsigned long long h[3] = {0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, d_counts, sizeof h, hipMemcpyDeviceToHost);
    if (e == hipSuccess && first_bad) e = hipMemcpy(first_bad, d_bad, 3 * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_counts);
    (void)hipFree(d_bad);
    if (mismatches) *mismatches = h[0];
    return (int)e;
}

}  // extern "C"
