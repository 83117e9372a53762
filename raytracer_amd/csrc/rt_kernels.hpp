// gfx950 device code of the path-tracing hot path.
//
// Execution shape: one lane carries one random stream and traces the samples
// that stream owns, one path after another (RT_RNG_PIXEL: the samples of one
// (pixel, batch); RT_RNG_ROW_SERIAL: every sample of a WorkOrder's rows, as the
// reference's worker thread does, main.cpp:370-392).  The 64 lanes of a wave
// step through ONE loop whose body is "start a path if idle; intersect; shade";
// a lane whose path ended picks up its next sample in the next iteration, so
// lanes at different bounce depths and different samples still execute the
// intersection - the only expensive part - together.  The wave leaves the loop
// when a ballot shows no lane has work.  The scene image is staged once per
// workgroup into LDS and every primitive record is read with a wave-uniform
// (broadcast) address: lanes are rays, the loop runs over primitives, where the
// reference puts primitives in SIMD lanes and loops over rays.
//
// Arithmetic: this file must be compiled with -ffp-contract=off.  Every scalar
// operation then rounds once, in the reference's source order, and the only
// fused multiply-adds are the __builtin_fmaf calls that stand where the reference
// calls FMulAdd/FMulSub (simd.h:91-93,200-202; main.cpp:67,86).  Division and
// square root are the correctly rounded forms.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>

#include "rt_layout.hpp"

namespace rt {

// The lanes of the wave for which `predicate` holds.  (The compiler's own ballot takes the predicate as a mask
// in scalar registers; hip's __ballot goes through an integer per lane and a vector compare.)
__device__ __forceinline__ unsigned long long ballot(bool predicate) { return __builtin_amdgcn_ballot_w64(predicate); }

struct KernelParams {
    uint32_t width, height, spp;
    uint32_t local_rows, row_begin, band_rows, band_index, band_count;
    uint32_t max_bounces;
    float min_hit, plane_tol;
    uint32_t seed_constant;
    uint32_t rows_per_order, order_count;               // ROW_SERIAL
    uint32_t lanes_per_pixel, samples_per_lane, batch_offset;  // PIXEL
    uint32_t tile_w, tile_h, tiles_x, tiles_y;          // PIXEL: pixels of one wave
    uint32_t output_sum, accumulate;                    // radiance receives the sample sum / is added to
    uint32_t pool_threshold, pool_max_samples, pool_capacity;             // tail compaction (STAGE 1 and 2), see render_kernel
    uint32_t lds_words;                                 // leading part of the scene image staged in LDS
    uint32_t sphere_count, sphere_offset, sphere_tag_offset;   // sphere sections of the global image (words)
    float cam_pos[3], cam_x[3], cam_y[3], film_center[3];
    float half_w, half_h, pixel_w, pixel_h;
    float cull_margin;      // PIXEL: > 0 enables the per-wave primary-ray cull; film-plane widening of a tile's footprint (see primary_pair_mask)
    float cull_reach;       // upper bound on the length of an un-normalised primary direction (film at distance 1)
    unsigned long long rect_about_y, rect_about_x;   // rectangle slots (bit k: slot k < 64) by sparsity class, see rect_sparsity_class
};

// Device workspace of the tail compaction: per-stream sums, the pool of unfinished streams and its fill count.
struct Workspace {
    float4* lane_sums;        // [pixel * lanes_per_pixel + batch]: finished sum of that stream (xyz), pending pixels only
    uint8_t* pending;         // [pixel]: 1 if the pixel has to be assembled from lane_sums
    uint4* pool;              // POOL_VEC4 x uint4 per deposited stream
    uint32_t* pool_count;     // streams deposited by STAGE 1
    unsigned long long* ray_slots;   // RAY_SLOTS partial ray counters, RAY_SLOT_STRIDE words apart (see render_kernel's last lines)
};
constexpr uint32_t RAY_SLOTS = 64, RAY_SLOT_STRIDE = 8;   // 64 counters, 64 bytes apart
// A pool entry: sid state samples_left bounce | sum.xyz ro.x | ro.yz dir.xy | dir.z att.xyz | result.xyz 0
constexpr uint32_t POOL_VEC4 = 5;
// Per-wave LDS slice of worlds with spheres: 64 staged sphere records (16 B each), then their 64 tags.
constexpr uint32_t SPHERE_SLICE_BYTES = 1280u;

// Diagnostic build only (-DRT_STAMPS, never the shipped library): every wave adds up the shader-clock
// cycles it spends in each region of the render loop, and a few event counts, and adds them to a global
// table when it ends.  The stamps cost about a tenth of the run time, so the table is read as shares.
enum : int { ST_START = 0, ST_CONTROL, ST_PLANES, ST_SPHERE_TERMS, ST_SPHERE_ROOTS, ST_RECTS, ST_RESOLVE, ST_SHADE, ST_EPILOGUE,
             CT_ITERATIONS, CT_LANES_TRACING, CT_ROOTS_BODIES, CT_ROOTS_LANES, CT_START_LANES, CT_HIT_LANES, CT_DIELECTRIC_BODIES,
             CT_DIELECTRIC_LANES, CT_RECT_PAIRS, CT_RECT_PAIRS_BEHIND, CT_RECTS_OUT_OF_RANGE, CT_RECT_RANGE_LANES,
             CT_RECT_PAIRS_SKIPPED, CT_RECT_PAIR_RAYS, ST_COUNT };
constexpr int MAX_STAMP_WAVES = 16;
#ifdef RT_STAMPS
__device__ unsigned long long rt_stamp_table[ST_COUNT];
// The wave's sums live in LDS and are updated by the first ACTIVE lane, so that marks inside divergent code
// are recorded whichever lanes happen to run it.
struct Stamps {
    unsigned long long* slot;   // [ST_COUNT + 1] of this wave; the last word is the time of the previous mark
    __device__ __forceinline__ bool first_active() const {
        return (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(ballot(true));
    }
    __device__ __forceinline__ void begin() {
        __shared__ unsigned long long table[MAX_STAMP_WAVES][ST_COUNT + 1];
        slot = table[threadIdx.x >> 6];
        if ((threadIdx.x & 63u) == 0) {
            for (int i = 0; i < ST_COUNT; ++i) slot[i] = 0;
            slot[ST_COUNT] = __builtin_amdgcn_s_memtime();
        }
    }
    __device__ __forceinline__ void mark(int region) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if (first_active()) { slot[region] += now - slot[ST_COUNT]; slot[ST_COUNT] = now; }
    }
    __device__ __forceinline__ void count(int counter, unsigned long long n) { if (first_active()) slot[counter] += n; }
    __device__ __forceinline__ void flush() {
        if ((threadIdx.x & 63u) == 0)
            for (int i = 0; i < ST_COUNT; ++i) atomicAdd(&rt_stamp_table[i], slot[i]);
    }
};
#define RT_LANES(pred) ((unsigned long long)__builtin_popcountll(ballot(pred)))
#else
struct Stamps {
    __device__ __forceinline__ void begin() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void count(int, unsigned long long) {}
    __device__ __forceinline__ void flush() {}
};
#define RT_LANES(pred) 0ull
#endif

// Which primitive loops a kernel variant contains (planes are always there): a world without spheres runs a
// variant without the sphere code and its registers, and the other way round.  PRIMS_BOTH in WIDE8 mode
// also orders sphere/rectangle ties by the reference's AVX lane (see intersect_world).
// PRIMS_FEW_SPHERES: spheres only, at most 64 of them - all records fit the wave's LDS slice, which is then
// filled once per wave instead of once per ray, and hits read centre and tag from it instead of from global
// memory; no register set is spent on reading ahead, so more waves fit a SIMD.
// PRIMS_CLASSED_RECTS: rectangles only, some of them turned about one axis (rt_layout.hpp, rect_sparsity_class):
// the rectangle loop dispatches on the class of every rectangle; the plain variants run the full transforms.
enum : int { PRIMS_RECTS = 1, PRIMS_SPHERES = 2, PRIMS_BOTH = 3, PRIMS_FEW = 4, PRIMS_FEW_SPHERES = PRIMS_SPHERES | PRIMS_FEW,
             PRIMS_CLASSED = 8, PRIMS_CLASSED_RECTS = PRIMS_RECTS | PRIMS_CLASSED };

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float f) { return mk(a.x * f, a.y * f, a.z * f); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
// math_vector.h:153-155: products and sums rounded one by one, left to right.
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// simd.h:91-93: fma(x,x', fma(y,y', z*z')).
__device__ __forceinline__ float dot_fma(V3 a, V3 b) {
    return __builtin_fmaf(a.x, b.x, __builtin_fmaf(a.y, b.y, a.z * b.z));
}
// Square root and reciprocal square root without the compiler's operand scaling, class tests and
// division fix-up.  Those are the identity for x in [2^-96, 2^96]; other arguments (zero, denormal,
// huge, inf, NaN) take the library forms.  A device self-test compares both functions with the library
// forms on every float.
__device__ __forceinline__ bool in_fast_sqrt_range(float x) {   // 2^-96 <= x <= 2^96
    return __float_as_uint(x) - 0x0F800000u <= 0x6F800000u - 0x0F800000u;
}
// The correctly rounded square root of x in the fast range: the hardware estimate is within one ulp,
// and the residuals x - (s -+ 1ulp) * s say which neighbour, if any, is the rounded root.
__device__ __forceinline__ float sqrt_fast_range(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    float root = s;
    if (__builtin_fmaf(-s_dn, s, x) <= 0.0f) root = s_dn;
    if (__builtin_fmaf(-s_up, s, x) > 0.0f) root = s_up;
    return root;
}
// sqrtf(x), correctly rounded for every argument.  The fast form is evaluated unconditionally (outside
// its range it yields a value that is thrown away) and the library form replaces it under a branch
// that no lane of a wave takes in practice, which keeps the common path free of an else-branch.
__device__ __forceinline__ float sqrt_exact(float x) {
    float root = sqrt_fast_range(x);
    if (!in_fast_sqrt_range(x)) {
        asm volatile("" ::: "memory");   // keep this a branch: the compiler would otherwise evaluate both forms and select
        root = __builtin_sqrtf(x);
    }
    return root;
}
// 1.0f / sqrtf(x) with both operations correctly rounded, i.e. exactly what the two library calls give:
// the reciprocal is the division refinement of div2 with numerator 1 (the first quotient estimate is the
// refined reciprocal itself).
__device__ __forceinline__ float inv_sqrt_exact(float x) {
    const float root = sqrt_fast_range(x);
    float y = __builtin_amdgcn_rcpf(root);
    y = __builtin_fmaf(__builtin_fmaf(-root, y, 1.0f), y, y);
    float q = __builtin_fmaf(__builtin_fmaf(-root, y, 1.0f), y, y);
    q = __builtin_fmaf(__builtin_fmaf(-root, q, 1.0f), y, q);
    if (!in_fast_sqrt_range(x)) {
        asm volatile("" ::: "memory");   // as in sqrt_exact
        q = 1.0f / __builtin_sqrtf(x);
    }
    return q;
}
// math_vector.h:167-171: v * (1 / sqrtf(v.v)).
__device__ __forceinline__ V3 normalize(V3 v) {
    const float f = inv_sqrt_exact(dot(v, v));
    return mk(v.x * f, v.y * f, v.z * f);
}
__device__ __forceinline__ V3 load3(const float* p) { return mk(p[0], p[1], p[2]); }

// Two independent floats handled side by side (two rectangles of a pair, the two roots of a sphere).
// Plain scalar instructions: a v_pk_* instruction does two lanes' worth of work in twice the issue time
// of a scalar v_fma/v_add/v_mul on this chip, so packing buys no throughput, and the scalar form
// schedules better (measured on the Cornell bench: 19.4 ms against 20.2 ms with v_pk_* arithmetic).
struct F2 { float x, y; };
__device__ __forceinline__ F2 f2(float a, float b) { F2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ F2 splat(float a) { return f2(a, a); }
__device__ __forceinline__ F2 fma2(F2 a, F2 b, F2 c) { return f2(__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)); }
__device__ __forceinline__ F2 operator+(F2 a, F2 b) { return f2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ F2 operator*(F2 a, F2 b) { return f2(a.x * b.x, a.y * b.y); }
__device__ __forceinline__ F2 operator-(F2 a) { return f2(-a.x, -a.y); }

// Two correctly rounded quotients a/b at once.  This is the refinement the compiler emits for
// an IEEE float division (reciprocal estimate, one Newton step on it, quotient, two residual
// corrections) without its operand pre-scaling and special-case fix-up.  It returns the IEEE quotient
// whenever that scaling would have been the identity: b normal, |a| >= 2^-100 or a == 0, and
// 2^-100 < |a/b| < 2^96.  Outside that range (a ray parallel to a rectangle's plane, a
// zero-padded slot) the result is inf, NaN or a value of the true quotient's huge or tiny
// magnitude; the caller's tests (t > min_hit, t < closest, |h| <= 1) reject all of those just
// as they reject the IEEE quotient there, given finite matrices of rank 3.
__device__ __forceinline__ F2 div2(F2 a, F2 b) {
    F2 y = f2(__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y));
    F2 e = fma2(-b, y, splat(1.0f));
    y = fma2(e, y, y);
    F2 q = a * y;
    F2 r = fma2(-b, q, a);
    q = fma2(r, y, q);
    r = fma2(-b, q, a);
    return fma2(r, y, q);
}

// One such quotient.
__device__ __forceinline__ float div1(float a, float b) {
    float y = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    float q = a * y;
    float r = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r, y, q);
}

// A row (m0 m1 m2 m3) of a rectangle's inverse transform applied to a point v (w = 1), in the arithmetic of the
// mode: WIDE8 is DotProduct(LaneVector4), fma(x,x', fma(y,y', fma(z,z', w*w'))) with w*1 exact; the others are
// DotProduct(Vector4), x*x' + y*y' + z*z' + w*w' from left to right (math_vector.h:341-343).  Zk (compile time):
// mk is an exact zero and its term is left out - the term is a zero of either sign (v is finite), and adding it
// would change at most the sign of a sum that is itself zero.
template <int IMODE, bool Z0, bool Z1, bool Z2>
__device__ __forceinline__ float affine_row(float m0, float m1, float m2, float m3, V3 v) {
    if (IMODE == RT_INTERSECT_WIDE8) {
        float acc = m3;
        if (!Z2) acc = __builtin_fmaf(m2, v.z, acc);
        if (!Z1) acc = __builtin_fmaf(m1, v.y, acc);
        if (!Z0) acc = __builtin_fmaf(m0, v.x, acc);
        return acc;
    }
    float acc = 0.0f;
    bool have = false;
    if (!Z0) { acc = m0 * v.x; have = true; }
    if (!Z1) { acc = have ? acc + m1 * v.y : m1 * v.y; have = true; }
    if (!Z2) { acc = have ? acc + m2 * v.z : m2 * v.z; have = true; }
    return have ? acc + m3 : m3;
}
// The same row applied to a direction (w = 0: the reference adds a zero there, which is left out as well).
template <int IMODE, bool Z0, bool Z1, bool Z2>
__device__ __forceinline__ float linear_row(float m0, float m1, float m2, V3 v) {
    float acc = 0.0f;
    bool have = false;
    if (IMODE == RT_INTERSECT_WIDE8) {
        if (!Z2) { acc = m2 * v.z; have = true; }
        if (!Z1) { acc = have ? __builtin_fmaf(m1, v.y, acc) : m1 * v.y; have = true; }
        if (!Z0) { acc = have ? __builtin_fmaf(m0, v.x, acc) : m0 * v.x; have = true; }
        return acc;
    }
    if (!Z0) { acc = m0 * v.x; have = true; }
    if (!Z1) { acc = have ? acc + m1 * v.y : m1 * v.y; have = true; }
    if (!Z2) { acc = have ? acc + m2 * v.z : m2 * v.z; have = true; }
    return acc;
}

// math_util.h:32-50.  (float)U32Max is 2^32, so the division is an exact scaling.
__device__ __forceinline__ uint32_t xorshift32(uint32_t& s) {
    uint32_t x = s;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    s = x;
    return x;
}
__device__ __forceinline__ float random_unilateral(uint32_t& s) { return (float)xorshift32(s) * 0x1.0p-32f; }
// 2u - 1: u = x * 2^-32 and 2u are exact scalings, so the one rounding of fma(x, 2^-31, -1) is
// the one rounding of the reference's subtraction.
__device__ __forceinline__ float random_bilateral(uint32_t& s) { return __builtin_fmaf((float)xorshift32(s), 0x1.0p-31f, -1.0f); }

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
// Stream of (pixel, batch): rooted in the reference's seed for the one-row
// WorkOrder [y, y+1) (main.cpp:370); never zero (math_util.h:32).
__device__ __forceinline__ uint32_t pixel_stream_seed(uint32_t seed_constant, uint32_t x, uint32_t y, uint32_t batch) {
    uint32_t root = seed_constant * (y + (y + 1u) * 36u);
    uint32_t h = fmix32(root + 0x9E3779B9u * (x + 1u));
    h = fmix32(h ^ (0x85EBCA6Bu * (batch + 1u)));
    return h ? h : 0x9E3779B9u;
}

__device__ __forceinline__ float clampf(float lo, float v, float hi) {  // math_util.h:14-22
    if (v < lo) return lo;
    else if (v > hi) return hi;
    else return v;
}

// math_util.h:102-125.  powf is evaluated in double and rounded once.
__device__ __forceinline__ float linear_to_srgb(float value) {
    value = clampf(0.0f, value, 1.0f);
    float result = value * 12.92f;
    if (value >= 0.0031308f) {
        float p = (float)pow((double)value, (double)(1.0f / 2.4f));
        result = (1.055f * p) - 0.055f;
    }
    return result;
}
__device__ __forceinline__ uint32_t pack_srgb(V3 c) {
    return (255u << 24) | ((uint32_t)(int32_t)(255 * linear_to_srgb(c.x)) << 16) |
           ((uint32_t)(int32_t)(255 * linear_to_srgb(c.y)) << 8) | ((uint32_t)(int32_t)(255 * linear_to_srgb(c.z)) << 0);
}

__device__ __forceinline__ uint32_t global_row(const KernelParams& p, uint32_t local) {
    if (p.band_rows == 0 || p.band_count <= 1) return p.row_begin + local;
    uint32_t band = local / p.band_rows, within = local % p.band_rows;
    return p.row_begin + (band * p.band_count + p.band_index) * p.band_rows + within;
}

struct SceneView {
    const float* f;       // LDS image as floats
    const uint32_t* u;    // ... and as words
    const float4* gspheres;   // sphere records (cx cy cz r*r) in the global image
    float4* wave_spheres;     // this wave's LDS slice: 64 staged sphere records, then (PRIMS_FEW) their 64 tags
    const uint32_t* gtags;    // sphere tags (material | lane << 24) in the global image
    uint32_t n_planes, n_spheres, n_rects;
    uint32_t off_planes, off_rects, off_mats;
    unsigned long long rect_about_y, rect_about_x;   // KernelParams' masks (scalar registers)
};

// World intersection.  best_id: 0 = miss, 1+i plane i, 1+np+k sphere slot k,
// 1+np+ns+k rectangle slot k (slots in the image's order).
struct Closest { float t; uint32_t id; };

// Closest-hit update of the rectangle test (main.cpp:121-127 / 229-237):
//   if (|hx| <= 1 && |hy| <= 1 && t < best_t && t > min_hit) { best_t = t; best_id = id; }
// written as four v_cmpx that narrow EXEC and two moves under the narrowed mask, instead of the
// compiler's four compares into scalar masks, three scalar ANDs and two v_cndmask: same predicate,
// same NaN behaviour (every comparison is false on NaN), about a quarter fewer issue cycles.  EXEC is
// saved and restored inside the block, so it may sit in divergent code.  `min_hit` and `id` are
// wave-uniform (scalar registers).
__device__ __forceinline__ void update_closest_rect(float hx, float hy, float t, float min_hit, uint32_t id, float& best_t,
                                                    uint32_t& best_id) {
    unsigned long long saved_exec;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "v_cmpx_lt_f32_e32 %[t], %[best]\n\t"
        "v_cmpx_ge_f32_e64 exec, 1.0, |%[hx]|\n\t"
        "v_cmpx_ge_f32_e64 exec, 1.0, |%[hy]|\n\t"
        "v_cmpx_lt_f32_e32 %[minhit], %[t]\n\t"
        "v_mov_b32_e32 %[best], %[t]\n\t"
        "v_mov_b32_e32 %[bid], %[id]\n\t"
        "s_mov_b64 exec, %[save]"
        : [save] "=&s"(saved_exec), [best] "+v"(best_t), [bid] "+v"(best_id)
        : [t] "v"(t), [hx] "v"(hx), [hy] "v"(hy), [minhit] "s"(min_hit), [id] "s"(id)
        : "vcc");
}

// The same for a sphere root: if (t > min_hit && t < best_t) { best_t = t; best_id = id; best_lane = lane; }
// (main.cpp:78-90 / 196-206); best_lane is only kept by the MIXED variant.
template <bool MIXED>
__device__ __forceinline__ void update_closest_sphere(float t, float min_hit, uint32_t id, uint32_t lane, float& best_t,
                                                      uint32_t& best_id, uint32_t& best_lane) {
    unsigned long long saved_exec;
    if (MIXED) {
        asm volatile(
            "s_mov_b64 %[save], exec\n\t"
            "v_cmpx_lt_f32_e32 %[t], %[best]\n\t"
            "v_cmpx_lt_f32_e32 %[minhit], %[t]\n\t"
            "v_mov_b32_e32 %[best], %[t]\n\t"
            "v_mov_b32_e32 %[bid], %[id]\n\t"
            "v_mov_b32_e32 %[blane], %[lane]\n\t"
            "s_mov_b64 exec, %[save]"
            : [save] "=&s"(saved_exec), [best] "+v"(best_t), [bid] "+v"(best_id), [blane] "+v"(best_lane)
            : [t] "v"(t), [minhit] "s"(min_hit), [id] "s"(id), [lane] "v"(lane)
            : "vcc");
    } else {
        asm volatile(
            "s_mov_b64 %[save], exec\n\t"
            "v_cmpx_lt_f32_e32 %[t], %[best]\n\t"
            "v_cmpx_lt_f32_e32 %[minhit], %[t]\n\t"
            "v_mov_b32_e32 %[best], %[t]\n\t"
            "v_mov_b32_e32 %[bid], %[id]\n\t"
            "s_mov_b64 exec, %[save]"
            : [save] "=&s"(saved_exec), [best] "+v"(best_t), [bid] "+v"(best_id)
            : [t] "v"(t), [minhit] "s"(min_hit), [id] "s"(id)
            : "vcc");
    }
}

// Primary-ray cull.  The pixels of a wave form a fixed tile of the image, so every primary ray the wave will
// ever trace leaves the camera position through one small patch of the film: the tile's pixel centres, the
// jitter of a quarter pixel around them (main.cpp:381-382), and a widening `cull_margin` (one pixel plus a bound
// on the rounding of the kernel's own ray set-up) chosen by the host.  In un-normalised direction space the
// patch is the parallelogram D(a, b) = D0 + a * camX + b * camY over a box [a_lo, a_hi] x [b_lo, b_hi].  For
// every rectangle the host has reduced the question "can a ray of direction D(a, b) from the camera pass the
// rectangle test of the loop below (main.cpp:96-129)?" to the signs of five functions that are LINEAR in
// (a, b) (cull table of the scene image, rt_layout.hpp): a linear function takes its extremes over a box at
// the corners picked by the signs of its two slopes, so each costs two multiply-adds here.  Lane j decides for
// rectangle pair j; the ballot of the answers is the set of pairs that iterations in which all rays are
// primary rays have to visit.  Anything not provably missed is kept, and culled pairs can never replace the
// closest hit, so skipping them changes neither the winner nor any later arithmetic.
__device__ __forceinline__ unsigned long long primary_pair_mask(const double* __restrict__ cull_table, const SceneView& sv,
                                                                const KernelParams& p, uint32_t x0, uint32_t x1, uint32_t row0,
                                                                uint32_t row1) {
    const uint32_t lane = threadIdx.x & 63u, n_pairs = sv.n_rects / 2u;
    const CullBox box = cull_box(p.width, p.height, p.pixel_w, p.pixel_h, p.half_w, p.half_h, p.cull_margin, x0, x1, row0, row1);
    bool keep = false;
#pragma unroll 1
    for (uint32_t which = 0; which < (lane < n_pairs ? 2u : 0u); ++which)
        keep = keep || cull_keeps(cull_table + (size_t)CULL_DOUBLES * (2u * lane + which), box, (double)p.cull_reach);
    return ballot(keep);
}

// `camera_origin` (wave-uniform): every lane's ray starts at the camera position, so the rectangles'
// origin transforms are read from the scene image, where the host stored them, instead of computed.
// `primary_pairs` (wave-uniform): the rectangle pairs such an iteration has to visit (bit i: pair i; pairs
// from 64 on are always visited).
// Variants with spheres must be called by ALL 64 lanes of the wave (the lanes stage the sphere records
// together); `active` says whether this lane has a ray, and only active lanes test anything.
template <int IMODE, int PRIMS>
__device__ __forceinline__ Closest intersect_world(const SceneView& sv, V3 o, V3 d, float min_hit, float plane_tol,
                                                   bool camera_origin, unsigned long long primary_pairs, bool active, Stamps& st) {
    constexpr bool MIXED = PRIMS == PRIMS_BOTH && IMODE == RT_INTERSECT_WIDE8;
    float best_t = 3.402823466e+38f;  // F32Max
    uint32_t best_id = 0;
    uint32_t best_lane = 0;     // WIDE8 + MIXED: AVX lane of the best hit when it is a sphere

    // Planes: scalar code in both reference routines (main.cpp:37-50, 168-180).
    for (uint32_t i = 0; i < (active ? sv.n_planes : 0u); ++i) {
        const float4 pl = *reinterpret_cast<const float4*>(sv.f + sv.off_planes + PLANE_WORDS * i);
        V3 n = mk(pl.x, pl.y, pl.z);
        float denom = dot(n, d);
        if ((denom < -plane_tol) || (denom > plane_tol)) {
            float t = (-pl.w - dot(n, o)) / denom;
            if (t > min_hit && t < best_t) { best_t = t; best_id = 1 + i; }
        }
    }

    st.mark(ST_PLANES);
    // Spheres (main.cpp:60-91 / 182-207).  Every lane of the wave tests the same sphere, so the records
    // (centre, r*r) are wanted in vector registers as wave-uniform values: arithmetic on vector registers
    // only issues at twice the rate of instructions that read a scalar register on this chip.  The wave
    // therefore stages 64 records at a time in its own 1 KiB slice of LDS - lane j loads record j of the
    // chunk from global memory (one coalesced 1 KiB load, requested one chunk ahead) and stores it - and
    // reads them back one by one with a wave-uniform address (a broadcast ds_read_b128).  LDS operations
    // of one wave execute in order, so the slice needs no barrier and no second buffer.  The part every
    // sphere pays (b, c, discriminant) is evaluated for four spheres at a time; the roots are only
    // computed, sphere by sphere in slot order, when some lane of the wave needs them.
    if ((PRIMS & PRIMS_SPHERES) && sv.n_spheres) {
        const float a = IMODE == RT_INTERSECT_WIDE8 ? dot_fma(d, d) : dot(d, d);
        const float denom = 2.0f * a;
        const float minus_four_a = -4.0f * a;   // -(4a * c) is (-4a) * c: negation commutes with the rounding
        // b = 2 * dot(d, cto) is computed as dot(2d, cto): scaling by two commutes with every rounding, unless a
        // product or partial sum is subnormal (a direction component times a centre offset below 1.2e-38, say
        // 1e-30 x 1e-9): such a term's last bit can then differ from the reference's, which reaches b only when all
        // three terms are that small.  The same precondition holds for minus_four_a * c above.
        const V3 d2 = mk(2.0f * d.x, 2.0f * d.y, 2.0f * d.z);
        // Both roots are divided by the same 2a: the reciprocal estimate and its Newton step (the
        // first three operations of the IEEE division refinement, see div2) are done once per ray.
        float recip = __builtin_amdgcn_rcpf(denom);
        recip = __builtin_fmaf(__builtin_fmaf(-denom, recip, 1.0f), recip, recip);
        const F2 denom2 = splat(denom), recip2 = splat(recip);
        const uint32_t sphere_base = 1 + sv.n_planes;

        // b, c and the discriminant of one sphere (main.cpp:66-72 / 188-192).
        auto terms = [&](const float4 sp, float& b, float& c, float& disc) {
            V3 cto = o - mk(sp.x, sp.y, sp.z);
            if (IMODE == RT_INTERSECT_WIDE8) {
                b = dot_fma(d2, cto);
                c = dot_fma(cto, cto) - sp.w;
                disc = __builtin_fmaf(b, b, minus_four_a * c);     // FMulSub(b, b, 4ac)
            } else {
                b = d2.x * cto.x + d2.y * cto.y + d2.z * cto.z;    // 2 * (d . cto), rounded as the products and sums of d . cto
                c = dot(cto, cto) - sp.w;
                disc = b * b + minus_four_a * c;                   // b*b - 4ac: adding the negated product is the subtraction
            }
        };
        // disc > 0: the ray's line meets the sphere.  With b > 0 and c > 0 (centre behind the
        // origin, origin outside) 4ac > 0, so sqrt(disc) <= sqrt(b*b) = b and both roots are <= 0:
        // the reference computes them and rejects them against min_hit >= 0; skip them unevaluated.
        auto roots = [&](uint32_t k, float b, float c, float disc) {
            if (!(disc > 0.0f) || (b > 0.0f && c > 0.0f)) return;
            st.count(CT_ROOTS_BODIES, 1);
            st.count(CT_ROOTS_LANES, RT_LANES(true));
            const float root = sqrt_exact(disc);
            // (tp, tn) = ((-b + root) / 2a, (-b - root) / 2a), correctly rounded (see div2)
            const F2 num = f2(-b + root, -b - root);
            F2 q = num * recip2;
            F2 r = fma2(-denom2, q, num);
            q = fma2(r, recip2, q);
            r = fma2(-denom2, q, num);
            q = fma2(r, recip2, q);
            // The reference takes the near root tn if (tn > min_hit && tn < tp), else the far root tp.
            // 2a > 0 and root > 0 give tn <= tp after rounding, and when they are equal either choice
            // is the same number, so the second condition never decides.
            const float tp = q.x, tn = q.y;
            const float t = tn > min_hit ? tn : tp;
            update_closest_sphere<MIXED>(t, min_hit, sphere_base + k, MIXED ? sv.gtags[k] >> TAG_LANE_SHIFT : 0u,
                                         best_t, best_id, best_lane);
        };
        // Is any of four discriminants possibly positive?  As integers, the AND of the four has its sign
        // bit set exactly when all four are negative or -0; +0 passes, and `roots` tests each one exactly.
        auto any_candidate = [](float d0, float d1, float d2_, float d3) {
            return (int32_t)(__float_as_uint(d0) & __float_as_uint(d1) & __float_as_uint(d2_) & __float_as_uint(d3)) >= 0;
        };

        if (PRIMS & PRIMS_FEW) {
            // At most 64 spheres: the kernel's prologue staged all of them in the wave's slice, once.
            const float4* rec = sv.wave_spheres;
            const uint32_t groups = sv.n_spheres / 4u;
            for (uint32_t g = 0; g < (active ? groups : 0u); ++g) {
                const float4* r4 = rec + 4u * g;
                const float4 s0 = r4[0], s1 = r4[1], s2 = r4[2], s3 = r4[3];
                float b0, b1, b2, b3, c0, c1, c2, c3, e0, e1, e2, e3;
                terms(s0, b0, c0, e0); terms(s1, b1, c1, e1); terms(s2, b2, c2, e2); terms(s3, b3, c3, e3);
                if (any_candidate(e0, e1, e2, e3)) {
                    st.mark(ST_SPHERE_TERMS);
                    roots(4 * g, b0, c0, e0);
                    roots(4 * g + 1, b1, c1, e1);
                    roots(4 * g + 2, b2, c2, e2);
                    roots(4 * g + 3, b3, c3, e3);
                    st.mark(ST_SPHERE_ROOTS);
                }
            }
            for (uint32_t k = 4 * groups; k < (active ? sv.n_spheres : 0u); ++k) {
                float b, c, disc;
                terms(rec[k], b, c, disc);
                roots(k, b, c, disc);
            }
        } else {
            const uint32_t lane = threadIdx.x & 63u;
            const uint32_t chunks = (sv.n_spheres + 63u) / 64u;
            const uint32_t last = sv.n_spheres - 1u;
            float4 staged = sv.gspheres[min(lane, last)];
            for (uint32_t chunk = 0; chunk < chunks; ++chunk) {
                sv.wave_spheres[lane] = staged;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (chunk + 1 < chunks) staged = sv.gspheres[min(64u * (chunk + 1) + lane, last)];   // next chunk: in flight while this one is tested
                const uint32_t first = 64u * chunk;
                const uint32_t count = min(64u, sv.n_spheres - first);
                const uint32_t groups = count / 4u;
                const float4* rec = sv.wave_spheres;
                if (active) {
                // Groups of four.  The records of the next group are read from LDS before the current group is
                // worked on, so their latency is covered by arithmetic even with few waves on the SIMD; two
                // register sets take turns (A, B), so nothing is copied from "next" to "current".
                auto group = [&](uint32_t g, const float4 s0, const float4 s1, const float4 s2, const float4 s3) {
                    float b0, b1, b2, b3, c0, c1, c2, c3, e0, e1, e2, e3;
                    terms(s0, b0, c0, e0); terms(s1, b1, c1, e1); terms(s2, b2, c2, e2); terms(s3, b3, c3, e3);
                    if (any_candidate(e0, e1, e2, e3)) {
                        st.mark(ST_SPHERE_TERMS);
                        roots(first + 4 * g, b0, c0, e0);
                        roots(first + 4 * g + 1, b1, c1, e1);
                        roots(first + 4 * g + 2, b2, c2, e2);
                        roots(first + 4 * g + 3, b3, c3, e3);
                        st.mark(ST_SPHERE_ROOTS);
                    }
                };
                float4 a0, a1, a2, a3;
                if (groups) { a0 = rec[0]; a1 = rec[1]; a2 = rec[2]; a3 = rec[3]; }
                uint32_t g = 0;
                for (; g + 2u <= groups; g += 2u) {
                    const float4* nb = rec + 4u * (g + 1u);
                    const float4 b0 = nb[0], b1 = nb[1], b2 = nb[2], b3 = nb[3];
                    group(g, a0, a1, a2, a3);
                    const float4* na = rec + 4u * min(g + 2u, groups - 1u);
                    a0 = na[0]; a1 = na[1]; a2 = na[2]; a3 = na[3];
                    group(g + 1u, b0, b1, b2, b3);
                }
                if (g < groups) group(g, a0, a1, a2, a3);
                for (uint32_t k = 4 * groups; k < count; ++k) {
                    float b, c, disc;
                    terms(rec[k], b, c, disc);
                    roots(first + k, b, c, disc);
                }
                }
                __builtin_amdgcn_wave_barrier();   // the next chunk's store stays behind these reads
            }
        }
    }

    st.mark(ST_SPHERE_TERMS);
    // Rectangles (main.cpp:96-129 / 212-239): ray into the rectangle's frame, hit the
    // z = 0 plane, test against the unit square.  The scene image stores them in pairs (rows 0 and 1 of A, of B,
    // then the two row-2 vectors) and a step of the loop tests the two rectangles of a pair.
    //   Plain variants: both full transforms side by side, which gives the scheduler twelve independent
    //   multiply-add chains and two divisions to interleave.
    //   PRIMS_CLASSED (worlds in which some rectangle is turned about one axis, rect_sparsity_class): each
    //   rectangle's transforms by the copy of the arithmetic that matches the exact zeros of its matrix, chosen by
    //   a scalar branch on the host's class masks - 10 multiply-adds instead of 18 for a classed rectangle; the
    //   division, the hit point and the closest-hit update follow once, after the copies have joined.
    constexpr bool CLASSED = (PRIMS & PRIMS_CLASSED) != 0;
    const uint32_t rect_base = 1 + sv.n_planes + sv.n_spheres;
    const uint32_t n_pairs = (PRIMS & PRIMS_RECTS) && active ? sv.n_rects / 2 : 0u;
    // One pair.  CAMERA is a compile-time copy of camera_origin: the loop exists twice, so that neither copy
    // branches on it.  about_y / about_x (CLASSED): sparsity classes of the pair's two slots in bits 0 (A) and
    // 1 (B); higher bits are ignored.
    auto test_pair = [&](uint32_t pair, uint32_t about_y, uint32_t about_x, auto camera_tag) {
        constexpr bool CAMERA = decltype(camera_tag)::value;
        const float4* rec = reinterpret_cast<const float4*>(sv.f + sv.off_rects + RECT_PAIR_WORDS * pair);
        // Row 2 of both rectangles (and the camera rows) are read once per pair.  (CLASSED: reading them per
        // rectangle would free four registers, but the two extra LDS reads per pair cost more than that brings:
        // 16.55 against 16.29 ms on the Cornell bench.)
        const float4 z01 = rec[4], z23 = rec[5];
        float4 c01 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), c2 = c01;
        if (CAMERA) { c01 = rec[8]; c2 = rec[9]; }
        float t_a, t_b;
        // if (min_hit < t < best_t inside the unit square) the rectangle becomes the closest hit
        auto update = [&](int which, float hx, float hy, float t) {
#ifdef RT_STAMPS
            // rectangles that no lane's t passes min_hit < t < closest so far
            st.count(CT_RECTS_OUT_OF_RANGE, ballot(t > min_hit && t < best_t) == 0ull ? 1 : 0);
            st.count(CT_RECT_RANGE_LANES, RT_LANES(t > min_hit && t < best_t));
#endif
            if (!MIXED) {
                update_closest_rect(hx, hy, t, min_hit, rect_base + 2 * pair + which, best_t, best_id);
            } else {
                // A sphere and a rectangle at exactly the same t: the reference keeps the lower
                // AVX lane, and within one lane the sphere, which it tested first.
                const uint32_t lane = sv.u[sv.off_rects + RECT_PAIR_WORDS * pair + 24 + 4 * which + 3] >> TAG_LANE_SHIFT;
                const bool closer = t < best_t || (t == best_t && lane < best_lane);
                if (__builtin_fabsf(hx) <= 1.0f && __builtin_fabsf(hy) <= 1.0f && closer && t > min_hit) {
                    best_t = t;
                    best_id = rect_base + 2 * pair + which;
                    best_lane = 0;
                }
            }
        };
        if (!CLASSED) {
            const float4 a01 = rec[0], a23 = rec[1], b01 = rec[2], b23 = rec[3];
            const F2 ox2 = splat(o.x), oy2 = splat(o.y), oz2 = splat(o.z), dx2 = splat(d.x), dy2 = splat(d.y), dz2 = splat(d.z);
            F2 a_lo, a_ld, b_lo, b_ld, z_lo, z_ld;   // (row0, row1) of A, of B; (A.row2, B.row2)
            if (CAMERA) {
                a_lo = f2(c01.x, c01.y); b_lo = f2(c01.z, c01.w); z_lo = f2(c2.x, c2.y);
            } else if (IMODE == RT_INTERSECT_WIDE8) {
                // DotProduct(LaneVector4): fma(x,x', fma(y,y', fma(z,z', w*w'))) with w' = 1 for the
                // origin (w*1 is exact).
                a_lo = fma2(f2(a01.x, a01.y), ox2, fma2(f2(a01.z, a01.w), oy2, fma2(f2(a23.x, a23.y), oz2, f2(a23.z, a23.w))));
                b_lo = fma2(f2(b01.x, b01.y), ox2, fma2(f2(b01.z, b01.w), oy2, fma2(f2(b23.x, b23.y), oz2, f2(b23.z, b23.w))));
                z_lo = fma2(f2(z01.x, z01.y), ox2, fma2(f2(z01.z, z01.w), oy2, fma2(f2(z23.x, z23.y), oz2, f2(z23.z, z23.w))));
            } else {
                // DotProduct(Vector4): x*x' + y*y' + z*z' + w*w', left to right (math_vector.h:341-343).
                a_lo = f2(a01.x, a01.y) * ox2 + f2(a01.z, a01.w) * oy2 + f2(a23.x, a23.y) * oz2 + f2(a23.z, a23.w);
                b_lo = f2(b01.x, b01.y) * ox2 + f2(b01.z, b01.w) * oy2 + f2(b23.x, b23.y) * oz2 + f2(b23.z, b23.w);
                z_lo = f2(z01.x, z01.y) * ox2 + f2(z01.z, z01.w) * oy2 + f2(z23.x, z23.y) * oz2 + f2(z23.z, z23.w);
            }
            if (IMODE == RT_INTERSECT_WIDE8) {
                // w' = 0 for the direction (adds a zero: only the sign of an exact zero could differ,
                // which no comparison below can see).
                a_ld = fma2(f2(a01.x, a01.y), dx2, fma2(f2(a01.z, a01.w), dy2, f2(a23.x, a23.y) * dz2));
                b_ld = fma2(f2(b01.x, b01.y), dx2, fma2(f2(b01.z, b01.w), dy2, f2(b23.x, b23.y) * dz2));
                z_ld = fma2(f2(z01.x, z01.y), dx2, fma2(f2(z01.z, z01.w), dy2, f2(z23.x, z23.y) * dz2));
            } else {
                a_ld = f2(a01.x, a01.y) * dx2 + f2(a01.z, a01.w) * dy2 + f2(a23.x, a23.y) * dz2;
                b_ld = f2(b01.x, b01.y) * dx2 + f2(b01.z, b01.w) * dy2 + f2(b23.x, b23.y) * dz2;
                z_ld = f2(z01.x, z01.y) * dx2 + f2(z01.z, z01.w) * dy2 + f2(z23.x, z23.y) * dz2;
            }
            const F2 t2 = div2(-z_lo, z_ld);   // t = (-origin'.z) / direction'.z for A and B
            t_a = t2.x; t_b = t2.y;
            const F2 ha = a_lo + a_ld * splat(t_a), hb = b_lo + b_ld * splat(t_b);
            update(0, ha.x, ha.y, t_a);
            update(1, hb.x, hb.y, t_b);
        } else {
            auto one = [&](auto which_tag) -> float {
                constexpr int which = decltype(which_tag)::value;
                const float4 r01 = rec[2 * which], r23 = rec[2 * which + 1];   // (m00 m10 m01 m11) (m02 m12 m03 m13)
                const float m20 = which ? z01.y : z01.x, m21 = which ? z01.w : z01.z, m22 = which ? z23.y : z23.x, m23 = which ? z23.w : z23.z;
                float lo_x, lo_y, lo_z, ld_x, ld_y, ld_z;   // origin and direction in the rectangle's frame
                auto transforms = [&](auto sparsity_tag) {
                    constexpr int S = decltype(sparsity_tag)::value;
                    constexpr bool Z01 = S != 0, Z10 = S != 0, Z12 = S == 1, Z21 = S == 1, Z02 = S == 2, Z20 = S == 2;
                    if (CAMERA) {
                        lo_x = which ? c01.z : c01.x; lo_y = which ? c01.w : c01.y; lo_z = which ? c2.y : c2.x;
                    } else {
                        lo_x = affine_row<IMODE, false, Z01, Z02>(r01.x, r01.z, r23.x, r23.z, o);
                        lo_y = affine_row<IMODE, Z10, false, Z12>(r01.y, r01.w, r23.y, r23.w, o);
                        lo_z = affine_row<IMODE, Z20, Z21, false>(m20, m21, m22, m23, o);
                    }
                    ld_x = linear_row<IMODE, false, Z01, Z02>(r01.x, r01.z, r23.x, d);
                    ld_y = linear_row<IMODE, Z10, false, Z12>(r01.y, r01.w, r23.y, d);
                    ld_z = linear_row<IMODE, Z20, Z21, false>(m20, m21, m22, d);
                };
                if ((about_y >> which) & 1u) transforms(std::integral_constant<int, 1>());
                else if ((about_x >> which) & 1u) transforms(std::integral_constant<int, 2>());
                else transforms(std::integral_constant<int, 0>());
                const float t = div1(-lo_z, ld_z);   // t = (-origin'.z) / direction'.z
                update(which, lo_x + ld_x * t, lo_y + ld_y * t, t);
                return t;
            };
            t_a = one(std::integral_constant<int, 0>());
            t_b = one(std::integral_constant<int, 1>());
        }
#ifdef RT_STAMPS
        // pairs whose two planes lie behind every ray of the wave (t <= 0 on all lanes)
        st.count(CT_RECT_PAIRS, 1);
        st.count(CT_RECT_PAIRS_BEHIND, ballot(t_a > 0.0f || t_b > 0.0f) == 0ull ? 1 : 0);
#else
        (void)t_a; (void)t_b;
#endif
    };
    if (camera_origin) {
        // All rays are primary rays: only the pairs one of them can hit (primary_pair_mask), in increasing order.
        unsigned long long todo = n_pairs >= 64u ? primary_pairs : primary_pairs & ((1ull << n_pairs) - 1ull);
        st.count(CT_RECT_PAIRS_SKIPPED, (n_pairs < 64u ? n_pairs : 64u) - (unsigned long long)__builtin_popcountll(todo));
        st.count(CT_RECT_PAIR_RAYS, RT_LANES(active) * ((unsigned long long)__builtin_popcountll(todo) + (n_pairs > 64u ? n_pairs - 64u : 0u)));
        // (one loop for the masked pairs and for those beyond the mask: one copy of the body)
        uint32_t pair = todo ? (uint32_t)__builtin_ctzll(todo) : 64u;
        while (pair < n_pairs) {
            test_pair(pair, pair < 32u ? (uint32_t)(sv.rect_about_y >> (2u * pair)) : 0u,
                      pair < 32u ? (uint32_t)(sv.rect_about_x >> (2u * pair)) : 0u, std::true_type());
            if (pair < 64u) {
                todo &= todo - 1ull;
                pair = todo ? (uint32_t)__builtin_ctzll(todo) : 64u;
            } else {
                ++pair;
            }
        }
    } else {
        st.count(CT_RECT_PAIR_RAYS, RT_LANES(active) * n_pairs);
        // the class masks move along with the loop, two bits a pair; past slot 63 they have run empty
        unsigned long long about_y = sv.rect_about_y, about_x = sv.rect_about_x;
        for (uint32_t pair = 0; pair < n_pairs; ++pair, about_y >>= 2, about_x >>= 2)
            test_pair(pair, (uint32_t)about_y, (uint32_t)about_x, std::false_type());
    }
    st.mark(ST_RECTS);
    Closest c;
    c.t = best_t;
    c.id = best_id;
    return c;
}

struct Hit { V3 normal; uint32_t material; };

// Normal and material of the winning primitive, as the reference computes them
// at the moment it accepts the hit.
template <int IMODE, int PRIMS>
__device__ __forceinline__ Hit resolve_hit(const SceneView& sv, uint32_t best_id, float t, V3 o, V3 d) {
    Hit h;
    uint32_t id = best_id - 1;
    if (id < sv.n_planes) {
        const float* pl = sv.f + sv.off_planes + PLANE_WORDS * id;
        h.normal = load3(pl);
        h.material = sv.u[sv.off_planes + PLANE_WORDS * id + 4];
    } else if ((PRIMS & PRIMS_SPHERES) && (!(PRIMS & PRIMS_RECTS) || id < sv.n_planes + sv.n_spheres)) {
        uint32_t k = id - sv.n_planes;
        const float4 sp = (PRIMS & PRIMS_FEW) ? sv.wave_spheres[k] : sv.gspheres[k];
        V3 c = mk(sp.x, sp.y, sp.z);
        if (IMODE == RT_INTERSECT_WIDE8) {
            // main.cpp:86-87: FMulAdd(d, t, o) - c, scaled by the reciprocal square root of its
            // fma-chain length.  The reference's factor is the x86 rsqrt approximation
            // (simd_avx2.h:139-144, relative error <= 1.5 * 2^-12, hardware specific); this is the
            // correctly rounded 1 / sqrt.
            V3 v = mk(__builtin_fmaf(d.x, t, o.x), __builtin_fmaf(d.y, t, o.y), __builtin_fmaf(d.z, t, o.z)) - c;
            float f = inv_sqrt_exact(dot_fma(v, v));
            h.normal = mk(v.x * f, v.y * f, v.z * f);
        } else {
            V3 hp = o + d * t;                 // main.cpp:203-204
            h.normal = normalize(hp - c);
        }
        h.material = ((PRIMS & PRIMS_FEW) ? reinterpret_cast<const uint32_t*>(sv.wave_spheres + 64)[k] : sv.gtags[k]) & TAG_MATERIAL_MASK;
    } else {
        uint32_t k = id - sv.n_planes - sv.n_spheres;
        const float4 tail = *reinterpret_cast<const float4*>(sv.f + sv.off_rects + RECT_PAIR_WORDS * (k >> 1) + 24 + 4 * (k & 1));
        V3 n = mk(tail.x, tail.y, tail.z);
        float dp = IMODE == RT_INTERSECT_WIDE8 ? dot_fma(n, d) : dot(n, d);
        h.normal = dp > 0.0f ? -n : n;          // two-sided (main.cpp:118-126, 228-236)
        h.material = __float_as_uint(tail.w) & TAG_MATERIAL_MASK;
    }
    return h;
}

// math_util.h:52-79.  inv_ri is the narrowed double quotient 1.0 / ri.
__device__ __forceinline__ bool refract(V3 incident, V3 normal, float ri, float inv_ri, V3& out) {
    float cos_i = clampf(-1.0f, dot(incident, normal), 1.0f);
    V3 hit_normal = normal;
    float ratio;
    if (cos_i < 0) {
        cos_i = -cos_i;
        ratio = inv_ri;
    } else {
        hit_normal = -normal;
        ratio = ri;
    }
    float discriminant = 1 - ratio * ratio * (1 - cos_i * cos_i);
    if (discriminant < 0) return false;
    out = incident * ratio + hit_normal * (ratio * cos_i - sqrt_exact(discriminant));
    return true;
}

// math_util.h:81-93.  powf(x, 5) is evaluated as a double product and rounded once.
__device__ __forceinline__ float schlick(V3 incident, V3 normal, float r0) {
    float cos_i = clampf(-1.0f, dot(incident, normal), 1.0f);
    float cosine = cos_i > 0 ? cos_i : -cos_i;
    double x = (double)(1 - cosine);
    double x2 = x * x;
    float p5 = (float)(x2 * x2 * x);
    return r0 + (1 - r0) * p5;
}

__device__ __forceinline__ void emit_pixel(const KernelParams& p, V3 sum, size_t index,
                                           uint32_t* pixels, float* radiance) {
    float n = (float)p.spp;
    V3 mean = mk(sum.x / n, sum.y / n, sum.z / n);   // color / sampleSize (main.cpp:393)
    if (pixels) pixels[index] = pack_srgb(mean);
    if (radiance) {
        V3 out = p.output_sum ? sum : mean;
        if (p.accumulate)   // progressive rendering: running sum + this pass, the order in which a host would add them
            out = mk(radiance[3 * index + 0], radiance[3 * index + 1], radiance[3 * index + 2]) + sum;
        radiance[3 * index + 0] = out.x;
        radiance[3 * index + 1] = out.y;
        radiance[3 * index + 2] = out.z;
    }
}

// Workgroup size is chosen per launch: 256 threads when the scene image is small, up to 1024
// when it is large, so that one LDS copy of the scene still serves 4 waves per SIMD.
constexpr int MAX_BLOCK_THREADS = 1024;
constexpr int BLOCK_THREADS = 256;   // resolve_kernel

// STAGE 0: the whole render in one launch (each wave finishes its streams and writes its pixels).
// STAGE 1 and 2 (PIXEL streams only) split it to compact the tails of the waves: towards the end of a wave
// most lanes have finished their samples and idle while a few streams go on.  A STAGE 1 wave therefore
// stops once at most `pool_threshold` lanes still have work and each of them has at most `pool_max_samples`
// samples left to start: finished lanes store their stream's sum in `lane_sums`, unfinished ones deposit
// their complete state (stream, sample count, partial sum, pending ray) in the pool.  STAGE 2 runs
// persistent waves whose lanes take deposited streams from the pool, one after another, and run them to
// the end with the same loop.  pixel_combine_kernel then adds the lanes_per_pixel sums of every pixel with a deposited stream in
// the order of the STAGE 0 tree and writes the pixel.  A stream is computed by the same operations in the
// same order wherever it runs, so the results are those of STAGE 0 bit for bit.
template <int IMODE, int RMODE, int PRIMS, int STAGE>
__global__ __launch_bounds__(MAX_BLOCK_THREADS) void render_kernel(KernelParams p, const uint4* __restrict__ scene_image,
                                                               uint32_t* __restrict__ pixels,
                                                               float* __restrict__ radiance,
                                                               unsigned long long* __restrict__ rays,
                                                               Workspace ws) {
    extern __shared__ uint4 lds_image[];
    for (uint32_t i = threadIdx.x; i < p.lds_words / 4; i += blockDim.x) lds_image[i] = scene_image[i];
    __syncthreads();

    SceneView sv;
    sv.f = reinterpret_cast<const float*>(lds_image);
    sv.u = reinterpret_cast<const uint32_t*>(lds_image);
    sv.n_planes = sv.u[H_NPLANES]; sv.n_spheres = p.sphere_count; sv.n_rects = sv.u[H_NRECTS];
    sv.gspheres = reinterpret_cast<const float4*>(scene_image) + p.sphere_offset / 4;
    sv.wave_spheres = reinterpret_cast<float4*>(lds_image + p.lds_words / 4) + (SPHERE_SLICE_BYTES / 16u) * (threadIdx.x >> 6);
    sv.gtags = reinterpret_cast<const uint32_t*>(scene_image) + p.sphere_tag_offset;
    sv.off_planes = sv.u[H_OFF_PLANES];
    sv.off_rects = sv.u[H_OFF_RECTS]; sv.off_mats = sv.u[H_OFF_MATS];
    sv.rect_about_y = p.rect_about_y; sv.rect_about_x = p.rect_about_x;
    if ((PRIMS & PRIMS_FEW) && sv.n_spheres) {
        // all records and tags into the wave's slice, once; LDS operations of a wave execute in order
        const uint32_t lane = threadIdx.x & 63u, k = min(lane, sv.n_spheres - 1u);
        sv.wave_spheres[lane] = sv.gspheres[k];
        reinterpret_cast<uint32_t*>(sv.wave_spheres + 64)[lane] = sv.gtags[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    const V3 cam_pos = load3(p.cam_pos), cam_x = load3(p.cam_x), cam_y = load3(p.cam_y);
    const V3 film_center = load3(p.film_center);

    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;

    // ---- this lane's stream and its list of samples --------------------------------
    uint32_t state = 1;
    uint32_t x = 0, ly = 0, ly_end = 0;   // current pixel (local row) and end of the lane's rows
    uint32_t samples_left = 0;            // samples still to START in the current pixel
    uint32_t sid = 0;                     // PIXEL: index of the lane's stream, pixel * lanes_per_pixel + batch
    bool has_pixel = false;               // the lane still owns an unfinished pixel
    bool pixel_valid = false;             // PIXEL: lane maps to a pixel inside the image
    unsigned long long primary_pairs = ~0ull;   // rectangle pairs that iterations of primary rays only have to visit

    if (RMODE == RT_RNG_ROW_SERIAL) {
        if (gid < p.order_count) {
            ly = gid * p.rows_per_order;
            ly_end = min(ly + p.rows_per_order, p.local_rows);
            uint32_t start_row = global_row(p, ly);
            uint32_t end_row = global_row(p, ly_end - 1) + 1;
            state = p.seed_constant * (start_row + end_row * 36u);   // main.cpp:370
            has_pixel = true;
            samples_left = p.spp;
        }
    } else if (STAGE == 2) {
        // the stream comes from the pool: restored after the common initialisation below
    } else {
        const uint32_t wave = gid >> 6, lane = gid & 63u;
        const uint32_t q = lane / p.lanes_per_pixel, b = lane % p.lanes_per_pixel;
        // Tile order: column by column, starting with the middle column and moving outwards.
        const uint32_t column_rank = wave / p.tiles_y, tile_y = wave % p.tiles_y;
        const uint32_t middle = p.tiles_x / 2u;
        const uint32_t tile_x = (column_rank & 1u) ? middle - (column_rank + 1u) / 2u : middle + column_rank / 2u;
        x = tile_x * p.tile_w + q % p.tile_w;
        ly = tile_y * p.tile_h + q / p.tile_w;
        pixel_valid = (column_rank < p.tiles_x) && (x < p.width) && (ly < p.local_rows);
        if (pixel_valid) {
            state = pixel_stream_seed(p.seed_constant, x, global_row(p, ly), p.batch_offset + b);
            has_pixel = true;
            samples_left = p.samples_per_lane;
            sid = (ly * p.width + x) * p.lanes_per_pixel + b;
        }
        // The wave's tile is fixed for its lifetime: decide once which rectangle pairs its primary rays can reach.
        if ((PRIMS & PRIMS_RECTS) && p.cull_margin > 0.0f && column_rank < p.tiles_x) {
            const uint32_t x0 = tile_x * p.tile_w, ly0 = tile_y * p.tile_h;
            primary_pairs = primary_pair_mask(reinterpret_cast<const double*>(reinterpret_cast<const uint32_t*>(scene_image) + sv.u[H_OFF_CULL]), sv, p, x0, min(x0 + p.tile_w, p.width) - 1u, global_row(p, ly0),
                                              global_row(p, min(ly0 + p.tile_h, p.local_rows) - 1u));
        }
    }
    // The mask is the same in every lane (a ballot), but it was assigned under conditions derived from the thread
    // index, which the compiler cannot see to be wave-uniform: say so, or the loop over its bits is compiled as a
    // per-lane loop in vector registers.
    primary_pairs = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(primary_pairs >> 32)) << 32) |
                    __builtin_amdgcn_readfirstlane((uint32_t)primary_pairs);
    const bool nothing_for_primary_rays = (PRIMS & ~PRIMS_CLASSED) == PRIMS_RECTS && sv.n_planes == 0 && sv.n_rects <= 128u &&
                                          (primary_pairs & (sv.n_rects >= 128u ? ~0ull : (1ull << (sv.n_rects / 2u)) - 1ull)) == 0ull;
    // STAGE 2: persistent waves; every lane fetches deposited streams from the pool until it is empty.
    const uint32_t pool_count = STAGE == 2 ? min(ws.pool_count[0], p.pool_capacity) : 0u;
    float film_x = 0.0f, film_y = 0.0f;
    if (has_pixel) {
        film_y = ((float)(int32_t)global_row(p, ly) / (float)(int32_t)p.height) * -2.0f + 1.0f;   // main.cpp:375
        film_x = (((float)(int32_t)x / (float)(int32_t)p.width) * 2.0f - 1.0f);                   // main.cpp:377
    }

    V3 pixel_sum = mk(0.0f, 0.0f, 0.0f);
    V3 ro = cam_pos, rd = mk(0.0f, 0.0f, 1.0f), attenuation = mk(1.0f, 1.0f, 1.0f), result = mk(0.0f, 0.0f, 0.0f);
    V3 next_dir = mk(0.0f, 0.0f, 1.0f);   // direction of the lane's next ray, before normalisation
    uint32_t bounce = 0;
    bool in_path = false;                 // the lane has a ray to intersect in this iteration
    bool need_sample = has_pixel;         // the lane's previous path ended (or it has not started one)
    unsigned long long wave_rays = 0;     // rays traced by the wave: the lanes with a ray, added up every iteration (a scalar)
    if (STAGE == 2) need_sample = true;   // every lane starts by fetching a stream

    // STAGE 2: the pool is dealt to the waves in groups of 64 entries, group g to wave g mod (number of
    // waves); a wave walks through its entries with a cursor that all its lanes advance together.
    uint32_t wave_cursor = 0;
    Stamps st;
    st.begin();
    while (true) {
        uint32_t fetch_slot = 0xFFFFFFFFu;
        if (STAGE == 2) {
            const unsigned long long asking = ballot(need_sample && samples_left == 0);
            const uint32_t mine = wave_cursor + (uint32_t)__builtin_popcountll(asking & ((1ull << (threadIdx.x & 63u)) - 1ull));
            const uint32_t waves = (gridDim.x * blockDim.x) >> 6;
            fetch_slot = ((gid >> 6) + (mine >> 6) * waves) * 64u + (mine & 63u);
            wave_cursor += (uint32_t)__builtin_popcountll(asking);
        }
        // ---- lanes between paths: close the pixel if it is complete, start the next sample
        //      (main.cpp:379-388).  Only the un-normalised primary direction is formed here; the
        //      normalisation below is shared with the lanes that continue a path.
        st.count(CT_START_LANES, RT_LANES(need_sample));
        if (need_sample) {
            need_sample = false;
            if (samples_left == 0) {
                if (RMODE == RT_RNG_ROW_SERIAL) {
                    emit_pixel(p, pixel_sum, (size_t)ly * p.width + x, pixels, radiance);
                    pixel_sum = mk(0.0f, 0.0f, 0.0f);
                    if (++x == p.width) {
                        x = 0;
                        ++ly;
                        if (ly < ly_end)
                            film_y = ((float)(int32_t)global_row(p, ly) / (float)(int32_t)p.height) * -2.0f + 1.0f;
                    }
                    if (ly < ly_end) {
                        film_x = (((float)(int32_t)x / (float)(int32_t)p.width) * 2.0f - 1.0f);
                        samples_left = p.spp;
                    } else {
                        has_pixel = false;
                    }
                } else if (STAGE == 2) {
                    // The lane's stream is complete: store its sum, then take the next deposited stream.
                    if (has_pixel) ws.lane_sums[sid] = make_float4(pixel_sum.x, pixel_sum.y, pixel_sum.z, 0.0f);
                    has_pixel = false;
                    const uint32_t slot = fetch_slot;
                    if (slot < pool_count) {
                        const uint4* e = ws.pool + (size_t)slot * POOL_VEC4;
                        const uint4 e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3], e4 = e[4];
                        sid = e0.x; state = e0.y; samples_left = e0.z; bounce = e0.w;
                        pixel_sum = mk(__uint_as_float(e1.x), __uint_as_float(e1.y), __uint_as_float(e1.z));
                        ro = mk(__uint_as_float(e1.w), __uint_as_float(e2.x), __uint_as_float(e2.y));
                        next_dir = mk(__uint_as_float(e2.z), __uint_as_float(e2.w), __uint_as_float(e3.x));
                        attenuation = mk(__uint_as_float(e3.y), __uint_as_float(e3.z), __uint_as_float(e3.w));
                        result = mk(__uint_as_float(e4.x), __uint_as_float(e4.y), __uint_as_float(e4.z));
                        const uint32_t pixel = sid / p.lanes_per_pixel;
                        x = pixel % p.width; ly = pixel / p.width;
                        film_y = ((float)(int32_t)global_row(p, ly) / (float)(int32_t)p.height) * -2.0f + 1.0f;
                        film_x = (((float)(int32_t)x / (float)(int32_t)p.width) * 2.0f - 1.0f);
                        in_path = true;     // the deposited stream continues with its pending ray
                    }
                } else {
                    has_pixel = false;   // the pixel is written after the loop (cross-lane sum)
                }
            }
            if (STAGE == 2 && in_path) has_pixel = true;   // fetched: no new sample in this iteration
            else if (has_pixel) {
                --samples_left;
                float off_x = film_x + random_bilateral(state) * p.pixel_w;
                float off_y = film_y + random_bilateral(state) * p.pixel_h;
                V3 film_pos = film_center + cam_x * off_x * p.half_w + cam_y * p.half_h * off_y;
                ro = cam_pos;
                next_dir = film_pos - cam_pos;
                attenuation = mk(1.0f, 1.0f, 1.0f);
                result = mk(0.0f, 0.0f, 0.0f);
                bounce = 0;
                in_path = true;
            }
        }
        st.mark(ST_START);
        const unsigned long long active = ballot(in_path);
        if (active == 0ull) break;   // no lane of the wave has a ray left
        if (STAGE == 1 && (uint32_t)__builtin_popcountll(active) <= p.pool_threshold &&
            ballot(in_path && samples_left > p.pool_max_samples) == 0ull)
            break;   // hand the tail over: few lanes left, each with little left to do

        wave_rays += (uint32_t)__builtin_popcountll(active);   // every lane with a ray traces it in this iteration (main.cpp:278)

        // ---- one ray per lane: a primary ray or a bounce (main.cpp:275-330) ------------------
        // Worlds with spheres: every lane of the wave enters the intersection (the lanes stage the sphere
        // records together), lanes without a ray only help with that.
        // All rays of the wave are primary rays?  (Not asked in STAGE 2, whose waves hold streams of many tiles in
        // the middle of their paths: without the all-primary copy of the rectangle loop that kernel keeps a wave
        // per SIMD more; computing the origin rows gives the bits the stored ones have.)
        const bool camera_origin = STAGE != 2 && ballot(in_path && bounce != 0) == 0ull;
        Closest c;
        st.count(CT_ITERATIONS, 1);
        st.count(CT_LANES_TRACING, RT_LANES(in_path));
        if (PRIMS & PRIMS_SPHERES) {
            if (in_path) rd = normalize(next_dir);
            st.mark(ST_CONTROL);
            c = intersect_world<IMODE, PRIMS>(sv, ro, rd, p.min_hit, p.plane_tol, camera_origin, primary_pairs, in_path, st);
        }
        if (in_path) {
            if (!(PRIMS & PRIMS_SPHERES)) {
                if ((PRIMS & ~PRIMS_CLASSED) == PRIMS_RECTS && camera_origin && nothing_for_primary_rays) {
                    // Primary rays of a tile from which no rectangle can be reached (and a world without planes): a
                    // miss, whatever the direction; the miss branch below does not read it, so it is not normalised.
                    c.t = 3.402823466e+38f;
                    c.id = 0;
                } else {
                    rd = normalize(next_dir);
                    st.mark(ST_CONTROL);
                    c = intersect_world<IMODE, PRIMS>(sv, ro, rd, p.min_hit, p.plane_tol, camera_origin, primary_pairs, true, st);
                }
            }
            ++bounce;
            st.count(CT_HIT_LANES, RT_LANES(c.id != 0));
            if (c.id != 0) {
                Hit h = resolve_hit<IMODE, PRIMS>(sv, c.id, c.t, ro, rd);
                st.mark(ST_RESOLVE);
                const float4* m = reinterpret_cast<const float4*>(sv.f + sv.off_mats + MAT_WORDS * h.material);
                const float4 m0 = m[0], m1 = m[1], m2 = m[2];   // color ri | emit reflection | 1/ri r0
                result = result + attenuation * mk(m1.x, m1.y, m1.z);
                attenuation = attenuation * mk(m0.x, m0.y, m0.z);
                ro = ro + rd * c.t;

                V3 n = h.normal;
                // g++ evaluates the constructor arguments right to left: z, y, x (main.cpp:291-294).
                float rz = random_bilateral(state);
                float ry = random_bilateral(state);
                float rx = random_bilateral(state);
                // The reference normalises the reflected and the refracted candidate and then picks
                // one (main.cpp:295-322); picking first and normalising once gives the same vector.
                V3 reflected = n + mk(rx, ry, rz);
                const float refl = m1.w;
                if (refl != 0.0f) {
                    V3 mirror = rd - n * dot(n, rd) * 2.0f;
                    reflected = reflected * (1.0f - refl) + mirror * refl;   // Lerp, math_util.h:24-26
                }
                // refl == 0 (diffuse): Lerp gives random_bounce * 1 + mirror * 0 = random_bounce + (+-0), which is
                // random_bounce bit for bit: the sum n + r is never -0 (r is never -0), and mirror is finite
                // (unit rd, scene values bounded at upload), so the block is skipped by waves with no mirror hit.
                const uint32_t pick = xorshift32(state);   // RandomUnilateral of main.cpp:318, drawn for every material
                next_dir = reflected;
                if (m0.w != 0.0f) {
                    st.count(CT_DIELECTRIC_BODIES, 1);
                    st.count(CT_DIELECTRIC_LANES, RT_LANES(true));
                    float fresnel = 1.0f;
                    V3 refracted = reflected;
                    if (refract(rd, n, m0.w, m2.x, refracted)) fresnel = schlick(rd, n, m2.y);
                    if (!((float)pick * 0x1.0p-32f <= fresnel)) next_dir = refracted;
                }
                // Opaque materials: both candidates are `reflected`, whatever the draw says.
                if (bounce >= p.max_bounces) in_path = false;
            } else {
                const float4 sky = *reinterpret_cast<const float4*>(sv.f + sv.off_mats + 4);   // materials[0].emit
                result = result + attenuation * mk(sky.x, sky.y, sky.z);
                in_path = false;
            }
            if (!in_path) {
                pixel_sum = pixel_sum + result;   // color += RaytraceWorld(...) (main.cpp:390)
                need_sample = true;
            }
        }
        st.mark(ST_SHADE);
    }

    // ---- epilogue ------------------------------------------------------------------
    bool pixel_pending = false;           // STAGE 1: some stream of the lane's pixel was handed over
    if (RMODE == RT_RNG_PIXEL && STAGE != 0) {
        // Finished streams leave their sum, unfinished ones (STAGE 1 only) their state.
        const unsigned long long unfinished = ballot(in_path);
        uint32_t base = 0;
        if (unfinished) {
            const uint32_t leader = (uint32_t)__builtin_ctzll(unfinished);
            if ((threadIdx.x & 63u) == leader) base = atomicAdd(ws.pool_count, (uint32_t)__builtin_popcountll(unfinished));
            base = __shfl(base, leader);
        }
        if (in_path) {
            const uint32_t slot = base + (uint32_t)__builtin_popcountll(unfinished & ((1ull << (threadIdx.x & 63u)) - 1ull));
            if (slot < p.pool_capacity) {   // always true: capacity = waves * pool_threshold
                uint4* e = ws.pool + (size_t)slot * POOL_VEC4;
                e[0] = make_uint4(sid, state, samples_left, bounce);
                e[1] = make_uint4(__float_as_uint(pixel_sum.x), __float_as_uint(pixel_sum.y), __float_as_uint(pixel_sum.z), __float_as_uint(ro.x));
                e[2] = make_uint4(__float_as_uint(ro.y), __float_as_uint(ro.z), __float_as_uint(next_dir.x), __float_as_uint(next_dir.y));
                e[3] = make_uint4(__float_as_uint(next_dir.z), __float_as_uint(attenuation.x), __float_as_uint(attenuation.y), __float_as_uint(attenuation.z));
                e[4] = make_uint4(__float_as_uint(result.x), __float_as_uint(result.y), __float_as_uint(result.z), 0u);
            }
        }
        if (STAGE == 1) {
            // A pixel whose lanes_per_pixel streams all ended in this wave is finished here, like in STAGE 0;
            // the others wait for pixel_combine_kernel, which reads the per-stream sums.
            const uint32_t lane = threadIdx.x & 63u, width_mask = p.lanes_per_pixel - 1u;
            const unsigned long long group = (p.lanes_per_pixel == 64u ? ~0ull : ((1ull << p.lanes_per_pixel) - 1ull)) << (lane & ~width_mask);
            pixel_pending = (unfinished & group) != 0ull;
            if (pixel_pending && !in_path && pixel_valid) ws.lane_sums[sid] = make_float4(pixel_sum.x, pixel_sum.y, pixel_sum.z, 0.0f);
            if (pixel_valid && (lane & width_mask) == 0u) ws.pending[(size_t)ly * p.width + x] = pixel_pending ? 1 : 0;
        }
    }
    if (RMODE == RT_RNG_PIXEL && STAGE != 2) {
        // Pairwise tree over the lanes_per_pixel partial sums of each pixel.
        for (uint32_t stride = 1; stride < p.lanes_per_pixel; stride <<= 1) {
            float sx = __shfl_down(pixel_sum.x, stride);
            float sy = __shfl_down(pixel_sum.y, stride);
            float sz = __shfl_down(pixel_sum.z, stride);
            if (((threadIdx.x % p.lanes_per_pixel) & (2 * stride - 1)) == 0) pixel_sum = pixel_sum + mk(sx, sy, sz);
        }
        if (pixel_valid && !pixel_pending && (threadIdx.x % p.lanes_per_pixel) == 0)
            emit_pixel(p, pixel_sum, (size_t)ly * p.width + x, pixels, radiance);
    }
    st.mark(ST_EPILOGUE);
    st.flush();
    // Ray counter: one 64-bit atomic per wave (main.cpp:397).  A wave keeps its slot until its atomic has been
    // performed, and atomics on ONE address are performed one after another: 129,600 waves of a 1080p render ending
    // within 8 ms (128 samples per pixel) queue up behind the counter (measured: 8.7 ms, 8.3 without the atomic).
    // The stages of a compacted render therefore add to one of RAY_SLOTS counters of the workspace, which
    // pixel_combine_kernel sums into `rays`; the single launch, which has no workspace, adds to `rays` itself.
    if ((threadIdx.x & 63) == 0 && wave_rays != 0 && rays)
        atomicAdd(STAGE != 0 ? ws.ray_slots + ((gid >> 6) % RAY_SLOTS) * RAY_SLOT_STRIDE : rays, wave_rays);
}

// Zeroes the ray counter (and the pool fill count) ahead of a render: a kernel rather than a memset
// call, so that a recorded HIP graph replays it faithfully.
__global__ void clear_counter_kernel(unsigned long long* counter, uint32_t* pool_count, unsigned long long* ray_slots) {
    if (threadIdx.x == 0) {
        if (counter) *counter = 0ull;
        if (pool_count) *pool_count = 0u;
    }
    if (ray_slots && threadIdx.x < RAY_SLOTS) ray_slots[threadIdx.x * RAY_SLOT_STRIDE] = 0ull;   // launched with RAY_SLOTS threads
}

// Last step of the compacted render, for the pixels STAGE 1 left pending: the pixel is the pairwise tree over its lanes_per_pixel stream sums,
// in the order of the in-wave tree of STAGE 0 (halves added left + right), then color / sampleSize and pack.
__device__ inline V3 tree_sum(const float4* sums, uint32_t n) {   // n: power of two <= 64
    V3 partial[7];   // partial[l]: sum of a completed left block of 2^l streams waiting for its right neighbour
    V3 v = mk(0.0f, 0.0f, 0.0f);
    for (uint32_t i = 0; i < n; ++i) {
        v = mk(sums[i].x, sums[i].y, sums[i].z);
        uint32_t level = 0;
        for (; (i >> level) & 1u; ++level) v = partial[level] + v;
        partial[level] = v;
    }
    return v;
}
__global__ __launch_bounds__(BLOCK_THREADS) void pixel_combine_kernel(KernelParams p, const float4* __restrict__ lane_sums,
                                                                      const uint8_t* __restrict__ pending,
                                                                      uint32_t* __restrict__ pixels, float* __restrict__ radiance,
                                                                      const unsigned long long* __restrict__ ray_slots,
                                                                      unsigned long long* __restrict__ rays) {
    if (rays && blockIdx.x == 0 && threadIdx.x == 0) {   // the render's ray count: the sum of the stages' partial counters
        unsigned long long total = 0;
        for (uint32_t k = 0; k < RAY_SLOTS; ++k) total += ray_slots[k * RAY_SLOT_STRIDE];
        *rays = total;
    }
    const size_t i = (size_t)blockIdx.x * BLOCK_THREADS + threadIdx.x;
    if (i >= (size_t)p.local_rows * p.width || !pending[i]) return;
    emit_pixel(p, tree_sum(lane_sums + i * p.lanes_per_pixel, p.lanes_per_pixel), i, pixels, radiance);
}

// Sum of `parts` radiance sums (one per GPU or per pass, `stride` floats apart) in the order of tree_sum:
// neighbours first, then neighbouring pairs, left operand first; a count that is not a power of two
// carries its odd block up unchanged.  For a power-of-two count this continues the per-pixel tree of the
// render kernel, so N GPUs with B streams per pixel each give the sums of one GPU with N*B streams.
__global__ __launch_bounds__(BLOCK_THREADS) void combine_kernel(const float* __restrict__ sums, uint32_t parts, uint64_t stride,
                                                                uint64_t count, float* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK_THREADS + threadIdx.x;
    if (i >= count) return;
    float partial[7];    // partial[l]: a completed block of 2^l parts waiting for its right neighbour (parts <= 64)
    uint32_t occupied = 0;
    float v = 0.0f;
    for (uint32_t k = 0; k < parts; ++k) {
        v = sums[(uint64_t)k * stride + i];
        uint32_t level = 0;
        for (; (k >> level) & 1u; ++level) v = partial[level] + v;
        partial[level] = v;
        occupied = (occupied & ~((1u << level) - 1u)) | (1u << level);
    }
    // leftover blocks of a count that is not a power of two: the smallest (rightmost) joins its left neighbour first
    bool first = true;
    for (uint32_t level = 0; level < 7; ++level) {
        if (!((occupied >> level) & 1u)) continue;
        v = first ? partial[level] : partial[level] + v;
        first = false;
    }
    out[i] = v;
}

// Tonemap + pack of an accumulated sum (epilogue of main.cpp:393 on its own).
__global__ __launch_bounds__(BLOCK_THREADS) void resolve_kernel(const float* __restrict__ sum, uint64_t pixel_count,
                                                                float total_samples, uint32_t* __restrict__ pixels,
                                                                float* __restrict__ mean_out) {
    uint64_t i = (uint64_t)blockIdx.x * BLOCK_THREADS + threadIdx.x;
    if (i >= pixel_count) return;
    V3 mean = mk(sum[3 * i] / total_samples, sum[3 * i + 1] / total_samples, sum[3 * i + 2] / total_samples);
    if (pixels) pixels[i] = pack_srgb(mean);
    if (mean_out) { mean_out[3 * i] = mean.x; mean_out[3 * i + 1] = mean.y; mean_out[3 * i + 2] = mean.z; }
}

}  // namespace rt
