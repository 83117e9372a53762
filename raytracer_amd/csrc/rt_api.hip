// C ABI of include/rt_hip.h: context, world upload, render launches.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (raytracer_amd/csrc/Makefile).
// -ffp-contract=off applies to the host code here too: the film constants below
// are computed with the same once-rounded float operations as main.cpp:350-368.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_kernels.hpp"
#include "rt_layout.hpp"

// Workspace of rt_render_device_noalloc: one fixed array per GPU, part of the code object (each GPU the
// library is loaded on holds its own copy until the process ends).  Host code never uses the array's
// name as an address: rt_create asks the runtime for this device's copy (hipGetSymbolAddress) and the
// context keeps that pointer.
constexpr size_t STATIC_SCRATCH_BYTES = 32u << 20;
// head of a compaction workspace: the pool's fill count, then (256 bytes on) the partial ray counters
constexpr size_t COUNTER_BLOCK_BYTES = 256 + rt::RAY_SLOTS * rt::RAY_SLOT_STRIDE * sizeof(unsigned long long);
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

struct rt_context {
    int device = -1;
    unsigned char* static_scratch = nullptr;   // this device's copy of `scratch`
    // stage2_waves_per_cu's answers (no heap: the no-allocation entry point may be the first to ask)
    struct Residency { const void* kernel; uint32_t block_threads; size_t lds_bytes; uint32_t waves; };
    mutable Residency stage2_residency[32];
    mutable uint32_t stage2_residency_count = 0;
    hipDeviceProp_t props{};
    std::string error;

    // world: one LDS image per intersection mode
    bool has_world = false;
    bool mixed = false;             // spheres and rectangles both present
    rt_camera camera{};
    uint32_t image_words[3] = {0, 0, 0};
    unsigned long long rect_about_y[3] = {0, 0, 0}, rect_about_x[3] = {0, 0, 0};   // the images' sparsity masks (rt_layout.hpp)
    uint32_t lds_words = 0;                 // part of the image each workgroup stages in LDS (same for both images)
    uint32_t sphere_count = 0, sphere_offset = 0, sphere_tag_offset = 0;
    uint4* d_image[3] = {nullptr, nullptr, nullptr};   // indexed by RT_INTERSECT_*

    // buffers behind rt_render (host outputs); grow-only
    uint32_t* d_pixels = nullptr;
    float* d_radiance = nullptr;
    float* d_resolve_sum = nullptr;  // rt_resolve's copy of the caller's host-resident sum
    size_t pixel_capacity = 0, radiance_capacity = 0, resolve_capacity = 0;
    // running sum of RT_FLAG_ACCUMULATE renders through rt_render: d_radiance, valid for the rows of this key
    bool accumulator_valid = false;
    uint32_t accumulator_key[6] = {0, 0, 0, 0, 0, 0};
    void* h_staging = nullptr;       // pinned host buffer the results are copied through
    // workspace of RT_FLAG_COMPACT_TAILS renders; grow-only
    rt::Workspace ws{nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t lane_sum_capacity = 0, pool_entry_capacity = 0, pending_capacity = 0;
    size_t staging_bytes = 0;
    unsigned long long* d_rays = nullptr;

    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timed = false;
};

namespace {

rt_status fail(rt_context* ctx, rt_status st, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->error = buf;
    return st;
}

#define RT_HIP_CHECK(ctx, call)                                                                          \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fail((ctx), e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP, "%s: %s", \
                        #call, hipGetErrorString(e_));                                                   \
    } while (0)

bool is_pow2(uint32_t v) { return v && !(v & (v - 1)); }

// Tail compaction (see render_kernel): a stage-1 wave stops when at most `pool_threshold()` of its 64 lanes
// still have work and none of them has more than `pool_max_samples()` samples left to start.  The second
// bound keeps the handed-over streams short: stage 2 only begins when stage 1 has ended, so a long stream
// would run there on its own (measured on the Cornell bench without it: stage 1 -1.1 ms, stage 2 +1.6 ms).
// Defaults from a sweep on the bench (T 16..48 x S 2..32).
// A lane with few samples hands over at most a quarter of them: the bound is min(12, samples_per_lane / 4 - 1),
// and renders with fewer than 8 samples per lane run as one launch (4096 spheres at 64 spp: 4 lanes per pixel
// with 16 samples each and a bound of 3 take 498 ms, one lane per pixel with the bound of 12 takes 566 ms).
//
// The shipped library reads nothing from the environment.  Only a build with -DRT_EXPERIMENT_KNOBS (make knobs,
// used by the sweep scripts) lets RT_POOL_THRESHOLD, RT_POOL_MAX_SAMPLES, RT_BLOCK_THREADS and
// RT_STAGE2_WAVES_PER_CU override these launch parameters.
uint32_t env_or(const char* name, long fallback, long lo, long hi) {
#ifdef RT_EXPERIMENT_KNOBS
    const char* env = std::getenv(name);
    long v = env ? std::strtol(env, nullptr, 10) : fallback;
#else
    (void)name;
    long v = fallback;
#endif
    return (uint32_t)(v < lo ? lo : v > hi ? hi : v);
}
uint32_t pool_threshold() { static const uint32_t v = env_or("RT_POOL_THRESHOLD", 32, 1, 63); return v; }
constexpr uint32_t MIN_SAMPLES_PER_LANE_TO_COMPACT = 8;
uint32_t pool_max_samples(uint32_t samples_per_lane) {
    static const uint32_t forced = env_or("RT_POOL_MAX_SAMPLES", -1, -1, 1 << 20);
    if (forced != (uint32_t)-1) return forced;
    const uint32_t quarter = samples_per_lane / 4u;
    return quarter ? (quarter - 1u < 12u ? quarter - 1u : 12u) : 0u;
}

struct Launch {
    rt::KernelParams kp;
    uint32_t blocks, block_threads;
    int imode, rmode;
    bool compact = false;            // RT_FLAG_COMPACT_TAILS on per-pixel streams: stages 1 + 2 + combine
};

// A CU has 160 KB of LDS and room for 2048 threads: the workgroup size is picked so that the LDS copies of
// the scene image leave the CU full (small image: 256 threads; > 40 KB: 512; > 80 KB: 1024).
uint32_t workgroup_threads(const rt_context* ctx) {
    const size_t image_bytes = (size_t)ctx->lds_words * 4;
    uint32_t threads = image_bytes > 80u * 1024u ? 1024u : image_bytes > 40u * 1024u ? 512u : 256u;
    // Worlds with spheres: one wave per workgroup.  Their waves last long and unevenly (every ray tests every
    // sphere), and a workgroup keeps its waves' slots until its last wave ends (measured with 256 threads on the
    // 4096-sphere scene: 2.8 of 4 possible waves resident on average; one wave per workgroup: +8 %).
    if (ctx->sphere_count && image_bytes <= 8u * 1024u) threads = 64u;
    // Worlds without spheres whose image is small enough for 32 copies per CU (eight one-wave workgroups per SIMD):
    // one wave per workgroup as well - a wave starts as soon as ITS copy is staged and no slot waits for the last
    // wave of a group (Cornell bench, alternating runs: 15.65 ms with 256 threads, 15.59 with 128, 15.53 with 64).
    if (image_bytes <= 4u * 1024u) threads = 64u;
    if (const uint32_t forced = env_or("RT_BLOCK_THREADS", 0, 0, 1024)) threads = forced & ~63u ? forced & ~63u : 64u;   // experiments
    return threads;
}

// Film constants of a render (main.cpp:350-368), and the margins of the primary-ray cull.
void film_and_cull_constants(const rt_camera& c, uint32_t width, uint32_t height, bool cull, rt::KernelParams& k) {
    // Film constants (main.cpp:350-368), rounded operation by operation.
    float aspect = (float)(int32_t)width / (float)(int32_t)height;
    float film_distance = 1.0f;
    for (int i = 0; i < 3; ++i) {
        k.cam_pos[i] = c.position[i];
        k.cam_x[i] = c.x_axis[i];
        k.cam_y[i] = c.y_axis[i];
        k.film_center[i] = c.position[i] - c.z_axis[i] * film_distance;
    }
    float film_w = 1.0f * aspect, film_h = 1.0f;
    k.half_w = film_w * 0.5f;
    k.half_h = film_h * 0.5f;
    k.pixel_w = 0.5f / (float)(int32_t)width;
    k.pixel_h = 0.5f / (float)(int32_t)height;

    // Primary-ray cull (rt_kernels.hpp, primary_pair_mask): a wave decides once which rectangles the primary rays
    // of its pixel tile can reach, from the tile's patch of the film widened by this margin on every side, in
    // film-plane units (one pixel is 1 / height wide in them).  The margin is one pixel plus a bound on how far
    // the kernel's rounded ray set-up (film position, direction, normalisation: about a dozen operations on
    // values of the camera position's magnitude) can move a ray on the film plane.  That bound reads the ray's
    // error through the camera axes, so the cull is used only with axes that are orthonormal to 1e-3.
    k.cull_margin = 0.0f;
    if (cull) {
        const float* axes[3] = {c.x_axis, c.y_axis, c.z_axis};
        bool orthonormal = true;
        double reach = 4.0;
        for (int i = 0; i < 3; ++i) {
            reach = std::fmax(reach, 4.0 + std::fabs((double)c.position[i]));
            for (int j = i; j < 3; ++j) {
                const double g = (double)axes[i][0] * axes[j][0] + (double)axes[i][1] * axes[j][1] + (double)axes[i][2] * axes[j][2];
                orthonormal = orthonormal && std::fabs(g - (i == j ? 1.0 : 0.0)) <= 1.0e-3;
            }
        }
        if (orthonormal) {
            k.cull_margin = (float)(1.0 / (double)height + 0x1.0p-20 * reach);
            // no primary direction D0 + a camX + b camY is longer than this (|a| <= half_w (1 + pixel_w) + margin, same for b)
            k.cull_reach = (float)(1.002 * (1.0 + ((double)k.half_w * (1.0 + k.pixel_w) + k.cull_margin) +
                                            ((double)k.half_h * (1.0 + k.pixel_h) + k.cull_margin)));
        }
    }
}

// `max_pool_entries` (0 = no limit) caps the pool of handed-over streams: the hand-over threshold is lowered
// until waves * threshold entries fit, and with no room at all the render is planned as a single launch.
rt_status validate_and_plan(rt_context* ctx, const rt_render_params* p, Launch* out, uint64_t max_pool_entries = 0) {
    if (!p) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "params is NULL");
    if (p->struct_size != sizeof(rt_render_params))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_render_params.struct_size %u != %zu (ABI mismatch)", p->struct_size,
                    sizeof(rt_render_params));
    if (p->width == 0 || p->height == 0 || p->width > (1u << 16) || p->height > (1u << 16))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "image size %ux%u out of range (1..65536)", p->width, p->height);
    if (p->samples_per_pixel == 0 || p->samples_per_pixel > (1u << 20))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "samples_per_pixel %u out of range (1..1048576)", p->samples_per_pixel);
    if (p->max_bounces == 0 || p->max_bounces > 1024)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "max_bounces %u out of range (1..1024)", p->max_bounces);
    if (p->row_begin >= p->row_end || p->row_end > p->height)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rows [%u,%u) not inside the image height %u", p->row_begin, p->row_end,
                    p->height);
    if (p->band_rows && (p->band_count == 0 || p->band_index >= p->band_count))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "band_index %u / band_count %u invalid", p->band_index, p->band_count);
    if (p->rng_mode != RT_RNG_ROW_SERIAL && p->rng_mode != RT_RNG_PIXEL)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "unknown rng_mode %u", p->rng_mode);
    if (p->intersect_mode != RT_INTERSECT_WIDE8 && p->intersect_mode != RT_INTERSECT_SCALAR && p->intersect_mode != RT_INTERSECT_WIDE4)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "unknown intersect_mode %u", p->intersect_mode);
    if (!(p->min_hit_distance >= 0.0f) || !(p->plane_tolerance >= 0.0f))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "min_hit_distance and plane_tolerance must be >= 0");
    if (p->seed_constant == 0 && p->rng_mode == RT_RNG_ROW_SERIAL)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "seed_constant 0 gives the all-zero xorshift state");

    rt::KernelParams& k = out->kp;
    std::memset(&k, 0, sizeof k);
    k.width = p->width; k.height = p->height; k.spp = p->samples_per_pixel;
    k.local_rows = rt_local_row_count(p);
    if (k.local_rows == 0) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "this band owns no rows");
    k.row_begin = p->row_begin;
    k.band_rows = p->band_rows; k.band_index = p->band_index; k.band_count = p->band_count ? p->band_count : 1;
    k.max_bounces = p->max_bounces;
    k.min_hit = p->min_hit_distance; k.plane_tol = p->plane_tolerance;
    k.seed_constant = p->seed_constant;
    k.accumulate = (p->flags & RT_FLAG_ACCUMULATE) ? 1u : 0u;
    k.output_sum = (p->flags & (RT_FLAG_OUTPUT_SUM | RT_FLAG_ACCUMULATE)) ? 1u : 0u;
    out->imode = (int)p->intersect_mode;
    out->rmode = (int)p->rng_mode;
    k.lds_words = ctx->lds_words;
    k.sphere_count = ctx->sphere_count;
    k.sphere_offset = ctx->sphere_offset;
    k.sphere_tag_offset = ctx->sphere_tag_offset;
    k.rect_about_y = (p->flags & RT_FLAG_DENSE_TRANSFORMS) ? 0ull : ctx->rect_about_y[out->imode];
    k.rect_about_x = (p->flags & RT_FLAG_DENSE_TRANSFORMS) ? 0ull : ctx->rect_about_x[out->imode];

    uint64_t lanes;
    if (p->rng_mode == RT_RNG_ROW_SERIAL) {
        k.rows_per_order = p->rows_per_order ? p->rows_per_order : k.local_rows;
        if (p->band_rows && k.band_count > 1 && (p->band_rows % k.rows_per_order))
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "band_rows %u must be a multiple of rows_per_order %u", p->band_rows,
                        k.rows_per_order);
        k.order_count = (k.local_rows + k.rows_per_order - 1) / k.rows_per_order;
        k.lanes_per_pixel = 1; k.samples_per_lane = k.spp;
        k.tile_w = k.tile_h = 1; k.tiles_x = k.tiles_y = 1;
        lanes = k.order_count;
    } else {
        uint32_t B = p->lanes_per_pixel;
        if (!is_pow2(B) || B > 64 || (p->samples_per_pixel % B))
            return fail(ctx, RT_ERR_INVALID_ARGUMENT,
                        "lanes_per_pixel %u must be a power of two <= 64 dividing samples_per_pixel %u", B,
                        p->samples_per_pixel);
        k.lanes_per_pixel = B;
        k.samples_per_lane = p->samples_per_pixel / B;
        k.batch_offset = p->batch_offset;
        // 64/B pixels per wave, as square a tile as the power of two allows.
        uint32_t pixels_per_wave = 64 / B, tw = 1;
        while (tw * tw < pixels_per_wave) tw <<= 1;
        k.tile_w = tw; k.tile_h = pixels_per_wave / tw;
        k.tiles_x = (k.width + k.tile_w - 1) / k.tile_w;
        k.tiles_y = (k.local_rows + k.tile_h - 1) / k.tile_h;
        k.rows_per_order = 1; k.order_count = 0;
        lanes = (uint64_t)k.tiles_x * k.tiles_y * 64;
        // worth it only when a lane's samples are many compared with what may be handed over
        out->compact = (p->flags & RT_FLAG_COMPACT_TAILS) != 0 && k.samples_per_lane >= MIN_SAMPLES_PER_LANE_TO_COMPACT &&
                       k.samples_per_lane >= 4 * (pool_max_samples(k.samples_per_lane) + 1);
    }
    out->block_threads = workgroup_threads(ctx);
    uint64_t blocks = (lanes + out->block_threads - 1) / out->block_threads;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "launch grid out of range");
    // The kernels index lanes and streams (pixel * lanes_per_pixel + batch) with 32-bit integers.
    if (blocks * out->block_threads > 0xFFFFFFFFull || (uint64_t)k.local_rows * k.width * k.lanes_per_pixel > 0xFFFFFFFFull)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rows x width x lanes_per_pixel must stay below 2^32");
    out->blocks = (uint32_t)blocks;
    if (out->compact) {
        // A stage-1 wave hands over at most pool_threshold streams, so waves * threshold entries always suffice.
        k.pool_threshold = pool_threshold();
        k.pool_max_samples = pool_max_samples(k.samples_per_lane);
        const uint64_t waves = blocks * (out->block_threads / 64u);
        if (max_pool_entries && waves * k.pool_threshold > max_pool_entries) k.pool_threshold = (uint32_t)(max_pool_entries / waves);
        if (k.pool_threshold == 0) out->compact = false;
    }
    if (out->compact) {
        const uint64_t capacity = blocks * (out->block_threads / 64u) * k.pool_threshold;
        if (capacity > 0x7FFFFFFFull) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "image too large for RT_FLAG_COMPACT_TAILS");
        k.pool_capacity = (uint32_t)capacity;
    }

    film_and_cull_constants(ctx->camera, p->width, p->height, p->rng_mode == RT_RNG_PIXEL && !(p->flags & RT_FLAG_NO_PRIMARY_CULL), k);
    return RT_OK;
}

// Persistent waves of stage 2 per CU: as many as the kernel variant keeps resident (it depends on the variant's
// registers: 32 for rectangles only, 28 / 20 with few / many spheres), but no more than 24 - beyond that the
// Cornell bench gets slower (sweep 8 / 16 / 24 / 32: 19.83 / 19.68 / 19.65 / 19.75 ms), below the residency the
// sphere scenes do (4096 spheres at 16 / 20: 493 / 489 ms).  Queried once per (kernel, workgroup size, LDS size).
uint32_t stage2_waves_per_cu(const rt_context* ctx, const void* kernel, uint32_t block_threads, size_t lds_bytes) {
    static const uint32_t forced = env_or("RT_STAGE2_WAVES_PER_CU", 0, 0, 64);
    if (forced) return forced;
    for (uint32_t i = 0; i < ctx->stage2_residency_count; ++i) {
        const rt_context::Residency& r = ctx->stage2_residency[i];
        if (r.kernel == kernel && r.block_threads == block_threads && r.lds_bytes == lds_bytes) return r.waves;
    }
    int blocks_per_cu = 0;
    uint32_t waves = 16;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kernel, (int)block_threads, lds_bytes) == hipSuccess && blocks_per_cu > 0)
        waves = (uint32_t)blocks_per_cu * (block_threads / 64u);
    waves = waves > 24u ? 24u : waves < 4u ? 4u : waves;
    if (ctx->stage2_residency_count == 32) ctx->stage2_residency_count = 0;   // a new world every time: start over
    ctx->stage2_residency[ctx->stage2_residency_count++] = {kernel, block_threads, lds_bytes, waves};
    return waves;
}

template <int IMODE, int RMODE, int PRIMS, int STAGE>
hipError_t launch_variant(const rt_context* ctx, const Launch& L, size_t lds_bytes, const uint4* image, uint32_t* d_pixels,
                          float* d_radiance, unsigned long long* d_rays, const rt::Workspace& ws, hipStream_t stream, bool plan_only) {
    auto kernel = rt::render_kernel<IMODE, RMODE, PRIMS, STAGE>;
    uint32_t blocks = L.blocks;
    if (STAGE == 2) {   // (also called with plan_only by rt_reserve, so that the query below never runs inside a captured launch)
        const uint64_t waves = (uint64_t)ctx->props.multiProcessorCount *
                               stage2_waves_per_cu(ctx, reinterpret_cast<const void*>(kernel), L.block_threads, lds_bytes);
        const uint64_t wanted = (waves * 64u + L.block_threads - 1) / L.block_threads;
        blocks = (uint32_t)(wanted < L.blocks ? wanted : L.blocks);
    }
    if (plan_only) return hipSuccess;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(L.block_threads), lds_bytes, stream, L.kp, image, d_pixels, d_radiance,
                       d_rays, ws);
    return hipGetLastError();
}

// Every instantiation of the render kernel: (intersect mode, stream layout, primitive kinds, stage).  WIDE8,
// the throughput mode, has a variant per primitive mix (rectangles only, with or without rectangles turned about
// one axis / spheres only / at most 64 spheres only / both); SCALAR runs the general one.  Stages 1 and 2 exist for the per-pixel stream layout only.
#define RT_FOR_EACH_STAGE(X, IM, PR)                                                             \
    X(IM, RT_RNG_ROW_SERIAL, PR, 0) X(IM, RT_RNG_PIXEL, PR, 0) X(IM, RT_RNG_PIXEL, PR, 1) X(IM, RT_RNG_PIXEL, PR, 2)
#define RT_FOR_EACH_KERNEL(X)                                                                    \
    RT_FOR_EACH_STAGE(X, RT_INTERSECT_WIDE8, rt::PRIMS_RECTS) RT_FOR_EACH_STAGE(X, RT_INTERSECT_WIDE8, rt::PRIMS_CLASSED_RECTS) \
    RT_FOR_EACH_STAGE(X, RT_INTERSECT_WIDE8, rt::PRIMS_SPHERES)                                  \
    RT_FOR_EACH_STAGE(X, RT_INTERSECT_WIDE8, rt::PRIMS_FEW_SPHERES)                              \
    RT_FOR_EACH_STAGE(X, RT_INTERSECT_WIDE8, rt::PRIMS_BOTH) RT_FOR_EACH_STAGE(X, RT_INTERSECT_SCALAR, rt::PRIMS_BOTH)

// Lets every render kernel variant use up to `bytes` of dynamic LDS on the current device.  Done once
// per context at creation (with the device's maximum), so that a launch is nothing but a launch and can be
// recorded into a HIP graph.
hipError_t allow_dynamic_lds(size_t bytes) {
#define RT_ALLOW(IM, RM, PR, ST)                                                                                             \
    if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rt::render_kernel<IM, RM, PR, ST>),                 \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); e != hipSuccess)         \
        return e;
    RT_FOR_EACH_KERNEL(RT_ALLOW)
#undef RT_ALLOW
    return hipSuccess;
}

hipError_t launch(const rt_context* ctx, const Launch& L, int stage, uint32_t* d_pixels, float* d_radiance,
                  unsigned long long* d_rays, const rt::Workspace& ws, hipStream_t stream, bool plan_only = false) {
    const uint4* image = ctx->d_image[L.imode];
    // the staged scene image, plus one sphere staging slice per wave when the world has spheres
    const size_t lds = (size_t)ctx->lds_words * 4 + (ctx->sphere_count ? (size_t)(L.block_threads / 64u) * rt::SPHERE_SLICE_BYTES : 0u);
    // RT_INTERSECT_WIDE4 is the wide routine on an image whose primitives are dealt to four lanes: same kernels.
    const int kernel_imode = L.imode == RT_INTERSECT_WIDE4 ? RT_INTERSECT_WIDE8 : L.imode;
    const int prims = kernel_imode != RT_INTERSECT_WIDE8 ? rt::PRIMS_BOTH : ctx->mixed ? rt::PRIMS_BOTH
                      : !ctx->sphere_count ? ((L.kp.rect_about_y | L.kp.rect_about_x) ? rt::PRIMS_CLASSED_RECTS : rt::PRIMS_RECTS)
                      : ctx->sphere_count <= 64u ? rt::PRIMS_FEW_SPHERES : rt::PRIMS_SPHERES;
#define RT_DISPATCH(IM, RM, PR, ST)                                                  \
    if (kernel_imode == IM && L.rmode == RM && prims == PR && stage == ST)           \
        return launch_variant<IM, RM, PR, ST>(ctx, L, lds, image, d_pixels, d_radiance, d_rays, ws, stream, plan_only);
    RT_FOR_EACH_KERNEL(RT_DISPATCH)
#undef RT_DISPATCH
    return hipErrorInvalidValue;
}

// Grow-only device buffers behind the entry points that return results in host memory; each is
// only allocated once a caller asks for that output (a device allocation costs milliseconds).
rt_status ensure_output_capacity(rt_context* ctx, size_t count, bool want_pixels, bool want_radiance) {
    if (want_pixels && count > ctx->pixel_capacity) {
        if (ctx->d_pixels) { RT_HIP_CHECK(ctx, hipFree(ctx->d_pixels)); ctx->d_pixels = nullptr; }
        ctx->pixel_capacity = 0;
        RT_HIP_CHECK(ctx, hipMalloc(&ctx->d_pixels, count * sizeof(uint32_t)));
        ctx->pixel_capacity = count;
    }
    if (want_radiance && count > ctx->radiance_capacity) {
        if (ctx->d_radiance) { RT_HIP_CHECK(ctx, hipFree(ctx->d_radiance)); ctx->d_radiance = nullptr; }
        ctx->radiance_capacity = 0;
        ctx->accumulator_valid = false;
        RT_HIP_CHECK(ctx, hipMalloc(&ctx->d_radiance, count * 3 * sizeof(float)));
        ctx->radiance_capacity = count;
    }
    return RT_OK;
}

rt_status ensure_workspace(rt_context* ctx, const Launch& L) {
    const size_t streams = (size_t)L.kp.local_rows * L.kp.width * L.kp.lanes_per_pixel;
    if (!ctx->ws.pool_count) {   // the pool's fill count and, 256 bytes on, the partial ray counters
        RT_HIP_CHECK(ctx, hipMalloc(&ctx->ws.pool_count, COUNTER_BLOCK_BYTES));
        ctx->ws.ray_slots = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(ctx->ws.pool_count) + 256);
    }
    if (streams > ctx->lane_sum_capacity) {
        if (ctx->ws.lane_sums) { RT_HIP_CHECK(ctx, hipFree(ctx->ws.lane_sums)); ctx->ws.lane_sums = nullptr; }
        ctx->lane_sum_capacity = 0;
        RT_HIP_CHECK(ctx, hipMalloc(&ctx->ws.lane_sums, streams * sizeof(float4)));
        ctx->lane_sum_capacity = streams;
    }
    const size_t pixel_count = (size_t)L.kp.local_rows * L.kp.width;
    if (pixel_count > ctx->pending_capacity) {
        if (ctx->ws.pending) { RT_HIP_CHECK(ctx, hipFree(ctx->ws.pending)); ctx->ws.pending = nullptr; }
        ctx->pending_capacity = 0;
        RT_HIP_CHECK(ctx, hipMalloc(&ctx->ws.pending, pixel_count));
        ctx->pending_capacity = pixel_count;
    }
    if (L.kp.pool_capacity > ctx->pool_entry_capacity) {
        if (ctx->ws.pool) { RT_HIP_CHECK(ctx, hipFree(ctx->ws.pool)); ctx->ws.pool = nullptr; }
        ctx->pool_entry_capacity = 0;
        RT_HIP_CHECK(ctx, hipMalloc(&ctx->ws.pool, (size_t)L.kp.pool_capacity * rt::POOL_VEC4 * sizeof(uint4)));
        ctx->pool_entry_capacity = L.kp.pool_capacity;
    }
    return RT_OK;
}

rt_status ensure_staging(rt_context* ctx, size_t bytes) {
    if (bytes <= ctx->staging_bytes) return RT_OK;
    if (ctx->h_staging) { RT_HIP_CHECK(ctx, hipHostFree(ctx->h_staging)); ctx->h_staging = nullptr; }
    ctx->staging_bytes = 0;
    RT_HIP_CHECK(ctx, hipHostMalloc(&ctx->h_staging, bytes, hipHostMallocDefault));
    ctx->staging_bytes = bytes;
    return RT_OK;
}

// Render + copy to host memory.  The device results travel through a pinned staging buffer of the
// context (a first copy into pageable memory costs the runtime ~8 ms of set-up, and pinned copies run
// at link speed), then row by row to their place: compact (local row order) for rt_render, or at the
// row's global position relative to `base_row` in full-size outputs (`scatter`, rt_render_multi).
rt_status render_to_host(rt_context* ctx, const rt_render_params* params, uint32_t* pixels, float* radiance, uint64_t* rays,
                         bool scatter, uint32_t base_row) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (!ctx->has_world) return fail(ctx, RT_ERR_NO_WORLD, "rt_upload_world has not been called");
    Launch L;
    rt_status st = validate_and_plan(ctx, params, &L);
    if (st != RT_OK) return st;
    RT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t width = L.kp.width, rows = L.kp.local_rows, count = rows * width;
    const size_t pixel_bytes = pixels ? count * sizeof(uint32_t) : 0, radiance_bytes = radiance ? count * 3 * sizeof(float) : 0;
    // The context's radiance buffer doubles as the running sum of progressive renders: a sum render starts
    // it, accumulating renders of the same rows add to it, anything else that writes the buffer ends it.
    const bool accumulate = params->flags & RT_FLAG_ACCUMULATE, starts_sum = !accumulate && (params->flags & RT_FLAG_OUTPUT_SUM);
    const uint32_t key[6] = {params->width, params->row_begin, params->row_end, params->band_rows, params->band_index, params->band_count};
    if (accumulate && (!ctx->accumulator_valid || std::memcmp(key, ctx->accumulator_key, sizeof key) != 0))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT,
                    "RT_FLAG_ACCUMULATE: no running sum for these rows (render them with RT_FLAG_OUTPUT_SUM or call rt_load_accumulator first)");
    const bool device_radiance = radiance || accumulate || starts_sum;
    st = ensure_output_capacity(ctx, count, pixels != nullptr, device_radiance);
    if (st != RT_OK) return st;
    st = ensure_staging(ctx, pixel_bytes + radiance_bytes + 16);
    if (st != RT_OK) return st;
    if (device_radiance && !accumulate) ctx->accumulator_valid = false;
    st = rt_render_device(ctx, params, pixels ? ctx->d_pixels : nullptr, device_radiance ? ctx->d_radiance : nullptr,
                          reinterpret_cast<uint64_t*>(ctx->d_rays), nullptr);
    if (st != RT_OK) return st;
    if (starts_sum) { ctx->accumulator_valid = true; std::memcpy(ctx->accumulator_key, key, sizeof key); }
    char* stage = static_cast<char*>(ctx->h_staging);
    if (pixels) RT_HIP_CHECK(ctx, hipMemcpyAsync(stage, ctx->d_pixels, pixel_bytes, hipMemcpyDeviceToHost, nullptr));
    if (radiance) RT_HIP_CHECK(ctx, hipMemcpyAsync(stage + pixel_bytes, ctx->d_radiance, radiance_bytes, hipMemcpyDeviceToHost, nullptr));
    RT_HIP_CHECK(ctx, hipMemcpyAsync(stage + pixel_bytes + radiance_bytes, ctx->d_rays, sizeof(unsigned long long), hipMemcpyDeviceToHost, nullptr));
    RT_HIP_CHECK(ctx, hipStreamSynchronize(nullptr));
    const uint32_t* staged_pixels = reinterpret_cast<const uint32_t*>(stage);
    const float* staged_radiance = reinterpret_cast<const float*>(stage + pixel_bytes);
    if (!scatter) {
        if (pixels) std::memcpy(pixels, staged_pixels, pixel_bytes);
        if (radiance) std::memcpy(radiance, staged_radiance, radiance_bytes);
    } else {
        for (uint32_t local = 0; local < rows; ++local) {
            const size_t dst = (size_t)(rt_global_row(params, local) - base_row) * width;
            if (pixels) std::memcpy(pixels + dst, staged_pixels + (size_t)local * width, width * sizeof(uint32_t));
            if (radiance) std::memcpy(radiance + dst * 3, staged_radiance + (size_t)local * width * 3, width * 3 * sizeof(float));
        }
    }
    if (rays) {
        unsigned long long r;
        std::memcpy(&r, stage + pixel_bytes + radiance_bytes, sizeof r);
        *rays = r;
    }
    return RT_OK;
}

// No exception may cross the C ABI: the entry points that touch std::vector, std::string or std::thread run
// their bodies through this.
template <typename Body>
rt_status guarded(Body&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return RT_ERR_OUT_OF_MEMORY;
    } catch (...) {
        return RT_ERR_HIP;
    }
}

// Layout of the fixed per-GPU workspace behind rt_render_device_noalloc (STATIC_SCRATCH_BYTES in all):
//   [0, 256)                     fill count of the pool
//   [256, 4352)                  partial ray counters (rt::RAY_SLOTS x 64 bytes)
//   [4352, 4352 + 1 MiB)         one pending flag per pixel          -> at most 2^20 pixels
//   [.., + 16 MiB)               one float4 sum per stream            -> at most 2^20 streams (pixels x lanes_per_pixel)
//   rest (about 15 MiB)          pool of handed-over streams, 80 bytes each
constexpr uint64_t STATIC_MAX_PIXELS = 1u << 20, STATIC_MAX_STREAMS = 1u << 20;
constexpr size_t STATIC_PENDING_OFFSET = COUNTER_BLOCK_BYTES;
constexpr size_t STATIC_LANE_SUMS_OFFSET = STATIC_PENDING_OFFSET + STATIC_MAX_PIXELS;
constexpr size_t STATIC_POOL_OFFSET = STATIC_LANE_SUMS_OFFSET + STATIC_MAX_STREAMS * sizeof(float4);
constexpr uint64_t STATIC_POOL_ENTRIES = (STATIC_SCRATCH_BYTES - STATIC_POOL_OFFSET) / (rt::POOL_VEC4 * sizeof(uint4));
static_assert(STATIC_POOL_OFFSET % 256 == 0 && STATIC_POOL_OFFSET < STATIC_SCRATCH_BYTES, "static workspace layout");

// Enqueues the kernels of one render on `hip_stream`.  static_workspace = false: the compacted render's
// workspace is the context's (allocated by the first render that needs it, or by rt_reserve).
// static_workspace = true: it is this GPU's fixed array and the call allocates nothing; sizes the array
// cannot hold are refused.
rt_status render_device(rt_context* ctx, const rt_render_params* params, uint32_t* d_pixels, float* d_radiance,
                        uint64_t* d_rays, void* hip_stream, bool static_workspace) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (!ctx->has_world) return fail(ctx, RT_ERR_NO_WORLD, "rt_upload_world has not been called");
    Launch L;
    rt_status st = validate_and_plan(ctx, params, &L, static_workspace ? STATIC_POOL_ENTRIES : 0);
    if (st != RT_OK) return st;
    if ((params->flags & RT_FLAG_ACCUMULATE) && (!d_radiance || d_pixels))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_FLAG_ACCUMULATE needs a radiance buffer and no pixel buffer");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    RT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    unsigned long long* rays = reinterpret_cast<unsigned long long*>(d_rays);
    rt::Workspace ws = ctx->ws;
    if (L.compact && static_workspace) {
        const uint64_t pixel_count = (uint64_t)L.kp.local_rows * L.kp.width, streams = pixel_count * L.kp.lanes_per_pixel;
        if (pixel_count > STATIC_MAX_PIXELS || streams > STATIC_MAX_STREAMS)
            return fail(ctx, RT_ERR_TOO_LARGE,
                        "rt_render_device_noalloc: %llu pixels x %u lanes per pixel exceed the fixed workspace (at most %llu pixels and %llu streams with RT_FLAG_COMPACT_TAILS)",
                        (unsigned long long)pixel_count, L.kp.lanes_per_pixel, (unsigned long long)STATIC_MAX_PIXELS,
                        (unsigned long long)STATIC_MAX_STREAMS);
        unsigned char* base = ctx->static_scratch;   // this device's copy of the array, from hipGetSymbolAddress
        ws.pool_count = reinterpret_cast<uint32_t*>(base);
        ws.ray_slots = reinterpret_cast<unsigned long long*>(base + 256);
        ws.pending = base + STATIC_PENDING_OFFSET;
        ws.lane_sums = reinterpret_cast<float4*>(base + STATIC_LANE_SUMS_OFFSET);
        ws.pool = reinterpret_cast<uint4*>(base + STATIC_POOL_OFFSET);
    } else if (L.compact) {
        st = ensure_workspace(ctx, L);   // allocates only when this render needs more than any before it
        if (st != RT_OK) return st;
        ws = ctx->ws;
    }
    if (rays || L.compact) {
        // the fill count of the pool must start at zero; everything else in the workspace is written before it is read
        hipLaunchKernelGGL(rt::clear_counter_kernel, dim3(1), dim3(rt::RAY_SLOTS), 0, stream, rays, L.compact ? ws.pool_count : nullptr,
                           L.compact ? ws.ray_slots : nullptr);
        RT_HIP_CHECK(ctx, hipGetLastError());
    }
    const bool timed = params->flags & RT_FLAG_TIME_KERNEL;
    if (timed) RT_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start, stream));
    if (!L.compact) {
        RT_HIP_CHECK(ctx, launch(ctx, L, 0, d_pixels, d_radiance, rays, ws, stream));
    } else {
        RT_HIP_CHECK(ctx, launch(ctx, L, 1, d_pixels, d_radiance, rays, ws, stream));
        RT_HIP_CHECK(ctx, launch(ctx, L, 2, d_pixels, d_radiance, rays, ws, stream));
        const size_t pixel_count = (size_t)L.kp.local_rows * L.kp.width;
        hipLaunchKernelGGL(rt::pixel_combine_kernel, dim3((uint32_t)((pixel_count + rt::BLOCK_THREADS - 1) / rt::BLOCK_THREADS)),
                           dim3(rt::BLOCK_THREADS), 0, stream, L.kp, ws.lane_sums, ws.pending, d_pixels, d_radiance,
                           ws.ray_slots, rays);
        RT_HIP_CHECK(ctx, hipGetLastError());
    }
    if (timed) RT_HIP_CHECK(ctx, hipEventRecord(ctx->ev_stop, stream));
    ctx->timed = timed;
    return RT_OK;
}

}  // namespace

extern "C" {

uint32_t rt_abi_version(void) { return RT_ABI_VERSION; }

const char* rt_status_string(rt_status status) {
    switch (status) {
        case RT_OK: return "ok";
        case RT_ERR_INVALID_ARGUMENT: return "invalid argument";
        case RT_ERR_NO_DEVICE: return "no usable HIP device";
        case RT_ERR_HIP: return "HIP runtime error";
        case RT_ERR_NO_WORLD: return "no world uploaded";
        case RT_ERR_TOO_LARGE: return "too large (planes, rectangles and materials for the LDS of a workgroup, or a render for the fixed workspace of rt_render_device_noalloc)";
        case RT_ERR_OUT_OF_MEMORY: return "out of device memory";
    }
    return "unknown status";
}

static rt_status create_impl(int device_ordinal, rt_context** out_context) {
    if (!out_context) return RT_ERR_INVALID_ARGUMENT;
    *out_context = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return RT_ERR_NO_DEVICE;
    if (device_ordinal < 0 || device_ordinal >= count) return RT_ERR_NO_DEVICE;
    rt_context* ctx = new (std::nothrow) rt_context;
    if (!ctx) return RT_ERR_OUT_OF_MEMORY;
    ctx->device = device_ordinal;
    if (hipSetDevice(device_ordinal) != hipSuccess || hipGetDeviceProperties(&ctx->props, device_ordinal) != hipSuccess ||
        hipMalloc(&ctx->d_rays, sizeof(unsigned long long)) != hipSuccess ||
        hipEventCreate(&ctx->ev_start) != hipSuccess || hipEventCreate(&ctx->ev_stop) != hipSuccess) {
        rt_destroy(ctx);
        return RT_ERR_HIP;
    }
    // One-off runtime set-up belongs to context creation, not to the first render: the first
    // device-to-host transfer of a process costs the HIP runtime several milliseconds (copy engine
    // queue, staging), so a 1 MB transfer through the pinned buffer is done here.
    // Setting the kernels' LDS allowance also makes the runtime load this library's code object now.
    constexpr size_t warm_bytes = 1u << 20;
    void* d_warm = nullptr;
    const size_t lds_max = ctx->props.sharedMemPerBlockOptin ? ctx->props.sharedMemPerBlockOptin : ctx->props.sharedMemPerBlock;
#ifdef RT_STAMPS
    const size_t lds_static = 4096;   // the diagnostic build's per-wave tables
#else
    const size_t lds_static = 0;
#endif
    if (allow_dynamic_lds(lds_max - lds_static) != hipSuccess ||
        ensure_staging(ctx, warm_bytes) != RT_OK || hipMalloc(&d_warm, warm_bytes) != hipSuccess ||
        hipMemsetAsync(d_warm, 0, warm_bytes, nullptr) != hipSuccess ||
        hipMemcpyAsync(ctx->h_staging, d_warm, warm_bytes, hipMemcpyDeviceToHost, nullptr) != hipSuccess ||
        hipStreamSynchronize(nullptr) != hipSuccess) {
        if (d_warm) (void)hipFree(d_warm);
        rt_destroy(ctx);
        return RT_ERR_HIP;
    }
    (void)hipFree(d_warm);
    // This device's copy of the fixed workspace (see `scratch`): looked up once per device, with that device current.
    {
        constexpr int MAX_DEVICES = 256;
        static std::mutex lookup_mutex;
        static unsigned char* per_device[MAX_DEVICES] = {};     // no heap: nothing here can throw bad_alloc
        if (device_ordinal >= MAX_DEVICES) {
            rt_destroy(ctx);
            return RT_ERR_NO_DEVICE;
        }
        std::lock_guard<std::mutex> lock(lookup_mutex);
        if (!per_device[device_ordinal]) {
            void* address = nullptr;
            if (hipGetSymbolAddress(&address, HIP_SYMBOL(scratch)) != hipSuccess || !address) {
                rt_destroy(ctx);
                return RT_ERR_HIP;
            }
            per_device[device_ordinal] = static_cast<unsigned char*>(address);
        }
        ctx->static_scratch = per_device[device_ordinal];
    }
    *out_context = ctx;
    return RT_OK;
}

rt_status rt_create(int device_ordinal, rt_context** out_context) {
    return guarded([&] { return create_impl(device_ordinal, out_context); });   // std::mutex may throw system_error
}

void rt_destroy(rt_context* ctx) {
    if (!ctx) return;
    if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
    for (int i = 0; i < 3; ++i)
        if (ctx->d_image[i]) (void)hipFree(ctx->d_image[i]);
    if (ctx->d_pixels) (void)hipFree(ctx->d_pixels);
    if (ctx->d_radiance) (void)hipFree(ctx->d_radiance);
    if (ctx->d_resolve_sum) (void)hipFree(ctx->d_resolve_sum);
    if (ctx->d_rays) (void)hipFree(ctx->d_rays);
    if (ctx->h_staging) (void)hipHostFree(ctx->h_staging);
    if (ctx->ws.lane_sums) (void)hipFree(ctx->ws.lane_sums);
    if (ctx->ws.pool) (void)hipFree(ctx->ws.pool);
    if (ctx->ws.pending) (void)hipFree(ctx->ws.pending);
    if (ctx->ws.pool_count) (void)hipFree(ctx->ws.pool_count);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    delete ctx;
}

const char* rt_error_string(const rt_context* ctx) { return ctx ? ctx->error.c_str() : "context is NULL"; }

rt_status rt_get_device_info(const rt_context* ctx, rt_device_info* info) {
    if (!ctx || !info) return RT_ERR_INVALID_ARGUMENT;
    std::memset(info, 0, sizeof *info);
    std::snprintf(info->name, sizeof info->name, "%s", ctx->props.name);
    std::snprintf(info->arch, sizeof info->arch, "%s", ctx->props.gcnArchName);
    if (char* colon = std::strchr(info->arch, ':')) *colon = 0;
    info->compute_units = (uint32_t)ctx->props.multiProcessorCount;
    info->clock_khz = (uint32_t)ctx->props.clockRate;
    info->lds_bytes_per_workgroup = (uint32_t)ctx->props.sharedMemPerBlockOptin;
    if (info->lds_bytes_per_workgroup == 0) info->lds_bytes_per_workgroup = (uint32_t)ctx->props.sharedMemPerBlock;
    info->global_memory_bytes = ctx->props.totalGlobalMem;
    // 4 SIMDs x 32 lanes per CU, one FMA (2 flop) per lane per clock.
    info->fp32_vector_peak_tflops = (double)info->compute_units * 128.0 * 2.0 * (double)info->clock_khz * 1.0e3 / 1.0e12;
    return RT_OK;
}

void rt_default_params(rt_render_params* p, uint32_t width, uint32_t height, uint32_t spp) {
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    p->struct_size = sizeof *p;
    p->width = width; p->height = height; p->samples_per_pixel = spp;
    p->row_begin = 0; p->row_end = height;
    p->band_rows = 0; p->band_index = 0; p->band_count = 1;
    p->max_bounces = 8;             // main.cpp:275
    p->min_hit_distance = 0.001f;   // main.cpp:29
    p->plane_tolerance = 0.001f;    // main.cpp:28
    p->seed_constant = 262346u;     // main.cpp:370
    p->rng_mode = RT_RNG_PIXEL;
    p->rows_per_order = 1;          // main.cpp:433
    p->lanes_per_pixel = 1;
    p->batch_offset = 0;
    p->intersect_mode = RT_INTERSECT_WIDE8;
    p->flags = 0;
}

uint32_t rt_local_row_count(const rt_render_params* p) {
    if (!p || p->row_end <= p->row_begin) return 0;
    uint32_t total = p->row_end - p->row_begin;
    if (p->band_rows == 0 || p->band_count <= 1) return total;
    uint32_t bands = (total + p->band_rows - 1) / p->band_rows, n = 0;
    for (uint32_t b = p->band_index; b < bands; b += p->band_count) {
        uint32_t lo = b * p->band_rows, hi = lo + p->band_rows < total ? lo + p->band_rows : total;
        n += hi - lo;
    }
    return n;
}

uint32_t rt_global_row(const rt_render_params* p, uint32_t local) {
    if (p->band_rows == 0 || p->band_count <= 1) return p->row_begin + local;
    uint32_t band = local / p->band_rows, within = local % p->band_rows;
    return p->row_begin + (band * p->band_count + p->band_index) * p->band_rows + within;
}

static rt_status upload_world_impl(rt_context* ctx, const rt_world* w) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (!w || !w->camera || !w->materials || w->material_count == 0)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "world needs a camera and at least the sky material");
    if ((w->plane_count && !w->planes) || (w->sphere_count && !w->spheres) || (w->rectangle_count && !w->rectangles))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "non-zero primitive count with a NULL array");
    if (w->material_count > rt::TAG_MATERIAL_MASK)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "too many materials");
    for (uint32_t i = 0; i < w->plane_count; ++i)
        if (w->planes[i].material_index >= w->material_count) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "plane %u: material index out of range", i);
    for (uint32_t i = 0; i < w->sphere_count; ++i)
        if (w->spheres[i].material_index >= w->material_count) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "sphere %u: material index out of range", i);
    for (uint32_t i = 0; i < w->rectangle_count; ++i)
        if (w->rectangles[i].material_index >= w->material_count) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rectangle %u: material index out of range", i);

    // Every scene value must be finite and of magnitude <= 1e18: with such values no intermediate of
    // the kernel's shortened arithmetic forms (see rt_kernels.hpp) overflows, which is what their
    // equality with the reference's operations rests on.  (The reference itself turns larger values
    // into inf/NaN rays.)
    auto bounded = [](const float* v, size_t n) {
        for (size_t i = 0; i < n; ++i)
            if (!(std::fabs(v[i]) <= 1.0e18f)) return false;   // false for NaN too
        return true;
    };
    for (uint32_t i = 0; i < w->material_count; ++i)
        if (!bounded(w->materials[i].color, 8)) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "material %u: non-finite or huge value", i);
    for (uint32_t i = 0; i < w->plane_count; ++i)
        if (!bounded(w->planes[i].normal, 4)) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "plane %u: non-finite or huge value", i);
    for (uint32_t i = 0; i < w->sphere_count; ++i)
        if (!bounded(w->spheres[i].position, 4)) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "sphere %u: non-finite or huge value", i);
    for (uint32_t i = 0; i < w->rectangle_count; ++i)
        if (!bounded(&w->rectangles[i].inverse_transform[0][0], 19))
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rectangle %u: non-finite or huge value", i);
    if (!bounded(w->camera->position, 12)) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "camera: non-finite or huge value");

    RT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    size_t lds_limit = ctx->props.sharedMemPerBlockOptin ? ctx->props.sharedMemPerBlockOptin : ctx->props.sharedMemPerBlock;
    std::vector<uint32_t> images[3] = {rt::pack_scene_image(*w, 8), rt::pack_scene_image(*w, 0), rt::pack_scene_image(*w, 4)};
    // a world with spheres also needs the waves' sphere staging slices, at most 16 waves per workgroup
    const size_t staging = w->sphere_count ? 16u * rt::SPHERE_SLICE_BYTES : 0u;
    for (int i = 0; i < 3; ++i)
        if ((size_t)images[i][rt::H_LDS_WORDS] * 4 + staging > lds_limit)
            return fail(ctx, RT_ERR_TOO_LARGE, "planes, rectangles and materials take %zu bytes, the LDS of one workgroup holds %zu",
                        (size_t)images[i][rt::H_LDS_WORDS] * 4 + staging, lds_limit);
    // The new images are allocated and filled before the context is touched, so a failure leaves the
    // previous world in place and usable.
    uint4* fresh[3] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < 3; ++i) {
        hipError_t e = hipMalloc(&fresh[i], images[i].size() * 4);
        if (e == hipSuccess) e = hipMemcpy(fresh[i], images[i].data(), images[i].size() * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            for (int j = 0; j <= i; ++j)
                if (fresh[j]) (void)hipFree(fresh[j]);
            return fail(ctx, e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP, "rt_upload_world: %s", hipGetErrorString(e));
        }
    }
    if (hipError_t e = hipDeviceSynchronize(); e != hipSuccess) {   // no render may still read the old images
        for (int j = 0; j < 3; ++j) (void)hipFree(fresh[j]);
        return fail(ctx, RT_ERR_HIP, "rt_upload_world: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < 3; ++i) {
        if (ctx->d_image[i]) (void)hipFree(ctx->d_image[i]);
        ctx->d_image[i] = fresh[i];
        ctx->image_words[i] = (uint32_t)images[i].size();
        ctx->rect_about_y[i] = images[i][rt::H_RECT_ABOUT_Y] | ((unsigned long long)images[i][rt::H_RECT_ABOUT_Y + 1] << 32);
        ctx->rect_about_x[i] = images[i][rt::H_RECT_ABOUT_X] | ((unsigned long long)images[i][rt::H_RECT_ABOUT_X + 1] << 32);
    }
    ctx->camera = *w->camera;
    ctx->lds_words = images[0][rt::H_LDS_WORDS];
    ctx->sphere_count = w->sphere_count;
    ctx->sphere_offset = images[0][rt::H_OFF_SPHERES];
    ctx->sphere_tag_offset = images[0][rt::H_OFF_SPHERE_TAGS];
    ctx->mixed = w->sphere_count && w->rectangle_count;
    ctx->has_world = true;
    ctx->accumulator_valid = false;
    // Size stage 2 for this world's kernel variants now (an occupancy query each), so that no later launch -
    // a captured one, or the entry point that must not allocate - is the first to ask.  Best effort: the world
    // is in place, and a failed query only means the default of 16 waves per CU (stage2_waves_per_cu).
    for (int imode : {RT_INTERSECT_WIDE8, RT_INTERSECT_SCALAR, RT_INTERSECT_WIDE4})
        for (int classed = 0; classed < 2; ++classed) {   // with the class masks, and as RT_FLAG_DENSE_TRANSFORMS renders
            Launch L{};
            L.imode = imode; L.rmode = RT_RNG_PIXEL; L.blocks = 1; L.block_threads = workgroup_threads(ctx);
            L.kp.rect_about_y = classed ? ctx->rect_about_y[imode] : 0ull;
            L.kp.rect_about_x = classed ? ctx->rect_about_x[imode] : 0ull;
            (void)launch(ctx, L, 2, nullptr, nullptr, nullptr, ctx->ws, nullptr, /*plan_only=*/true);
        }
    return RT_OK;
}

rt_status rt_render_device(rt_context* ctx, const rt_render_params* params, uint32_t* d_pixels, float* d_radiance,
                           uint64_t* d_rays, void* hip_stream) {
    return guarded([&] { return render_device(ctx, params, d_pixels, d_radiance, d_rays, hip_stream, false); });
}

rt_status rt_render_device_noalloc(rt_context* ctx, const rt_render_params* params, uint32_t* d_pixels, float* d_radiance,
                                   uint64_t* d_rays, void* hip_stream) {
    return guarded([&] { return render_device(ctx, params, d_pixels, d_radiance, d_rays, hip_stream, true); });
}

static rt_status reserve_impl(rt_context* ctx, const rt_render_params* params) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (!ctx->has_world) return fail(ctx, RT_ERR_NO_WORLD, "rt_upload_world has not been called");
    Launch L;
    rt_status st = validate_and_plan(ctx, params, &L);
    if (st != RT_OK) return st;
    RT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (!L.compact) return RT_OK;
    RT_HIP_CHECK(ctx, launch(ctx, L, 2, nullptr, nullptr, nullptr, ctx->ws, nullptr, /*plan_only=*/true));   // sizes stage 2 now
    return ensure_workspace(ctx, L);
}

static rt_status load_accumulator_impl(rt_context* ctx, const rt_render_params* params, const float* radiance_sum) {
    if (!ctx || !radiance_sum) return RT_ERR_INVALID_ARGUMENT;
    if (!ctx->has_world) return fail(ctx, RT_ERR_NO_WORLD, "rt_upload_world has not been called");
    Launch L;
    rt_status st = validate_and_plan(ctx, params, &L);
    if (st != RT_OK) return st;
    RT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t count = (size_t)L.kp.local_rows * L.kp.width;
    st = ensure_output_capacity(ctx, count, false, true);
    if (st != RT_OK) return st;
    ctx->accumulator_valid = false;
    RT_HIP_CHECK(ctx, hipMemcpy(ctx->d_radiance, radiance_sum, count * 3 * sizeof(float), hipMemcpyHostToDevice));
    const uint32_t key[6] = {params->width, params->row_begin, params->row_end, params->band_rows, params->band_index, params->band_count};
    std::memcpy(ctx->accumulator_key, key, sizeof key);
    ctx->accumulator_valid = true;
    return RT_OK;
}

rt_status rt_render(rt_context* ctx, const rt_render_params* params, uint32_t* pixels, float* radiance, uint64_t* rays) {
    return guarded([&] { return render_to_host(ctx, params, pixels, radiance, rays, false, 0); });
}

rt_status rt_upload_world(rt_context* ctx, const rt_world* w) {
    return guarded([&] { return upload_world_impl(ctx, w); });
}

rt_status rt_reserve(rt_context* ctx, const rt_render_params* params) {
    return guarded([&] { return reserve_impl(ctx, params); });
}

rt_status rt_load_accumulator(rt_context* ctx, const rt_render_params* params, const float* radiance_sum) {
    return guarded([&] { return load_accumulator_impl(ctx, params, radiance_sum); });
}

static rt_status render_multi_impl(rt_context* const* contexts, uint32_t context_count, const rt_render_params* params,
                                   uint32_t* pixels, float* radiance, uint64_t* rays) {
    if (!contexts || context_count == 0 || !params) return RT_ERR_INVALID_ARGUMENT;
    for (uint32_t i = 0; i < context_count; ++i)
        if (!contexts[i]) return RT_ERR_INVALID_ARGUMENT;
    if (params->struct_size != sizeof(rt_render_params) || params->row_end <= params->row_begin || params->width == 0)
        return fail(contexts[0], RT_ERR_INVALID_ARGUMENT, "rt_render_multi: bad struct_size, width or row range");
    // A context keeps ONE running sum: listed twice, its second share would overwrite the first one's.
    if (params->flags & (RT_FLAG_OUTPUT_SUM | RT_FLAG_ACCUMULATE))
        for (uint32_t i = 0; i < context_count; ++i)
            for (uint32_t j = 0; j < i; ++j)
                if (contexts[i] == contexts[j])
                    return fail(contexts[0], RT_ERR_INVALID_ARGUMENT,
                                "rt_render_multi: a context listed more than once cannot render with RT_FLAG_OUTPUT_SUM or RT_FLAG_ACCUMULATE");
    std::vector<rt_render_params> shard_params(context_count, *params);
    std::vector<uint64_t> shard_rays(context_count, 0);
    std::vector<rt_status> status(context_count, RT_OK);
    for (uint32_t i = 0; i < context_count; ++i) {
        rt_render_params& p = shard_params[i];
        p.band_rows = context_count > 1 ? 8u : 0u;
        p.band_index = i;
        p.band_count = context_count;
    }
    // Each share goes from its GPU's staging buffer straight to its rows of the full-size outputs.
    auto render_share = [&](uint32_t j) {
        if (rt_local_row_count(&shard_params[j]) == 0) return;   // more GPUs than bands
        status[j] = render_to_host(contexts[j], &shard_params[j], pixels, radiance, &shard_rays[j], true, params->row_begin);
    };
    if (context_count == 1) {
        render_share(0);
    } else {
        // One thread per distinct context; a context listed twice renders its shares back to back.
        std::vector<std::thread> workers;
        for (uint32_t i = 0; i < context_count; ++i) {
            bool first = true;
            for (uint32_t j = 0; j < i; ++j) first = first && contexts[j] != contexts[i];
            if (!first) continue;
            workers.emplace_back([&, i] {
                for (uint32_t j = i; j < context_count; ++j)
                    if (contexts[j] == contexts[i]) render_share(j);
            });
        }
        for (std::thread& t : workers) t.join();
    }
    uint64_t total = 0;
    for (uint32_t i = 0; i < context_count; ++i) {
        if (status[i] != RT_OK) {
            if (contexts[i] != contexts[0]) contexts[0]->error = contexts[i]->error;
            return status[i];
        }
        total += shard_rays[i];
    }
    if (rays) *rays = total;
    return RT_OK;
}

rt_status rt_combine_device(rt_context* ctx, const float* d_sums, uint32_t part_count, uint64_t part_stride,
                            uint64_t value_count, float* d_out, void* hip_stream) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (!d_sums || !d_out || part_count == 0 || part_count > 64 || value_count == 0 || value_count > (1ull << 35) ||
        (part_count > 1 && part_stride < value_count))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_combine_device: bad pointers, part count (1..64), stride or value count");
    RT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t blocks = (uint32_t)((value_count + rt::BLOCK_THREADS - 1) / rt::BLOCK_THREADS);
    hipLaunchKernelGGL(rt::combine_kernel, dim3(blocks), dim3(rt::BLOCK_THREADS), 0, static_cast<hipStream_t>(hip_stream), d_sums,
                       part_count, part_stride, value_count, d_out);
    RT_HIP_CHECK(ctx, hipGetLastError());
    return RT_OK;
}

rt_status rt_resolve_device(rt_context* ctx, const float* d_sum, uint64_t pixel_count, uint32_t total_samples,
                            uint32_t* d_pixels, float* d_mean, void* hip_stream) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (!d_sum || pixel_count == 0 || total_samples == 0 || pixel_count > (1ull << 33))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_resolve_device: bad sum pointer, pixel count or sample count");
    RT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    uint32_t blocks = (uint32_t)((pixel_count + rt::BLOCK_THREADS - 1) / rt::BLOCK_THREADS);
    hipLaunchKernelGGL(rt::resolve_kernel, dim3(blocks), dim3(rt::BLOCK_THREADS), 0, static_cast<hipStream_t>(hip_stream), d_sum,
                       pixel_count, (float)total_samples, d_pixels, d_mean);
    RT_HIP_CHECK(ctx, hipGetLastError());
    return RT_OK;
}

static rt_status resolve_impl(rt_context* ctx, const float* radiance_sum, uint64_t pixel_count, uint32_t total_samples,
                               uint32_t* pixels, float* radiance_mean) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (!radiance_sum || pixel_count == 0 || total_samples == 0 || pixel_count > (1ull << 33))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_resolve: bad sum pointer, pixel count or sample count");
    RT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    rt_status st = ensure_output_capacity(ctx, (size_t)pixel_count, pixels != nullptr, radiance_mean != nullptr);
    if (radiance_mean) ctx->accumulator_valid = false;   // the mean is staged in the radiance buffer
    if (st != RT_OK) return st;
    const size_t sum_bytes = (size_t)pixel_count * 3 * sizeof(float);
    if ((size_t)pixel_count > ctx->resolve_capacity) {   // grow-only buffer of the context: repeated resolves allocate nothing
        if (ctx->d_resolve_sum) { RT_HIP_CHECK(ctx, hipFree(ctx->d_resolve_sum)); ctx->d_resolve_sum = nullptr; }
        ctx->resolve_capacity = 0;
        RT_HIP_CHECK(ctx, hipMalloc(&ctx->d_resolve_sum, sum_bytes));
        ctx->resolve_capacity = (size_t)pixel_count;
    }
    float* d_sum = ctx->d_resolve_sum;
    hipError_t e = hipMemcpy(d_sum, radiance_sum, sum_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        st = rt_resolve_device(ctx, d_sum, pixel_count, total_samples, pixels ? ctx->d_pixels : nullptr,
                               radiance_mean ? ctx->d_radiance : nullptr, nullptr);
        if (st == RT_OK) e = hipStreamSynchronize(nullptr);
        if (st == RT_OK && e == hipSuccess && pixels)
            e = hipMemcpy(pixels, ctx->d_pixels, (size_t)pixel_count * sizeof(uint32_t), hipMemcpyDeviceToHost);
        if (st == RT_OK && e == hipSuccess && radiance_mean)
            e = hipMemcpy(radiance_mean, ctx->d_radiance, sum_bytes, hipMemcpyDeviceToHost);
    }
    if (st != RT_OK) return st;
    if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, "rt_resolve: %s", hipGetErrorString(e));
    return RT_OK;
}

rt_status rt_render_multi(rt_context* const* contexts, uint32_t context_count, const rt_render_params* params,
                          uint32_t* pixels, float* radiance, uint64_t* rays) {
    return guarded([&] { return render_multi_impl(contexts, context_count, params, pixels, radiance, rays); });
}

rt_status rt_resolve(rt_context* ctx, const float* radiance_sum, uint64_t pixel_count, uint32_t total_samples,
                     uint32_t* pixels, float* radiance_mean) {
    return guarded([&] { return resolve_impl(ctx, radiance_sum, pixel_count, total_samples, pixels, radiance_mean); });
}

// Test hook (not declared in include/rt_hip.h, runs on the host, needs no GPU): which rectangles of `world` would a
// wave whose pixels are [x0, x1] x [row0, row1] of a width x height image keep for its primary rays?  The scene
// image, the cull table, the film constants, the margins and the evaluation are the ones the render uses
// (pack_scene_image, film_and_cull_constants, cull_box, cull_keeps); reachable[i] answers for rectangle i of the
// caller's array.  intersect_mode selects the image (their stored camera rows differ in rounding).
rt_status rt_debug_cull_reachable(const rt_world* world, uint32_t width, uint32_t height, uint32_t intersect_mode, uint32_t x0,
                                  uint32_t x1, uint32_t row0, uint32_t row1, uint8_t* reachable) {
    return guarded([&]() -> rt_status {
        if (!world || !world->camera || !reachable || width == 0 || height == 0 || x1 < x0 || row1 < row0 || x1 >= width || row1 >= height)
            return RT_ERR_INVALID_ARGUMENT;
        const uint32_t lanes = intersect_mode == RT_INTERSECT_WIDE8 ? 8u : intersect_mode == RT_INTERSECT_WIDE4 ? 4u : 0u;
        const std::vector<uint32_t> image = rt::pack_scene_image(*world, lanes);
        rt::KernelParams k;
        std::memset(&k, 0, sizeof k);
        film_and_cull_constants(*world->camera, width, height, true, k);
        const std::vector<uint32_t> order = lanes ? rt::wide_order(world->rectangle_count, lanes) : std::vector<uint32_t>();
        const rt::CullBox box = rt::cull_box(width, height, k.pixel_w, k.pixel_h, k.half_w, k.half_h, k.cull_margin, x0, x1, row0, row1);
        for (uint32_t slot = 0; slot < world->rectangle_count; ++slot) {
            double rec[rt::CULL_DOUBLES];
            std::memcpy(rec, &image[image[rt::H_OFF_CULL] + 2u * rt::CULL_DOUBLES * slot], sizeof rec);
            const bool keeps = k.cull_margin > 0.0f ? rt::cull_keeps(rec, box, (double)k.cull_reach) : true;
            reachable[lanes ? order[slot] : slot] = keeps ? 1 : 0;
        }
        return RT_OK;
    });
}

// Test hook (host only, no GPU): the sparsity class (rt_layout.hpp, rect_sparsity_class) the scene image of
// `intersect_mode` records for rectangle i of the caller's array; 0 for rectangles beyond the first 64 slots.
rt_status rt_debug_rect_classes(const rt_world* world, uint32_t intersect_mode, uint8_t* classes) {
    return guarded([&]() -> rt_status {
        if (!world || !world->camera || !classes) return RT_ERR_INVALID_ARGUMENT;
        const uint32_t lanes = intersect_mode == RT_INTERSECT_WIDE8 ? 8u : intersect_mode == RT_INTERSECT_WIDE4 ? 4u : 0u;
        const std::vector<uint32_t> image = rt::pack_scene_image(*world, lanes);
        const std::vector<uint32_t> order = lanes ? rt::wide_order(world->rectangle_count, lanes) : std::vector<uint32_t>();
        for (uint32_t slot = 0; slot < world->rectangle_count; ++slot) {
            const bool y = slot < 64u && ((image[rt::H_RECT_ABOUT_Y + slot / 32u] >> (slot % 32u)) & 1u);
            const bool x = slot < 64u && ((image[rt::H_RECT_ABOUT_X + slot / 32u] >> (slot % 32u)) & 1u);
            classes[lanes ? order[slot] : slot] = y ? 1 : x ? 2 : 0;
        }
        return RT_OK;
    });
}

#ifdef RT_STAMPS
// Diagnostic build only: reads (and optionally clears) the table of per-region cycle sums and event counts.
int rt_debug_read_stamps(unsigned long long* out, int clear) {
    void* address = nullptr;
    if (hipGetSymbolAddress(&address, HIP_SYMBOL(rt::rt_stamp_table)) != hipSuccess) return 1;
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (out && hipMemcpy(out, address, sizeof(unsigned long long) * rt::ST_COUNT, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (clear && hipMemset(address, 0, sizeof(unsigned long long) * rt::ST_COUNT) != hipSuccess) return 1;
    return 0;
}
#endif

rt_status rt_last_kernel_ms(rt_context* ctx, float* out_ms) {
    if (!ctx || !out_ms) return RT_ERR_INVALID_ARGUMENT;
    if (!ctx->timed) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "the last render was not launched with RT_FLAG_TIME_KERNEL");
    RT_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev_stop));
    RT_HIP_CHECK(ctx, hipEventElapsedTime(out_ms, ctx->ev_start, ctx->ev_stop));
    return RT_OK;
}

}  // extern "C"
