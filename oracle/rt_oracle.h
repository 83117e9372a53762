/* TEST INFRASTRUCTURE - CPU restatement of the reference renderer's hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker (or as the timed CPU baseline).  The
 * product (raytracer_amd/, include/rt_hip.h) never links or calls it.
 *
 * What is restated (file:line into the reference tree):
 *   rto_xorshift32 / rto_unilateral / rto_bilateral   math_util.h:32-50
 *   rto_refract / rto_schlick                          math_util.h:52-93
 *   rto_linear_to_srgb / rto_pack_srgb                 math_util.h:102-125
 *   rto_intersect_scalar                               main.cpp:163-242
 *   rto_intersect_wide  (8 lanes over primitives)      main.cpp:26-161, simd.h, simd_avx2.h
 *   rto_raytrace_world                                 main.cpp:265-334
 *   rto_render          (row work orders, thread pool) main.cpp:336-400, 432-461
 *
 * Arithmetic convention: every scalar operation rounds once, in source order
 * (the library is built with -ffp-contract=off); fused multiply-adds appear only
 * where the reference calls FMulAdd / FMulSub.  That is the behaviour of the
 * reference itself when built with -ffp-contract=off (oracle/_ref/ref_harness_strict),
 * against which this file is pinned bit for bit by tests/golden.
 *
 * Pinned by: tests/golden/ (vectors captured from the reference's own functions by
 * oracle/ref_harness.cpp; generator oracle/make_golden.py).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* AoS scene records, byte-compatible with scene.h:13-47 and scene.h:163-168. */
typedef struct { float color[3]; float refractive_index; float emit_color[3]; float reflection; } rto_material;
typedef struct { float position[3]; float radius; uint32_t material_index; } rto_sphere;
typedef struct { float normal[3]; float d; uint32_t material_index; } rto_plane;
typedef struct { float m[4][4]; float normal[3]; uint32_t material_index; } rto_rectangle; /* m = inverse transform, row-major */
typedef struct { float position[3]; float z_axis[3]; float y_axis[3]; float x_axis[3]; } rto_camera;

typedef struct rto_scene rto_scene;

/* WIDE8 / WIDE8_AVX2 scale sphere normals by the x86 rsqrt approximation, as the reference does
 * (simd_avx2.h:139-144; hardware specific, relative error <= 1.5 * 2^-12).  WIDE8_IEEE uses the
 * correctly rounded 1/sqrtf instead and is otherwise identical: it is what the HIP kernels compute,
 * since no GPU instruction reproduces the x86 approximation. */
enum { RTO_INTERSECT_WIDE8 = 0, RTO_INTERSECT_SCALAR = 1, RTO_INTERSECT_WIDE8_AVX2 = 2, RTO_INTERSECT_WIDE8_IEEE = 3,
       /* LANE_WIDTH 4 (simd.h:17-26, simd_sse.h): same arithmetic, primitives dealt to 4 lanes.  The reference cannot
        * be built at that width without editing simd.h, so these two are PARITY UNPINNED where lanes tie.
        * WIDE8_LANES: the width-generic routine at width 8 (a test holds it equal to WIDE8). */
       RTO_INTERSECT_WIDE4 = 4, RTO_INTERSECT_WIDE4_IEEE = 5, RTO_INTERSECT_WIDE8_LANES = 6 };
enum { RTO_RNG_ROW_SERIAL = 0, RTO_RNG_PIXEL = 1 };

typedef struct {
    uint32_t width, height, samples_per_pixel;
    uint32_t row_begin, row_end;      /* rows rendered (global coordinates) */
    uint32_t band_rows, band_index, band_count; /* band_rows = 0: every row in [row_begin,row_end) */
    uint32_t max_bounces;             /* reference: 8          main.cpp:275 */
    float    min_hit_distance;        /* reference: 0.001      main.cpp:29  */
    float    plane_tolerance;         /* reference: 0.001      main.cpp:28  */
    uint32_t seed_constant;           /* reference: 262346     main.cpp:370 */
    uint32_t rng_mode;                /* RTO_RNG_*  */
    uint32_t rows_per_order;          /* ROW_SERIAL: rows per WorkOrder (reference: 1; 0 = whole range) */
    uint32_t lanes_per_pixel;         /* PIXEL: sample batches per pixel (power of two, divides spp) */
    uint32_t batch_offset;            /* PIXEL: index of this render's first batch */
    uint32_t intersect_mode;          /* RTO_INTERSECT_* */
    uint32_t output_sum;              /* 1: radiance receives the un-normalised sample sum */
} rto_params;

rto_scene* rto_scene_create(const rto_material* materials, uint32_t material_count,
                            const rto_plane* planes, uint32_t plane_count,
                            const rto_sphere* spheres, uint32_t sphere_count,
                            const rto_rectangle* rectangles, uint32_t rectangle_count,
                            const rto_camera* camera);
void rto_scene_destroy(rto_scene* scene);

void rto_default_params(rto_params* p, uint32_t width, uint32_t height, uint32_t spp);
uint32_t rto_local_row_count(const rto_params* p);
/* Global row of local row `local`. */
uint32_t rto_global_row(const rto_params* p, uint32_t local);

/* Small functions (known-answer tests). */
uint32_t rto_xorshift32(uint32_t* state);
float rto_unilateral(uint32_t* state);
float rto_bilateral(uint32_t* state);
int   rto_refract(const float incident[3], const float normal[3], float refractive_index, float out[3]);
float rto_schlick(const float incident[3], const float normal[3], float refractive_index);
float rto_linear_to_srgb(float v);
uint32_t rto_pack_srgb(const float rgb[3]);
void rto_pack_srgb_many(const float* rgb, uint64_t count, uint32_t* out);   /* count pixels of 3 floats */
uint32_t rto_pixel_stream_seed(uint32_t seed_constant, uint32_t x, uint32_t y, uint32_t batch);

/* One world intersection.  Returns 1 on hit; t, material, normal as the reference leaves them. */
int rto_intersect(const rto_scene* scene, int intersect_mode, float min_hit, float plane_tol,
                  const float origin[3], const float direction[3],
                  float* t, uint32_t* material_index, float normal[3]);

/* One path (main.cpp:265-334).  Returns the number of world intersections executed. */
uint32_t rto_raytrace_world(const rto_scene* scene, const rto_params* p,
                            const float origin[3], const float direction[3],
                            uint32_t* rng_state, float radiance[3]);

/* Full render of the rows selected by `p`, on `threads` threads pulling row work
 * orders from an atomic queue (main.cpp:336-343, 450-461).  Outputs are compact:
 * local_rows x width.  Any output pointer may be NULL.  Returns 0 on success. */
int rto_render(const rto_scene* scene, const rto_params* p, int threads,
               uint32_t* pixels, float* radiance, uint64_t* rays);

/* Event counts of the last rto_render (summed over its threads) for the algorithmic flop count of a
 * configuration: the terms of F_ray = 17 P + 20 S + 11 S_disc + 18 S_upd + 45 R + 6 R_upd + 62 [hit]
 * + 60 [refracting hit] + 40 / (rays per sample).  Counted by the lane-by-lane and scalar intersection
 * routines (RTO_INTERSECT_WIDE8, _WIDE8_IEEE, _SCALAR); the AVX2 form counts rays and samples only. */
typedef struct {
    uint64_t rays;             /* world intersections (main.cpp:277-278) */
    uint64_t samples;          /* primary rays (main.cpp:379-390) */
    uint64_t hits;             /* intersections that hit something */
    uint64_t refracting_hits;  /* hits on a dielectric whose Refract succeeded (math_util.h:52-79) */
    uint64_t sphere_roots;     /* (ray, sphere) pairs with disc > 0: roots evaluated (main.cpp:73-77) */
    uint64_t sphere_updates;   /* closest-hit updates by a sphere (main.cpp:78-90) */
    uint64_t rectangle_updates;/* closest-hit updates by a rectangle (main.cpp:121-127) */
} rto_statistics;
void rto_last_statistics(rto_statistics* out);

#ifdef __cplusplus
}
#endif
#endif
