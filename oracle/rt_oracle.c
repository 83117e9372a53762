/* TEST INFRASTRUCTURE - CPU restatement of the reference renderer's hot path.
 * See rt_oracle.h for the scope, the arithmetic convention and what pins it.
 * Must be compiled with -ffp-contract=off (oracle/Makefile does).
 */
#define _GNU_SOURCE
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <xmmintrin.h>

#include "rt_oracle_internal.h"

/* Event counters (see rto_statistics): per thread while rendering, folded into `last_statistics` when a
 * worker of rto_render ends. */
static __thread rto_statistics tls_stats;
static rto_statistics last_statistics;
static pthread_mutex_t statistics_mutex = PTHREAD_MUTEX_INITIALIZER;

void rto_last_statistics(rto_statistics* out) {
    pthread_mutex_lock(&statistics_mutex);
    *out = last_statistics;
    pthread_mutex_unlock(&statistics_mutex);
}

/* ---------------------------------------------------------------- vectors */
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float f) { return V(a.x * f, a.y * f, a.z * f); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* math_vector.h:153-155: left to right, each product and sum rounded. */
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* math_vector.h:167-171 */
static inline v3 vnormalize(v3 v) {
    const float dot = vdot(v, v);
    const float factor = 1 / sqrtf(dot);
    return V(v.x * factor, v.y * factor, v.z * factor);
}
static inline v3 from3(const float p[3]) { return V(p[0], p[1], p[2]); }

/* ---------------------------------------------------------------- scene */
rto_scene* rto_scene_create(const rto_material* materials, uint32_t material_count,
                            const rto_plane* planes, uint32_t plane_count,
                            const rto_sphere* spheres, uint32_t sphere_count,
                            const rto_rectangle* rectangles, uint32_t rectangle_count,
                            const rto_camera* camera) {
    if (!materials || material_count == 0 || !camera) return NULL;
    rto_scene* s = (rto_scene*)calloc(1, sizeof *s);
    s->material_count = material_count;
    s->plane_count = plane_count;
    s->sphere_count = sphere_count;
    s->rectangle_count = rectangle_count;
    s->materials = (rto_material*)malloc(sizeof(rto_material) * material_count);
    memcpy(s->materials, materials, sizeof(rto_material) * material_count);
    s->planes = (rto_plane*)malloc(sizeof(rto_plane) * (plane_count + 1));
    if (plane_count) memcpy(s->planes, planes, sizeof(rto_plane) * plane_count);
    s->spheres = (rto_sphere*)malloc(sizeof(rto_sphere) * (sphere_count + 1));
    if (sphere_count) memcpy(s->spheres, spheres, sizeof(rto_sphere) * sphere_count);
    s->rectangles = (rto_rectangle*)malloc(sizeof(rto_rectangle) * (rectangle_count + 1));
    if (rectangle_count) memcpy(s->rectangles, rectangles, sizeof(rto_rectangle) * rectangle_count);
    s->camera = *camera;

    /* scene.h:255-286.  The reference leaves the tail lanes of the last group
     * uninitialised; here they hold primitives that can never be hit. */
    s->sphere_groups = (sphere_count + LANES - 1) / LANES;
    s->sphere_soa = NULL;
    if (s->sphere_groups) {
        if (posix_memalign((void**)&s->sphere_soa, 32, sizeof(float) * 5 * LANES * s->sphere_groups)) return NULL;
        for (uint32_t g = 0; g < s->sphere_groups; ++g) {
            float* blk = s->sphere_soa + (size_t)g * 5 * LANES;
            for (uint32_t j = 0; j < LANES; ++j) {
                uint32_t k = g * LANES + j;
                if (k < sphere_count) {
                    const rto_sphere* sp = &s->spheres[k];
                    blk[0 * LANES + j] = sp->position[0];
                    blk[1 * LANES + j] = sp->position[1];
                    blk[2 * LANES + j] = sp->position[2];
                    blk[3 * LANES + j] = sp->radius * sp->radius;
                    blk[4 * LANES + j] = (float)sp->material_index;
                } else {
                    blk[0 * LANES + j] = blk[1 * LANES + j] = blk[2 * LANES + j] = 1.0e30f;
                    blk[3 * LANES + j] = 0.0f;
                    blk[4 * LANES + j] = 0.0f;
                }
            }
        }
    }
    /* scene.h:413-448 */
    s->rectangle_groups = (rectangle_count + LANES - 1) / LANES;
    s->rectangle_soa = NULL;
    if (s->rectangle_groups) {
        if (posix_memalign((void**)&s->rectangle_soa, 32, sizeof(float) * 20 * LANES * s->rectangle_groups)) return NULL;
        memset(s->rectangle_soa, 0, sizeof(float) * 20 * LANES * s->rectangle_groups);
        for (uint32_t g = 0; g < s->rectangle_groups; ++g) {
            float* blk = s->rectangle_soa + (size_t)g * 20 * LANES;
            for (uint32_t j = 0; j < LANES; ++j) {
                uint32_t k = g * LANES + j;
                if (k >= rectangle_count) continue; /* zero matrix: t = -0/0 = NaN, never hit */
                const rto_rectangle* r = &s->rectangles[k];
                for (int a = 0; a < 4; ++a)
                    for (int b = 0; b < 4; ++b) blk[(a * 4 + b) * LANES + j] = r->m[a][b];
                blk[16 * LANES + j] = r->normal[0];
                blk[17 * LANES + j] = r->normal[1];
                blk[18 * LANES + j] = r->normal[2];
                blk[19 * LANES + j] = (float)r->material_index;
            }
        }
    }
    return s;
}

void rto_scene_destroy(rto_scene* s) {
    if (!s) return;
    free(s->materials); free(s->planes); free(s->spheres); free(s->rectangles);
    free(s->sphere_soa); free(s->rectangle_soa);
    free(s);
}

void rto_default_params(rto_params* p, uint32_t width, uint32_t height, uint32_t spp) {
    memset(p, 0, sizeof *p);
    p->width = width; p->height = height; p->samples_per_pixel = spp;
    p->row_begin = 0; p->row_end = height;
    p->band_rows = 0; p->band_index = 0; p->band_count = 1;
    p->max_bounces = 8;
    p->min_hit_distance = 0.001f;
    p->plane_tolerance = 0.001f;
    p->seed_constant = 262346u;
    p->rng_mode = RTO_RNG_ROW_SERIAL;
    p->rows_per_order = 1;
    p->lanes_per_pixel = 1;
    p->batch_offset = 0;
    p->intersect_mode = RTO_INTERSECT_WIDE8;
    p->output_sum = 0;
}

/* Row selection: rows [row_begin,row_end) are cut into bands of band_rows rows;
 * this render owns the bands whose index is band_index modulo band_count. */
uint32_t rto_local_row_count(const rto_params* p) {
    uint32_t total = p->row_end - p->row_begin;
    if (p->band_rows == 0 || p->band_count <= 1) return total;
    uint32_t n = 0;
    for (uint32_t r = 0; r < total; ++r)
        if ((r / p->band_rows) % p->band_count == p->band_index) ++n;
    return n;
}

uint32_t rto_global_row(const rto_params* p, uint32_t local) {
    if (p->band_rows == 0 || p->band_count <= 1) return p->row_begin + local;
    uint32_t band = local / p->band_rows, within = local % p->band_rows;
    return p->row_begin + (band * p->band_count + p->band_index) * p->band_rows + within;
}

/* ---------------------------------------------------------------- RNG  math_util.h:32-50 */
uint32_t rto_xorshift32(uint32_t* state) {
    uint32_t x = *state;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    *state = x;
    return x;
}

float rto_unilateral(uint32_t* state) {
    return (float)rto_xorshift32(state) / (float)0xFFFFFFFFu; /* (float)U32Max == 2^32 */
}

float rto_bilateral(uint32_t* state) {
    return 2.0f * rto_unilateral(state) - 1.0f;
}

static inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

/* PIXEL mode: one xorshift32 stream per (pixel, sample batch).  The root is the
 * seed the reference gives the one-row WorkOrder [y, y+1) (main.cpp:370); x and
 * the batch index are folded in with a 32-bit finaliser so neighbouring streams
 * are decorrelated; the state is never zero (math_util.h:32). */
uint32_t rto_pixel_stream_seed(uint32_t seed_constant, uint32_t x, uint32_t y, uint32_t batch) {
    uint32_t root = seed_constant * (y + (y + 1u) * 36u);
    uint32_t h = fmix32(root + 0x9E3779B9u * (x + 1u));
    h = fmix32(h ^ (0x85EBCA6Bu * (batch + 1u)));
    return h ? h : 0x9E3779B9u;
}

/* ---------------------------------------------------------------- math_util.h:14-22, 52-93 */
static inline float clampf(float low, float value, float high) {
    if (value < low) return low;
    else if (value > high) return high;
    else return value;
}

static int refract(v3 incident, v3 normal, float refractive_index, v3* out) {
    float cos_i = clampf(-1.0f, vdot(incident, normal), 1.0f);
    v3 hit_normal = normal;
    float ratio;
    if (cos_i < 0) {
        cos_i = -cos_i;
        ratio = (float)(1.0 / refractive_index); /* double division, then narrowed (math_util.h:62) */
    } else {
        hit_normal = vneg(normal);
        ratio = refractive_index;
    }
    float discriminant = 1 - ratio * ratio * (1 - cos_i * cos_i);
    if (discriminant < 0) return 0;
    *out = vadd(vscale(incident, ratio), vscale(hit_normal, (ratio * cos_i - sqrtf(discriminant))));
    return 1;
}

static float schlick(v3 incident, v3 normal, float refractive_index) {
    float cos_i = clampf(-1.0f, vdot(incident, normal), 1.0f);
    float cosine = cos_i > 0 ? cos_i : -cos_i;
    float r0 = (1 - refractive_index) / (1 + refractive_index);
    r0 = r0 * r0;
    return r0 + (1 - r0) * powf((1 - cosine), 5);
}

int rto_refract(const float incident[3], const float normal[3], float refractive_index, float out[3]) {
    v3 o = V(out[0], out[1], out[2]);
    int ok = refract(from3(incident), from3(normal), refractive_index, &o);
    out[0] = o.x; out[1] = o.y; out[2] = o.z;
    return ok;
}

float rto_schlick(const float incident[3], const float normal[3], float refractive_index) {
    return schlick(from3(incident), from3(normal), refractive_index);
}

/* ---------------------------------------------------------------- math_util.h:102-125 */
float rto_linear_to_srgb(float value) {
    value = clampf(0.0f, value, 1.0f);
    float result = value * 12.92f;
    if (value >= 0.0031308f) result = (1.055f * powf(value, 1.0f / 2.4f)) - 0.055f;
    return result;
}

uint32_t rto_pack_srgb(const float rgb[3]) {
    return (255u << 24 | (uint32_t)((int32_t)(255 * rto_linear_to_srgb(rgb[0])) << 16) |
            (uint32_t)((int32_t)(255 * rto_linear_to_srgb(rgb[1])) << 8) |
            (uint32_t)((int32_t)(255 * rto_linear_to_srgb(rgb[2])) << 0));
}

void rto_pack_srgb_many(const float* rgb, uint64_t count, uint32_t* out) {
    for (uint64_t i = 0; i < count; ++i) out[i] = rto_pack_srgb(rgb + 3 * i);
}

/* ---------------------------------------------------------------- main.cpp:163-242 */
static int intersect_scalar(const rto_scene* s, float min_hit, float plane_tol, v3 o, v3 d,
                            float* t_out, uint32_t* mat_out, v3* n_out) {
    float best_t = FLT_MAX;
    uint32_t best_mat = 0;
    v3 best_n = V(0, 0, 0);

    for (uint32_t i = 0; i < s->plane_count; ++i) {
        const rto_plane* pl = &s->planes[i];
        v3 n = from3(pl->normal);
        float denom = vdot(n, d);
        if ((denom < -plane_tol) || (denom > plane_tol)) {
            float hit = (-pl->d - vdot(n, o)) / denom;
            if (hit > min_hit && hit < best_t) { best_t = hit; best_mat = pl->material_index; best_n = n; }
        }
    }
    for (uint32_t i = 0; i < s->sphere_count; ++i) {
        const rto_sphere* sp = &s->spheres[i];
        v3 c = from3(sp->position);
        v3 cto = vsub(o, c);
        float a = vdot(d, d);
        float b = 2 * vdot(d, cto);
        float cc = vdot(cto, cto) - (sp->radius * sp->radius);
        float disc = b * b - 4 * a * cc;
        float denom = 2 * a;
        if (disc > 0) {
            ++tls_stats.sphere_roots;
            float tp = (-b + sqrtf(disc)) / denom;
            float tn = (-b - sqrtf(disc)) / denom;
            float hit = tp;
            if (tn > min_hit && tn < tp) hit = tn;
            if (hit > min_hit && hit < best_t) {
                ++tls_stats.sphere_updates;
                best_t = hit;
                best_mat = sp->material_index;
                v3 hp = vadd(o, vscale(d, hit));
                best_n = vnormalize(vsub(hp, c));
            }
        }
    }
    for (uint32_t i = 0; i < s->rectangle_count; ++i) {
        const rto_rectangle* r = &s->rectangles[i];
        /* Matrix4 * Vector4 (math_matrix.h:47-55) with the 4-wide dot of math_vector.h:341-343 */
        float lo[3], ld[3];
        for (int row = 0; row < 3; ++row) {
            lo[row] = r->m[row][0] * o.x + r->m[row][1] * o.y + r->m[row][2] * o.z + r->m[row][3] * 1.0f;
            ld[row] = r->m[row][0] * d.x + r->m[row][1] * d.y + r->m[row][2] * d.z + r->m[row][3] * 0.0f;
        }
        float t = (-lo[2]) / ld[2];
        float hx = lo[0] + ld[0] * t;
        float hy = lo[1] + ld[1] * t;
        int hit = hx <= 1.0f && hx >= -1.0f && hy <= 1.0f && hy >= -1.0f;
        if (hit && t < best_t && t > min_hit) {
            ++tls_stats.rectangle_updates;
            best_t = t;
            best_mat = r->material_index;
            v3 n = from3(r->normal);
            float dot = vdot(n, d);
            best_n = dot > 0 ? vneg(n) : n;
        }
    }
    *t_out = best_t; *mat_out = best_mat; *n_out = best_n;
    return best_t < FLT_MAX;
}

/* ---------------------------------------------------------------- main.cpp:26-161
 * The 8 lanes of the reference's LaneF32 are written out as arrays; FMulAdd /
 * FMulSub become fmaf; compare+blend becomes a per-lane conditional.  Lane
 * DotProduct is fma(x,x', fma(y,y', z*z')) (simd.h:91-93) and the 4-wide one
 * fma(x,x', fma(y,y', fma(z,z', w*w'))) (simd.h:200-202).
 */
static inline float lane_dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return fmaf(ax, bx, fmaf(ay, by, az * bz));
}
static inline float lane_dot4(float ax, float ay, float az, float aw, float bx, float by, float bz, float bw) {
    return fmaf(ax, bx, fmaf(ay, by, fmaf(az, bz, aw * bw)));
}
static inline float hw_rsqrt(float v) { /* _mm256_rsqrt_ps, one lane (simd_avx2.h:139-144) */
    return _mm_cvtss_f32(_mm_rsqrt_ss(_mm_set_ss(v)));
}

static int intersect_wide(const rto_scene* s, int ieee_rsqrt, float min_hit, float plane_tol, v3 o, v3 d,
                          float* t_out, uint32_t* mat_out, v3* n_out) {
    float closest = FLT_MAX;
    uint32_t hit_mat = 0;
    v3 hit_n = V(0, 0, 0);
    int any_hit = 0;

    for (uint32_t i = 0; i < s->plane_count; ++i) {
        const rto_plane* pl = &s->planes[i];
        v3 n = from3(pl->normal);
        float denom = vdot(n, d);
        if ((denom < -plane_tol) || (denom > plane_tol)) {
            float hit = (-pl->d - vdot(n, o)) / denom;
            if (hit > min_hit && hit < closest) { closest = hit; hit_mat = pl->material_index; hit_n = n; any_hit = 1; }
        }
    }

    float lane_t[LANES], lane_mat[LANES], lane_nx[LANES], lane_ny[LANES], lane_nz[LANES];
    for (int j = 0; j < LANES; ++j) {
        lane_t[j] = closest; lane_mat[j] = (float)hit_mat;
        lane_nx[j] = hit_n.x; lane_ny[j] = hit_n.y; lane_nz[j] = hit_n.z;
    }

    for (uint32_t g = 0; g < s->sphere_groups; ++g) {
        const float* blk = s->sphere_soa + (size_t)g * 5 * LANES;
        for (int j = 0; j < LANES; ++j) {
            float cx = blk[0 * LANES + j], cy = blk[1 * LANES + j], cz = blk[2 * LANES + j];
            float r2 = blk[3 * LANES + j];
            float tx = o.x - cx, ty = o.y - cy, tz = o.z - cz;
            float a = lane_dot3(d.x, d.y, d.z, d.x, d.y, d.z);
            float b = 2.0f * lane_dot3(d.x, d.y, d.z, tx, ty, tz);
            float c = lane_dot3(tx, ty, tz, tx, ty, tz) - r2;
            float disc = fmaf(b, b, -(4.0f * a * c));
            float denom = 2.0f * a;
            if (!(disc > 0.0f)) continue;
            ++tls_stats.sphere_roots;
            float root = sqrtf(disc);
            float tp = (-b + root) / denom;
            float tn = (-b - root) / denom;
            float hit = tp;
            if (tn > min_hit && tn < tp) hit = tn;
            if (hit > min_hit && hit < lane_t[j]) {
                ++tls_stats.sphere_updates;
                lane_t[j] = hit;
                lane_mat[j] = blk[4 * LANES + j];
                float hx = fmaf(d.x, hit, o.x), hy = fmaf(d.y, hit, o.y), hz = fmaf(d.z, hit, o.z);
                float vx = hx - cx, vy = hy - cy, vz = hz - cz;
                float len2 = lane_dot3(vx, vy, vz, vx, vy, vz);
                float f = ieee_rsqrt ? 1.0f / sqrtf(len2) : hw_rsqrt(len2);
                lane_nx[j] = vx * f; lane_ny[j] = vy * f; lane_nz[j] = vz * f;
                any_hit = 1;
            }
        }
    }

    for (uint32_t g = 0; g < s->rectangle_groups; ++g) {
        const float* blk = s->rectangle_soa + (size_t)g * 20 * LANES;
        for (int j = 0; j < LANES; ++j) {
            float lo[3], ld[3];
            for (int row = 0; row < 3; ++row) {
                float m0 = blk[(row * 4 + 0) * LANES + j], m1 = blk[(row * 4 + 1) * LANES + j];
                float m2 = blk[(row * 4 + 2) * LANES + j], m3 = blk[(row * 4 + 3) * LANES + j];
                lo[row] = lane_dot4(m0, m1, m2, m3, o.x, o.y, o.z, 1.0f);
                ld[row] = lane_dot4(m0, m1, m2, m3, d.x, d.y, d.z, 0.0f);
            }
            float t = (-lo[2]) / ld[2];
            float hx = lo[0] + ld[0] * t;
            float hy = lo[1] + ld[1] * t;
            int hit = hx <= 1.0f && hx >= -1.0f && hy <= 1.0f && hy >= -1.0f;
            if (hit && t < lane_t[j] && t > min_hit) {
                ++tls_stats.rectangle_updates;
                lane_t[j] = t;
                lane_mat[j] = blk[19 * LANES + j];
                float nx = blk[16 * LANES + j], ny = blk[17 * LANES + j], nz = blk[18 * LANES + j];
                float dot = lane_dot3(nx, ny, nz, d.x, d.y, d.z);
                if (dot > 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
                lane_nx[j] = nx; lane_ny[j] = ny; lane_nz[j] = nz;
                any_hit = 1;
            }
        }
    }

    if (any_hit) { /* main.cpp:131-154: first strictly smaller lane wins; the plane wins ties */
        for (int j = 0; j < LANES; ++j) {
            float t = lane_t[j];
            if (t < closest) {
                closest = t;
                hit_mat = (uint32_t)lane_mat[j];
                hit_n = V(lane_nx[j], lane_ny[j], lane_nz[j]);
            }
        }
    }
    *t_out = closest; *mat_out = hit_mat; *n_out = hit_n;
    return any_hit;
}

/* The same routine for any lane width (simd.h:17 fixes LANE_WIDTH at 8; its 4-wide branch, simd_sse.h, has
 * the same operations: fused FMulAdd / FMulSub under __FMA__, _mm_rsqrt_ps).  Primitive k lives in lane
 * k % L of group k / L (scene.h:255-286, 414-447); each lane keeps its own running closest hit and the
 * first strictly smaller lane wins at the end (main.cpp:131-154).  With L = 8 this is intersect_wide
 * reading the records in place (a test holds the two equal); L = 4 restates a LANE_WIDTH 4 build, which
 * cannot be produced from the reference without editing simd.h: PARITY UNPINNED for ties between lanes. */
static int intersect_wide_lanes(const rto_scene* s, uint32_t L, int ieee_rsqrt, float min_hit, float plane_tol, v3 o, v3 d,
                                float* t_out, uint32_t* mat_out, v3* n_out) {
    float closest = FLT_MAX;
    uint32_t hit_mat = 0;
    v3 hit_n = V(0, 0, 0);
    int any_hit = 0;
    for (uint32_t i = 0; i < s->plane_count; ++i) {
        const rto_plane* pl = &s->planes[i];
        v3 n = from3(pl->normal);
        float denom = vdot(n, d);
        if ((denom < -plane_tol) || (denom > plane_tol)) {
            float hit = (-pl->d - vdot(n, o)) / denom;
            if (hit > min_hit && hit < closest) { closest = hit; hit_mat = pl->material_index; hit_n = n; any_hit = 1; }
        }
    }
    float lane_t[LANES], lane_mat[LANES], lane_nx[LANES], lane_ny[LANES], lane_nz[LANES];
    for (uint32_t j = 0; j < L; ++j) {
        lane_t[j] = closest; lane_mat[j] = (float)hit_mat;
        lane_nx[j] = hit_n.x; lane_ny[j] = hit_n.y; lane_nz[j] = hit_n.z;
    }
    for (uint32_t k = 0; k < s->sphere_count; ++k) {      /* group k / L, lane k % L: groups in order, lanes independent */
        const rto_sphere* sp = &s->spheres[k];
        const uint32_t j = k % L;
        float cx = sp->position[0], cy = sp->position[1], cz = sp->position[2];
        float r2 = sp->radius * sp->radius;
        float tx = o.x - cx, ty = o.y - cy, tz = o.z - cz;
        float a = lane_dot3(d.x, d.y, d.z, d.x, d.y, d.z);
        float b = 2.0f * lane_dot3(d.x, d.y, d.z, tx, ty, tz);
        float c = lane_dot3(tx, ty, tz, tx, ty, tz) - r2;
        float disc = fmaf(b, b, -(4.0f * a * c));
        float denom = 2.0f * a;
        if (!(disc > 0.0f)) continue;
        float root = sqrtf(disc);
        float tp = (-b + root) / denom;
        float tn = (-b - root) / denom;
        float hit = tp;
        if (tn > min_hit && tn < tp) hit = tn;
        if (hit > min_hit && hit < lane_t[j]) {
            lane_t[j] = hit;
            lane_mat[j] = (float)sp->material_index;
            float hx = fmaf(d.x, hit, o.x), hy = fmaf(d.y, hit, o.y), hz = fmaf(d.z, hit, o.z);
            float vx = hx - cx, vy = hy - cy, vz = hz - cz;
            float len2 = lane_dot3(vx, vy, vz, vx, vy, vz);
            float f = ieee_rsqrt ? 1.0f / sqrtf(len2) : hw_rsqrt(len2);
            lane_nx[j] = vx * f; lane_ny[j] = vy * f; lane_nz[j] = vz * f;
            any_hit = 1;
        }
    }
    for (uint32_t k = 0; k < s->rectangle_count; ++k) {
        const rto_rectangle* r = &s->rectangles[k];
        const uint32_t j = k % L;
        float lo[3], ld[3];
        for (int row = 0; row < 3; ++row) {
            lo[row] = lane_dot4(r->m[row][0], r->m[row][1], r->m[row][2], r->m[row][3], o.x, o.y, o.z, 1.0f);
            ld[row] = lane_dot4(r->m[row][0], r->m[row][1], r->m[row][2], r->m[row][3], d.x, d.y, d.z, 0.0f);
        }
        float t = (-lo[2]) / ld[2];
        float hx = lo[0] + ld[0] * t;
        float hy = lo[1] + ld[1] * t;
        int hit = hx <= 1.0f && hx >= -1.0f && hy <= 1.0f && hy >= -1.0f;
        if (hit && t < lane_t[j] && t > min_hit) {
            lane_t[j] = t;
            lane_mat[j] = (float)r->material_index;
            float nx = r->normal[0], ny = r->normal[1], nz = r->normal[2];
            float dot = lane_dot3(nx, ny, nz, d.x, d.y, d.z);
            if (dot > 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
            lane_nx[j] = nx; lane_ny[j] = ny; lane_nz[j] = nz;
            any_hit = 1;
        }
    }
    if (any_hit) {
        for (uint32_t j = 0; j < L; ++j) {
            float t = lane_t[j];
            if (t < closest) {
                closest = t;
                hit_mat = (uint32_t)lane_mat[j];
                hit_n = V(lane_nx[j], lane_ny[j], lane_nz[j]);
            }
        }
    }
    *t_out = closest; *mat_out = hit_mat; *n_out = hit_n;
    return any_hit;
}

static inline int intersect(const rto_scene* s, int mode, float min_hit, float plane_tol, v3 o, v3 d,
                            float* t, uint32_t* mat, v3* n) {
    if (mode == RTO_INTERSECT_SCALAR) return intersect_scalar(s, min_hit, plane_tol, o, d, t, mat, n);
    if (mode == RTO_INTERSECT_WIDE8_AVX2) return rto_intersect_wide_avx2(s, min_hit, plane_tol, o, d, t, mat, n);
    if (mode == RTO_INTERSECT_WIDE4 || mode == RTO_INTERSECT_WIDE4_IEEE)
        return intersect_wide_lanes(s, 4, mode == RTO_INTERSECT_WIDE4_IEEE, min_hit, plane_tol, o, d, t, mat, n);
    if (mode == RTO_INTERSECT_WIDE8_LANES) return intersect_wide_lanes(s, 8, 0, min_hit, plane_tol, o, d, t, mat, n);
    return intersect_wide(s, mode == RTO_INTERSECT_WIDE8_IEEE, min_hit, plane_tol, o, d, t, mat, n);
}

int rto_intersect(const rto_scene* scene, int intersect_mode, float min_hit, float plane_tol,
                  const float origin[3], const float direction[3],
                  float* t, uint32_t* material_index, float normal[3]) {
    v3 n = V(0, 0, 0);
    int hit = intersect(scene, intersect_mode, min_hit, plane_tol, from3(origin), from3(direction), t, material_index, &n);
    normal[0] = n.x; normal[1] = n.y; normal[2] = n.z;
    return hit;
}

/* ---------------------------------------------------------------- main.cpp:265-334 */
static uint32_t raytrace_world(const rto_scene* s, const rto_params* p, v3 origin, v3 direction,
                               uint32_t* state, v3* result_out) {
    v3 result = V(0.0f, 0.0f, 0.0f);
    v3 ro = origin, rd = direction;
    uint32_t bounces = 0;
    v3 attenuation = V(1.0f, 1.0f, 1.0f);
    for (uint32_t bounce = 0; bounce < p->max_bounces; ++bounce) {
        float t; uint32_t mat_index; v3 n;
        int hit = intersect(s, (int)p->intersect_mode, p->min_hit_distance, p->plane_tolerance, ro, rd, &t, &mat_index, &n);
        ++bounces;
        ++tls_stats.rays;
        const rto_material* mat = &s->materials[mat_index];
        v3 emit = from3(mat->emit_color);
        if (hit) {
            ++tls_stats.hits;
            result = vadd(result, vmul(attenuation, emit));
            attenuation = vmul(attenuation, from3(mat->color));
            ro = vadd(ro, vscale(rd, t));

            v3 mirror = vsub(rd, vscale(vscale(n, vdot(n, rd)), 2.0f));
            /* g++ evaluates the three constructor arguments right to left: z, y, x. */
            float rz = rto_bilateral(state);
            float ry = rto_bilateral(state);
            float rx = rto_bilateral(state);
            v3 random_bounce = vadd(n, V(rx, ry, rz));
            /* Lerp(left, f, right) = left*(1-f) + right*f  (math_util.h:24-26) */
            float f = mat->reflection;
            v3 reflected = vnormalize(vadd(vscale(random_bounce, (1.0f - f)), vscale(mirror, f)));

            float fresnel = 1.0f;
            v3 refracted = reflected;
            if (mat->refractive_index != 0.0f) {
                if (refract(rd, n, mat->refractive_index, &refracted)) {
                    ++tls_stats.refracting_hits;
                    refracted = vnormalize(refracted);
                    fresnel = schlick(rd, n, mat->refractive_index);
                }
            }
            if (rto_unilateral(state) <= fresnel) rd = reflected;
            else rd = refracted;
        } else {
            result = vadd(result, vmul(attenuation, emit));
            break;
        }
    }
    *result_out = result;
    return bounces;
}

uint32_t rto_raytrace_world(const rto_scene* scene, const rto_params* p,
                            const float origin[3], const float direction[3],
                            uint32_t* rng_state, float radiance[3]) {
    v3 r;
    uint32_t b = raytrace_world(scene, p, from3(origin), from3(direction), rng_state, &r);
    radiance[0] = r.x; radiance[1] = r.y; radiance[2] = r.z;
    return b;
}

/* ---------------------------------------------------------------- main.cpp:336-400 */
typedef struct {
    v3 cam_pos, cam_x, cam_y, film_center;
    float half_w, half_h, pixel_w, pixel_h;
} film_t;

static film_t make_film(const rto_scene* s, const rto_params* p) {
    film_t f;
    float aspect = (float)p->width / (float)p->height;
    f.cam_pos = from3(s->camera.position);
    v3 cam_z = from3(s->camera.z_axis);
    f.cam_x = from3(s->camera.x_axis);
    f.cam_y = from3(s->camera.y_axis);
    float film_distance = 1.0f;
    f.film_center = vsub(f.cam_pos, vscale(cam_z, film_distance));
    float film_w = 1.0f * aspect, film_h = 1.0f;
    f.half_w = film_w * 0.5f;
    f.half_h = film_h * 0.5f;
    f.pixel_w = 0.5f / (float)(int32_t)p->width;
    f.pixel_h = 0.5f / (float)(int32_t)p->height;
    return f;
}

static inline float film_y_of(const rto_params* p, uint32_t y) { return ((float)(int32_t)y / (float)(int32_t)p->height) * -2.0f + 1.0f; }
static inline float film_x_of(const rto_params* p, uint32_t x) { return (((float)(int32_t)x / (float)(int32_t)p->width) * 2.0f - 1.0f); }

/* One sample: two draws for the jitter, then the path (main.cpp:381-390). */
static inline uint32_t trace_sample(const rto_scene* s, const rto_params* p, const film_t* f,
                                    float film_x, float film_y, uint32_t* state, v3* color) {
    float off_x = film_x + rto_bilateral(state) * f->pixel_w;
    float off_y = film_y + rto_bilateral(state) * f->pixel_h;
    v3 film_pos = vadd(vadd(f->film_center, vscale(vscale(f->cam_x, off_x), f->half_w)),
                       vscale(vscale(f->cam_y, f->half_h), off_y));
    v3 dir = vnormalize(vsub(film_pos, f->cam_pos));
    ++tls_stats.samples;
    v3 c;
    uint32_t b = raytrace_world(s, p, f->cam_pos, dir, state, &c);
    *color = vadd(*color, c);
    return b;
}

static inline void emit_pixel(const rto_params* p, v3 color, size_t index, uint32_t* pixels, float* radiance) {
    float n = (float)p->samples_per_pixel;
    v3 mean = V(color.x / n, color.y / n, color.z / n);
    if (pixels) { float rgb[3] = {mean.x, mean.y, mean.z}; pixels[index] = rto_pack_srgb(rgb); }
    if (radiance) {
        v3 out = p->output_sum ? color : mean;
        radiance[3 * index + 0] = out.x; radiance[3 * index + 1] = out.y; radiance[3 * index + 2] = out.z;
    }
}

/* Work order = `rows_per_order` consecutive local rows; one xorshift32 stream
 * per order, consumed serially across x, samples and bounces (main.cpp:370-392). */
static uint64_t render_order_row_serial(const rto_scene* s, const rto_params* p, const film_t* f,
                                        uint32_t local_begin, uint32_t local_end,
                                        uint32_t* pixels, float* radiance) {
    uint32_t start_row = rto_global_row(p, local_begin);
    uint32_t end_row = rto_global_row(p, local_end - 1) + 1;
    uint32_t state = p->seed_constant * (start_row + end_row * 36u);
    uint64_t bounces = 0;
    for (uint32_t ly = local_begin; ly < local_end; ++ly) {
        uint32_t y = rto_global_row(p, ly);
        float film_y = film_y_of(p, y);
        for (uint32_t x = 0; x < p->width; ++x) {
            float film_x = film_x_of(p, x);
            v3 color = V(0.0f, 0.0f, 0.0f);
            for (uint32_t sidx = 0; sidx < p->samples_per_pixel; ++sidx)
                bounces += trace_sample(s, p, f, film_x, film_y, &state, &color);
            emit_pixel(p, color, (size_t)ly * p->width + x, pixels, radiance);
        }
    }
    return bounces;
}

/* PIXEL mode: lanes_per_pixel streams per pixel, each tracing spp/lanes_per_pixel
 * consecutive samples; the per-stream sums are combined by a pairwise tree
 * (stride 1, 2, 4, ...), which for one stream is the reference's serial sum. */
static uint64_t render_order_pixel(const rto_scene* s, const rto_params* p, const film_t* f,
                                   uint32_t local_begin, uint32_t local_end,
                                   uint32_t* pixels, float* radiance) {
    uint32_t B = p->lanes_per_pixel;
    uint32_t K = p->samples_per_pixel / B;
    uint64_t bounces = 0;
    v3 partial[64];
    for (uint32_t ly = local_begin; ly < local_end; ++ly) {
        uint32_t y = rto_global_row(p, ly);
        float film_y = film_y_of(p, y);
        for (uint32_t x = 0; x < p->width; ++x) {
            float film_x = film_x_of(p, x);
            for (uint32_t b = 0; b < B; ++b) {
                uint32_t state = rto_pixel_stream_seed(p->seed_constant, x, y, p->batch_offset + b);
                v3 color = V(0.0f, 0.0f, 0.0f);
                for (uint32_t k = 0; k < K; ++k) bounces += trace_sample(s, p, f, film_x, film_y, &state, &color);
                partial[b] = color;
            }
            for (uint32_t stride = 1; stride < B; stride <<= 1)
                for (uint32_t b = 0; b + stride < B; b += 2 * stride) partial[b] = vadd(partial[b], partial[b + stride]);
            emit_pixel(p, partial[0], (size_t)ly * p->width + x, pixels, radiance);
        }
    }
    return bounces;
}

typedef struct {
    const rto_scene* scene;
    const rto_params* params;
    film_t film;
    uint32_t local_rows, rows_per_order, order_count;
    uint32_t* pixels;
    float* radiance;
    volatile uint64_t next_order;
    volatile uint64_t total_bounces;
} work_queue;

static int raytrace_work(work_queue* q) { /* main.cpp:336-343, 397-399 */
    uint64_t order = __sync_fetch_and_add(&q->next_order, 1);
    if (order >= q->order_count) return 0;
    uint32_t lb = (uint32_t)order * q->rows_per_order;
    uint32_t le = lb + q->rows_per_order < q->local_rows ? lb + q->rows_per_order : q->local_rows;
    uint64_t bounces;
    if (q->params->rng_mode == RTO_RNG_PIXEL)
        bounces = render_order_pixel(q->scene, q->params, &q->film, lb, le, q->pixels, q->radiance);
    else
        bounces = render_order_row_serial(q->scene, q->params, &q->film, lb, le, q->pixels, q->radiance);
    __sync_fetch_and_add(&q->total_bounces, bounces);
    return 1;
}

static void* thread_proc(void* arg) {
    memset(&tls_stats, 0, sizeof tls_stats);
    while (raytrace_work((work_queue*)arg)) {}
    pthread_mutex_lock(&statistics_mutex);
    last_statistics.rays += tls_stats.rays;
    last_statistics.samples += tls_stats.samples;
    last_statistics.hits += tls_stats.hits;
    last_statistics.refracting_hits += tls_stats.refracting_hits;
    last_statistics.sphere_roots += tls_stats.sphere_roots;
    last_statistics.sphere_updates += tls_stats.sphere_updates;
    last_statistics.rectangle_updates += tls_stats.rectangle_updates;
    pthread_mutex_unlock(&statistics_mutex);
    return NULL;
}

int rto_render(const rto_scene* scene, const rto_params* p, int threads,
               uint32_t* pixels, float* radiance, uint64_t* rays) {
    if (!scene || !p || p->width == 0 || p->height == 0 || p->samples_per_pixel == 0) return 1;
    if (p->row_end > p->height || p->row_begin >= p->row_end) return 1;
    if (p->rng_mode == RTO_RNG_PIXEL) {
        uint32_t B = p->lanes_per_pixel;
        if (B == 0 || B > 64 || (B & (B - 1)) || p->samples_per_pixel % B) return 1;
    }
    work_queue q;
    memset(&q, 0, sizeof q);
    q.scene = scene; q.params = p; q.film = make_film(scene, p);
    q.local_rows = rto_local_row_count(p);
    q.rows_per_order = p->rows_per_order ? p->rows_per_order : q.local_rows;
    if (p->rng_mode == RTO_RNG_PIXEL) q.rows_per_order = 1;
    if (p->rng_mode == RTO_RNG_ROW_SERIAL && p->band_rows && p->band_count > 1 && p->band_rows % q.rows_per_order) return 1;
    q.order_count = (q.local_rows + q.rows_per_order - 1) / q.rows_per_order;
    q.pixels = pixels; q.radiance = radiance;

    pthread_mutex_lock(&statistics_mutex);
    memset(&last_statistics, 0, sizeof last_statistics);
    pthread_mutex_unlock(&statistics_mutex);
    if (threads < 1) threads = 1;
    pthread_t* pool = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    for (int i = 1; i < threads; ++i) pthread_create(&pool[i], NULL, thread_proc, &q);
    thread_proc(&q); /* the calling thread takes orders too (main.cpp:456-461) */
    for (int i = 1; i < threads; ++i) pthread_join(pool[i], NULL);
    free(pool);
    if (rays) *rays = q.total_bounces;
    return 0;
}
