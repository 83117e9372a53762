/* TEST INFRASTRUCTURE - AVX2 form of the reference's world intersection
 * (main.cpp:26-161 with simd.h / simd_avx2.h): 8 primitives per step, one ray.
 * It is the timed "port" CPU baseline and must agree bit for bit with the
 * lane-by-lane C form in rt_oracle.c (tests check that).
 * Compiled with -ffp-contract=off: the only fused operations are the explicit
 * _mm256_fmadd_ps / _mm256_fmsub_ps, as in the reference.
 */
#include <float.h>
#include <immintrin.h>

#include "rt_oracle_internal.h"

typedef __m256 f8;

static inline f8 bcast(float v) { return _mm256_set1_ps(v); }
static inline f8 dot3(f8 ax, f8 ay, f8 az, f8 bx, f8 by, f8 bz) { /* simd.h:91-93 */
    return _mm256_fmadd_ps(ax, bx, _mm256_fmadd_ps(ay, by, _mm256_mul_ps(az, bz)));
}
static inline f8 dot4(f8 ax, f8 ay, f8 az, f8 aw, f8 bx, f8 by, f8 bz, f8 bw) { /* simd.h:200-202 */
    return _mm256_fmadd_ps(ax, bx, _mm256_fmadd_ps(ay, by, _mm256_fmadd_ps(az, bz, _mm256_mul_ps(aw, bw))));
}
static inline f8 neg(f8 v) { return _mm256_xor_ps(v, bcast(-0.0f)); }
static inline f8 gt(f8 a, f8 b) { return _mm256_cmp_ps(a, b, _CMP_GT_OQ); }
static inline f8 lt(f8 a, f8 b) { return _mm256_cmp_ps(a, b, _CMP_LT_OQ); }
static inline f8 le(f8 a, f8 b) { return _mm256_cmp_ps(a, b, _CMP_LE_OQ); }
static inline f8 ge(f8 a, f8 b) { return _mm256_cmp_ps(a, b, _CMP_GE_OQ); }
static inline f8 and8(f8 a, f8 b) { return _mm256_and_ps(a, b); }
static inline f8 sel(f8 old, f8 mask, f8 val) { return _mm256_blendv_ps(old, val, mask); }
static inline int none(f8 mask) { return _mm256_movemask_ps(mask) == 0; }

int rto_intersect_wide_avx2(const rto_scene* s, float min_hit, float plane_tol, v3 o, v3 d,
                            float* t_out, uint32_t* mat_out, v3* n_out) {
    float closest = FLT_MAX;
    uint32_t hit_mat = 0;
    v3 hit_n = {0, 0, 0};
    int any_hit = 0;

    for (uint32_t i = 0; i < s->plane_count; ++i) {
        const rto_plane* pl = &s->planes[i];
        float denom = pl->normal[0] * d.x + pl->normal[1] * d.y + pl->normal[2] * d.z;
        if ((denom < -plane_tol) || (denom > plane_tol)) {
            float hit = (-pl->d - (pl->normal[0] * o.x + pl->normal[1] * o.y + pl->normal[2] * o.z)) / denom;
            if (hit > min_hit && hit < closest) {
                closest = hit; hit_mat = pl->material_index;
                hit_n.x = pl->normal[0]; hit_n.y = pl->normal[1]; hit_n.z = pl->normal[2];
                any_hit = 1;
            }
        }
    }

    const f8 ox = bcast(o.x), oy = bcast(o.y), oz = bcast(o.z);
    const f8 dx = bcast(d.x), dy = bcast(d.y), dz = bcast(d.z);
    const f8 vmin = bcast(min_hit);
    f8 best_t = bcast(closest), best_m = bcast((float)hit_mat);
    f8 best_nx = bcast(hit_n.x), best_ny = bcast(hit_n.y), best_nz = bcast(hit_n.z);

    for (uint32_t g = 0; g < s->sphere_groups; ++g) {
        const float* blk = s->sphere_soa + (size_t)g * 5 * LANES;
        f8 cx = _mm256_load_ps(blk), cy = _mm256_load_ps(blk + 8), cz = _mm256_load_ps(blk + 16);
        f8 r2 = _mm256_load_ps(blk + 24);
        f8 tx = _mm256_sub_ps(ox, cx), ty = _mm256_sub_ps(oy, cy), tz = _mm256_sub_ps(oz, cz);
        f8 a = dot3(dx, dy, dz, dx, dy, dz);
        f8 b = _mm256_mul_ps(bcast(2.0f), dot3(dx, dy, dz, tx, ty, tz));
        f8 c = _mm256_sub_ps(dot3(tx, ty, tz, tx, ty, tz), r2);
        f8 disc = _mm256_fmsub_ps(b, b, _mm256_mul_ps(_mm256_mul_ps(bcast(4.0f), a), c));
        f8 denom = _mm256_mul_ps(bcast(2.0f), a);
        f8 root_mask = gt(disc, bcast(0.0f));
        if (none(root_mask)) continue;
        f8 root = _mm256_sqrt_ps(disc);
        f8 tp = _mm256_div_ps(_mm256_add_ps(neg(b), root), denom);
        f8 tn = _mm256_div_ps(_mm256_sub_ps(neg(b), root), denom);
        f8 t = sel(tp, and8(gt(tn, vmin), lt(tn, tp)), tn);
        f8 hit = and8(root_mask, and8(gt(t, vmin), lt(t, best_t)));
        if (none(hit)) continue;
        best_t = sel(best_t, hit, t);
        best_m = sel(best_m, hit, _mm256_load_ps(blk + 32));
        f8 vx = _mm256_sub_ps(_mm256_fmadd_ps(dx, t, ox), cx);
        f8 vy = _mm256_sub_ps(_mm256_fmadd_ps(dy, t, oy), cy);
        f8 vz = _mm256_sub_ps(_mm256_fmadd_ps(dz, t, oz), cz);
        f8 f = _mm256_rsqrt_ps(dot3(vx, vy, vz, vx, vy, vz));
        best_nx = sel(best_nx, hit, _mm256_mul_ps(vx, f));
        best_ny = sel(best_ny, hit, _mm256_mul_ps(vy, f));
        best_nz = sel(best_nz, hit, _mm256_mul_ps(vz, f));
        any_hit = 1;
    }

    const f8 one = bcast(1.0f), zero = bcast(0.0f), minus_one = bcast(-1.0f);
    for (uint32_t g = 0; g < s->rectangle_groups; ++g) {
        const float* m = s->rectangle_soa + (size_t)g * 20 * LANES;
#define M(r, c) _mm256_load_ps(m + ((r) * 4 + (c)) * LANES)
        f8 lox = dot4(M(0, 0), M(0, 1), M(0, 2), M(0, 3), ox, oy, oz, one);
        f8 loy = dot4(M(1, 0), M(1, 1), M(1, 2), M(1, 3), ox, oy, oz, one);
        f8 loz = dot4(M(2, 0), M(2, 1), M(2, 2), M(2, 3), ox, oy, oz, one);
        f8 ldx = dot4(M(0, 0), M(0, 1), M(0, 2), M(0, 3), dx, dy, dz, zero);
        f8 ldy = dot4(M(1, 0), M(1, 1), M(1, 2), M(1, 3), dx, dy, dz, zero);
        f8 ldz = dot4(M(2, 0), M(2, 1), M(2, 2), M(2, 3), dx, dy, dz, zero);
#undef M
        f8 t = _mm256_div_ps(neg(loz), ldz);
        f8 hx = _mm256_add_ps(lox, _mm256_mul_ps(ldx, t));
        f8 hy = _mm256_add_ps(loy, _mm256_mul_ps(ldy, t));
        f8 inside = and8(and8(le(hx, one), ge(hx, minus_one)), and8(le(hy, one), ge(hy, minus_one)));
        f8 hit = and8(and8(inside, lt(t, best_t)), gt(t, vmin));
        if (none(hit)) continue;
        best_t = sel(best_t, hit, t);
        best_m = sel(best_m, hit, _mm256_load_ps(m + 19 * LANES));
        f8 nx = _mm256_load_ps(m + 16 * LANES), ny = _mm256_load_ps(m + 17 * LANES), nz = _mm256_load_ps(m + 18 * LANES);
        f8 flip = gt(dot3(nx, ny, nz, dx, dy, dz), zero);
        best_nx = sel(best_nx, hit, sel(nx, flip, neg(nx)));
        best_ny = sel(best_ny, hit, sel(ny, flip, neg(ny)));
        best_nz = sel(best_nz, hit, sel(nz, flip, neg(nz)));
        any_hit = 1;
    }

    if (any_hit) {
        _Alignas(32) float lt_[LANES], lm[LANES], lx[LANES], ly[LANES], lz[LANES];
        _mm256_store_ps(lt_, best_t); _mm256_store_ps(lm, best_m);
        _mm256_store_ps(lx, best_nx); _mm256_store_ps(ly, best_ny); _mm256_store_ps(lz, best_nz);
        for (int j = 0; j < LANES; ++j) {
            if (lt_[j] < closest) {
                closest = lt_[j]; hit_mat = (uint32_t)lm[j];
                hit_n.x = lx[j]; hit_n.y = ly[j]; hit_n.z = lz[j];
            }
        }
    }
    *t_out = closest; *mat_out = hit_mat; *n_out = hit_n;
    return any_hit;
}
