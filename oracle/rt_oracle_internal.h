/* TEST INFRASTRUCTURE - shared between rt_oracle.c and rt_oracle_wide.c. */
#ifndef RT_ORACLE_INTERNAL_H
#define RT_ORACLE_INTERNAL_H

#include "rt_oracle.h"

#define LANES 8 /* simd.h:17 */

typedef struct { float x, y, z; } v3;

struct rto_scene {
    uint32_t material_count, plane_count, sphere_count, rectangle_count;
    rto_material* materials;
    rto_plane* planes;
    rto_sphere* spheres;
    rto_rectangle* rectangles;
    rto_camera camera;
    /* AoSoA mirrors, LANES primitives per group (scene.h:26-30, 49-53), 32-byte aligned. */
    uint32_t sphere_groups, rectangle_groups;
    float* sphere_soa;    /* [group][5][LANES]: x y z r^2 material */
    float* rectangle_soa; /* [group][20][LANES]: m[4][4] normal[3] material */
};

/* rt_oracle_wide.c: main.cpp:26-161 with AVX2 intrinsics over the same AoSoA data. */
int rto_intersect_wide_avx2(const rto_scene* s, float min_hit, float plane_tol, v3 o, v3 d,
                            float* t_out, uint32_t* mat_out, v3* n_out);

#endif
