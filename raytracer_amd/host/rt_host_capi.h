/* C exports of the host-side scene and image code (librt_host.so), for callers
 * that are not C++ (the Python tests and bench use them through ctypes). */
#ifndef RT_HOST_CAPI_H
#define RT_HOST_CAPI_H

#include <stdint.h>
#include "../../include/rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_world rth_world;

/* name: "cornell" (CreateCornellBoxScene), "spheres" (createScene), "random_spheres"
 * (count, seed), "glass_spheres" (count = lattice side), or the path of an RTSC file.
 * Returns NULL if the name is unknown or the file unreadable. */
rth_world* rth_world_create(const char* name, uint32_t count, uint64_t seed);
void rth_world_destroy(rth_world* world);
/* Fills *out with pointers into the world's arrays (valid until rth_world_destroy). */
void rth_world_view(const rth_world* world, rt_world* out);
int rth_world_write(const rth_world* world, const char* path);   /* 0 on success */
int rth_write_bmp(const char* path, const uint32_t* pixels, int32_t width, int32_t height); /* 0 on success */

/* Resumable float framebuffer files (rt_framebuffer.h).  The header travels as 8 x uint64:
 * width, height, samples, next_batch, lanes_per_pixel, scene_hash, settings_hash, rays. */
uint64_t rth_world_hash(const rth_world* world);
uint64_t rth_settings_hash(const rt_render_params* params);
int rth_framebuffer_write(const char* path, const uint64_t header[8], const float* sum);       /* 0 on success */
int rth_framebuffer_read_header(const char* path, uint64_t header[8]);                          /* 0 on success */
int rth_framebuffer_read(const char* path, uint64_t header[8], float* sum, uint64_t capacity);  /* capacity in floats */

#ifdef __cplusplus
}
#endif
#endif
