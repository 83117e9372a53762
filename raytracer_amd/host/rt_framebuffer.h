// Resumable float framebuffer: the running per-pixel radiance SUM of a progressive render plus
// what is needed to continue it (how many samples it holds, which sample batches are used up,
// which scene and settings it belongs to).  The reference's CPU renderer has no such state; its
// GPU front end keeps a running average over frames (win32gpu/compute.glsl:382-391), and this
// is that accumulator made exact (a sum, not an average) and storable.
//
// File layout ("RTFB", little-endian), 64-byte header then height*width*3 float32 sums, rows
// top-down like the image:
//   0  char[4] "RTFB"      4  u32 version (1)     8  u32 width        12 u32 height
//   16 u32 channels (3)    20 u32 samples         24 u32 next_batch   28 u32 lanes_per_pixel
//   32 u64 scene_hash      40 u64 settings_hash   48 u64 rays         56 8 zero bytes
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"

namespace rth {

struct Accumulator {
    uint32_t width = 0, height = 0;
    uint32_t samples = 0;          // samples per pixel summed so far
    uint32_t next_batch = 0;       // first unused sample-batch index (rt_render_params.batch_offset of the next pass)
    uint32_t lanes_per_pixel = 1;  // batches consumed per pass
    uint64_t scene_hash = 0;       // HashWorld of the scene the sums belong to
    uint64_t settings_hash = 0;    // HashSettings of the render parameters that must not change between passes
    uint64_t rays = 0;             // world intersections executed so far
    std::vector<float> sum;        // height*width*3

    void Reset(uint32_t w, uint32_t h);
    // sum += pass (element by element, in pass order: the result does not depend on where a
    // checkpoint was taken); advances samples, next_batch and rays.
    void AddPass(const float* pass_sum, uint32_t pass_samples, uint64_t pass_rays);
    // The counters' part of AddPass, for a sum that was accumulated elsewhere (on the GPU) and copied into `sum`.
    void Advance(uint32_t pass_samples, uint64_t pass_rays);
};

// FNV-1a over the world's arrays and camera / over the parameters that define the estimator
// (image size, bounce cap, tolerances, seed constant, modes, lanes per pixel, samples per pass).
uint64_t HashWorld(const rt_world& world);
uint64_t HashSettings(const rt_render_params& params);

// Writes to "<path>.tmp" and renames, so an interrupted write leaves the previous checkpoint intact.
bool WriteAccumulatorFile(const Accumulator& acc, const char* path);
// Returns false and sets *error on a missing, truncated or malformed file.
bool ReadAccumulatorFile(const char* path, Accumulator* acc, std::string* error);

}  // namespace rth
