#include "rt_host_capi.h"

#include <cstring>

#include "rt_framebuffer.h"
#include "rt_image.h"
#include "rt_scene.h"

struct rth_world { rth::World* world; };

extern "C" {

rth_world* rth_world_create(const char* name, uint32_t count, uint64_t seed) {
    if (!name) return nullptr;
    rth::World* w = nullptr;
    if (!std::strcmp(name, "cornell")) w = rth::CreateCornellBoxScene();
    else if (!std::strcmp(name, "spheres")) w = rth::createScene();
    else if (!std::strcmp(name, "random_spheres")) w = rth::CreateRandomSpheresScene(count ? count : 4096, seed);
    else if (!std::strcmp(name, "glass_spheres")) w = rth::CreateGlassSpheresScene(count ? count : 5);
    else w = rth::ReadSceneFile(name);
    if (!w) return nullptr;
    return new rth_world{w};
}

void rth_world_destroy(rth_world* world) {
    if (!world) return;
    delete world->world;
    delete world;
}

void rth_world_view(const rth_world* world, rt_world* out) { *out = world->world->View(); }

int rth_world_write(const rth_world* world, const char* path) { return rth::WriteSceneFile(*world->world, path) ? 0 : 1; }

int rth_write_bmp(const char* path, const uint32_t* pixels, int32_t width, int32_t height) {
    rth::Image image;
    image.width = width;
    image.height = height;
    image.pixelData = const_cast<uint32_t*>(pixels);
    return rth::WriteImageFile(&image, path) ? 0 : 1;
}

uint64_t rth_world_hash(const rth_world* world) { return rth::HashWorld(world->world->View()); }

uint64_t rth_settings_hash(const rt_render_params* params) { return rth::HashSettings(*params); }

namespace {
void to_header(const rth::Accumulator& a, uint64_t h[8]) {
    h[0] = a.width; h[1] = a.height; h[2] = a.samples; h[3] = a.next_batch; h[4] = a.lanes_per_pixel;
    h[5] = a.scene_hash; h[6] = a.settings_hash; h[7] = a.rays;
}
}  // namespace

int rth_framebuffer_write(const char* path, const uint64_t h[8], const float* sum) {
    rth::Accumulator a;
    a.width = (uint32_t)h[0]; a.height = (uint32_t)h[1]; a.samples = (uint32_t)h[2]; a.next_batch = (uint32_t)h[3];
    a.lanes_per_pixel = (uint32_t)h[4]; a.scene_hash = h[5]; a.settings_hash = h[6]; a.rays = h[7];
    a.sum.assign(sum, sum + (size_t)a.width * a.height * 3);
    return rth::WriteAccumulatorFile(a, path) ? 0 : 1;
}

int rth_framebuffer_read(const char* path, uint64_t h[8], float* sum, uint64_t capacity) {
    rth::Accumulator a;
    if (!rth::ReadAccumulatorFile(path, &a, nullptr)) return 1;
    to_header(a, h);
    if (sum) {
        if (capacity < a.sum.size()) return 2;
        std::memcpy(sum, a.sum.data(), a.sum.size() * sizeof(float));
    }
    return 0;
}

int rth_framebuffer_read_header(const char* path, uint64_t h[8]) { return rth_framebuffer_read(path, h, nullptr, 0); }

}  // extern "C"
