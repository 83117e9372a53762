#include "rt_host_capi.h"

#include <cstring>

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

}  // extern "C"
