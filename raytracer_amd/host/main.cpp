// Render driver: the counterpart of the reference's main() (main.cpp:409-475).
// It keeps that program's shape - create the image, build the world, render,
// print time / ray count / Mray/s, write render.bmp - but the work-order queue
// and thread pool (main.cpp:432-461) are replaced by one call into the HIP
// library per GPU, and the sizes the reference hard-codes are options.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_image.h"
#include "rt_scene.h"

using namespace rth;

namespace {

struct Options {
    std::string scene = "cornell";
    int width = 1280, height = 720;          // main.cpp:410
    unsigned spp = 512;                      // main.cpp:415
    unsigned bounces = 8;                    // main.cpp:275
    unsigned seed_constant = 262346;         // main.cpp:370
    std::string rng = "pixel";
    unsigned lanes_per_pixel = 1;
    std::string intersect = "wide";
    unsigned count = 0;
    unsigned long long scene_seed = 1;
    int gpus = 1;
    std::string out = "render.bmp";
};

void Usage() {
    std::fprintf(stderr,
                 "usage: raytracer [--scene cornell|spheres|random_spheres|glass_spheres|FILE.rtsc] [--count N] [--scene-seed S]\n"
                 "                 [--width W] [--height H] [--spp N] [--bounces N] [--seed-constant N]\n"
                 "                 [--rng pixel|row] [--lanes-per-pixel N] [--intersect wide|scalar]\n"
                 "                 [--gpus N] [--out FILE.bmp]\n");
}

bool Parse(int argc, char** argv, Options* o) {
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto value = [&]() -> const char* { return i + 1 < argc ? argv[++i] : nullptr; };
        const char* v = nullptr;
        if (a == "--help" || a == "-h") return false;
        if (!(v = value())) return false;
        if (a == "--scene") o->scene = v;
        else if (a == "--count") o->count = (unsigned)std::strtoul(v, nullptr, 0);
        else if (a == "--scene-seed") o->scene_seed = std::strtoull(v, nullptr, 0);
        else if (a == "--width") o->width = std::atoi(v);
        else if (a == "--height") o->height = std::atoi(v);
        else if (a == "--spp") o->spp = (unsigned)std::strtoul(v, nullptr, 0);
        else if (a == "--bounces") o->bounces = (unsigned)std::strtoul(v, nullptr, 0);
        else if (a == "--seed-constant") o->seed_constant = (unsigned)std::strtoul(v, nullptr, 0);
        else if (a == "--rng") o->rng = v;
        else if (a == "--lanes-per-pixel") o->lanes_per_pixel = (unsigned)std::strtoul(v, nullptr, 0);
        else if (a == "--intersect") o->intersect = v;
        else if (a == "--gpus") o->gpus = std::atoi(v);
        else if (a == "--out") o->out = v;
        else return false;
    }
    return o->width > 0 && o->height > 0 && o->gpus > 0;
}

World* BuildWorld(const Options& o) {
    if (o.scene == "cornell") return CreateCornellBoxScene();
    if (o.scene == "spheres") return createScene();
    if (o.scene == "random_spheres") return CreateRandomSpheresScene(o.count ? o.count : 4096, o.scene_seed);
    if (o.scene == "glass_spheres") return CreateGlassSpheresScene(o.count ? o.count : 5);
    return ReadSceneFile(o.scene.c_str());
}

}  // namespace

int main(int argc, char** argv) {
    Options opt;
    if (!Parse(argc, argv, &opt)) { Usage(); return 2; }

    Image image = CreateImage(opt.width, opt.height);
    World* world = BuildWorld(opt);
    if (!world || !image.pixelData) { std::fprintf(stderr, "cannot build scene '%s'\n", opt.scene.c_str()); return 1; }
    rt_world view = world->View();

    // One context per GPU; each renders an interleaved set of 8-row bands straight
    // into its own compact buffer, and the bands are interleaved back on the host.
    const int gpus = opt.gpus;
    std::vector<rt_context*> contexts(gpus, nullptr);
    for (int g = 0; g < gpus; ++g) {
        rt_status st = rt_create(g, &contexts[g]);
        if (st != RT_OK) { std::fprintf(stderr, "rt_create(%d): %s\n", g, rt_status_string(st)); return 1; }
        st = rt_upload_world(contexts[g], &view);
        if (st != RT_OK) { std::fprintf(stderr, "rt_upload_world: %s\n", rt_error_string(contexts[g])); return 1; }
    }

    std::vector<rt_render_params> params(gpus);
    std::vector<std::vector<uint32_t>> shards(gpus);
    std::vector<uint64_t> rays(gpus, 0);
    std::vector<rt_status> status(gpus, RT_OK);
    for (int g = 0; g < gpus; ++g) {
        rt_render_params& p = params[g];
        rt_default_params(&p, (uint32_t)opt.width, (uint32_t)opt.height, opt.spp);
        p.max_bounces = opt.bounces;
        p.seed_constant = opt.seed_constant;
        p.rng_mode = opt.rng == "row" ? RT_RNG_ROW_SERIAL : RT_RNG_PIXEL;
        p.lanes_per_pixel = opt.lanes_per_pixel;
        p.intersect_mode = opt.intersect == "scalar" ? RT_INTERSECT_SCALAR : RT_INTERSECT_WIDE8;
        if (gpus > 1) { p.band_rows = 8; p.band_index = (uint32_t)g; p.band_count = (uint32_t)gpus; }
        shards[g].resize((size_t)rt_local_row_count(&p) * opt.width);
    }

    auto start = std::chrono::steady_clock::now();
    std::vector<std::thread> workers;
    for (int g = 0; g < gpus; ++g)
        workers.emplace_back([&, g] {
            if (shards[g].empty()) return;
            status[g] = rt_render(contexts[g], &params[g], shards[g].data(), nullptr, &rays[g]);
        });
    for (std::thread& t : workers) t.join();
    for (int g = 0; g < gpus; ++g) {
        if (status[g] != RT_OK) { std::fprintf(stderr, "rt_render on GPU %d: %s\n", g, rt_error_string(contexts[g])); return 1; }
        uint32_t rows = rt_local_row_count(&params[g]);
        for (uint32_t local = 0; local < rows; ++local)
            std::memcpy(image.pixelData + (size_t)rt_global_row(&params[g], local) * opt.width,
                        shards[g].data() + (size_t)local * opt.width, (size_t)opt.width * 4);
    }
    auto end = std::chrono::steady_clock::now();

    unsigned long long bouncesComputed = 0;
    for (uint64_t r : rays) bouncesComputed += r;
    double timeElapsedMs = std::chrono::duration<double, std::milli>(end - start).count();
    // The reference's report lines (main.cpp:468-471).
    std::printf("Raytracing time: %.0fms\n", timeElapsedMs);
    std::printf("Total computed rays: %llu\n", bouncesComputed);
    std::printf("Performance: %.1fMray/s, %fms/ray\n", (bouncesComputed / 1000.0) / timeElapsedMs,
                timeElapsedMs / (double)bouncesComputed);

    bool ok = WriteImageFile(&image, opt.out.c_str());
    if (!ok) std::fprintf(stderr, "cannot write %s\n", opt.out.c_str());
    for (rt_context* c : contexts) rt_destroy(c);
    delete world;
    FreeImage(&image);
    return ok ? 0 : 1;
}
