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
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_framebuffer.h"
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
    bool compact = false;                    // RT_FLAG_COMPACT_TAILS
    std::string intersect = "wide";
    unsigned count = 0;
    unsigned long long scene_seed = 1;
    int gpus = 1;
    std::string devices;                     // explicit device ordinals, e.g. "0,1,2,3" (overrides --gpus)
    std::string out = "render.bmp";
    // Progressive rendering: `passes` renders of `spp` samples each are summed in a float
    // framebuffer, which can be stored after every pass and picked up again later.
    unsigned passes = 1;
    std::string checkpoint;                  // RTFB file written after every pass
    std::string resume;                      // RTFB file to continue from
};

void Usage() {
    std::fprintf(stderr,
                 "usage: raytracer [--scene cornell|spheres|random_spheres|glass_spheres|FILE.rtsc] [--count N] [--scene-seed S]\n"
                 "                 [--width W] [--height H] [--spp N] [--bounces N] [--seed-constant N]\n"
                 "                 [--rng pixel|row] [--lanes-per-pixel N] [--compact] [--intersect wide|wide4|scalar]\n"
                 "                 [--gpus N | --devices 0,1,...] [--out FILE.bmp]\n"
                 "                 [--passes N] [--checkpoint FILE.rtfb] [--resume FILE.rtfb]\n");
}

bool Parse(int argc, char** argv, Options* o) {
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto value = [&]() -> const char* { return i + 1 < argc ? argv[++i] : nullptr; };
        const char* v = nullptr;
        if (a == "--help" || a == "-h") return false;
        if (a == "--compact") { o->compact = true; continue; }
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
        else if (a == "--devices") o->devices = v;
        else if (a == "--out") o->out = v;
        else if (a == "--passes") o->passes = (unsigned)std::strtoul(v, nullptr, 0);
        else if (a == "--checkpoint") o->checkpoint = v;
        else if (a == "--resume") o->resume = v;
        else return false;
    }
    return o->width > 0 && o->height > 0 && o->gpus > 0 && o->passes > 0;
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

    // One context per GPU (the counterpart of the worker threads of main.cpp:450-454).
    std::vector<int> ordinals;
    if (!opt.devices.empty()) {
        for (size_t pos = 0; pos < opt.devices.size();) {
            ordinals.push_back(std::atoi(opt.devices.c_str() + pos));
            size_t comma = opt.devices.find(',', pos);
            pos = comma == std::string::npos ? opt.devices.size() : comma + 1;
        }
    } else {
        for (int g = 0; g < opt.gpus; ++g) ordinals.push_back(g);
    }
    std::vector<rt_context*> contexts(ordinals.size(), nullptr);
    for (size_t g = 0; g < ordinals.size(); ++g) {
        rt_status st = rt_create(ordinals[g], &contexts[g]);
        if (st != RT_OK) { std::fprintf(stderr, "rt_create(%d): %s\n", ordinals[g], rt_status_string(st)); return 1; }
        st = rt_upload_world(contexts[g], &view);
        if (st != RT_OK) { std::fprintf(stderr, "rt_upload_world: %s\n", rt_error_string(contexts[g])); return 1; }
    }

    rt_render_params params;
    rt_default_params(&params, (uint32_t)opt.width, (uint32_t)opt.height, opt.spp);
    params.max_bounces = opt.bounces;
    params.seed_constant = opt.seed_constant;
    params.rng_mode = opt.rng == "row" ? RT_RNG_ROW_SERIAL : RT_RNG_PIXEL;
    params.lanes_per_pixel = opt.lanes_per_pixel;
    if (opt.compact) params.flags |= RT_FLAG_COMPACT_TAILS;
    params.intersect_mode = opt.intersect == "scalar" ? RT_INTERSECT_SCALAR : opt.intersect == "wide4" ? RT_INTERSECT_WIDE4 : RT_INTERSECT_WIDE8;

    // The render: interleaved 8-row bands, one host thread per GPU, assembled in image.pixelData.
    auto start = std::chrono::steady_clock::now();
    uint64_t rays = 0;
    rt_status st = RT_OK;
    const bool progressive = opt.passes > 1 || !opt.checkpoint.empty() || !opt.resume.empty();
    if (!progressive) {
        st = rt_render_multi(contexts.data(), (uint32_t)contexts.size(), &params, image.pixelData, nullptr, &rays);
    } else {
        // Every pass draws fresh sample batches (batch_offset), returns its un-normalised sum, and the
        // sums are added in pass order; the image is the resolve of the total (the color / sampleSize
        // and pack of main.cpp:393 applied once, at the end).
        if (params.rng_mode != RT_RNG_PIXEL) {
            std::fprintf(stderr, "--passes/--checkpoint/--resume need --rng pixel: a row-serial stream has no second batch\n");
            return 2;
        }
        Accumulator acc;
        acc.Reset((uint32_t)opt.width, (uint32_t)opt.height);
        acc.lanes_per_pixel = params.lanes_per_pixel;
        acc.scene_hash = HashWorld(view);
        acc.settings_hash = HashSettings(params);
        if (!opt.resume.empty()) {
            Accumulator loaded;
            std::string why;
            if (!ReadAccumulatorFile(opt.resume.c_str(), &loaded, &why)) { std::fprintf(stderr, "%s\n", why.c_str()); return 1; }
            if (loaded.width != acc.width || loaded.height != acc.height || loaded.lanes_per_pixel != acc.lanes_per_pixel ||
                loaded.scene_hash != acc.scene_hash || loaded.settings_hash != acc.settings_hash) {
                std::fprintf(stderr, "%s was rendered with another scene, image size or settings\n", opt.resume.c_str());
                return 1;
            }
            acc = std::move(loaded);
            std::printf("Resumed at %u samples per pixel\n", acc.samples);
        }
        // The running sum stays on the GPUs (RT_FLAG_ACCUMULATE); it comes back to the host only where it is
        // needed: for a checkpoint and for the final resolve.
        const uint32_t width = (uint32_t)opt.width;
        if (!opt.resume.empty()) {
            // hand every GPU the rows it renders (interleaved 8-row bands, as rt_render_multi deals them)
            for (size_t g = 0; g < contexts.size() && st == RT_OK; ++g) {
                rt_render_params share = params;
                share.band_rows = contexts.size() > 1 ? 8u : 0u;
                share.band_index = (uint32_t)g;
                share.band_count = (uint32_t)contexts.size();
                const uint32_t rows = rt_local_row_count(&share);
                if (rows == 0) continue;
                std::vector<float> compact((size_t)rows * width * 3);
                for (uint32_t local = 0; local < rows; ++local)
                    std::memcpy(&compact[(size_t)local * width * 3], &acc.sum[(size_t)rt_global_row(&share, local) * width * 3],
                                (size_t)width * 3 * sizeof(float));
                st = rt_load_accumulator(contexts[g], &share, compact.data());
            }
        }
        bool started = !opt.resume.empty();
        for (unsigned pass = 0; pass < opt.passes && st == RT_OK; ++pass) {
            params.batch_offset = acc.next_batch;
            params.flags = (params.flags & ~(uint32_t)(RT_FLAG_OUTPUT_SUM | RT_FLAG_ACCUMULATE)) |
                           (started ? RT_FLAG_ACCUMULATE : RT_FLAG_OUTPUT_SUM);
            const bool need_sum = !opt.checkpoint.empty() || pass + 1 == opt.passes;
            uint64_t pass_rays = 0;
            st = rt_render_multi(contexts.data(), (uint32_t)contexts.size(), &params, nullptr, need_sum ? acc.sum.data() : nullptr,
                                 &pass_rays);
            if (st != RT_OK) break;
            started = true;
            acc.Advance(opt.spp, pass_rays);
            rays += pass_rays;
            if (!opt.checkpoint.empty() && !WriteAccumulatorFile(acc, opt.checkpoint.c_str())) {
                std::fprintf(stderr, "cannot write %s\n", opt.checkpoint.c_str());
                return 1;
            }
            std::printf("Pass %u/%u: %u samples per pixel\n", pass + 1, opt.passes, acc.samples);
        }
        if (st == RT_OK)
            st = rt_resolve(contexts[0], acc.sum.data(), (uint64_t)opt.width * (uint64_t)opt.height, acc.samples, image.pixelData, nullptr);
    }
    auto end = std::chrono::steady_clock::now();
    if (st != RT_OK) { std::fprintf(stderr, "render failed: %s\n", rt_error_string(contexts[0])); return 1; }

    unsigned long long bouncesComputed = rays;
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
