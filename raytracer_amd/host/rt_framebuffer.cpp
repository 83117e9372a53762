#include "rt_framebuffer.h"

#include <cstdio>
#include <cstring>

namespace rth {

namespace {

constexpr uint32_t kVersion = 1;
constexpr size_t kHeaderBytes = 64;

struct Fnv {
    uint64_t h = 1469598103934665603ull;
    void bytes(const void* p, size_t n) {
        const unsigned char* b = static_cast<const unsigned char*>(p);
        for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    }
    void u32(uint32_t v) { bytes(&v, 4); }
    void f32(float v) { bytes(&v, 4); }
};

void put32(unsigned char* p, uint32_t v) { for (int i = 0; i < 4; ++i) p[i] = (unsigned char)(v >> (8 * i)); }
void put64(unsigned char* p, uint64_t v) { for (int i = 0; i < 8; ++i) p[i] = (unsigned char)(v >> (8 * i)); }
uint32_t get32(const unsigned char* p) { uint32_t v = 0; for (int i = 0; i < 4; ++i) v |= (uint32_t)p[i] << (8 * i); return v; }
uint64_t get64(const unsigned char* p) { uint64_t v = 0; for (int i = 0; i < 8; ++i) v |= (uint64_t)p[i] << (8 * i); return v; }

}  // namespace

void Accumulator::Reset(uint32_t w, uint32_t h) {
    width = w; height = h;
    samples = 0; next_batch = 0; rays = 0;
    sum.assign((size_t)w * h * 3, 0.0f);
}

void Accumulator::AddPass(const float* pass_sum, uint32_t pass_samples, uint64_t pass_rays) {
    for (size_t i = 0; i < sum.size(); ++i) sum[i] += pass_sum[i];
    Advance(pass_samples, pass_rays);
}

void Accumulator::Advance(uint32_t pass_samples, uint64_t pass_rays) {
    samples += pass_samples;
    next_batch += lanes_per_pixel;
    rays += pass_rays;
}

uint64_t HashWorld(const rt_world& w) {
    Fnv f;
    f.u32(w.material_count); f.bytes(w.materials, (size_t)w.material_count * sizeof(rt_material));
    f.u32(w.plane_count);
    for (uint32_t i = 0; i < w.plane_count; ++i) {   // field by field: the records carry no padding, but stay layout-independent
        f.bytes(w.planes[i].normal, 12); f.f32(w.planes[i].d); f.u32(w.planes[i].material_index);
    }
    f.u32(w.sphere_count);
    for (uint32_t i = 0; i < w.sphere_count; ++i) {
        f.bytes(w.spheres[i].position, 12); f.f32(w.spheres[i].radius); f.u32(w.spheres[i].material_index);
    }
    f.u32(w.rectangle_count);
    for (uint32_t i = 0; i < w.rectangle_count; ++i) {
        f.bytes(w.rectangles[i].inverse_transform, 64); f.bytes(w.rectangles[i].normal, 12); f.u32(w.rectangles[i].material_index);
    }
    if (w.camera) f.bytes(w.camera, sizeof(rt_camera));
    return f.h;
}

uint64_t HashSettings(const rt_render_params& p) {
    Fnv f;
    f.u32(p.width); f.u32(p.height); f.u32(p.samples_per_pixel);
    f.u32(p.max_bounces); f.f32(p.min_hit_distance); f.f32(p.plane_tolerance);
    f.u32(p.seed_constant); f.u32(p.rng_mode); f.u32(p.lanes_per_pixel); f.u32(p.intersect_mode);
    return f.h;
}

bool WriteAccumulatorFile(const Accumulator& acc, const char* path) {
    if (acc.sum.size() != (size_t)acc.width * acc.height * 3) return false;
    unsigned char header[kHeaderBytes] = {0};
    std::memcpy(header, "RTFB", 4);
    put32(header + 4, kVersion);
    put32(header + 8, acc.width); put32(header + 12, acc.height);
    put32(header + 16, 3);
    put32(header + 20, acc.samples); put32(header + 24, acc.next_batch); put32(header + 28, acc.lanes_per_pixel);
    put64(header + 32, acc.scene_hash); put64(header + 40, acc.settings_hash); put64(header + 48, acc.rays);
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite(header, sizeof header, 1, f) == 1 &&
              (acc.sum.empty() || std::fwrite(acc.sum.data(), acc.sum.size() * sizeof(float), 1, f) == 1);
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) { std::remove(tmp.c_str()); return false; }
    return std::rename(tmp.c_str(), path) == 0;
}

bool ReadAccumulatorFile(const char* path, Accumulator* acc, std::string* error) {
    auto bad = [&](const char* why) { if (error) *error = std::string(path) + ": " + why; return false; };
    FILE* f = std::fopen(path, "rb");
    if (!f) return bad("cannot open");
    unsigned char header[kHeaderBytes];
    if (std::fread(header, sizeof header, 1, f) != 1) { std::fclose(f); return bad("shorter than the 64-byte header"); }
    if (std::memcmp(header, "RTFB", 4) != 0) { std::fclose(f); return bad("not an RTFB file"); }
    if (get32(header + 4) != kVersion) { std::fclose(f); return bad("unsupported version"); }
    Accumulator a;
    a.width = get32(header + 8); a.height = get32(header + 12);
    const uint32_t channels = get32(header + 16);
    a.samples = get32(header + 20); a.next_batch = get32(header + 24); a.lanes_per_pixel = get32(header + 28);
    a.scene_hash = get64(header + 32); a.settings_hash = get64(header + 40); a.rays = get64(header + 48);
    if (channels != 3 || a.width == 0 || a.height == 0 || a.width > 65536 || a.height > 65536) {
        std::fclose(f);
        return bad("bad image size or channel count");
    }
    a.sum.resize((size_t)a.width * a.height * 3);
    const bool ok = std::fread(a.sum.data(), a.sum.size() * sizeof(float), 1, f) == 1;
    const bool at_end = ok && std::fgetc(f) == EOF;
    std::fclose(f);
    if (!ok) return bad("pixel data truncated");
    if (!at_end) return bad("trailing bytes after the pixel data");
    *acc = std::move(a);
    return true;
}

}  // namespace rth
