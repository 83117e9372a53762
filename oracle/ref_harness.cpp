// TEST INFRASTRUCTURE - not part of the product path.
//
// Drives the reference renderer's own hot-path functions (RaytraceWork,
// RaytraceWorld, IntersectWorldWide, IntersectWorld, the scene builders) so that
// their outputs can be captured as golden vectors and timed as a CPU baseline.
// The reference sources are compiled where they lie (-I/root/reference); nothing
// of them is copied here.  This file only adds:
//   * const-reference forwarding overloads for the four operator* functions that
//     the reference declares with non-const lvalue references and calls with
//     temporaries (math_matrix.h:47,57 and simd.h:282,292 - accepted by MSVC only).
//     Each forwards to the reference's own operator, so the arithmetic is theirs.
//   * a capture hook around RGBPackToUInt32WithsRGB (main.cpp:393) that records the
//     linear radiance the reference is about to tonemap, then calls the real packer.
//   * a driver (counterpart of main.cpp:409-475) with run-time width/height/spp/
//     thread count, because the reference hard-codes them.
//
// Build: see oracle/Makefile (outputs go to oracle/_ref/ only).
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>
#include <vector>

#include "platform.h"
#include "math_util.h"

inline Vector4 operator*(const Matrix4& left, const Vector4& right) {
    Matrix4 l(left);
    Vector4 r(right);
    return l * r;  // both lvalues: binds to the reference's operator*(Matrix4&, Vector4&)
}

inline Matrix4 operator*(const Matrix4& left, const Matrix4& right) {
    Matrix4 l(left);
    Matrix4 r(right);
    return l * r;
}

#include "simd.h"

inline LaneVector4 operator*(const LaneMatrix4& left, const LaneVector4& right) {
    LaneMatrix4 l(left);
    LaneVector4 r(right);
    return l * r;
}

inline LaneMatrix4 operator*(const LaneMatrix4& left, const LaneMatrix4& right) {
    LaneMatrix4 l(left);
    LaneMatrix4 r(right);
    return l * r;
}

#include "image.h"

// Linear-radiance capture (W*H*3 floats), filled from inside RaytraceWork.
static float* gRadianceCapture = nullptr;

static inline uint32_t RefRealPack(Vector3 c) { return RGBPackToUInt32WithsRGB(c); }

static inline uint32_t RefCapturePack(int32_t x, int32_t y, Image* image, Vector3 c) {
    if (gRadianceCapture) {
        size_t index = (size_t)y * (size_t)image->width + (size_t)x;
        gRadianceCapture[3 * index + 0] = c.x;
        gRadianceCapture[3 * index + 1] = c.y;
        gRadianceCapture[3 * index + 2] = c.z;
    }
    return RefRealPack(c);
}

// Expands inside RaytraceWork's pixel loop, where `x`, `y` and `image` are in scope
// (`frameBuffer` is not usable: its post-increment is unsequenced with this call).
#define RGBPackToUInt32WithsRGB(c) RefCapturePack(x, y, image, (c))
#define main reference_main
#include "main.cpp"
#undef main
#undef RGBPackToUInt32WithsRGB

// ----------------------------------------------------------------------------
// Scene file ("RTSC" v1): header, camera, then the four AoS arrays as the
// reference lays them out (Material 32 B, Plane 20 B, Sphere 20 B, RectangleXY 80 B).
// ----------------------------------------------------------------------------
struct SceneFileHeader {
    char magic[4];
    uint32_t version;
    uint32_t materialCount, planeCount, sphereCount, rectangleCount;
};

static_assert(sizeof(Material) == 32, "Material layout");
static_assert(sizeof(Plane) == 20, "Plane layout");
static_assert(sizeof(Sphere) == 20, "Sphere layout");
static_assert(sizeof(RectangleXY) == 80, "RectangleXY layout");
static_assert(sizeof(Camera) == 48, "Camera layout");

static void SanitizeWorld(World* w, bool hasSpheres, bool hasRects) {
    // The builders leave the fields of the primitive kinds they do not use
    // uninitialised (scene.h:332-342, 456-466); give them defined empty values.
    if (!hasSpheres) { w->sphereCount = 0; w->spheres = nullptr; w->sphereSoAArrayCount = 0; w->sphereSoAArray = nullptr; }
    if (!hasRects) { w->rectangleCount = 0; w->rectangles = nullptr; w->rectangleLaneArrayCount = 0; w->rectangleLaneArray = nullptr; }
}

static World* BuildNamedScene(const char* name) {
    if (!strcmp(name, "cornell")) {
        World* w = CreateCornellBoxScene();
        w->planes = nullptr;
        SanitizeWorld(w, false, true);
        return w;
    }
    if (!strcmp(name, "spheres")) {
        World* w = createScene();
        SanitizeWorld(w, true, false);
        return w;
    }
    return nullptr;
}

static int WriteSceneFile(World* w, const char* path) {
    FILE* f = fopen(path, "wb");
    if (!f) return 1;
    SceneFileHeader h = {{'R', 'T', 'S', 'C'}, 1, w->materialCount, w->planeCount, w->sphereCount, w->rectangleCount};
    fwrite(&h, sizeof h, 1, f);
    fwrite(w->camera, sizeof(Camera), 1, f);
    fwrite(w->materials, sizeof(Material), w->materialCount, f);
    if (w->planeCount) fwrite(w->planes, sizeof(Plane), w->planeCount, f);
    if (w->sphereCount) fwrite(w->spheres, sizeof(Sphere), w->sphereCount, f);
    if (w->rectangleCount) fwrite(w->rectangles, sizeof(RectangleXY), w->rectangleCount, f);
    fclose(f);
    return 0;
}

// Loads a scene file into a World, packing the AoSoA mirrors the way the
// builders do (scene.h:255-286, 413-448) but with zero-filled tail lanes that
// can never be hit (radius^2 = 0 at a far position; singular matrices).
static World* LoadSceneFile(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return nullptr;
    SceneFileHeader h;
    if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, "RTSC", 4) || h.version != 1) { fclose(f); return nullptr; }
    World* w = new World;
    memset((void*)w, 0, sizeof(World));
    w->camera = (Camera*)malloc(sizeof(Camera));
    if (fread((void*)w->camera, sizeof(Camera), 1, f) != 1) return nullptr;
    w->materialCount = h.materialCount;
    w->materials = (Material*)malloc(sizeof(Material) * (h.materialCount + 1));
    if (fread((void*)w->materials, sizeof(Material), h.materialCount, f) != h.materialCount) return nullptr;
    w->planeCount = h.planeCount;
    w->planes = (Plane*)malloc(sizeof(Plane) * (h.planeCount + 1));
    if (h.planeCount && fread((void*)w->planes, sizeof(Plane), h.planeCount, f) != h.planeCount) return nullptr;
    w->sphereCount = h.sphereCount;
    w->spheres = (Sphere*)malloc(sizeof(Sphere) * (h.sphereCount + 1));
    if (h.sphereCount && fread((void*)w->spheres, sizeof(Sphere), h.sphereCount, f) != h.sphereCount) return nullptr;
    w->rectangleCount = h.rectangleCount;
    w->rectangles = (RectangleXY*)malloc(sizeof(RectangleXY) * (h.rectangleCount + 1));
    if (h.rectangleCount && fread((void*)w->rectangles, sizeof(RectangleXY), h.rectangleCount, f) != h.rectangleCount) return nullptr;
    fclose(f);

    w->sphereSoAArrayCount = (w->sphereCount + LANE_WIDTH - 1) / LANE_WIDTH;
    w->sphereSoAArray = w->sphereSoAArrayCount ? new SphereSoALane[w->sphereSoAArrayCount] : nullptr;
    for (uint32_t i = 0; i < w->sphereSoAArrayCount; ++i) {
        ALIGN_LANE float px[LANE_WIDTH], py[LANE_WIDTH], pz[LANE_WIDTH], r2[LANE_WIDTH], mi[LANE_WIDTH];
        for (uint32_t j = 0; j < LANE_WIDTH; ++j) {
            uint32_t k = i * LANE_WIDTH + j;
            if (k < w->sphereCount) {
                Sphere s = w->spheres[k];
                px[j] = s.position.x; py[j] = s.position.y; pz[j] = s.position.z;
                r2[j] = s.radius * s.radius; mi[j] = (float)s.materialIndex;
            } else {
                px[j] = py[j] = pz[j] = 1.0e30f; r2[j] = 0.0f; mi[j] = 0.0f;
            }
        }
        SphereSoALane lane = {};
        lane.position = LaneVector3(LaneF32(px), LaneF32(py), LaneF32(pz));
        lane.radiusSquared = LaneF32(r2);
        lane.materialIndex = LaneF32(mi);
        w->sphereSoAArray[i] = lane;
    }

    w->rectangleLaneArrayCount = (w->rectangleCount + LANE_WIDTH - 1) / LANE_WIDTH;
    w->rectangleLaneArray = w->rectangleLaneArrayCount ? new RectangleLane[w->rectangleLaneArrayCount] : nullptr;
    for (uint32_t i = 0; i < w->rectangleLaneArrayCount; ++i) {
        ALIGN_LANE float m[4][4][LANE_WIDTH];
        ALIGN_LANE float n[3][LANE_WIDTH];
        ALIGN_LANE float mi[LANE_WIDTH];
        memset(m, 0, sizeof m); memset(n, 0, sizeof n); memset(mi, 0, sizeof mi);
        for (uint32_t j = 0; j < LANE_WIDTH; ++j) {
            uint32_t k = i * LANE_WIDTH + j;
            if (k >= w->rectangleCount) continue;  // all-zero matrix: t = -0/0 = NaN, never a hit
            RectangleXY* r = w->rectangles + k;
            for (uint32_t a = 0; a < 4; ++a) for (uint32_t b = 0; b < 4; ++b) m[a][b][j] = r->transformMatrix[a][b];
            n[0][j] = r->normal.x; n[1][j] = r->normal.y; n[2][j] = r->normal.z;
            mi[j] = (float)r->materialIndex;
        }
        RectangleLane lane = {};
        lane.transformMatrix = LaneMatrix4(m);
        lane.normal = LaneVector3(n);
        lane.materialIndex = LaneF32(mi);
        w->rectangleLaneArray[i] = lane;
    }
    return w;
}

static World* GetScene(const char* nameOrPath) {
    World* w = BuildNamedScene(nameOrPath);
    if (!w) w = LoadSceneFile(nameOrPath);
    if (!w) { fprintf(stderr, "cannot build or load scene '%s'\n", nameOrPath); exit(2); }
    return w;
}

static int WriteBlob(const char* path, const void* data, size_t bytes) {
    FILE* f = fopen(path, "wb");
    if (!f) return 1;
    fwrite(data, 1, bytes, f);
    fclose(f);
    return 0;
}

static double NowMs() {
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1000.0 + t.tv_nsec / 1.0e6;
}

// render <scene> W H spp threads rowsPerOrder outPrefix|-
// One WorkOrder per `rowsPerOrder` rows (the reference ships 1, main.cpp:433;
// 0 means one order for the whole image, its SINGLE_THREAD shape, main.cpp:418-430).
static int CmdRender(int argc, char** argv) {
    if (argc < 9) return 2;
    World* world = GetScene(argv[2]);
    int W = atoi(argv[3]), H = atoi(argv[4]);
    uint32_t spp = (uint32_t)atoi(argv[5]);
    int threads = atoi(argv[6]);
    int stride = atoi(argv[7]);
    const char* prefix = argv[8];
    bool save = strcmp(prefix, "-") != 0;

    Image image = CreateImage(W, H);
    std::vector<float> radiance;
    if (save) { radiance.assign((size_t)W * H * 3, 0.0f); gRadianceCapture = radiance.data(); }

    if (stride <= 0) stride = H;
    uint32_t orderCount = (uint32_t)((H + stride - 1) / stride);
    WorkQueue queue = {};
    queue.workOrders = (WorkOrder*)malloc(orderCount * sizeof(WorkOrder));
    queue.workOrderCount = orderCount;
    for (uint32_t i = 0; i < orderCount; ++i) {
        WorkOrder* o = queue.workOrders + i;
        o->image = &image;
        o->world = world;
        o->startRowIndex = i * stride;
        o->endRowIndex = (i + 1) * stride < (uint32_t)H ? (i + 1) * stride : (uint32_t)H;
        o->sampleSize = spp;
    }

    double t0 = NowMs();
    std::vector<pthread_t> pool;
    for (int i = 1; i < threads; ++i) {
        pthread_t th;
        pthread_create(&th, NULL, ThreadProc, &queue);
        pool.push_back(th);
    }
    while (RaytraceWork(&queue)) {}
    for (pthread_t th : pool) pthread_join(th, NULL);
    double t1 = NowMs();

    unsigned long long rays = queue.totalBouncesComputed;
    double ms = t1 - t0;
    printf("{\"width\": %d, \"height\": %d, \"spp\": %u, \"threads\": %d, \"rows_per_order\": %d, "
           "\"rays\": %llu, \"ms\": %.3f, \"mrays_per_s\": %.3f}\n",
           W, H, spp, threads, stride, rays, ms, rays / 1000.0 / ms);
    if (save) {
        char path[1024];
        snprintf(path, sizeof path, "%s.radiance.f32", prefix);
        WriteBlob(path, radiance.data(), radiance.size() * sizeof(float));
        snprintf(path, sizeof path, "%s.pixels.u32", prefix);
        WriteBlob(path, image.pixelData, (size_t)W * H * 4);
        snprintf(path, sizeof path, "%s.bmp", prefix);
        WriteImageFile(&image, path);
    }
    return 0;
}

static uint32_t HarnessRand(uint64_t* s) {  // splitmix-style generator for test inputs only
    *s += 0x9E3779B97F4A7C15ull;
    uint64_t z = *s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}

static float HarnessUniform(uint64_t* s, float lo, float hi) {
    return lo + (hi - lo) * (float)(HarnessRand(s) >> 8) * (1.0f / 16777216.0f);
}

struct RayRecord {      // 24 floats/words per ray
    float origin[3];
    float direction[3];
    float wideT; uint32_t wideHit; uint32_t wideMaterial; float wideNormal[3];
    float scalarT; uint32_t scalarHit; uint32_t scalarMaterial; float scalarNormal[3];
};

// rays <scene> N seed out.bin : random rays through IntersectWorldWide and IntersectWorld.
// Half the rays start at the camera and aim into the scene, half start at a
// surface point found by the first half and leave in a random direction.
static int CmdRays(int argc, char** argv) {
    if (argc < 6) return 2;
    World* world = GetScene(argv[2]);
    int N = atoi(argv[3]);
    uint64_t seed = strtoull(argv[4], NULL, 0);
    std::vector<RayRecord> records((size_t)N);
    Vector3 lastHitPoint = world->camera->position;
    for (int i = 0; i < N; ++i) {
        Ray ray = {};
        if ((i & 1) == 0) {
            ray.origin = world->camera->position;
            Vector3 target(HarnessUniform(&seed, -9.0f, 9.0f), HarnessUniform(&seed, -9.0f, 9.0f), HarnessUniform(&seed, -15.0f, 2.0f));
            ray.direction = Normalize(target - ray.origin);
        } else {
            ray.origin = lastHitPoint;
            Vector3 d(HarnessUniform(&seed, -1.0f, 1.0f), HarnessUniform(&seed, -1.0f, 1.0f), HarnessUniform(&seed, -1.0f, 1.0f));
            if (d.x == 0.0f && d.y == 0.0f && d.z == 0.0f) d.x = 1.0f;
            ray.direction = Normalize(d);
        }
        RayRecord& r = records[i];
        memset(&r, 0, sizeof r);
        r.origin[0] = ray.origin.x; r.origin[1] = ray.origin.y; r.origin[2] = ray.origin.z;
        r.direction[0] = ray.direction.x; r.direction[1] = ray.direction.y; r.direction[2] = ray.direction.z;

        WorldIntersectionResult wide = {};
        bool wideHit = IntersectWorldWide(world, &ray, &wide);
        r.wideHit = wideHit; r.wideT = wide.t; r.wideMaterial = wide.hitMaterialIndex;
        if (wideHit) { r.wideNormal[0] = wide.hitNormal.x; r.wideNormal[1] = wide.hitNormal.y; r.wideNormal[2] = wide.hitNormal.z; }

        WorldIntersectionResult scalar = {};
        scalar.hitMaterialIndex = 0;
        bool scalarHit = IntersectWorld(world, &ray, &scalar);
        r.scalarHit = scalarHit; r.scalarT = scalar.t; r.scalarMaterial = scalar.hitMaterialIndex;
        if (scalarHit) { r.scalarNormal[0] = scalar.hitNormal.x; r.scalarNormal[1] = scalar.hitNormal.y; r.scalarNormal[2] = scalar.hitNormal.z; }

        if (wideHit) lastHitPoint = ray.origin + ray.direction * wide.t;
    }
    return WriteBlob(argv[5], records.data(), records.size() * sizeof(RayRecord));
}

struct PathRecord {     // 16 words per path
    float origin[3];
    float direction[3];
    uint32_t stateIn;
    float radiance[3];
    uint32_t stateOut;
    uint32_t bounces;
    uint32_t pad[4];
};

// paths <scene> N seed out.bin : camera-like primary rays through RaytraceWorld.
static int CmdPaths(int argc, char** argv) {
    if (argc < 6) return 2;
    World* world = GetScene(argv[2]);
    int N = atoi(argv[3]);
    uint64_t seed = strtoull(argv[4], NULL, 0);
    std::vector<PathRecord> records((size_t)N);
    WorkQueue queue = {};
    for (int i = 0; i < N; ++i) {
        Ray ray = {};
        ray.origin = world->camera->position;
        Vector3 target(HarnessUniform(&seed, -8.0f, 8.0f), HarnessUniform(&seed, -8.0f, 8.0f), HarnessUniform(&seed, -13.0f, 3.0f));
        ray.direction = Normalize(target - ray.origin);
        uint32_t state = HarnessRand(&seed) | 1u;
        PathRecord& r = records[i];
        memset(&r, 0, sizeof r);
        r.origin[0] = ray.origin.x; r.origin[1] = ray.origin.y; r.origin[2] = ray.origin.z;
        r.direction[0] = ray.direction.x; r.direction[1] = ray.direction.y; r.direction[2] = ray.direction.z;
        r.stateIn = state;
        uint64_t bounces = 0;
        Vector3 c = RaytraceWorld(world, &ray, &state, &queue, &bounces);
        r.radiance[0] = c.x; r.radiance[1] = c.y; r.radiance[2] = c.z;
        r.stateOut = state;
        r.bounces = (uint32_t)bounces;
    }
    return WriteBlob(argv[5], records.data(), records.size() * sizeof(PathRecord));
}

// misc <out.bin> : known answers for the small functions (a4, a5, a7 of the scope table).
static int CmdMisc(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 1;
    // 1. xorshift32 stream + unilateral/bilateral from a few seeds: (state, u, b) x 64 per seed
    uint32_t seeds[4] = {1u, 262346u * (0u + 1u * 36u), 262346u * (719u + 720u * 36u), 0xFFFFFFFFu};
    for (int k = 0; k < 4; ++k) {
        uint32_t s = seeds[k];
        for (int i = 0; i < 64; ++i) {
            uint32_t s1 = s; float u = RandomUnilateral(&s1);
            uint32_t s2 = s; float b = RandomBilateral(&s2);
            fwrite(&s, 4, 1, f); fwrite(&s1, 4, 1, f); fwrite(&u, 4, 1, f); fwrite(&b, 4, 1, f);
            s = s1;
        }
    }
    // 2. LinearTosRGB + pack on a sweep of values: (v, srgb, packed) x 4096
    uint64_t seed = 12345;
    for (int i = 0; i < 4096; ++i) {
        float v = (i < 64) ? (float)i * 0.0001f : HarnessUniform(&seed, -0.1f, 1.2f);
        float s = LinearTosRGB(v);
        uint32_t p = RefRealPack(Vector3(v, v * 0.5f, v * 0.25f));
        fwrite(&v, 4, 1, f); fwrite(&s, 4, 1, f); fwrite(&p, 4, 1, f);
    }
    // 3. Refract / Schlick: (i[3], n[3], ri, ok, out[3], schlick) x 2048
    for (int i = 0; i < 2048; ++i) {
        Vector3 in = Normalize(Vector3(HarnessUniform(&seed, -1, 1), HarnessUniform(&seed, -1, 1), HarnessUniform(&seed, -1, 1)));
        Vector3 n = Normalize(Vector3(HarnessUniform(&seed, -1, 1), HarnessUniform(&seed, -1, 1), HarnessUniform(&seed, -1, 1)));
        float ri = (i & 1) ? 1.5f : HarnessUniform(&seed, 1.05f, 2.4f);
        Vector3 out(0.0f, 0.0f, 0.0f);
        uint32_t ok = Refract(in, n, ri, &out);
        float sch = Schlick(in, n, ri);
        float rec[12] = {in.x, in.y, in.z, n.x, n.y, n.z, ri, 0.0f, out.x, out.y, out.z, sch};
        memcpy(&rec[7], &ok, 4);
        fwrite(rec, 4, 12, f);
    }
    fclose(f);
    return 0;
}

static int CmdScene(int argc, char** argv) {
    if (argc < 4) return 2;
    return WriteSceneFile(GetScene(argv[2]), argv[3]);
}

int main(int argc, char** argv) {
    int rc = 2;
    if (argc >= 2) {
        if (!strcmp(argv[1], "render")) rc = CmdRender(argc, argv);
        else if (!strcmp(argv[1], "rays")) rc = CmdRays(argc, argv);
        else if (!strcmp(argv[1], "paths")) rc = CmdPaths(argc, argv);
        else if (!strcmp(argv[1], "misc")) rc = CmdMisc(argc, argv);
        else if (!strcmp(argv[1], "scene")) rc = CmdScene(argc, argv);
    }
    if (rc == 2) {
        fprintf(stderr,
                "usage: ref_harness render <cornell|spheres|file.rtsc> W H spp threads rowsPerOrder <outPrefix|->\n"
                "       ref_harness rays   <scene> N seed out.bin\n"
                "       ref_harness paths  <scene> N seed out.bin\n"
                "       ref_harness misc   out.bin\n"
                "       ref_harness scene  <cornell|spheres> out.rtsc\n");
    }
    return rc;
}
