// Scene builders.  The two hard-coded scenes reproduce the reference's
// construction arithmetic step by step (scene.h:59-161, 193-469) so that the
// rectangle matrices and camera axes are bit-identical to what the reference
// computes when built with -ffp-contract=off.
#include "rt_scene.h"

#include <cstdio>
#include <cstring>

namespace rth {

const Vec3 XAxis(1.0f, 0.0f, 0.0f);
const Vec3 YAxis(0.0f, 1.0f, 0.0f);
const Vec3 ZAxis(0.0f, 0.0f, 1.0f);

static const float kPi = 3.1415926535f;       // math_util.h:11
static const float kHalfPi = kPi * 0.5f;      // math_util.h:12

Camera::Camera(Vec3 cameraPosition) {
    position = cameraPosition;
    zVec = Normalize(cameraPosition);
    xVec = Normalize(Cross(Vec3(0.0f, 1.0f, 0.0f), zVec));
    yVec = Normalize(Cross(zVec, xVec));
}

rt_world World::View() const {
    rt_world w;
    w.material_count = (uint32_t)materials.size();
    w.materials = reinterpret_cast<const rt_material*>(materials.data());
    w.plane_count = (uint32_t)planes.size();
    w.planes = planes.empty() ? nullptr : reinterpret_cast<const rt_plane*>(planes.data());
    w.sphere_count = (uint32_t)spheres.size();
    w.spheres = spheres.empty() ? nullptr : reinterpret_cast<const rt_sphere*>(spheres.data());
    w.rectangle_count = (uint32_t)rectangles.size();
    w.rectangles = rectangles.empty() ? nullptr : reinterpret_cast<const rt_rectangle*>(rectangles.data());
    w.camera = reinterpret_cast<const rt_camera*>(&camera);
    return w;
}

// Rotation about a principal axis, written over the identity; any other axis leaves the identity.
static Mat4 AxisRotation(Vec3 axis, float angle, bool* supported) {
    Mat4 r = Mat4::Identity();
    *supported = true;
    if (axis == XAxis) RotateMatrixXAxis(r, angle);
    else if (axis == YAxis) RotateMatrixYAxis(r, angle);
    else if (axis == ZAxis) RotateMatrixZAxis(r, angle);
    else *supported = false;
    return r;
}

RectangleXY CreateRectangle(Vec3 position, Vec3 scale, uint32_t materialIndex, Vec3 initialRotationAxis,
                            float initialRotationAngle) {
    RectangleXY rect;
    Mat4 s = Mat4::Identity(), t = Mat4::Identity();
    ScaleMatrix(s, scale);
    TranslateMatrix(t, position);
    bool supported;
    Mat4 r = AxisRotation(initialRotationAxis, initialRotationAngle, &supported);
    rect.transformMatrix = (t * r) * s;
    rect.normal = (r * Vec4(rect.normal, 0.0f)).xyz();
    rect.materialIndex = materialIndex;
    return rect;
}

void RotateRectangle(RectangleXY* rect, Vec3 axis, float angle) {
    bool supported;
    Mat4 r = AxisRotation(axis, angle, &supported);
    if (!supported) return;   // the reference throws a C string here (scene.h:68); arbitrary axes are not supported
    rect->normal = (r * Vec4(rect->normal, 0.0f)).xyz();
    rect->transformMatrix = r * rect->transformMatrix;
}

Box CreateBox(Vec3 position, Vec3 scale, uint32_t materialIndex) {
    Box box;
    box.position = position;
    auto moved = [&](int axis, float delta) {   // one coordinate of the centre shifted by +-half extent
        Vec3 p = position;
        (&p.x)[axis] += delta;
        return p;
    };
    const Vec3 lid(scale.x, scale.z, 1.0f), side(scale.z, scale.y, 1.0f), face(scale.x, scale.y, 1.0f);
    box.rectangles[0] = CreateRectangle(moved(2, scale.z), face, materialIndex);                       // front
    box.rectangles[1] = CreateRectangle(moved(2, -scale.z), face, materialIndex);                      // back
    box.rectangles[2] = CreateRectangle(moved(0, scale.x), side, materialIndex, YAxis, -kHalfPi);      // right
    box.rectangles[3] = CreateRectangle(moved(0, -scale.x), side, materialIndex, YAxis, kHalfPi);      // left
    box.rectangles[4] = CreateRectangle(moved(1, scale.y), lid, materialIndex, XAxis, -kHalfPi);       // top
    box.rectangles[5] = CreateRectangle(moved(1, -scale.y), lid, materialIndex, XAxis, -kHalfPi);      // bottom
    return box;
}

void RotateBox(Box* box, Vec3 axis, float angle) {
    // Rotate every face about the box centre: move the centre to the origin, rotate, move back.
    Mat4 toBox = Mat4::Identity();
    TranslateMatrix(toBox, box->position);
    for (RectangleXY& rect : box->rectangles) {
        rect.transformMatrix = Inverse(toBox) * rect.transformMatrix;
        RotateRectangle(&rect, axis, angle);
        rect.transformMatrix = toBox * rect.transformMatrix;
    }
}

static Material Diffuse(float r, float g, float b) {
    Material m;
    m.color = Vec3(r, g, b);
    return m;
}

World* createScene() {
    World* world = new World;
    Plane ground;
    ground.normal = Vec3(0.0f, 1.0f, 0.0f);
    ground.d = 0;
    ground.materialIndex = 1;
    world->planes.push_back(ground);

    struct SphereSpec { float x, y, z, radius; uint32_t material; };
    static const SphereSpec specs[8] = {
        {0.0f, 1.0f, 0.0f, 1.0f, 2}, {-2.0f, 1.0f, 0.0f, 1.0f, 3}, {-4.0f, 2.0f, 1.0f, 1.0f, 4}, {2.0f, 1.0f, -1.0f, 1.0f, 5},
        {-2.0f, 1.0f, 5.0f, 1.0f, 2}, {5.0f, 2.0f, -6.0f, 2.0f, 3}, {-4.0f, 4.0f, 5.0f, 1.0f, 6}, {2.0f, 1.0f, 4.0f, 1.0f, 5},
    };
    for (const SphereSpec& s : specs) {
        Sphere sphere;
        sphere.position = Vec3(s.x, s.y, s.z);
        sphere.radius = s.radius;
        sphere.materialIndex = s.material;
        world->spheres.push_back(sphere);
    }

    Material sky;                                        // 0: black sky
    Material ground_mat = Diffuse(0.8f, 0.8f, 0.8f);      // 1
    Material red = Diffuse(0.8f, 0.3f, 0.3f);             // 2
    Material mirror;                                     // 3
    mirror.color = Vec3(1.0f, 1.0f, 1.0f) * 0.9f;
    mirror.reflection = 1.0f;
    Material gold = Diffuse(0.8f, 0.6f, 0.2f);            // 4: glossy
    gold.reflection = 0.9f;
    Material glass;                                      // 5
    glass.color = Vec3(1.0f, 1.0f, 1.0f) * 0.9f;
    glass.refractiveIndex = 1.5f;
    glass.reflection = 1.0f;
    Material warm_light;                                 // 6
    warm_light.emitColor = Vec3(20.0f, 15.0f, 10.0f) * 0.2f;
    Material blue_light;                                 // 7 (unused by any primitive)
    blue_light.emitColor = Vec3(1.0f, 1.0f, 18.0f);
    world->materials = {sky, ground_mat, red, mirror, gold, glass, warm_light, blue_light};

    world->camera = Camera(Vec3(0.0f, 4.0f, 10.0f));
    return world;
}

World* CreateCornellBoxScene() {
    World* world = new World;
    Material sky;
    Material white = Diffuse(0.73f, 0.73f, 0.73f);
    Material green = Diffuse(0.12f, 0.45f, 0.15f);
    Material red = Diffuse(0.65f, 0.05f, 0.05f);
    Material light;
    light.emitColor = Vec3(15.0f, 15.0f, 15.0f);
    world->materials = {sky, white, green, red, light};

    std::vector<RectangleXY>& r = world->rectangles;
    r.push_back(CreateRectangle(Vec3(0.0f, 7.99f, -6.0f), Vec3(2.0f, 2.0f, 1.0f), 4, XAxis, -kHalfPi));    // light
    r.push_back(CreateRectangle(Vec3(0.0f, -8.0f, -8.0f), Vec3(8.0f, 10.0f, 1.0f), 1, XAxis, -kHalfPi));   // floor
    r.push_back(CreateRectangle(Vec3(8.0f, 0.0f, -8.0f), Vec3(10.0f, 8.0f, 1.0f), 3, YAxis, -kHalfPi));    // right, red
    r.push_back(CreateRectangle(Vec3(-8.0f, 0.0f, -8.0f), Vec3(10.0f, 8.0f, 1.0f), 2, YAxis, kHalfPi));    // left, green
    r.push_back(CreateRectangle(Vec3(0.0f, 0.0f, -14.0f), Vec3(8.0f, 8.0f, 1.0f), 1));                     // back
    r.push_back(CreateRectangle(Vec3(0.0f, 8.0f, -8.0f), Vec3(8.0f, 10.0f, 1.0f), 1, XAxis, -kHalfPi));    // ceiling

    Box small_box = CreateBox(Vec3(2.0f, -6.0f, -3.0f), Vec3(2.0f, 2.0f, 2.0f), 1);
    RotateBox(&small_box, Vec3(0.0f, 1.0f, 0.0f), -0.3f);
    Box tall_box = CreateBox(Vec3(-2.0f, -4.0f, -8.0f), Vec3(2.0f, 4.0f, 2.0f), 1);
    RotateBox(&tall_box, Vec3(0.0f, 1.0f, 0.0f), 0.3f);
    for (const RectangleXY& face : small_box.rectangles) r.push_back(face);
    for (const RectangleXY& face : tall_box.rectangles) r.push_back(face);

    for (RectangleXY& rect : r) rect.transformMatrix = Inverse(rect.transformMatrix);   // scene.h:408-411

    world->camera = Camera(Vec3(0.0f, 1.0f, 20.0f));
    return world;
}

// ---- generated scenes ------------------------------------------------------
namespace {
struct SplitMix64 {
    uint64_t s;
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    float uniform(float lo, float hi) { return lo + (hi - lo) * ((float)(next() >> 40) * (1.0f / 16777216.0f)); }
};
}  // namespace

World* CreateRandomSpheresScene(uint32_t count, uint64_t seed, float extent) {
    World* world = new World;
    SplitMix64 rng{seed};
    Plane ground;
    ground.normal = Vec3(0.0f, 1.0f, 0.0f);
    ground.d = 0;
    ground.materialIndex = 1;
    world->planes.push_back(ground);

    Material sky;
    sky.emitColor = Vec3(0.35f, 0.45f, 0.6f);
    world->materials.push_back(sky);
    world->materials.push_back(Diffuse(0.6f, 0.6f, 0.6f));
    // 62 more materials: 50 % diffuse, 15 % glossy, 15 % mirror, 15 % glass, 5 % emissive.
    for (int i = 0; i < 62; ++i) {
        float pick = rng.uniform(0.0f, 1.0f);
        Material m = Diffuse(rng.uniform(0.1f, 0.95f), rng.uniform(0.1f, 0.95f), rng.uniform(0.1f, 0.95f));
        if (pick < 0.50f) {
        } else if (pick < 0.65f) {
            m.reflection = rng.uniform(0.3f, 0.95f);
        } else if (pick < 0.80f) {
            m.reflection = 1.0f;
        } else if (pick < 0.95f) {
            m.color = Vec3(0.95f, 0.95f, 0.95f);
            m.refractiveIndex = 1.5f;
            m.reflection = 1.0f;
        } else {
            m.color = Vec3(0.0f, 0.0f, 0.0f);
            m.emitColor = Vec3(rng.uniform(2.0f, 8.0f), rng.uniform(2.0f, 8.0f), rng.uniform(2.0f, 8.0f));
        }
        world->materials.push_back(m);
    }
    for (uint32_t i = 0; i < count; ++i) {
        Sphere s;
        s.radius = rng.uniform(0.15f, 0.45f);
        s.position = Vec3(rng.uniform(-extent, extent), s.radius + rng.uniform(0.0f, 3.0f), rng.uniform(-extent, extent));
        s.materialIndex = 2 + (uint32_t)(rng.next() % 62);
        world->spheres.push_back(s);
    }
    world->camera = Camera(Vec3(0.0f, 6.0f, 18.0f));
    return world;
}

World* CreateGlassSpheresScene(uint32_t grid) {
    World* world = new World;
    Plane ground;
    ground.normal = Vec3(0.0f, 1.0f, 0.0f);
    ground.d = 0;
    ground.materialIndex = 1;
    world->planes.push_back(ground);

    Material sky;
    sky.emitColor = Vec3(0.05f, 0.07f, 0.1f);
    Material floor_mat = Diffuse(0.75f, 0.75f, 0.75f);
    Material glass;                                   // material precedent: scene.h:305-308
    glass.color = Vec3(1.0f, 1.0f, 1.0f) * 0.9f;
    glass.refractiveIndex = 1.5f;
    glass.reflection = 1.0f;
    Material lamp;
    lamp.emitColor = Vec3(12.0f, 11.0f, 10.0f);
    world->materials = {sky, floor_mat, glass, lamp};

    const float pitch = 2.2f, radius = 0.9f;
    const float half = 0.5f * pitch * (float)(grid - 1);
    for (uint32_t iz = 0; iz < grid; ++iz)
        for (uint32_t ix = 0; ix < grid; ++ix) {
            Sphere s;
            s.position = Vec3((float)ix * pitch - half, radius, (float)iz * pitch - half);
            s.radius = radius;
            s.materialIndex = 2;
            world->spheres.push_back(s);
        }
    Sphere bulb;
    bulb.position = Vec3(0.0f, 7.0f, 0.0f);
    bulb.radius = 2.0f;
    bulb.materialIndex = 3;
    world->spheres.push_back(bulb);
    world->camera = Camera(Vec3(0.0f, 5.0f, 14.0f));
    return world;
}

// ---- RTSC v1 -----------------------------------------------------------------
namespace {
struct SceneFileHeader {
    char magic[4];
    uint32_t version, materialCount, planeCount, sphereCount, rectangleCount;
};
}  // namespace

bool WriteSceneFile(const World& world, const char* path) {
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    SceneFileHeader h = {{'R', 'T', 'S', 'C'}, 1, (uint32_t)world.materials.size(), (uint32_t)world.planes.size(),
                         (uint32_t)world.spheres.size(), (uint32_t)world.rectangles.size()};
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1 && std::fwrite(&world.camera, sizeof(Camera), 1, f) == 1;
    auto put = [&](const void* data, size_t size, size_t count) {
        if (count) ok = ok && std::fwrite(data, size, count, f) == count;
    };
    put(world.materials.data(), sizeof(Material), world.materials.size());
    put(world.planes.data(), sizeof(Plane), world.planes.size());
    put(world.spheres.data(), sizeof(Sphere), world.spheres.size());
    put(world.rectangles.data(), sizeof(RectangleXY), world.rectangles.size());
    return std::fclose(f) == 0 && ok;
}

World* ReadSceneFile(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return nullptr;
    SceneFileHeader h;
    World* world = nullptr;
    if (std::fread(&h, sizeof h, 1, f) == 1 && !std::memcmp(h.magic, "RTSC", 4) && h.version == 1 && h.materialCount >= 1 &&
        h.materialCount < (1u << 24) && h.planeCount < (1u << 24) && h.sphereCount < (1u << 24) && h.rectangleCount < (1u << 24)) {
        world = new World;
        world->materials.resize(h.materialCount);
        world->planes.resize(h.planeCount);
        world->spheres.resize(h.sphereCount);
        world->rectangles.resize(h.rectangleCount);
        bool ok = std::fread(&world->camera, sizeof(Camera), 1, f) == 1;
        auto get = [&](void* data, size_t size, size_t count) {
            if (count) ok = ok && std::fread(data, size, count, f) == count;
        };
        get(world->materials.data(), sizeof(Material), h.materialCount);
        get(world->planes.data(), sizeof(Plane), h.planeCount);
        get(world->spheres.data(), sizeof(Sphere), h.sphereCount);
        get(world->rectangles.data(), sizeof(RectangleXY), h.rectangleCount);
        auto index_ok = [&](uint32_t m) { return m < h.materialCount; };
        for (const Plane& p : world->planes) ok = ok && index_ok(p.materialIndex);
        for (const Sphere& s : world->spheres) ok = ok && index_ok(s.materialIndex);
        for (const RectangleXY& r : world->rectangles) ok = ok && index_ok(r.materialIndex);
        if (!ok) { delete world; world = nullptr; }
    }
    std::fclose(f);
    return world;
}

}  // namespace rth
