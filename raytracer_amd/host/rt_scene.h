// Host-side scene model: the reference's scene types and builders (scene.h),
// kept byte-compatible with the records of include/rt_hip.h so a World can be
// handed to rt_upload_world without conversion.
//
//   Material / Sphere / Plane / RectangleXY / Camera / World   scene.h:13-47, 163-191
//   CreateRectangle / RotateRectangle / CreateBox / RotateBox  scene.h:59-161
//   createScene (spheres + plane)                              scene.h:193-345
//   CreateCornellBoxScene                                      scene.h:347-469
// plus two generated scenes the reference cannot express without recompiling:
//   CreateRandomSpheresScene   N random spheres of mixed materials over a ground plane
//   CreateGlassSpheresScene    a lattice of glass spheres over a plane with one emitter
//
// The CPU-only AoSoA lane mirrors (SphereSoALane, RectangleLane) have no
// counterpart: the GPU library derives its own LDS image from the AoS arrays.
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_math.h"

namespace rth {

struct Material {
    Vec3 color{0.0f, 0.0f, 0.0f};
    float refractiveIndex = 0.0f;   // 0 = no refraction
    Vec3 emitColor{0.0f, 0.0f, 0.0f};
    float reflection = 0.0f;        // 0 = diffuse, 1 = mirror
};

struct Sphere {
    Vec3 position;
    float radius = 0.0f;
    uint32_t materialIndex = 0;
};

struct Plane {
    Vec3 normal;
    float d = 0.0f;
    uint32_t materialIndex = 0;
};

// Unit square [-1,1]^2 in the XY plane under transformMatrix; the scene builders
// store the INVERSE transform once construction is finished.
struct RectangleXY {
    Mat4 transformMatrix;
    Vec3 normal{0.0f, 0.0f, 1.0f};
    uint32_t materialIndex = 0;
};

struct Camera {
    Vec3 position, zVec, yVec, xVec;
    Camera() = default;
    explicit Camera(Vec3 cameraPosition);   // looks at the origin, +Y up (scene.h:169-174)
};

static_assert(sizeof(Material) == sizeof(rt_material), "Material layout");
static_assert(sizeof(Sphere) == sizeof(rt_sphere), "Sphere layout");
static_assert(sizeof(Plane) == sizeof(rt_plane), "Plane layout");
static_assert(sizeof(RectangleXY) == sizeof(rt_rectangle), "RectangleXY layout");
static_assert(sizeof(Camera) == sizeof(rt_camera), "Camera layout");

struct World {
    std::vector<Material> materials;
    std::vector<Plane> planes;
    std::vector<Sphere> spheres;
    std::vector<RectangleXY> rectangles;
    Camera camera;

    // View for rt_upload_world; valid while this World is alive and unmodified.
    rt_world View() const;
};

extern const Vec3 XAxis, YAxis, ZAxis;

RectangleXY CreateRectangle(Vec3 position, Vec3 scale, uint32_t materialIndex,
                            Vec3 initialRotationAxis = Vec3(0.0f, 0.0f, 0.0f), float initialRotationAngle = 0.0f);
void RotateRectangle(RectangleXY* rect, Vec3 axis, float angle);

struct Box {
    Vec3 position;
    RectangleXY rectangles[6];   // front, back, right, left, top, bottom
};
Box CreateBox(Vec3 position, Vec3 scale, uint32_t materialIndex);
void RotateBox(Box* box, Vec3 axis, float angle);

World* createScene();
World* CreateCornellBoxScene();

// count spheres (radius 0.15-0.45) with centres uniform in x,z in [-extent, extent] and resting
// between y = radius and y = radius + 3 over the plane y = 0; materials drawn 50 % diffuse,
// 15 % glossy (reflection 0.3-0.95), 15 % mirror, 15 % glass (ri 1.5), 5 % emissive; generator:
// splitmix64 from `seed`.  Camera at (0, 6, 18).
World* CreateRandomSpheresScene(uint32_t count, uint64_t seed, float extent = 12.0f);
// grid x grid glass spheres (ri 1.5, reflection 1) of radius 0.9 on a 2.2 pitch, resting on a
// diffuse plane, lit by one emissive sphere above the lattice.  Camera at (0, 5, 14).
World* CreateGlassSpheresScene(uint32_t grid = 5);

// RTSC v1 scene file (header, camera, materials, planes, spheres, rectangles).
bool WriteSceneFile(const World& world, const char* path);
World* ReadSceneFile(const char* path);

}  // namespace rth
