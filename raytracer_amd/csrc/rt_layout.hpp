// Device scene image: the word layout of the scene in device memory, and the host
// code that packs it from the caller's AoS arrays (rt_world, include/rt_hip.h).
//
// One image per intersection mode, each self-contained (header + sections).  The first
// H_LDS_WORDS words (header, planes, rectangles, materials) are staged in LDS by every
// workgroup with a flat 16-byte copy; the sphere records stay in global memory, from where
// every wave copies them, 64 at a time, into its own LDS slice (rt_kernels.hpp, intersect_world).
//
//   header   16 words  counts and section offsets (in words); words 12-15: sparsity classes of the first 64 rectangle slots
//   planes    8 words  nx ny nz d | material 0 0 0
//   rects    40 words per PAIR of rectangles (A, B), the two handled side by side:
//              A.r0x A.r1x A.r0y A.r1y | A.r0z A.r1z A.r0w A.r1w      rows 0,1 of A's inverse transform, interleaved
//              B.r0x B.r1x B.r0y B.r1y | B.r0z B.r1z B.r0w B.r1w      same for B
//              A.r2x B.r2x A.r2y B.r2y | A.r2z B.r2z A.r2w B.r2w      row 2 of A and of B, interleaved
//              A.nx A.ny A.nz A.tag    | B.nx B.ny B.nz B.tag         tag = material | wide_lane << 24
//              A.r0.c A.r1.c B.r0.c B.r1.c | A.r2.c B.r2.c 0 0            rows applied to the CAMERA position c: every
//                                                                     primary ray starts there, so a wave whose lanes all
//                                                                     trace primary rays reads these instead of computing them
//            an odd count is padded with an all-zero rectangle (t = -0/0 = NaN: never hit)
//   mats     12 words  color[3] ri emit[3] reflection | 1/ri r0 0 0
//   -------- end of the LDS part --------
//   spheres   4 words  cx cy cz r*r                      (16-byte aligned)
//   sph.tag   1 word   material | wide_lane << 24
//   cull     36 words  per rectangle SLOT (pair k: slots 2k, 2k+1): 18 doubles that let a wave decide whether any
//                      primary ray through its pixel tile can hit the rectangle (see cull_record below)
//
// RT_INTERSECT_WIDE8 order: the reference tests primitives 8 at a time, lane j of
// group g being primitive 8g+j, keeps a running closest hit PER LANE and finally
// takes the first strictly smaller lane (main.cpp:60-154).  The winner is therefore
// the hit with the smallest t, ties going to the lowest lane j, then the lowest
// group g.  Storing the primitives sorted by (j, g) turns that into a plain
// first-strictly-smaller scan.  RT_INTERSECT_SCALAR keeps the caller's order
// (main.cpp:182-239).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/rt_hip.h"

namespace rt {

enum : uint32_t {
    H_NPLANES = 0, H_NSPHERES, H_NRECTS, H_NMATS,
    H_OFF_PLANES, H_OFF_SPHERES, H_OFF_SPHERE_TAGS, H_OFF_RECTS, H_OFF_MATS, H_TOTAL_WORDS, H_LDS_WORDS, H_OFF_CULL,
    H_RECT_ABOUT_Y = 12, H_RECT_ABOUT_X = 14,   // two words each: bit k set = rectangle slot k (k < 64) has that sparsity
    H_WORDS = 16
};

// Sparsity classes of a rectangle's inverse transform.  Scenes are mostly built from rectangles that were
// scaled, turned about ONE coordinate axis and moved, and their matrices then hold exact zeros:
//   1, about y:  m01 = m10 = m12 = m21 = 0        2, about x:  m01 = m02 = m10 = m20 = 0        0: neither
// The product of an exact zero and a finite coordinate is a zero, and adding a zero changes at most the sign of a
// sum that is itself zero, which none of the rectangle test's comparisons can see: the kernel leaves such terms
// out (rt_kernels.hpp, rectangle loop).  Precondition: the ray's origin and direction are finite (0 * inf would
// be a NaN in the reference's full transform).  Hit points lie on the scene's surfaces, so this holds as long as
// those are of ordinary size; a rectangle whose matrix has entries above 2^40 or a row shorter than 2^-40 (an
// extent beyond 2^40) is left unclassified.
inline uint32_t rect_sparsity_class(const float (*m)[4]) {
    for (int row = 0; row < 3; ++row) {
        const double n = (double)m[row][0] * m[row][0] + (double)m[row][1] * m[row][1] + (double)m[row][2] * m[row][2];
        if (!(n >= 0x1.0p-80) || !(n <= 0x1.0p80) || !(std::fabs(m[row][3]) <= 0x1.0p40f)) return 0u;
    }
    if (m[0][1] == 0.0f && m[1][0] == 0.0f && m[1][2] == 0.0f && m[2][1] == 0.0f) return 1u;
    if (m[0][1] == 0.0f && m[0][2] == 0.0f && m[1][0] == 0.0f && m[2][0] == 0.0f) return 2u;
    return 0u;
}

constexpr uint32_t PLANE_WORDS = 8, SPHERE_WORDS = 4, RECT_PAIR_WORDS = 40, MAT_WORDS = 12;
constexpr uint32_t WIDE_LANES = 8;        // simd.h:17; the SSE branch of simd.h:20-26 is 4 wide (RT_INTERSECT_WIDE4)
constexpr uint32_t TAG_LANE_SHIFT = 24;   // tag = material | lane << 24
constexpr uint32_t TAG_MATERIAL_MASK = 0x00FFFFFFu;

inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// Cull record of one rectangle (doubles).  A primary ray has the un-normalised direction
// D(a, b) = D0 + a * camX + b * camY (D0 = film centre - camera position; a, b film-plane coordinates) and, in
// the rectangle's frame, the origin o' (the floats stored for camera rays) and the direction
// (u, v, w) = (r0.D, r1.D, r2.D), r0..r2 being the first three columns of the inverse transform's rows.  Each
// triple below is (alpha, beta, gamma) of a function alpha + beta a + gamma b:
//   W      w itself.  The rectangle is decided only where w keeps one sign over the tile's patch and
//          |w| >= GRAZING * reach (reach >= |D|, so no ray is within 1e-3, as a cosine, of parallel to the plane):
//          then the kernel's rounded d'z (error <= 4 ulp of |r2||d|) has that sign and is within 2e-4 of r2.d,
//          relatively.  o'z == 0 or o'z of w's sign: t = -o'z / d'z <= 0 fails `t > min_hit`, no hit.
//   RIGHT  (o'x - mx) w - u o'z, LEFT (o'x + mx) w - u o'z, ABOVE / BELOW the same with o'y, my, v.  Otherwise the
//          exact hit point (o'x - u o'z / w, o'y - v o'z / w) is a projective image of the patch with a
//          denominator of one sign, hence inside the convex hull of the patch corners' images, and
//          sign(w) * RIGHT > 0 at the corners says they all have x > mx = 1 + M, with
//          M = 1e-3 (1 + |o'x| + |r0|/|r2| |o'z|); the kernel's rounded hit point is within M / 4 of the exact one
//          (relative error 2e-4 of t, 4 ulp of |r0||d| on d'x, times |t| <= |o'z| / (1e-3 |r2|)), so it fails
//          |x| <= 1.  LEFT: all x < -mx; ABOVE / BELOW: the same for y.
//   GRAZING  1e-3 |r2|; negative: nothing can hit the rectangle (zero padding, or a zero third row: t is NaN or
//          infinite); +infinity: never decided (magnitudes outside [2^-40, 2^40], where the loop's shortened
//          division is not in its exact range).
//   OZ     o'z.
enum : uint32_t { CULL_W = 0, CULL_RIGHT = 3, CULL_LEFT = 6, CULL_ABOVE = 9, CULL_BELOW = 12, CULL_GRAZING = 15, CULL_OZ = 16, CULL_DOUBLES = 18 };

#if defined(__HIPCC__)
#define RT_HOST_DEVICE __host__ __device__
#else
#define RT_HOST_DEVICE
#endif

// The patch of the film through which the primary rays of the pixels [x0, x1] x [row0, row1] (global rows) leave the
// camera, as a box in film-plane coordinates (a along camX, b along camY): film coordinates as main.cpp:375-388
// forms them (in float, like the render: their rounding is four orders of magnitude below the widening), the
// quarter-pixel jitter, and the margin.
struct CullBox { double a_lo, a_hi, b_lo, b_hi; };
RT_HOST_DEVICE inline CullBox cull_box(uint32_t width, uint32_t height, float pixel_w, float pixel_h, float half_w, float half_h,
                                       float margin, uint32_t x0, uint32_t x1, uint32_t row0, uint32_t row1) {
    const float fx0 = ((float)(int32_t)x0 / (float)(int32_t)width) * 2.0f - 1.0f, fx1 = ((float)(int32_t)x1 / (float)(int32_t)width) * 2.0f - 1.0f;
    const float fy0 = ((float)(int32_t)row0 / (float)(int32_t)height) * -2.0f + 1.0f, fy1 = ((float)(int32_t)row1 / (float)(int32_t)height) * -2.0f + 1.0f;   // fy1 <= fy0
    CullBox box;
    box.a_lo = (double)((fx0 - pixel_w) * half_w - margin); box.a_hi = (double)((fx1 + pixel_w) * half_w + margin);
    box.b_lo = (double)((fy1 - pixel_h) * half_h - margin); box.b_hi = (double)((fy0 + pixel_h) * half_h + margin);
    return box;
}

// Must the rectangle of this cull record be tested for the primary rays of the box?  false only if no such ray can
// pass the rectangle test (see the record's description above).  A linear function takes its extremes over a box at
// the corners picked by the signs of its two slopes.
RT_HOST_DEVICE inline bool cull_keeps(const double* rec, const CullBox& box, double reach) {
    auto lowest = [&](const double* f) {
        return f[0] + (f[1] > 0.0 ? f[1] * box.a_lo : f[1] * box.a_hi) + (f[2] > 0.0 ? f[2] * box.b_lo : f[2] * box.b_hi);
    };
    auto highest = [&](const double* f) {
        return f[0] + (f[1] > 0.0 ? f[1] * box.a_hi : f[1] * box.a_lo) + (f[2] > 0.0 ? f[2] * box.b_hi : f[2] * box.b_lo);
    };
    const double grazing = rec[CULL_GRAZING];
    if (grazing < 0.0) return false;                       // a rectangle that nothing can hit (zero padding)
    // d'z = r2 . D keeps one sign over the patch and no ray is within 1e-3 (as a cosine) of parallel to the plane?
    const double limit = grazing * reach;
    const bool all_pos = lowest(rec + CULL_W) >= limit, all_neg = highest(rec + CULL_W) <= -limit;
    if (!(all_pos || all_neg)) return true;                // not decided here
    const double oz = rec[CULL_OZ];
    if (oz == 0.0 || (oz > 0.0) == all_pos) return false;  // t = -o'z / d'z <= 0 for every ray
    // In front: do all rays pass the plane beyond one edge of the square, margin included?
    const bool right = all_pos ? lowest(rec + CULL_RIGHT) > 0.0 : highest(rec + CULL_RIGHT) < 0.0;
    const bool left = all_pos ? highest(rec + CULL_LEFT) < 0.0 : lowest(rec + CULL_LEFT) > 0.0;
    const bool above = all_pos ? lowest(rec + CULL_ABOVE) > 0.0 : highest(rec + CULL_ABOVE) < 0.0;
    const bool below = all_pos ? highest(rec + CULL_BELOW) < 0.0 : lowest(rec + CULL_BELOW) > 0.0;
    return !(right || left || above || below);
}

// m: the rectangle's inverse transform; o: the stored camera-origin transform (o'x, o'y, o'z).
inline void cull_record(const float (*m)[4], const float o[3], const rt_camera& cam, double* rec) {
    const double ox = o[0], oy = o[1], oz = o[2];
    double r[3][3], n[3];
    for (int row = 0; row < 3; ++row) {
        for (int col = 0; col < 3; ++col) r[row][col] = (double)m[row][col];
        n[row] = r[row][0] * r[row][0] + r[row][1] * r[row][1] + r[row][2] * r[row][2];
    }
    for (uint32_t i = 0; i < CULL_DOUBLES; ++i) rec[i] = 0.0;
    rec[CULL_OZ] = oz;
    if (n[2] == 0.0) { rec[CULL_GRAZING] = -1.0; return; }
    const double big = 0x1.0p40, tiny = 0x1.0p-40;
    const bool in_range = n[0] <= big * big && n[1] <= big * big && n[2] <= big * big && n[2] >= 0x1.0p-60 && std::fabs(ox) <= big &&
                          std::fabs(oy) <= big && std::fabs(oz) <= big && (oz == 0.0 || std::fabs(oz) >= tiny);
    if (!in_range) { rec[CULL_GRAZING] = INFINITY; return; }
    rec[CULL_GRAZING] = 1.0e-3 * std::sqrt(n[2]);
    const double mx = 1.0 + 1.0e-3 * (1.0 + std::fabs(ox) + std::sqrt(n[0] / n[2]) * std::fabs(oz));
    const double my = 1.0 + 1.0e-3 * (1.0 + std::fabs(oy) + std::sqrt(n[1] / n[2]) * std::fabs(oz));
    // D0 = film centre - camera position, the film centre rounded as the render does it (main.cpp:358-359)
    double d0[3];
    for (int i = 0; i < 3; ++i) {
        const float film_center = cam.position[i] - cam.z_axis[i];
        d0[i] = (double)film_center - (double)cam.position[i];
    }
    auto put = [&](uint32_t at, double w_scale, const double* other, double other_scale) {   // c = w_scale r2 + other_scale other
        double c[3];
        for (int i = 0; i < 3; ++i) c[i] = w_scale * r[2][i] + (other ? other_scale * other[i] : 0.0);
        rec[at + 0] = c[0] * d0[0] + c[1] * d0[1] + c[2] * d0[2];
        rec[at + 1] = c[0] * cam.x_axis[0] + c[1] * cam.x_axis[1] + c[2] * cam.x_axis[2];
        rec[at + 2] = c[0] * cam.y_axis[0] + c[1] * cam.y_axis[1] + c[2] * cam.y_axis[2];
    };
    put(CULL_W, 1.0, nullptr, 0.0);
    put(CULL_RIGHT, ox - mx, r[0], -oz);
    put(CULL_LEFT, ox + mx, r[0], -oz);
    put(CULL_ABOVE, oy - my, r[1], -oz);
    put(CULL_BELOW, oy + my, r[1], -oz);
}

// Indices 0..n-1 sorted by (i % lanes, i / lanes).
inline std::vector<uint32_t> wide_order(uint32_t n, uint32_t lanes) {
    std::vector<uint32_t> order;
    order.reserve(n);
    for (uint32_t j = 0; j < lanes; ++j)
        for (uint32_t i = j; i < n; i += lanes) order.push_back(i);
    return order;
}

// wide_lanes: 0 = the scalar routine's image (index order, unfused arithmetic for the camera rows), else the
// lane width of the wide routine (8, or 4 for a LANE_WIDTH 4 build).
inline std::vector<uint32_t> pack_scene_image(const rt_world& w, uint32_t wide_lanes) {
    const bool wide = wide_lanes != 0;
    auto round4 = [](uint32_t v) { return (v + 3u) & ~3u; };
    uint32_t off_planes = H_WORDS;
    uint32_t off_rects = off_planes + PLANE_WORDS * w.plane_count;
    const uint32_t rect_slots = (w.rectangle_count + 1u) & ~1u;   // padded to whole pairs
    uint32_t off_mats = off_rects + RECT_PAIR_WORDS * (rect_slots / 2);
    uint32_t lds_words = round4(off_mats + MAT_WORDS * w.material_count);
    uint32_t off_spheres = lds_words;
    uint32_t off_tags = off_spheres + SPHERE_WORDS * w.sphere_count;
    uint32_t off_cull = round4(off_tags + w.sphere_count);          // doubles: 16-byte aligned
    uint32_t total = off_cull + 2u * CULL_DOUBLES * rect_slots;

    std::vector<uint32_t> img(total, 0u);
    img[H_NPLANES] = w.plane_count; img[H_NSPHERES] = w.sphere_count;
    img[H_NRECTS] = rect_slots; img[H_NMATS] = w.material_count;
    img[H_OFF_PLANES] = off_planes; img[H_OFF_SPHERES] = off_spheres; img[H_OFF_SPHERE_TAGS] = off_tags;
    img[H_OFF_RECTS] = off_rects; img[H_OFF_MATS] = off_mats; img[H_TOTAL_WORDS] = total; img[H_LDS_WORDS] = lds_words;
    img[H_OFF_CULL] = off_cull;
    for (uint32_t k = w.rectangle_count; k < rect_slots; ++k) {    // the padding slot: never hit
        double rec[CULL_DOUBLES] = {0.0};
        rec[CULL_GRAZING] = -1.0;
        std::memcpy(&img[off_cull + 2u * CULL_DOUBLES * k], rec, sizeof rec);
    }

    for (uint32_t i = 0; i < w.plane_count; ++i) {
        const rt_plane& p = w.planes[i];
        uint32_t* q = &img[off_planes + PLANE_WORDS * i];
        q[0] = f2u(p.normal[0]); q[1] = f2u(p.normal[1]); q[2] = f2u(p.normal[2]); q[3] = f2u(p.d);
        q[4] = p.material_index;
    }

    std::vector<uint32_t> sphere_order = wide ? wide_order(w.sphere_count, wide_lanes) : std::vector<uint32_t>();
    for (uint32_t k = 0; k < w.sphere_count; ++k) {
        uint32_t i = wide ? sphere_order[k] : k;
        const rt_sphere& s = w.spheres[i];
        uint32_t* q = &img[off_spheres + SPHERE_WORDS * k];
        q[0] = f2u(s.position[0]); q[1] = f2u(s.position[1]); q[2] = f2u(s.position[2]);
        q[3] = f2u(s.radius * s.radius);   // scene.h:274 / main.cpp:187: rounded once, then used
        img[off_tags + k] = s.material_index | ((wide ? i % wide_lanes : 0u) << TAG_LANE_SHIFT);
    }

    std::vector<uint32_t> rect_order = wide ? wide_order(w.rectangle_count, wide_lanes) : std::vector<uint32_t>();
    for (uint32_t k = 0; k < w.rectangle_count; ++k) {
        uint32_t i = wide ? rect_order[k] : k;
        const rt_rectangle& r = w.rectangles[i];
        uint32_t* q = &img[off_rects + RECT_PAIR_WORDS * (k / 2)];
        const uint32_t which = k & 1u;
        const auto& m = r.inverse_transform;             // the 4th row only feeds the unused w component
        for (int col = 0; col < 4; ++col) {
            uint32_t* rows01 = q + 8 * which + 4 * (col / 2) + 2 * (col % 2);
            rows01[0] = f2u(m[0][col]);
            rows01[1] = f2u(m[1][col]);
            q[16 + 4 * (col / 2) + 2 * (col % 2) + which] = f2u(m[2][col]);
        }
        // The camera position through rows 0..2, in the arithmetic of the mode's origin transform
        // (rt_kernels.hpp, rectangle loop): an fma chain for WIDE8, left-to-right products and sums otherwise.
        {
            const float* c = w.camera->position;
            float rows[3];
            for (int row = 0; row < 3; ++row)
                rows[row] = wide ? std::fmaf(m[row][0], c[0], std::fmaf(m[row][1], c[1], std::fmaf(m[row][2], c[2], m[row][3])))
                                 : m[row][0] * c[0] + m[row][1] * c[1] + m[row][2] * c[2] + m[row][3];
            q[32 + 2 * which] = f2u(rows[0]);
            q[33 + 2 * which] = f2u(rows[1]);
            q[36 + which] = f2u(rows[2]);
            double rec[CULL_DOUBLES];
            cull_record(m, rows, *w.camera, rec);
            std::memcpy(&img[off_cull + 2u * CULL_DOUBLES * k], rec, sizeof rec);
        }
        if (k < 64u) {
            const uint32_t cls = rect_sparsity_class(m);
            if (cls) img[(cls == 1u ? H_RECT_ABOUT_Y : H_RECT_ABOUT_X) + k / 32u] |= 1u << (k % 32u);
        }
        uint32_t* tail = q + 24 + 4 * which;
        tail[0] = f2u(r.normal[0]); tail[1] = f2u(r.normal[1]); tail[2] = f2u(r.normal[2]);
        tail[3] = r.material_index | ((wide ? i % wide_lanes : 0u) << TAG_LANE_SHIFT);
    }

    for (uint32_t i = 0; i < w.material_count; ++i) {
        const rt_material& m = w.materials[i];
        uint32_t* q = &img[off_mats + MAT_WORDS * i];
        q[0] = f2u(m.color[0]); q[1] = f2u(m.color[1]); q[2] = f2u(m.color[2]); q[3] = f2u(m.refractive_index);
        q[4] = f2u(m.emit_color[0]); q[5] = f2u(m.emit_color[1]); q[6] = f2u(m.emit_color[2]); q[7] = f2u(m.reflection);
        float ri = m.refractive_index;
        // Refract: `1.0 / refractiveIndex` is a double division narrowed to float (math_util.h:62).
        float inv_ri = ri != 0.0f ? (float)(1.0 / (double)ri) : 0.0f;
        // Schlick: r0 = ((1 - ri) / (1 + ri))^2 in float (math_util.h:90-91).
        float r0 = (1 - ri) / (1 + ri);
        r0 = r0 * r0;
        q[8] = f2u(inv_ri); q[9] = f2u(r0);
    }
    return img;
}

}  // namespace rt
