// Host-side vector/matrix arithmetic for scene construction.
//
// Scene matrices must come out bit-identical to the reference's, so every
// operation here rounds in the same order as its counterpart:
//   dot / normalize / cross        math_vector.h:153-171, 341-343
//   Mat4 * Vec4, Mat4 * Mat4       math_matrix.h:47-70
//   inverse (adjugate, double det) math_matrix.h:82-158
//   scale / translate / rotate     math_matrix.h:161-192
// Build with -ffp-contract=off (raytracer_amd/host/Makefile).
#pragma once

#include <cmath>
#include <cstdint>

namespace rth {

struct Vec3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
    Vec3() = default;
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};

inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, float f) { return {a.x * f, a.y * f, a.z * f}; }
inline bool operator==(Vec3 a, Vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline float Dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 Cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Vec3 Normalize(Vec3 v) {
    const float f = 1 / sqrtf(Dot(v, v));
    return {v.x * f, v.y * f, v.z * f};
}

struct Vec4 {
    float x = 0.0f, y = 0.0f, z = 0.0f, w = 0.0f;
    Vec4() = default;
    Vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    Vec4(Vec3 v, float w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
    Vec3 xyz() const { return {x, y, z}; }
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
inline float Dot(Vec4 a, Vec4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

struct Mat4 {
    Vec4 row[4];
    Vec4& operator[](int i) { return row[i]; }
    const Vec4& operator[](int i) const { return row[i]; }
    static Mat4 Identity() {
        Mat4 m;
        for (int i = 0; i < 4; ++i) m[i][i] = 1.0f;
        return m;
    }
};

inline Vec4 operator*(const Mat4& m, Vec4 v) { return {Dot(m[0], v), Dot(m[1], v), Dot(m[2], v), Dot(m[3], v)}; }

inline Mat4 operator*(const Mat4& a, const Mat4& b) {
    Mat4 out;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out[r][c] = Dot(a[r], Vec4(b[0][c], b[1][c], b[2][c], b[3][c]));
    return out;
}

// Adjugate inverse.  The reference spells out 18 2x2 and 16 3x3 sub-determinants;
// they all follow two patterns, generated here from index tables:
//   d2(r0 r1 | c0 c1)    = m[r0][c0]*m[r1][c1] - m[r0][c1]*m[r1][c0]
//   d3(r0 r1 r2 | c0 c1 c2) = m[r0][c0]*d2(r1 r2|c1 c2) - m[r0][c1]*d2(r1 r2|c0 c2) + m[r0][c2]*d2(r1 r2|c0 c1)
// all in float, left to right; the determinant is widened to double, 1.0f/det and the
// final products are double, narrowed on store.
inline Mat4 Inverse(const Mat4& m) {
    auto d2 = [&](int r0, int r1, int c0, int c1) -> float { return m[r0][c0] * m[r1][c1] - m[r0][c1] * m[r1][c0]; };
    auto d3 = [&](const int r[3], const int c[3]) -> float {
        return m[r[0]][c[0]] * d2(r[1], r[2], c[1], c[2]) - m[r[0]][c[1]] * d2(r[1], r[2], c[0], c[2]) +
               m[r[0]][c[2]] * d2(r[1], r[2], c[0], c[1]);
    };
    static const int kCols[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};   // columns left after dropping i
    static const int kRows[4][3] = {{2, 1, 3}, {2, 0, 3}, {3, 0, 1}, {2, 0, 1}};   // expansion row first
    static const float kColumnSign[4] = {-1.0f, 1.0f, 1.0f, -1.0f};

    const float c0 = d3(kRows[3], kCols[0]), c1 = d3(kRows[3], kCols[1]);
    const float c2 = d3(kRows[3], kCols[2]), c3 = d3(kRows[3], kCols[3]);
    const double det = (-c0 * m[3][0] + c1 * m[3][1] - c2 * m[3][2] + c3 * m[3][3]);
    const double inv_det = 1.0f / det;

    Mat4 out;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            float minor = d3(kRows[j], kCols[i]);
            float sign = (i & 1) ? -kColumnSign[j] : kColumnSign[j];
            out[i][j] = (float)((sign < 0 ? -minor : minor) * inv_det);
        }
    return out;
}

inline void ScaleMatrix(Mat4& m, Vec3 s) { m[0][0] *= s.x; m[1][1] *= s.y; m[2][2] *= s.z; }
inline void TranslateMatrix(Mat4& m, Vec3 t) { m[0][3] += t.x; m[1][3] += t.y; m[2][3] += t.z; }
// The rotations overwrite the four entries of their plane (they are only ever applied to the identity).
inline void RotateMatrixXAxis(Mat4& m, float rad) { m[1][1] = cosf(rad); m[1][2] = -sinf(rad); m[2][1] = sinf(rad); m[2][2] = cosf(rad); }
inline void RotateMatrixYAxis(Mat4& m, float rad) { m[0][0] = cosf(rad); m[0][2] = sinf(rad); m[2][0] = -sinf(rad); m[2][2] = cosf(rad); }
inline void RotateMatrixZAxis(Mat4& m, float rad) { m[0][0] = cosf(rad); m[0][1] = -sinf(rad); m[1][0] = sinf(rad); m[1][1] = cosf(rad); }

}  // namespace rth
