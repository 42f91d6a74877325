// Small host-side value types used by the scene tools (OBJ loader, BVH builder,
// ray generators).  Plays the role of the reference's src/driver/{float3,bbox,tri}.h.
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>

namespace rodent {

struct V3 {
    float x = 0, y = 0, z = 0;
    V3() = default;
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit V3(float a) : x(a), y(a), z(a) {}
    float  operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& operator[](int i)       { return i == 0 ? x : (i == 1 ? y : z); }
};
struct V2 { float x = 0, y = 0; };

inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return a * s; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3& operator+=(V3& a, V3 b) { a = a + b; return a; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a * (1.0f / length(a)); }
inline V3 vmin(V3 a, V3 b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }

struct Box {
    V3 lo{FLT_MAX, FLT_MAX, FLT_MAX}, hi{-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void grow(V3 p) { lo = vmin(lo, p); hi = vmax(hi, p); }
    void grow(const Box& b) { lo = vmin(lo, b.lo); hi = vmax(hi, b.hi); }
    void clip(const Box& b) { lo = vmax(lo, b.lo); hi = vmin(hi, b.hi); }
    bool empty() const { return lo.x > hi.x || lo.y > hi.y || lo.z > hi.z; }
    float half_area() const {
        const float kx = std::max(hi.x - lo.x, 0.0f), ky = std::max(hi.y - lo.y, 0.0f), kz = std::max(hi.z - lo.z, 0.0f);
        return kx * (ky + kz) + ky * kz;
    }
};

struct Triangle { V3 v0, v1, v2; };

} // namespace rodent
