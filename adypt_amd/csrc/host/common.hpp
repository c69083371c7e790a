// Shared host-side records and small math of the scene/BVH producer side (CPU, no GPU needed).
// Layouts are the kernel's input contract (reference: src/Util/Shape.hpp:70-74, src/BVH/WideBVH.hpp:13-26,
// src/Tracer/OglScene.hpp:19-28).  Built with -ffp-contract=off: every float expression is evaluated as written.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace adypt {

struct Vec3 {
	float x, y, z;
	float &operator[](int i) { return (&x)[i]; }
	float operator[](int i) const { return (&x)[i]; }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
// component-wise select with the argument order of glm::min/max: min(x,y) = y<x ? y : x ; max(x,y) = x<y ? y : x
inline float fmin_glm(float x, float y) { return y < x ? y : x; }
inline float fmax_glm(float x, float y) { return x < y ? y : x; }
inline Vec3 vmin(Vec3 a, Vec3 b) { return {fmin_glm(a.x, b.x), fmin_glm(a.y, b.y), fmin_glm(a.z, b.z)}; }
inline Vec3 vmax(Vec3 a, Vec3 b) { return {fmax_glm(a.x, b.x), fmax_glm(a.y, b.y), fmax_glm(a.z, b.z)}; }

// Axis-aligned box, empty = (+FLT_MAX, -FLT_MAX)  (reference: src/Util/Shape.hpp:33-67)
struct Box {
	Vec3 lo{FLT_MAX, FLT_MAX, FLT_MAX}, hi{-FLT_MAX, -FLT_MAX, -FLT_MAX};
	Box() = default;
	Box(Vec3 l, Vec3 h) : lo(l), hi(h) {}
	static Box join(const Box &a, const Box &b) { return Box(vmin(a.lo, b.lo), vmax(a.hi, b.hi)); }
	void grow(Vec3 p) { lo = vmin(p, lo); hi = vmax(p, hi); }
	void grow(const Box &b) { lo = vmin(b.lo, lo); hi = vmax(b.hi, hi); }
	void clip(const Box &b) { lo = vmax(lo, b.lo); hi = vmin(hi, b.hi); }
	Vec3 center() const { return (lo + hi) * 0.5f; }
	Vec3 extent() const { return hi - lo; }
	float area() const { Vec3 e = extent(); return (e.x * (e.y + e.z) + e.y * e.z) * 2.0f; }
};

#pragma pack(push, 1)
struct TriRec {  // 100 B
	Vec3 p[3], n[3];
	float tc[3][2];
	int32_t matid;
	Box bounds() const { return Box(vmin(p[0], vmin(p[1], p[2])), vmax(p[0], vmax(p[1], p[2]))); }
};
struct MatRec {  // 64 B
	int32_t dtex; float dr, dg, db;
	int32_t etex; float er, eg, eb;
	int32_t stex; float sr, sg, sb;
	int32_t illum; float shininess, dissolve, ior;
};
struct NodeRec {  // 80 B
	float px, py, pz;
	uint8_t ex, ey, ez, imask;
	uint32_t child_base, tri_base;
	uint8_t meta[8], qlox[8], qloy[8], qloz[8], qhix[8], qhiy[8], qhiz[8];
};
#pragma pack(pop)
static_assert(sizeof(TriRec) == 100 && sizeof(MatRec) == 64 && sizeof(NodeRec) == 80, "record layouts");

struct TextureImage { int32_t w = 0, h = 0; std::vector<uint8_t> rgb; };

// binary SBVH node, one triangle reference per leaf (reference: src/BVH/SBVH.hpp:11-16; right child = index + 1)
struct BinNode { Box box; int32_t tri; int32_t left; };

void set_host_error(const std::string &msg);
bool decode_image_rgb8(const std::string &path, TextureImage *out, std::string *err);
bool decode_jpeg(const std::vector<uint8_t> &bytes, TextureImage *out, std::string *err); // jpeg_decoder.cpp: stb_image's pixels

}  // namespace adypt
