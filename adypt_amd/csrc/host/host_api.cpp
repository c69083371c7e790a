// Host-side C entry points around the scene/BVH producers plus the small restated host functions of the tracer's
// driver: Woop precompute, camera matrices, Sobol stream, per-pixel shift bytes, .bvh cache, OpenEXR output.
#include "builders.hpp"
#include "exact_sort.hpp"
#include "../../../include/adypt_hip.h"

#include <algorithm>
#include <cstdio>
#include <fstream>
#include <zlib.h>
#include <sched.h>
#include <thread>

using namespace adypt;

struct adypt_bvh {
	std::vector<NodeRec> nodes;
	std::vector<int32_t> tri_indices;
};

namespace {

// ---- glm::inverse(mat4) (dep/glm/detail/func_matrix.inl:294-351): cofactor expansion, column-major m[c*4+r] ----
void inverse4(const float *a, float *out)
{
	auto m = [&](int c, int r) { return a[c * 4 + r]; };
	float c00 = m(2, 2) * m(3, 3) - m(3, 2) * m(2, 3), c02 = m(1, 2) * m(3, 3) - m(3, 2) * m(1, 3), c03 = m(1, 2) * m(2, 3) - m(2, 2) * m(1, 3);
	float c04 = m(2, 1) * m(3, 3) - m(3, 1) * m(2, 3), c06 = m(1, 1) * m(3, 3) - m(3, 1) * m(1, 3), c07 = m(1, 1) * m(2, 3) - m(2, 1) * m(1, 3);
	float c08 = m(2, 1) * m(3, 2) - m(3, 1) * m(2, 2), c10 = m(1, 1) * m(3, 2) - m(3, 1) * m(1, 2), c11 = m(1, 1) * m(2, 2) - m(2, 1) * m(1, 2);
	float c12 = m(2, 0) * m(3, 3) - m(3, 0) * m(2, 3), c14 = m(1, 0) * m(3, 3) - m(3, 0) * m(1, 3), c15 = m(1, 0) * m(2, 3) - m(2, 0) * m(1, 3);
	float c16 = m(2, 0) * m(3, 2) - m(3, 0) * m(2, 2), c18 = m(1, 0) * m(3, 2) - m(3, 0) * m(1, 2), c19 = m(1, 0) * m(2, 2) - m(2, 0) * m(1, 2);
	float c20 = m(2, 0) * m(3, 1) - m(3, 0) * m(2, 1), c22 = m(1, 0) * m(3, 1) - m(3, 0) * m(1, 1), c23 = m(1, 0) * m(2, 1) - m(2, 0) * m(1, 1);
	const float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
	const float f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
	const float v0[4] = {m(1, 0), m(0, 0), m(0, 0), m(0, 0)}, v1[4] = {m(1, 1), m(0, 1), m(0, 1), m(0, 1)};
	const float v2[4] = {m(1, 2), m(0, 2), m(0, 2), m(0, 2)}, v3[4] = {m(1, 3), m(0, 3), m(0, 3), m(0, 3)};
	float inv[4][4];
	for(int i = 0; i < 4; ++i)
	{
		const float sa = (i & 1) ? -1.0f : 1.0f, sb = -sa;
		inv[0][i] = (v1[i] * f0[i] - v2[i] * f1[i] + v3[i] * f2[i]) * sa;
		inv[1][i] = (v0[i] * f0[i] - v2[i] * f3[i] + v3[i] * f4[i]) * sb;
		inv[2][i] = (v0[i] * f1[i] - v1[i] * f3[i] + v3[i] * f5[i]) * sa;
		inv[3][i] = (v0[i] * f2[i] - v1[i] * f4[i] + v2[i] * f5[i]) * sb;
	}
	float det = (m(0, 0) * inv[0][0] + m(0, 1) * inv[1][0]) + (m(0, 2) * inv[2][0] + m(0, 3) * inv[3][0]);
	float ood = 1.0f / det;
	for(int c = 0; c < 4; ++c) for(int r = 0; r < 4; ++r) out[c * 4 + r] = inv[c][r] * ood;
}

// glm::rotate (dep/glm/ext/matrix_transform.inl:18-46)
void rotate4(const float *m, float angle, const float axis[3], float *out)
{
	const float c = std::cos(angle), s = std::sin(angle);
	const float inv_len = 1.0f / std::sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
	const float ax[3] = {axis[0] * inv_len, axis[1] * inv_len, axis[2] * inv_len};
	const float t[3] = {(1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2]};
	float R[3][3];
	R[0][0] = c + t[0] * ax[0]; R[0][1] = t[0] * ax[1] + s * ax[2]; R[0][2] = t[0] * ax[2] - s * ax[1];
	R[1][0] = t[1] * ax[0] - s * ax[2]; R[1][1] = c + t[1] * ax[1]; R[1][2] = t[1] * ax[2] + s * ax[0];
	R[2][0] = t[2] * ax[0] + s * ax[1]; R[2][1] = t[2] * ax[1] - s * ax[0]; R[2][2] = c + t[2] * ax[2];
	for(int col = 0; col < 3; ++col)
		for(int r = 0; r < 4; ++r) out[col * 4 + r] = m[r] * R[col][0] + m[4 + r] * R[col][1] + m[8 + r] * R[col][2];
	for(int r = 0; r < 4; ++r) out[12 + r] = m[12 + r];
}

const uint32_t kSobolMatrices[64][32] = {
#include "sobol_table.inc"
};

// std::mt19937 (MT19937, 32-bit) restated: seeding by the 1812433253 recurrence, standard tempering
struct Mt19937 {
	uint32_t s[624];
	int idx;
	explicit Mt19937(uint32_t seed)
	{
		s[0] = seed;
		for(int i = 1; i < 624; ++i) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i;
		idx = 624;
	}
	uint32_t next()
	{
		if(idx >= 624)
		{
			for(int i = 0; i < 624; ++i)
			{
				uint32_t y = (s[i] & 0x80000000u) | (s[(i + 1) % 624] & 0x7fffffffu);
				s[i] = s[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
			}
			idx = 0;
		}
		uint32_t y = s[idx++];
		y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
		return y;
	}
};

// ---- OpenEXR (scanline, single part) --------------------------------------------------------------------------
uint16_t to_half(float f)
{
	// tinyexr's float_to_half_full (dep/tinyexr.h:7160): truncate, round up when the first dropped bit is set
	uint32_t u; memcpy(&u, &f, 4);
	uint32_t sign = u >> 31, exp = (u >> 23) & 0xff, man = u & 0x7fffff;
	uint16_t o = 0;
	if(exp == 0) o = 0;
	else if(exp == 255) o = (uint16_t)((31u << 10) | (man ? 0x200u : 0u));
	else
	{
		int ne = (int)exp - 127 + 15;
		if(ne >= 31) o = (uint16_t)(31u << 10);
		else if(ne <= 0)
		{
			if(14 - ne <= 24)
			{
				uint32_t m = man | 0x800000u;
				o = (uint16_t)(m >> (14 - ne));
				if((m >> (13 - ne)) & 1u) ++o;
			}
		}
		else
		{
			o = (uint16_t)(((uint32_t)ne << 10) | (man >> 13));
			if(man & 0x1000u) ++o;
		}
	}
	return (uint16_t)(o | (sign << 15));
}
float from_half(uint16_t h)
{
	uint32_t sign = (uint32_t)(h >> 15) << 31, exp = (h >> 10) & 31, man = h & 0x3ff, u;
	if(exp == 0)
	{
		if(man == 0) u = sign;
		else { int e = -1; do { ++e; man <<= 1; } while(!(man & 0x400)); u = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3ff) << 13; }
	}
	else if(exp == 31) u = sign | 0x7f800000u | man << 13;
	else u = sign | (exp + 112) << 23 | man << 13;
	float f; memcpy(&f, &u, 4);
	return f;
}

void put_attr(std::vector<uint8_t> *o, const char *name, const char *type, const void *data, int32_t size)
{
	o->insert(o->end(), name, name + strlen(name) + 1);
	o->insert(o->end(), type, type + strlen(type) + 1);
	const uint8_t *s = (const uint8_t *)&size;
	o->insert(o->end(), s, s + 4);
	o->insert(o->end(), (const uint8_t *)data, (const uint8_t *)data + size);
}

}  // namespace

extern "C" int adypt_host_get_threads(void);
// independent items over the host worker threads (contiguous ranges; small jobs stay on the calling thread)
template <class F> static void parallel_ranges(int64_t n, int64_t min_per_thread, F body)
{
	const int threads = (int)std::max<int64_t>(1, std::min<int64_t>(adypt_host_get_threads(), n / std::max<int64_t>(1, min_per_thread)));
	if(threads <= 1) { body((int64_t)0, n); return; }
	std::vector<std::thread> pool;
	for(int t = 1; t < threads; ++t) pool.emplace_back(body, n * t / threads, n * (t + 1) / threads);
	body((int64_t)0, n / threads);
	for(std::thread &th : pool) th.join();
}


extern "C" {

// ---------------------------------------------------------------------------------------------------------------
// worker threads of the BVH build: explicit setting > $ADYPT_BUILD_THREADS > the cores this process may use
// (affinity mask capped by the cgroup CPU quota)
static int g_host_threads = 0;

static int available_cores()
{
	int n = (int)std::thread::hardware_concurrency();
	cpu_set_t set;
	if(sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
	if(FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r"))
	{
		char quota[32] = {0};
		long long period = 0;
		if(fscanf(f, "%31s %lld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0)
			n = std::min(n, (int)((atoll(quota) + period / 2) / period));
		fclose(f);
	}
	return std::max(1, std::min(n, 64));
}

int adypt_host_set_threads(int n)
{
	if(n < 0) { set_host_error("adypt_host_set_threads: n must be >= 0 (0 = automatic)"); return ADYPT_E_INVALID; }
	g_host_threads = n;
	return ADYPT_OK;
}

int adypt_host_get_threads(void)
{
	if(g_host_threads > 0) return g_host_threads;
	if(const char *ev = getenv("ADYPT_BUILD_THREADS")) { int n = atoi(ev); if(n > 0) return std::min(n, 256); }
	return available_cores();
}

// exact_sort.hpp against std::sort: same bytes out for the same bytes in
namespace {
struct SortRec { float key; int32_t tri; uint32_t payload; };
bool sort_rec_less(const SortRec &l, const SortRec &r) { return l.key < r.key || (l.key == r.key && l.tri < r.tri); }

// McIlroy's adversary ("A Killer Adversary for Quicksort", 1999) run against std::sort: the keys it freezes make
// this library's introsort degenerate until it falls back to heap sort
std::vector<int> killer_keys(int n)
{
	std::vector<int> val((size_t)n, n), idx((size_t)n);
	int solid = 0, candidate = 0;
	for(int i = 0; i < n; ++i) idx[(size_t)i] = i;
	std::sort(idx.begin(), idx.end(), [&](int x, int y) {
		if(val[(size_t)x] == n && val[(size_t)y] == n) { if(x == candidate) val[(size_t)x] = solid++; else val[(size_t)y] = solid++; }
		if(val[(size_t)x] == n) candidate = x; else if(val[(size_t)y] == n) candidate = y;
		return val[(size_t)x] < val[(size_t)y];
	});
	return val;
}
}  // namespace

int adypt_host_selftest_sort(int64_t n, uint32_t seed, int pattern, int threads, int64_t min_task)
{
	if(n < 0 || n > ((int64_t)1 << 28) || threads < 1 || min_task < 1) { set_host_error("adypt_host_selftest_sort: bad argument"); return ADYPT_E_INVALID; }
	std::vector<SortRec> a((size_t)n);
	uint32_t x = seed * 2654435761u + 12345u;
	auto rnd = [&] { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
	std::vector<int> killer;
	if(pattern == 5) killer = killer_keys((int)n);
	for(int64_t i = 0; i < n; ++i)
	{
		SortRec &r = a[(size_t)i];
		r.payload = (uint32_t)i;
		switch(pattern)
		{
		case 0: r.key = (float)(rnd() >> 8); r.tri = (int32_t)(rnd() & 0xffff); break;               // mostly distinct
		case 1: r.key = (float)(rnd() % (uint32_t)std::max<int64_t>(1, n / 64)); r.tri = (int32_t)(rnd() & 3); break; // heavy ties
		case 2: r.key = (float)(i / 3); r.tri = 7; break;                                              // sorted, tied triples
		case 3: r.key = (float)((n - i) / 2); r.tri = (int32_t)(i & 1); break;                         // reversed
		case 4: r.key = 1.0f; r.tri = 0; break;                                                        // all equal
		case 5: r.key = (float)killer[(size_t)i]; r.tri = 0; break;                                    // quicksort killer
		default: set_host_error("adypt_host_selftest_sort: pattern must be 0..5"); return ADYPT_E_INVALID;
		}
	}
	std::vector<SortRec> b = a;
	std::sort(a.begin(), a.end(), sort_rec_less);
	bool (*less)(const SortRec &, const SortRec &) = sort_rec_less;
	ExactSort<SortRec, decltype(less)>(less, threads, min_task).sort(b.data(), b.data() + b.size());
	if(n && memcmp(a.data(), b.data(), (size_t)n * sizeof(SortRec)) != 0) { set_host_error("adypt_host_selftest_sort: permutation differs from std::sort"); return ADYPT_E_INVALID; }
	return ADYPT_OK;
}

int adypt_bvh_build(const adypt_scene *s, const adypt_bvh_params *p, adypt_bvh **out, adypt_build_info *info)
{
	if(!s || !p || !out) { set_host_error("adypt_bvh_build: null argument"); return ADYPT_E_INVALID; }
	const void *tp; int64_t nt = adypt_scene_triangles(s, &tp);
	if(nt <= 0) { set_host_error("adypt_bvh_build: scene has no triangles"); return ADYPT_E_INVALID; }
	Box box;
	{ float lo[3], hi[3]; adypt_scene_aabb(s, lo, hi); box = Box({lo[0], lo[1], lo[2]}, {hi[0], hi[1], hi[2]}); }
	std::vector<BinNode> bin;
	double sbvh_ms = 0, wide_ms = 0;
	int64_t leaves = build_sbvh((const TriRec *)tp, nt, box, *p, &bin, &sbvh_ms, adypt_host_get_threads());
	adypt_bvh *b = new adypt_bvh();
	build_wide_bvh(bin, leaves, *p, &b->nodes, &b->tri_indices, &wide_ms, adypt_host_get_threads());
	if(info) { info->sbvh_nodes = (int64_t)bin.size(); info->refs = leaves; info->wide_nodes = (int64_t)b->nodes.size(); info->sbvh_ms = sbvh_ms; info->wide_ms = wide_ms; }
	*out = b;
	return ADYPT_OK;
}

// `.bvh` cache (src/BVH/WideBVH.cpp:9-66): "CWBVH_1.0\0" | InstanceConfig::BVH (12 B) | u32 n | i32[n] | nodes to EOF
static const char kBvhMagic[] = "CWBVH_1.0";

int adypt_bvh_save(const adypt_bvh *b, const char *path, const adypt_bvh_params *p)
{
	if(!b || !path || !p) { set_host_error("adypt_bvh_save: null argument"); return ADYPT_E_INVALID; }
	std::ofstream os(path, std::ios::binary);
	if(!os.is_open()) { set_host_error(std::string("cannot write ") + path); return ADYPT_E_IO; }
	os.write(kBvhMagic, sizeof(kBvhMagic));
	os.write((const char *)p, sizeof(*p));
	uint32_t n = (uint32_t)b->tri_indices.size();
	os.write((const char *)&n, 4);
	os.write((const char *)b->tri_indices.data(), (std::streamsize)(b->tri_indices.size() * 4));
	os.write((const char *)b->nodes.data(), (std::streamsize)(b->nodes.size() * sizeof(NodeRec)));
	return os.good() ? ADYPT_OK : ADYPT_E_IO;
}

int adypt_bvh_load(const char *path, const adypt_bvh_params *expected, adypt_bvh **out)
{
	if(!path || !out) { set_host_error("adypt_bvh_load: null argument"); return ADYPT_E_INVALID; }
	std::ifstream is(path, std::ios::binary);
	if(!is.is_open()) { set_host_error(std::string("cannot open ") + path); return ADYPT_E_IO; }
	std::vector<char> buf((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
	const size_t hdr = sizeof(kBvhMagic) + sizeof(adypt_bvh_params) + 4;
	if(buf.size() < hdr || memcmp(buf.data(), kBvhMagic, sizeof(kBvhMagic)) != 0) { set_host_error("not a CWBVH_1.0 file"); return ADYPT_E_PARSE; }
	adypt_bvh_params got;
	memcpy(&got, buf.data() + sizeof(kBvhMagic), sizeof(got));
	if(expected && (got.node_sah != expected->node_sah || got.triangle_sah != expected->triangle_sah || got.max_spatial_depth != expected->max_spatial_depth))
	{ set_host_error("bvh cache was built with different parameters"); return ADYPT_E_STATE; }
	uint32_t n; memcpy(&n, buf.data() + sizeof(kBvhMagic) + sizeof(got), 4);
	if(hdr + (size_t)n * 4 > buf.size()) { set_host_error("truncated bvh cache"); return ADYPT_E_PARSE; }
	adypt_bvh *b = new adypt_bvh();
	b->tri_indices.resize(n);
	if(n) memcpy(b->tri_indices.data(), buf.data() + hdr, (size_t)n * 4);
	size_t rest = buf.size() - hdr - (size_t)n * 4;
	b->nodes.resize(rest / sizeof(NodeRec));
	if(!b->nodes.empty()) memcpy(b->nodes.data(), buf.data() + hdr + (size_t)n * 4, b->nodes.size() * sizeof(NodeRec));
	*out = b;
	return ADYPT_OK;
}

void adypt_bvh_free(adypt_bvh *b) { delete b; }
int64_t adypt_bvh_nodes(const adypt_bvh *b, const void **nodes) { if(nodes) *nodes = b->nodes.data(); return (int64_t)b->nodes.size(); }
int64_t adypt_bvh_tri_indices(const adypt_bvh *b, const int32_t **idx) { if(idx) *idx = b->tri_indices.data(); return (int64_t)b->tri_indices.size(); }

// ---------------------------------------------------------------------------------------------------------------
void adypt_woop_matrices(const void *tris_, const int32_t *tri_indices, int64_t n_refs, float *out)
{
	const TriRec *tris = (const TriRec *)tris_;
	parallel_ranges(n_refs, 1 << 16, [=](int64_t begin, int64_t end) {
	for(int64_t i = begin; i < end; ++i)
	{
		const TriRec &t = tris[tri_indices[i]];
		Vec3 e0 = t.p[0] - t.p[2], e1 = t.p[1] - t.p[2];
		Vec3 n = {e0.y * e1.z - e1.y * e0.z, e0.z * e1.x - e1.z * e0.x, e0.x * e1.y - e1.x * e0.y};
		const float A[16] = {e0.x, e1.x, n.x, t.p[2].x, e0.y, e1.y, n.y, t.p[2].y, e0.z, e1.z, n.z, t.p[2].z, 0.0f, 0.0f, 0.0f, 1.0f};
		float inv[16];
		inverse4(A, inv);
		float *o = out + i * 12;
		o[0] = inv[8]; o[1] = inv[9]; o[2] = inv[10]; o[3] = -inv[11];
		memcpy(o + 4, inv, 8 * sizeof(float));
	}
	});
}

void adypt_camera_matrices(float fov, float yaw, float pitch, int width, int height, float inv_proj[16], float inv_view[16])
{
	const float kDeg = 0.01745329251994329576923690768489f;
	const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
	const float ax_x[3] = {1, 0, 0}, ax_y[3] = {0, 1, 0};
	float r1[16], view[16];
	rotate4(ident, (-pitch) * kDeg, ax_x, r1);
	rotate4(r1, (-yaw) * kDeg, ax_y, view);
	// glm::tweakedInfinitePerspective(fovy, aspect, 0.01f) with ep = FLT_EPSILON (dep/glm/ext/matrix_clip_space.inl:512-533)
	const float fovy = fov * kDeg, aspect = width / (float)height, zn = 0.01f, ep = FLT_EPSILON;
	const float range = std::tan(fovy / 2.0f) * zn;
	const float left = -range * aspect, right = range * aspect, bottom = -range, top = range;
	float proj[16] = {0};
	proj[0] = (2.0f * zn) / (right - left);
	proj[5] = (2.0f * zn) / (top - bottom);
	proj[10] = ep - 1.0f;
	proj[11] = -1.0f;
	proj[14] = (ep - 2.0f) * zn;
	inverse4(proj, inv_proj);
	inverse4(view, inv_view);
}

// 8-bit RGBA -> PNG (colour type 6, one zlib stream, filter 0 on every row): the headless stand-in for the window
int adypt_save_png(const char *path, const uint8_t *rgba8, int width, int height)
{
	if(!path || !rgba8 || width <= 0 || height <= 0) { set_host_error("adypt_save_png: bad argument"); return ADYPT_E_INVALID; }
	std::vector<uint8_t> raw((size_t)height * ((size_t)width * 4 + 1));
	for(int y = 0; y < height; ++y)
	{
		uint8_t *row = raw.data() + (size_t)y * ((size_t)width * 4 + 1);
		row[0] = 0;
		memcpy(row + 1, rgba8 + (size_t)y * width * 4, (size_t)width * 4);
	}
	uLongf zlen = compressBound((uLong)raw.size());
	std::vector<uint8_t> z(zlen);
	if(compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { set_host_error("adypt_save_png: zlib failed"); return ADYPT_E_IO; }
	std::ofstream os(path, std::ios::binary);
	if(!os.is_open()) { set_host_error(std::string("cannot write ") + path); return ADYPT_E_IO; }
	auto be32 = [](uint8_t *o, uint32_t v) { o[0] = (uint8_t)(v >> 24); o[1] = (uint8_t)(v >> 16); o[2] = (uint8_t)(v >> 8); o[3] = (uint8_t)v; };
	auto chunk = [&](const char *type, const uint8_t *data, size_t n) {
		uint8_t hdr[8];
		be32(hdr, (uint32_t)n); memcpy(hdr + 4, type, 4);
		os.write((const char *)hdr, 8);
		if(n) os.write((const char *)data, (std::streamsize)n);
		uLong crc = crc32(0L, (const Bytef *)type, 4);
		if(n) crc = crc32(crc, data, (uInt)n);
		uint8_t tail[4]; be32(tail, (uint32_t)crc);
		os.write((const char *)tail, 4);
	};
	static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
	os.write((const char *)sig, 8);
	uint8_t ihdr[13];
	be32(ihdr, (uint32_t)width); be32(ihdr + 4, (uint32_t)height);
	ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
	chunk("IHDR", ihdr, 13);
	chunk("IDAT", z.data(), (size_t)zlen);
	chunk("IEND", nullptr, 0);
	return os.good() ? ADYPT_OK : ADYPT_E_IO;
}

// Camera::Control (src/Tracer/Camera.cpp:25-59) + move_forward (Camera.hpp:19-24), driven by explicit input state
void adypt_camera_control(adypt_config *cfg, uint32_t keys, float mouse_dx, float mouse_dy, float frame_seconds)
{
	if(!cfg) return;
	const float kDeg = 0.01745329251994329576923690768489f; // glm::radians
	const float speed = frame_seconds * cfg->speed;
	auto move_forward = [&](float dist, float dir) {
		const float rad = kDeg * (cfg->yaw + dir);
		cfg->position[0] -= std::sin(rad) * dist;
		cfg->position[2] -= std::cos(rad) * dist;
	};
	if(keys & ADYPT_KEY_W) move_forward(speed, 0.0f);
	if(keys & ADYPT_KEY_A) move_forward(speed, 90.0f);
	if(keys & ADYPT_KEY_D) move_forward(speed, -90.0f);
	if(keys & ADYPT_KEY_S) move_forward(speed, 180.0f);
	if(keys & ADYPT_KEY_SPACE) cfg->position[1] += speed;
	if(keys & ADYPT_KEY_LEFT_SHIFT) cfg->position[1] -= speed;
	if(mouse_dx != 0.0f || mouse_dy != 0.0f)
	{
		cfg->yaw -= mouse_dx * cfg->mouse_sensitive;
		cfg->pitch -= mouse_dy * cfg->mouse_sensitive;
		cfg->pitch = fmin_glm(fmax_glm(cfg->pitch, -90.0f), 90.0f);          // glm::clamp = min(max(x, lo), hi)
		cfg->yaw = cfg->yaw - 360.0f * std::floor(cfg->yaw / 360.0f);         // glm::mod(x, y) = x - y * floor(x / y)
	}
}

int adypt_sobol_points(int dim, int first, int n, float *out)
{
	if(dim < 0 || dim > 64 || first < 0 || n < 0 || !out) { set_host_error("adypt_sobol_points: dim must be <= 64"); return ADYPT_E_INVALID; }
	uint32_t x[64] = {0};
	for(int idx = 0; idx < first + n; ++idx)
	{
		unsigned c = 0; // position of the lowest zero bit of the frame index (gray-code order)
		while(c < 31 && (((unsigned)idx >> c) & 1u)) ++c;
		for(int j = 0; j < dim; ++j)
		{
			x[j] ^= kSobolMatrices[j][c];
			if(idx >= first) out[(size_t)(idx - first) * dim + j] = (float)(x[j] / 4294967296.0);
		}
	}
	return ADYPT_OK;
}

void adypt_shift_bytes(uint32_t seed, int width, int height, uint8_t *out)
{
	Mt19937 gen(seed);
	const size_t n = (size_t)width * height * 2;
	for(size_t i = 0; i < n; ++i) out[i] = (uint8_t)gen.next();
}

// ---------------------------------------------------------------------------------------------------------------
int adypt_save_exr(const char *path, const float *rgb, int width, int height, int save_as_fp16)
{
	if(!path || !rgb || width <= 0 || height <= 0) { set_host_error("adypt_save_exr: bad argument"); return ADYPT_E_INVALID; }
	const bool zip = !(width < 16 && height < 16); // tinyexr SaveEXR: no compression for tiny images
	const int ptype = save_as_fp16 ? 1 : 2, bpc = save_as_fp16 ? 2 : 4;
	std::vector<uint8_t> hdr = {0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0};
	{
		std::vector<uint8_t> ch;
		for(const char *name : {"B", "G", "R"})
		{
			ch.push_back((uint8_t)name[0]); ch.push_back(0);
			int32_t rec[4] = {ptype, 0, 1, 1}; // pixel type, pLinear + 3 reserved bytes, xSampling, ySampling
			ch.insert(ch.end(), (uint8_t *)rec, (uint8_t *)rec + 16);
		}
		ch.push_back(0);
		put_attr(&hdr, "channels", "chlist", ch.data(), (int32_t)ch.size());
		uint8_t comp = zip ? 3 : 0;
		put_attr(&hdr, "compression", "compression", &comp, 1);
		int32_t win[4] = {0, 0, width - 1, height - 1};
		put_attr(&hdr, "dataWindow", "box2i", win, 16);
		put_attr(&hdr, "displayWindow", "box2i", win, 16);
		uint8_t lo = 0;
		put_attr(&hdr, "lineOrder", "lineOrder", &lo, 1);
		float one = 1.0f, center[2] = {0.0f, 0.0f};
		put_attr(&hdr, "pixelAspectRatio", "float", &one, 4);
		put_attr(&hdr, "screenWindowCenter", "v2f", center, 8);
		put_attr(&hdr, "screenWindowWidth", "float", &one, 4);
		hdr.push_back(0);
	}
	const int lines_per_block = zip ? 16 : 1;
	const int n_blocks = (height + lines_per_block - 1) / lines_per_block;
	std::vector<std::vector<uint8_t>> blocks((size_t)n_blocks);
	const size_t line_bytes = (size_t)width * 3 * bpc;
	std::vector<uint8_t> raw, tmp;
	for(int b = 0; b < n_blocks; ++b)
	{
		const int y0 = b * lines_per_block, y1 = std::min(height, y0 + lines_per_block);
		raw.resize(line_bytes * (size_t)(y1 - y0));
		for(int y = y0; y < y1; ++y)
			for(int c = 0; c < 3; ++c) // B, G, R planes
			{
				uint8_t *dst = raw.data() + line_bytes * (size_t)(y - y0) + (size_t)c * width * bpc;
				const float *src = rgb + (size_t)y * width * 3 + (2 - c);
				if(save_as_fp16) for(int x = 0; x < width; ++x) { uint16_t h = to_half(src[(size_t)x * 3]); memcpy(dst + (size_t)x * 2, &h, 2); }
				else for(int x = 0; x < width; ++x) memcpy(dst + (size_t)x * 4, &src[(size_t)x * 3], 4);
			}
		std::vector<uint8_t> &out = blocks[(size_t)b];
		if(!zip) { out = raw; continue; }
		// OpenEXR ZIP: de-interleave even/odd bytes, delta-predict, deflate; stored raw if that does not shrink it
		tmp.resize(raw.size());
		{
			uint8_t *t1 = tmp.data(), *t2 = tmp.data() + (raw.size() + 1) / 2;
			for(size_t i = 0; i < raw.size(); ++i) { if(i & 1) *t2++ = raw[i]; else *t1++ = raw[i]; }
			int p = tmp[0];
			for(size_t i = 1; i < tmp.size(); ++i) { int d = (int)tmp[i] - p + (128 + 256); p = tmp[i]; tmp[i] = (uint8_t)d; }
		}
		uLongf clen = compressBound((uLong)tmp.size());
		out.resize(clen);
		if(compress2(out.data(), &clen, tmp.data(), (uLong)tmp.size(), Z_DEFAULT_COMPRESSION) != Z_OK) { set_host_error("zlib compress failed"); return ADYPT_E_IO; }
		if(clen >= raw.size()) out = raw; else out.resize(clen);
	}
	FILE *f = fopen(path, "wb");
	if(!f) { set_host_error(std::string("cannot write ") + path); return ADYPT_E_IO; }
	fwrite(hdr.data(), 1, hdr.size(), f);
	uint64_t off = hdr.size() + (uint64_t)n_blocks * 8;
	for(int b = 0; b < n_blocks; ++b) { fwrite(&off, 8, 1, f); off += 8 + blocks[(size_t)b].size(); }
	for(int b = 0; b < n_blocks; ++b)
	{
		int32_t y = b * lines_per_block, sz = (int32_t)blocks[(size_t)b].size();
		fwrite(&y, 4, 1, f); fwrite(&sz, 4, 1, f);
		fwrite(blocks[(size_t)b].data(), 1, blocks[(size_t)b].size(), f);
	}
	fclose(f);
	return ADYPT_OK;
}

int adypt_load_exr(const char *path, float **rgb_out, int *width, int *height)
{
	std::ifstream is(path, std::ios::binary);
	if(!is.is_open()) { set_host_error(std::string("cannot open ") + path); return ADYPT_E_IO; }
	std::vector<uint8_t> b((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
	if(b.size() < 8 || b[0] != 0x76 || b[1] != 0x2f || b[2] != 0x31 || b[3] != 0x01) { set_host_error("not an EXR file"); return ADYPT_E_PARSE; }
	size_t pos = 8;
	int comp = -1, w = 0, h = 0;
	struct Ch { std::string name; int type; };
	std::vector<Ch> chans;
	while(pos < b.size() && b[pos] != 0)
	{
		std::string name((const char *)&b[pos]); pos += name.size() + 1;
		std::string type((const char *)&b[pos]); pos += type.size() + 1;
		int32_t size; memcpy(&size, &b[pos], 4); pos += 4;
		if(name == "channels")
		{
			size_t q = pos;
			while(b[q] != 0)
			{
				Ch c; c.name = (const char *)&b[q]; q += c.name.size() + 1;
				int32_t t; memcpy(&t, &b[q], 4); c.type = t; q += 16;
				chans.push_back(c);
			}
		}
		else if(name == "compression") comp = b[pos];
		else if(name == "dataWindow") { int32_t win[4]; memcpy(win, &b[pos], 16); w = win[2] - win[0] + 1; h = win[3] - win[1] + 1; }
		pos += (size_t)size;
	}
	++pos;
	if((comp != 0 && comp != 3 && comp != 2) || w <= 0 || h <= 0 || chans.empty()) { set_host_error("unsupported EXR variant"); return ADYPT_E_PARSE; }
	const int lpb = comp == 3 ? 16 : 1, n_blocks = (h + lpb - 1) / lpb;
	size_t line_bytes = 0;
	for(const Ch &c : chans) line_bytes += (size_t)w * (c.type == 1 ? 2 : 4);
	float *rgb = (float *)calloc((size_t)w * h * 3, sizeof(float));
	std::vector<uint8_t> raw, tmp;
	for(int k = 0; k < n_blocks; ++k)
	{
		uint64_t off; memcpy(&off, &b[pos + (size_t)k * 8], 8);
		int32_t y0, sz; memcpy(&y0, &b[off], 4); memcpy(&sz, &b[off + 4], 4);
		const int lines = std::min(lpb, h - y0);
		raw.resize(line_bytes * (size_t)lines);
		if((size_t)sz == raw.size()) memcpy(raw.data(), &b[off + 8], raw.size());
		else
		{
			tmp.resize(raw.size());
			uLongf n = (uLongf)tmp.size();
			if(uncompress(tmp.data(), &n, &b[off + 8], (uLong)sz) != Z_OK || n != tmp.size()) { free(rgb); set_host_error("EXR inflate failed"); return ADYPT_E_PARSE; }
			for(size_t i = 1; i < tmp.size(); ++i) tmp[i] = (uint8_t)(tmp[i - 1] + tmp[i] - 128);
			const uint8_t *t1 = tmp.data(), *t2 = tmp.data() + (tmp.size() + 1) / 2;
			for(size_t i = 0; i < raw.size(); ++i) raw[i] = (i & 1) ? *t2++ : *t1++;
		}
		for(int l = 0; l < lines; ++l)
		{
			const uint8_t *p = raw.data() + line_bytes * (size_t)l;
			for(const Ch &c : chans)
			{
				int dst = c.name == "R" ? 0 : c.name == "G" ? 1 : c.name == "B" ? 2 : -1;
				for(int x = 0; x < w; ++x)
				{
					float v;
					if(c.type == 1) { uint16_t hv; memcpy(&hv, p, 2); v = from_half(hv); p += 2; }
					else { memcpy(&v, p, 4); p += 4; }
					if(dst >= 0) rgb[((size_t)(y0 + l) * w + x) * 3 + dst] = v;
				}
			}
		}
	}
	*rgb_out = rgb; *width = w; *height = h;
	return ADYPT_OK;
}

int adypt_load_image_rgb8(const char *path, uint8_t **rgb, int32_t *width, int32_t *height)
{
	if(!path || !rgb || !width || !height) { set_host_error("adypt_load_image_rgb8: null argument"); return ADYPT_E_INVALID; }
	adypt::TextureImage img;
	std::string err;
	try
	{
		if(!adypt::decode_image_rgb8(path, &img, &err)) { set_host_error(std::string(path) + ": " + err); return ADYPT_E_PARSE; }
	}
	catch(const std::bad_alloc &) { set_host_error(std::string(path) + ": out of memory while decoding"); return ADYPT_E_OOM; } // nothing may unwind across the C ABI
	catch(const std::exception &e) { set_host_error(std::string(path) + ": " + e.what()); return ADYPT_E_PARSE; }
	*rgb = (uint8_t *)malloc(std::max<size_t>(1, img.rgb.size()));
	if(!*rgb) { set_host_error("out of memory"); return ADYPT_E_OOM; }
	memcpy(*rgb, img.rgb.data(), img.rgb.size());
	*width = img.w; *height = img.h;
	return ADYPT_OK;
}

void adypt_free(void *p) { free(p); }

}  // extern "C"
