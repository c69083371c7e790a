// =====================================================================================================
// TEST INFRASTRUCTURE — CPU restatement ("oracle") of Adypt's GPU hot path.  NOT product code.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so; the
// product (adypt_amd/csrc, libadypt_hip.so) never includes, links or calls anything in oracle/.
//
// What it restates (reference file:line, all relative to /root/reference):
//   * BVHIntersection closest hit ........ shaders/traversal.glsl:14-255
//   * Render / FetchInfo / sampling ...... shaders/pathtracer.glsl:49-227
//   * primary-ray viewer .................. shaders/primaryray.glsl:39-94
//   * Woop matrix precompute .............. src/Tracer/OglScene.cpp:93-116 (+ glm::inverse, dep/glm/detail/func_matrix.inl:294-351)
//   * camera matrices ..................... src/Tracer/Camera.cpp:13-23, src/Tracer/OglPathTracer.cpp:27-32
//   * Sobol stream / per-pixel shift ...... src/Util/Sobol.cpp:5-21, src/Tracer/OglPathTracer.cpp:48-49,153-162
//   * frame cadence (spp, tmpLife cache) .. src/Tracer/OglPathTracer.cpp:34-61, shaders/pathtracer.glsl:107-128,206-227
//
// Parity pinning (see tests/test_oracle_golden.py):
//   * Woop / camera / Sobol / shift bytes are checked bit-for-bit against the reference's own CPU code
//     compiled from /root/reference (oracle/_ref/adypt_ref) through the committed fixtures in tests/golden/.
//   * The traversal is checked against an fp64 brute-force ray/triangle intersection on the fixtures.
//   * PARITY UNPINNED for the shader half: the reference has no tests, ships no fixtures for this path, and its GLSL
//     cannot be compiled or run here, so no output of the reference's traversal / Render pins this restatement; it is
//     a line-by-line reading of the shaders, checked by fp64 brute force (traversal) and by construction (shading).
//   * The GLSL shaders themselves cannot be compiled or run here (GL 4.5 + bindless textures, no GPU, no GL):
//     GLSL leaves the rounding of normalize/dot/1/x/pow/sin/cos and fma contraction unspecified, so this file
//     *defines* the canonical arithmetic both sides (oracle and HIP kernels) implement:
//        - binary32 IEEE-754 round-to-nearest-even for + - * / sqrt, evaluated in exactly the order written;
//        - fused multiply-add ONLY where fmaf() is written (built with -ffp-contract=off);
//        - dot3(a,b) = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x));  normalize(v) = v * (1/sqrt(dot3(v,v)));
//        - sin/cos/pow: fixed IEEE binary64 series (below), rounded once to binary32;
//        - min/max: GLSL rule min(x,y) = y<x ? y : x, max(x,y) = x<y ? y : x in shading code; the slab test of the
//          traversal uses IEEE maxNum/minNum (same values for non-NaN inputs; NaN inputs are undefined in GLSL).
// =====================================================================================================
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#define ORC_API extern "C" __attribute__((visibility("default")))

namespace {

// ---------------------------------------------------------------------------------------------------
// small vector helpers with the canonical evaluation order
// ---------------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
static inline V3 fma3(V3 a, float s, V3 c) { return v3(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z)); }
static inline V3 fma3(V3 a, V3 b, V3 c) { return v3(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z)); }
static inline float dot3(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline V3 cross3(V3 a, V3 b)
{
	return v3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline V3 normalize3(V3 a) { float inv = 1.0f / sqrtf(dot3(a, a)); return a * inv; }
// GLSL reflect(I, N) = I - 2*dot(N, I)*N
static inline V3 reflect3(V3 i, V3 n) { float k = 2.0f * dot3(n, i); return fma3(n, -k, i); }
static inline float gl_min(float x, float y) { return y < x ? y : x; }
static inline float gl_max(float x, float y) { return x < y ? y : x; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// ---------------------------------------------------------------------------------------------------
// canonical transcendental functions: fixed binary64 series, one final rounding to binary32
// ---------------------------------------------------------------------------------------------------
static void canon_sincos(float xf, float *s_out, float *c_out)
{
	const double TWO_OVER_PI = 0.63661977236758134308;
	const double PIO2_HI = 1.57079632679489655800e+00; // 0x3FF921FB54442D18
	const double PIO2_LO = 6.12323399573676603587e-17; // 0x3C91A62633145C07
	double x = (double)xf;
	double kd = nearbyint(x * TWO_OVER_PI);
	double y = fma(-kd, PIO2_HI, x);
	y = fma(-kd, PIO2_LO, y);
	double y2 = y * y;
	// sin(y) = y * (1 + y2*(-1/3! + y2*(1/5! + ... - 1/19!)))
	double ps = -1.0 / 121645100408832000.0;                // -1/19!
	ps = fma(ps, y2, 1.0 / 355687428096000.0);              //  1/17!
	ps = fma(ps, y2, -1.0 / 1307674368000.0);               // -1/15!
	ps = fma(ps, y2, 1.0 / 6227020800.0);                   //  1/13!
	ps = fma(ps, y2, -1.0 / 39916800.0);                    // -1/11!
	ps = fma(ps, y2, 1.0 / 362880.0);                       //  1/9!
	ps = fma(ps, y2, -1.0 / 5040.0);                        // -1/7!
	ps = fma(ps, y2, 1.0 / 120.0);                          //  1/5!
	ps = fma(ps, y2, -1.0 / 6.0);                           // -1/3!
	double sn = fma(y * y2, ps, y);
	// cos(y) = 1 + y2*(-1/2! + y2*(1/4! - ... + 1/20!))
	double pc = 1.0 / 2432902008176640000.0;                //  1/20!
	pc = fma(pc, y2, -1.0 / 6402373705728000.0);            // -1/18!
	pc = fma(pc, y2, 1.0 / 20922789888000.0);               //  1/16!
	pc = fma(pc, y2, -1.0 / 87178291200.0);                 // -1/14!
	pc = fma(pc, y2, 1.0 / 479001600.0);                    //  1/12!
	pc = fma(pc, y2, -1.0 / 3628800.0);                     // -1/10!
	pc = fma(pc, y2, 1.0 / 40320.0);                        //  1/8!
	pc = fma(pc, y2, -1.0 / 720.0);                         // -1/6!
	pc = fma(pc, y2, 1.0 / 24.0);                           //  1/4!
	pc = fma(pc, y2, -0.5);                                 // -1/2!
	double cs = fma(y2, pc, 1.0);
	long long k = (long long)kd;
	double s, c;
	switch(k & 3)
	{
		case 0: s = sn; c = cs; break;
		case 1: s = cs; c = -sn; break;
		case 2: s = -sn; c = -cs; break;
		default: s = -cs; c = sn; break;
	}
	*s_out = (float)s;
	*c_out = (float)c;
}

static float canon_pow_series(float xf, float yf);
static float canon_pow(float xf, float yf)
{
	// x^1 = x for positive finite x: what the series returns anyway (orc_pow_series lets the tests check that), spelled out
	// so that the product may skip the series for the diffuse lobe
	if(yf == 1.0f && xf > 0.0f && xf < INFINITY) return xf;
	return canon_pow_series(xf, yf);
}
static float canon_pow_series(float xf, float yf)
{
	if(yf == 0.0f) return 1.0f;
	if(xf != xf || yf != yf) return xf + yf;
	if(xf < 0.0f) return u2f(0x7fc00000u);                 // non-integer exponents only on this path
	if(xf == 0.0f) return yf > 0.0f ? 0.0f : u2f(0x7f800000u);
	if(xf == u2f(0x7f800000u)) return yf > 0.0f ? xf : 0.0f;
	double x = (double)xf;                                    // always a normal double
	uint64_t bits; memcpy(&bits, &x, 8);
	int e = (int)((bits >> 52) & 0x7ff) - 1023;
	bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
	double m; memcpy(&m, &bits, 8);                           // m in [1,2)
	if(m > 1.41421356237309514547) { m *= 0.5; e += 1; }     // m in (sqrt(.5), sqrt(2)]
	double s = (m - 1.0) / (m + 1.0);
	double s2 = s * s;
	double p = 1.0 / 23.0;
	p = fma(p, s2, 1.0 / 21.0);
	p = fma(p, s2, 1.0 / 19.0);
	p = fma(p, s2, 1.0 / 17.0);
	p = fma(p, s2, 1.0 / 15.0);
	p = fma(p, s2, 1.0 / 13.0);
	p = fma(p, s2, 1.0 / 11.0);
	p = fma(p, s2, 1.0 / 9.0);
	p = fma(p, s2, 1.0 / 7.0);
	p = fma(p, s2, 1.0 / 5.0);
	p = fma(p, s2, 1.0 / 3.0);
	double ln_m = 2.0 * fma(s * s2, p, s);                    // ln(m) = 2 atanh(s)
	const double LOG2E = 1.44269504088896338700;
	double log2x = fma(ln_m, LOG2E, (double)e);
	double t = (double)yf * log2x;
	if(t >= 129.0) return u2f(0x7f800000u);
	if(t <= -151.0) return 0.0f;
	double n = nearbyint(t);
	double f = t - n;
	const double LN2 = 0.69314718055994528623;
	double z = f * LN2;
	double q = 1.0 / 6227020800.0;                            // 1/13!
	q = fma(q, z, 1.0 / 479001600.0);
	q = fma(q, z, 1.0 / 39916800.0);
	q = fma(q, z, 1.0 / 3628800.0);
	q = fma(q, z, 1.0 / 362880.0);
	q = fma(q, z, 1.0 / 40320.0);
	q = fma(q, z, 1.0 / 5040.0);
	q = fma(q, z, 1.0 / 720.0);
	q = fma(q, z, 1.0 / 120.0);
	q = fma(q, z, 1.0 / 24.0);
	q = fma(q, z, 1.0 / 6.0);
	q = fma(q, z, 0.5);
	q = fma(q, z, 1.0);
	q = fma(q, z, 1.0);                                        // e^z
	uint64_t sb = (uint64_t)((long long)n + 1023) << 52;       // 2^n, n in [-151,129] -> normal double
	double scale; memcpy(&scale, &sb, 8);
	return (float)(q * scale);
}

// ---------------------------------------------------------------------------------------------------
// data layouts (the kernel's input contract; SURVEY.md §2 resource table)
// ---------------------------------------------------------------------------------------------------
struct Node // 80 B  (src/BVH/WideBVH.hpp:13-26  <->  shaders/traversal.glsl:1-5)
{
	float px, py, pz;
	uint8_t ex, ey, ez, imask;
	uint32_t child_base, tri_base;
	uint8_t meta[8], qlox[8], qloy[8], qloz[8], qhix[8], qhiy[8], qhiz[8];
};
static_assert(sizeof(Node) == 80, "node layout");
struct Woop { float m0[4], m1[4], m2[4]; };
struct Tri // 100 B (src/Util/Shape.hpp:70-74 <-> shaders/pathtracer.glsl:2-8)
{
	float p[3][3], n[3][3], tc[3][2];
	int32_t matid;
};
static_assert(sizeof(Tri) == 100, "triangle layout");
struct Mat // 64 B (src/Tracer/OglScene.hpp:19-28 <-> shaders/pathtracer.glsl:9-18)
{
	int32_t dtex; float dr, dg, db;
	int32_t etex; float er, eg, eb;
	int32_t stex; float sr, sg, sb;
	int32_t illum; float shininess, dissolve, ior;
};
static_assert(sizeof(Mat) == 64, "material layout");

} // namespace

extern "C" {
struct OrcTexture { int32_t w, h; const uint8_t *rgb; };
struct OrcScene
{
	const void *nodes; int64_t n_nodes;
	const float *woop; const int32_t *tri_indices; int64_t n_refs;
	const void *triangles; int64_t n_tris;
	const void *materials; int64_t n_mats;
	const OrcTexture *textures; int64_t n_tex;
};
struct OrcParams
{
	int32_t width, height, stack_size, max_bounce, subpixel, tmp_life;
	float tmin, clamp, sun[3], origin[3];
	float inv_proj[16], inv_view[16]; // column-major like glm
	// the occlusion test the reference has commented out (pathtracer.glsl:132): when a path escapes, the sun term is
	// only added if an any-hit ray from the last position towards normalize(sun_dir) finds nothing.  0 = reference behaviour.
	int32_t sun_visibility;
	float sun_dir[3];
};
struct OrcHit { int32_t ref_idx, tri_id; float u, v, t; uint32_t nodes, tris, hash, max_depth; };
struct OrcStats { uint64_t rays, nodes, tris, hits, shaded, texel_fetches, stack_overflows; uint32_t max_depth; uint32_t pad; };
}

namespace {

struct Counters { uint64_t rays = 0, nodes = 0, tris = 0, hits = 0, shaded = 0, texels = 0, overflows = 0; uint32_t max_depth = 0; };

// ---------------------------------------------------------------------------------------------------
// BVHIntersection, closest hit  (shaders/traversal.glsl:14-255)
// ---------------------------------------------------------------------------------------------------
static inline int find_msb(uint32_t x) { return 31 - __builtin_clz(x); }
static inline int find_lsb(uint32_t x) { return __builtin_ctz(x); }

static const int kMaxStack = 64;

// any_hit = the second overload, shaders/traversal.glsl:257-494: identical, returns at the first accepted triangle
static void bvh_intersect(const OrcScene &sc, int stack_size, const float o4[4], V3 dir, OrcHit *out, bool any_hit = false)
{
	const Node *nodes = (const Node *)sc.nodes;
	const Woop *woop = (const Woop *)sc.woop;
	// :16-23 ray setup
	const float ooeps = u2f((127u - 64u) << 23); // exp2(-64)
	dir.x = fabsf(dir.x) > ooeps ? dir.x : (dir.x >= 0 ? ooeps : -ooeps);
	dir.y = fabsf(dir.y) > ooeps ? dir.y : (dir.y >= 0 ? ooeps : -ooeps);
	dir.z = fabsf(dir.z) > ooeps ? dir.z : (dir.z >= 0 ? ooeps : -ooeps);
	dir = normalize3(dir);
	V3 idir = v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
	uint32_t octinv = 7u - ((dir.x < 0 ? 1u : 0u) | (dir.y < 0 ? 2u : 0u) | (dir.z < 0 ? 4u : 0u));
	uint32_t octinv4 = octinv * 0x01010101u;
	V3 origin = v3(o4[0], o4[1], o4[2]);
	float hit_tmin = o4[3];
	float hit_t = 1e9f;
	int32_t hit_idx = -1;
	float hit_u = 0.0f, hit_v = 0.0f; // GLSL: inout, left untouched on a miss; we define 0 for the record

	uint32_t stack[kMaxStack][2];
	int stack_ptr = 0;
	uint32_t tg_x = 0, tg_y = 0, ng_x = 0, ng_y = 0x80000000u; // :35

	uint32_t n_nodes = 0, n_tris = 0, hash = 0x811c9dc5u, max_depth = 0;
	bool overflow = false;

	while(true)
	{
		if(ng_y > 0x00ffffffu) // :47
		{
			uint32_t imask = ng_y;
			uint32_t child_bit_index = (uint32_t)find_msb(ng_y);
			uint32_t child_node_base_index = ng_x;
			ng_y &= ~(1u << child_bit_index);
			if(ng_y > 0x00ffffffu) // :59-60 push
			{
				if(stack_ptr < stack_size && stack_ptr < kMaxStack)
				{
					stack[stack_ptr][0] = ng_x; stack[stack_ptr][1] = ng_y;
					++stack_ptr;
					if((uint32_t)stack_ptr > max_depth) max_depth = (uint32_t)stack_ptr;
				}
				else
					overflow = true; // the reference has undefined behaviour here; the build reports it
			}
			uint32_t slot_index = (child_bit_index - 24u) ^ octinv;
			uint32_t relative_index = (uint32_t)__builtin_popcount(imask & ~(0xffffffffu << slot_index));
			uint32_t child_node_index = child_node_base_index + relative_index;

			const Node &nd = nodes[child_node_index]; // :69-74
			++n_nodes;
			hash = (hash * 0x01000193u) ^ child_node_index;

			float adj_idir_x = u2f((uint32_t)nd.ex << 23) * idir.x; // :76-78
			float adj_idir_y = u2f((uint32_t)nd.ey << 23) * idir.y;
			float adj_idir_z = u2f((uint32_t)nd.ez << 23) * idir.z;
			V3 adj_org = (v3(nd.px, nd.py, nd.pz) - origin) * idir; // :79

			ng_x = nd.child_base; // :81-83
			tg_x = nd.tri_base;
			tg_y = 0;
			uint32_t hitmask = 0;
			const uint8_t *lox = idir.x < 0 ? nd.qhix : nd.qlox, *hix = idir.x < 0 ? nd.qlox : nd.qhix; // :92-99
			const uint8_t *loy = idir.y < 0 ? nd.qhiy : nd.qloy, *hiy = idir.y < 0 ? nd.qloy : nd.qhiy;
			const uint8_t *loz = idir.z < 0 ? nd.qhiz : nd.qloz, *hiz = idir.z < 0 ? nd.qloz : nd.qhiz;
			for(int i = 0; i < 8; ++i)
			{
				uint32_t meta = nd.meta[i];
				uint32_t is_inner = (meta & (meta << 1)) & 0x10u;                       // :88
				uint32_t bit_index = (meta ^ (is_inner ? octinv : 0u)) & 0x1fu;         // :89
				uint32_t child_bits = (meta >> 5) & 0x07u;                              // :90
				float txmin = fmaf((float)lox[i], adj_idir_x, adj_org.x);               // :101-126
				float tymin = fmaf((float)loy[i], adj_idir_y, adj_org.y);
				float tzmin = fmaf((float)loz[i], adj_idir_z, adj_org.z);
				float txmax = fmaf((float)hix[i], adj_idir_x, adj_org.x);
				float tymax = fmaf((float)hiy[i], adj_idir_y, adj_org.y);
				float tzmax = fmaf((float)hiz[i], adj_idir_z, adj_org.z);
				// :128-129.  GLSL leaves min/max of NaN undefined; the canonical choice is IEEE-754 maxNum/minNum
				// (fmaxf/fminf = gfx950 v_max_f32/v_min_f32), identical to the GLSL rule for every non-NaN input.
				float ctmin = fmaxf(fmaxf(txmin, tymin), fmaxf(tzmin, hit_tmin));
				float ctmax = fminf(fminf(txmax, tymax), fminf(tzmax, hit_t));
				if(ctmin <= ctmax) hitmask |= child_bits << bit_index;                  // :130
			}
			(void)octinv4;
			ng_y = (hitmask & 0xff000000u) | (uint32_t)nd.imask; // :204-205
			tg_y = hitmask & 0x00ffffffu;
		}
		else // :207-211 (dead in practice: only node groups are ever pushed)
		{
			tg_x = ng_x; tg_y = ng_y;
			ng_x = 0; ng_y = 0;
		}

		while(tg_y != 0) // :213-243
		{
			uint32_t tridx = (uint32_t)find_lsb(tg_y);
			tg_y &= ~(1u << tridx);
			tridx += tg_x;
			const Woop &w = woop[tridx];
			++n_tris;
			V3 m0 = v3(w.m0[0], w.m0[1], w.m0[2]), m1 = v3(w.m1[0], w.m1[1], w.m1[2]), m2 = v3(w.m2[0], w.m2[1], w.m2[2]);
			float toz = w.m0[3] - dot3(origin, m0);
			float tidz = 1.0f / dot3(dir, m0);
			float tt = toz * tidz;
			float tox = w.m1[3] + dot3(origin, m1);
			float tdx = dot3(dir, m1);
			float tu = fmaf(tt, tdx, tox);
			float toy = w.m2[3] + dot3(origin, m2);
			float tdy = dot3(dir, m2);
			float tv = fmaf(tt, tdy, toy);
			if(tt > hit_tmin && tt < hit_t)
				if(tu >= 0.0f && tu <= 1.0f)
					if(tv >= 0.0f && tu + tv <= 1.0f)
					{
						hit_t = tt; hit_u = tu; hit_v = tv; hit_idx = (int32_t)tridx;
						if(any_hit) goto done; // traversal.glsl:477-483 `return true`
					}
		}

		if(ng_y <= 0x00ffffffu) // :245-250
		{
			if(stack_ptr == 0) break;
			--stack_ptr;
			ng_x = stack[stack_ptr][0]; ng_y = stack[stack_ptr][1];
		}
	}
done:
	out->ref_idx = hit_idx;
	out->tri_id = hit_idx != -1 ? sc.tri_indices[hit_idx] : -1; // :253-254
	out->u = hit_u; out->v = hit_v; out->t = hit_t;
	out->nodes = n_nodes; out->tris = n_tris; out->hash = hash;
	out->max_depth = overflow ? 0xffffffffu : max_depth;
}

// ---------------------------------------------------------------------------------------------------
// shading helpers (shaders/pathtracer.glsl)
// ---------------------------------------------------------------------------------------------------
static inline V3 mat4_mul_xyz(const float m[16], float x, float y, float z, float w)
{
	V3 r;
	r.x = fmaf(m[12], w, fmaf(m[8], z, fmaf(m[4], y, m[0] * x)));
	r.y = fmaf(m[13], w, fmaf(m[9], z, fmaf(m[5], y, m[1] * x)));
	r.z = fmaf(m[14], w, fmaf(m[10], z, fmaf(m[6], y, m[2] * x)));
	return r;
}
static inline V3 mat3_mul(const float m[16], V3 v)
{
	V3 r;
	r.x = fmaf(m[8], v.z, fmaf(m[4], v.y, m[0] * v.x));
	r.y = fmaf(m[9], v.z, fmaf(m[5], v.y, m[1] * v.x));
	r.z = fmaf(m[10], v.z, fmaf(m[6], v.y, m[2] * v.x));
	return r;
}
// Camera(bias): pathtracer.glsl:213-218 / primaryray.glsl:39-44 (bias = 0)
static inline V3 camera_dir(const OrcParams &p, int px, int py, float bx, float by)
{
	float sx = (2.0f * ((float)px + bx)) / (float)p.width - 1.0f;
	float sy = (2.0f * ((float)py + by)) / (float)p.height - 1.0f;
	sy = -sy;
	V3 t = mat4_mul_xyz(p.inv_proj, sx, sy, 1.0f, 1.0f);
	return normalize3(mat3_mul(p.inv_view, t));
}

static inline int pos_mod(int a, int n) { int r = a % n; return r < 0 ? r + n : r; }

// GL_LINEAR / GL_REPEAT / GL_RGB8 single level (src/Tracer/OglScene.cpp:33-38); canonical fp32 weights
static V3 sample_texture(const OrcTexture &t, float s, float tt, Counters &cn)
{
	float uu = fmaf(s, (float)t.w, -0.5f), vv = fmaf(tt, (float)t.h, -0.5f);
	float fu = floorf(uu), fv = floorf(vv);
	float a = uu - fu, b = vv - fv;
	// clamp the float before the int conversion so absurd coordinates stay defined
	fu = gl_min(gl_max(fu, -1e9f), 1e9f); fv = gl_min(gl_max(fv, -1e9f), 1e9f);
	int i0 = pos_mod((int)fu, t.w), j0 = pos_mod((int)fv, t.h);
	int i1 = i0 + 1 == t.w ? 0 : i0 + 1, j1 = j0 + 1 == t.h ? 0 : j0 + 1;
	auto tex = [&](int i, int j) {
		const uint8_t *p = t.rgb + ((size_t)j * t.w + i) * 3;
		return v3((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f);
	};
	cn.texels += 4;
	float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
	V3 r = tex(i0, j0) * w00;
	r = fma3(tex(i1, j0), w10, r);
	r = fma3(tex(i0, j1), w01, r);
	r = fma3(tex(i1, j1), w11, r);
	return r;
}

static inline V3 bary3(const float a[3], const float b[3], const float c[3], float u, float v, float w)
{
	V3 r = v3(a[0], a[1], a[2]) * u;
	r = fma3(v3(b[0], b[1], b[2]), v, r);
	r = fma3(v3(c[0], c[1], c[2]), w, r);
	return r;
}

struct PixelRng { float sx, sy; const float *sobol; }; // sobol: 2*maxBounce floats of this frame
// Sobol(i): pathtracer.glsl:49
static inline void sobol2(const PixelRng &r, int i, float *x, float *y)
{
	float a = r.sobol[2 * i] + r.sx, b = r.sobol[2 * i + 1] + r.sy;
	*x = a - floorf(a); *y = b - floorf(b);
}
// SampleHemisphere: pathtracer.glsl:52-64
static V3 sample_hemisphere(const PixelRng &rng, int b, float e)
{
	float rx, ry; sobol2(rng, b, &rx, &ry);
	rx *= 6.28318530718f;
	float sin_phi, cos_phi; canon_sincos(rx, &sin_phi, &cos_phi);
	float cos_theta = canon_pow(1.0f - ry, 1.0f / (e + 1.0f));
	float sin_theta = sqrtf(fmaf(-cos_theta, cos_theta, 1.0f));
	return normalize3(v3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta));
}
// AlignDirection: pathtracer.glsl:66-71
static V3 align_direction(V3 dir, V3 target)
{
	V3 a = fabsf(target.x) > 0.01f ? v3(0, 1, 0) : v3(1, 0, 0);
	V3 u = normalize3(cross3(a, target));
	V3 v = cross3(target, u);
	V3 r = u * dir.x;
	r = fma3(v, dir.y, r);
	r = fma3(target, dir.z, r);
	return r;
}

struct Surface { V3 position, normal, emissive, diffuse, specular; const Mat *mtl; };

// FetchInfo: pathtracer.glsl:73-100.  Returns false when the material id is out of range (the reference reads
// uMaterials[-1]; the build defines this as "terminate the path, report").
static bool fetch_info(const OrcScene &sc, int tri_idx, float u, float v, Surface *s, Counters &cn)
{
	const Tri &tri = ((const Tri *)sc.triangles)[tri_idx];
	if(tri.matid < 0 || tri.matid >= sc.n_mats) return false;
	const Mat &m = ((const Mat *)sc.materials)[tri.matid];
	float w = 1.0f - u - v;
	s->mtl = &m;
	s->normal = normalize3(bary3(tri.n[0], tri.n[1], tri.n[2], u, v, w));
	s->position = bary3(tri.p[0], tri.p[1], tri.p[2], u, v, w);
	s->emissive = v3(m.er, m.eg, m.eb);
	if(sc.n_tex != 0 && m.dtex != -1 && m.dtex >= 0 && m.dtex < sc.n_tex)
	{
		float ts = fmaf(tri.tc[2][0], w, fmaf(tri.tc[1][0], v, tri.tc[0][0] * u));
		float tt = fmaf(tri.tc[2][1], w, fmaf(tri.tc[1][1], v, tri.tc[0][1] * u));
		s->diffuse = sample_texture(sc.textures[m.dtex], ts, tt, cn);
	}
	else
		s->diffuse = v3(m.dr, m.dg, m.db);
	s->specular = v3(m.sr, m.sg, m.sb);
	return true;
}

// The material response of one loop iteration of Render (pathtracer.glsl:141-201): normal flip, `illum` switch, new direction
// and throughput.  Returns false where the GLSL has `return ret` (glossy sample below the surface, :151).  A separate function
// only so that tests can drive it with hand-made surfaces (orc_scatter); render() below is its one caller.
static bool scatter_step(const Surface &s, const PixelRng &rng, int b, V3 &dir, V3 &color, float *fresnel_out)
{
	const Mat &m = *s.mtl;
	V3 normal = s.normal;
	if(m.illum < 6 && dot3(dir, normal) > 0) normal = -normal;

	int illum = m.illum;
	if(illum == 2)
	{
		float e = m.shininess * 0.01f;
		if(e > 0.3f)
		{
			V3 r = reflect3(dir, normal), sh = sample_hemisphere(rng, b, e);
			dir = align_direction(sh, r);
			if(dot3(dir, normal) < 0.0f) return false;
			float pw = canon_pow(dot3(dir, r), e);
			color = color * fma3(s.specular, pw, s.diffuse);
			return true;
		}
		illum = 1; // falls through to diffuse
	}
	if(illum == 1)
	{
		dir = align_direction(sample_hemisphere(rng, b, 0.0f), normal);
		color = color * s.diffuse;
	}
	else if(illum >= 3 && illum <= 5)
	{
		color = color * s.specular;
		dir = reflect3(dir, normal);
	}
	else if(illum == 6 || illum == 7)
	{
		float eta = m.ior;
		float cosi = dot3(dir, normal);
		float fresnel, etai, etat;
		if(cosi > 0) { etai = eta; etat = 1.0f; }
		else { etai = 1.0f; etat = eta; normal = -normal; cosi = -cosi; }
		eta = etai / etat;
		float sint = (etai / etat) * sqrtf(gl_max(0.0f, fmaf(-cosi, cosi, 1.0f)));
		if(sint >= 1.0f) fresnel = 1.0f;
		else
		{
			float cost = sqrtf(gl_max(0.0f, fmaf(-sint, sint, 1.0f)));
			float A = etat * cosi, B = etai * cost, C = etai * cosi, D = etat * cost;
			float Rs = (A - B) / (A + B);
			float Rp = (C - D) / (C + D);
			fresnel = fmaf(Rs, Rs, Rp * Rp) * 0.5f;
		}
		if(fresnel_out) *fresnel_out = fresnel;
		float cos2 = fmaf(-(eta * eta), fmaf(-cosi, cosi, 1.0f), 1.0f);
		float sx, sy; sobol2(rng, b, &sx, &sy);
		if(cos2 > 0 && sx >= fresnel)
		{
			float k = fmaf(eta, cosi, sqrtf(cos2));
			dir = normalize3(fma3(normal, k, dir * eta));
		}
		else
			dir = reflect3(dir, normal);
	}
	// any other illum: the ray continues straight through (no case in the reference's switch)
	return true;
}

// Render: pathtracer.glsl:101-204.  cache = this pixel's {tri_idx, u, v} slot of uPrimaryTmpImg.
static V3 render(const OrcScene &sc, const OrcParams &p, int spp, V3 dir, const PixelRng &rng, int32_t *cache_tri,
				 float *cache_uv, Counters &cn)
{
	float o4[4] = {p.origin[0], p.origin[1], p.origin[2], p.tmin};
	V3 ret = v3(0, 0, 0), color = v3(1, 1, 1);
	V3 sun = v3(p.sun[0], p.sun[1], p.sun[2]);
	int tri_idx = -1; float tu = 0, tv = 0;
	for(int b = 0; b < p.max_bounce; ++b)
	{
		if(b > 0 || spp % p.tmp_life == 0)
		{
			OrcHit h;
			bvh_intersect(sc, p.stack_size, o4, dir, &h);
			++cn.rays; cn.nodes += h.nodes; cn.tris += h.tris;
			if(h.max_depth == 0xffffffffu) ++cn.overflows; else if(h.max_depth > cn.max_depth) cn.max_depth = h.max_depth;
			tri_idx = h.tri_id; tu = h.u; tv = h.v;
			if(tri_idx != -1) ++cn.hits;
			if(b == 0) { *cache_tri = tri_idx; cache_uv[0] = tu; cache_uv[1] = tv; }
		}
		else { tri_idx = *cache_tri; tu = cache_uv[0]; tv = cache_uv[1]; }

		if(tri_idx == -1)
		{
			bool lit = true;
			if(p.sun_visibility) // if(!BVHIntersection(origin, normalize(vec3(0.6, 1, 0.2))))   (pathtracer.glsl:132)
			{
				OrcHit sh;
				bvh_intersect(sc, p.stack_size, o4, normalize3(v3(p.sun_dir[0], p.sun_dir[1], p.sun_dir[2])), &sh, true);
				++cn.rays; cn.nodes += sh.nodes; cn.tris += sh.tris;
				if(sh.max_depth == 0xffffffffu) ++cn.overflows; else if(sh.max_depth > cn.max_depth) cn.max_depth = sh.max_depth;
				if(sh.tri_id != -1) ++cn.hits;
				lit = sh.tri_id == -1;
			}
			if(lit) ret = fma3(color, sun, ret);
			break;
		}

		Surface s;
		if(!fetch_info(sc, tri_idx, tu, tv, &s, cn)) break;
		++cn.shaded;
		o4[0] = s.position.x; o4[1] = s.position.y; o4[2] = s.position.z;
		ret = fma3(color, s.emissive, ret);
		if(!scatter_step(s, rng, b, dir, color, nullptr)) return ret;
	}
	return ret;
}

template <class F> static void parallel_rows(int height, int n_threads, F &&fn)
{
	if(n_threads <= 1) { for(int y = 0; y < height; ++y) fn(y, 0); return; }
	std::atomic<int> next{0};
	std::vector<std::thread> th;
	for(int t = 0; t < n_threads; ++t)
		th.emplace_back([&, t]() { for(int y; (y = next.fetch_add(1)) < height;) fn(y, t); });
	for(auto &x : th) x.join();
}

static void add_stats(OrcStats *st, const std::vector<Counters> &cs)
{
	if(!st) return;
	for(const Counters &c : cs)
	{
		st->rays += c.rays; st->nodes += c.nodes; st->tris += c.tris; st->hits += c.hits; st->shaded += c.shaded;
		st->texel_fetches += c.texels; st->stack_overflows += c.overflows;
		if(c.max_depth > st->max_depth) st->max_depth = c.max_depth;
	}
}

} // namespace

// =====================================================================================================
// C entry points (ctypes)
// =====================================================================================================

// --- glm::inverse(mat4) restated: dep/glm/detail/func_matrix.inl:294-351 (column-major m[col][row]) ---
static void mat4_inverse(const float m_[16], float out[16])
{
	auto m = [&](int c, int r) { return m_[c * 4 + r]; };
	float Coef00 = m(2, 2) * m(3, 3) - m(3, 2) * m(2, 3);
	float Coef02 = m(1, 2) * m(3, 3) - m(3, 2) * m(1, 3);
	float Coef03 = m(1, 2) * m(2, 3) - m(2, 2) * m(1, 3);
	float Coef04 = m(2, 1) * m(3, 3) - m(3, 1) * m(2, 3);
	float Coef06 = m(1, 1) * m(3, 3) - m(3, 1) * m(1, 3);
	float Coef07 = m(1, 1) * m(2, 3) - m(2, 1) * m(1, 3);
	float Coef08 = m(2, 1) * m(3, 2) - m(3, 1) * m(2, 2);
	float Coef10 = m(1, 1) * m(3, 2) - m(3, 1) * m(1, 2);
	float Coef11 = m(1, 1) * m(2, 2) - m(2, 1) * m(1, 2);
	float Coef12 = m(2, 0) * m(3, 3) - m(3, 0) * m(2, 3);
	float Coef14 = m(1, 0) * m(3, 3) - m(3, 0) * m(1, 3);
	float Coef15 = m(1, 0) * m(2, 3) - m(2, 0) * m(1, 3);
	float Coef16 = m(2, 0) * m(3, 2) - m(3, 0) * m(2, 2);
	float Coef18 = m(1, 0) * m(3, 2) - m(3, 0) * m(1, 2);
	float Coef19 = m(1, 0) * m(2, 2) - m(2, 0) * m(1, 2);
	float Coef20 = m(2, 0) * m(3, 1) - m(3, 0) * m(2, 1);
	float Coef22 = m(1, 0) * m(3, 1) - m(3, 0) * m(1, 1);
	float Coef23 = m(1, 0) * m(2, 1) - m(2, 0) * m(1, 1);
	float Fac0[4] = {Coef00, Coef00, Coef02, Coef03}, Fac1[4] = {Coef04, Coef04, Coef06, Coef07};
	float Fac2[4] = {Coef08, Coef08, Coef10, Coef11}, Fac3[4] = {Coef12, Coef12, Coef14, Coef15};
	float Fac4[4] = {Coef16, Coef16, Coef18, Coef19}, Fac5[4] = {Coef20, Coef20, Coef22, Coef23};
	float Vec0[4] = {m(1, 0), m(0, 0), m(0, 0), m(0, 0)}, Vec1[4] = {m(1, 1), m(0, 1), m(0, 1), m(0, 1)};
	float Vec2[4] = {m(1, 2), m(0, 2), m(0, 2), m(0, 2)}, Vec3[4] = {m(1, 3), m(0, 3), m(0, 3), m(0, 3)};
	const float SignA[4] = {+1, -1, +1, -1}, SignB[4] = {-1, +1, -1, +1};
	float inv[4][4];
	for(int i = 0; i < 4; ++i)
	{
		inv[0][i] = (Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i] + Vec3[i] * Fac2[i]) * SignA[i];
		inv[1][i] = (Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i] + Vec3[i] * Fac4[i]) * SignB[i];
		inv[2][i] = (Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i] + Vec3[i] * Fac5[i]) * SignA[i];
		inv[3][i] = (Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i] + Vec2[i] * Fac5[i]) * SignB[i];
	}
	float d0 = m(0, 0) * inv[0][0], d1 = m(0, 1) * inv[1][0], d2 = m(0, 2) * inv[2][0], d3 = m(0, 3) * inv[3][0];
	float det = (d0 + d1) + (d2 + d3);
	float ood = 1.0f / det;
	for(int c = 0; c < 4; ++c)
		for(int r = 0; r < 4; ++r) out[c * 4 + r] = inv[c][r] * ood;
}

ORC_API void orc_mat4_inverse(const float *m, float *out) { mat4_inverse(m, out); }

// init_triangles: src/Tracer/OglScene.cpp:93-116
ORC_API void orc_woop(const void *triangles, const int32_t *tri_indices, int64_t n_refs, float *out)
{
	const Tri *tris = (const Tri *)triangles;
	for(int64_t i = 0; i < n_refs; ++i)
	{
		const Tri &t = tris[tri_indices[i]];
		V3 v0 = v3(t.p[0][0], t.p[0][1], t.p[0][2]), v1 = v3(t.p[1][0], t.p[1][1], t.p[1][2]), v2 = v3(t.p[2][0], t.p[2][1], t.p[2][2]);
		V3 c0 = v0 - v2, c1 = v1 - v2;
		// glm::cross: (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y), no contraction
		V3 c2 = v3(c0.y * c1.z - c1.y * c0.z, c0.z * c1.x - c1.z * c0.x, c0.x * c1.y - c1.x * c0.y);
		// glm::mat4 ctor takes column-major scalars: mtx[0] = (c0.x,c1.x,c2.x,c3.x) ...
		float mtx[16] = {c0.x, c1.x, c2.x, v2.x, c0.y, c1.y, c2.y, v2.y, c0.z, c1.z, c2.z, v2.z, 0.0f, 0.0f, 0.0f, 1.0f};
		float inv[16];
		mat4_inverse(mtx, inv);
		float *o = out + i * 12;
		o[0] = inv[8]; o[1] = inv[9]; o[2] = inv[10]; o[3] = -inv[11];
		o[4] = inv[0]; o[5] = inv[1]; o[6] = inv[2]; o[7] = inv[3];
		o[8] = inv[4]; o[9] = inv[5]; o[10] = inv[6]; o[11] = inv[7];
	}
}

// glm::rotate(m, angle, axis) restated: dep/glm/ext/matrix_transform.inl:18-46
static void mat4_rotate(const float m[16], float angle, V3 axis_in, float out[16])
{
	float c = cosf(angle), s = sinf(angle);
	// glm::normalize = v * inversesqrt(dot(v,v)) with inversesqrt(x) = 1/sqrt(x); dot = x*x + y*y + z*z (plain)
	float d = axis_in.x * axis_in.x + axis_in.y * axis_in.y + axis_in.z * axis_in.z;
	float inv = 1.0f / sqrtf(d);
	float ax[3] = {axis_in.x * inv, axis_in.y * inv, axis_in.z * inv};
	float t[3] = {(1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2]};
	float R[3][3];
	R[0][0] = c + t[0] * ax[0]; R[0][1] = t[0] * ax[1] + s * ax[2]; R[0][2] = t[0] * ax[2] - s * ax[1];
	R[1][0] = t[1] * ax[0] - s * ax[2]; R[1][1] = c + t[1] * ax[1]; R[1][2] = t[1] * ax[2] + s * ax[0];
	R[2][0] = t[2] * ax[0] + s * ax[1]; R[2][1] = t[2] * ax[1] - s * ax[0]; R[2][2] = c + t[2] * ax[2];
	for(int col = 0; col < 3; ++col)
		for(int r = 0; r < 4; ++r)
			out[col * 4 + r] = m[0 * 4 + r] * R[col][0] + m[1 * 4 + r] * R[col][1] + m[2 * 4 + r] * R[col][2];
	for(int r = 0; r < 4; ++r) out[12 + r] = m[12 + r];
}

// Camera::GetView/GetProjection + SetCamera: src/Tracer/Camera.cpp:13-23, OglPathTracer.cpp:27-32.
// NOTE: tan/cos/sin here are host libm calls in the reference as well (glm forwards to std::), so the result is
// pinned by the golden fixture generated from the reference's glm on this toolchain.
ORC_API void orc_camera(float fov_deg, float yaw_deg, float pitch_deg, int width, int height, float *inv_proj, float *inv_view)
{
	const float deg = 0.01745329251994329576923690768489f; // glm::radians: degrees * 0.0174532925...
	float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
	float v1[16], view[16];
	mat4_rotate(ident, (-pitch_deg) * deg, v3(1, 0, 0), v1);
	mat4_rotate(v1, (-yaw_deg) * deg, v3(0, 1, 0), view);
	// tweakedInfinitePerspective(fovy, aspect, zNear, ep = epsilon<float>()): dep/glm/ext/matrix_clip_space.inl:512-533
	float fovy = fov_deg * deg, aspect = width / (float)height, zn = 0.01f, ep = 1.1920928955078125e-7f;
	float range = tanf(fovy / 2.0f) * zn;
	float left = -range * aspect, right = range * aspect, bottom = -range, top = range;
	float proj[16] = {0};
	proj[0] = (2.0f * zn) / (right - left);
	proj[5] = (2.0f * zn) / (top - bottom);
	proj[10] = ep - 1.0f;
	proj[11] = -1.0f;
	proj[14] = (ep - 2.0f) * zn;
	mat4_inverse(proj, inv_proj);
	mat4_inverse(view, inv_view);
}

// Sobol::Next for `n` consecutive frames starting at frame index `first` (src/Util/Sobol.cpp:5-21).
// matrices: rows [0,dim) of kMatrices (32 u32 each).  out: n*dim floats.
ORC_API void orc_sobol(const uint32_t *matrices, int dim, int first, int n, float *out)
{
	std::vector<uint32_t> x((size_t)dim, 0u);
	for(int idx = 0; idx < first + n; ++idx)
	{
		unsigned c = 0; // index of the lowest zero bit of idx
		while(c < 32 && ((unsigned)idx >> c & 1u)) ++c;
		for(int j = 0; j < dim; ++j)
		{
			x[j] ^= matrices[(size_t)j * 32 + c];
			if(idx >= first) out[(size_t)(idx - first) * dim + j] = (float)(x[j] / 4294967296.0);
		}
	}
}

// per-pixel Cranley-Patterson shift bytes (src/Tracer/OglPathTracer.cpp:153-162) with an explicit seed
// instead of std::random_device.  out: width*height*2 bytes, pixel-major (R then G), row 0 = top row.
ORC_API void orc_shift_bytes(uint32_t seed, int width, int height, uint8_t *out)
{
	std::mt19937 gen{seed};
	size_t n = (size_t)width * height * 2;
	for(size_t i = 0; i < n; ++i) out[i] = (uint8_t)(signed char)gen();
}

ORC_API void orc_sincos(const float *x, int n, float *s, float *c) { for(int i = 0; i < n; ++i) canon_sincos(x[i], s + i, c + i); }
ORC_API void orc_pow(const float *x, const float *y, int n, float *out) { for(int i = 0; i < n; ++i) out[i] = canon_pow(x[i], y[i]); }
ORC_API void orc_pow_series(const float *x, const float *y, int n, float *out) { for(int i = 0; i < n; ++i) out[i] = canon_pow_series(x[i], y[i]); }

// rays: n * 8 floats (ox, oy, oz, tmin, dx, dy, dz, unused)
static void trace_batch(const OrcScene *sc, int stack_size, const float *rays, int64_t n, OrcHit *hits, int n_threads, bool any_hit)
{
	const int64_t chunk = 4096;
	int n_chunks = (int)((n + chunk - 1) / chunk);
	parallel_rows(n_chunks, n_threads, [&](int ci, int) {
		int64_t b = (int64_t)ci * chunk, e = b + chunk < n ? b + chunk : n;
		for(int64_t i = b; i < e; ++i)
			bvh_intersect(*sc, stack_size, rays + i * 8, v3(rays[i * 8 + 4], rays[i * 8 + 5], rays[i * 8 + 6]), hits + i, any_hit);
	});
}
ORC_API void orc_trace(const OrcScene *sc, int stack_size, const float *rays, int64_t n, OrcHit *hits, int n_threads)
{
	trace_batch(sc, stack_size, rays, n, hits, n_threads, false);
}
// any-hit overload (traversal.glsl:257-494): hits[i].tri_id != -1  <=>  the GLSL function returns true
ORC_API void orc_trace_any(const OrcScene *sc, int stack_size, const float *rays, int64_t n, OrcHit *hits, int n_threads)
{
	trace_batch(sc, stack_size, rays, n, hits, n_threads, true);
}

// screen.glsl main (:15-21) + the RGBA8 UNORM colour buffer it is drawn into: rgba W*H*4 floats -> out W*H*4 bytes.
// uType <= 3: pow(v.xyz, 1/2.2) (exponent = the binary32 quotient 1.0f / 2.2f); else normalize(v.xyz) * 0.5 + 0.5.
static uint8_t unorm8(float c)
{
	if(!(c > 0.0f)) return 0; // also NaN
	if(c >= 1.0f) return 255;
	return (uint8_t)floorf(fmaf(c, 255.0f, 0.5f));
}
ORC_API void orc_display(const float *rgba, int64_t n_px, int viewer_type, uint8_t *out)
{
	for(int64_t i = 0; i < n_px; ++i)
	{
		const float *v = rgba + i * 4;
		float c[3];
		if(viewer_type <= 3)
		{
			const float g = 1.0f / 2.2f;
			for(int k = 0; k < 3; ++k) c[k] = canon_pow(v[k], g);
		}
		else
		{
			V3 nn = normalize3(v3(v[0], v[1], v[2]));
			c[0] = nn.x * 0.5f + 0.5f; c[1] = nn.y * 0.5f + 0.5f; c[2] = nn.z * 0.5f + 0.5f;
		}
		out[i * 4 + 0] = unorm8(c[0]); out[i * 4 + 1] = unorm8(c[1]); out[i * 4 + 2] = unorm8(c[2]); out[i * 4 + 3] = 255;
	}
}

// primaryray.glsl main (:46-94).  rgba: W*H*4.  hits (optional): W*H OrcHit records.
ORC_API void orc_primary_frame(const OrcScene *sc, const OrcParams *p, int viewer_type, float *rgba, OrcHit *hits,
							   OrcStats *stats, int n_threads)
{
	std::vector<Counters> cs((size_t)(n_threads > 1 ? n_threads : 1));
	parallel_rows(p->height, n_threads, [&](int y, int t) {
		Counters &cn = cs[(size_t)t];
		for(int x = 0; x < p->width; ++x)
		{
			float o4[4] = {p->origin[0], p->origin[1], p->origin[2], p->tmin};
			OrcHit h;
			bvh_intersect(*sc, p->stack_size, o4, camera_dir(*p, x, y, 0.0f, 0.0f), &h);
			++cn.rays; cn.nodes += h.nodes; cn.tris += h.tris;
			if(h.max_depth == 0xffffffffu) ++cn.overflows; else if(h.max_depth > cn.max_depth) cn.max_depth = h.max_depth;
			if(hits) hits[(size_t)y * p->width + x] = h;
			float *o = rgba + ((size_t)y * p->width + x) * 4;
			o[0] = o[1] = o[2] = 0.0f; o[3] = 1.0f;
			if(h.tri_id == -1) continue;
			++cn.hits;
			const Tri &tri = ((const Tri *)sc->triangles)[h.tri_id];
			if(tri.matid < 0 || tri.matid >= sc->n_mats) continue;
			const Mat &m = ((const Mat *)sc->materials)[tri.matid];
			float u = h.u, v = h.v, w = 1.0f - u - v;
			V3 color = v3(0, 0, 0); // viewer type 3 (PT radiance) writes nothing defined in the reference; we define 0
			if(viewer_type == 0)
			{
				if(sc->n_tex != 0 && m.dtex != -1 && m.dtex >= 0 && m.dtex < sc->n_tex)
				{
					float ts = fmaf(tri.tc[2][0], w, fmaf(tri.tc[1][0], v, tri.tc[0][0] * u));
					float tt = fmaf(tri.tc[2][1], w, fmaf(tri.tc[1][1], v, tri.tc[0][1] * u));
					color = sample_texture(sc->textures[m.dtex], ts, tt, cn);
				}
				else color = v3(m.dr, m.dg, m.db);
			}
			else if(viewer_type == 1) color = v3(m.sr, m.sg, m.sb);
			else if(viewer_type == 2) color = v3(m.er, m.eg, m.eb);
			else if(viewer_type == 4) color = normalize3(bary3(tri.n[0], tri.n[1], tri.n[2], u, v, w));
			else if(viewer_type == 5) color = bary3(tri.p[0], tri.p[1], tri.p[2], u, v, w);
			o[0] = color.x; o[1] = color.y; o[2] = color.z;
		}
	});
	add_stats(stats, cs);
}

// OglPathTracer::Trace(true) x n_spp frames, frames [spp_begin, spp_begin+n_spp) (OglPathTracer.cpp:34-61 +
// pathtracer.glsl main :220-227).  accum: W*H*4 running mean (RGBA, A = 1), cache_tri / cache_uv: image 1.
// sobol: (spp_begin+n_spp) x (2*max_bounce) floats is NOT required: pass the points of frames
// [spp_begin, spp_begin+n_spp) only, row-major n_spp x (2*max_bounce).  mask (optional): W*H bytes, 0 = skip pixel.
ORC_API void orc_pt_frames(const OrcScene *sc, const OrcParams *p, const uint8_t *shift, const float *sobol, int spp_begin,
						   int n_spp, float *accum, int32_t *cache_tri, float *cache_uv, const uint8_t *mask,
						   OrcStats *stats, int n_threads)
{
	std::vector<Counters> cs((size_t)(n_threads > 1 ? n_threads : 1));
	const int dim = 2 * p->max_bounce;
	for(int f = 0; f < n_spp; ++f)
	{
		const int spp = spp_begin + f;
		const float *pts = sobol + (size_t)f * dim;
		// SubPixel(): pathtracer.glsl:206-211
		int sub_idx = (spp / p->tmp_life) % (p->subpixel * p->subpixel);
		const float unit = 1.0f / (float)p->subpixel;
		float bx = (float)(sub_idx / p->subpixel) * unit, by = (float)(sub_idx % p->subpixel) * unit;
		parallel_rows(p->height, n_threads, [&](int y, int t) {
			Counters &cn = cs[(size_t)t];
			for(int x = 0; x < p->width; ++x)
			{
				size_t pix = (size_t)y * p->width + x;
				if(mask && !mask[pix]) continue;
				PixelRng rng{(float)shift[pix * 2] / 255.0f, (float)shift[pix * 2 + 1] / 255.0f, pts};
				V3 r = render(*sc, *p, spp, camera_dir(*p, x, y, bx, by), rng, cache_tri + pix, cache_uv + pix * 2, cn);
				r = v3(gl_min(r.x, p->clamp), gl_min(r.y, p->clamp), gl_min(r.z, p->clamp));
				float *o = accum + pix * 4;
				float fs = (float)spp, fs1 = (float)(spp + 1);
				o[0] = fmaf(o[0], fs, r.x) / fs1;
				o[1] = fmaf(o[1], fs, r.y) / fs1;
				o[2] = fmaf(o[2], fs, r.z) / fs1;
				o[3] = 1.0f;
			}
		});
	}
	add_stats(stats, cs);
}

// fp64 brute-force closest hit over *scene* triangles (independent check of the traversal; Moller-Trumbore).
// Returns scene triangle id and t for each ray; used only by tests.
ORC_API void orc_brute_force(const void *triangles, int64_t n_tris, const float *rays, int64_t n, int32_t *tri_out, double *t_out,
							 int n_threads)
{
	const Tri *tris = (const Tri *)triangles;
	parallel_rows((int)n, n_threads, [&](int i, int) {
		const float *r = rays + (size_t)i * 8;
		double o[3] = {r[0], r[1], r[2]}, d[3] = {r[4], r[5], r[6]};
		double len = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
		for(double &c : d) c /= len;
		double best = 1e9; int32_t bi = -1;
		for(int64_t k = 0; k < n_tris; ++k)
		{
			const Tri &t = tris[k];
			double e1[3], e2[3], pv[3], tv[3], qv[3];
			for(int c = 0; c < 3; ++c) { e1[c] = (double)t.p[1][c] - t.p[0][c]; e2[c] = (double)t.p[2][c] - t.p[0][c]; }
			pv[0] = d[1] * e2[2] - d[2] * e2[1]; pv[1] = d[2] * e2[0] - d[0] * e2[2]; pv[2] = d[0] * e2[1] - d[1] * e2[0];
			double det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
			if(det == 0.0) continue;
			double id = 1.0 / det;
			for(int c = 0; c < 3; ++c) tv[c] = o[c] - t.p[0][c];
			double u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * id;
			if(u < 0.0 || u > 1.0) continue;
			qv[0] = tv[1] * e1[2] - tv[2] * e1[1]; qv[1] = tv[2] * e1[0] - tv[0] * e1[2]; qv[2] = tv[0] * e1[1] - tv[1] * e1[0];
			double v = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) * id;
			if(v < 0.0 || u + v > 1.0) continue;
			double tt = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * id;
			if(tt > (double)r[3] && tt < best) { best = tt; bi = (int32_t)k; }
		}
		tri_out[i] = bi; t_out[i] = best;
	});
}


// ---- test hooks: the shading pieces on their own (tests/test_oracle_shading.py checks them against fp64 closed forms) --------
// SampleHemisphere (pathtracer.glsl:52-64) for n (r.x, r.y) pairs in [0,1): the pairs are used as Sobol(b) directly (shift 0)
ORC_API void orc_sample_hemisphere(const float *r, int n, float e, float *out)
{
	for(int i = 0; i < n; ++i)
	{
		PixelRng rng{0.0f, 0.0f, r + (size_t)i * 2};
		V3 d = sample_hemisphere(rng, 0, e);
		out[i * 3] = d.x; out[i * 3 + 1] = d.y; out[i * 3 + 2] = d.z;
	}
}
// AlignDirection (pathtracer.glsl:66-71)
ORC_API void orc_align_direction(const float *dir, const float *target, int n, float *out)
{
	for(int i = 0; i < n; ++i)
	{
		V3 r = align_direction(v3(dir[i * 3], dir[i * 3 + 1], dir[i * 3 + 2]), v3(target[i * 3], target[i * 3 + 1], target[i * 3 + 2]));
		out[i * 3] = r.x; out[i * 3 + 1] = r.y; out[i * 3 + 2] = r.z;
	}
}
// one material response (pathtracer.glsl:141-201) per record: material (64 B), unit normal, incoming direction, Sobol point of
// the bounce (2 floats, shift 0); diffuse / specular are the material's Kd / Ks.  out per record: new direction (3), throughput
// factor (3), Fresnel term (dielectrics, else -1), alive (1 / 0).
ORC_API void orc_scatter(const void *materials, const float *normal, const float *dir_in, const float *r, int n, float *out)
{
	const Mat *mats = (const Mat *)materials;
	for(int i = 0; i < n; ++i)
	{
		Surface s;
		s.mtl = &mats[i];
		s.normal = v3(normal[i * 3], normal[i * 3 + 1], normal[i * 3 + 2]);
		s.position = v3(0, 0, 0);
		s.emissive = v3(mats[i].er, mats[i].eg, mats[i].eb);
		s.diffuse = v3(mats[i].dr, mats[i].dg, mats[i].db);
		s.specular = v3(mats[i].sr, mats[i].sg, mats[i].sb);
		PixelRng rng{0.0f, 0.0f, r + (size_t)i * 2};
		V3 dir = v3(dir_in[i * 3], dir_in[i * 3 + 1], dir_in[i * 3 + 2]), color = v3(1, 1, 1);
		float fresnel = -1.0f;
		const bool alive = scatter_step(s, rng, 0, dir, color, &fresnel);
		float *o = out + (size_t)i * 8;
		o[0] = dir.x; o[1] = dir.y; o[2] = dir.z; o[3] = color.x; o[4] = color.y; o[5] = color.z; o[6] = fresnel; o[7] = alive ? 1.0f : 0.0f;
	}
}

ORC_API int orc_abi_version(void) { return 2; }
