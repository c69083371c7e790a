// MEASUREMENT ONLY — never part of the shipped library: the product sources do not mention this file or its hooks.
// Bodies of the hook points that csrc/measure/k_trace_hooks.py puts into a scratch copy of the device sources: redundant work that prices one
// pipeline of the CU at a time.  Results stay correct with every one of them except SHADE_TRI_L2 and FP16_NODES' node layout (bit-identical
// images, different upload); the numbers they produced are in profiles/r2_ablations_k_trace.txt and profiles/r3_ablations_k_trace.txt.
//   tools/build_variant.sh <name> --transform adypt_amd/csrc/measure/k_trace_hooks.py -DADYPT_ABLATE_EXTRA_LOADS ; tools/ab.py default <name>
#pragma once

#ifdef ADYPT_ABLATE_EXTRA_TRI_LOADS   // the triangle loads a second time (L1 hits)
#define ADYPT_MEASURE_AFTER_TRI_LOADS(w0)                                                                                       \
	{                                                                                                                           \
		const float4 *x0 = (w0);                                                                                                \
		asm volatile("" : "+v"(x0));                                                                                            \
		const float4 e0 = x0[0], e1 = x0[1], e2 = x0[2];                                                                        \
		asm volatile("" ::"v"(e0.x), "v"(e0.w), "v"(e1.x), "v"(e1.w), "v"(e2.x), "v"(e2.w));                                    \
	}
#else
#define ADYPT_MEASURE_AFTER_TRI_LOADS(w0)
#endif

// the 5 node loads issued twice (L1 hits, no new traffic): price of the vector-memory issue path; _FEW: only from every 4th lane — does
// the price follow lanes or instructions?
#if defined(ADYPT_ABLATE_EXTRA_LOADS) || defined(ADYPT_ABLATE_EXTRA_LOADS_FEW)
#ifdef ADYPT_ABLATE_EXTRA_LOADS_FEW
#define ADYPT_MEASURE_NODE_LOAD_LANES(lane) (((lane) & 3) == 0)
#else
#define ADYPT_MEASURE_NODE_LOAD_LANES(lane) true
#endif
#define ADYPT_MEASURE_AFTER_NODE_LOADS(np, lane)                                                                                \
	if(ADYPT_MEASURE_NODE_LOAD_LANES(lane))                                                                                     \
	{                                                                                                                           \
		const uint4 *np2 = (np);                                                                                                \
		asm volatile("" : "+v"(np2)); /* launder the pointer so the duplicate loads are not CSE'd */                            \
		const uint4 e0 = np2[0], e1 = np2[1], e2 = np2[2], e3 = np2[3], e4 = np2[4];                                            \
		asm volatile("" ::"v"(e0.x), "v"(e0.w), "v"(e1.x), "v"(e1.w), "v"(e2.x), "v"(e2.w), "v"(e3.x), "v"(e3.w), "v"(e4.x), "v"(e4.w)); \
	}
#else
#define ADYPT_MEASURE_AFTER_NODE_LOADS(np, lane)
#endif

#ifdef ADYPT_ABLATE_EXTRA_VALU        // 24 extra independent VALU instructions per slab test (+7 % issued instructions)
#define ADYPT_MEASURE_AFTER_SLAB_TEST(aox, aoy, aoz, aix, aiy)                                                                  \
	{                                                                                                                           \
		float e0 = (aox), e1 = (aoy), e2 = (aoz), e3 = (aix);                                                                   \
		_Pragma("unroll") for(int k = 0; k < 6; ++k)                                                                            \
			asm volatile("v_add_f32 %0, %0, %4\nv_add_f32 %1, %1, %4\nv_add_f32 %2, %2, %4\nv_add_f32 %3, %3, %4" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(aiy)); \
		asm volatile("" ::"v"(e0), "v"(e1), "v"(e2), "v"(e3));                                                                  \
	}
#else
#define ADYPT_MEASURE_AFTER_SLAB_TEST(aox, aoy, aoz, aix, aiy)
#endif

// Wave timeline of ONE traversal launch (tools/wave_timeline.py): when do the persistent waves start, get their first rays, find the queue
// dry, end?  100 MHz wall clock, one record of 4 timestamps per wave in a device array (atomics on shared words would cost more than
// the launch: ~88 same-address atomics per microsecond), read back through adypt_debug_read_timeline (this build only).
#ifdef ADYPT_ABLATE_WAVE_TIMELINE
#define ADYPT_MEASURE_WAVE_TIMELINE
namespace adypt { __device__ unsigned long long g_wave_timeline[8192 * 4]; }
#define ADYPT_MEASURE_WAVE_BEGIN() const unsigned long long tl_begin = wall_clock64(); unsigned long long tl_first = 0, tl_dry = 0
#define ADYPT_MEASURE_WAVE_FIRST_RAYS() if(tl_first == 0) tl_first = wall_clock64()
#define ADYPT_MEASURE_WAVE_QUEUE_DRY() if(tl_dry == 0) tl_dry = wall_clock64()
#define ADYPT_MEASURE_WAVE_END(stats)                                                                                           \
	if(!STATS && lane == 0)                                                                                                     \
	{                                                                                                                           \
		const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;                                                        \
		if(w < 8192) { unsigned long long *o = adypt::g_wave_timeline + (size_t)w * 4; o[0] = tl_begin; o[1] = tl_first; o[2] = tl_dry; o[3] = wall_clock64(); } \
	}
#ifdef ADYPT_TRACER_TU
extern "C" int adypt_debug_read_timeline(unsigned long long *out) { (void)hipDeviceSynchronize(); return hipMemcpyFromSymbol(out, HIP_SYMBOL(adypt::g_wave_timeline), sizeof(unsigned long long) * 8192 * 4) == hipSuccess ? 0 : -3; }
#endif
#endif

// fp16 node bounds: the 48 quantised bounds of a node re-encoded at upload as binary16 (0..255 are exact), so that one
// v_fma_mix_f32 (fp16 source, fp32 fma: the same product and sum, bit-identical) replaces v_cvt_f32_ubyte + half a v_pk_fma_f32 per bound.
// The node grows from 80 to 128 bytes: 8 instead of 5 loads per visit.  Layout: [0] px py pz (ex ey ez imask)  [1] child base, triangle
// base, meta[8]  [2..7] qlox qloy qloz qhix qhiy qhiz, 8 halves each.
#ifdef ADYPT_ABLATE_FP16_NODES
#define ADYPT_MEASURE_FP16_NODES
namespace adypt {
constexpr int kNodeUint4 = 8;
typedef _Float16 adypt_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t slab_test_fp16_nodes(uint4 n1, uint4 n2, uint4 n3, uint4 n4, uint4 n5, uint4 n6, uint4 n7, bool nx, bool ny, bool nz, uint32_t octinv4,
                                                         float aix, float aiy, float aiz, float aox, float aoy, float aoz, float tmin, float hit_t)
{
	uint32_t hitmask = 0;
	// entry / exit planes per axis by ray sign, as in the product (traversal.glsl:92-99)
	const uint4 slox = nx ? n5 : n2, shix = nx ? n2 : n5, sloy = ny ? n6 : n3, shiy = ny ? n3 : n6, sloz = nz ? n7 : n4, shiz = nz ? n4 : n7;
	const uint32_t lx[4] = {slox.x, slox.y, slox.z, slox.w}, hx[4] = {shix.x, shix.y, shix.z, shix.w};
	const uint32_t ly[4] = {sloy.x, sloy.y, sloy.z, sloy.w}, hy[4] = {shiy.x, shiy.y, shiy.z, shiy.w};
	const uint32_t lz[4] = {sloz.x, sloz.y, sloz.z, sloz.w}, hz[4] = {shiz.x, shiz.y, shiz.z, shiz.w};
#pragma unroll
	for(int g = 0; g < 2; ++g)
	{
		const uint32_t meta4 = g ? n1.w : n1.z;
		const uint32_t is_inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
		const uint32_t inner1 = is_inner4 >> 4;
		const uint32_t bit_index4 = (meta4 ^ (octinv4 & ((inner1 << 3) - inner1))) & 0x1f1f1f1fu;
		const uint32_t child_bits4 = (meta4 >> 5) & 0x07070707u;
#pragma unroll
		for(int j = 0; j < 4; ++j)
		{
			const int c = 4 * g + j, w = c >> 1, h = c & 1, sh = 8 * j;
			auto q = [&](const uint32_t *a) { return (float)__builtin_bit_cast(adypt_h2, a[w])[h]; };
			const float tx0 = fmaf(q(lx), aix, aox), tx1 = fmaf(q(hx), aix, aox);
			const float ty0 = fmaf(q(ly), aiy, aoy), ty1 = fmaf(q(hy), aiy, aoy);
			const float tz0 = fmaf(q(lz), aiz, aoz), tz1 = fmaf(q(hz), aiz, aoz);
			const float cmin = fmaxf(fmaxf(tx0, ty0), fmaxf(tz0, tmin));
			const float cmax = fminf(fminf(tx1, ty1), fminf(tz1, hit_t));
			if(cmin <= cmax) hitmask |= ((child_bits4 >> sh) & 0xffu) << ((bit_index4 >> sh) & 0xffu);
		}
	}
	return hitmask;
}
}  // namespace adypt
#define ADYPT_MEASURE_MORE_NODE_REGS() uint4 n5, n6, n7; ADYPT_DEF4(n5); ADYPT_DEF4(n6); ADYPT_DEF4(n7)
#define ADYPT_MEASURE_LOAD_MORE_NODE(np) n5 = (np)[5]; n6 = (np)[6]; n7 = (np)[7]
#ifdef ADYPT_TRACER_TU
#include <cstring>
#include <vector>
namespace adypt {
inline uint16_t half_bits_of_byte(uint32_t k) // exact: 0..255 need 8 significant bits, binary16 has 11
{
	if(k == 0) return 0;
	int e = 31 - __builtin_clz(k);
	return (uint16_t)(((e + 15) << 10) | ((k << (10 - e)) & 0x3ffu));
}
inline std::vector<uint8_t> nodes_as_fp16(const uint8_t *nodes, size_t n)
{
	std::vector<uint8_t> out(n * 128, 0);
	for(size_t i = 0; i < n; ++i)
	{
		const uint8_t *s = nodes + i * 80;
		uint8_t *d = out.data() + i * 128;
		memcpy(d, s, 32);                                   // origin, exponents + imask, child / triangle base, meta
		for(int arr = 0; arr < 6; ++arr)                    // qlox qloy qloz qhix qhiy qhiz
			for(int k = 0; k < 8; ++k)
			{
				const uint16_t h = half_bits_of_byte(s[32 + arr * 8 + k]);
				memcpy(d + 32 + arr * 16 + k * 2, &h, 2);
			}
	}
	return out;
}
}  // namespace adypt
#endif
#endif
