// MEASUREMENT ONLY — never part of the shipped library (adypt_amd/csrc/Makefile does not define ADYPT_MEASUREMENT_BUILD).
// Bodies of the hooks in device/traverse.hpp: redundant work that prices one pipeline of the CU at a time.  Results stay correct with
// every one of them (the extra loads / adds feed nothing); the numbers they produced are in profiles/r2_ablations_k_trace.txt.
//   make -C adypt_amd/csrc HIPFLAGS="$(HIPFLAGS) -DADYPT_MEASUREMENT_BUILD -DADYPT_ABLATE_EXTRA_LOADS" OUT=... ; tools/ab.py default <variant>.so
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
