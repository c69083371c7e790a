// Developer tool (GPU box): issue cost in cycles of the vector instructions the traversal kernel is made of, one wave per
// SIMD, independent instructions (8 register chains), measured with s_memtime around an unrolled loop.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/microbench/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY(NAME, ASM)                                                                                         \
	__global__ void NAME(unsigned long long *out, float seed)                                                    \
	{                                                                                                            \
		float r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, r7 = seed + 7; \
		float a = seed * 0.5f, b = seed + 0.25f;                                                                 \
		unsigned long long t0 = __builtin_readcyclecounter();                                                    \
		for(int it = 0; it < 64; ++it)                                                                           \
		{                                                                                                        \
			_Pragma("unroll") for(int k = 0; k < 4; ++k)                                                         \
			{                                                                                                    \
				asm volatile(ASM : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b)); \
			}                                                                                                    \
		}                                                                                                        \
		unsigned long long t1 = __builtin_readcyclecounter();                                                    \
		if(threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                          \
		if(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[1000] = 1;                                   \
	}

#define I8(OP) OP " %0\n" OP " %1\n" OP " %2\n" OP " %3\n" OP " %4\n" OP " %5\n" OP " %6\n" OP " %7\n"
// one-destination forms: "op dst, dst, a" etc. are written out per instruction
BODY(k_add_u32, "v_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_add_u32 %3, %3, %8\nv_add_u32 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_add_u32 %6, %6, %8\nv_add_u32 %7, %7, %8\n")
BODY(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %8\nv_mul_lo_u32 %1, %1, %8\nv_mul_lo_u32 %2, %2, %8\nv_mul_lo_u32 %3, %3, %8\nv_mul_lo_u32 %4, %4, %8\nv_mul_lo_u32 %5, %5, %8\nv_mul_lo_u32 %6, %6, %8\nv_mul_lo_u32 %7, %7, %8\n")
BODY(k_fma_f32, "v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\nv_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n")
BODY(k_cvt_ubyte, "v_cvt_f32_ubyte1 %0, %0\nv_cvt_f32_ubyte1 %1, %1\nv_cvt_f32_ubyte1 %2, %2\nv_cvt_f32_ubyte1 %3, %3\nv_cvt_f32_ubyte1 %4, %4\nv_cvt_f32_ubyte1 %5, %5\nv_cvt_f32_ubyte1 %6, %6\nv_cvt_f32_ubyte1 %7, %7\n")
BODY(k_rcp_f32, "v_rcp_f32 %0, %0\nv_rcp_f32 %1, %1\nv_rcp_f32 %2, %2\nv_rcp_f32 %3, %3\nv_rcp_f32 %4, %4\nv_rcp_f32 %5, %5\nv_rcp_f32 %6, %6\nv_rcp_f32 %7, %7\n")
BODY(k_max3_f32, "v_max3_f32 %0, %0, %8, %9\nv_max3_f32 %1, %1, %8, %9\nv_max3_f32 %2, %2, %8, %9\nv_max3_f32 %3, %3, %8, %9\nv_max3_f32 %4, %4, %8, %9\nv_max3_f32 %5, %5, %8, %9\nv_max3_f32 %6, %6, %8, %9\nv_max3_f32 %7, %7, %8, %9\n")
BODY(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n")
BODY(k_bfe_u32, "v_bfe_u32 %0, %0, 5, 3\nv_bfe_u32 %1, %1, 5, 3\nv_bfe_u32 %2, %2, 5, 3\nv_bfe_u32 %3, %3, 5, 3\nv_bfe_u32 %4, %4, 5, 3\nv_bfe_u32 %5, %5, 5, 3\nv_bfe_u32 %6, %6, 5, 3\nv_bfe_u32 %7, %7, 5, 3\n")
BODY(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %8\nv_lshl_add_u32 %1, %1, 1, %8\nv_lshl_add_u32 %2, %2, 1, %8\nv_lshl_add_u32 %3, %3, 1, %8\nv_lshl_add_u32 %4, %4, 1, %8\nv_lshl_add_u32 %5, %5, 1, %8\nv_lshl_add_u32 %6, %6, 1, %8\nv_lshl_add_u32 %7, %7, 1, %8\n")
BODY(k_div_fixup, "v_div_fixup_f32 %0, %0, %8, %9\nv_div_fixup_f32 %1, %1, %8, %9\nv_div_fixup_f32 %2, %2, %8, %9\nv_div_fixup_f32 %3, %3, %8, %9\nv_div_fixup_f32 %4, %4, %8, %9\nv_div_fixup_f32 %5, %5, %8, %9\nv_div_fixup_f32 %6, %6, %8, %9\nv_div_fixup_f32 %7, %7, %8, %9\n")
BODY(k_cmp_le, "v_cmp_le_f32 vcc, %0, %8\nv_cmp_le_f32 vcc, %1, %8\nv_cmp_le_f32 vcc, %2, %8\nv_cmp_le_f32 vcc, %3, %8\nv_cmp_le_f32 vcc, %4, %8\nv_cmp_le_f32 vcc, %5, %8\nv_cmp_le_f32 vcc, %6, %8\nv_cmp_le_f32 vcc, %7, %8\n")

// 64-bit destinations: separate kernels with register pairs
__global__ void k_pk_fma(unsigned long long *out, float seed)
{
	typedef float V2 __attribute__((ext_vector_type(2)));
	V2 r0 = {seed, seed}, r1 = r0 + 1.0f, r2 = r0 + 2.0f, r3 = r0 + 3.0f, r4 = r0 + 4.0f, r5 = r0 + 5.0f, r6 = r0 + 6.0f, r7 = r0 + 7.0f, a = r0 * 0.5f, b = r0 + 0.25f;
	unsigned long long t0 = __builtin_readcyclecounter();
	for(int it = 0; it < 64; ++it)
	{
#pragma unroll
		for(int k = 0; k < 4; ++k)
			asm volatile("v_pk_fma_f32 %0, %0, %8, %9\nv_pk_fma_f32 %1, %1, %8, %9\nv_pk_fma_f32 %2, %2, %8, %9\nv_pk_fma_f32 %3, %3, %8, %9\nv_pk_fma_f32 %4, %4, %8, %9\nv_pk_fma_f32 %5, %5, %8, %9\nv_pk_fma_f32 %6, %6, %8, %9\nv_pk_fma_f32 %7, %7, %8, %9\n"
						 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
	}
	unsigned long long t1 = __builtin_readcyclecounter();
	if(threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
	if(r0.x + r1.x + r2.x + r3.x + r4.y + r5.y + r6.y + r7.y == 12345.678f) out[1000] = 1;
}
__global__ void k_mad_u64(unsigned long long *out, unsigned seed)
{
	unsigned long long r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, r7 = seed + 7;
	unsigned a = seed * 3u;
	unsigned long long t0 = __builtin_readcyclecounter();
	for(int it = 0; it < 64; ++it)
	{
#pragma unroll
		for(int k = 0; k < 4; ++k)
			asm volatile("v_mad_u64_u32 %0, vcc, %8, 48, %0\nv_mad_u64_u32 %1, vcc, %8, 48, %1\nv_mad_u64_u32 %2, vcc, %8, 48, %2\nv_mad_u64_u32 %3, vcc, %8, 48, %3\nv_mad_u64_u32 %4, vcc, %8, 48, %4\nv_mad_u64_u32 %5, vcc, %8, 48, %5\nv_mad_u64_u32 %6, vcc, %8, 48, %6\nv_mad_u64_u32 %7, vcc, %8, 48, %7\n"
						 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "vcc");
	}
	unsigned long long t1 = __builtin_readcyclecounter();
	if(threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
	if(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345) out[1000] = 1;
}

template <class K, class... A> static void run(const char *name, K kern, unsigned long long *d, A... args)
{
	std::vector<unsigned long long> h(4);
	// one wave per SIMD would need placement control; one 64-thread workgroup on an otherwise idle GPU measures one wave's stream
	hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d, args...);
	hipDeviceSynchronize();
	hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d, args...);
	hipDeviceSynchronize();
	hipMemcpy(h.data(), d, 8, hipMemcpyDeviceToHost);
	// s_memtime / readcyclecounter ticks at the 100 MHz reference clock on gfx9: report relative to v_add_u32 as well
	printf("%-16s %8llu ticks for %d instructions\n", name, h[0], 64 * 4 * 8);
}

int main()
{
	unsigned long long *d;
	hipMalloc(&d, 8192);
	hipMemset(d, 0, 8192);
	run("v_add_u32", k_add_u32, d, 1.5f);
	run("v_mul_lo_u32", k_mul_lo_u32, d, 1.5f);
	run("v_mad_u64_u32", k_mad_u64, d, 3u);
	run("v_fma_f32", k_fma_f32, d, 1.5f);
	run("v_pk_fma_f32", k_pk_fma, d, 1.5f);
	run("v_cvt_f32_ubyte1", k_cvt_ubyte, d, 1.5f);
	run("v_rcp_f32", k_rcp_f32, d, 1.5f);
	run("v_max3_f32", k_max3_f32, d, 1.5f);
	run("v_cndmask_b32", k_cndmask, d, 1.5f);
	run("v_bfe_u32", k_bfe_u32, d, 1.5f);
	run("v_lshl_add_u32", k_lshl_add, d, 1.5f);
	run("v_div_fixup_f32", k_div_fixup, d, 1.5f);
	run("v_cmp_le_f32", k_cmp_le, d, 1.5f);
	return 0;
}
