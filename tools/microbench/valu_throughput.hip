// Developer tool (GPU box): THROUGHPUT of vector-ALU instructions with every SIMD holding several waves (valu_rates.hip measures
// one wave alone = the issue interval of a single instruction stream, 4 cycles; with >= 2 waves per SIMD a full-rate instruction
// issues every 2 cycles on gfx950's 32-lane SIMDs).  Reports cycles per wave-instruction per SIMD at an assumed 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/valu_throughput.bin tools/microbench/valu_throughput.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define BODY(NAME, ASM)                                                                                         \
	__global__ __launch_bounds__(256) void NAME(float *out, float seed, int iters)                               \
	{                                                                                                            \
		float r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, r7 = seed + 7; \
		float a = seed * 0.5f, b = seed + 0.25f;                                                                 \
		for(int it = 0; it < iters; ++it)                                                                        \
		{                                                                                                        \
			_Pragma("unroll") for(int k = 0; k < 8; ++k)                                                         \
				asm volatile(ASM : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b)); \
		}                                                                                                        \
		if(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[0] = 1;                                     \
	}
#define R8(OP, TAIL) OP " %0, %0" TAIL "\n" OP " %1, %1" TAIL "\n" OP " %2, %2" TAIL "\n" OP " %3, %3" TAIL "\n" OP " %4, %4" TAIL "\n" OP " %5, %5" TAIL "\n" OP " %6, %6" TAIL "\n" OP " %7, %7" TAIL "\n"
BODY(k_add_f32, R8("v_add_f32", ", %8"))
BODY(k_fma_f32, R8("v_fma_f32", ", %8, %9"))
BODY(k_max_f32, R8("v_max_f32", ", %8"))
BODY(k_max3_f32, R8("v_max3_f32", ", %8, %9"))
BODY(k_cvt_ubyte, R8("v_cvt_f32_ubyte1", ""))
BODY(k_and_b32, R8("v_and_b32", ", %8"))
BODY(k_lshl, R8("v_lshlrev_b32", ", %8"))
BODY(k_bfe, R8("v_bfe_u32", ", 5, 3"))
BODY(k_lshl_or, R8("v_lshl_or_b32", ", 1, %8"))
BODY(k_or3, R8("v_or3_b32", ", %8, %9"))
BODY(k_perm, R8("v_perm_b32", ", %8, %9"))
BODY(k_mul_lo, R8("v_mul_lo_u32", ", %8"))
BODY(k_rcp, R8("v_rcp_f32", ""))
BODY(k_sqrt, R8("v_sqrt_f32", ""))
BODY(k_div_fixup, R8("v_div_fixup_f32", ", %8, %9"))
BODY(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n")
BODY(k_cmp, "v_cmp_le_f32 vcc, %0, %8\nv_cmp_le_f32 vcc, %1, %8\nv_cmp_le_f32 vcc, %2, %8\nv_cmp_le_f32 vcc, %3, %8\nv_cmp_le_f32 vcc, %4, %8\nv_cmp_le_f32 vcc, %5, %8\nv_cmp_le_f32 vcc, %6, %8\nv_cmp_le_f32 vcc, %7, %8\n")
BODY(k_cmp_sgpr, "v_cmp_le_f32 s[20:21], %0, %8\nv_cmp_le_f32 s[22:23], %1, %8\nv_cmp_le_f32 s[20:21], %2, %8\nv_cmp_le_f32 s[22:23], %3, %8\nv_cmp_le_f32 s[20:21], %4, %8\nv_cmp_le_f32 s[22:23], %5, %8\nv_cmp_le_f32 s[20:21], %6, %8\nv_cmp_le_f32 s[22:23], %7, %8\n")
BODY(k_mov, "v_mov_b32 %0, %8\nv_mov_b32 %1, %8\nv_mov_b32 %2, %8\nv_mov_b32 %3, %8\nv_mov_b32 %4, %8\nv_mov_b32 %5, %8\nv_mov_b32 %6, %8\nv_mov_b32 %7, %8\n")

__global__ __launch_bounds__(256) void k_pk_fma(float *out, float seed, int iters)
{
	typedef float V2 __attribute__((ext_vector_type(2)));
	V2 r0 = {seed, seed}, r1 = r0 + 1.0f, r2 = r0 + 2.0f, r3 = r0 + 3.0f, r4 = r0 + 4.0f, r5 = r0 + 5.0f, r6 = r0 + 6.0f, r7 = r0 + 7.0f, a = r0 * 0.5f, b = r0 + 0.25f;
	for(int it = 0; it < iters; ++it)
#pragma unroll
		for(int k = 0; k < 8; ++k)
			asm volatile("v_pk_fma_f32 %0, %0, %8, %9\nv_pk_fma_f32 %1, %1, %8, %9\nv_pk_fma_f32 %2, %2, %8, %9\nv_pk_fma_f32 %3, %3, %8, %9\nv_pk_fma_f32 %4, %4, %8, %9\nv_pk_fma_f32 %5, %5, %8, %9\nv_pk_fma_f32 %6, %6, %8, %9\nv_pk_fma_f32 %7, %7, %8, %9\n"
						 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
	if(r0.x + r1.x + r2.x + r3.x + r4.y + r5.y + r6.y + r7.y == 12345.678f) out[0] = 1;
}
__global__ __launch_bounds__(256) void k_mad_u64(float *out, unsigned seed, int iters)
{
	unsigned long long r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, r7 = seed + 7;
	unsigned a = seed * 3u;
	for(int it = 0; it < iters; ++it)
#pragma unroll
		for(int k = 0; k < 8; ++k)
			asm volatile("v_mad_u64_u32 %0, vcc, %8, 48, %0\nv_mad_u64_u32 %1, vcc, %8, 48, %1\nv_mad_u64_u32 %2, vcc, %8, 48, %2\nv_mad_u64_u32 %3, vcc, %8, 48, %3\nv_mad_u64_u32 %4, vcc, %8, 48, %4\nv_mad_u64_u32 %5, vcc, %8, 48, %5\nv_mad_u64_u32 %6, vcc, %8, 48, %6\nv_mad_u64_u32 %7, vcc, %8, 48, %7\n"
						 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "vcc");
	if(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345) out[0] = 1;
}

template <class K, class A> static void run(const char *name, K kern, float *d, A arg, int cus)
{
	const int iters = 2000, waves_per_simd = 5;
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipLaunchKernelGGL(kern, dim3(cus * waves_per_simd), dim3(256), 0, 0, d, arg, 10);
	hipEventRecord(a);
	hipLaunchKernelGGL(kern, dim3(cus * waves_per_simd), dim3(256), 0, 0, d, arg, iters);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms = 0; hipEventElapsedTime(&ms, a, b);
	const double instr_per_simd = (double)iters * 64 * waves_per_simd;
	printf("%-18s %7.3f ms   %5.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}

int main()
{
	float *d; hipMalloc(&d, 64);
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	run("v_add_f32", k_add_f32, d, 1.5f, cus); run("v_fma_f32", k_fma_f32, d, 1.5f, cus); run("v_pk_fma_f32", k_pk_fma, d, 1.5f, cus);
	run("v_max_f32", k_max_f32, d, 1.5f, cus); run("v_max3_f32", k_max3_f32, d, 1.5f, cus); run("v_cvt_f32_ubyte1", k_cvt_ubyte, d, 1.5f, cus);
	run("v_and_b32", k_and_b32, d, 1.5f, cus); run("v_lshlrev_b32", k_lshl, d, 1.5f, cus); run("v_bfe_u32", k_bfe, d, 1.5f, cus);
	run("v_lshl_or_b32", k_lshl_or, d, 1.5f, cus); run("v_or3_b32", k_or3, d, 1.5f, cus); run("v_perm_b32", k_perm, d, 1.5f, cus);
	run("v_mul_lo_u32", k_mul_lo, d, 1.5f, cus); run("v_mad_u64_u32", k_mad_u64, d, 3u, cus); run("v_rcp_f32", k_rcp, d, 1.5f, cus);
	run("v_sqrt_f32", k_sqrt, d, 1.5f, cus); run("v_div_fixup_f32", k_div_fixup, d, 1.5f, cus); run("v_cndmask_b32", k_cndmask, d, 1.5f, cus);
	run("v_cmp_le_f32 vcc", k_cmp, d, 1.5f, cus); run("v_cmp_le_f32 sgpr", k_cmp_sgpr, d, 1.5f, cus); run("v_mov_b32", k_mov, d, 1.5f, cus);
	return 0;
}
