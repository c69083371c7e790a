// Developer tool (GPU box): follow-up to valu_throughput.hip — select / compare / min-max forms the slab test is made of.
#include <hip/hip_runtime.h>
#include <cstdio>
#define BODY(NAME, PRE, ASM)                                                                                    \
	__global__ __launch_bounds__(256) void NAME(float *out, float seed, int iters)                               \
	{                                                                                                            \
		float r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, r7 = seed + 7; \
		float a = seed * 0.5f, b = seed + 0.25f;                                                                 \
		asm volatile(PRE ::: "vcc", "s20", "s21", "s22", "s23");                                                 \
		for(int it = 0; it < iters; ++it)                                                                        \
		{                                                                                                        \
			_Pragma("unroll") for(int k = 0; k < 8; ++k)                                                         \
				asm volatile(ASM : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc", "s20", "s21", "s22", "s23"); \
		}                                                                                                        \
		if(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[0] = 1;                                     \
	}
#define R8(OP, TAIL) OP " %0, %0" TAIL "\n" OP " %1, %1" TAIL "\n" OP " %2, %2" TAIL "\n" OP " %3, %3" TAIL "\n" OP " %4, %4" TAIL "\n" OP " %5, %5" TAIL "\n" OP " %6, %6" TAIL "\n" OP " %7, %7" TAIL "\n"
BODY(k_cnd_vcc, "s_mov_b32 vcc_lo, 0x55555555\ns_mov_b32 vcc_hi, 0x55555555", R8("v_cndmask_b32", ", %8, vcc"))
BODY(k_cnd_sgpr, "s_mov_b32 s20, 0x55555555\ns_mov_b32 s21, 0x55555555", R8("v_cndmask_b32_e64", ", %8, s[20:21]"))
BODY(k_cnd_vcc_zero, "s_mov_b64 vcc, 0", R8("v_cndmask_b32", ", %8, vcc"))
BODY(k_cmp_cnd, "", "v_cmp_le_f32 vcc, %0, %8\nv_cndmask_b32 %1, %1, %9, vcc\nv_cmp_le_f32 vcc, %2, %8\nv_cndmask_b32 %3, %3, %9, vcc\nv_cmp_le_f32 vcc, %4, %8\nv_cndmask_b32 %5, %5, %9, vcc\nv_cmp_le_f32 vcc, %6, %8\nv_cndmask_b32 %7, %7, %9, vcc\n")
BODY(k_min_f32, "", R8("v_min_f32", ", %8"))
BODY(k_mul_f32, "", R8("v_mul_f32", ", %8"))
BODY(k_sub_f32, "", R8("v_sub_f32", ", %8"))
BODY(k_or_b32, "", R8("v_or_b32", ", %8"))
BODY(k_xor_b32, "", R8("v_xor_b32", ", %8"))
BODY(k_add_u32, "", R8("v_add_u32", ", %8"))
BODY(k_lshl_add, "", R8("v_lshl_add_u32", ", 3, %8"))
BODY(k_bfi, "", R8("v_bfi_b32", ", %8, %9"))
BODY(k_and_or, "", R8("v_and_or_b32", ", %8, %9"))
BODY(k_ashr, "", R8("v_ashrrev_i32", ", %8"))
BODY(k_med3, "", R8("v_med3_f32", ", %8, %9"))
BODY(k_fmamix, "", R8("v_fma_mix_f32", ", %8, %9"))
BODY(k_bcnt, "", R8("v_bcnt_u32_b32", ", %8"))
BODY(k_ffbh, "", "v_ffbh_u32 %0, %0\nv_ffbh_u32 %1, %1\nv_ffbh_u32 %2, %2\nv_ffbh_u32 %3, %3\nv_ffbh_u32 %4, %4\nv_ffbh_u32 %5, %5\nv_ffbh_u32 %6, %6\nv_ffbh_u32 %7, %7\n")
BODY(k_cvt_u32, "", "v_cvt_f32_u32 %0, %0\nv_cvt_f32_u32 %1, %1\nv_cvt_f32_u32 %2, %2\nv_cvt_f32_u32 %3, %3\nv_cvt_f32_u32 %4, %4\nv_cvt_f32_u32 %5, %5\nv_cvt_f32_u32 %6, %6\nv_cvt_f32_u32 %7, %7\n")
BODY(k_sdwa_cvt, "", "v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\nv_cvt_f32_u32_sdwa %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\nv_cvt_f32_u32_sdwa %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\nv_cvt_f32_u32_sdwa %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\nv_cvt_f32_u32_sdwa %4, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\nv_cvt_f32_u32_sdwa %5, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\nv_cvt_f32_u32_sdwa %6, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\nv_cvt_f32_u32_sdwa %7, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n")
template <class K> static void run(const char *name, K kern, float *d, int cus)
{
	const int iters = 2000, waves_per_simd = 5;
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipLaunchKernelGGL(kern, dim3(cus * waves_per_simd), dim3(256), 0, 0, d, 1.5f, 10);
	hipEventRecord(a);
	hipLaunchKernelGGL(kern, dim3(cus * waves_per_simd), dim3(256), 0, 0, d, 1.5f, iters);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms = 0; hipEventElapsedTime(&ms, a, b);
	printf("%-22s %7.3f ms   %5.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * waves_per_simd));
}
int main()
{
	float *d; hipMalloc(&d, 64);
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	run("v_cndmask vcc=0x55..", k_cnd_vcc, d, cus); run("v_cndmask sgpr=0x55..", k_cnd_sgpr, d, cus); run("v_cndmask vcc=0", k_cnd_vcc_zero, d, cus);
	run("v_cmp + v_cndmask", k_cmp_cnd, d, cus); run("v_min_f32", k_min_f32, d, cus); run("v_mul_f32", k_mul_f32, d, cus); run("v_sub_f32", k_sub_f32, d, cus);
	run("v_or_b32", k_or_b32, d, cus); run("v_xor_b32", k_xor_b32, d, cus); run("v_add_u32", k_add_u32, d, cus); run("v_lshl_add_u32", k_lshl_add, d, cus);
	run("v_bfi_b32", k_bfi, d, cus); run("v_and_or_b32", k_and_or, d, cus); run("v_ashrrev_i32", k_ashr, d, cus); run("v_med3_f32", k_med3, d, cus);
	run("v_fma_mix_f32", k_fmamix, d, cus); run("v_bcnt_u32_b32", k_bcnt, d, cus); run("v_ffbh_u32", k_ffbh, d, cus); run("v_cvt_f32_u32", k_cvt_u32, d, cus);
	run("v_cvt_f32_u32 sdwa", k_sdwa_cvt, d, cus);
	return 0;
}
