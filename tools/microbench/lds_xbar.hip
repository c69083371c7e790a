// Developer tool (GPU box): what the LDS crossbar costs.  ds_bpermute_b32 / ds_read / ds_write issue rates per CU (6 waves per SIMD, every SIMD
// busy), alone and interleaved with vector-ALU work the way the trip issues them, against the DPP moves they replace.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/lds_xbar.bin tools/microbench/lds_xbar.hip && tools/microbench/lds_xbar.bin
// Output: one JSON line per row; cycles are shader cycles measured with s_memtime inside the kernel (wave 0 of workgroup 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define KERNEL(NAME, BODY8, NOPS)                                                                                       \
	__global__ __launch_bounds__(256) void NAME(unsigned long long *out, float seed, int iters, int pattern)             \
	{                                                                                                                   \
		__shared__ uint32_t lds[1024];                                                                                  \
		const uint32_t lane = threadIdx.x & 63u;                                                                        \
		uint32_t a0 = (pattern == 0 ? lane : pattern == 1 ? (lane * 17u + 5u) & 63u : (lane ^ 1u)) << 2;                \
		uint32_t la = (threadIdx.x * 4u) & 4095u;                                                                       \
		float r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, r7 = seed + 7; \
		float f = seed * 0.5f;                                                                                          \
		lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 256] = 0; lds[threadIdx.x + 512] = 0; lds[threadIdx.x + 768] = 0; \
		__syncthreads();                                                                                                \
		const unsigned long long t0 = __builtin_readcyclecounter();                                                     \
		for(int it = 0; it < iters; ++it)                                                                               \
		{                                                                                                               \
			asm volatile(BODY8 "s_waitcnt lgkmcnt(0)\n" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a0), "v"(la), "v"(f) : "memory"); \
		}                                                                                                               \
		const unsigned long long t1 = __builtin_readcyclecounter();                                                     \
		if(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[1] = 1;                                            \
		if(blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;                                                       \
	}

#define BP(D) "ds_bpermute_b32 " D ", %8, " D "\n"
#define RD(D) "ds_read_b32 " D ", %9\n"
#define WR8(D) "ds_write_b8 %9, " D "\n"
#define DPP(D) "v_mov_b32_dpp " D ", " D " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define FMA(D) "v_fma_f32 " D ", " D ", %10, %10\n"
#define CVT(D) "v_max_f32 " D ", " D ", %10\n"

KERNEL(k_bpermute8, BP("%0") BP("%1") BP("%2") BP("%3") BP("%4") BP("%5") BP("%6") BP("%7"), 8)
KERNEL(k_read8, RD("%0") RD("%1") RD("%2") RD("%3") RD("%4") RD("%5") RD("%6") RD("%7"), 8)
KERNEL(k_write8, WR8("%0") WR8("%1") WR8("%2") WR8("%3") WR8("%4") WR8("%5") WR8("%6") WR8("%7"), 8)
KERNEL(k_dpp8, DPP("%0") DPP("%1") DPP("%2") DPP("%3") DPP("%4") DPP("%5") DPP("%6") DPP("%7"), 8)
KERNEL(k_fma8, FMA("%0") FMA("%1") FMA("%2") FMA("%3") FMA("%4") FMA("%5") FMA("%6") FMA("%7"), 8)
KERNEL(k_max8, CVT("%0") CVT("%1") CVT("%2") CVT("%3") CVT("%4") CVT("%5") CVT("%6") CVT("%7"), 8)
// the trip's mix: ~300 vector instructions (here 32 "normal" v_max + 32 full-rate v_fma per group of 8 LDS ops, scaled: 64 VALU : N LDS)
#define V8 CVT("%0") FMA("%1") CVT("%2") FMA("%3") CVT("%4") FMA("%5") CVT("%6") FMA("%7")
#define V64 V8 V8 V8 V8 V8 V8 V8 V8
KERNEL(k_valu64, V64, 64)
KERNEL(k_valu64_bp4, V64 BP("%0") BP("%1") BP("%2") BP("%3"), 68)
KERNEL(k_valu64_bp8, V64 BP("%0") BP("%1") BP("%2") BP("%3") BP("%4") BP("%5") BP("%6") BP("%7"), 72)
KERNEL(k_valu64_dpp4, V64 DPP("%0") DPP("%1") DPP("%2") DPP("%3"), 68)

template <class K> static void run(const char *name, K kern, unsigned long long *d, int cus, int n_ops, int pattern, int waves_per_simd)
{
	const int iters = 4000;
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipLaunchKernelGGL(kern, dim3(cus * waves_per_simd), dim3(256), 0, 0, d, 1.5f, 10, pattern);
	hipEventRecord(a);
	hipLaunchKernelGGL(kern, dim3(cus * waves_per_simd), dim3(256), 0, 0, d, 1.5f, iters, pattern);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms = 0; hipEventElapsedTime(&ms, a, b);
	unsigned long long cyc = 0; hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
	// per CU: waves_per_simd workgroups of 4 waves = 4 * waves_per_simd waves, each issuing n_ops per iteration
	const double per_wave_iter = (double)cyc / iters;
	printf("{\"kernel\": \"%s\", \"pattern\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"cycles_per_iteration_of_one_wave\": %.1f, \"ops_per_iteration\": %d, "
	       "\"cycles_per_op_per_simd\": %.2f, \"cycles_per_op_per_cu\": %.2f}\n",
	       name, pattern, waves_per_simd, ms, per_wave_iter, n_ops, per_wave_iter / (n_ops * waves_per_simd), per_wave_iter / (n_ops * waves_per_simd * 4));
	fflush(stdout);
}
int main(int argc, char **argv)
{
	unsigned long long *d; hipMalloc(&d, 64);
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	for(int w : {6, 1})
	{
		for(int p = 0; p < 3; ++p) run("ds_bpermute_b32 x8", k_bpermute8, d, cus, 8, p, w);
		run("ds_read_b32 x8", k_read8, d, cus, 8, 0, w);
		run("ds_write_b8 x8", k_write8, d, cus, 8, 0, w);
		run("v_mov_b32_dpp x8", k_dpp8, d, cus, 8, 0, w);
		run("v_fma_f32 x8", k_fma8, d, cus, 8, 0, w);
		run("v_max_f32 x8", k_max8, d, cus, 8, 0, w);
		run("64 valu", k_valu64, d, cus, 64, 0, w);
		run("64 valu + 4 bpermute", k_valu64_bp4, d, cus, 68, 1, w);
		run("64 valu + 8 bpermute", k_valu64_bp8, d, cus, 72, 1, w);
		run("64 valu + 4 dpp", k_valu64_dpp4, d, cus, 68, 0, w);
	}
	return 0;
}
