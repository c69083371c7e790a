// Developer tool (GPU box): is  r = v_rcp_f32(x); e = fma(-x, r, 1); r' = fma(e, r, r)  the correctly rounded 1/x?  Exhaustive over all
// 2^32 bit patterns against the compiler's IEEE division (v_div_scale / v_div_fmas / v_div_fixup sequence), by exponent class.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(unsigned long long *mism /* [256] by biased exponent */, unsigned long long *mism2)
{
	const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 256ull;
	for(uint32_t i = 0; i < 256; ++i)
	{
		const uint32_t bits = (uint32_t)(base + i);
		const float x = __uint_as_float(bits);
		const float want = 1.0f / x;
		const float r = __builtin_amdgcn_rcpf(x);
		const float e = __builtin_fmaf(-x, r, 1.0f);
		const float r1 = __builtin_fmaf(e, r, r);
		const float e2 = __builtin_fmaf(-x, r1, 1.0f);
		const float r2 = __builtin_fmaf(e2, r1, r1);
		const uint32_t ex = (bits >> 23) & 255u;
		if(__float_as_uint(r1) != __float_as_uint(want) && !(want != want && r1 != r1)) atomicAdd(&mism[ex], 1ull);
		if(__float_as_uint(r2) != __float_as_uint(want) && !(want != want && r2 != r2)) atomicAdd(&mism2[ex], 1ull);
	}
}
int main()
{
	unsigned long long *d, h[512];
	hipMalloc(&d, sizeof(h)); hipMemset(d, 0, sizeof(h));
	hipLaunchKernelGGL(k, dim3(65536), dim3(256), 0, 0, d, d + 256);
	hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	unsigned long long t1 = 0, t2 = 0;
	for(int e = 0; e < 256; ++e) { t1 += h[e]; t2 += h[256 + e]; if(h[e] || h[256 + e]) printf("exponent %3d: one step %llu mismatches, two steps %llu\n", e, h[e], h[256 + e]); }
	printf("total: one Newton step %llu mismatches, two steps %llu, of 2^32\n", t1, t2);
	return 0;
}
