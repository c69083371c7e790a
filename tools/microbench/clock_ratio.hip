// Developer tool (GPU box): is s_memtime the shader clock?  One wave per SIMD runs a dependent fma chain; s_memtime and s_memrealtime (100 MHz)
// are read before and after.  Prints delta(s_memtime) / delta(s_memrealtime) x 100 MHz for a light and a fully loaded chip.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/clock_ratio.bin tools/microbench/clock_ratio.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long *o, float seed, int iters)
{
	const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
	float a = seed, b = seed + 1, c = seed + 2, d = seed + 3;
	for(int i = 0; i < iters; ++i) { a = fmaf(a, 1.0001f, 0.5f); b = fmaf(b, 1.0001f, 0.5f); c = fmaf(c, 1.0001f, 0.5f); d = fmaf(d, 1.0001f, 0.5f); }
	const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
	if(a + b + c + d == 12345.0f) o[4] = 1;
	if(blockIdx.x == 0 && threadIdx.x == 0) { o[0] = c1 - c0; o[1] = r1 - r0; }
}
int main()
{
	unsigned long long *d, h[2];
	hipMalloc(&d, 64);
	for(int blocks : {1, 256 * 6})
	{
		hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1.5f, 2000000);
		hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
		printf("%5d workgroups: s_memtime %llu, s_memrealtime %llu -> %.3f GHz if s_memtime counts shader cycles\n", blocks, h[0], h[1], (double)h[0] / (double)h[1] * 0.1);
	}
	return 0;
}
