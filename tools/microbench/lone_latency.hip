// Developer tool (GPU box): what one dependent fetch costs a LONE wave — the end of a k_path launch, isolated.
// One wave chases pointers through a table (every lane its own chain, 16-byte records, random permutation), on an otherwise idle chip and with a
// second kernel keeping every CU busy on another stream; one load per step and two independent loads per step (do two round trips overlap?).
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/lone_latency.bin tools/microbench/lone_latency.hip && tools/microbench/lone_latency.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

__global__ __launch_bounds__(64) void k_chase(const uint4 *tab, uint32_t n, int steps, int lanes, unsigned long long *out, int two)
{
	uint32_t i = (threadIdx.x * 2654435761u) % n, j = (threadIdx.x * 40503u + 12345u) % n;
	const bool on = (int)threadIdx.x < lanes;
	const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
	for(int s = 0; s < steps; ++s)
	{
		if(on)
		{
			const uint4 a = tab[i];
			if(two) { const uint4 b = tab[j]; j = b.y % n; }
			i = a.x;
		}
	}
	const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
	if(threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
	if(i == 0xffffffffu || j == 0xffffffffu) out[2] = 1;
}
__global__ __launch_bounds__(256) void k_busy(float *out, int iters, const uint4 *tab, uint32_t n)
{
	float a = threadIdx.x * 0.5f, b = 1.0001f;
	uint32_t i = (blockIdx.x * 256u + threadIdx.x) % n;
	for(int it = 0; it < iters; ++it)
	{
		for(int k = 0; k < 64; ++k) a = fmaf(a, b, 0.5f);
		i = tab[i].x; // a dependent gather now and then: the memory system stays awake too
	}
	if(a == 12345.0f || i == 0xffffffffu) out[0] = a;
}
int main()
{
	const uint32_t n = 4u << 20; // 4 Mi records of 16 bytes = 64 MiB
	std::vector<uint32_t> perm(n); std::iota(perm.begin(), perm.end(), 0u);
	std::mt19937 rng(1); std::shuffle(perm.begin(), perm.end(), rng);
	std::vector<uint4> h(n);
	for(uint32_t k = 0; k < n; ++k) h[perm[k]] = make_uint4(perm[(k + 1) % n], perm[(k + 7) % n], 0, 0);
	uint4 *d; hipMalloc(&d, (size_t)n * 16); hipMemcpy(d, h.data(), (size_t)n * 16, hipMemcpyHostToDevice);
	unsigned long long *o; hipMalloc(&o, 64); float *f; hipMalloc(&f, 64);
	hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	for(int busy = 0; busy < 2; ++busy)
		for(int two = 0; two < 2; ++two)
			for(int lanes : {1, 16, 64})
			{
				const int steps = 2000;
				if(busy) hipLaunchKernelGGL(k_busy, dim3(prop.multiProcessorCount * 6), dim3(256), 0, s2, f, 40000, d, n);
				hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, s1, d, n, 200, lanes, o, two); // warm
				hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, s1, d, n, steps, lanes, o, two);
				hipStreamSynchronize(s1);
				unsigned long long r[2]; hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
				hipDeviceSynchronize();
				printf("{\"chip\": \"%s\", \"loads_per_step\": %d, \"lanes\": %d, \"cycles_per_step\": %.0f, \"ns_per_step\": %.0f, \"shader_GHz\": %.2f}\n", busy ? "busy" : "idle", two ? 2 : 1, lanes,
				       (double)r[0] / steps, (double)r[1] * 10.0 / steps, (double)r[0] / ((double)r[1] * 10.0));
				fflush(stdout);
			}
	return 0;
}
