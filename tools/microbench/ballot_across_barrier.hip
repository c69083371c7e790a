// gfx950 probe for the round-3 heisenbug (adypt_amd/csrc/device/shade.hpp append_slot note): a wave-vote mask kept across two workgroup
// barriers, rank = popcount(mask & ((1 << lane) - 1)) afterwards.  In one build of k_gen_primary (exactly 16 VGPRs) whole waves saw rank 0.
// This reproduces the structure — but NOT the failure (0 double claims in 200 launches on MI355X: the trigger needs more than these ingredients) (vote -> LDS counts -> barrier -> thread 0: one device atomic -> LDS bases -> barrier -> rank -> claim a slot)
// with a register budget of 16 and counts the slots claimed more than once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

template <int VARIANT> __global__ __attribute__((amdgpu_num_vgpr(16))) __launch_bounds__(256) void k_claim(uint32_t *counter, uint32_t *claims, const float *junk, float *sink, int n_threads, float scale)
{
	__shared__ uint32_t wave_base[4];
	const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	// a mostly-alive pattern with dead 8-lane groups, like the right / bottom image border
	const bool alive = gid < (uint32_t)n_threads && ((gid * 2654435761u) >> 29) != 0u;
	const unsigned long long mask = __ballot(alive);
	if(lane == 0) wave_base[wave] = (uint32_t)__popcll(mask);
	__syncthreads();
	if(threadIdx.x == 0)
	{
		uint32_t c[4], total = 0;
		for(int w = 0; w < 4; ++w) { c[w] = wave_base[w]; total += c[w]; }
		uint32_t base = total ? atomicAdd(counter, total) : 0u;
		for(int w = 0; w < 4; ++w) { wave_base[w] = base; base += c[w]; }
	}
	__syncthreads();
	// some floating-point work between the barrier and the rank, as the camera-ray arithmetic of k_gen_primary (divisions, sqrt)
	float a = junk[gid & 1023] * scale + 1.0f, b = junk[(gid + 7) & 1023] + 2.0f;
	if(VARIANT >= 1) { a = a / b; b = sqrtf(a * a + b * b); a = (a + 1.0f) / (b + 1.0f); }
	const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
	if(!alive) return;
	atomicAdd(&claims[wave_base[wave] + rank], 1u);
	sink[gid] = a + b;
}

template <int VARIANT> void run(const char *name)
{
	const int n_blocks = 48 * 8 * 16, n = n_blocks * 256;
	uint32_t *counter, *claims; float *junk, *sink;
	CK(hipMalloc(&counter, 4)); CK(hipMalloc(&claims, (size_t)n * 4)); CK(hipMalloc(&junk, 4096)); CK(hipMalloc(&sink, (size_t)n * 4));
	CK(hipMemset(junk, 0, 4096));
	unsigned long long bad_total = 0;
	for(int rep = 0; rep < 200; ++rep)
	{
		CK(hipMemsetAsync(counter, 0, 4, 0)); CK(hipMemsetAsync(claims, 0, (size_t)n * 4, 0));
		hipLaunchKernelGGL(k_claim<VARIANT>, dim3(n_blocks), dim3(256), 0, 0, counter, claims, junk, sink, n, 0.5f);
		static uint32_t *h = (uint32_t *)malloc((size_t)n * 4);
		CK(hipMemcpy(h, claims, (size_t)n * 4, hipMemcpyDeviceToHost));
		for(int i = 0; i < n; ++i) bad_total += h[i] > 1;
	}
	printf("%s: slots claimed more than once over 200 launches of %d workgroups: %llu\n", name, n_blocks, bad_total);
}
int main() { run<0>("plain"); run<1>("with fp work between barrier and rank"); return 0; }
