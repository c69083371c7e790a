// Developer tool (GPU box): the roof of k_path's memory access pattern — what 256 CUs can GATHER from a table of small records when every
// lane follows its own chain of dependent, uniformly random record reads (the two dependent fetch sites of the traversal,
// shaders/traversal.glsl:69-74 node and :216-218 Woop triangle, and the triangle record of FetchInfo, pathtracer.glsl:76-84).
//   records : 80 B read as 5 x dwordx4 (WideBVHNode), 48 B as 3 x dwordx4 (Woop), 128 B as 8 x dwordx4 (repacked triangle record)
//   tables  : sizes given on the command line in MiB (default 20, 600, 2700: cache-resident BVH, the 10 M-triangle BVH, BVH + per-reference copy)
//   launch  : k_path's shape — 256-thread workgroups, `wgs` per CU pinned by their LDS allocation (6 -> 24 waves per CU), persistent: every lane
//             makes `steps` reads on each of `chains` independent chains (1 = a ray's own chain; 2, 4 = the memory-level parallelism more
//             chains per lane would buy)
// Output: one JSON line per configuration (useful GB/s = records x record bytes / time, records/s, latency per dependent read).
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/gather_roof.bin tools/microbench/gather_roof.hip
//   ./gather_roof.bin [--pmc] [--sizes 20,600,2700] [--wgs 6,8] [--chains 1,2,4] [--steps 256]
// --pmc: exactly ONE launch per configuration, no warm-up (the dispatch order then maps rocprofv3's counter rows to the JSON lines).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define HIP_OK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while(0)

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
	x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
	return x;
}

// record i: dword 0 = a random word (the chain's next hop is a hash of it and of the chain's salt), the rest noise
template <int Q>
__global__ void k_fill(uint4 *table, uint32_t n_rec, uint32_t seed)
{
	for(uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += gridDim.x * blockDim.x)
	{
		const uint32_t h = mix32(i * 0x9e3779b9u + seed);
		const uint32_t next = (uint32_t)(((unsigned long long)h * n_rec) >> 32);
#pragma unroll
		for(int q = 0; q < Q; ++q) table[(size_t)i * Q + q] = make_uint4(q == 0 ? next : h + q, h ^ 0x55u, i, (uint32_t)q);
	}
}

template <int Q, int CHAINS>
__global__ __launch_bounds__(256) void k_gather(const uint4 *__restrict__ table, uint32_t n_rec, int steps, uint32_t seed, uint32_t *sink)
{
	extern __shared__ uint32_t lds_pin[]; // (sizes the workgroups per CU, nothing else)
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	// The next hop is a function of the record read AND of the chain: walkers of one random map x -> next(x) coalesce (after k steps they
	// occupy ~2N/k distinct records, a cache-resident set), walkers that each salt the map do not.
	uint32_t idx[CHAINS], salt[CHAINS], acc = 0;
#pragma unroll
	for(int c = 0; c < CHAINS; ++c) salt[c] = mix32((gid * CHAINS + c) * 0x9e3779b9u + seed * 0x85ebca6bu + 1u);
#pragma unroll
	for(int c = 0; c < CHAINS; ++c) idx[c] = (uint32_t)(((unsigned long long)mix32(gid * CHAINS + c + seed) * n_rec) >> 32);
	for(int s = 0; s < steps; ++s)
	{
		uint4 r[CHAINS][Q];
#pragma unroll
		for(int c = 0; c < CHAINS; ++c)
		{
			const uint4 *p = table + (size_t)idx[c] * Q;
#pragma unroll
			for(int q = 0; q < Q; ++q) r[c][q] = p[q];
		}
#pragma unroll
		for(int c = 0; c < CHAINS; ++c)
		{
			idx[c] = (uint32_t)(((unsigned long long)mix32(r[c][0].x ^ salt[c]) * n_rec) >> 32);
#pragma unroll
			for(int q = 0; q < Q; ++q) acc += (r[c][q].y ^ r[c][q].w) + (r[c][q].z ^ (q ? r[c][q].x : 0u)); // every dword is consumed: the loads stay dwordx4
		}
	}
	if(acc == 0x12345678u) { sink[0] = acc; lds_pin[0] = acc; }
}

struct Cfg { int q, chains, wgs, steps; size_t mib; };

template <int Q, int CHAINS>
static void launch(const uint4 *t, uint32_t n_rec, const Cfg &c, int blocks, size_t lds, uint32_t seed, uint32_t *sink, hipStream_t st)
{
	HIP_OK(hipFuncSetAttribute((const void *)k_gather<Q, CHAINS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipLaunchKernelGGL((k_gather<Q, CHAINS>), dim3(blocks), dim3(256), lds, st, t, n_rec, c.steps, seed, sink);
}
template <int Q>
static void launch_q(const uint4 *t, uint32_t n_rec, const Cfg &c, int blocks, size_t lds, uint32_t seed, uint32_t *sink, hipStream_t st)
{
	switch(c.chains)
	{
	case 1: launch<Q, 1>(t, n_rec, c, blocks, lds, seed, sink, st); break;
	case 2: launch<Q, 2>(t, n_rec, c, blocks, lds, seed, sink, st); break;
	case 4: launch<Q, 4>(t, n_rec, c, blocks, lds, seed, sink, st); break;
	default: fprintf(stderr, "chains must be 1, 2 or 4\n"); exit(2);
	}
}
static void launch_any(const uint4 *t, uint32_t n_rec, const Cfg &c, int blocks, size_t lds, uint32_t seed, uint32_t *sink, hipStream_t st)
{
	switch(c.q)
	{
	case 3: launch_q<3>(t, n_rec, c, blocks, lds, seed, sink, st); break;
	case 5: launch_q<5>(t, n_rec, c, blocks, lds, seed, sink, st); break;
	case 8: launch_q<8>(t, n_rec, c, blocks, lds, seed, sink, st); break;
	}
}

static std::vector<int> ints(const char *s)
{
	std::vector<int> v;
	for(const char *p = s; *p;) { v.push_back(atoi(p)); while(*p && *p != ',') ++p; if(*p) ++p; }
	return v;
}

int main(int argc, char **argv)
{
	bool pmc = false;
	std::vector<int> sizes = {20, 600, 2700}, wgs = {6}, chains = {1, 2}, quads = {5, 3, 8};
	int steps = 256;
	for(int i = 1; i < argc; ++i)
	{
		if(!strcmp(argv[i], "--pmc")) pmc = true;
		else if(!strcmp(argv[i], "--sizes") && i + 1 < argc) sizes = ints(argv[++i]);
		else if(!strcmp(argv[i], "--wgs") && i + 1 < argc) wgs = ints(argv[++i]);
		else if(!strcmp(argv[i], "--chains") && i + 1 < argc) chains = ints(argv[++i]);
		else if(!strcmp(argv[i], "--quads") && i + 1 < argc) quads = ints(argv[++i]);
		else if(!strcmp(argv[i], "--steps") && i + 1 < argc) steps = atoi(argv[++i]);
		else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
	}
	hipDeviceProp_t prop;
	HIP_OK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	const size_t lds_cu = prop.maxSharedMemoryPerMultiProcessor;
	hipStream_t st;
	HIP_OK(hipStreamCreate(&st));
	uint32_t *sink;
	HIP_OK(hipMalloc(&sink, 64));
	hipEvent_t e0, e1;
	HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
	int dispatch = 0;
	for(int mib : sizes)
	{
		for(int q : quads)
		{
			if(q != 3 && q != 5 && q != 8) { fprintf(stderr, "quads must be 3, 5 or 8\n"); return 2; }
			const size_t rec = (size_t)q * 16;
			const uint32_t n_rec = (uint32_t)(((size_t)mib << 20) / rec);
			uint4 *table;
			HIP_OK(hipMalloc(&table, (size_t)n_rec * rec));
			switch(q)
			{
			case 3: hipLaunchKernelGGL((k_fill<3>), dim3(cus * 8), dim3(256), 0, st, table, n_rec, 12345u); break;
			case 5: hipLaunchKernelGGL((k_fill<5>), dim3(cus * 8), dim3(256), 0, st, table, n_rec, 12345u); break;
			case 8: hipLaunchKernelGGL((k_fill<8>), dim3(cus * 8), dim3(256), 0, st, table, n_rec, 12345u); break;
			}
			HIP_OK(hipStreamSynchronize(st));
			for(int w : wgs)
				for(int ch : chains)
				{
					Cfg c{q, ch, w, steps, (size_t)mib};
					const int blocks = cus * w;
					// LDS per workgroup such that exactly w fit a CU (k_path: 26 304 B of the 26 624 a CU gives each of 6)
					size_t lds = (lds_cu / (size_t)w) & ~(size_t)255;
					if(lds > 65536) lds = 65536;
					float ms_best = 1e30f, ms_sum = 0;
					const int reps = pmc ? 1 : 3;
					if(!pmc) { launch_any(table, n_rec, c, blocks, lds, 1u, sink, st); HIP_OK(hipStreamSynchronize(st)); }
					for(int r = 0; r < reps; ++r)
					{
						HIP_OK(hipEventRecord(e0, st));
						launch_any(table, n_rec, c, blocks, lds, 77u + (uint32_t)r, sink, st);
						HIP_OK(hipEventRecord(e1, st));
						HIP_OK(hipEventSynchronize(e1));
						HIP_OK(hipGetLastError());
						float ms = 0;
						HIP_OK(hipEventElapsedTime(&ms, e0, e1));
						ms_best = ms < ms_best ? ms : ms_best; ms_sum += ms;
						++dispatch;
					}
					const double lanes = (double)blocks * 256.0;
					const double recs = lanes * ch * steps;
					const double s = ms_best * 1e-3;
					printf("{\"gather_launches_so_far\": %d, \"table_MiB\": %d, \"record_B\": %zu, \"n_records\": %u, \"wgs_per_cu\": %d, \"waves_per_cu\": %d, \"chains\": %d, \"steps\": %d, "
					       "\"ms\": %.4f, \"ms_mean\": %.4f, \"useful_GBs\": %.1f, \"Grecords_s\": %.3f, \"ns_per_dependent_read\": %.1f}\n",
					       dispatch, mib, rec, n_rec, w, w * 4, ch, steps, ms_best, ms_sum / reps, recs * rec / s * 1e-9, recs / s * 1e-9, s / steps * 1e9);
					fflush(stdout);
				}
			HIP_OK(hipFree(table));
		}
	}
	return 0;
}
