// Developer tool (GPU box): what a 16-byte-per-lane vector load costs on gfx950 as a function of how the 64 lane addresses fall
// on cache lines and how many lanes are active.  Every workgroup re-reads a small table (L1 resident after the first touch) with
// a fixed per-lane address pattern; all CUs are kept busy at 5 waves per SIMD like the traversal kernel.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l1_patterns tools/microbench/l1_patterns.hip && /tmp/l1_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kIters = 2048, kUnroll = 8;

__global__ __launch_bounds__(256, 5) void k_loads(const uint4 *table, const uint32_t *offsets /* [pattern rows][64] in 16-byte units */, int rows,
                                                  unsigned long long lane_mask, uint4 *sink, unsigned long long *cycles)
{
	const int lane = threadIdx.x & 63;
	const bool on = (lane_mask >> lane) & 1ull;
	uint32_t off[kUnroll];
	for(int k = 0; k < kUnroll; ++k) off[k] = offsets[((blockIdx.x * 4 + (threadIdx.x >> 6)) * kUnroll + k) % rows * 64 + lane];
	uint4 acc = make_uint4(0, 0, 0, 0);
	const unsigned long long t0 = __builtin_readcyclecounter();
	if(on)
		for(int it = 0; it < kIters; ++it)
		{
#pragma unroll
			for(int k = 0; k < kUnroll; ++k)
			{
				const uint4 v = table[off[k]];
				acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
				off[k] += v.x; // the table holds x = 0: the address stays put, but only the hardware knows (no hoisting of the load)
			}
		}
	const unsigned long long t1 = __builtin_readcyclecounter();
	if(acc.x == 0x12345678u && acc.y == 42u) sink[0] = acc;
	if(threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main()
{
	const int table_chunks = 16384 / 16; // 16 KB table: L1 resident
	std::vector<uint4> h_table(table_chunks, make_uint4(0, 1, 2, 3)); // x = 0, see the kernel
	uint4 *d_table, *d_sink; uint32_t *d_off; unsigned long long *d_cyc;
	hipMalloc(&d_table, h_table.size() * 16); hipMemcpy(d_table, h_table.data(), h_table.size() * 16, hipMemcpyHostToDevice);
	hipMalloc(&d_sink, 64); hipMalloc(&d_cyc, 8 * 4096);
	const int rows = 64;
	hipMalloc(&d_off, rows * 64 * 4);
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int blocks = prop.multiProcessorCount * 5;
	struct Pattern { const char *name; int kind; unsigned long long mask; };
	const Pattern pats[] = {
		{"64 lanes, 64 distinct 128-B lines (random 16 B in each)", 0, ~0ull},
		{"64 lanes, 8 lines x 8 consecutive 16-B chunks", 1, ~0ull},
		{"64 lanes, 1 KB contiguous", 2, ~0ull},
		{"64 lanes, 12 records of 80 B (5 lanes each) + 4 lanes idle", 3, 0x0fffffffffffffffull},
		{"64 lanes, 21 records of 48 B (3 lanes each)", 4, 0x7fffffffffffffffull},
		{"16 lanes (every 4th), distinct lines", 0, 0x1111111111111111ull},
		{"16 lanes (0..15), distinct lines", 0, 0xffffull},
		{"48 lanes (random 3 of 4), distinct lines", 0, 0x7777777777777777ull},
		{"64 lanes, 64 distinct 64-B halves, same 16-B slot", 5, ~0ull},
	};
	srand(1);
	for(const Pattern &p : pats)
	{
		std::vector<uint32_t> off(rows * 64);
		for(int r = 0; r < rows; ++r)
			for(int l = 0; l < 64; ++l)
			{
				uint32_t o = 0;
				const int lines = table_chunks / 8;
				switch(p.kind)
				{
				case 0: o = (uint32_t)((rand() % lines) * 8 + rand() % 8); break;
				case 1: { static uint32_t base[8]; if(l % 8 == 0) base[l / 8] = (uint32_t)(rand() % lines) * 8; o = base[l / 8] + l % 8; break; }
				case 2: { static uint32_t b; if(l == 0) b = (uint32_t)(rand() % (lines - 8)) * 8; o = b + l; break; }
				case 3: { static uint32_t b[13]; if(l % 5 == 0) b[l / 5] = (uint32_t)(rand() % (table_chunks / 5 - 1)) * 5; o = b[l / 5] + l % 5; break; }
				case 4: { static uint32_t b[22]; if(l % 3 == 0) b[l / 3] = (uint32_t)(rand() % (table_chunks / 3 - 1)) * 3; o = b[l / 3] + l % 3; break; }
				case 5: o = (uint32_t)((rand() % (table_chunks / 4)) * 4); break;
				}
				off[r * 64 + l] = o % table_chunks;
			}
		hipMemcpy(d_off, off.data(), off.size() * 4, hipMemcpyHostToDevice);
		hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
		hipLaunchKernelGGL(k_loads, dim3(blocks), dim3(256), 0, 0, d_table, d_off, rows, p.mask, d_sink, d_cyc);
		hipEventRecord(a);
		hipLaunchKernelGGL(k_loads, dim3(blocks), dim3(256), 0, 0, d_table, d_off, rows, p.mask, d_sink, d_cyc);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms = 0; hipEventElapsedTime(&ms, a, b);
		const double instr_per_cu = 20.0 * kIters * kUnroll; // 20 waves per CU
		printf("%-62s %7.3f ms  %6.1f ns per wave-load per CU  (~%5.1f cycles at 2.4 GHz)\n", p.name, ms, ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
	}
	return 0;
}
