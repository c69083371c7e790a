// Developer tool (GPU box): cost of one vector load on the CU's vector-memory pipeline by WIDTH (4 / 8 / 12 / 16 bytes per lane) and address
// pattern, L1-resident table, 5 waves per SIMD on every CU (companion of l1_patterns.hip, which only measured 16-byte loads).
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/l1_patterns2.bin tools/microbench/l1_patterns2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kIters = 2048, kUnroll = 8;
struct __attribute__((packed, aligned(4))) W3 { uint32_t x, y, z; };
template <int W> struct Word;
template <> struct Word<1> { typedef uint32_t T; };
template <> struct Word<2> { typedef uint2 T; };
template <> struct Word<3> { typedef W3 T; };
template <> struct Word<4> { typedef uint4 T; };

template <int W>
__global__ __launch_bounds__(256, 5) void k_loads(const uint32_t *table, const uint32_t *offsets /* [rows][64], dword units */, int rows, uint32_t *sink)
{
	typedef typename Word<W>::T T;
	const int lane = threadIdx.x & 63;
	uint32_t off[kUnroll];
	for(int k = 0; k < kUnroll; ++k) off[k] = offsets[((blockIdx.x * 4 + (threadIdx.x >> 6)) * kUnroll + k) % rows * 64 + lane];
	uint32_t acc = 0;
	for(int it = 0; it < kIters; ++it)
	{
#pragma unroll
		for(int k = 0; k < kUnroll; ++k)
		{
			const T v = *(const T *)(table + off[k]);
			const uint32_t first = *(const uint32_t *)&v;
			acc += first;
			if(W > 1) acc ^= ((const uint32_t *)&v)[W - 1];
			off[k] += first; // the table holds zeros: the address stays put, but only the hardware knows
		}
	}
	if(acc == 0x12345678u) sink[0] = acc;
}

template <int W> static void run(const char *name, int kind, const uint32_t *d_table, uint32_t *d_off, uint32_t *d_sink, int blocks, int table_dwords)
{
	const int rows = 64;
	std::vector<uint32_t> off(rows * 64);
	const int lines = table_dwords / 32;
	for(int r = 0; r < rows; ++r)
	{
		uint32_t few[8]; for(uint32_t &f : few) f = (uint32_t)(rand() % lines);
		const uint32_t start = (uint32_t)(rand() % (lines - 8)) * 32, same = (uint32_t)(rand() % lines) * 32;
		for(int l = 0; l < 64; ++l)
		{
			uint32_t o = 0;
			const int slots = 32 / (W == 3 ? 4 : W); // aligned slots of a 128-byte line (12-byte records: on 16-byte slots)
			const int step = W == 3 ? 4 : W;
			switch(kind)
			{
			case 0: o = (uint32_t)(rand() % lines) * 32 + (uint32_t)(rand() % slots) * step; break;  // every lane its own line
			case 1: o = start + (uint32_t)l * W; break;                                              // contiguous records
			case 2: o = few[rand() % 8] * 32 + (uint32_t)(rand() % slots) * step; break;             // 8 lines, lanes scattered over them
			case 3: o = same; break;                                                                 // one address for all lanes
			}
			off[r * 64 + l] = o;
		}
	}
	hipMemcpy(d_off, off.data(), off.size() * 4, hipMemcpyHostToDevice);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipLaunchKernelGGL(k_loads<W>, dim3(blocks), dim3(256), 0, 0, d_table, d_off, rows, d_sink);
	hipEventRecord(a);
	hipLaunchKernelGGL(k_loads<W>, dim3(blocks), dim3(256), 0, 0, d_table, d_off, rows, d_sink);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms = 0; hipEventElapsedTime(&ms, a, b);
	const double instr_per_cu = 20.0 * kIters * kUnroll;
	printf("%2d B per lane, %-44s %7.3f ms  %6.2f ns per wave-load per CU\n", W * 4, name, ms, ms * 1e6 / instr_per_cu);
}

int main()
{
	const int table_dwords = 16384 / 4 + 64; // 16 KB table: L1 resident
	uint32_t *d_table, *d_off, *d_sink;
	hipMalloc(&d_table, table_dwords * 4); hipMemset(d_table, 0, table_dwords * 4);
	hipMalloc(&d_off, 64 * 64 * 4); hipMalloc(&d_sink, 64);
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int blocks = prop.multiProcessorCount * 5;
	srand(1);
	const char *names[] = {"64 distinct lines", "contiguous records", "8 lines, lanes scattered", "one address"};
	for(int kind = 0; kind < 4; ++kind)
	{
		run<1>(names[kind], kind, d_table, d_off, d_sink, blocks, table_dwords - 64);
		run<2>(names[kind], kind, d_table, d_off, d_sink, blocks, table_dwords - 64);
		run<3>(names[kind], kind, d_table, d_off, d_sink, blocks, table_dwords - 64);
		run<4>(names[kind], kind, d_table, d_off, d_sink, blocks, table_dwords - 64);
	}
	return 0;
}
