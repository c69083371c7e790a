// Round-4 attempt at the root cause of round 3's append_slot fault ("The append_slot fault"): the FAILING form of the function — the
// wave vote taken before the two workgroup barriers, kept in an SGPR pair across them, rank = popcount(mask & ((1 << lane) - 1)) — inside a kernel
// shaped like the k_gen_primary that failed: 16 VGPRs, 256 threads, workgroup b appends to queue segment b & 7, camera-ray arithmetic (divisions,
// rsqrt) and two record stores after the claim; launched the way the product launched it when the fault showed: several chains on separate
// non-blocking streams, each alternating the claiming kernel with a long, register-heavy persistent kernel (the traversal's stand-in) so that
// the queues are oversubscribed and waves of the small kernel sit at their barriers while others are dispatched, and with the segment counters
// cleared by a small kernel between launches.  Every launch poisons nothing and checks everything: claims[slot] must be exactly 1 for every slot
// below the segment's counter, 0 above.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/append_slot_repro.hip -o /tmp/append_slot_repro && /tmp/append_slot_repro [launches per stream]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

constexpr int kSeg = 8, kStride = 32;

struct Cam { float m[12]; float origin[3]; int width, height; };

// FORM 0: round 3's failing form (mask across the barriers).  FORM 1: the product's form since (vote again after the barriers, v_mbcnt).
template <int FORM> __device__ __forceinline__ uint32_t append_slot(bool alive, uint32_t *seg_counter, uint32_t seg_base)
{
	__shared__ uint32_t wave_base[4];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const unsigned long long mask = __ballot(alive);
	if(lane == 0) wave_base[wave] = (uint32_t)__popcll(mask);
	__syncthreads();
	if(threadIdx.x == 0)
	{
		uint32_t c[4], total = 0;
#pragma unroll
		for(int w = 0; w < 4; ++w) { c[w] = wave_base[w]; total += c[w]; }
		uint32_t base = total ? atomicAdd(seg_counter, total) : 0u;
#pragma unroll
		for(int w = 0; w < 4; ++w) { wave_base[w] = base; base += c[w]; }
	}
	__syncthreads();
	if(FORM == 0) return seg_base + wave_base[wave] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
	const unsigned long long again = __ballot(alive);
	return seg_base + wave_base[wave] + __builtin_amdgcn_mbcnt_hi((uint32_t)(again >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)again, 0u));
}

template <int FORM> __global__ __attribute__((amdgpu_num_vgpr(16))) __launch_bounds__(256) void k_gen(Cam cam, uint32_t *counters, uint32_t seg_cap, uint32_t seg_paths,
                                                                                                      float *out_o, float4 *out_d, uint32_t *claims)
{
	const uint32_t seg = blockIdx.x & (kSeg - 1), chunk = blockIdx.x >> 3;
	const uint32_t local = chunk * 256 + threadIdx.x;
	const uint32_t pi = (chunk * kSeg + seg) * 256 + threadIdx.x;
	const int x = (int)(pi % (uint32_t)cam.width), y = (int)(pi / (uint32_t)cam.width);
	const bool alive = local < seg_paths && y < cam.height && ((x >> 3) % 5) != 0; // dead 8-pixel groups like the image border of the product
	const uint32_t slot = append_slot<FORM>(alive, counters + seg * kStride, seg * seg_cap);
	if(!alive) return;
	const float sx = (2.0f * (float)x) / (float)cam.width - 1.0f, sy = 1.0f - (2.0f * (float)y) / (float)cam.height;
	float dx = fmaf(cam.m[2], 1.0f, fmaf(cam.m[1], sy, cam.m[0] * sx)), dy = fmaf(cam.m[5], 1.0f, fmaf(cam.m[4], sy, cam.m[3] * sx)), dz = fmaf(cam.m[8], 1.0f, fmaf(cam.m[7], sy, cam.m[6] * sx));
	const float inv = 1.0f / sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
	dx *= inv; dy *= inv; dz *= inv;
	out_d[slot] = make_float4(dx, dy, dz, __uint_as_float(pi));
	out_o[slot] = cam.origin[0];
	atomicAdd(&claims[slot], 1u);
}

__global__ void k_clear(uint32_t *p, uint32_t n) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if(i < n) p[i] = 0u; }

// the traversal's stand-in: persistent, 4 waves per workgroup, many registers, dependent loads + arithmetic for `iters` rounds
__global__ __launch_bounds__(256, 5) void k_busy(const float4 *data, uint32_t n, float4 *sink, int iters)
{
	float4 acc[12];
	uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
#pragma unroll
	for(int k = 0; k < 12; ++k) acc[k] = make_float4(0, 0, 0, 0);
	for(int it = 0; it < iters; ++it)
	{
#pragma unroll
		for(int k = 0; k < 12; ++k)
		{
			const float4 v = data[(idx + (uint32_t)k * 7919u) % n];
			acc[k].x = fmaf(v.x, acc[k].y + 1.0f, acc[k].x); acc[k].y = fmaf(v.y, 0.5f, acc[k].z); acc[k].z = v.z * acc[k].w + v.w; acc[k].w = fminf(v.w, acc[k].x);
			idx = idx * 1664525u + 1013904223u + __float_as_uint(acc[k].x);
		}
	}
	float4 r = make_float4(0, 0, 0, 0);
#pragma unroll
	for(int k = 0; k < 12; ++k) { r.x += acc[k].x; r.y += acc[k].y; r.z += acc[k].z; r.w += acc[k].w; }
	sink[blockIdx.x * 256 + threadIdx.x] = r;
}

// per stream: counters of the 8 segments, claims per slot, a verdict per launch
__global__ void k_check(const uint32_t *counters, const uint32_t *claims, uint32_t seg_cap, unsigned long long *bad /* [0] double claims [1] missing [2] claimed above the counter */)
{
	const uint32_t seg = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= seg_cap) return;
	const uint32_t n = counters[seg * kStride], c = claims[seg * seg_cap + i];
	if(i < n) { if(c > 1u) atomicAdd(&bad[0], 1ull); if(c == 0u) atomicAdd(&bad[1], 1ull); }
	else if(c != 0u) atomicAdd(&bad[2], 1ull);
}

template <int FORM> void run(const char *name, int launches, int n_streams, bool with_busy)
{
	Cam cam;
	const float m[12] = {0.9f, 0.01f, -0.2f, 0.02f, 0.55f, 0.1f, 0.3f, -0.1f, -1.0f, 0, 0, 0};
	for(int i = 0; i < 12; ++i) cam.m[i] = m[i];
	cam.origin[0] = -8; cam.origin[1] = 2; cam.origin[2] = 0; cam.width = 1920; cam.height = 1080;
	const uint32_t paths = 1920u * 1080u, chunks = (paths + 255) / 256, seg_chunks = (chunks + kSeg - 1) / kSeg, seg_cap = seg_chunks * 256, seg_paths = seg_cap;
	const uint32_t grid = kSeg * seg_chunks;
	struct S { hipStream_t s; uint32_t *counters, *claims; float *o; float4 *d; unsigned long long *bad; float4 *busy_sink; };
	std::vector<S> st((size_t)n_streams);
	float4 *data; const uint32_t n_data = 1u << 20;
	CK(hipMalloc(&data, (size_t)n_data * 16)); CK(hipMemset(data, 0x3c, (size_t)n_data * 16));
	for(S &q : st)
	{
		CK(hipStreamCreateWithFlags(&q.s, hipStreamNonBlocking));
		CK(hipMalloc(&q.counters, kSeg * kStride * 4)); CK(hipMalloc(&q.claims, (size_t)kSeg * seg_cap * 4));
		CK(hipMalloc(&q.o, (size_t)kSeg * seg_cap * 12)); CK(hipMalloc(&q.d, (size_t)kSeg * seg_cap * 16)); CK(hipMalloc(&q.bad, 24)); CK(hipMemset(q.bad, 0, 24));
		CK(hipMalloc(&q.busy_sink, (size_t)256 * 5 * 256 * 16));
	}
	CK(hipDeviceSynchronize());
	for(int l = 0; l < launches; ++l)
		for(S &q : st) // launch by launch over the streams: the chains reach the GPU interleaved, as the product's pipes did
		{
			hipLaunchKernelGGL(k_clear, dim3(1), dim3(256), 0, q.s, q.counters, (uint32_t)(kSeg * kStride));
			hipLaunchKernelGGL(k_clear, dim3((kSeg * seg_cap + 255) / 256), dim3(256), 0, q.s, q.claims, kSeg * seg_cap);
			hipLaunchKernelGGL(k_gen<FORM>, dim3(grid), dim3(256), 0, q.s, cam, q.counters, seg_cap, seg_paths, q.o, q.d, q.claims);
			hipLaunchKernelGGL(k_check, dim3((seg_cap + 255) / 256, kSeg), dim3(256), 0, q.s, q.counters, q.claims, seg_cap, q.bad);
			if(with_busy) hipLaunchKernelGGL(k_busy, dim3(256 * 5), dim3(256), 0, q.s, data, n_data, q.busy_sink, 40);
		}
	CK(hipDeviceSynchronize());
	unsigned long long tot[3] = {0, 0, 0};
	for(S &q : st) { unsigned long long b[3]; CK(hipMemcpy(b, q.bad, 24, hipMemcpyDeviceToHost)); for(int i = 0; i < 3; ++i) tot[i] += b[i]; }
	printf("%-46s %d streams x %d launches of %u workgroups%s: slots claimed twice %llu, never claimed below the counter %llu, claimed above the counter %llu\n", name, n_streams,
	       launches, grid, with_busy ? " (+ persistent stand-in between)" : "", tot[0], tot[1], tot[2]);
	for(S &q : st) { CK(hipStreamDestroy(q.s)); CK(hipFree(q.counters)); CK(hipFree(q.claims)); CK(hipFree(q.o)); CK(hipFree(q.d)); CK(hipFree(q.bad)); CK(hipFree(q.busy_sink)); }
	CK(hipFree(data));
}

int main(int argc, char **argv)
{
	const int launches = argc > 1 ? atoi(argv[1]) : 600;
	run<0>("mask across the barriers (round 3's failing form)", launches, 1, false);
	run<0>("mask across the barriers (round 3's failing form)", launches, 4, true);
	run<0>("mask across the barriers (round 3's failing form)", launches, 8, true);
	run<1>("vote again after the barriers (the product)", launches, 8, true);
	return 0;
}
