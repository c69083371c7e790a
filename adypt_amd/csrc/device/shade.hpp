// gfx950 kernels of the wavefront path tracer (everything except the traversal kernels, which live in traverse.hpp and path.hpp).
//
//   k_gen_primary       camera rays + path-state init, into the queue   (shaders/pathtracer.glsl:206-224, primaryray.glsl:39-49; the launch-per-bounce
//                                                                        pipeline: k_trace_camera computes camera rays itself)
//   k_shade_first       camera ray + bounce 0 of every frame of a batch from the cached primary hit
//   k_shade             one bounce of Render(): FetchInfo, scatter, compaction, accumulate on termination
//                                                                       (shaders/pathtracer.glsl:73-204,224-226)
//   viewer_color        primary-ray viewer colouring, called by k_trace_camera when a ray has finished (shaders/primaryray.glsl:50-94)
//   k_resolve, k_display, k_untile   running mean in frame order; the window's display transform; compact block-major radiance -> W x H RGB
//
// Ray queue = 8 XCD-affine segments.  Workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MB
// L2), so workgroup b of every kernel works on segment b & 7: rays stay with "their" XCD from bounce to bounce, the
// queue tail of each segment is its own counter on its own 128-byte line (a single tail word saturates at ~88
// atomics/us on this chip), and one atomic per workgroup appends the survivors.  Segment s owns the slot range
// [s * seg_cap, (s+1) * seg_cap).  None of this affects results — per-pixel work is independent of slot order.
//
// Queue layout (SoA, slot-indexed, rewritten compacted every bounce) — 52 B per path and bounce (round 2: 64 B):
//   ray_o = origin.xyz (12 B; tmin is the pass's)   ray_d = (dir.xyz, path word) (16 B)   <- the 28 B the traversal reads
//   col   = throughput rgb (12 B)                                                          <- carried by the shade kernel only
//   hit   = (bits(scene triangle id), u, v) (12 B)                                         <- what the traversal writes
//   path word: bits 30..0 path id (batch frame * local pixels + local pixel), bit 31 "radiance parked in done[path id]"
// (adypt_trace_rays batches and the sun-visibility queue keep float4 records: per-ray tmin in ray_o.w, t in hit.w)
// Per-local-pixel buffers (block-major, 32x32 blocks of 16 8x8 wave tiles): accum (RGBA32F running mean),
// cache (primary hit: bits(tri), u, v, -), shift (2 bytes).
// Scene triangles are repacked at upload from the 100-byte Triangle (src/Util/Shape.hpp:70-74) to 128 bytes =
// 8 x float4: [p0 p1 p2 n0 n1 n2 | matid | class word] (5 x 16 B, always read; class word: 1 = glossy lobe or dielectric, what k_path's shading rounds defer) + [tc0 tc1 tc2 | pad] (2 x 16 B, textured only) + 16 B pad:
// one 128-byte line per gather (at 112 bytes a record straddled two lines three times out of four).
#pragma once
#include "canon_math.hpp"

namespace adypt {

constexpr int kBlockShift = 5;                 // 32x32 pixel shard blocks
constexpr int kBlockDim = 1 << kBlockShift;
constexpr int kBlockPixels = kBlockDim * kBlockDim;
constexpr int kTraceThreads = 256;             // 4 waves per workgroup
constexpr int kLdsStackMax = 8;                // stack entries kept in LDS per lane; deeper entries spill to HBM
constexpr int kNumSegments = 8;                // one ray-queue segment per XCD
constexpr int kCursorStride = 32;              // uint32 words between per-segment counters: one 128-byte line each
constexpr int kShadeThreads = 256;             // workgroup of the gen / shade / viewer kernels = one queue chunk
constexpr int kTriFloat4 = 8;                  // device triangle record: 8 x float4 = one 128-byte line
constexpr int kMatFloat4 = 5;                  // device material record: the reference's 64 bytes + (texel offset, w, h, 0) of its diffuse texture
constexpr uint32_t kPathParked = 0x80000000u;  // path word: the path's radiance so far is parked in FrameArgs::done[path id]
constexpr uint32_t kPathIdMask = 0x7fffffffu;
constexpr uint32_t kPathShadow = 0x40000000u;  // path word of the queue k_path reads (ids < 2^26 there): the ray is a path's sun-visibility query (pathtracer.glsl:132), not a bounce

struct DeviceStats {                           // accumulated until adypt_reset_stats
	unsigned long long rays, nodes, tris, hits, shaded, overflows, bad_materials;
	uint32_t max_stack, pad;
	unsigned long long wave_profile[8];          // adypt_get_wave_profile (instrumented traversal only)
	unsigned long long clock_cycles, clock_ticks; // adypt_get_shader_clock: shader cycles / 100 MHz ticks of workgroup 0 over the traversal launches
	unsigned long long path_rays, path_nodes, path_tris, path_hits, path_shaded; // k_path's share of rays / nodes / tris / hits / shaded
	unsigned long long audit_errors;             // slot-claim audit (adypt_set_instrumentation flag 4): slots of an appended queue not written exactly once
	uint32_t *host_overflow;                     // NOT a counter (set at adypt_create, kept by adypt_reset_stats): pinned host word a traversal kernel sets when a
	                                             //   stack overflows, so that the host learns of it from the stream synchronisation alone (no copy per call)
};
// a traversal kernel's report of a stack overflow (traversal.glsl has none: its stack is a fixed local array)
__device__ __forceinline__ void report_overflow(DeviceStats *st) { atomicAdd(&st->overflows, 1ull); *st->host_overflow = 1u; }

struct RayStats { int32_t ref_idx; uint32_t nodes, tris, hash, max_depth, pad0, pad1, pad2; }; // 32 B, STATS variant

struct TraceArgs {
	const uint4 *nodes;
	const float4 *woop;
	const int32_t *tri_indices;
	const float4 *ray_o, *ray_d;   // packed: ray_o is 3 floats per slot and tmin is the pass's; otherwise (origin, tmin) per slot
	float4 *hit;                   // packed: 3 floats per slot (bits(tri), u, v); otherwise (bits(tri), u, v, t)
	RayStats *ray_stats;           // STATS only (may be null)
	const uint32_t *count;         // rays per segment: count[s * kCursorStride] (device memory)
	uint32_t *cursor;              // fetch cursor per segment: cursor[s * kCursorStride], zero at launch
	uint2 *spill;                  // [(stack_size - lds_depth)][total lanes]
	DeviceStats *stats;
	uint32_t seg_cap;              // slots per segment
	int32_t stack_size, lds_depth;
	uint32_t refill_min, chunk, bite, endgame; // tunables of the persistent fetch (traverse.hpp)
	uint32_t packed;               // the path tracer's own queues (12-byte origins and hits) / float4 records
	float tmin;                    // packed: tmin of every ray of the pass
};

// ---------------------------------------------------------------------------------------------------------------
// frame-level parameters (uuCamera + uuPT UBOs of the reference, plus shard geometry)
// ---------------------------------------------------------------------------------------------------------------
struct FrameArgs {
	float inv_proj[16], inv_view[16];
	float origin[3], tmin;
	float sun[3], clamp;
	const float *sobol;            // device: [frame in batch][64] — each frame's Sobol point, 2*max_bounce floats used (pathtracer.glsl:46,49)
	float4 *done;                  // device: [frame in batch][local pixel] finished sample radiance (batches of > 1 frame only)
	int32_t width, height;
	int32_t spp, subpixel, tmp_life, max_bounce; // spp = index of the FIRST frame of the batch
	float sun_query_dir[3];        // sun_query: the (normalised) direction of the occlusion query the reference has commented out (pathtracer.glsl:132)
	int32_t sun_query;             // 1: the one-launch pipeline traces that query for every escaped path (k_shade_first emits it for bounce 0, k_path for the others)
	int32_t n_frames;              // frames of this pass (queue position = frame ordinal * n_local_px + local pixel)
	int32_t frame_first, frame_stride; // batch frame of ordinal r = frame_first + r * frame_stride (a sub-batch of the main pass:
	                               //   its first frame, 1; the primary-only pass runs just the re-tracing frames: first one, tmp_life)
	int32_t batched;               // the pass belongs to a batch of several frames: path id = batch frame * n_local_px + local pixel,
	                               //   finished samples are parked in done[path id] and applied in frame order by k_resolve
	int32_t n_local_px;            // owned blocks * 1024
	int32_t blocks_x;              // image width in 32-px blocks
	int32_t rank, nranks;
	int32_t n_tris, n_mats, n_tex;
	int32_t deal_chunks;           // k_gen_primary: queue chunks of 256 paths are dealt round-robin to the 8 segments instead of one contiguous run each
};

struct SceneArgs {
	const float4 *triangles;       // kTriFloat4 float4 per triangle (repacked, see header)
	const float4 *materials;       // kMatFloat4 x float4 per material
	const uint32_t *texels;        // RGBA8, all textures back to back; every row is followed by a copy of its first texel (w + 1 per row)
	const int32_t *local_blocks;   // global block id of each owned block
	const uint8_t *tri_class;      // per triangle: shading class of its material, 1..6 (material_class); null = k_shade does not bin
};

// Shading class of a material = which code of Render() a path that hits it runs (pathtracer.glsl:144-201) — only a SORT KEY for
// k_shade's in-workgroup binning, never an input of the arithmetic.  0 is the key of a miss, 7 of a thread without a path.
constexpr uint32_t kClassMiss = 0, kClassNone = 7;
inline uint32_t material_class(int illum, float shininess, bool textured)
{
	if(illum == 2 && shininess * 0.01f > 0.3f) return textured ? 4u : 3u; // glossy lobe: two canon_pow + sincos
	if(illum == 1 || illum == 2) return textured ? 2u : 1u;                // diffuse lobe
	if(illum == 6 || illum == 7) return 6u;                                // dielectric
	return 5u;                                                             // mirror (3..5) and pass-through (0, 8+)
}

struct QueueArgs {
	float *ray_o; float4 *ray_d; float *col; // queue being read (shade) / written (gen): 3 floats, float4, 3 floats per slot
	float *hit;                    // 3 floats per slot
	float *out_o; float4 *out_d; float *out_col;
	const uint32_t *count_in;      // [s * kCursorStride]
	uint32_t *count_out;           // [s * kCursorStride]
	uint32_t seg_cap;              // allocated slots per segment: segment s owns [s * seg_cap, (s + 1) * seg_cap)
	uint32_t seg_paths;            // paths per segment of THIS pass (<= seg_cap, multiple of kShadeThreads): a pass of fewer frames
	                               //   than fit still spreads evenly over the 8 segments, and the grids cover seg_paths only
};

// Optional sun-visibility test = the occlusion query the reference has commented out (pathtracer.glsl:132): escaped
// paths go to this queue instead of receiving the sun term at once; an any-hit traversal and k_shadow_resolve follow.
struct ShadowArgs {
	float4 *o, *d, *col;           // (origin, tmin), (normalised sun direction, bits(path)), (throughput, radiance parked?)
	float4 *hit;                   // any-hit result per slot
	uint32_t *count;               // [s * kCursorStride] escaped paths per segment of this bounce
	float dir[3];
	int32_t enabled;
};

struct PixelArgs {
	float4 *accum;                 // running mean RGBA per local pixel   (image 0)
	float4 *cache;                 // cached primary hit per local pixel  (image 1)
	float4 *cache_next;            // batches spanning several tmpLifetime groups: primary hits of group 1, 2, ... of the
	                               //   batch, [group - 1][local pixel]; group 0 (the batch's first frame) lives in `cache`
	const uint8_t *shift;          // 2 bytes per local pixel             (image 2)
	DeviceStats *stats;
};

// Keeps VGPRs allocated beyond the kernel's highest register (the clobber makes `reg`, the first one of the next granule of 8,
// count as used).  Precaution
// that goes with the append_slot note below: the misbehaving build of k_gen_primary had exactly 16 VGPRs, the same instructions with
// 24 allocated did not misbehave.  tools/check_vgpr.py lists every kernel's count and fails when one of these small kernels lands on a multiple of 8.
#define ADYPT_VGPR_SLACK(reg) asm volatile("; one spare VGPR granule" ::: reg)

__device__ __forceinline__ bool local_pixel_xy(const FrameArgs &f, const int32_t *local_blocks, int L, int *x, int *y)
{
	const int blk = local_blocks[L >> 10];
	const int in = L & 1023, wt = in >> 6, ln = in & 63;
	const int bx = blk % f.blocks_x, by = blk / f.blocks_x;
	*x = bx * kBlockDim + (wt & 3) * 8 + (ln & 7);
	*y = by * kBlockDim + (wt >> 2) * 8 + (ln >> 3);
	return *x < f.width && *y < f.height;
}

// 12-byte records (one global_load_dwordx3 / global_store_dwordx3 each)
struct __attribute__((packed, aligned(4))) Rec3 { float x, y, z; };
__device__ __forceinline__ F3 ld3(const float *base, size_t i) { const Rec3 r = *(const Rec3 *)(base + 3 * i); return f3(r.x, r.y, r.z); }
__device__ __forceinline__ void st3(float *base, size_t i, float x, float y, float z) { Rec3 r; r.x = x; r.y = y; r.z = z; *(Rec3 *)(base + 3 * i) = r; }

// tmpLifetime group of batch frame `frame`, relative to the group of the batch's first frame (f.spp)
__device__ __forceinline__ int frame_group(const FrameArgs &f, int frame) { return (f.spp + frame) / f.tmp_life - f.spp / f.tmp_life; }
__device__ __forceinline__ float4 *cache_of_group(const FrameArgs &f, const PixelArgs &px, int group)
{
	return group == 0 ? px.cache : px.cache_next + (size_t)(group - 1) * (size_t)f.n_local_px;
}

__device__ __forceinline__ F3 camera_dir(const FrameArgs &f, int px, int py, float bx, float by)
{
	float sx = (2.0f * ((float)px + bx)) / (float)f.width - 1.0f;
	float sy = (2.0f * ((float)py + by)) / (float)f.height - 1.0f;
	sy = -sy;
	const float *m = f.inv_proj;
	F3 t;
	t.x = fmaf(m[12], 1.0f, fmaf(m[8], 1.0f, fmaf(m[4], sy, m[0] * sx)));
	t.y = fmaf(m[13], 1.0f, fmaf(m[9], 1.0f, fmaf(m[5], sy, m[1] * sx)));
	t.z = fmaf(m[14], 1.0f, fmaf(m[10], 1.0f, fmaf(m[6], sy, m[2] * sx)));
	const float *v = f.inv_view;
	F3 r;
	r.x = fmaf(v[8], t.z, fmaf(v[4], t.y, v[0] * t.x));
	r.y = fmaf(v[9], t.z, fmaf(v[5], t.y, v[1] * t.x));
	r.z = fmaf(v[10], t.z, fmaf(v[6], t.y, v[2] * t.x));
	return normalize3(r);
}

// Workgroup-level stream compaction into the workgroup's queue segment: returns this thread's output slot (valid
// only if `alive`).  Wave vote (__ballot) -> scan of the 4 wave counts -> ONE device atomic per workgroup -> rank of the lane
// among its wave's survivors (v_mbcnt).
// The vote is taken AGAIN after the barriers for the rank instead of keeping the first mask alive across them.  Round 3: with the
// mask held in an SGPR pair over the two s_barriers and the rank computed as popcount(mask & ((1 << lane) - 1)), whole waves of
// k_gen_primary occasionally (a few workgroups per launch, depending on timing) saw rank 0 in every lane and wrote all their paths
// to the wave's first slot — in a build whose ISA reads correctly and only when the kernel's VGPR count came out as exactly 16; any
// perturbation of the code hid it (tools/debug notes in DESIGN.md §10).  The second vote costs one v_cmp and is immune to whatever
// that was; v_mbcnt is also two instructions instead of the shift / select / popcount five.
__device__ __forceinline__ uint32_t append_slot(bool alive, uint32_t *seg_counter, uint32_t seg_base)
{
	__shared__ uint32_t wave_base[kShadeThreads / 64];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	{
		const unsigned long long mask = __ballot(alive);
		if(lane == 0) wave_base[wave] = (uint32_t)__popcll(mask);
	}
	__syncthreads();
	if(threadIdx.x == 0)
	{
		uint32_t c[kShadeThreads / 64], total = 0;
#pragma unroll
		for(int w = 0; w < kShadeThreads / 64; ++w) { c[w] = wave_base[w]; total += c[w]; }
		uint32_t base = total ? atomicAdd(seg_counter, total) : 0u;
#pragma unroll
		for(int w = 0; w < kShadeThreads / 64; ++w) { wave_base[w] = base; base += c[w]; }
	}
	__syncthreads();
	const unsigned long long mask = __ballot(alive);
	return seg_base + wave_base[wave] + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// In-workgroup counting sort of the workgroup's 256 paths by an 8-valued key: returns the index (0..255) of the thread whose path
// THIS thread takes over, such that keys ascend with threadIdx.x — waves then run one class of material code instead of all of them
// under exec masks.  Stable (ties keep thread order).  Per wave 8 votes; the 32 (class, wave) counts meet in LDS.
__device__ __forceinline__ uint32_t bin_by_key(uint32_t key)
{
	__shared__ uint32_t count[8][kShadeThreads / 64];
	__shared__ uint16_t source[kShadeThreads];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t rank = 0;
#pragma unroll
	for(uint32_t c = 0; c < 8; ++c)
	{
		const unsigned long long mask = __ballot(key == c);
		if(lane == 0) count[c][wave] = (uint32_t)__popcll(mask);
		if(key == c) rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
	}
	__syncthreads();
	uint32_t before = 0, base = 0; // paths of lower classes; + paths of this class in lower waves
#pragma unroll
	for(uint32_t c = 0; c < 8; ++c)
	{
		uint32_t lower_waves = 0, all = 0;
#pragma unroll
		for(int w = 0; w < kShadeThreads / 64; ++w) { const uint32_t n = count[c][w]; if(w < wave) lower_waves += n; all += n; }
		if(key == c) base = before + lower_waves;
		before += all;
	}
	source[base + rank] = (uint16_t)threadIdx.x;
	__syncthreads();
	return source[threadIdx.x];
}

// use_cache: the frame reuses the cached primary hit (spp % tmpLife != 0, pathtracer.glsl:115-120) — the hit
// record is copied next to the ray and the host skips the bounce-0 traversal launch.
// bias_mode 0: Camera() of primaryray.glsl (no sub-pixel bias); 1: Camera(SubPixel()) of pathtracer.glsl
__global__ __launch_bounds__(kShadeThreads) void k_gen_primary(FrameArgs f, SceneArgs sc, QueueArgs q, PixelArgs px, int use_cache, int bias_mode)
{
	ADYPT_VGPR_SLACK("v16");
	const uint32_t seg = blockIdx.x & (kNumSegments - 1), chunk = blockIdx.x >> 3;
	// Which paths a segment (= an XCD's share of the queue; a path stays in its segment for life) starts with.  Rounds 1-2: one contiguous run
	// of local pixels each, for locality.  Round 3: chunks of 256 paths (four 8x8 tiles) dealt round-robin — every segment then holds an even
	// sample of the image, so the eight XCDs finish together instead of the one with the deepest geometry last: primary rays only +4.5 %,
	// 10 M-triangle frames +1.8 %, k_shade -3.5 %, nothing slower (profiles/r3_ablations_k_trace.txt item 13)
	const uint32_t local = chunk * kShadeThreads + threadIdx.x;
	// path index = frame ordinal * n_local_px + local pixel.  A segment takes one contiguous run of paths, or (deal_chunks) every 8th chunk
	const uint32_t pi = f.deal_chunks ? (chunk * kNumSegments + seg) * kShadeThreads + threadIdx.x : seg * q.seg_paths + local;
	const int ordinal = (int)(pi / (uint32_t)f.n_local_px), L = (int)(pi % (uint32_t)f.n_local_px);
	const int frame = f.frame_first + ordinal * f.frame_stride;
	int x = 0, y = 0;
	const bool alive = local < q.seg_paths && ordinal < f.n_frames && local_pixel_xy(f, sc.local_blocks, L, &x, &y);
	float bx = 0.0f, by = 0.0f;
	if(bias_mode)
	{
		const int sub_idx = ((f.spp + frame) / f.tmp_life) % (f.subpixel * f.subpixel);
		const float unit = 1.0f / (float)f.subpixel;
		bx = (float)(sub_idx / f.subpixel) * unit;
		by = (float)(sub_idx % f.subpixel) * unit;
	}
	const uint32_t slot = append_slot(alive, q.count_out + seg * kCursorStride, seg * q.seg_cap);
	if(!alive) return;
	const F3 d = camera_dir(f, x, y, bx, by);
	st3(q.out_o, slot, f.origin[0], f.origin[1], f.origin[2]);
	q.out_d[slot] = make_float4(d.x, d.y, d.z, __int_as_float(frame * f.n_local_px + L)); // path word: id (batch frame, local pixel), nothing parked
	st3(q.out_col, slot, 1.0f, 1.0f, 1.0f);
	if(use_cache) { const float4 h = cache_of_group(f, px, frame_group(f, frame))[L]; st3(q.hit, slot, h.x, h.y, h.z); }
}

// Applies finished samples of a batch to the running mean in frame order (pathtracer.glsl:224-226): batch frames
// [first, first + count); f.spp = index of the batch's first frame.  The whole batch at once, or — when the caller asks for
// one frame per call and the batch was traced ahead (adypt_set_lookahead) — a few frames per call.
__global__ __launch_bounds__(256) void k_resolve(FrameArgs f, SceneArgs sc, PixelArgs px, int first, int count)
{
	const int L = blockIdx.x * blockDim.x + threadIdx.x;
	int x, y;
	if(L >= f.n_local_px || !local_pixel_xy(f, sc.local_blocks, L, &x, &y)) return;
	float4 acc = px.accum[L];
	for(int k = first; k < first + count; ++k)
	{
		const float4 r = f.done[(size_t)k * f.n_local_px + L];
		const float fs = (float)(f.spp + k), fs1 = (float)(f.spp + k + 1);
		acc = make_float4(fmaf(acc.x, fs, r.x) / fs1, fmaf(acc.y, fs, r.y) / fs1, fmaf(acc.z, fs, r.z) / fs1, 1.0f);
	}
	px.accum[L] = acc;
}

__device__ __forceinline__ int pos_mod(int a, int n) { int r = a % n; return r < 0 ? r + n : r; }

// GL_LINEAR / GL_REPEAT / RGB8, single level, canonical fp32 weights (see oracle.cpp sample_texture).  d = (texel offset, w, h) from the
// material record.  The two texels of a row come in one 8-byte load: rows are stored w + 1 texels long, the last one a copy of the
// first, so (i0, i0 + 1) is the wrapped pair too — 2 vector-memory instructions per sample instead of 4.
struct __attribute__((packed, aligned(4))) TexelPair { uint32_t a, b; };
__device__ inline F3 sample_texture(const SceneArgs &sc, int4 d, float s, float t)
{
	const int w = d.y, h = d.z;
	const uint32_t *tx = sc.texels + d.x;
	const float uu = fmaf(s, (float)w, -0.5f), vv = fmaf(t, (float)h, -0.5f);
	float fu = floorf(uu), fv = floorf(vv);
	const float a = uu - fu, b = vv - fv;
	fu = gl_min(gl_max(fu, -1e9f), 1e9f); fv = gl_min(gl_max(fv, -1e9f), 1e9f);
	const int i0 = pos_mod((int)fu, w), j0 = pos_mod((int)fv, h);
	const int j1 = j0 + 1 == h ? 0 : j0 + 1;
	const TexelPair r0 = *(const TexelPair *)(tx + (size_t)j0 * (size_t)(w + 1) + i0), r1 = *(const TexelPair *)(tx + (size_t)j1 * (size_t)(w + 1) + i0);
	auto rgb = [](uint32_t p) { return f3(unorm8_to_float(p), unorm8_to_float(p >> 8), unorm8_to_float(p >> 16)); };
	const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
	F3 r = rgb(r0.a) * w00;
	r = fma3(rgb(r0.b), w10, r);
	r = fma3(rgb(r1.a), w01, r);
	r = fma3(rgb(r1.b), w11, r);
	return r;
}

// the always-needed 80 bytes of a triangle: positions, normals, material id
struct TriCore { float v[20]; };
__device__ __forceinline__ TriCore load_tri_core(const SceneArgs &sc, int tri_idx)
{
	const float4 *p = sc.triangles + (size_t)tri_idx * kTriFloat4;
	const float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
	TriCore t;
	t.v[0] = a.x; t.v[1] = a.y; t.v[2] = a.z; t.v[3] = a.w; t.v[4] = b.x; t.v[5] = b.y; t.v[6] = b.z; t.v[7] = b.w;
	t.v[8] = c.x; t.v[9] = c.y; t.v[10] = c.z; t.v[11] = c.w; t.v[12] = d.x; t.v[13] = d.y; t.v[14] = d.z; t.v[15] = d.w;
	t.v[16] = e.x; t.v[17] = e.y; t.v[18] = e.z; t.v[19] = e.w;
	return t;
}
__device__ __forceinline__ F3 bary3(const float *a, const float *b, const float *c, float u, float v, float w)
{
	F3 r = f3(a[0], a[1], a[2]) * u;
	r = fma3(f3(b[0], b[1], b[2]), v, r);
	r = fma3(f3(c[0], c[1], c[2]), w, r);
	return r;
}
__device__ __forceinline__ F3 textured_diffuse(const SceneArgs &sc, int tri_idx, int4 desc, float u, float v, float w)
{
	const float4 *p = sc.triangles + (size_t)tri_idx * kTriFloat4 + 5;
	const float4 a = p[0], b = p[1]; // tc0.xy tc1.xy | tc2.xy pad
	const float ts = fmaf(b.x, w, fmaf(a.z, v, a.x * u));
	const float tt = fmaf(b.y, w, fmaf(a.w, v, a.y * u));
	return sample_texture(sc, desc, ts, tt);
}

struct Rng { float sx, sy; const float *sobol; };
__device__ __forceinline__ void sobol2(const Rng &r, int i, float *x, float *y)
{
	const float a = r.sobol[2 * i] + r.sx, b = r.sobol[2 * i + 1] + r.sy;
	*x = a - floorf(a); *y = b - floorf(b);
}
// the same with the frame's point of bounce `b` fetched by the caller (k_path issues that load with the round's other independent fetches)
struct RngPoint { float sx, sy, qx, qy; };
__device__ __forceinline__ void sobol2(const RngPoint &r, int, float *x, float *y)
{
	const float a = r.qx + r.sx, b = r.qy + r.sy;
	*x = a - floorf(a); *y = b - floorf(b);
}
template <class RNG>
__device__ inline F3 sample_hemisphere(const RNG &rng, int b, float e)
{
	float rx, ry; sobol2(rng, b, &rx, &ry);
	rx *= 6.28318530718f;
	float sin_phi, cos_phi; canon_sincos(rx, &sin_phi, &cos_phi);
	const float cos_theta = canon_pow(1.0f - ry, 1.0f / (e + 1.0f));
	const float sin_theta = sqrtf(fmaf(-cos_theta, cos_theta, 1.0f));
	return normalize3(f3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta));
}
__device__ inline F3 align_direction(F3 dir, F3 target)
{
	const F3 a = fabsf(target.x) > 0.01f ? f3(0, 1, 0) : f3(1, 0, 0);
	const F3 u = normalize3(cross3(a, target));
	const F3 v = cross3(target, u);
	F3 r = u * dir.x;
	r = fma3(v, dir.y, r);
	r = fma3(target, dir.z, r);
	return r;
}

// main() accumulate of pathtracer.glsl:224-226, executed once per pixel and frame when its path ends
__device__ __forceinline__ void finish_path(const FrameArgs &f, const PixelArgs &px, int pi, int L, F3 ret)
{
	const F3 r = f3(gl_min(ret.x, f.clamp), gl_min(ret.y, f.clamp), gl_min(ret.z, f.clamp));
	if(f.batched) { f.done[pi] = make_float4(r.x, r.y, r.z, 1.0f); return; }      // applied in frame order by k_resolve (the
	                                                                               // slot doubles as the parked radiance of a live path)
	const float4 old = px.accum[L];
	const float fs = (float)f.spp, fs1 = (float)(f.spp + 1);
	px.accum[L] = make_float4(fmaf(old.x, fs, r.x) / fs1, fmaf(old.y, fs, r.y) / fs1, fmaf(old.z, fs, r.z) / fs1, 1.0f);
}

// FetchInfo (pathtracer.glsl:73-100): what the surface at a hit is made of.  Depends on the hit alone (triangle, u, v) — not on the ray —
// so the frames of one tmpLifetime group, which share their primary hit, share it too (k_shade_first).
struct SurfaceInfo { F3 origin, normal, diffuse, specular, emission; int illum0; float shininess, ior; bool bad_mat; };
__device__ __forceinline__ SurfaceInfo fetch_info(const FrameArgs &f, const SceneArgs &sc, const TriCore &tc, int tri_idx, float tu, float tv)
{
	SurfaceInfo s;
	const float *tri = tc.v;
	const int matid = __float_as_int(tri[18]);
	// the hit's geometry before the material is looked at: all five loads of the record are then in flight together (with the
	// interpolation under the material test the compiler fetched the material id first and the rest one round trip later)
	const float w = 1.0f - tu - tv;
	s.normal = normalize3(bary3(tri + 9, tri + 12, tri + 15, tu, tv, w));
	s.origin = bary3(tri + 0, tri + 3, tri + 6, tu, tv, w);
	s.bad_mat = matid < 0 || matid >= f.n_mats;
	s.diffuse = s.specular = s.emission = f3(0, 0, 0); s.illum0 = 0; s.shininess = 0.0f; s.ior = 1.0f;
	if(!s.bad_mat)
	{
		const float4 *mp = sc.materials + (size_t)matid * kMatFloat4;
		const float4 md = mp[0], me = mp[1], ms = mp[2], mx = mp[3], mt = mp[4];
		const int dtex = __float_as_int(md.x);
		s.illum0 = __float_as_int(mx.x); s.shininess = mx.y; s.ior = mx.w;
		const int4 tex_desc = make_int4(__float_as_int(mt.x), __float_as_int(mt.y), __float_as_int(mt.z), 0);
		if(f.n_tex != 0 && dtex != -1 && dtex >= 0 && dtex < f.n_tex) s.diffuse = textured_diffuse(sc, tri_idx, tex_desc, tu, tv, w);
		else s.diffuse = f3(md.y, md.z, md.w);
		s.specular = f3(ms.y, ms.z, ms.w);
		s.emission = f3(me.y, me.z, me.w);
	}
	return s;
}

__device__ __forceinline__ SurfaceInfo fetch_info(const FrameArgs &f, const SceneArgs &sc, int tri_idx, float tu, float tv)
{
	return fetch_info(f, sc, load_tri_core(sc, tri_idx), tri_idx, tu, tv);
}

// The rest of one iteration `b` of Render()'s loop (pathtracer.glsl:101-104, 144-201) at a surface with a valid material: emission picked
// up, the `illum` switch, the new direction and throughput.  Returns whether the path goes on.
template <class RNG>
__device__ __forceinline__ bool respond(const FrameArgs &f, const SurfaceInfo &si, const RNG &rng, int b, F3 &dir, F3 &color, F3 &ret)
{
	bool alive = true;
	F3 normal = si.normal;
	const F3 diffuse = si.diffuse, specular = si.specular;
	const int illum0 = si.illum0;
	const float shininess = si.shininess, ior = si.ior;
	ret = fma3(color, si.emission, ret);
	// last iteration of the loop (pathtracer.glsl:107): the switch below only produces the next direction and throughput, which nothing reads any more
	if(b + 1 >= f.max_bounce) return false;
	if(illum0 < 6 && dot3(dir, normal) > 0) normal = -normal;
	int illum = illum0;
	bool done = false;
	if(illum == 2)
	{
		const float e = shininess * 0.01f;
		if(e > 0.3f)
		{
			const F3 r = reflect3(dir, normal), shv = sample_hemisphere(rng, b, e);
			dir = align_direction(shv, r);
			if(dot3(dir, normal) < 0.0f) alive = false;
			else
			{
				const float pw = canon_pow(dot3(dir, r), e);
				color = color * fma3(specular, pw, diffuse);
			}
			done = true;
		}
		else illum = 1;
	}
	if(!done)
	{
		if(illum == 1)
		{
			dir = align_direction(sample_hemisphere(rng, b, 0.0f), normal);
			color = color * diffuse;
		}
		else if(illum >= 3 && illum <= 5)
		{
			color = color * specular;
			dir = reflect3(dir, normal);
		}
		else if(illum == 6 || illum == 7)
		{
			float eta = ior;
			float cosi = dot3(dir, normal);
			float fresnel, etai, etat;
			if(cosi > 0) { etai = eta; etat = 1.0f; }
			else { etai = 1.0f; etat = eta; normal = -normal; cosi = -cosi; }
			eta = etai / etat;
			const float sint = (etai / etat) * sqrtf(gl_max(0.0f, fmaf(-cosi, cosi, 1.0f)));
			if(sint >= 1.0f) fresnel = 1.0f;
			else
			{
				const float cost = sqrtf(gl_max(0.0f, fmaf(-sint, sint, 1.0f)));
				const float A = etat * cosi, B = etai * cost, C = etai * cosi, D = etat * cost;
				const float Rs = (A - B) / (A + B);
				const float Rp = (C - D) / (C + D);
				fresnel = fmaf(Rs, Rs, Rp * Rp) * 0.5f;
			}
			const float cos2 = fmaf(-(eta * eta), fmaf(-cosi, cosi, 1.0f), 1.0f);
			float sx, sy; sobol2(rng, b, &sx, &sy);
			if(cos2 > 0 && sx >= fresnel)
			{
				const float k = fmaf(eta, cosi, sqrtf(cos2));
				dir = normalize3(fma3(normal, k, dir * eta));
			}
			else dir = reflect3(dir, normal);
		}
	}
	return alive;
}

// One iteration `b` of the for-loop of Render() (pathtracer.glsl:107-202) for the 256 live paths of queue chunk `chunk` of segment `seg`.
// store_cache: bounce 0 of a frame that traced its primary rays (pathtracer.glsl:121-127).
__device__ __forceinline__ void shade_chunk(const FrameArgs &f, const SceneArgs &sc, const QueueArgs &q, const PixelArgs &px, const ShadowArgs &sh, int b,
											 int store_cache, int count_stats, uint32_t seg, uint32_t chunk, uint32_t n_in)
{
	uint32_t local = chunk * kShadeThreads + threadIdx.x;
	if(sc.tri_class) // (kernel argument: uniform branch around the barriers of bin_by_key)
	{
		// the workgroup's paths change hands so that every wave shades one class of material: the hit's triangle names the class
		uint32_t key = kClassNone;
		if(local < n_in)
		{
			const int tri = __float_as_int(q.hit[3 * ((size_t)seg * q.seg_cap + local)]);
			key = tri < 0 || tri >= f.n_tris ? kClassMiss : (uint32_t)sc.tri_class[tri];
		}
		local = chunk * kShadeThreads + bin_by_key(key);
	}
	const uint32_t slot_in = seg * q.seg_cap + local;
	bool alive = local < n_in;
	F3 origin = f3(0, 0, 0), dir = f3(0, 0, 1), color = f3(0, 0, 0), ret = f3(0, 0, 0);
	int L = 0, pi = 0;
	const float *sobol = f.sobol;
	bool shaded = false, bad_mat = false, parked = false, escaped = false;
	F3 ret_in = f3(0, 0, 0);
	if(alive)
	{
		const float4 rd = q.ray_d[slot_in];
		const F3 h = ld3(q.hit, slot_in);
		color = ld3(q.col, slot_in);
		dir = f3(rd.x, rd.y, rd.z);
		pi = (int)(__float_as_uint(rd.w) & kPathIdMask);
		L = pi;
		if(f.batched)
		{
			const int frame = (int)((uint32_t)pi / (uint32_t)f.n_local_px);
			L = pi - frame * f.n_local_px;
			sobol += frame * 64;
		}
		// radiance picked up so far: zero for almost every path, so it is parked per path (FrameArgs::done[pi]) and only
		// touched when it changes, instead of being read and re-written (32 B) by every bounce of every path
		parked = (__float_as_uint(rd.w) & kPathParked) != 0u;
		if(parked) { const float4 r4 = f.done[pi]; ret = f3(r4.x, r4.y, r4.z); }
		ret_in = ret;
		const int tri_idx = __float_as_int(h.x);
		const float tu = h.y, tv = h.z;
		if(store_cache) px.cache[L] = make_float4(h.x, tu, tv, 0.0f);

		if(tri_idx == -1)
		{
			if(sh.enabled) escaped = true; // the sun term waits for the visibility query (k_shadow_resolve)
			else ret = fma3(color, f3(f.sun[0], f.sun[1], f.sun[2]), ret);
			alive = false;
		}
		else
		{
			const SurfaceInfo si = fetch_info(f, sc, tri_idx, tu, tv);
			origin = si.origin;
			if(si.bad_mat) { alive = false; bad_mat = true; }
			else
			{
				shaded = true;
				const uint8_t *sh = px.shift + (size_t)L * 2;
				const Rng rng{unorm8_to_float(sh[0]), unorm8_to_float(sh[1]), sobol};
				alive = respond(f, si, rng, b, dir, color, ret);
			}
		}
		if(!alive && !escaped) finish_path(f, px, pi, L, ret);
	}
	// statistics.  The FetchInfo count is only collected in instrumented runs, one atomic per workgroup: per-wave atomics
	// on a single word (23 k per launch) cost ~0.27 ms on this chip (~88 same-address atomics/us) — 75 % of this kernel.
	{
		const unsigned long long mb = __ballot(bad_mat);
		if(mb && (threadIdx.x & 63) == 0) atomicAdd(&px.stats->bad_materials, (unsigned long long)__popcll(mb));
		if(count_stats) // (kernel argument: uniform branch around the barrier)
		{
			const int n_shaded = __syncthreads_count(shaded);
			if(n_shaded && threadIdx.x == 0) atomicAdd(&px.stats->shaded, (unsigned long long)n_shaded);
		}
	}
	const uint32_t slot = append_slot(alive, q.count_out + seg * kCursorStride, seg * q.seg_cap);
	if(alive)
	{
		st3(q.out_o, slot, origin.x, origin.y, origin.z);
		if(__float_as_uint(ret.x) != __float_as_uint(ret_in.x) || __float_as_uint(ret.y) != __float_as_uint(ret_in.y) ||
		   __float_as_uint(ret.z) != __float_as_uint(ret_in.z))
		{
			f.done[pi] = make_float4(ret.x, ret.y, ret.z, 0.0f);
			parked = true;
		}
		q.out_d[slot] = make_float4(dir.x, dir.y, dir.z, __uint_as_float((uint32_t)pi | (parked ? kPathParked : 0u)));
		st3(q.out_col, slot, color.x, color.y, color.z);
	}
	if(sh.enabled) // (kernel argument: uniform branch around the barriers of append_slot)
	{
		const uint32_t sslot = append_slot(escaped, sh.count + seg * kCursorStride, seg * q.seg_cap);
		if(escaped)
		{
			const F3 ro = ld3(q.ray_o, slot_in); // the position the path escaped from (camera or last hit)
			sh.o[sslot] = make_float4(ro.x, ro.y, ro.z, f.tmin);
			sh.d[sslot] = make_float4(sh.dir[0], sh.dir[1], sh.dir[2], __int_as_float(pi));
			sh.col[sslot] = make_float4(color.x, color.y, color.z, parked ? 1.0f : 0.0f);
		}
	}
}

// One workgroup per chunk.  Round 3 measured the alternatives (profiles/r3_ablations_k_trace.txt items 9-12): several chunks per workgroup (the
// kernel is not bound by the dispatch rate), 8 waves per SIMD (+4 %), the material table in LDS, binning by material class (+10 %).
// 7 waves per SIMD (72 VGPRs): measured optimum (6: -2.4 %)
__global__ __launch_bounds__(kShadeThreads, 7) void k_shade(FrameArgs f, SceneArgs sc, QueueArgs q, PixelArgs px, ShadowArgs sh, int b, int store_cache, int count_stats)
{
	const uint32_t seg = blockIdx.x & (kNumSegments - 1), chunk = blockIdx.x >> 3;
	const uint32_t n_in = q.count_in[seg * kCursorStride];
	if(chunk * kShadeThreads >= n_in) return; // whole workgroup beyond the segment's live range (uniform exit)
	shade_chunk(f, sc, q, px, sh, b, store_cache, count_stats, seg, chunk, n_in);
}

// Bounce 0 of a whole batch in one kernel, one thread per local pixel, looping over the batch's frames (batches only; every frame starts
// from the cached primary hit of its tmpLifetime group, pathtracer.glsl:113-127).  The frames of a group share camera ray, hit and therefore
// FetchInfo: the surface is fetched ONCE per pixel and group, and per frame only the material response runs — instead of k_gen_primary
// writing a ray and a hit per path and k_shade reading them back and gathering the same triangle, material and texels 16 times.
// Same arithmetic per path as k_gen_primary + k_shade(b = 0) (fetch_info / respond are the very functions k_shade calls); which segment a
// path lands in is scheduling only.  With the sun-visibility query on (FrameArgs::sun_query) a path that escapes at bounce 0 is not finished here: its query ray
// (camera origin -> sun direction, throughput 1) goes into the queue, flagged kPathShadow, and k_path traces it and adds the sun term if nothing is hit.
__global__ __launch_bounds__(kShadeThreads, 6) void k_shade_first(FrameArgs f, SceneArgs sc, QueueArgs q, PixelArgs px, int count_stats)
{
	const uint32_t L = blockIdx.x * kShadeThreads + threadIdx.x;      // local pixel; a workgroup = four 8x8 tiles
	int x = 0, y = 0;
	const bool valid = L < (uint32_t)f.n_local_px && local_pixel_xy(f, sc.local_blocks, (int)L, &x, &y);
	Rng rng{0.0f, 0.0f, f.sobol};
	if(valid)
	{
		const uint8_t *sh = px.shift + (size_t)L * 2;
		rng.sx = unorm8_to_float(sh[0]); rng.sy = unorm8_to_float(sh[1]);
	}
	SurfaceInfo si;
	si.origin = si.normal = si.diffuse = si.specular = si.emission = f3(0, 0, 0); si.illum0 = 0; si.shininess = 0.0f; si.ior = 1.0f; si.bad_mat = false;
	F3 cam_dir = f3(0, 0, 1);
	bool hit = false;
	int group_now = -1;
	unsigned long long n_shaded = 0, n_bad = 0;
	for(int ordinal = 0; ordinal < f.n_frames; ++ordinal)
	{
		const int frame = f.frame_first + ordinal * f.frame_stride;
		const int group = frame_group(f, frame);
		if(group != group_now) // (uniform) a new tmpLifetime group: its sub-pixel offset, its cached primary hit, the surface there
		{
			group_now = group;
			if(valid)
			{
				const int sub_idx = ((f.spp + frame) / f.tmp_life) % (f.subpixel * f.subpixel);
				const float unit = 1.0f / (float)f.subpixel;
				cam_dir = camera_dir(f, x, y, (float)(sub_idx / f.subpixel) * unit, (float)(sub_idx % f.subpixel) * unit);
				const float4 h = cache_of_group(f, px, group)[L];
				const int tri_idx = __float_as_int(h.x);
				hit = tri_idx != -1;
				if(hit) si = fetch_info(f, sc, tri_idx, h.y, h.z);
			}
		}
		const int pi = frame * f.n_local_px + (int)L;
		bool alive = valid, query = false;
		F3 dir = cam_dir, color = f3(1.0f, 1.0f, 1.0f), ret = f3(0, 0, 0);
		if(valid)
		{
			if(!hit && f.sun_query) { query = true; dir = f3(f.sun_query_dir[0], f.sun_query_dir[1], f.sun_query_dir[2]); }
			else if(!hit) { ret = fma3(color, f3(f.sun[0], f.sun[1], f.sun[2]), ret); alive = false; }
			else if(si.bad_mat) { alive = false; ++n_bad; }
			else
			{
				++n_shaded;
				Rng r = rng; r.sobol = f.sobol + frame * 64;
				alive = respond(f, si, r, 0, dir, color, ret);
			}
			if(!alive) finish_path(f, px, pi, (int)L, ret);
		}
		// the (frame, workgroup) chunks of the pass are dealt round-robin to the segments: even load, and never more chunks in a segment than
		// the pass's capacity ceil(chunks / 8) (tracer.hip: seg_slots_for)
		const uint32_t seg = ((uint32_t)ordinal * gridDim.x + blockIdx.x) & (kNumSegments - 1);
		const uint32_t slot = append_slot(alive, q.count_out + seg * kCursorStride, seg * q.seg_cap);
		if(alive)
		{
			if(query) st3(q.out_o, slot, f.origin[0], f.origin[1], f.origin[2]); // (the position the path escapes from: the camera)
			else st3(q.out_o, slot, si.origin.x, si.origin.y, si.origin.z);
			bool parked = false;
			if(__float_as_uint(ret.x) != 0u || __float_as_uint(ret.y) != 0u || __float_as_uint(ret.z) != 0u)
			{
				f.done[pi] = make_float4(ret.x, ret.y, ret.z, 0.0f);
				parked = true;
			}
			q.out_d[slot] = make_float4(dir.x, dir.y, dir.z, __uint_as_float((uint32_t)pi | (parked ? kPathParked : 0u) | (query ? kPathShadow : 0u)));
			st3(q.out_col, slot, color.x, color.y, color.z);
		}
	}
	// statistics: one count per path-bounce as in k_shade, summed over the frames first
	{
		const unsigned long long mb = __ballot(n_bad != 0);
		if(mb)
		{
			unsigned long long t = n_bad;
			for(int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
			if((threadIdx.x & 63) == 0) atomicAdd(&px.stats->bad_materials, t);
		}
		if(count_stats)
		{
			unsigned long long t = n_shaded;
			for(int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
			if((threadIdx.x & 63) == 0 && t) atomicAdd(&px.stats->shaded, t);
		}
	}
}

// pathtracer.glsl:130-135 with the commented-out condition enabled: the escaped path receives the sun term only if the
// any-hit query towards the sun found nothing; then main()'s clamp + accumulate (:224-226)
__global__ __launch_bounds__(kShadeThreads) void k_shadow_resolve(FrameArgs f, QueueArgs q, PixelArgs px, ShadowArgs sh)
{
	const uint32_t seg = blockIdx.x & (kNumSegments - 1), chunk = blockIdx.x >> 3;
	const uint32_t local = chunk * kShadeThreads + threadIdx.x;
	if(local >= sh.count[seg * kCursorStride]) return;
	const uint32_t slot = seg * q.seg_cap + local;
	const float4 c4 = sh.col[slot];
	const int pi = __float_as_int(sh.d[slot].w);
	int L = pi;
	if(f.batched) L = pi - (int)((uint32_t)pi / (uint32_t)f.n_local_px) * f.n_local_px;
	F3 ret = f3(0, 0, 0);
	if(c4.w != 0.0f) { const float4 r4 = f.done[pi]; ret = f3(r4.x, r4.y, r4.z); }
	if(__float_as_int(sh.hit[slot].x) == -1) ret = fma3(f3(c4.x, c4.y, c4.z), f3(f.sun[0], f.sun[1], f.sun[2]), ret);
	finish_path(f, px, pi, L, ret);
}

// primaryray.glsl main (:46-94): the colour of a primary hit by viewer type (k_trace_camera colours a pixel when its ray has finished)
__device__ __forceinline__ F3 viewer_color(const FrameArgs &f, const SceneArgs &sc, DeviceStats *stats, int tri_idx, float u, float v, int viewer_type)
{
	F3 color = f3(0, 0, 0);
	if(tri_idx == -1) return color;
	const int matid = __float_as_int(sc.triangles[(size_t)tri_idx * kTriFloat4 + 4].z);
	if(matid < 0 || matid >= f.n_mats) { atomicAdd(&stats->bad_materials, 1ull); return color; }
	const float4 *mp = sc.materials + (size_t)matid * kMatFloat4;
	const float w = 1.0f - u - v;
	if(viewer_type == 0)
	{
		const float4 md = mp[0];
		const int dtex = __float_as_int(md.x);
		if(f.n_tex != 0 && dtex != -1 && dtex >= 0 && dtex < f.n_tex)
		{
			const float4 mt = mp[4];
			color = textured_diffuse(sc, tri_idx, make_int4(__float_as_int(mt.x), __float_as_int(mt.y), __float_as_int(mt.z), 0), u, v, w);
		}
		else color = f3(md.y, md.z, md.w);
	}
	else if(viewer_type == 1) { const float4 ms = mp[2]; color = f3(ms.y, ms.z, ms.w); }
	else if(viewer_type == 2) { const float4 me = mp[1]; color = f3(me.y, me.z, me.w); }
	else if(viewer_type == 4 || viewer_type == 5)
	{
		const TriCore tc = load_tri_core(sc, tri_idx);
		const float *tri = tc.v;
		color = viewer_type == 4 ? normalize3(bary3(tri + 9, tri + 12, tri + 15, u, v, w)) : bary3(tri + 0, tri + 3, tri + 6, u, v, w);
	}
	return color;
}

// screen.glsl main (:15-21): what the reference's full-screen quad shows for the result image — gamma 1/2.2 for the
// colour viewers and the path-traced radiance (uType <= 3), normalize(v) * 0.5 + 0.5 for the normal / position viewers —
// quantised like a GL RGBA8 UNORM colour buffer: round(clamp(c, 0, 1) * 255), NaN -> 0.  Canonical arithmetic: the
// exponent is the binary32 quotient 1.0f / 2.2f, pow is canon_pow.  One RGBA8 word per local pixel (R in the low byte).
__device__ __forceinline__ uint32_t unorm8(float c)
{
	if(!(c > 0.0f)) return 0u; // also NaN
	if(c >= 1.0f) return 255u;
	return (uint32_t)floorf(fmaf(c, 255.0f, 0.5f));
}
__global__ __launch_bounds__(256) void k_display(const float4 *accum, int n_local_px, int viewer_type, uint32_t *out)
{
	const int L = blockIdx.x * blockDim.x + threadIdx.x;
	if(L >= n_local_px) return;
	const float4 v = accum[L];
	F3 c;
	if(viewer_type <= 3)
	{
		const float g = 1.0f / 2.2f;
		c = f3(canon_pow(v.x, g), canon_pow(v.y, g), canon_pow(v.z, g));
	}
	else
	{
		const F3 n = normalize3(f3(v.x, v.y, v.z));
		c = f3(n.x * 0.5f + 0.5f, n.y * 0.5f + 0.5f, n.z * 0.5f + 0.5f);
	}
	out[L] = unorm8(c.x) | unorm8(c.y) << 8 | unorm8(c.z) << 16 | 0xff000000u;
}

// compact block-major RGBA (one rank's buffer) -> rows of the W x H x 3 image (row 0 = top)
__global__ void k_untile(const float4 *local, const int32_t *local_blocks, int n_local_px, int blocks_x, int width, int height, float *rgb)
{
	const int L = blockIdx.x * blockDim.x + threadIdx.x;
	if(L >= n_local_px) return;
	const int blk = local_blocks[L >> 10];
	const int in = L & 1023, wt = in >> 6, ln = in & 63;
	const int x = (blk % blocks_x) * kBlockDim + (wt & 3) * 8 + (ln & 7);
	const int y = (blk / blocks_x) * kBlockDim + (wt >> 2) * 8 + (ln >> 3);
	if(x >= width || y >= height) return;
	const float4 v = local[L];
	float *o = rgb + ((size_t)y * width + x) * 3;
	o[0] = v.x; o[1] = v.y; o[2] = v.z;
}

}  // namespace adypt
