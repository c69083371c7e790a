// C-ABI implementation of include/adypt_hip.h: context, HBM residency of the scene, launch sequencing of the
// wavefront path tracer on one HIP stream.  Replaces OglScene + OglPathTracer (src/Tracer/*.cpp) of the reference.
//
// Per frame (= one OglPathTracer::Trace(true), OglPathTracer.cpp:34-61):
//     memset counters -> k_gen_primary -> [ k_trace -> k_shade ] x maxBounce      (no host sync; queue sizes live in
//     device memory, the traversal kernel is persistent, the shade grid covers the worst case)
// A batch of several frames is cut into sub-batches ("pipes"), each the chain above on its OWN HIP stream over its own window
// of the ray queues: while one pipe's traversal launch drains (its last, longest rays) or its shade kernel streams the queues
// through HBM, the other pipe's traversal keeps the vector ALUs busy.  The reference has no barrier between bounces at all
// (one dispatch runs the whole for(b < uMaxBounce) loop, shaders/pathtracer.glsl:107); results do not depend on any of this:
// per-path work is independent of queue order and k_resolve applies the finished samples in frame order.
// There is no CPU fallback anywhere in this file: without a HIP device adypt_create fails with ADYPT_E_NO_DEVICE.
#include "traverse.hpp"
#include "path.hpp"
#include "ctx_access.hpp"
#include "tunables.hpp"
#include "../../../include/adypt_hip.h"
#include "../../../include/adypt_host.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace adypt;

namespace {

constexpr int kMaxBounce = 32;
constexpr int kMaxFramesInFlight = 128;
constexpr int kMaxPipes = 4;           // sub-batches of a batch that run as concurrent chains (adypt_set_pipeline)
constexpr long kRefTrianglesAutoMaxMB = 1l << 20; // ADYPT_REF_TRIANGLES_MAX_MB unset: the per-reference triangle copy is made whatever its size
constexpr int kRollMaxPixels = 1 << 22;   // single frames in a row overlap on two streams up to this many local pixels (trace_rolling_frame)
constexpr int kDefaultPipes = 1;       // measured: a second chain overlaps but recovers nothing (profiles/r3_ablations_k_trace.txt)

struct FrameCounters {                 // one memset per frame; every counter on its own 128-byte line
	uint32_t count[kMaxBounce + 1][kNumSegments * kCursorStride];   // live rays per queue segment after bounce b
	uint32_t cursor[kMaxBounce + 1][kNumSegments * kCursorStride];  // traversal fetch cursors per segment
	uint32_t sh_count[kMaxBounce + 1][kNumSegments * kCursorStride];  // sun-visibility queries per segment of bounce b
	uint32_t sh_cursor[kMaxBounce + 1][kNumSegments * kCursorStride];
};

thread_local std::string g_create_error;

struct EventPair { hipEvent_t a, b; int kind; };

// Zeroes the counters of `n` pipes.  A kernel rather than hipMemsetAsync: while round 3's queue corruption was being hunted (the
// append_slot fault) the memset was suspected of not being ordered against the kernels around it and replaced; that changed the timing, not the
// fault — the suspicion was never established.  The kernel stays because it is one launch for all pipes' counters and certainly stream-ordered.
__global__ void k_clear_counters(uint4 *p, uint32_t n16) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if(i < n16) p[i] = make_uint4(0, 0, 0, 0); }

// Slot-claim audit (adypt_set_instrumentation flag 4).  append_slot (shade.hpp) hands every surviving path of a workgroup a slot of its queue
// segment; round 3 saw a build in which whole waves claimed ONE slot (the append_slot fault).  With the audit on, the queue a kernel is
// about to append to is filled with a poison path word, and after the kernel every slot below the segment's counter must hold a real path word, every
// path id must appear once (a bitmap over the path ids), and every slot above the counter must still be poison.
constexpr uint32_t kAuditPoison = 0xffffffffu;
__global__ void k_audit_poison(float4 *out_d, size_t n, uint32_t *seen, size_t n_seen)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(i < n) out_d[i].w = __uint_as_float(kAuditPoison);
	if(i < n_seen) seen[i] = 0u;
}
__global__ void k_audit_check(const float4 *out_d, const uint32_t *count, uint32_t seg_cap, uint32_t *seen, uint32_t id_limit, unsigned long long *errors)
{
	const uint32_t seg = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= seg_cap) return;
	const uint32_t w = __float_as_uint(out_d[(size_t)seg * seg_cap + i].w);
	bool bad;
	if(i < count[seg * kCursorStride])
	{
		const uint32_t id = w & kPathIdMask;
		// never written, a path id the bitmap has no bit for (a corrupted word: counted, the bitmap is not touched), or a path that holds two slots
		bad = w == kAuditPoison || id >= id_limit || ((atomicOr(&seen[id >> 5], 1u << (id & 31u)) >> (id & 31u)) & 1u);
	}
	else bad = w != kAuditPoison;                                                                     // written beyond what the counter admits
	if(bad) atomicAdd(errors, 1ull);
}

// out[r] = the 128-byte device record of triangle tri_indices[r]: the uTriIndices remap (traversal.glsl:253-254) applied to the data once, so that k_path
// looks a hit's triangle up by the traversal's own reference index.  One thread per 16 bytes.
__global__ void k_expand_references(const float4 *triangles, const int32_t *tri_indices, size_t n_refs, float4 *out)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(i < n_refs * kTriFloat4) out[i] = triangles[(size_t)tri_indices[i / kTriFloat4] * kTriFloat4 + i % kTriFloat4];
}

// One sub-batch chain of a pipelined batch.  Pipe 0 runs on the context's stream.
struct Pipe {
	hipStream_t stream = nullptr;
	hipEvent_t done = nullptr;         // end of the pipe's chain; the context's stream waits for it before k_resolve
	FrameCounters *counters = nullptr;
	uint2 *spill = nullptr;            // traversal stack spill of this pipe's launches (two pipes' launches overlap)
};

// The part of the ray queues a pass works in: slots [offset, offset + kNumSegments * seg_cap)
struct QueueWindow { size_t offset; uint32_t seg_cap; };

}  // namespace

struct adypt_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	std::string error;
	Tunables tun;              // the environment as adypt_create found it (tunables.hpp)

	// scene (immutable after create)
	void *d_nodes = nullptr, *d_woop = nullptr, *d_tri_indices = nullptr, *d_triangles = nullptr, *d_materials = nullptr, *d_tri_class = nullptr;
	void *d_texels = nullptr, *d_local_blocks = nullptr;
	void *d_ref_triangles = nullptr;          // k_path: the triangle records once per REFERENCE (uTriIndices order), made at adypt_create; null = k_path remaps through d_tri_indices
	void *d_all_blocks = nullptr;             // adypt_assemble_radiance: block lists of all ranks
	std::vector<int64_t> all_blocks_offset;
	int64_t n_nodes = 0, n_refs = 0, n_tris = 0, n_mats = 0;
	int n_tex = 0;
	int width = 0, height = 0, blocks_x = 0, blocks_y = 0, rank = 0, nranks = 1;
	int n_local_blocks = 0, n_local_px = 0;
	int64_t n_image_px = 0; // pixels of the owned blocks that lie inside the image (= camera rays per frame)
	std::vector<int32_t> local_blocks;

	// per local pixel
	float4 *d_accum = nullptr, *d_cache = nullptr;
	float4 *d_cache_next = nullptr; // primary hits of the 2nd, 3rd, ... tmpLifetime group of a batch (shade.hpp PixelArgs)
	int cache_next_slices = 0;
	uint8_t *d_shift = nullptr;

	// wavefront queues
	int64_t capacity = 0;      // queue slots = kNumSegments * seg_cap
	uint32_t seg_cap = 0;      // slots per XCD-affine segment (multiple of kShadeThreads)
	size_t alloc_slots = 0;    // allocated slots (>= capacity: the windows of a pipelined batch round up separately)
	int pipeline = kDefaultPipes; // sub-batches per batch (adypt_set_pipeline); 1 = one chain on the context's stream
	Pipe pipes[kMaxPipes];
	hipEvent_t fork_ev = nullptr;
	// Single frames in a row (one frame per wavefront pass): frame k's k_path runs on pipe 1 + (k & 1) while frame k + 1's bounce 0 and k_path are already
	// enqueued behind it on the other pipe, so the END of frame k's launch (its last paths' sequential bounces, ~0.5 ms of 2 ms) is covered by frame k + 1.
	// roll_frame[s] = the frame whose k_path is in flight (or finished, not yet applied) in slot s, -1 = none
	int roll_frame[2] = {-1, -1};
	hipEvent_t roll_ready[2] = {nullptr, nullptr}; // bounce 0 of the slot's frame is done (context's stream) -> its k_path may start (pipe's stream)
	int single_overlap = 1;        // ADYPT_SINGLE_OVERLAP=0: single frames strictly one after the other
	float4 *q_o[2] = {nullptr, nullptr}, *q_d[2] = {nullptr, nullptr}, *q_col[2] = {nullptr, nullptr};
	float4 *d_hit = nullptr;
	float4 *sh_o = nullptr, *sh_d = nullptr, *sh_col = nullptr, *sh_hit = nullptr; // sun-visibility queue (allocated when enabled)
	int sun_visibility = 0;
	float sun_dir[3] = {0.6f, 1.0f, 0.2f}; // normalised at the time it is set
	float4 *d_done = nullptr;  // [frames_in_flight][local pixels] finished samples of a multi-frame batch
	float *d_sobol = nullptr;  // [kMaxFramesInFlight][64] Sobol points of the frames of the current batch
	// pinned staging of the Sobol points, one slot per batch in flight on the stream: the upload is then a true
	// asynchronous copy and enqueueing a batch never waits for the GPU (adypt_trace_spp_async)
	static constexpr int kSobolSlots = 4;
	float *h_sobol[kSobolSlots] = {nullptr, nullptr, nullptr, nullptr};
	hipEvent_t sobol_done[kSobolSlots] = {nullptr, nullptr, nullptr, nullptr};
	int sobol_next = 0;
	int frames_in_flight = 1;
	bool queues_ok = false;    // false after a failed (re)allocation of the queues: trace calls return ADYPT_E_STATE
	uint32_t *d_display = nullptr; // adypt_read_display: one RGBA8 word per local pixel (allocated on first use)
	RayStats *d_ray_stats = nullptr;
	uint32_t *d_audit_seen = nullptr; // slot-claim audit: one bit per path id (instrumentation flag 4)
	bool audit_selftest = false;      // ADYPT_AUDIT_SELFTEST=1: a double claim is planted before every check (tests that the detector detects)
	size_t audit_words = 0;
	FrameCounters *d_counters = nullptr; // [kMaxPipes]; pipe k uses d_counters + k
	uint32_t *d_camera_cursors = nullptr; // k_trace_camera's own fetch cursors [kNumSegments][kCursorStride] + its counts of workgroups that have left [kNumSegments + 1][kCursorStride]: zero between launches (the kernel leaves them so)
	DeviceStats *d_stats = nullptr;
	uint32_t *h_overflow = nullptr; // pinned: DeviceStats::host_overflow
	uint2 *d_spill = nullptr;  // [kMaxPipes][stack_size - lds_depth][total lanes]
	size_t spill_bytes = 0;
	size_t lds_bytes = 0;      // dynamic LDS of a traversal launch (>= the stack's: padded when it has to cap the workgroups per CU)

	// launch geometry of the persistent traversal kernel
	int num_cus = 0, trace_blocks = 0, lds_depth = 0, occupancy_api = 0;
	size_t lds_per_cu = (size_t)160 * 1024; // hipDeviceProp_t::maxSharedMemoryPerMultiProcessor
	uint32_t refill_min = kRefillMin, chunk = kChunk, bite = kBite, endgame = kEndgame;
	// Camera rays come in queue order = 8x8 pixel tiles, so a wave's rays are coherent and finish together: a wave takes a WHOLE tile when all its
	// lanes are idle (refill threshold 64, bites of 64 from the workgroup's reservation) and its lanes then walk the same nodes.  Measured (round 4,
	// k_trace_camera, primary rays only, 1080p; profiles/r4_ablations_k_path.txt item 17): threshold / bite 8/8, 16/16, 32/32, 48/48, 64/64 ->
	// 0.422 / 0.361 / 0.326 / 0.381 / 0.291 ms per launch; thresholds off the bite (28, 36 with bite 32) cost 8-38 %.  Secondary rays are
	// incoherent and keep the low threshold.
	uint32_t refill_min_primary = 64, bite_primary = 64;
	int first_fused = 1;           // ADYPT_FIRST_FUSED=0: camera rays and bounce 0 of a batch as k_gen_primary + k_shade (rounds 1-2)
	int fused_bounces = 1;         // bounces 1 .. maxBounce-1 of a batch in ONE launch (k_path, path.hpp); ADYPT_FUSED_BOUNCES=0: k_trace + k_shade per bounce
	int single_fused = 1;          // a single frame runs as a batch of one through the same pipeline (ADYPT_SINGLE_FUSED=0: gen -> [trace -> shade] x maxBounce)
	int path_blocks = 0, path_lds_depth = 0; // launch geometry of k_path
	size_t path_lds = 0;
	uint32_t shade_min = 64;       // deposited hits a wave of k_path waits for before it shades a batch
	uint32_t defer_max = 24;       // ... and a round defers them only when it holds at most this many (tunables.hpp)
	uint32_t rare_min = 48;        // deferred hits (glossy lobe / dielectric) a shading round of k_path waits for; 0 = nothing is deferred
	int deal_chunks = 1;           // k_gen_primary deals 256-path chunks round-robin to the 8 queue segments (ADYPT_GEN_DEAL=0: one contiguous run each)

	// state
	adypt_pt_params params{}, pending{};
	bool have_params = false, have_camera = false, pt_started = false;
	float origin[3] = {0, 0, 0}, inv_proj[16] = {0}, inv_view[16] = {0};
	int spp = 0;
	// frames traced ahead (adypt_set_lookahead): the last wavefront batch covered frames [batch_spp, batch_spp + batch_frames);
	// frames [batch_spp + ahead_pos, batch_spp + batch_frames) are finished samples parked in d_done, not yet in the image
	int lookahead = 0, batch_spp = 0, batch_frames = 0, ahead_pos = 0, ahead_count = 0;
	int cache_group = 0;        // which tmpLifetime group of that batch image 1 (d_cache) currently holds (0 = the batch's first)
	uint32_t shift_seed_loaded = 0;
	bool shift_loaded = false;
	int instrumentation = 0;
	int view_type = 0;          // uuViewer.uType of the image in d_accum: the viewer type of the last primary frame, 3 after path tracing

	// RCCL communicator state of the native multi-GPU path (multi.hip owns and frees it)
	void *comm = nullptr;
	void (*comm_free)(void *) = nullptr;

	std::vector<EventPair> events;
	std::vector<EventPair> free_events;
	double trace_ms = 0, shade_ms = 0, path_ms = 0;
	uint32_t trace_launches = 0, path_launches = 0;
	bool last_batch_fused = false;
};

namespace {

#define HIP_TRY(ctx, expr)                                                                             \
	do {                                                                                               \
		hipError_t e_ = (expr);                                                                        \
		if(e_ != hipSuccess) {                                                                         \
			(ctx)->error = std::string(#expr) + ": " + hipGetErrorString(e_);                          \
			return e_ == hipErrorOutOfMemory ? ADYPT_E_OOM : ADYPT_E_HIP;                              \
		}                                                                                              \
	} while(0)

int fail(adypt_ctx *c, int code, const std::string &msg) { c->error = msg; return code; }

// block ownership of the pixel-tile shard: diagonal interleave so that every rank gets sky and floor alike
inline int block_owner(int bx, int by, int nranks) { return (bx + by) % nranks; }

std::vector<int32_t> owned_blocks(int width, int height, int rank, int nranks)
{
	const int nbx = (width + kBlockDim - 1) / kBlockDim, nby = (height + kBlockDim - 1) / kBlockDim;
	std::vector<int32_t> v;
	for(int by = 0; by < nby; ++by)
		for(int bx = 0; bx < nbx; ++bx)
			if(block_owner(bx, by, nranks) == rank) v.push_back(by * nbx + bx);
	return v;
}

template <class T> int upload(adypt_ctx *c, void **dst, const T *src, size_t n)
{
	size_t bytes = std::max<size_t>(n * sizeof(T), 16);
	HIP_TRY(c, hipMalloc(dst, bytes));
	if(n) HIP_TRY(c, hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
	return ADYPT_OK;
}

// Structural validation of the BVH arrays: a corrupt child/triangle range would make the kernel read out of
// bounds, which on this hardware can reset the GPU.  Mirrors exactly the address arithmetic of k_trace.
bool validate_bvh(const adypt_scene_desc &d, std::string *why)
{
	if(d.n_nodes <= 0) { *why = "empty node array"; return false; }
	const uint8_t *nodes = (const uint8_t *)d.nodes;
	for(int64_t i = 0; i < d.n_nodes; ++i)
	{
		const uint8_t *n = nodes + i * 80;
		uint32_t child_base, tri_base;
		memcpy(&child_base, n + 16, 4);
		memcpy(&tri_base, n + 20, 4);
		const uint8_t imask = n[15];
		const uint8_t *meta = n + 24;
		int n_inner = __builtin_popcount(imask);
		if(n_inner && (uint64_t)child_base + (uint64_t)n_inner > (uint64_t)d.n_nodes) { *why = "node " + std::to_string(i) + ": child range out of bounds"; return false; }
		for(int s = 0; s < 8; ++s)
		{
			const uint32_t m = meta[s];
			if(m == 0) continue;
			const bool inner = (m & (m << 1)) & 0x10;
			if(inner)
			{
				// hit bit (24 + widx) must address a set imask bit, and ONLY that bit may be raised: the kernel ORs
				// child_bits << bit_index into the hit mask, so child_bits other than 0b001 would mark slots that are not in
				// imask and the child index base + popcount(imask below slot) could step one past the validated range
				const uint32_t widx = (m & 31u) - 24u;
				if((m >> 5) != 1u) { *why = "node " + std::to_string(i) + ": inner child with child bits != 001"; return false; }
				if(widx > 7 || !((imask >> widx) & 1u)) { *why = "node " + std::to_string(i) + ": inner child not in imask"; return false; }
			}
			else
			{
				const uint32_t off = m & 31u, bits = (m >> 5) & 7u;
				const uint32_t top = bits ? 32u - (uint32_t)__builtin_clz(bits) : 0u;
				if(off + top > 24u) { *why = "node " + std::to_string(i) + ": leaf bits exceed 24"; return false; }
				if((uint64_t)tri_base + off + top > (uint64_t)d.n_refs) { *why = "node " + std::to_string(i) + ": triangle range out of bounds"; return false; }
			}
		}
	}
	for(int64_t i = 0; i < d.n_refs; ++i)
		if(d.tri_indices[i] < 0 || d.tri_indices[i] >= d.n_tris) { *why = "tri_indices[" + std::to_string(i) + "] out of range"; return false; }
	return true;
}

hipEvent_t *begin_timing(adypt_ctx *c, int kind, hipStream_t stream)
{
	if(!(c->instrumentation & 1)) return nullptr;
	EventPair p;
	if(!c->free_events.empty()) { p = c->free_events.back(); c->free_events.pop_back(); }
	else { if(hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return nullptr; }
	p.kind = kind;
	c->events.push_back(p);
	(void)hipEventRecord(c->events.back().a, stream);
	return &c->events.back().b;
}
inline void end_timing(hipEvent_t *stop, hipStream_t stream) { if(stop) (void)hipEventRecord(*stop, stream); }

void harvest_events(adypt_ctx *c)
{
	std::vector<EventPair> pending; // launches of a frame started ahead on its own stream may still be running: their turn comes later
	for(EventPair &p : c->events)
	{
		if(hipEventQuery(p.b) == hipErrorNotReady) { pending.push_back(p); continue; }
		float ms = 0.0f;
		if(hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess)
		{
			if(p.kind == 0 || p.kind == 2) { c->trace_ms += ms; ++c->trace_launches; } else c->shade_ms += ms;
			if(p.kind == 2) { c->path_ms += ms; ++c->path_launches; }
		}
		c->free_events.push_back(p);
	}
	c->events.swap(pending);
}

void clear_counters(adypt_ctx *c, FrameCounters *first, int n, hipStream_t stream)
{
	static_assert(sizeof(FrameCounters) % 16 == 0, "FrameCounters is cleared 16 bytes at a time");
	const uint32_t n16 = (uint32_t)(sizeof(FrameCounters) / 16) * (uint32_t)n;
	hipLaunchKernelGGL(k_clear_counters, dim3((n16 + 255) / 256), dim3(256), 0, stream, (uint4 *)first, n16);
}

int ensure_spill(adypt_ctx *c, int stack_size)
{
	const int extra = stack_size - c->lds_depth, extra_path = c->path_blocks ? stack_size - c->path_lds_depth : 0;
	if(extra <= 0 && extra_path <= 0) return ADYPT_OK;
	const size_t per_pipe = std::max((size_t)std::max(extra, 0) * (size_t)c->trace_blocks, (size_t)std::max(extra_path, 0) * (size_t)c->path_blocks) * kTraceThreads;
	const size_t need = per_pipe * kMaxPipes * sizeof(uint2);
	if(need > c->spill_bytes)
	{
		HIP_TRY(c, hipDeviceSynchronize()); // launches of any pipe may still be using the old array
		if(c->d_spill) (void)hipFree(c->d_spill);
		c->d_spill = nullptr; c->spill_bytes = 0;
		HIP_TRY(c, hipMalloc((void **)&c->d_spill, need));
		c->spill_bytes = need;
	}
	for(int k = 0; k < kMaxPipes; ++k) c->pipes[k].spill = c->d_spill + (size_t)k * per_pipe;
	return ADYPT_OK;
}

// geometry of the persistent traversal launch for a given stack size
int configure_trace(adypt_ctx *c, int stack_size)
{
	c->lds_depth = std::max(1, std::min(stack_size, kLdsStackMax));
	// testing / tuning hook: a smaller LDS part pushes stack entries into the global spill array (tests cover that path)
	if(c->tun.lds_stack_depth > 0) c->lds_depth = std::max(1, std::min(c->lds_depth, c->tun.lds_stack_depth));
	size_t lds = (size_t)(kTraceThreads / 64) * c->lds_depth * 64 * sizeof(uint2) + sizeof(WgPool) + kTripTabBytes; // stacks + the workgroup's ray pool + the waves' triangle hand-out tables
	int per_cu = 0;
	HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace<false>, kTraceThreads, lds));
	c->occupancy_api = per_cu;
	per_cu = std::max(1, std::min(per_cu, 8));
	// 80 VGPRs allow 6 waves per SIMD.  Measured (round 3, tools/sweep_env.py): 6 instead of 5 workgroups per CU is +2.6 % on the bench scene
	// (BVH 20 MB), +2 % on the 1.2 M-triangle scene (69 MB) and -1 % on the 10 M-triangle scene (602 MB), where the sixth wave's lines evict
	// the others' from the caches: a BVH beyond the 256 MB Infinity Cache keeps 5
	const size_t bvh_bytes = (size_t)c->n_nodes * 80 + (size_t)c->n_refs * 52;
	if(bvh_bytes > ((size_t)256 << 20)) per_cu = std::min(per_cu, 5);
	if(c->tun.trace_blocks_per_cu > 0) // tuning override
	{
		const int want = c->tun.trace_blocks_per_cu;
		// Fewer workgroups per CU than the registers allow: the traversal launches of two pipes overlap, and together they would
		// fill the CU again — so the launch asks for as much LDS as makes `want` workgroups the most that fit in a CU's 160 KB
		if(want < per_cu) lds = std::max(lds, std::min<size_t>(64 * 1024, ((c->lds_per_cu - 2048) / (size_t)want) & ~(size_t)1023));
		per_cu = want;
	}
	c->lds_bytes = lds;
	c->trace_blocks = c->num_cus * per_cu;
	// k_path (path.hpp): as many workgroups per CU as its registers allow, with the deepest LDS stack that still fits next to the path table
	{
		const int want0 = c->tun.path_blocks_per_cu > 0 ? c->tun.path_blocks_per_cu : 6;
		const size_t fixed = path_lds_bytes(0), per_entry = (size_t)(kTraceThreads / 64) * 64 * sizeof(uint2);
		int chosen = 0, depth = 1;
		for(int want = want0; want >= 1 && !chosen; --want)
		{
			// LDS is handed out in granules; the occupancy query does not know: measured, it answers 6 for 26944 bytes per workgroup, of which a
			// compute unit then runs 5 at a time (the sixth of every six waits for a slot: -5 %).  So the budget is whole KiB of (LDS per CU) / want.
			// ... of (LDS per CU - 2 KiB) / want: measured (round 6, tools/sweep_env.py), 5 workgroups of 32384 bytes = 161920 of the CU's 163840 do NOT run together — the
			// "5 per CU" of rounds 4-5 ran 4 (the rate of ADYPT_PATH_BLOCKS_PER_CU=4 to the percent) — while 5 of 30336 do
			const size_t budget = std::min<size_t>((((c->lds_per_cu - 2048) / 1024) / (size_t)want) * 1024, 64 * 1024);
			if(budget < fixed + per_entry) continue;
			int d = (int)std::min<size_t>((size_t)std::min(stack_size, kLdsStackMax), (budget - fixed) / per_entry);
			if(c->tun.path_lds_depth > 0) d = std::max(1, std::min(d, c->tun.path_lds_depth));
			int got = 0; // the query is made with THIS want's depth, and judged against this want
			HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&got, (k_path<false, false>), kTraceThreads, path_lds_bytes(d)));
			int got_sun = 0; // (the variant with sun-visibility queries among its rays must fit as well: the launch geometry is one)
			HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&got_sun, (k_path<false, true>), kTraceThreads, path_lds_bytes(d)));
			got = std::min(got, got_sun);
			if(got >= want) { chosen = want; depth = d; }
		}
		if(!chosen) return fail(c, ADYPT_E_HIP, "k_path does not fit a compute unit");
		c->path_lds_depth = depth; c->path_lds = path_lds_bytes(depth);
		c->path_blocks = c->num_cus * chosen;
		if(c->tun.path_verbose) fprintf(stderr, "[adypt] k_path: %d workgroups per CU, %d path slots each, LDS stack depth %d, %zu bytes of LDS; a hit's triangle record by %s\n", chosen, kPathSlots, depth, c->path_lds,
		                                c->d_ref_triangles ? "reference index (per-reference copy of the records)" : "uTriIndices remap in the shading round (no per-reference copy)");
	}
	return ensure_spill(c, stack_size);
}

// bounces b0 .. maxBounce-1 of a batched pass in one launch: the queue of parity `parity` holds the pass's rays of bounce b0
int launch_path(adypt_ctx *c, const Pipe &pipe, const QueueWindow &win, int parity, const uint32_t *count, uint32_t *cursor, const FrameArgs &f, const SceneArgs &sc,
				const PixelArgs &px, int b0, bool stats)
{
	PathArgs a;
	a.nodes = (const uint4 *)c->d_nodes; a.woop = (const float4 *)c->d_woop;
	a.in_o = (const float *)c->q_o[parity] + 3 * win.offset; a.in_d = c->q_d[parity] + win.offset; a.in_col = (const float *)c->q_col[parity] + 3 * win.offset;
	a.ray_stats = nullptr;
	a.tri_remap = c->d_ref_triangles ? nullptr : (const int32_t *)c->d_tri_indices;
	a.count = count; a.cursor = cursor;
	a.spill = pipe.spill; a.stats = c->d_stats;
	a.seg_cap = win.seg_cap;
	a.stack_size = c->params.stack_size; a.lds_depth = c->path_lds_depth;
	a.refill_min = c->refill_min; a.shade_min = c->shade_min; a.rare_min = c->rare_min; a.defer_max = c->defer_max;
	a.b0 = b0; a.tmin = c->params.ray_tmin;
	hipEvent_t *stop = begin_timing(c, 2, pipe.stream);
	const PathKernArgs K{a, f, sc, px, stats ? 1 : 0};
	// (the SUN variant: rays that end at their first accepted triangle among the others — only where the queue can hold such queries)
	if(f.sun_query)
	{
		if(stats) hipLaunchKernelGGL((k_path<true, true>), dim3(c->path_blocks), dim3(kTraceThreads), c->path_lds, pipe.stream, K);
		else hipLaunchKernelGGL((k_path<false, true>), dim3(c->path_blocks), dim3(kTraceThreads), c->path_lds, pipe.stream, K);
	}
	else if(stats) hipLaunchKernelGGL((k_path<true, false>), dim3(c->path_blocks), dim3(kTraceThreads), c->path_lds, pipe.stream, K);
	else hipLaunchKernelGGL((k_path<false, false>), dim3(c->path_blocks), dim3(kTraceThreads), c->path_lds, pipe.stream, K);
	end_timing(stop, pipe.stream);
	HIP_TRY(c, hipGetLastError());
	return ADYPT_OK;
}

int launch_trace(adypt_ctx *c, const Pipe &pipe, const QueueWindow &win, int parity, const uint32_t *count, uint32_t *cursor, int stack_size, bool stats,
				 RayStats *ray_stats, bool any_hit = false, bool shadow_queue = false, bool packed = false, bool camera_rays = false)
{
	TraceArgs a;
	a.nodes = (const uint4 *)c->d_nodes;
	a.woop = (const float4 *)c->d_woop;
	a.tri_indices = (const int32_t *)c->d_tri_indices;
	// the path tracer's own queues hold 12-byte origins and hits (shade.hpp); ray batches handed in by the caller and the sun-visibility
	// queue are float4 records.  The buffers are allocated 16 bytes per slot either way; a window starts 3 (or 4) floats x offset in.
	a.packed = packed ? 1u : 0u; a.tmin = c->params.ray_tmin;
	if(packed)
	{
		a.ray_o = (const float4 *)((const float *)c->q_o[parity] + 3 * win.offset);
		a.hit = (float4 *)((float *)c->d_hit + 3 * win.offset);
		a.ray_d = c->q_d[parity] + win.offset;
	}
	else
	{
		a.ray_o = (shadow_queue ? c->sh_o : c->q_o[parity]) + win.offset; a.ray_d = (shadow_queue ? c->sh_d : c->q_d[parity]) + win.offset;
		a.hit = (shadow_queue ? c->sh_hit : c->d_hit) + win.offset;
	}
	a.ray_stats = ray_stats;
	a.count = count; a.cursor = cursor;
	a.spill = pipe.spill;
	a.stats = c->d_stats;
	a.seg_cap = win.seg_cap;
	a.refill_min = camera_rays ? c->refill_min_primary : c->refill_min; a.chunk = c->chunk; a.bite = camera_rays ? c->bite_primary : c->bite; a.endgame = c->endgame;
	a.stack_size = stack_size; a.lds_depth = c->lds_depth;
	const size_t lds = c->lds_bytes;
	hipEvent_t *stop = begin_timing(c, 0, pipe.stream);
	if(any_hit)
	{
		if(stats) hipLaunchKernelGGL((k_trace<true, true>), dim3(c->trace_blocks), dim3(kTraceThreads), lds, pipe.stream, a);
		else hipLaunchKernelGGL((k_trace<false, true>), dim3(c->trace_blocks), dim3(kTraceThreads), lds, pipe.stream, a);
	}
	else if(stats) hipLaunchKernelGGL(k_trace<true>, dim3(c->trace_blocks), dim3(kTraceThreads), lds, pipe.stream, a);
	else hipLaunchKernelGGL(k_trace<false>, dim3(c->trace_blocks), dim3(kTraceThreads), lds, pipe.stream, a);
	end_timing(stop, pipe.stream);
	HIP_TRY(c, hipGetLastError());
	return ADYPT_OK;
}

void fill_frame(const adypt_ctx *c, FrameArgs *f)
{
	memset(f, 0, sizeof(*f));
	memcpy(f->inv_proj, c->inv_proj, 64); memcpy(f->inv_view, c->inv_view, 64);
	memcpy(f->origin, c->origin, 12);
	f->tmin = c->params.ray_tmin;
	memcpy(f->sun, c->params.sun, 12);
	f->clamp = c->params.clamp;
	f->width = c->width; f->height = c->height;
	f->spp = c->spp; f->subpixel = c->params.subpixel; f->tmp_life = c->params.tmp_lifetime; f->max_bounce = c->params.max_bounce;
	f->sobol = c->d_sobol; f->done = c->d_done; f->n_frames = 1; f->frame_first = 0; f->frame_stride = 1; f->batched = 0;
	f->n_local_px = c->n_local_px; f->blocks_x = c->blocks_x; f->rank = c->rank; f->nranks = c->nranks;
	f->n_tris = (int32_t)c->n_tris; f->n_mats = (int32_t)c->n_mats; f->n_tex = c->n_tex;
	f->deal_chunks = c->deal_chunks;
	memcpy(f->sun_query_dir, c->sun_dir, 12); f->sun_query = 0; // (set by the frame driver where the one-launch pipeline carries the query)
}
void fill_scene(const adypt_ctx *c, SceneArgs *s)
{
	s->triangles = (const float4 *)c->d_triangles;
	s->materials = (const float4 *)c->d_materials;
	s->texels = (const uint32_t *)c->d_texels;
	s->local_blocks = (const int32_t *)c->d_local_blocks;
	s->tri_class = (const uint8_t *)c->d_tri_class;
}
void fill_pixels(const adypt_ctx *c, PixelArgs *p)
{
	p->accum = c->d_accum; p->cache = c->d_cache; p->cache_next = c->d_cache_next; p->shift = c->d_shift; p->stats = c->d_stats;
}
// slots per segment that `frames` frames of this context's pixels need (multiple of kShadeThreads)
uint32_t seg_slots_for(const adypt_ctx *c, int frames)
{
	const size_t paths = (size_t)std::max(c->n_local_px, 64) * (size_t)std::max(1, frames);
	const size_t chunks = (paths + kShadeThreads - 1) / kShadeThreads;
	return (uint32_t)(((chunks + kNumSegments - 1) / kNumSegments) * kShadeThreads);
}
// paths per queue segment of a pass over `frames` frames (QueueArgs::seg_paths); its kernels run 8 x seg_paths / 256 workgroups.
// Sizing the grids for the allocated capacity instead cost ~6 ns per empty workgroup: 13 ms per 8-bounce batch at 66 M slots.
uint32_t pass_seg_paths(const adypt_ctx *c, const QueueWindow &win, int frames) { return std::min(win.seg_cap, seg_slots_for(c, frames)); }

inline QueueWindow full_window(const adypt_ctx *c) { return QueueWindow{0, c->seg_cap}; }
// window of sub-batch k when a batch is cut into n_pipes sub-batches of at most ceil(frames_in_flight / n_pipes) frames
inline QueueWindow pipe_window(const adypt_ctx *c, int k, int n_pipes)
{
	if(n_pipes <= 1) return full_window(c);
	const uint32_t cap = seg_slots_for(c, (c->frames_in_flight + n_pipes - 1) / n_pipes);
	return QueueWindow{(size_t)k * (size_t)cap * kNumSegments, cap};
}

// Camera rays of a pass -> traversal -> cache images, one launch (k_trace_camera) and nothing in front of it.  `f` names the frames of the pass
// (n_frames, frame_first, frame_stride).  On the context's stream only: the launches share one set of fetch cursors.
int launch_trace_camera(adypt_ctx *c, const Pipe &pipe, const QueueWindow &win, const FrameArgs &f, const PixelArgs &px, int bias_mode, bool stats, int viewer_type = -1)
{
	TraceCameraArgs K;
	memset(&K, 0, sizeof(K));
	TraceArgs &a = K.a;
	a.nodes = (const uint4 *)c->d_nodes; a.woop = (const float4 *)c->d_woop; a.tri_indices = (const int32_t *)c->d_tri_indices;
	a.packed = 1u; a.tmin = c->params.ray_tmin;
	a.cursor = c->d_camera_cursors; K.left = c->d_camera_cursors + kNumSegments * kCursorStride;
	a.spill = pipe.spill; a.stats = c->d_stats;
	K.seg_paths = pass_seg_paths(c, win, f.n_frames);
	K.seg_shift = 8;
	while((1u << K.seg_shift) < K.seg_paths) ++K.seg_shift;
	a.seg_cap = 1u << K.seg_shift; // (positions are numbers: nothing is stored at them)
	K.rays = (unsigned long long)c->n_image_px * (unsigned long long)f.n_frames;
	a.refill_min = c->refill_min_primary; a.chunk = c->chunk; a.bite = c->bite_primary; a.endgame = c->endgame;
	a.stack_size = c->params.stack_size; a.lds_depth = c->lds_depth;
	K.f = f; K.local_blocks = (const int32_t *)c->d_local_blocks; K.px = px; K.bias_mode = bias_mode;
	hipEvent_t *stop = begin_timing(c, 0, pipe.stream);
	const bool viewer = viewer_type >= 0; // a primary-only call: the pixel is coloured when its ray has finished (no viewer launch)
	if(viewer) { fill_scene(c, &K.sc); K.viewer_type = viewer_type; }
	const dim3 grid(c->trace_blocks), block(kTraceThreads);
	if(viewer)
	{
		if(stats) hipLaunchKernelGGL((k_trace_camera<true, true>), grid, block, c->lds_bytes, pipe.stream, K);
		else hipLaunchKernelGGL((k_trace_camera<false, true>), grid, block, c->lds_bytes, pipe.stream, K);
	}
	else if(stats) hipLaunchKernelGGL((k_trace_camera<true, false>), grid, block, c->lds_bytes, pipe.stream, K);
	else hipLaunchKernelGGL((k_trace_camera<false, false>), grid, block, c->lds_bytes, pipe.stream, K);
	end_timing(stop, pipe.stream);
	HIP_TRY(c, hipGetLastError());
	return ADYPT_OK;
}

QueueArgs queue_args(adypt_ctx *c, const QueueWindow &win, int in, const uint32_t *count_in, uint32_t *count_out, int frames = 0)
{
	QueueArgs q;
	q.seg_paths = frames > 0 ? pass_seg_paths(c, win, frames) : win.seg_cap;
	q.ray_o = (float *)c->q_o[in] + 3 * win.offset; q.ray_d = c->q_d[in] + win.offset; q.col = (float *)c->q_col[in] + 3 * win.offset;
	q.hit = (float *)c->d_hit + 3 * win.offset;
	q.out_o = (float *)c->q_o[in ^ 1] + 3 * win.offset; q.out_d = c->q_d[in ^ 1] + win.offset; q.out_col = (float *)c->q_col[in ^ 1] + 3 * win.offset;
	q.count_in = count_in; q.count_out = count_out;
	q.seg_cap = win.seg_cap;
	return q;
}

int check_async_errors(adypt_ctx *c)
{
	// (after a synchronisation of the streams the kernels ran on: the word is in pinned host memory, the kernels write it themselves)
	if(*(volatile uint32_t *)c->h_overflow) return fail(c, ADYPT_E_STACK_OVERFLOW, "traversal stack overflow: increase pathTracer.stackSize (currently " + std::to_string(c->params.stack_size) + ")");
	return ADYPT_OK;
}

int load_shift(adypt_ctx *c)
{
	if(c->shift_loaded && c->shift_seed_loaded == c->params.shift_seed) return ADYPT_OK;
	std::vector<uint8_t> full((size_t)c->width * c->height * 2), local((size_t)c->n_local_px * 2, 0);
	adypt_shift_bytes(c->params.shift_seed, c->width, c->height, full.data());
	for(int L = 0; L < c->n_local_px; ++L)
	{
		const int blk = c->local_blocks[(size_t)(L >> 10)];
		const int in = L & 1023, wt = in >> 6, ln = in & 63;
		const int x = (blk % c->blocks_x) * kBlockDim + (wt & 3) * 8 + (ln & 7), y = (blk / c->blocks_x) * kBlockDim + (wt >> 2) * 8 + (ln >> 3);
		if(x < c->width && y < c->height)
		{
			local[(size_t)L * 2] = full[((size_t)y * c->width + x) * 2];
			local[(size_t)L * 2 + 1] = full[((size_t)y * c->width + x) * 2 + 1];
		}
	}
	// frames enqueued earlier (adypt_trace_spp_async, then adypt_reset + adypt_set_params) may still be reading d_shift
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	if(!local.empty()) HIP_TRY(c, hipMemcpy(c->d_shift, local.data(), local.size(), hipMemcpyHostToDevice));
	c->shift_loaded = true; c->shift_seed_loaded = c->params.shift_seed;
	return ADYPT_OK;
}

// (re)allocate the wavefront queues for `fif` frames in flight: capacity = fif x local pixels, cut into 8 segments
int alloc_queues_raw(adypt_ctx *c, int fif)
{
	void *old[] = {c->q_o[0], c->q_o[1], c->q_d[0], c->q_d[1], c->q_col[0], c->q_col[1], c->d_hit, c->d_done, c->d_ray_stats,
				   c->sh_o, c->sh_d, c->sh_col, c->sh_hit};
	for(void *b : old) if(b) (void)hipFree(b);
	c->q_o[0] = c->q_o[1] = c->q_d[0] = c->q_d[1] = c->q_col[0] = c->q_col[1] = c->d_hit = c->d_done = nullptr;
	c->sh_o = c->sh_d = c->sh_col = c->sh_hit = nullptr; // re-created by ensure_shadow_queue when the option is on
	c->d_ray_stats = nullptr;
	const size_t npx = (size_t)std::max(c->n_local_px, 64);
	const size_t paths = npx * (size_t)fif;
	if(paths >= ((size_t)1 << 31)) return fail(c, ADYPT_E_INVALID, "frames in flight x pixels exceeds 2^31 paths");
	const size_t chunks = (paths + kShadeThreads - 1) / kShadeThreads;
	c->seg_cap = (uint32_t)(((chunks + kNumSegments - 1) / kNumSegments) * kShadeThreads);
	c->capacity = (int64_t)c->seg_cap * kNumSegments;
	c->frames_in_flight = fif;
	size_t nq = (size_t)c->capacity;
	for(int n = 2; n <= kMaxPipes; ++n) // the windows of an n-way split round up one by one
		nq = std::max(nq, (size_t)n * kNumSegments * (size_t)seg_slots_for(c, (fif + n - 1) / n));
	c->alloc_slots = nq;
	for(int i = 0; i < 2; ++i)
	{
		HIP_TRY(c, hipMalloc((void **)&c->q_o[i], nq * sizeof(float4)));
		HIP_TRY(c, hipMalloc((void **)&c->q_d[i], nq * sizeof(float4)));
		HIP_TRY(c, hipMalloc((void **)&c->q_col[i], nq * sizeof(float4)));
	}
	HIP_TRY(c, hipMalloc((void **)&c->d_hit, nq * sizeof(float4)));
	// finished samples of a batch / parked radiance of live paths; two frames at least: single frames in a row alternate between two slots
	HIP_TRY(c, hipMalloc((void **)&c->d_done, npx * (size_t)std::max(fif, 2) * sizeof(float4)));
	return ADYPT_OK;
}

// A failed (re)allocation must not leave a context that launches kernels on null queues: fall back to the previous
// frames-in-flight; if even that cannot be had, the context is marked unusable and every trace call returns ADYPT_E_STATE.
int alloc_queues(adypt_ctx *c, int fif)
{
	const int previous = c->queues_ok ? c->frames_in_flight : 0;
	int r = alloc_queues_raw(c, fif);
	c->queues_ok = r == ADYPT_OK;
	if(r == ADYPT_OK) return r;
	const std::string why = c->error;
	(void)hipGetLastError();
	if(previous > 0 && previous != fif && alloc_queues_raw(c, previous) == ADYPT_OK)
	{
		c->queues_ok = true;
		c->error = why + " (kept " + std::to_string(previous) + " frames in flight)";
		return r;
	}
	// nothing usable is left: free the partial allocation of the failed attempt
	void *bufs[] = {c->q_o[0], c->q_o[1], c->q_d[0], c->q_d[1], c->q_col[0], c->q_col[1], c->d_hit, c->d_done};
	for(void *b : bufs) if(b) (void)hipFree(b);
	c->q_o[0] = c->q_o[1] = c->q_d[0] = c->q_d[1] = c->q_col[0] = c->q_col[1] = c->d_hit = c->d_done = nullptr;
	c->capacity = 0; c->seg_cap = 0; c->alloc_slots = 0; c->frames_in_flight = previous;
	c->error = why + " (the context has no ray queues left: destroy it)";
	return r;
}

int ensure_shadow_queue(adypt_ctx *c)
{
	if(c->sh_o) return ADYPT_OK;
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	const size_t nq = c->alloc_slots;
	HIP_TRY(c, hipMalloc((void **)&c->sh_o, nq * sizeof(float4)));
	HIP_TRY(c, hipMalloc((void **)&c->sh_d, nq * sizeof(float4)));
	HIP_TRY(c, hipMalloc((void **)&c->sh_col, nq * sizeof(float4)));
	HIP_TRY(c, hipMalloc((void **)&c->sh_hit, nq * sizeof(float4)));
	return ADYPT_OK;
}

int ensure_cache_slices(adypt_ctx *c, int extra)
{
	if(extra <= c->cache_next_slices) return ADYPT_OK;
	HIP_TRY(c, hipStreamSynchronize(c->stream)); // the previous batch may still be reading the slices about to be freed
	if(c->d_cache_next) (void)hipFree(c->d_cache_next); // (they hold nothing between batches)
	c->d_cache_next = nullptr; c->cache_next_slices = 0;
	HIP_TRY(c, hipMalloc((void **)&c->d_cache_next, (size_t)extra * (size_t)std::max(c->n_local_px, 64) * sizeof(float4)));
	c->cache_next_slices = extra;
	return ADYPT_OK;
}

// the audit's bitmap over the path ids (one bit per queue slot is enough: ids are < capacity)
int ensure_audit(adypt_ctx *c)
{
	const size_t words = ((size_t)c->alloc_slots + 31) / 32 + 1;
	if(c->d_audit_seen && c->audit_words >= words) return ADYPT_OK;
	HIP_TRY(c, hipDeviceSynchronize());
	if(c->d_audit_seen) (void)hipFree(c->d_audit_seen);
	c->d_audit_seen = nullptr; c->audit_words = 0;
	HIP_TRY(c, hipMalloc((void **)&c->d_audit_seen, words * kMaxPipes * sizeof(uint32_t))); // one bitmap per chain: they check concurrently
	c->audit_words = words;
	return ADYPT_OK;
}
// around a kernel that appends to q's output queue: poison before, check after (both on the launch's stream)
void audit_before(adypt_ctx *c, const QueueArgs &q, hipStream_t stream, int pipe = 0)
{
	if(!(c->instrumentation & 4) || !c->d_audit_seen || c->audit_words * 32 < c->alloc_slots) return;
	const size_t n = (size_t)kNumSegments * q.seg_cap, n_seen = c->audit_words; // path ids are numbered over the whole batch, not over the chain's window
	hipLaunchKernelGGL(k_audit_poison, dim3((unsigned)((std::max(n, n_seen) + 255) / 256)), dim3(256), 0, stream, q.out_d, n, c->d_audit_seen + (size_t)pipe * c->audit_words, n_seen);
}
__global__ void k_audit_plant(float4 *out_d, const uint32_t *count) { if(count[0] >= 2u) out_d[1].w = out_d[0].w; } // (self-test of the detector: two slots, one path)
void audit_after(adypt_ctx *c, const QueueArgs &q, hipStream_t stream, int pipe = 0)
{
	if(!(c->instrumentation & 4) || !c->d_audit_seen || c->audit_words * 32 < c->alloc_slots) return;
	if(c->audit_selftest) hipLaunchKernelGGL(k_audit_plant, dim3(1), dim3(1), 0, stream, q.out_d, (const uint32_t *)q.count_out);
	hipLaunchKernelGGL(k_audit_check, dim3((q.seg_cap + 255) / 256, kNumSegments), dim3(256), 0, stream, (const float4 *)q.out_d, (const uint32_t *)q.count_out, q.seg_cap, c->d_audit_seen + (size_t)pipe * c->audit_words,
					   (uint32_t)std::min<size_t>(c->audit_words * 32, 0xffffffffu), &c->d_stats->audit_errors);
}

int ensure_ray_stats(adypt_ctx *c)
{
	if(c->d_ray_stats) return ADYPT_OK;
	HIP_TRY(c, hipMalloc((void **)&c->d_ray_stats, (size_t)c->capacity * sizeof(RayStats)));
	return ADYPT_OK;
}

// update_config_args (OglPathTracer.cpp:214-225): pending parameters become active
int apply_params(adypt_ctx *c)
{
	c->params = c->pending;
	int r = configure_trace(c, c->params.stack_size);
	if(r != ADYPT_OK) return r;
	// cache slices for the tmpLifetime groups a batch of frames_in_flight frames can span (none for one frame at a time)
	const int life = std::max(1, c->params.tmp_lifetime);
	if(c->frames_in_flight > 1)
	{
		r = ensure_cache_slices(c, (c->frames_in_flight - 2) / life + 1);
		if(r != ADYPT_OK) return r;
	}
	return load_shift(c);
}

// Running-mean step (pathtracer.glsl:224-226) of frames [first, first + count) of the batch last traced (its finished samples
// are parked in d_done), in frame order; afterwards image 1 holds the primary hits of the tmpLifetime group of the last frame
// applied — what frame-by-frame tracing leaves there (pathtracer.glsl:121-127).
int resolve_batch_frames(adypt_ctx *c, const SceneArgs &sc, const PixelArgs &px, int first, int count)
{
	if(count <= 0) return ADYPT_OK;
	FrameArgs f;
	fill_frame(c, &f);
	f.spp = c->batch_spp; f.n_frames = c->batch_frames;
	hipEvent_t *stop = begin_timing(c, 1, c->stream);
	hipLaunchKernelGGL(k_resolve, dim3((c->n_local_px + 255) / 256), dim3(256), 0, c->stream, f, sc, px, first, count);
	end_timing(stop, c->stream);
	HIP_TRY(c, hipGetLastError());
	const int life = std::max(1, c->params.tmp_lifetime);
	const int group = (c->batch_spp + first + count - 1) / life - c->batch_spp / life;
	if(group > c->cache_group)
	{
		HIP_TRY(c, hipMemcpyAsync(c->d_cache, c->d_cache_next + (size_t)(group - 1) * (size_t)c->n_local_px, (size_t)c->n_local_px * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
		c->cache_group = group;
	}
	return ADYPT_OK;
}

// frames traced ahead belong to the camera / parameters / queues they were traced with: anything that changes those drops
// them (they are re-traced on demand — the sample sequence is a function of the frame index alone)
// The same for single frames whose k_path was started ahead in a rolling slot: waited for (their kernels read queues, counters and the camera's
// cache image) and forgotten.
void drop_rolling(adypt_ctx *c)
{
	if(c->roll_frame[0] < 0 && c->roll_frame[1] < 0) return;
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream);
	for(int s = 0; s < 2; ++s) { (void)hipStreamSynchronize(c->pipes[1 + s].stream); c->roll_frame[s] = -1; }
}
inline void drop_lookahead(adypt_ctx *c) { c->ahead_count = 0; c->ahead_pos = 0; drop_rolling(c); }


// Sobol::Next (src/Util/Sobol.cpp:16-21) for frames [first, first + m): staged in a pinned slot, copied to `dst` on the context's stream
int upload_sobol(adypt_ctx *c, int first, int m, float *dst)
{
	const int max_bounce = c->params.max_bounce;
	const int slot = c->sobol_next;
	c->sobol_next = (slot + 1) % adypt_ctx::kSobolSlots;
	HIP_TRY(c, hipEventSynchronize(c->sobol_done[slot])); // the copy that last used this slot has left it
	std::vector<float> pts((size_t)m * 2 * max_bounce);
	int r = adypt_sobol_points(2 * max_bounce, first, m, pts.data());
	if(r != ADYPT_OK) return fail(c, r, adypt_host_last_error());
	float *padded = c->h_sobol[slot];
	memset(padded, 0, (size_t)m * 64 * sizeof(float));
	for(int k = 0; k < m; ++k) memcpy(&padded[(size_t)k * 64], &pts[(size_t)k * 2 * max_bounce], sizeof(float) * 2 * (size_t)max_bounce);
	HIP_TRY(c, hipMemcpyAsync(dst, padded, (size_t)m * 64 * sizeof(float), hipMemcpyHostToDevice, c->stream));
	HIP_TRY(c, hipEventRecord(c->sobol_done[slot], c->stream));
	return ADYPT_OK;
}

// the arguments of single frame `frame` in rolling slot `s`: a batch of one whose Sobol points, finished samples, queue window, counters and
// stream are the slot's
void roll_frame_args(const adypt_ctx *c, int frame, int s, FrameArgs *f)
{
	fill_frame(c, f);
	f->spp = frame; f->n_frames = 1; f->frame_first = 0; f->frame_stride = 1; f->batched = 1;
	f->sun_query = c->sun_visibility; // (a rolling frame always takes the one-launch pipeline: the escaped paths' queries travel with it)
	f->sobol = c->d_sobol + (size_t)s * 64;
	f->done = c->d_done + (size_t)s * (size_t)std::max(c->n_local_px, 64);
}

// Enqueues single frame `frame` in rolling slot `s`: [camera rays of a re-tracing frame ->] counters -> k_shade_first on the CONTEXT's stream (it
// reads the primary-hit cache, which the next re-tracing frame rewrites on that stream), then k_path on the slot's own stream behind an event.
// Nothing here waits for the slot's previous frame: the caller has enqueued that frame's running-mean step — which waits for its k_path — on the
// context's stream before it calls this.
int roll_launch(adypt_ctx *c, const SceneArgs &sc, const PixelArgs &px, bool stats, int frame, int s)
{
	const Pipe &pipe = c->pipes[1 + s];
	const QueueWindow win = pipe_window(c, s, 2);
	FrameArgs f;
	roll_frame_args(c, frame, s, &f);
	int r = upload_sobol(c, frame, 1, c->d_sobol + (size_t)s * 64);
	if(r != ADYPT_OK) return r;
	if(frame % std::max(1, c->params.tmp_lifetime) == 0)
	{
		// the frame re-traces its primary rays (pathtracer.glsl:113-127): one camera launch into the cache image, on the context's stream
		FrameArgs fc = f;
		fc.frame_stride = std::max(1, c->params.tmp_lifetime);
		r = launch_trace_camera(c, c->pipes[0], full_window(c), fc, px, 1, stats);
		if(r != ADYPT_OK) return r;
	}
	clear_counters(c, pipe.counters, 1, c->stream);
	hipEvent_t *stop = begin_timing(c, 1, c->stream);
	QueueArgs q = queue_args(c, win, 0, pipe.counters->count[0], pipe.counters->count[1], 1); // out = queue 1 = bounce 1's rays
	audit_before(c, q, c->stream, 1 + s);
	hipLaunchKernelGGL(k_shade_first, dim3((unsigned)(c->n_local_px / kShadeThreads)), dim3(kShadeThreads), 0, c->stream, f, sc, q, px, stats ? 1 : 0);
	audit_after(c, q, c->stream, 1 + s);
	end_timing(stop, c->stream);
	HIP_TRY(c, hipGetLastError());
	c->last_batch_fused = true;
	if(c->params.max_bounce > 1 || c->sun_visibility) // (with one bounce the queue still holds the sun-visibility queries of the paths that escaped at once)
	{
		HIP_TRY(c, hipEventRecord(c->roll_ready[s], c->stream));
		HIP_TRY(c, hipStreamWaitEvent(pipe.stream, c->roll_ready[s], 0));
		SceneArgs sc_ref = sc;
		if(c->d_ref_triangles) sc_ref.triangles = (const float4 *)c->d_ref_triangles;
		r = launch_path(c, pipe, win, 1, pipe.counters->count[1], pipe.counters->cursor[1], f, sc_ref, px, 1, stats);
		if(r != ADYPT_OK) return r;
	}
	HIP_TRY(c, hipEventRecord(pipe.done, pipe.stream));
	c->roll_frame[s] = frame;
	return ADYPT_OK;
}

// frame c->spp as a rolling single frame; `more` = the call wants the frame after it too
int trace_rolling_frame(adypt_ctx *c, const SceneArgs &sc, const PixelArgs &px, bool stats, bool more)
{
	const int frame = c->spp, s = frame & 1;
	// While frames come in order the slots hold nothing but `frame` (slot s: started ahead by the previous call) and `frame + 1` (slot s ^ 1); anything else
	// is waited for and forgotten first.
	if((c->roll_frame[s] >= 0 && c->roll_frame[s] != frame) || (c->roll_frame[s ^ 1] >= 0 && c->roll_frame[s ^ 1] != frame + 1)) drop_rolling(c);
	if(c->roll_frame[s] != frame)
	{
		const int r = roll_launch(c, sc, px, stats, frame, s);
		if(r != ADYPT_OK) { drop_rolling(c); return r; }
	}
	// The frame after it, when this call asks for it (or the caller switched look-ahead on and it belongs to the same tmpLifetime group, so that image 1
	// stays what frame-by-frame tracing leaves there): enqueued NOW, behind frame `frame`'s k_path — it fills the compute units as that launch's workgroups
	// end.  Its slot's previous frame (frame - 1) had its running-mean step enqueued by the previous call of this function.  Only while a frame is small
	// enough for the end of its launch to matter: at 4096 x 4096 (99 M rays, 14 ms per frame) the next frame's bounce 0 running beside the current k_path
	// costs the 3 % the launch's end is worth (6566 against 6777 Mrays/s, profiles/r5_ablations.txt 3).
	const int life = std::max(1, c->params.tmp_lifetime);
	const bool ahead = c->single_overlap && c->n_local_px <= kRollMaxPixels && (more || (c->lookahead && (frame + 1) % life != 0));
	if(ahead && c->roll_frame[s ^ 1] != frame + 1)
	{
		const int r = roll_launch(c, sc, px, stats, frame + 1, s ^ 1);
		if(r != ADYPT_OK) { drop_rolling(c); return r; }
	}
	// running mean of frame `frame` (pathtracer.glsl:224-226) once its k_path has ended
	HIP_TRY(c, hipStreamWaitEvent(c->stream, c->pipes[1 + s].done, 0));
	FrameArgs f;
	roll_frame_args(c, frame, s, &f);
	hipEvent_t *stop = begin_timing(c, 1, c->stream);
	hipLaunchKernelGGL(k_resolve, dim3((c->n_local_px + 255) / 256), dim3(256), 0, c->stream, f, sc, px, 0, 1);
	end_timing(stop, c->stream);
	HIP_TRY(c, hipGetLastError());
	c->roll_frame[s] = -1;
	c->batch_spp = frame; c->batch_frames = 1; c->cache_group = 0; c->ahead_pos = 1; c->ahead_count = 0;
	c->spp += 1;
	return ADYPT_OK;
}

}  // namespace

namespace adypt {

namespace { std::atomic<bool> g_test_hooks{false}; }
bool test_hooks_enabled() { return g_test_hooks.load(std::memory_order_acquire); }

// The one place of csrc/device that reads the environment (tunables.hpp has the table).
Tunables read_tunables()
{
	Tunables t;
	auto num = [](const char *name, long lo, long hi, long unset) -> long {
		const char *v = getenv(name);
		if(!v || !*v) return unset;
		char *end = nullptr;
		const long x = strtol(v, &end, 10);
		if(end == v) return unset;
		return std::max(lo, std::min(hi, x));
	};
	auto flag = [&](const char *name, int unset) -> int { return num(name, 0, 1 << 30, unset) != 0 ? 1 : 0; };
	t.frames_in_flight = (int)num("ADYPT_FRAMES_IN_FLIGHT", 1, kMaxFramesInFlight, 0);
	t.pipeline = (int)num("ADYPT_PIPELINE", 1, kMaxPipes, kDefaultPipes);
	t.fused_bounces = flag("ADYPT_FUSED_BOUNCES", 1); t.first_fused = flag("ADYPT_FIRST_FUSED", 1); t.single_fused = flag("ADYPT_SINGLE_FUSED", 1);
	t.gen_deal = flag("ADYPT_GEN_DEAL", 1); t.shade_bin = flag("ADYPT_SHADE_BIN", 0); t.single_overlap = flag("ADYPT_SINGLE_OVERLAP", 1);
	t.refill_min = (int)num("ADYPT_REFILL_MIN", 1, 64, 0); t.refill_min_primary = (int)num("ADYPT_REFILL_MIN_PRIMARY", 1, 64, 0);
	t.bite = (int)num("ADYPT_BITE", 1, 4096, 0); t.bite_primary = (int)num("ADYPT_BITE_PRIMARY", 1, 4096, 0);
	t.chunk = (int)num("ADYPT_CHUNK", 16, 4096, 0); t.endgame = (int)num("ADYPT_ENDGAME", 0, 1024, -1);
	t.shade_min = (int)num("ADYPT_SHADE_MIN", 1, 64, 0);
	t.rare_min = (int)num("ADYPT_RARE_MIN", 0, 64, -1); t.defer_max = (int)num("ADYPT_DEFER_MAX", 0, 64, -1);
	t.lds_stack_depth = (int)num("ADYPT_LDS_STACK_DEPTH", 1, kLdsStackMax, 0); t.trace_blocks_per_cu = (int)num("ADYPT_TRACE_BLOCKS_PER_CU", 1, 16, 0);
	t.path_blocks_per_cu = (int)num("ADYPT_PATH_BLOCKS_PER_CU", 1, 8, 0); t.path_lds_depth = (int)num("ADYPT_PATH_LDS_DEPTH", 1, kLdsStackMax, 0);
	t.path_verbose = flag("ADYPT_PATH_VERBOSE", 0);
	t.ref_triangles_max_mb = num("ADYPT_REF_TRIANGLES_MAX_MB", 0, 1 << 20, -1);
	if(const char *v = getenv("ADYPT_RCCL_LIB")) t.rccl_lib = v;
	if(const char *v = getenv("ADYPT_GATHER_TIMEOUT"))
	{
		// (a value that is not a number leaves the default in place: atof would read "abc" as 0 = no watchdog)
		char *end = nullptr;
		const double x = strtod(v, &end);
		if(end != v && *end == '\0' && x >= 0.0 && x <= 86400.0) t.gather_timeout_s = x;
		else fprintf(stderr, "[adypt] ADYPT_GATHER_TIMEOUT=\"%s\" is not a number of seconds in [0, 86400]: keeping %g s\n", v, t.gather_timeout_s);
	}
	if(test_hooks_enabled())
	{
		t.multi_shared_device = flag("ADYPT_MULTI_SHARED_DEVICE", 0) != 0;
		t.audit_selftest = flag("ADYPT_AUDIT_SELFTEST", 0) != 0;
		t.gather_stall_test = flag("ADYPT_GATHER_STALL_TEST", 0) != 0;
		if(const char *v = getenv("ADYPT_COMM_TRANSPORT")) t.comm_transport_host = !strcmp(v, "host");
		if(const char *v = getenv("ADYPT_HOST_TRANSPORT_TIMEOUT")) t.host_transport_timeout_s = std::max(0.1, atof(v));
	}
	return t;
}

CtxInfo ctx_info(adypt_ctx *c)
{
	CtxInfo i;
	i.device = c->device; i.stream = c->stream; i.rank = c->rank; i.nranks = c->nranks; i.width = c->width; i.height = c->height;
	i.n_local_px = c->n_local_px; i.accum = c->d_accum;
	return i;
}
void ctx_set_error(adypt_ctx *c, const std::string &msg) { c->error = msg; }
void **ctx_comm_slot(adypt_ctx *c, void (***free_fn)(void *)) { *free_fn = &c->comm_free; return &c->comm; }
}  // namespace adypt

extern "C" {

int adypt_abi_version(void) { return ADYPT_ABI_VERSION; }

int adypt_enable_test_hooks(uint64_t magic)
{
	if(magic != ADYPT_TEST_HOOKS_MAGIC) return ADYPT_E_INVALID;
	g_test_hooks.store(true, std::memory_order_release);
	return ADYPT_OK;
}
int adypt_test_hooks_enabled(void) { return test_hooks_enabled() ? 1 : 0; }

const char *adypt_last_error(const adypt_ctx *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int64_t adypt_shard_block_count(int width, int height, int rank, int nranks)
{
	if(width <= 0 || height <= 0 || nranks <= 0 || rank < 0 || rank >= nranks) return -1;
	return (int64_t)owned_blocks(width, height, rank, nranks).size();
}

int adypt_untile_host(int width, int height, int rank, int nranks, const float *local_rgba, float *rgb)
{
	if(width <= 0 || height <= 0 || nranks <= 0 || rank < 0 || rank >= nranks || !local_rgba || !rgb) return ADYPT_E_INVALID;
	const std::vector<int32_t> blocks = owned_blocks(width, height, rank, nranks);
	const int nbx = (width + kBlockDim - 1) / kBlockDim;
	for(size_t bi = 0; bi < blocks.size(); ++bi)
		for(int in = 0; in < kBlockPixels; ++in)
		{
			const int wt = in >> 6, ln = in & 63;
			const int x = (blocks[bi] % nbx) * kBlockDim + (wt & 3) * 8 + (ln & 7), y = (blocks[bi] / nbx) * kBlockDim + (wt >> 2) * 8 + (ln >> 3);
			if(x >= width || y >= height) continue;
			const float *s = local_rgba + (bi * kBlockPixels + (size_t)in) * 4;
			float *o = rgb + ((size_t)y * width + x) * 3;
			o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
		}
	return ADYPT_OK;
}

int adypt_create(adypt_ctx **out, const adypt_scene_desc *d)
{
	g_create_error.clear();
	if(!out || !d) { g_create_error = "adypt_create: null argument"; return ADYPT_E_INVALID; }
	*out = nullptr;
	if(!d->nodes || !d->tri_indices || !d->triangles || d->n_nodes <= 0 || d->n_refs < 0 || d->n_tris <= 0 || d->n_mats < 0 ||
	   d->width <= 0 || d->height <= 0 || d->n_textures < 0 || (d->n_mats > 0 && !d->materials) || (d->n_textures > 0 && !d->textures) ||
	   d->tile_nranks <= 0 || d->tile_rank < 0 || d->tile_rank >= d->tile_nranks)
	{ g_create_error = "adypt_create: inconsistent scene description"; return ADYPT_E_INVALID; }
	if((int64_t)d->width * d->height > (int64_t)1 << 30) { g_create_error = "adypt_create: image too large"; return ADYPT_E_INVALID; }
	{
		std::string why;
		if(!validate_bvh(*d, &why)) { g_create_error = "adypt_create: invalid BVH arrays: " + why; return ADYPT_E_INVALID; }
	}
	int n_dev = 0;
	if(hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { g_create_error = "adypt_create: no HIP device available (this library has no CPU path)"; return ADYPT_E_NO_DEVICE; }
	if(d->device < 0 || d->device >= n_dev) { g_create_error = "adypt_create: device ordinal out of range"; return ADYPT_E_INVALID; }

	adypt_ctx *c = new adypt_ctx();
	auto bail = [&](int code) { g_create_error = c->error; adypt_destroy(c); return code; };
	c->device = d->device;
	int r;
#define TRY_CREATE(expr) do { r = (expr); if(r != ADYPT_OK) return bail(r); } while(0)
#define HIP_CREATE(expr) do { hipError_t e_ = (expr); if(e_ != hipSuccess) { c->error = std::string(#expr) + ": " + hipGetErrorString(e_); return bail(e_ == hipErrorOutOfMemory ? ADYPT_E_OOM : ADYPT_E_HIP); } } while(0)
	HIP_CREATE(hipSetDevice(c->device));
	HIP_CREATE(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	c->pipes[0].stream = c->stream;
	for(int k = 1; k < kMaxPipes; ++k) HIP_CREATE(hipStreamCreateWithFlags(&c->pipes[k].stream, hipStreamNonBlocking));
	for(int k = 0; k < kMaxPipes; ++k) HIP_CREATE(hipEventCreateWithFlags(&c->pipes[k].done, hipEventDisableTiming));
	HIP_CREATE(hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));
	for(int s = 0; s < 2; ++s) HIP_CREATE(hipEventCreateWithFlags(&c->roll_ready[s], hipEventDisableTiming));
	c->tun = read_tunables();
	c->pipeline = c->tun.pipeline;
	hipDeviceProp_t prop;
	HIP_CREATE(hipGetDeviceProperties(&prop, c->device));
	c->num_cus = prop.multiProcessorCount;
	if(prop.maxSharedMemoryPerMultiProcessor >= 64 * 1024) c->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
	{
		const Tunables &t = c->tun;
		if(t.refill_min > 0) c->refill_min = c->refill_min_primary = (uint32_t)t.refill_min;
		if(t.refill_min_primary > 0) c->refill_min_primary = (uint32_t)t.refill_min_primary;
		c->deal_chunks = t.gen_deal; c->first_fused = t.first_fused; c->fused_bounces = t.fused_bounces; c->single_fused = t.single_fused;
		c->audit_selftest = t.audit_selftest; c->single_overlap = t.single_overlap;
		if(t.shade_min > 0) c->shade_min = (uint32_t)t.shade_min;
		if(t.rare_min >= 0) c->rare_min = (uint32_t)t.rare_min;
		if(t.defer_max >= 0) c->defer_max = (uint32_t)t.defer_max;
		if(t.chunk > 0) c->chunk = (uint32_t)t.chunk;
		if(t.endgame >= 0) c->endgame = (uint32_t)t.endgame;
		if(t.bite > 0) c->bite = c->bite_primary = (uint32_t)t.bite;
		if(t.bite_primary > 0) c->bite_primary = (uint32_t)t.bite_primary;
	}

	c->n_nodes = d->n_nodes; c->n_refs = d->n_refs; c->n_tris = d->n_tris; c->n_mats = d->n_mats; c->n_tex = d->n_textures;
	c->width = d->width; c->height = d->height; c->rank = d->tile_rank; c->nranks = d->tile_nranks;
	c->blocks_x = (c->width + kBlockDim - 1) / kBlockDim; c->blocks_y = (c->height + kBlockDim - 1) / kBlockDim;
	c->local_blocks = owned_blocks(c->width, c->height, c->rank, c->nranks);
	c->n_local_blocks = (int)c->local_blocks.size();
	c->n_local_px = c->n_local_blocks * kBlockPixels;
	c->n_image_px = 0;
	for(int32_t blk : c->local_blocks)
	{
		const int bx = blk % c->blocks_x, by = blk / c->blocks_x;
		c->n_image_px += (int64_t)std::min(kBlockDim, c->width - bx * kBlockDim) * (int64_t)std::min(kBlockDim, c->height - by * kBlockDim);
	}

	TRY_CREATE(upload(c, &c->d_nodes, (const uint8_t *)d->nodes, (size_t)d->n_nodes * 80));
	TRY_CREATE(upload(c, &c->d_tri_indices, d->tri_indices, (size_t)d->n_refs));
	{
		std::vector<float> woop;
		const float *wp = d->woop;
		if(!wp) { woop.resize((size_t)d->n_refs * 12); adypt_woop_matrices(d->triangles, d->tri_indices, d->n_refs, woop.data()); wp = woop.data(); }
		TRY_CREATE(upload(c, &c->d_woop, wp, (size_t)d->n_refs * 12));
	}
	{
		// 100-byte Triangle -> 112-byte device record (shade.hpp): [p n matid pad] + [tc pad]
		std::vector<float> packed((size_t)d->n_tris * kTriFloat4 * 4, 0.0f);
		const uint8_t *src = (const uint8_t *)d->triangles;
		for(int64_t i = 0; i < d->n_tris; ++i)
		{
			float *o = packed.data() + (size_t)i * kTriFloat4 * 4;
			memcpy(o, src + i * 100, 72);            // positions + normals
			memcpy(o + 18, src + i * 100 + 96, 4);   // material id
			// class word (shade.hpp): 1 = a hit here runs the glossy lobe or the dielectric branch of Render() — what k_path's shading rounds defer to a
			// round of their own (path.hpp).  A grouping hint only: never an input of the arithmetic.
			int32_t matid; memcpy(&matid, src + i * 100 + 96, 4);
			if(matid >= 0 && matid < d->n_mats)
			{
				const uint8_t *mat = (const uint8_t *)d->materials + (size_t)matid * 64;
				int32_t dtex, illum; float shininess;
				memcpy(&dtex, mat, 4); memcpy(&illum, mat + 48, 4); memcpy(&shininess, mat + 52, 4);
				const uint32_t cls = material_class(illum, shininess, false), word = (cls == 3u || cls == 6u) ? 1u : 0u;
				memcpy(o + 19, &word, 4);
			}
			memcpy(o + 20, src + i * 100 + 72, 24);  // texture coordinates
		}
		TRY_CREATE(upload(c, &c->d_triangles, packed.data(), packed.size()));
	}
	{
		// k_shade's sort key per triangle (shade.hpp: material_class).  Off unless ADYPT_SHADE_BIN=1: measured +10 % k_shade time on both
		// bench scenes (profiles/r3_ablations_k_trace.txt item 9) — the kernel waits on its gathers, not on divergent vector-ALU work
		if(c->tun.shade_bin)
		{
			std::vector<uint8_t> cls((size_t)std::max<int64_t>(d->n_tris, 1), (uint8_t)5);
			const uint8_t *tri = (const uint8_t *)d->triangles, *mat = (const uint8_t *)d->materials;
			for(int64_t i = 0; i < d->n_tris; ++i)
			{
				int32_t matid, dtex, illum; float shininess;
				memcpy(&matid, tri + i * 100 + 96, 4);
				if(matid < 0 || matid >= d->n_mats) continue;
				memcpy(&dtex, mat + (size_t)matid * 64, 4); memcpy(&illum, mat + (size_t)matid * 64 + 48, 4); memcpy(&shininess, mat + (size_t)matid * 64 + 52, 4);
				cls[(size_t)i] = (uint8_t)material_class(illum, shininess, d->n_textures != 0 && dtex >= 0 && dtex < d->n_textures);
			}
			TRY_CREATE(upload(c, &c->d_tri_class, cls.data(), cls.size()));
		}
	}
	{
		// textures: RGB8 -> RGBA8 words, every row w + 1 texels long — the extra one repeats the row's first texel, so the horizontal
		// neighbour of the last column (GL_REPEAT) sits next to it and sample_texture fetches a row's two texels in one 8-byte load
		std::vector<uint32_t> texels;
		std::vector<int32_t> desc;
		for(int t = 0; t < d->n_textures; ++t)
		{
			const adypt_texture &tx = d->textures[t];
			if(tx.width <= 0 || tx.height <= 0 || !tx.rgb) { c->error = "adypt_create: bad texture " + std::to_string(t); return bail(ADYPT_E_INVALID); }
			desc.push_back((int32_t)texels.size()); desc.push_back(tx.width); desc.push_back(tx.height); desc.push_back(0);
			const size_t base = texels.size(), row = (size_t)tx.width + 1;
			if(base + row * (size_t)tx.height >= ((size_t)1 << 31)) { c->error = "adypt_create: more than 2^31 texels"; return bail(ADYPT_E_INVALID); }
			texels.resize(base + row * (size_t)tx.height);
			for(int y = 0; y < tx.height; ++y)
			{
				uint32_t *o = texels.data() + base + row * (size_t)y;
				const uint8_t *in = tx.rgb + (size_t)y * tx.width * 3;
				for(int x = 0; x < tx.width; ++x) o[x] = (uint32_t)in[x * 3] | (uint32_t)in[x * 3 + 1] << 8 | (uint32_t)in[x * 3 + 2] << 16 | 0xff000000u;
				o[tx.width] = o[0];
			}
		}
		TRY_CREATE(upload(c, &c->d_texels, texels.data(), texels.size()));
		// materials: the reference's 64 bytes + the descriptor of the diffuse texture (one fetch less per textured hit)
		std::vector<uint8_t> mats((size_t)std::max<int64_t>(d->n_mats, 1) * kMatFloat4 * 16, 0);
		for(int64_t m = 0; m < d->n_mats; ++m)
		{
			uint8_t *o = mats.data() + (size_t)m * kMatFloat4 * 16;
			memcpy(o, (const uint8_t *)d->materials + (size_t)m * 64, 64);
			int32_t dtex; memcpy(&dtex, o, 4);
			if(dtex >= 0 && dtex < d->n_textures) memcpy(o + 64, desc.data() + (size_t)dtex * 4, 16);
		}
		TRY_CREATE(upload(c, &c->d_materials, mats.data(), mats.size()));
	}
	TRY_CREATE(upload(c, &c->d_local_blocks, c->local_blocks.data(), c->local_blocks.size()));
	{
		// k_path looks a hit's triangle up by reference index in a second copy of the records (path.hpp): made here, once, so that nothing is
		// allocated while frames are traced.  Above the size threshold, or when the memory cannot be had, k_path applies the 4-byte
		// uTriIndices remap (traversal.glsl:253-254) in its shading round instead — same image.
		const size_t n16 = (size_t)c->n_refs * kTriFloat4, bytes = std::max<size_t>(n16, 1) * sizeof(float4);
		const long max_mb = c->tun.ref_triangles_max_mb >= 0 ? c->tun.ref_triangles_max_mb : kRefTrianglesAutoMaxMB;
		if(max_mb != 0 && (bytes >> 20) <= (size_t)max_mb) // (0 = never, whatever the size: the tests' way into the remap path with scenes of a few triangles)
		{
			if(hipMalloc(&c->d_ref_triangles, bytes) != hipSuccess) { c->d_ref_triangles = nullptr; (void)hipGetLastError(); }
			else if(n16)
			{
				hipLaunchKernelGGL(k_expand_references, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, c->stream, (const float4 *)c->d_triangles, (const int32_t *)c->d_tri_indices, (size_t)c->n_refs, (float4 *)c->d_ref_triangles);
				HIP_CREATE(hipGetLastError());
			}
		}
	}

	const size_t npx = (size_t)std::max(c->n_local_px, 64);
	HIP_CREATE(hipMalloc((void **)&c->d_accum, npx * sizeof(float4)));
	HIP_CREATE(hipMalloc((void **)&c->d_cache, npx * sizeof(float4)));
	HIP_CREATE(hipMalloc((void **)&c->d_shift, npx * 2));
	HIP_CREATE(hipMemset(c->d_accum, 0, npx * sizeof(float4)));
	HIP_CREATE(hipMemset(c->d_cache, 0xff, npx * sizeof(float4)));
	HIP_CREATE(hipMemset(c->d_shift, 0, npx * 2));
	{
		// frames in flight: enough consecutive frames per wavefront pass to keep ~64 Mi paths in flight (32 frames of a
		// 1080p image, 128 frames = the maximum for the 260 k-pixel tile shard of an 8-GPU run): the drain of a persistent
		// launch (its longest rays) is amortised over more work; ADYPT_FRAMES_IN_FLIGHT overrides
		int fif = (int)std::min<size_t>(kMaxFramesInFlight, std::max<size_t>(1, ((size_t)64 << 20) / npx));
		if(c->tun.frames_in_flight > 0) fif = std::min(kMaxFramesInFlight, c->tun.frames_in_flight);
		TRY_CREATE(alloc_queues(c, fif));
	}
	HIP_CREATE(hipMalloc((void **)&c->d_sobol, (size_t)kMaxFramesInFlight * 64 * sizeof(float)));
	for(int i = 0; i < adypt_ctx::kSobolSlots; ++i) // allocated here, not lazily: nothing is allocated while frames are traced
	{
		HIP_CREATE(hipHostMalloc((void **)&c->h_sobol[i], (size_t)kMaxFramesInFlight * 64 * sizeof(float), hipHostMallocDefault));
		HIP_CREATE(hipEventCreateWithFlags(&c->sobol_done[i], hipEventDisableTiming));
	}
	HIP_CREATE(hipMalloc((void **)&c->d_counters, sizeof(FrameCounters) * kMaxPipes));
	for(int k = 0; k < kMaxPipes; ++k) c->pipes[k].counters = c->d_counters + k;
	HIP_CREATE(hipMalloc((void **)&c->d_stats, sizeof(DeviceStats)));
	HIP_CREATE(hipMemset(c->d_counters, 0, sizeof(FrameCounters) * kMaxPipes));
	HIP_CREATE(hipMalloc((void **)&c->d_camera_cursors, sizeof(uint32_t) * (2 * kNumSegments + 1) * kCursorStride));
	HIP_CREATE(hipMemset(c->d_camera_cursors, 0, sizeof(uint32_t) * (2 * kNumSegments + 1) * kCursorStride));
	HIP_CREATE(hipMemset(c->d_stats, 0, sizeof(DeviceStats)));
	HIP_CREATE(hipHostMalloc((void **)&c->h_overflow, sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
	*c->h_overflow = 0u;
	HIP_CREATE(hipMemcpy(&c->d_stats->host_overflow, &c->h_overflow, sizeof(uint32_t *), hipMemcpyHostToDevice));

	// defaults of InstanceConfig::PT (src/InstanceConfig.hpp:21-27), seed 0
	c->pending.stack_size = 12; c->pending.max_bounce = 5; c->pending.subpixel = 8; c->pending.tmp_lifetime = 16;
	c->pending.ray_tmin = 0.0001f; c->pending.clamp = 4.0f; c->pending.sun[0] = c->pending.sun[1] = c->pending.sun[2] = 0.0f;
	c->pending.shift_seed = 0;
	TRY_CREATE(apply_params(c));
	HIP_CREATE(hipDeviceSynchronize()); // the uploads / memsets above ran on the legacy stream
	HIP_CREATE(hipStreamSynchronize(c->stream));
#undef TRY_CREATE
#undef HIP_CREATE
	*out = c;
	return ADYPT_OK;
}

void adypt_destroy(adypt_ctx *c)
{
	if(!c) return;
	(void)hipSetDevice(c->device);
	for(int k = 0; k < kMaxPipes; ++k) if(c->pipes[k].stream) (void)hipStreamSynchronize(c->pipes[k].stream);
	if(c->comm && c->comm_free) c->comm_free(c->comm);
	c->comm = nullptr;
	for(EventPair &p : c->events) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
	for(EventPair &p : c->free_events) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
	void *bufs[] = {c->sh_o, c->sh_d, c->sh_col, c->sh_hit, c->d_all_blocks, c->d_nodes, c->d_woop, c->d_tri_indices, c->d_triangles, c->d_materials, c->d_tri_class, c->d_texels, c->d_ref_triangles, c->d_local_blocks,
					c->d_accum, c->d_cache, c->d_cache_next, c->d_shift, c->q_o[0], c->q_o[1], c->q_d[0], c->q_d[1], c->q_col[0], c->q_col[1],
					c->d_hit, c->d_ray_stats, c->d_audit_seen, c->d_counters, c->d_camera_cursors, c->d_stats, c->d_spill, c->d_done, c->d_sobol, c->d_display};
	for(void *b : bufs) if(b) (void)hipFree(b);
	for(int i = 0; i < adypt_ctx::kSobolSlots; ++i)
	{
		if(c->h_sobol[i]) (void)hipHostFree(c->h_sobol[i]);
		if(c->sobol_done[i]) (void)hipEventDestroy(c->sobol_done[i]);
	}
	for(int k = 0; k < kMaxPipes; ++k)
	{
		if(c->pipes[k].done) (void)hipEventDestroy(c->pipes[k].done);
		if(k > 0 && c->pipes[k].stream) (void)hipStreamDestroy(c->pipes[k].stream);
	}
	if(c->h_overflow) (void)hipHostFree(c->h_overflow);
	if(c->fork_ev) (void)hipEventDestroy(c->fork_ev);
	for(int s = 0; s < 2; ++s) if(c->roll_ready[s]) (void)hipEventDestroy(c->roll_ready[s]);
	if(c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
}

int adypt_set_params(adypt_ctx *c, const adypt_pt_params *p)
{
	if(!c || !p) return ADYPT_E_INVALID;
	// subpixel * subpixel is an int in the kernels (pathtracer.glsl:207 does the same in GLSL int): 46340^2 < 2^31
	if(p->stack_size < 1 || p->stack_size > 64 || p->max_bounce < 1 || p->max_bounce > kMaxBounce || p->subpixel < 1 || p->subpixel > 46340 || p->tmp_lifetime < 1)
		return fail(c, ADYPT_E_INVALID, "adypt_set_params: stackSize must be in [1,64], maxBounce in [1,32], subpixel in [1,46340], tmpLifetime >= 1");
	c->pending = *p;
	c->have_params = true;
	if(!c->pt_started)
	{
		HIP_TRY(c, hipSetDevice(c->device));
		return apply_params(c);
	}
	return ADYPT_OK;
}

int adypt_set_camera(adypt_ctx *c, const float origin[3], const float inv_proj[16], const float inv_view[16])
{
	if(!c || !origin || !inv_proj || !inv_view) return ADYPT_E_INVALID;
	memcpy(c->origin, origin, 12); memcpy(c->inv_proj, inv_proj, 64); memcpy(c->inv_view, inv_view, 64);
	c->have_camera = true;
	drop_lookahead(c); // frames traced ahead saw the previous camera
	return ADYPT_OK;
}

int adypt_reset(adypt_ctx *c)
{
	if(!c) return ADYPT_E_INVALID;
	c->pt_started = false;
	c->spp = 0;
	drop_lookahead(c);
	return ADYPT_OK;
}

int adypt_set_lookahead(adypt_ctx *c, int enabled)
{
	if(!c) return ADYPT_E_INVALID;
	c->lookahead = enabled ? 1 : 0;
	if(!enabled) drop_lookahead(c);
	return ADYPT_OK;
}

int adypt_get_lookahead_frames(const adypt_ctx *c) { return c ? c->ahead_count : ADYPT_E_INVALID; }

int adypt_get_spp(const adypt_ctx *c) { return c ? c->spp : ADYPT_E_INVALID; }

int adypt_set_instrumentation(adypt_ctx *c, int flags)
{
	if(!c) return ADYPT_E_INVALID;
	c->instrumentation = flags;
	if(flags & 4) { HIP_TRY(c, hipSetDevice(c->device)); int r = ensure_audit(c); if(r != ADYPT_OK) return r; }
	if(flags & 1)
	{
		// a pool of event pairs for the kernel timing, created here rather than while frames are being traced
		HIP_TRY(c, hipSetDevice(c->device));
		while(c->free_events.size() + c->events.size() < 96)
		{
			EventPair p;
			p.kind = 0;
			HIP_TRY(c, hipEventCreate(&p.a));
			HIP_TRY(c, hipEventCreate(&p.b));
			c->free_events.push_back(p);
		}
	}
	return ADYPT_OK;
}

int adypt_set_sun_visibility(adypt_ctx *c, int enabled, const float dir[3])
{
	if(!c) return ADYPT_E_INVALID;
	float d[3] = {0.6f, 1.0f, 0.2f}; // the direction of the reference's commented-out query (pathtracer.glsl:132)
	if(dir) memcpy(d, dir, sizeof(d));
	const float len2 = fmaf(d[2], d[2], fmaf(d[1], d[1], d[0] * d[0]));
	if(!(len2 > 0.0f) || !(len2 < INFINITY)) return fail(c, ADYPT_E_INVALID, "adypt_set_sun_visibility: direction must be finite and non-zero");
	const float inv = 1.0f / sqrtf(len2); // normalize() in the canonical arithmetic (canon_math.hpp normalize3)
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	c->sun_dir[0] = d[0] * inv; c->sun_dir[1] = d[1] * inv; c->sun_dir[2] = d[2] * inv;
	c->sun_visibility = enabled ? 1 : 0;
	drop_lookahead(c);
	return ADYPT_OK;
}

int adypt_set_frames_in_flight(adypt_ctx *c, int n)
{
	if(!c) return ADYPT_E_INVALID;
	if(n < 1 || n > kMaxFramesInFlight) return fail(c, ADYPT_E_INVALID, "adypt_set_frames_in_flight: n_frames must be in [1, " + std::to_string(kMaxFramesInFlight) + "]");
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	if(n == c->frames_in_flight && c->queues_ok) return ADYPT_OK;
	drop_lookahead(c); // the parked samples live in the buffers about to be reallocated
	int r = alloc_queues(c, n);
	if(r == ADYPT_OK && (c->instrumentation & 4)) r = ensure_audit(c);
	return r;
}

int adypt_get_frames_in_flight(const adypt_ctx *c) { return c ? c->frames_in_flight : ADYPT_E_INVALID; }

int adypt_set_pipeline(adypt_ctx *c, int n_pipes)
{
	if(!c) return ADYPT_E_INVALID;
	if(n_pipes < 1 || n_pipes > kMaxPipes) return fail(c, ADYPT_E_INVALID, "adypt_set_pipeline: n_pipes must be in [1, " + std::to_string(kMaxPipes) + "]");
	c->pipeline = n_pipes; // takes effect with the next batch; nothing in flight depends on it
	return ADYPT_OK;
}

int adypt_get_pipeline(const adypt_ctx *c) { return c ? c->pipeline : ADYPT_E_INVALID; }

int adypt_set_fused_bounces(adypt_ctx *c, int enabled)
{
	if(!c) return ADYPT_E_INVALID;
	c->fused_bounces = enabled ? 1 : 0; // takes effect with the next batch; both pipelines leave the same image and the same state behind
	return ADYPT_OK;
}

int adypt_get_fused_bounces(const adypt_ctx *c) { return c ? (c->last_batch_fused ? 1 : 0) : ADYPT_E_INVALID; }

int adypt_trace_primary(adypt_ctx *c, int viewer_type)
{
	if(!c) return ADYPT_E_INVALID;
	if(!c->have_camera) return fail(c, ADYPT_E_STATE, "adypt_trace_primary: call adypt_set_camera first");
	if(!c->queues_ok) return fail(c, ADYPT_E_STATE, "adypt_trace_primary: the context lost its ray queues (failed adypt_set_frames_in_flight)");
	HIP_TRY(c, hipSetDevice(c->device));
	// Trace(false): leaves path-tracing mode (OglPathTracer.cpp:53-58)
	c->pt_started = false; c->spp = 0;
	drop_lookahead(c);
	c->view_type = viewer_type;
	int r = apply_params(c);
	if(r != ADYPT_OK) return r;
	if(c->n_local_px == 0) return ADYPT_OK; // a tile shard that owns no 32x32 block (more ranks than block diagonals): nothing to render
	FrameArgs f; SceneArgs sc; PixelArgs px;
	fill_frame(c, &f); fill_scene(c, &sc); fill_pixels(c, &px);
	const Pipe &pipe = c->pipes[0];
	const QueueWindow win = full_window(c);
	// camera rays -> traversal -> cache image and the viewer's colour of every pixel (primaryray.glsl:46-94): one launch is the whole call
	r = launch_trace_camera(c, pipe, win, f, px, 0, (c->instrumentation & 2) != 0, viewer_type);
	if(r != ADYPT_OK) return r;
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	harvest_events(c);
	return check_async_errors(c);
}

int adypt_wait(adypt_ctx *c)
{
	if(!c) return ADYPT_E_INVALID;
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	harvest_events(c);
	return check_async_errors(c);
}

int adypt_trace_spp(adypt_ctx *c, int n_spp)
{
	int r = adypt_trace_spp_async(c, n_spp);
	return r != ADYPT_OK ? r : adypt_wait(c);
}

int adypt_trace_spp_async(adypt_ctx *c, int n_spp)
{
	if(!c || n_spp < 0) return ADYPT_E_INVALID;
	if(!c->have_camera) return fail(c, ADYPT_E_STATE, "adypt_trace_spp: call adypt_set_camera first");
	if(!c->queues_ok) return fail(c, ADYPT_E_STATE, "adypt_trace_spp: the context lost its ray queues (failed adypt_set_frames_in_flight)");
	HIP_TRY(c, hipSetDevice(c->device));
	if(c->n_local_px == 0)
	{
		// a tile shard that owns no 32x32 block (more ranks than block diagonals, e.g. 64x36 on 4 ranks): the frame counter
		// and the parameter hand-over advance like everywhere else, no kernel runs (they would divide by n_local_px)
		if(n_spp > 0 && !c->pt_started)
		{
			int r = apply_params(c);
			if(r != ADYPT_OK) return r;
			c->spp = 0; c->pt_started = true; c->view_type = 3;
		}
		c->spp += n_spp;
		return ADYPT_OK;
	}
	SceneArgs sc; PixelArgs px;
	fill_scene(c, &sc); fill_pixels(c, &px);
	const bool stats = (c->instrumentation & 2) != 0;
	for(int remaining = n_spp; remaining > 0;)
	{
		if(c->ahead_count > 0)
		{
			// frames already traced ahead by an earlier call: only their running-mean step is left (frame order is kept)
			const int k = std::min(remaining, c->ahead_count);
			int r = resolve_batch_frames(c, sc, px, c->ahead_pos, k);
			if(r != ADYPT_OK) return r;
			c->ahead_pos += k; c->ahead_count -= k; c->spp += k; remaining -= k;
			continue;
		}
		if(!c->pt_started)
		{
			// first path-traced frame (OglPathTracer.cpp:39-46): apply config, clear the result image, restart Sobol
			int r = apply_params(c);
			if(r != ADYPT_OK) return r;
			HIP_TRY(c, hipMemsetAsync(c->d_accum, 0, (size_t)std::max(c->n_local_px, 64) * sizeof(float4), c->stream));
			c->spp = 0;
			c->pt_started = true;
			c->view_type = 3; // kPTRadiance (OglPathTracer.cpp:38)
		}
		const int max_bounce = c->params.max_bounce, life = c->params.tmp_lifetime;
		// Batch = up to frames_in_flight consecutive frames traced as ONE wavefront (frames are independent samples; the
		// running mean is applied afterwards in frame order, so the result is bit-identical to frame-by-frame).  A batch
		// may span several tmpLifetime groups: the frames that re-trace their primary rays (spp % tmpLifetime == 0) run
		// first, as one primary-only pass, and park their hits in the cache image of their group.
		// With look-ahead on, a call for fewer frames than fit in a pass (Instance::Update asks for ONE, src/Instance.cpp:44-57)
		// still traces a full pass: frames are independent samples of a deterministic sequence, so the frames beyond the ones
		// asked for are simply finished early and parked; later calls hand them out one running-mean step at a time.
		const int m = c->lookahead ? c->frames_in_flight : std::min(remaining, c->frames_in_flight);
		const int hand_out = std::min(remaining, m);
		// One frame per pass through the one-launch pipeline: a rolling single frame (frame k + 1 is enqueued under the end of frame k's k_path)
		// (the sun-visibility query rides in k_path as bounce index kPwShadow = 31: with 32 bounces configured the launch-per-bounce pipeline keeps it)
		const bool sun_ok = !c->sun_visibility || max_bounce <= (int)kPwShadow;
		if(m == 1 && c->single_fused && c->first_fused && c->fused_bounces && sun_ok && (int64_t)c->n_local_px <= kPathMaxPaths)
		{
			int r = trace_rolling_frame(c, sc, px, stats, remaining > 1);
			if(r != ADYPT_OK) return r;
			remaining -= 1;
			continue;
		}
		drop_rolling(c); // (a batch works in the whole queues)
		const int first_retrace = (life - c->spp % life) % life;                       // batch index of the first re-tracing frame
		const int n_retrace = first_retrace < m ? (m - 1 - first_retrace) / life + 1 : 0;
		const int n_groups = (c->spp + m - 1) / life - c->spp / life + 1;
		if(m > 1 && n_groups > 1)
		{
			int r = ensure_cache_slices(c, n_groups - 1);
			if(r != ADYPT_OK) return r;
			fill_pixels(c, &px);
		}
		FrameArgs f;
		fill_frame(c, &f);
		{ int r = upload_sobol(c, c->spp, m, c->d_sobol); if(r != ADYPT_OK) return r; } // Sobol::Next (src/Util/Sobol.cpp:16-21) for the m frames of the batch
		// A single frame (no look-ahead, or one frame in flight) runs as a batch of one — camera launch, k_shade_first, k_path, k_resolve: 4 launches
		// instead of 1 + 2 x maxBounce — whenever a batch would take the one-launch pipeline (ADYPT_SINGLE_FUSED=0: the launch-per-bounce frame)
		const bool as_batch = m > 1 || (c->single_fused && c->first_fused && c->fused_bounces && sun_ok && (int64_t)c->n_local_px <= kPathMaxPaths);
		const int use_cache = (!as_batch && n_retrace) ? 0 : 1;
		f.batched = as_batch ? 1 : 0;
		if(as_batch && n_retrace)
		{
			// primary-only pass of the re-tracing frames: camera rays -> traversal -> cache image of each frame's group
			// (on the context's stream, in the whole queue: every sub-batch below starts from these cache images)
			const Pipe &pipe = c->pipes[0];
			const QueueWindow win = full_window(c);
			f.n_frames = n_retrace; f.frame_first = first_retrace; f.frame_stride = life;
			int r = launch_trace_camera(c, pipe, win, f, px, 1, stats);
			if(r != ADYPT_OK) return r;
			f.frame_stride = 1;
		}
		// The main pass, cut into n_pipes sub-batches of consecutive frames; sub-batch k = the chain gen -> [trace -> shade] x
		// maxBounce on pipe k's stream in window k of the queues.  Everything before this point (Sobol upload, primary-only
		// pass, the previous batch's k_resolve) is ordered before every chain by the fork event, every chain before k_resolve.
		const int n_pipes = m > 1 ? std::max(1, std::min(std::min(c->pipeline, kMaxPipes), m)) : 1;
		// batches start every frame from a cached primary hit: camera rays and bounce 0 in one kernel (k_shade_first).  With the sun-visibility query on, only when
		// k_path follows (it traces the queries k_shade_first emits for the paths that escape at once); else the launch-per-bounce pipeline and its query queue
		const bool path_ok = n_pipes == 1 && c->fused_bounces && (int64_t)m * (int64_t)c->n_local_px <= kPathMaxPaths;
		const bool fused_first = as_batch && use_cache && c->first_fused && (!c->sun_visibility || (path_ok && sun_ok));
		f.sun_query = (c->sun_visibility && fused_first) ? 1 : 0;
		if(c->sun_visibility && !f.sun_query) { int r = ensure_shadow_queue(c); if(r != ADYPT_OK) return r; }
		// the counters of all pipes are contiguous: one clearing launch, on the context's stream, before the chains fork
		clear_counters(c, c->d_counters, n_pipes, c->stream);
		// a launch or HIP call that fails between the fork and the join must not leave the other chains running unjoined: what follows on the
		// context's stream (or the caller's next call) only synchronises c->stream, and those chains would still be writing queues, done[] and
		// counters.  Every early return from here to the join goes through abandon().
		auto abandon = [&](int code) { for(int k = 1; k < kMaxPipes; ++k) (void)hipStreamSynchronize(c->pipes[k].stream); return code; };
#define HIP_TRY_JOINED(expr)                                                                          \
		do {                                                                                           \
			hipError_t e_ = (expr);                                                                    \
			if(e_ != hipSuccess) {                                                                     \
				c->error = std::string(#expr) + ": " + hipGetErrorString(e_);                          \
				return abandon(e_ == hipErrorOutOfMemory ? ADYPT_E_OOM : ADYPT_E_HIP);                 \
			}                                                                                          \
		} while(0)
		if(n_pipes > 1)
		{
			HIP_TRY_JOINED(hipEventRecord(c->fork_ev, c->stream));
			for(int k = 1; k < n_pipes; ++k) HIP_TRY_JOINED(hipStreamWaitEvent(c->pipes[k].stream, c->fork_ev, 0));
		}
		struct Sub { QueueWindow win; FrameArgs f; int grid; };
		Sub sub[kMaxPipes];
		for(int k = 0, frame0 = 0; k < n_pipes; ++k)
		{
			const int frames_k = m / n_pipes + (k < m % n_pipes ? 1 : 0);
			sub[k].win = pipe_window(c, k, n_pipes);
			sub[k].f = f;
			sub[k].f.n_frames = frames_k; sub[k].f.frame_first = frame0;
			sub[k].grid = (int)(kNumSegments * (pass_seg_paths(c, sub[k].win, frames_k) / kShadeThreads)); // kNumSegments x chunks per segment
			frame0 += frames_k;
			const Pipe &pipe = c->pipes[k];
			hipEvent_t *stop = begin_timing(c, 1, pipe.stream);
			if(fused_first)
			{
				// camera rays + bounce 0 of every frame from the cached primary hits, the surface fetched once per pixel and tmpLifetime group
				QueueArgs q = queue_args(c, sub[k].win, 0, pipe.counters->count[0], pipe.counters->count[1], frames_k); // out = queue 1 = bounce 1's rays
				audit_before(c, q, pipe.stream, k);
				hipLaunchKernelGGL(k_shade_first, dim3((unsigned)(c->n_local_px / kShadeThreads)), dim3(kShadeThreads), 0, pipe.stream, sub[k].f, sc, q, px, stats ? 1 : 0);
				audit_after(c, q, pipe.stream, k);
			}
			else
			{
				QueueArgs q = queue_args(c, sub[k].win, 1, pipe.counters->count[0], pipe.counters->count[0], frames_k); // out = queue 0
				audit_before(c, q, pipe.stream, k);
				hipLaunchKernelGGL(k_gen_primary, dim3(sub[k].grid), dim3(kShadeThreads), 0, pipe.stream, sub[k].f, sc, q, px, use_cache, 1);
				audit_after(c, q, pipe.stream, k);
			}
			end_timing(stop, pipe.stream);
		}
		// every bounce after the first in ONE launch (k_path): the reference's for(b < uMaxBounce) inside a single dispatch
		const bool fused_bounces = fused_first && n_pipes == 1 && c->fused_bounces && (int64_t)m * (int64_t)c->n_local_px <= kPathMaxPaths;
		c->last_batch_fused = fused_bounces;
		if(fused_bounces && (max_bounce > 1 || f.sun_query))
		{
			const Pipe &pipe = c->pipes[0];
			SceneArgs sc_ref = sc; // the triangle records by REFERENCE index when the context holds that copy, else the uTriIndices remap inside k_path
			if(c->d_ref_triangles) sc_ref.triangles = (const float4 *)c->d_ref_triangles;
			int r = launch_path(c, pipe, sub[0].win, 1, pipe.counters->count[1], pipe.counters->cursor[1], sub[0].f, sc_ref, px, 1, stats);
			if(r != ADYPT_OK) return abandon(r);
		}
		for(int b = fused_first ? 1 : 0; b < max_bounce && !fused_bounces; ++b)
		{
			const int in = b & 1;
			for(int k = 0; k < n_pipes; ++k) // bounce by bounce over the pipes: their launches reach the GPU interleaved
			{
				const Pipe &pipe = c->pipes[k];
				FrameCounters *ctr = pipe.counters;
				if(!(b == 0 && use_cache))
				{
					int r = launch_trace(c, pipe, sub[k].win, in, ctr->count[b], ctr->cursor[b], c->params.stack_size, stats, nullptr, false, false, true, b == 0); // (b == 0: camera rays from the queue, tile by tile)
					if(r != ADYPT_OK) return abandon(r);
				}
				QueueArgs q = queue_args(c, sub[k].win, in, ctr->count[b], ctr->count[b + 1], sub[k].f.n_frames);
				ShadowArgs sh;
				sh.o = c->sh_o + sub[k].win.offset; sh.d = c->sh_d + sub[k].win.offset; sh.col = c->sh_col + sub[k].win.offset; sh.hit = c->sh_hit + sub[k].win.offset;
				sh.count = ctr->sh_count[b];
				memcpy(sh.dir, c->sun_dir, sizeof(sh.dir));
				sh.enabled = c->sun_visibility;
				hipEvent_t *stop = begin_timing(c, 1, pipe.stream);
				audit_before(c, q, pipe.stream, k);
				hipLaunchKernelGGL(k_shade, dim3(sub[k].grid), dim3(kShadeThreads), 0, pipe.stream, sub[k].f, sc, q, px, sh, b, (b == 0 && !use_cache) ? 1 : 0, stats ? 1 : 0);
				audit_after(c, q, pipe.stream, k);
				end_timing(stop, pipe.stream);
				if(c->sun_visibility)
				{
					// the escaped paths of this bounce: any-hit query towards the sun, then sun term + accumulate (pathtracer.glsl:130-135)
					int r = launch_trace(c, pipe, sub[k].win, 0, ctr->sh_count[b], ctr->sh_cursor[b], c->params.stack_size, stats, nullptr, true, true);
					if(r != ADYPT_OK) return abandon(r);
					stop = begin_timing(c, 1, pipe.stream);
					hipLaunchKernelGGL(k_shadow_resolve, dim3(sub[k].grid), dim3(kShadeThreads), 0, pipe.stream, sub[k].f, q, px, sh);
					end_timing(stop, pipe.stream);
				}
			}
		}
		for(int k = 1; k < n_pipes; ++k)
		{
			HIP_TRY_JOINED(hipEventRecord(c->pipes[k].done, c->pipes[k].stream));
			HIP_TRY_JOINED(hipStreamWaitEvent(c->stream, c->pipes[k].done, 0));
		}
		HIP_TRY_JOINED(hipGetLastError());
#undef HIP_TRY_JOINED
		if(as_batch)
		{
			c->batch_spp = c->spp; c->batch_frames = m; c->cache_group = 0;
			int r = resolve_batch_frames(c, sc, px, 0, hand_out);
			if(r != ADYPT_OK) return r;
			c->ahead_pos = hand_out; c->ahead_count = m - hand_out;
		}
		c->spp += hand_out;
		remaining -= hand_out;
	}
	return ADYPT_OK; // everything is enqueued on the context's stream; adypt_wait collects errors and kernel timings
}

int adypt_read_radiance(adypt_ctx *c, float *rgb)
{
	if(!c || !rgb) return ADYPT_E_INVALID;
	HIP_TRY(c, hipSetDevice(c->device));
	// the context's stream is non-blocking: a legacy-stream copy is not ordered after the frames enqueued by
	// adypt_trace_spp_async, so wait for them here (include/adypt_hip.h: entry points that read results synchronise)
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	if(c->n_local_px == 0) return ADYPT_OK; // a shard that owns no block: nothing of the image is this context's
	std::vector<float> local((size_t)c->n_local_px * 4);
	HIP_TRY(c, hipMemcpy(local.data(), c->d_accum, local.size() * sizeof(float), hipMemcpyDeviceToHost));
	return adypt_untile_host(c->width, c->height, c->rank, c->nranks, local.data(), rgb);
}

int adypt_read_display(adypt_ctx *c, uint8_t *rgba8)
{
	if(!c || !rgba8) return ADYPT_E_INVALID;
	HIP_TRY(c, hipSetDevice(c->device));
	if(c->n_local_px == 0) return ADYPT_OK;
	if(!c->d_display) HIP_TRY(c, hipMalloc((void **)&c->d_display, (size_t)c->n_local_px * sizeof(uint32_t))); // once: the size never changes
	hipLaunchKernelGGL(k_display, dim3((c->n_local_px + 255) / 256), dim3(256), 0, c->stream, (const float4 *)c->d_accum, c->n_local_px, c->view_type, c->d_display);
	std::vector<uint32_t> local((size_t)c->n_local_px);
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipMemcpyAsync(local.data(), c->d_display, local.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	for(int L = 0; L < c->n_local_px; ++L)
	{
		const int blk = c->local_blocks[(size_t)(L >> 10)];
		const int in = L & 1023, wt = in >> 6, ln = in & 63;
		const int x = (blk % c->blocks_x) * kBlockDim + (wt & 3) * 8 + (ln & 7), y = (blk / c->blocks_x) * kBlockDim + (wt >> 2) * 8 + (ln >> 3);
		if(x >= c->width || y >= c->height) continue;
		memcpy(rgba8 + ((size_t)y * c->width + x) * 4, &local[(size_t)L], 4);
	}
	return ADYPT_OK;
}

int adypt_read_hits(adypt_ctx *c, int32_t *tri, float *uv)
{
	if(!c || !tri || !uv) return ADYPT_E_INVALID;
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, hipStreamSynchronize(c->stream)); // see adypt_read_radiance
	if(c->n_local_px == 0) return ADYPT_OK;
	std::vector<float> local((size_t)c->n_local_px * 4);
	HIP_TRY(c, hipMemcpy(local.data(), c->d_cache, local.size() * sizeof(float), hipMemcpyDeviceToHost));
	for(int L = 0; L < c->n_local_px; ++L)
	{
		const int blk = c->local_blocks[(size_t)(L >> 10)];
		const int in = L & 1023, wt = in >> 6, ln = in & 63;
		const int x = (blk % c->blocks_x) * kBlockDim + (wt & 3) * 8 + (ln & 7), y = (blk / c->blocks_x) * kBlockDim + (wt >> 2) * 8 + (ln >> 3);
		if(x >= c->width || y >= c->height) continue;
		const size_t p = (size_t)y * c->width + x;
		memcpy(&tri[p], &local[(size_t)L * 4], 4);
		uv[p * 2] = local[(size_t)L * 4 + 1]; uv[p * 2 + 1] = local[(size_t)L * 4 + 2];
	}
	return ADYPT_OK;
}

static int trace_rays_impl(adypt_ctx *c, const float *rays, int64_t n, adypt_hit *hits, int with_stats, bool any_hit)
{
	if(!c || n < 0 || (n > 0 && (!rays || !hits))) return ADYPT_E_INVALID;
	if(!c->queues_ok) return fail(c, ADYPT_E_STATE, "adypt_trace_rays: the context lost its ray queues (failed adypt_set_frames_in_flight)");
	HIP_TRY(c, hipSetDevice(c->device));
	drop_rolling(c); // (a frame started ahead works in a window of the queues this call is about to fill)
	if(!c->pt_started) { int r = apply_params(c); if(r != ADYPT_OK) return r; }
	if(with_stats) { int r = ensure_ray_stats(c); if(r != ADYPT_OK) return r; }
	std::vector<float4> o, d, h;
	std::vector<RayStats> rs;
	for(int64_t done = 0; done < n;)
	{
		const int64_t m = std::min<int64_t>(n - done, c->capacity);
		o.resize((size_t)m); d.resize((size_t)m); h.resize((size_t)m);
		for(int64_t i = 0; i < m; ++i)
		{
			const float *r = rays + (size_t)(done + i) * 8;
			o[(size_t)i] = make_float4(r[0], r[1], r[2], r[3]);
			d[(size_t)i] = make_float4(r[4], r[5], r[6], 0.0f);
		}
		// the batch is cut into kNumSegments consecutive pieces, piece s occupying the head of queue segment s
		clear_counters(c, c->d_counters, 1, c->stream);
		const int64_t piece = (m + kNumSegments - 1) / kNumSegments; // <= seg_cap because m <= capacity
		uint32_t counts[kNumSegments * kCursorStride] = {0};
		if(with_stats) rs.resize((size_t)m);
		for(int s = 0; s < kNumSegments; ++s)
		{
			const int64_t b0 = std::min<int64_t>(piece * s, m), n_s = std::min<int64_t>(piece, m - b0);
			counts[s * kCursorStride] = (uint32_t)n_s;
			if(n_s <= 0) continue;
			const size_t off = (size_t)s * c->seg_cap;
			HIP_TRY(c, hipMemcpyAsync(c->q_o[0] + off, o.data() + b0, (size_t)n_s * sizeof(float4), hipMemcpyHostToDevice, c->stream));
			HIP_TRY(c, hipMemcpyAsync(c->q_d[0] + off, d.data() + b0, (size_t)n_s * sizeof(float4), hipMemcpyHostToDevice, c->stream));
		}
		HIP_TRY(c, hipMemcpyAsync(c->d_counters->count[0], counts, sizeof(counts), hipMemcpyHostToDevice, c->stream));
		int r = launch_trace(c, c->pipes[0], full_window(c), 0, c->d_counters->count[0], c->d_counters->cursor[0], c->params.stack_size, with_stats != 0, with_stats ? c->d_ray_stats : nullptr, any_hit);
		if(r != ADYPT_OK) return r;
		for(int s = 0; s < kNumSegments; ++s)
		{
			const int64_t b0 = std::min<int64_t>(piece * s, m), n_s = std::min<int64_t>(piece, m - b0);
			if(n_s <= 0) continue;
			const size_t off = (size_t)s * c->seg_cap;
			HIP_TRY(c, hipMemcpyAsync(h.data() + b0, c->d_hit + off, (size_t)n_s * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
			if(with_stats) HIP_TRY(c, hipMemcpyAsync(rs.data() + b0, c->d_ray_stats + off, (size_t)n_s * sizeof(RayStats), hipMemcpyDeviceToHost, c->stream));
		}
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		for(int64_t i = 0; i < m; ++i)
		{
			adypt_hit &out = hits[done + i];
			memcpy(&out.tri_id, &h[(size_t)i].x, 4);
			out.u = h[(size_t)i].y; out.v = h[(size_t)i].z; out.t = h[(size_t)i].w;
			if(with_stats)
			{
				const RayStats &s = rs[(size_t)i];
				out.ref_idx = s.ref_idx; out.nodes = s.nodes; out.tris = s.tris; out.hash = s.hash; out.max_depth = s.max_depth;
			}
			else { out.ref_idx = out.tri_id == -1 ? -1 : 0; out.nodes = out.tris = out.hash = out.max_depth = 0; }
		}
		done += m;
	}
	harvest_events(c);
	return ADYPT_OK; // stack overflows of arbitrary batches are reported per ray (max_depth = 0xffffffff) and in the stats
}

int adypt_trace_rays(adypt_ctx *c, const float *rays, int64_t n, adypt_hit *hits, int with_stats)
{
	return trace_rays_impl(c, rays, n, hits, with_stats, false);
}

int adypt_trace_rays_any(adypt_ctx *c, const float *rays, int64_t n, adypt_hit *hits, int with_stats)
{
	return trace_rays_impl(c, rays, n, hits, with_stats, true);
}

int adypt_get_stats(adypt_ctx *c, adypt_stats *out)
{
	if(!c || !out) return ADYPT_E_INVALID;
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	harvest_events(c);
	DeviceStats st;
	HIP_TRY(c, hipMemcpy(&st, c->d_stats, sizeof(st), hipMemcpyDeviceToHost));
	out->rays = st.rays; out->nodes_visited = st.nodes; out->tris_tested = st.tris; out->hits = st.hits; out->shaded = st.shaded;
	out->stack_overflows = st.overflows; out->bad_materials = st.bad_materials; out->max_stack = st.max_stack;
	out->trace_launches = c->trace_launches; out->trace_ms = c->trace_ms; out->shade_ms = c->shade_ms;
	out->path_ms = c->path_ms; out->path_launches = c->path_launches; out->audit_errors = (uint32_t)std::min<unsigned long long>(st.audit_errors, 0xffffffffull);
	out->path_rays = st.path_rays; out->path_nodes = st.path_nodes; out->path_tris = st.path_tris; out->path_hits = st.path_hits; out->path_shaded = st.path_shaded;
	return ADYPT_OK;
}

int adypt_get_shader_clock(adypt_ctx *c, uint64_t out[2])
{
	if(!c || !out) return ADYPT_E_INVALID;
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	DeviceStats st;
	HIP_TRY(c, hipMemcpy(&st, c->d_stats, sizeof(st), hipMemcpyDeviceToHost));
	out[0] = st.clock_cycles; out[1] = st.clock_ticks;
	return ADYPT_OK;
}

int adypt_get_wave_profile(adypt_ctx *c, uint64_t out[8])
{
	if(!c || !out) return ADYPT_E_INVALID;
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	DeviceStats st;
	HIP_TRY(c, hipMemcpy(&st, c->d_stats, sizeof(st), hipMemcpyDeviceToHost));
	for(int i = 0; i < 8; ++i) out[i] = st.wave_profile[i];
	return ADYPT_OK;
}

int adypt_reset_stats(adypt_ctx *c)
{
	if(!c) return ADYPT_E_INVALID;
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	harvest_events(c);
	// on the context's own stream: a legacy-stream hipMemset is not ordered against a non-blocking stream
	HIP_TRY(c, hipMemsetAsync(c->d_stats, 0, offsetof(DeviceStats, host_overflow), c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	*c->h_overflow = 0u;
	c->trace_ms = c->shade_ms = c->path_ms = 0; c->trace_launches = c->path_launches = 0;
	return ADYPT_OK;
}

int64_t adypt_local_pixel_count(const adypt_ctx *c) { return c ? c->n_local_px : ADYPT_E_INVALID; }

int adypt_copy_local_radiance(adypt_ctx *c, void *dst, int64_t capacity_float4)
{
	if(!c || !dst || capacity_float4 < c->n_local_px) return ADYPT_E_INVALID;
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, hipMemcpyAsync(dst, c->d_accum, (size_t)c->n_local_px * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
	if(capacity_float4 > c->n_local_px)
		HIP_TRY(c, hipMemsetAsync((char *)dst + (size_t)c->n_local_px * sizeof(float4), 0, (size_t)(capacity_float4 - c->n_local_px) * sizeof(float4), c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	return ADYPT_OK;
}

int adypt_assemble_radiance(adypt_ctx *c, const void *gathered, int64_t stride_float4, void *rgb_device)
{
	if(!c || !gathered || !rgb_device || stride_float4 < 0) return ADYPT_E_INVALID;
	HIP_TRY(c, hipSetDevice(c->device));
	if(!c->d_all_blocks)
	{
		// block lists of every rank of the shard, back to back (uploaded once)
		std::vector<int32_t> all;
		c->all_blocks_offset.assign(1, 0);
		for(int r = 0; r < c->nranks; ++r)
		{
			const std::vector<int32_t> b = owned_blocks(c->width, c->height, r, c->nranks);
			all.insert(all.end(), b.begin(), b.end());
			c->all_blocks_offset.push_back((int64_t)all.size());
		}
		int rr = upload(c, &c->d_all_blocks, all.data(), all.size());
		if(rr != ADYPT_OK) return rr;
	}
	for(int r = 0; r < c->nranks; ++r)
	{
		const int64_t n_blocks = c->all_blocks_offset[(size_t)r + 1] - c->all_blocks_offset[(size_t)r];
		const int n_px = (int)(n_blocks * kBlockPixels);
		if(n_px == 0) continue;
		if(stride_float4 < n_px) return fail(c, ADYPT_E_INVALID, "adypt_assemble_radiance: stride smaller than a rank's buffer");
		hipLaunchKernelGGL(k_untile, dim3((n_px + 255) / 256), dim3(256), 0, c->stream, (const float4 *)gathered + (size_t)r * (size_t)stride_float4,
						   (const int32_t *)c->d_all_blocks + c->all_blocks_offset[(size_t)r], n_px, c->blocks_x, c->width, c->height, (float *)rgb_device);
	}
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	return ADYPT_OK;
}

int adypt_local_radiance_device(adypt_ctx *c, void **dptr)
{
	if(!c || !dptr) return ADYPT_E_INVALID;
	*dptr = c->d_accum;
	return ADYPT_OK;
}

}  // extern "C"
