// k_trace — persistent wave64 CWBVH8 closest-hit traversal for gfx950 (the hot kernel).
//
// Semantics: exactly BVHIntersection of shaders/traversal.glsl:14-255 — same node-visit order, same triangle test
// order, same arithmetic (canon_math.hpp) — so hit ids, u/v/t bits and the visit hash equal the oracle's.
// What is re-designed for the machine (none of it changes a result):
//   * persistent waves with per-lane ray replacement: a lane whose ray has finished is refilled as soon as
//     >= kRefillMin lanes of its wave are idle (wave vote with __ballot / __popcll), instead of the whole wave
//     waiting for its slowest ray (measured SIMT utilisation of the batch-synchronous first version: 19 %).
//     Rays are reserved from the queue kChunk at a time (one device atomic per chunk, wave-private range handed out
//     to idle lanes without further atomics): a per-refill atomic was measured 2.5x slower end to end.
//   * software-pipelined node fetch: which node comes next (closest remaining child of the current group, or the
//     popped group) is decided by the *previous* slab test and does not depend on the triangle tests in between, so
//     its 5 x 16-byte loads are issued together with the triangle loads of the current node and the latencies
//     overlap.  The slab test itself still runs after the triangle tests (it needs the shortened hit_t), exactly
//     like the reference.
//   * the wave's triangles are ONE list: a lane offers up to three triangles of its node per trip and lane j of the wave tests
//     entry j, whoever owns it — rays and verdicts travel through the LDS crossbar (ds_bpermute_b32), the owner applies the
//     verdicts in the reference's order (traverse_trip.inc, sections B / C).
//   * the node-group stack lives in LDS, laid out [depth][lane] (ds_write_b64 / ds_read_b64, conflict free); only
//     entries deeper than kLdsStackMax spill to a global scratch array.  Overflow beyond stackSize is reported.
//   * XCD-affine queue segments with separate fetch cursors on separate cache lines (see shade.hpp).
#pragma once
#include "shade.hpp"

namespace adypt {

// 16-byte records of the trip's two tables through a pointer that KEEPS the global address space (k_path launders its base pointers through an empty
// asm to pin them; as generic pointers the loads would become flat_load, which also occupies the LDS queue) or through a plain one (k_trace)
typedef float NativeF4 __attribute__((ext_vector_type(4)));
typedef uint32_t NativeU4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) NativeF4 *GlobalF4;
typedef const __attribute__((address_space(1))) NativeU4 *GlobalU4;
__device__ __forceinline__ float4 trip_ld(GlobalF4 p) { const NativeF4 v = *p; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint4 trip_ld(GlobalU4 p) { const NativeU4 v = *p; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float4 trip_ld(const float4 *p) { return *p; }
__device__ __forceinline__ uint4 trip_ld(const uint4 *p) { return *p; }

constexpr int kRefillMin = 16; // default: refill when at least this many lanes of the wave are idle (or all are)
constexpr int kChunk = 128;    // default: rays a workgroup reserves per queue atomic
constexpr int kBite = 32;      // default: rays a wave takes from its workgroup's reservation at a time (end of a launch)
constexpr int kEndgame = 4;    // default: the end of a launch = fewer than this many more chunks per wave left in the segment
struct WgPool { unsigned long long range; uint32_t lock, dry, left, pad; }; // range = (end << 32) | next: reserved, not yet handed to a wave; left: waves that have left the loop (k_trace_camera)
constexpr int kNodeUint4 = 5;  // the 80-byte WideBVHNode, verbatim (src/BVH/WideBVH.hpp:13-26)
constexpr size_t kTripTabBytes = (size_t)(kTraceThreads / 64) * 64; // LDS: one byte per lane and wave — which lane owns entry j of the wave's triangle list

// The lane's column of the HBM spill array (stack entries beyond the LDS depth): my_spill[depth x lanes of the launch].  AT_USE: the column's
// address is computed where it is used, behind a value the compiler cannot hoist — as a pointer it holds two VGPRs through the whole persistent
// loop for the sake of a rare branch (k_path, k_trace_camera: the registers are needed; k_trace keeps the pointer, it has them).
template <bool AT_USE> struct SpillColumn;
template <> struct SpillColumn<false> {
	uint2 *column;
	__device__ __forceinline__ explicit SpillColumn(uint2 *base) : column(base + (blockIdx.x * (uint32_t)kTraceThreads + threadIdx.x)) {}
	__device__ __forceinline__ uint2 &operator[](size_t i) const { return column[i]; }
};
template <> struct SpillColumn<true> {
	uint2 *base;
	__device__ __forceinline__ explicit SpillColumn(uint2 *b) : base(b) {}
	__device__ __forceinline__ uint2 &operator[](size_t i) const
	{
		uint32_t t = threadIdx.x;
		asm volatile("" : "+v"(t));
		return base[i + (size_t)(blockIdx.x * (uint32_t)kTraceThreads + t)];
	}
};

// Reserve up to `want` consecutive rays: first from the segment of "our" XCD (blockIdx & 7 groups the workgroups
// that share an L2 under the observed round-robin dispatch — a speed hint only), then steal from the others.
// One device atomic and nothing else per reservation: the segment lengths are read once per wave (lane s of
// `seg_len_lanes` holds count[s]) and a segment found exhausted is remembered in `seg_done`, so the three dependent
// global round trips of the first version (count, cursor pre-check, atomic) shrink to one — measured: reservations
// were 18 % of the kernel's time at 128 rays per reservation.
__device__ __forceinline__ uint32_t fetch_rays(uint32_t seg_len_lanes, uint32_t &seg_done, uint32_t *cursor, uint32_t seg_cap, int home, uint32_t want, uint32_t *begin,
                                               uint32_t *left)
{
	for(int k = 0; k < kNumSegments; ++k)
	{
		const int s = (home + k) & (kNumSegments - 1);
		if((seg_done >> s) & 1u) continue;
		const uint32_t seg_len = (uint32_t)__builtin_amdgcn_readlane((int)seg_len_lanes, s);
		uint32_t rel = 0;
		if(threadIdx.x % 64 == 0) rel = atomicAdd(&cursor[s * kCursorStride], want);
		rel = (uint32_t)__builtin_amdgcn_readfirstlane((int)rel);
		if(rel < seg_len)
		{
			*begin = (uint32_t)s * seg_cap + rel;
			const uint32_t got = min(want, seg_len - rel);
			*left = seg_len - rel - got; // rays of this segment nobody had reserved yet
			return got;
		}
		seg_done |= 1u << s;
	}
	*left = 0;
	return 0;
}

// ANY = the any-hit overload of the reference (traversal.glsl:257-494, never called by its shaders — SURVEY.md §8 f1):
// identical traversal, the ray ends at the FIRST accepted triangle in traversal order.
//
// CAMERA = the rays are the camera rays of a pass (primaryray.glsl:23-44 Camera(), pathtracer.glsl:51-71 with the sub-pixel bias): nothing is read
// from a ray queue and no hit goes to one — the lane computes its ray from the queue POSITION it reserved (the position -> path mapping of
// k_gen_primary, so the XCD segments hold what they held) and writes the finished hit into the cache image of its pixel (what k_gen_primary ->
// k_trace -> k_viewer / k_store_cache did in three launches and 2 x 40 bytes of queue traffic per ray).
struct TraceCameraArgs {            // the ONE parameter of k_trace_camera (offset 0 of the kernarg segment: camera_args())
	TraceArgs a;                    //   a.count, a.ray_o, a.ray_d, a.hit are unused; a.seg_cap = 1 << seg_shift (positions are numbers, not memory)
	uint32_t seg_paths;             //   queue positions per segment (QueueArgs::seg_paths of the pass; a multiple of 256)
	uint32_t seg_shift;             //   position = segment << seg_shift | place in the segment
	unsigned long long rays;        //   camera rays of the pass (in-image pixels of the owned blocks x frames)
	uint32_t *left;                 //   [kNumSegments + 1][kCursorStride] workgroups that have left the launch, per home segment, and segments complete.
	                                //   a.cursor and these words belong to camera launches alone and are zero between them: the last workgroup to
	                                //   leave a launch puts them back (no clearing launch in front of a primary-only call)
	FrameArgs f;                    //   n_frames / frame_first / frame_stride: the frames of the pass
	const int32_t *local_blocks;
	PixelArgs px;                   //   cache, cache_next: where the hits go; accum: VIEWER's colours
	SceneArgs sc;                   //   VIEWER only
	int32_t viewer_type;            //   VIEWER only (primaryray.glsl's uType)
	int32_t bias_mode;              //   0: Camera() of primaryray.glsl; 1: Camera(SubPixel()) of pathtracer.glsl
};
// read where they are used (the refill block), not held in SGPRs across the persistent loop (see k_path's rare_args())
__device__ __forceinline__ const TraceCameraArgs &camera_args()
{
	unsigned long long k = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("" : "+s"(k));
	return *(const TraceCameraArgs *)(const __attribute__((address_space(4))) TraceCameraArgs *)k;
}
// The hit of a camera ray -> cache image of the frame's tmpLifetime group (pathtracer.glsl:121-127, primaryray.glsl:93).  `where` = group x
// local pixels + local pixel: group 0 is the cache image, the later groups of a batch follow in cache_next.
__device__ __forceinline__ void camera_store_hit(const TraceCameraArgs &R, uint32_t where, int32_t tri, float u, float v)
{
	const uint32_t n = (uint32_t)R.f.n_local_px;
	float4 *dst = where < n ? R.px.cache + where : R.px.cache_next + (where - n);
	*dst = make_float4(__int_as_float(tri), u, v, 0.0f);
}
// VIEWER (a primary-only call, one frame): the pixel's colour as well (primaryray.glsl:46-94) — `where` is the local pixel
__device__ __forceinline__ void camera_store_view(const TraceCameraArgs &R, uint32_t where, int32_t tri, float u, float v)
{
	const F3 c = viewer_color(R.f, R.sc, R.px.stats, tri, u, v, R.viewer_type);
	R.px.accum[where] = make_float4(c.x, c.y, c.z, 1.0f);
}

template <bool STATS, bool ANY, bool CAMERA, bool VIEWER = false>
__device__ __forceinline__ void trace_loop(const TraceArgs a)
{
	constexpr bool kUniformTmin = CAMERA; // ray batches handed in by the caller carry a tmin per ray; camera rays the pass's
	extern __shared__ uint2 lds_stack[]; // [waves][lds_depth][64], then the workgroup's ray pool (WgPool)
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	uint2 *my_stack = lds_stack + (size_t)wave * a.lds_depth * 64 + lane;
	// Workgroup ray pool, for the END of a launch.  A reservation from the global queue is a.chunk rays for one device atomic, and while
	// the queue is long the wave that makes it keeps all of them (the pool stays empty: round 2's behaviour).  What a launch loses at
	// its end is mostly the spread of the moments at which the waves next need rays (tools/wave_timeline.py: 110 us at 128 rays per
	// wave — a wave that reserved just before the cursors ran out works on while 5119 others are done).  So once a reservation leaves
	// fewer rays in its segment than one more chunk for every wave, the chunk goes into the pool and the 4 waves take it a.bite rays
	// at a time (LDS compare-and-swap): the rays a wave can be left holding shrink 4x at the same number of device atomics.
	WgPool *pool = (WgPool *)(lds_stack + (size_t)(kTraceThreads / 64) * a.lds_depth * 64);
	uint8_t *const trip_tab = (uint8_t *)(pool + 1) + wave * 64; // the wave's triangle hand-out table (traverse_trip.inc, section B)
	if(threadIdx.x == 0) { pool->range = 0ull; pool->lock = 0u; pool->dry = 0u; pool->left = 0u; }
	__syncthreads();
	// a.endgame more chunks for every wave of the segment (64-bit product: the tuning overrides allow 1024 x 4096 x 1024 waves)
	const uint32_t endgame_rays = (uint32_t)min((unsigned long long)a.endgame * a.chunk * max(1u, (gridDim.x * (kTraceThreads / 64)) / kNumSegments), 0xffffffffull);
	const uint32_t total_lanes = gridDim.x * (uint32_t)kTraceThreads;
	const SpillColumn<CAMERA> my_spill(a.spill);
	const int home = blockIdx.x & (kNumSegments - 1);

	// the clock the chip holds under THIS launch: shader cycles (s_memtime) against the constant 100 MHz counter (s_memrealtime) over the life
	// of workgroup 0's first wave — bench.py's vector-ALU roof is 1024 SIMDs x this, measured in the run it prices (adypt_get_shader_clock)
	const unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_r0 = __builtin_amdgcn_s_memrealtime();
	if(blockIdx.x == 0 && threadIdx.x == 0)
	{
		unsigned long long total = CAMERA ? camera_args().rays : 0ull;
		if(!CAMERA) for(int s = 0; s < kNumSegments; ++s) total += a.count[s * kCursorStride];
		atomicAdd(&a.stats->rays, total);
	}

	// per-lane ray state
	bool active = false;
	uint32_t ray = 0;
	// origin and direction are kept interleaved per axis — (o.x,d.x), (o.y,d.y), (o.z,d.z) in adjacent registers — so one
	// v_pk_fma_f32 advances o·m and d·m of a Woop row together (same products, same fma order as dot3: bit-identical)
	V2 od_x = v2(0, 0), od_y = v2(0, 0), od_z = v2(0, 1);
	F3 idir = f3(0, 0, 1);
	bool nx = false, ny = false, nz = false;
	uint32_t octinv = 7u;
	float tmin = CAMERA ? a.tmin : 0.0f, hit_t = 1e9f, hit_u = 0.0f, hit_v = 0.0f;
	int32_t hit_idx = -1;
	int sp = 0;
	uint32_t ng_x = 0, ng_y = 0, tg_x = 0, tg_y = 0;
	uint32_t n_nodes = 0, n_tris = 0, hash = 0, max_depth = 0;
	bool overflow = false;
	bool flush = false;                          // the lane's ray is finished but its result is not written yet (see E / refill)
	bool pending = false;                        // a next node is chosen (and pushed for) but not yet slab-tested
	uint32_t node = 0, depth_after_push = 0;     // push bookkeeping is committed when the node is actually visited (D)
	bool push_overflow = false;

	unsigned long long st_nodes = 0, st_tris = 0, st_hits = 0;
	uint32_t st_maxdepth = 0;
	bool any_overflow = false, exhausted = false;
	uint32_t loc_next = 0, loc_end = 0; // wave-uniform: reserved but not yet started rays
	const uint32_t seg_len_lanes = lane < kNumSegments ? (CAMERA ? camera_args().seg_paths : a.count[lane * kCursorStride]) : 0u;
	uint32_t seg_done = 0;              // wave-uniform: segments found empty or exhausted
	for(int sgm = 0; sgm < kNumSegments; ++sgm)
		if(__builtin_amdgcn_readlane((int)seg_len_lanes, sgm) == 0) seg_done |= 1u << sgm;
	unsigned long long wp[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // STATS: wave-occupancy profile (adypt_get_wave_profile)
	// one count per wave-level event, taken by the first lane that executes it, plus the number of lanes executing it
	auto wave_event = [&](int slot) {
		const unsigned long long m = __ballot(true);
		if(lane == (int)__builtin_ctzll(m)) { wp[slot] += 1; wp[slot + 1] += (unsigned long long)__popcll(m); }
	};

	constexpr bool kOverflowPerRay = true;
	constexpr bool kTripShadowRays = false; // (a launch of k_trace is closest-hit or any-hit as a whole: ANY)
	const float4 *const trip_woop = a.woop; // (the trip's two base pointers: traverse_trip.inc)
	const uint4 *const trip_nodes = a.nodes;
	for(;;)
	{
		// ---------------- refill idle lanes ----------------
		const unsigned long long idle = __ballot(!active);
		const uint32_t n_idle = (uint32_t)__popcll(idle);
		if((!exhausted || loc_next < loc_end) && (n_idle >= a.refill_min))
		{
			if(loc_next == loc_end)
			{
				// a bite from the workgroup's pool (lane 0 does the LDS work, the wave shares the result)
				uint32_t cb = 0, cn = 0, dry = 0;
				if(lane == 0)
				{
					unsigned long long r = __hip_atomic_load(&pool->range, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					for(;;)
					{
						const uint32_t nx_ = (uint32_t)r, en_ = (uint32_t)(r >> 32);
						if(nx_ >= en_) break;
						const uint32_t take_ = min(a.bite, en_ - nx_);
						const unsigned long long want_ = ((unsigned long long)en_ << 32) | (nx_ + take_);
						if(__hip_atomic_compare_exchange_strong(&pool->range, &r, want_, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { cb = nx_; cn = take_; break; }
					}
					if(cn == 0) dry = __hip_atomic_load(&pool->dry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
				cb = (uint32_t)__builtin_amdgcn_readfirstlane((int)cb); cn = (uint32_t)__builtin_amdgcn_readfirstlane((int)cn);
				dry = (uint32_t)__builtin_amdgcn_readfirstlane((int)dry);
				if(cn == 0 && !dry)
				{
					// the pool is empty: the wave that gets the lock reserves the next chunk for the workgroup (and takes its own bite at once);
					// the others carry on with the rays they have and look again next trip
					uint32_t mine = 0;
					if(lane == 0)
					{
						mine = atomicCAS(&pool->lock, 0u, 1u) == 0u ? 1u : 0u;
						if(mine)
						{	// another wave may have refilled the pool between our look at it and our getting the lock: never overwrite rays
							const unsigned long long r = __hip_atomic_load(&pool->range, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
							if((uint32_t)r < (uint32_t)(r >> 32) || __hip_atomic_load(&pool->dry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
							{
								__hip_atomic_store(&pool->lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
								mine = 0; // look again next trip
							}
						}
					}
					if(__builtin_amdgcn_readfirstlane((int)mine))
					{
						uint32_t gb = 0, left = 0;
						const uint32_t gn = fetch_rays(seg_len_lanes, seg_done, a.cursor, a.seg_cap, home, a.chunk, &gb, &left);
						cb = gb; cn = left < endgame_rays ? min(a.bite, gn) : gn;
						if(lane == 0)
						{
							if(gn == 0) __hip_atomic_store(&pool->dry, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
							else __hip_atomic_store(&pool->range, ((unsigned long long)(gb + gn) << 32) | (gb + cn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
							__hip_atomic_store(&pool->lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
						}
						dry = gn == 0 ? 1u : 0u;
					}
				}
				loc_next = cb; loc_end = cb + cn;
				if(cn == 0 && dry) exhausted = true;
			}
			const uint32_t begin = loc_next;
			uint32_t got = min(n_idle, loc_end - loc_next);
			if(CAMERA) got = min(got, 256u - (begin & 255u)); // (one refill stays inside one 256-position chunk = inside one 32x32 block of one frame: see below)
			loc_next += got;
			if(STATS && lane == 0) wp[6] += 1;
			const uint32_t my_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u)); // idle lanes below this one
			// results of the rays the idle lanes finished since the last refill (traversal.glsl:253-254: remap, write):
			// written here, >= refill_min lanes at a time and with the remap load in flight next to the loads of the new
			// rays, instead of by the 5 lanes that finish in an average trip with a dependent load-then-store of their own
			int32_t flush_tri = -1;
			if(flush && hit_idx != -1) flush_tri = a.tri_indices[hit_idx];
			const bool take_slot = !active && my_rank < got;
			const uint32_t slot_ray = begin + my_rank;
			bool take_cam = false;       // CAMERA: the position holds a pixel of the image
			uint32_t cam_ray = 0;        // CAMERA: `ray` = where the hit goes (tmpLifetime group x local pixels + local pixel), not the queue position
			float4 ro = make_float4(0, 0, 0, 0), rd = make_float4(0, 0, 1, 0);
			if(CAMERA)
			{
				const TraceCameraArgs &R = camera_args();
				if(flush) { camera_store_hit(R, ray, flush_tri, hit_u, hit_v); if(VIEWER) camera_store_view(R, ray, flush_tri, hit_u, hit_v); flush = false; }
				// Queue position -> path.  The positions of a refill are consecutive and lie inside one 256-position chunk of one segment, and
				// chunks map to runs of 256 paths (k_gen_primary's dealing: chunk c of segment s is chunk c * 8 + s of the pass, or the segments
				// are contiguous runs), which lie inside one 1024-pixel block of one frame: frame, tmpLifetime group, sub-pixel bias and the
				// block's place in the image are the same for the whole wave — scalar arithmetic, and the block table is read with a scalar load.
				const uint32_t seg = begin >> R.seg_shift, rel = begin & ((1u << R.seg_shift) - 1u);
				const uint32_t q0 = R.f.deal_chunks ? ((((rel >> 8) * kNumSegments + seg) << 8) | (rel & 255u)) : seg * R.seg_paths + rel;
				const uint32_t npx = (uint32_t)R.f.n_local_px;
				uint32_t ordinal = 0, l0 = q0;
				if(R.f.n_frames != 1) { ordinal = q0 / npx; l0 = q0 - ordinal * npx; }
				const bool in_pass = q0 < npx * (uint32_t)R.f.n_frames; // (the last chunks of the segments may lie beyond the pass)
				const int blk = in_pass ? R.local_blocks[l0 >> 10] : 0;
				const int blk_x = (blk % R.f.blocks_x) * kBlockDim, blk_y = (blk / R.f.blocks_x) * kBlockDim;
				const int frame = R.f.frame_first + (int)ordinal * R.f.frame_stride;
				const uint32_t group = frame == 0 ? 0u : (uint32_t)frame_group(R.f, frame);
				float bx = 0.0f, by = 0.0f;
				if(R.bias_mode)
				{
					const int sub_idx = ((R.f.spp + frame) / R.f.tmp_life) % (R.f.subpixel * R.f.subpixel);
					const float unit = 1.0f / (float)R.f.subpixel;
					bx = (float)(sub_idx / R.f.subpixel) * unit;
					by = (float)(sub_idx % R.f.subpixel) * unit;
				}
				// per lane: the pixel inside the block (local_pixel_xy), the ray
				const uint32_t in = (l0 & 1023u) + my_rank, wt = in >> 6, ln = in & 63u;
				const int x = blk_x + (int)((wt & 3u) * 8u + (ln & 7u)), y = blk_y + (int)((wt >> 2) * 8u + (ln >> 3));
				const bool take = take_slot && in_pass && x < R.f.width && y < R.f.height; // (pixels of a border block beyond the image have no ray)
				const uint32_t new_ray = group * npx + l0 + my_rank; // `ray` of a camera pass: where the hit goes (camera_store_hit)
				if(take)
				{
					const F3 d = camera_dir(R.f, x, y, bx, by);
					ro = make_float4(R.f.origin[0], R.f.origin[1], R.f.origin[2], 0.0f);
					rd = make_float4(d.x, d.y, d.z, 0.0f);
				}
				take_cam = take; cam_ray = new_ray;
			}
			const bool take = CAMERA ? take_cam : take_slot;
			const uint32_t new_ray = CAMERA ? cam_ray : slot_ray;
			if(!CAMERA)
			{
				if(take)
				{
					if(a.packed) { const F3 o3 = ld3((const float *)a.ray_o, new_ray); ro = make_float4(o3.x, o3.y, o3.z, a.tmin); }
					else ro = a.ray_o[new_ray];
					rd = a.ray_d[new_ray];
				}
				if(flush)
				{
					if(a.packed) st3((float *)a.hit, ray, __int_as_float(flush_tri), hit_u, hit_v);
					else a.hit[ray] = make_float4(__int_as_float(flush_tri), hit_u, hit_v, hit_t);
					flush = false;
				}
			}
			if(take)
			{
				// ---- ray setup (traversal.glsl:16-35) ----
				ray = new_ray;
				const float ooeps = __uint_as_float((127u - 64u) << 23);
				F3 dir = f3(rd.x, rd.y, rd.z);
				dir.x = fabsf(dir.x) > ooeps ? dir.x : (dir.x >= 0 ? ooeps : -ooeps);
				dir.y = fabsf(dir.y) > ooeps ? dir.y : (dir.y >= 0 ? ooeps : -ooeps);
				dir.z = fabsf(dir.z) > ooeps ? dir.z : (dir.z >= 0 ? ooeps : -ooeps);
				dir = normalize3(dir);
				idir = f3(rcp_ieee(dir.x), rcp_ieee(dir.y), rcp_ieee(dir.z));
				nx = dir.x < 0; ny = dir.y < 0; nz = dir.z < 0;
				octinv = 7u - ((nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u));
				F3 origin = f3(ro.x, ro.y, ro.z);
				if(!CAMERA) tmin = ro.w;
				// make the ray loads complete inside this (rare) refill block: otherwise the compiler's s_waitcnt
				// bookkeeping carries them into the traversal loop and drains the software-pipelined node fetch
				if(CAMERA) asm volatile("" : "+v"(origin.x), "+v"(origin.y), "+v"(origin.z));
				else asm volatile("" : "+v"(origin.x), "+v"(origin.y), "+v"(origin.z), "+v"(tmin));
				od_x = v2(origin.x, dir.x); od_y = v2(origin.y, dir.y); od_z = v2(origin.z, dir.z);
				hit_t = 1e9f; hit_u = 0.0f; hit_v = 0.0f; hit_idx = -1;
				sp = 0;
				ng_x = 0; ng_y = 0x80000000u; tg_x = 0; tg_y = 0;
				if(STATS) { n_nodes = 0; n_tris = 0; hash = 0x811c9dc5u; max_depth = 0; }
				overflow = false; pending = false; push_overflow = false; depth_after_push = 0;
				active = true;
			}
		}
		const unsigned long long live = __ballot(active);
		if(STATS && lane == 0) { wp[0] += 1; wp[1] += (unsigned long long)__popcll(live); }
		// (no `continue` around the trip: with one, the loop-carried ray state lives in two register sets and is copied at the loop's latch — path.hpp)
		if(live == 0ull)
		{
			if(STATS && lane == 0) wp[7] += 1;
			if(exhausted && loc_next == loc_end) break;
		}
		else
#define ADYPT_TRIP_TAKE_HIT(u, v, idx) { hit_u = (u); hit_v = (v); hit_idx = (int32_t)(idx); }
#define ADYPT_TRIP_SHADOW false
#include "traverse_trip.inc"
#undef ADYPT_TRIP_SHADOW
#undef ADYPT_TRIP_TAKE_HIT
	}

	if(flush) // rays finished after the queue ran dry
	{
		const int32_t tri_id = hit_idx != -1 ? a.tri_indices[hit_idx] : -1;
		if(CAMERA) { const TraceCameraArgs &R = camera_args(); camera_store_hit(R, ray, tri_id, hit_u, hit_v); if(VIEWER) camera_store_view(R, ray, tri_id, hit_u, hit_v); }
		else if(a.packed) st3((float *)a.hit, ray, __int_as_float(tri_id), hit_u, hit_v);
		else a.hit[ray] = make_float4(__int_as_float(tri_id), hit_u, hit_v, hit_t);
	}
	if(CAMERA && lane == 0)
	{
		// Every fetch of this wave has returned (the loop used their results).  The last wave of the launch to get here zeroes the cursors — counted
		// wave -> workgroup (LDS) -> home segment -> launch, so that no counter takes more than a few hundred atomics (all 6144 waves on one word
		// held the end of the launch up by 35 us: a single address takes ~88 atomics per us on this chip)
		if(__hip_atomic_fetch_add(&pool->left, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == (uint32_t)(kTraceThreads / 64) - 1u)
		{
			const TraceCameraArgs &R = camera_args();
			const uint32_t in_segment = (gridDim.x + (uint32_t)(kNumSegments - 1 - home)) / (uint32_t)kNumSegments; // workgroups whose home this segment is
			if(atomicAdd(&R.left[home * kCursorStride], 1u) == in_segment - 1u && atomicAdd(&R.left[kNumSegments * kCursorStride], 1u) == (uint32_t)min((uint32_t)kNumSegments, gridDim.x) - 1u)
			{
				for(int s = 0; s < kNumSegments; ++s) { R.a.cursor[s * kCursorStride] = 0u; R.left[s * kCursorStride] = 0u; }
				R.left[kNumSegments * kCursorStride] = 0u;
			}
		}
	}
	if(blockIdx.x == 0 && threadIdx.x == 0)
	{
		atomicAdd(&a.stats->clock_cycles, __builtin_readcyclecounter() - clk_c0);
		atomicAdd(&a.stats->clock_ticks, __builtin_amdgcn_s_memrealtime() - clk_r0);
	}
	if(any_overflow) report_overflow(a.stats);
	if(STATS)
	{
		for(int off = 32; off > 0; off >>= 1)
		{
			st_nodes += __shfl_down(st_nodes, off);
			st_tris += __shfl_down(st_tris, off);
			st_hits += __shfl_down(st_hits, off);
			st_maxdepth = max(st_maxdepth, (uint32_t)__shfl_down((int)st_maxdepth, off));
			for(int i = 0; i < 8; ++i) wp[i] += __shfl_down(wp[i], off);
		}
		if(lane == 0)
		{
			for(int i = 0; i < 8; ++i) atomicAdd(&a.stats->wave_profile[i], wp[i]);
			atomicAdd(&a.stats->nodes, st_nodes);
			atomicAdd(&a.stats->tris, st_tris);
			atomicAdd(&a.stats->hits, st_hits);
			atomicMax(&a.stats->max_stack, st_maxdepth);
		}
	}
}

template <bool STATS, bool ANY = false>
__global__ __launch_bounds__(kTraceThreads, ((STATS || ANY) ? 4 : 6)) void k_trace(TraceArgs a) // hot variant: <= 80 VGPRs, 6 waves per SIMD
{
	trace_loop<STATS, ANY, false>(a);
}
template <bool STATS, bool VIEWER>
__global__ __launch_bounds__(kTraceThreads, (STATS ? 4 : 6)) void k_trace_camera(TraceCameraArgs K)
{
	trace_loop<STATS, false, true, VIEWER>(K.a);
}

}  // namespace adypt
