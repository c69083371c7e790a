// k_path — bounces 1 .. maxBounce-1 of a whole batch in ONE persistent launch: the reference's for(b < uMaxBounce) loop inside a single
// dispatch (shaders/pathtracer.glsl:107-202, src/Tracer/OglPathTracer.cpp:60) instead of [k_trace -> k_shade] per bounce.
//
// Why: every persistent traversal launch ends with its longest rays (~0.1 ms with the chip draining, profiles/r3_ablations_k_trace.txt item 14);
// seven of those per batch are 2-3 % of a 1-GPU batch and the whole strong-scaling loss of a pixel-tile shard; and k_shade, on its own,
// waits on its gathers with half the vector ALUs idle while ~1 GB of queue records per frame travel through HBM between the two kernels.
//
// How (results unchanged: per-path arithmetic is fetch_info / respond / finish_path of shade.hpp and the trip of traverse_trip.inc, and
// k_resolve applies the finished samples in frame order):
//   * a workgroup owns kPathSlots PATHS for their whole life, kept in an LDS table (path word, direction, throughput, and origin-or-hit:
//     40 bytes) — more paths than lanes, so a lane whose ray has finished finds the next ray waiting;
//   * a lane traverses the ray of one path slot.  When >= refill_min lanes of a wave are idle the wave DEPOSITS its finished rays (hit into
//     the slot, slot index onto the workgroup's to-shade list) and takes ready rays off the to-trace list — LDS only, a short spin lock;
//   * whichever wave finds >= shade_min deposited hits takes 64 of them and runs one bounce of Render() for them with all 64 lanes —
//     FetchInfo, the illum switch, accumulate on termination: exactly k_shade's work at k_shade's lane occupancy — while the other waves
//     of the CU keep traversing and hide its gather latency; its own rays wait, ten registers of their state (hit, node groups, pending
//     node, slot and stack pointer) parked in LDS for the round so that the shading code has them (one shading wave per workgroup at a time);
//   * a path that ends is replaced from the global queue (the bounce-1 rays k_shade_first wrote) by the shading wave itself: one device
//     atomic per shading round, issued BEFORE the gathers for the paths that are certain to end (miss, last bounce) so that its latency is hidden;
//   * the hit's triangle record is looked up by REFERENCE index (the traversal's own index: SceneArgs::triangles here is the per-reference copy
//     of the records, made once at upload), so the uTriIndices remap (traversal.glsl:253-254) — a dependent load in front of every deposit —
//     disappears from this kernel; the triangle id itself is never an output of these bounces.  A context without that copy (too large, or
//     the allocation failed) passes uTriIndices as PathArgs::tri_remap and the shading round applies it;
//   * a shading round runs every branch of Render()'s illum switch some lane needs, and the glossy lobe (two fp64 pow series and a sincos) or the
//     dielectric branch are needed by a few lanes of almost every round: a round that finds such hits among the 64 it took DEFERS them to a second
//     small ring (the class is a flag in the triangle record, known when the record arrives) and shades the others without those branches; when the
//     ring holds rare_min of them a round takes them together.  A grouping of the work only: every path is shaded by the same arithmetic;
//   * the optional sun-visibility query (pathtracer.glsl:132, commented out in the reference) stays inside the launch: a path that hits nothing keeps
//     its slot and its origin, takes the sun direction and the bounce index kPwShadow, and is traced once more — ending at its first accepted triangle
//     (traversal.glsl:257-494 per lane: the SUN variant of the kernel) — before the round that finds it again adds the sun term, or not;
//   * no inter-workgroup communication of any kind, so none of the cross-XCD visibility questions of a streaming queue.
// The kernel ends when the global queue is dry and every workgroup has finished the paths it holds.
#pragma once
#include "traverse.hpp"

namespace adypt {

#ifndef ADYPT_PATH_SLOTS
#define ADYPT_PATH_SLOTS 352
#endif
constexpr int kPathSlots = ADYPT_PATH_SLOTS;     // paths a workgroup holds: its 256 lanes' + those ready or waiting to be shaded (tuning: tools/build_variant.sh)
static_assert(kPathSlots >= kTraceThreads && kPathSlots <= 2 * kTraceThreads && kPathSlots % 32 == 0, "k_path: 256 <= slots <= 512");
constexpr int kTabFields = 10;                   // path word | direction | throughput | origin (to-trace) or hit (to-shade)
constexpr int kRareCap = 96;                     // entries of the ring of deferred hits (glossy lobe / dielectric): what does not fit is shaded at once
static_assert(kRareCap >= 64 && kRareCap % 8 == 0 && kRareCap + 64 <= kTraceThreads, "k_path: the deferred ring holds a round's worth (rare_min <= 64), keeps PathCtl 16-byte aligned behind it, and leaves the to-shade ring a full batch when every path waits");
constexpr int kParkDwords = 8;                   // per-lane ray state a shading wave parks in LDS for the round (node and triangle groups | hit distance, node, slot | stack pointer)
constexpr uint32_t kPwBounceShift = 26;          // path word in the table: bits 25..0 path id, 30..26 bounce index (kPwShadow: the ray is the path's sun-visibility query), 31 radiance parked
constexpr uint32_t kPwShadow = 31;               // (so the query needs max_bounce <= 31: tracer.hip keeps the launch-per-bounce pipeline otherwise)
constexpr uint32_t kRingMiss = 0x8000u;          // to-shade ring entry: the slot's ray hit nothing (its origin fields still hold the origin)
constexpr uint32_t kPwIdMask = (1u << kPwBounceShift) - 1u;
constexpr int64_t kPathMaxPaths = (int64_t)1 << kPwBounceShift; // batches with more paths keep the launch-per-bounce pipeline
enum { T_PW = 0, T_DX, T_DY, T_DZ, T_CX, T_CY, T_CZ, T_OX, T_OY, T_OZ };

struct PathCtl {                                 // workgroup control block in LDS (zeroed at start)
	uint32_t n_shade, n_trace, live, busy;         // deposited hits, ready rays, paths alive in this workgroup, (bit 0: a wave is shading | deferred hits << 16)   <- one 16-byte peek
	uint32_t h_shade, h_trace, rays, shaded;       // heads of the two rings (FIFO: no path waits behind younger ones)
	uint32_t lock, init_have, h_rare, pad[5];      // head of the ring of deferred hits (its count lives in `busy`)
};

struct PathArgs {
	const uint4 *nodes;
	const float4 *woop;
	const float *in_o; const float4 *in_d; const float *in_col; // the batch's ray queue as k_shade_first leaves it (12 / 16 / 12 bytes per path)
	RayStats *ray_stats;           // always null (traverse_trip.inc's per-ray record belongs to adypt_trace_rays)
	const int32_t *tri_remap;      // null: SceneArgs::triangles is the per-reference copy; else uTriIndices, applied in the shading round (traversal.glsl:253-254)
	const uint32_t *count;         // paths per queue segment: count[s * kCursorStride]
	uint32_t *cursor;              // fetch cursor per segment, zero at launch
	uint2 *spill;                  // [(stack_size - lds_depth)][total lanes]
	DeviceStats *stats;
	uint32_t seg_cap;
	int32_t stack_size, lds_depth;
	uint32_t refill_min, shade_min;
	uint32_t rare_min;             // deferred hits (glossy lobe / dielectric) a shading round waits for; 0 = no round defers anything
	uint32_t defer_max;            // a round defers such hits only when it holds at most this many of them (more: those branches are well occupied where they are)
	int32_t b0;                    // bounce index of the queue's rays (1: k_shade_first did bounce 0)
	float tmin;
};

inline size_t path_lds_bytes(int lds_depth)
{
	return (size_t)(kTraceThreads / 64) * (size_t)lds_depth * 64 * sizeof(uint2) + (size_t)kTabFields * kPathSlots * 4 + (size_t)kParkDwords * 64 * 4 + 2 * (size_t)kPathSlots * 2 +
	       (size_t)kRareCap * 2 + sizeof(PathCtl) + kTripTabBytes;
}

__device__ __forceinline__ void wg_lock(PathCtl *ctl, int lane)
{
	if(lane == 0)
	{
		uint32_t expect = 0u;
		while(!__hip_atomic_compare_exchange_strong(&ctl->lock, &expect, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
		{
			expect = 0u;
			__builtin_amdgcn_s_sleep(1);
		}
	}
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void wg_unlock(PathCtl *ctl, int lane)
{
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (one wave's LDS operations execute in order; this keeps the compiler from moving them)
	if(lane == 0) __hip_atomic_store(&ctl->lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t lane_rank(unsigned long long mask) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

#ifndef ADYPT_PATH_PRIO
#define ADYPT_PATH_PRIO 2
#endif
#ifndef ADYPT_PATH_WAVES
#define ADYPT_PATH_WAVES 6  // waves per SIMD the register allocation is held to (6: <= 80 VGPRs, like k_trace)
#endif

struct PathKernArgs { PathArgs a; FrameArgs f; SceneArgs sc; PixelArgs px; int count_stats; }; // the kernel's one parameter: offset 0 of the kernarg segment
__device__ __forceinline__ const PathKernArgs &rare_args()
{
	unsigned long long k = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("" : "+s"(k)); // (not hoisted out of the block that calls this, not merged with the by-value parameter)
	return *(const PathKernArgs *)(const __attribute__((address_space(4))) PathKernArgs *)k;
}

template <bool STATS, bool SUN>
__global__ __launch_bounds__(kTraceThreads, STATS ? 4 : ADYPT_PATH_WAVES) void k_path(PathKernArgs K)
{
	// The kernel's arguments are read in two ways.  What the traversal loop and the prologue need comes from the by-value parameter (the compiler
	// keeps those fields in SGPRs).  What only the shading round and the epilogue need — two dozen pointers and scalars — is loaded where it is
	// used, through the dispatch's kernarg segment behind a value the compiler cannot see through (rare_args()): held in SGPRs across the
	// persistent loop they would push the loop's own scalars into spills.
	const PathArgs &a = K.a;
	constexpr bool ANY = false;
	constexpr bool kOverflowPerRay = false;
	constexpr bool kUniformTmin = true; // every ray of the pass has the pass's tmin
	constexpr bool kTripShadowRays = SUN; // sun-visibility queries among the rays: they end at their first accepted triangle
	extern __shared__ uint2 lds_stack[]; // [waves][lds_depth][64] | path table [kTabFields][kPathSlots] | parking [kParkDwords][64] | to-shade, to-trace, deferred rings | PathCtl | triangle hand-out tables
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	uint2 *my_stack = lds_stack + (size_t)wave * a.lds_depth * 64 + lane;
	uint32_t *tab = (uint32_t *)(lds_stack + (size_t)(kTraceThreads / 64) * a.lds_depth * 64);
	uint32_t *park = tab + kTabFields * kPathSlots;
	uint16_t *to_shade = (uint16_t *)(park + kParkDwords * 64), *to_trace = to_shade + kPathSlots;
	uint16_t *to_rare = to_trace + kPathSlots;
	PathCtl *ctl = (PathCtl *)(to_rare + kRareCap);
	uint8_t *const trip_tab = (uint8_t *)(ctl + 1) + wave * 64; // the wave's triangle hand-out table (traverse_trip.inc, section B)
	const uint32_t total_lanes = gridDim.x * (uint32_t)kTraceThreads;
	const SpillColumn<true> my_spill(a.spill); // (addressed where it is used: traverse.hpp)
	const int home = blockIdx.x & (kNumSegments - 1);
	const float tmin = a.tmin;

	const unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_r0 = __builtin_amdgcn_s_memrealtime();
	if(threadIdx.x < sizeof(PathCtl) / 4) ((uint32_t *)ctl)[threadIdx.x] = 0u;
	__syncthreads();

	const uint32_t seg_len_lanes = lane < kNumSegments ? a.count[lane * kCursorStride] : 0u;
	uint32_t seg_done = 0;              // wave-uniform: segments found empty or exhausted
	for(int sgm = 0; sgm < kNumSegments; ++sgm)
		if(__builtin_amdgcn_readlane((int)seg_len_lanes, sgm) == 0) seg_done |= 1u << sgm;

	// a path of the global queue moves into table slot `slot`
	auto load_path = [&](const PathArgs &a, uint32_t idx, uint32_t slot) {
		const F3 o = ld3(a.in_o, idx);
		const float4 d4 = a.in_d[idx];
		const F3 c3 = ld3(a.in_col, idx);
		const uint32_t w = __float_as_uint(d4.w); // path word of the queue: bits 30..0 path id (< 2^26 here), bit 31 radiance parked
		tab[T_PW * kPathSlots + slot] = (w & (kPathParked | kPwIdMask)) | ((SUN && (w & kPathShadow) ? kPwShadow : (uint32_t)a.b0) << kPwBounceShift);
		tab[T_DX * kPathSlots + slot] = __float_as_uint(d4.x); tab[T_DY * kPathSlots + slot] = __float_as_uint(d4.y); tab[T_DZ * kPathSlots + slot] = __float_as_uint(d4.z);
		tab[T_CX * kPathSlots + slot] = __float_as_uint(c3.x); tab[T_CY * kPathSlots + slot] = __float_as_uint(c3.y); tab[T_CZ * kPathSlots + slot] = __float_as_uint(c3.z);
		tab[T_OX * kPathSlots + slot] = __float_as_uint(o.x); tab[T_OY * kPathSlots + slot] = __float_as_uint(o.y); tab[T_OZ * kPathSlots + slot] = __float_as_uint(o.z);
	};

	// ---------------- the workgroup's first paths: wave 0 reserves them, every thread moves its share into the table ----------------
	{
		uint32_t *init_idx = (uint32_t *)lds_stack; // (the stacks are not in use yet)
		if(wave == 0)
		{
			uint32_t have = 0;
			while(have < (uint32_t)kPathSlots)
			{
				uint32_t gb = 0, left = 0;
				const uint32_t gn = fetch_rays(seg_len_lanes, seg_done, a.cursor, a.seg_cap, home, (uint32_t)kPathSlots - have, &gb, &left);
				if(gn == 0) break;
				for(uint32_t i = (uint32_t)lane; i < gn; i += 64u) init_idx[have + i] = gb + i;
				have += gn;
			}
			if(lane == 0) { ctl->live = have; ctl->init_have = have; ctl->n_trace = have > (uint32_t)kTraceThreads ? have - (uint32_t)kTraceThreads : 0u; }
		}
		__syncthreads();
		const uint32_t have = ctl->init_have;
		uint32_t i0 = 0, i1 = 0;
		if(threadIdx.x < have) i0 = init_idx[threadIdx.x];
		if(threadIdx.x + kTraceThreads < have) i1 = init_idx[threadIdx.x + kTraceThreads];
		__syncthreads(); // the stack area is free again
		if(threadIdx.x < have) load_path(a, i0, threadIdx.x);
		if(threadIdx.x + kTraceThreads < have) { load_path(a, i1, threadIdx.x + kTraceThreads); to_trace[threadIdx.x] = (uint16_t)(threadIdx.x + kTraceThreads); }
		__syncthreads(); // slots beyond the lanes are on the to-trace list before any wave looks at it
	}

	// per-lane ray state (the names traverse_trip.inc works on)
	bool active = false;
	uint32_t ray = threadIdx.x;                  // the path slot whose ray this lane traverses
	bool setup = threadIdx.x < ctl->init_have;   // the lane starts the ray of slot `ray` at the top of the loop
	V2 od_x = v2(0, 0), od_y = v2(0, 0), od_z = v2(0, 1);
	F3 idir = f3(0, 0, 1);
	bool nx = false, ny = false, nz = false;
	uint32_t octinv = 7u;
	float hit_t = 1e9f;
	int32_t hit_idx = -1; // (STATS only: the hit itself — reference index, u, v — lives in the path's table slot, below)
	int sp = 0;
	uint32_t ng_x = 0, ng_y = 0, tg_x = 0, tg_y = 0;
	uint32_t n_nodes = 0, n_tris = 0, hash = 0, max_depth = 0;
	bool overflow = false;
	bool flush = false;
	bool pending = false;
	uint32_t node = 0, depth_after_push = 0;
	bool push_overflow = false;
	unsigned long long st_nodes = 0, st_tris = 0, st_hits = 0;
	uint32_t st_maxdepth = 0;
	bool any_overflow = false;
	bool shadow = false; // SUN: the lane's ray is a sun-visibility query
	uint32_t wave_rays = 0, wave_shaded = 0, wave_bad = 0; // wave-uniform totals, added up per workgroup at the end
	unsigned long long wp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	auto wave_event = [&](int slot) {
		const unsigned long long m = __ballot(true);
		if(lane == (int)__builtin_ctzll(m)) { wp[slot] += 1; wp[slot + 1] += (unsigned long long)__popcll(m); }
	};

	// The part of the ray setup (traversal.glsl:16-23) that is a function of the path slot alone: origin and direction as the table holds them
	// while the ray is traced.
	auto aim = [&]() {
		const float ox = __uint_as_float(tab[T_OX * kPathSlots + ray]), oy = __uint_as_float(tab[T_OY * kPathSlots + ray]), oz = __uint_as_float(tab[T_OZ * kPathSlots + ray]);
		F3 dir = f3(__uint_as_float(tab[T_DX * kPathSlots + ray]), __uint_as_float(tab[T_DY * kPathSlots + ray]), __uint_as_float(tab[T_DZ * kPathSlots + ray]));
		const float ooeps = __uint_as_float((127u - 64u) << 23);
		dir.x = fabsf(dir.x) > ooeps ? dir.x : (dir.x >= 0 ? ooeps : -ooeps);
		dir.y = fabsf(dir.y) > ooeps ? dir.y : (dir.y >= 0 ? ooeps : -ooeps);
		dir.z = fabsf(dir.z) > ooeps ? dir.z : (dir.z >= 0 ? ooeps : -ooeps);
		dir = normalize3(dir);
		idir = f3(rcp_ieee(dir.x), rcp_ieee(dir.y), rcp_ieee(dir.z));
		nx = dir.x < 0; ny = dir.y < 0; nz = dir.z < 0;
		octinv = 7u - ((nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u));
		od_x = v2(ox, dir.x); od_y = v2(oy, dir.y); od_z = v2(oz, dir.z);
	};

	// the two base pointers of the trip's loads, pinned for the whole loop (as plain kernel arguments the compiler re-loads them from the kernarg
	// segment inside the trip: a scalar load and its wait in front of every node / triangle fetch) — as GLOBAL pointers: laundered as generic ones the
	// eight loads of a trip became flat_load, which also passes through the LDS queue (-0.6 %, profiles/r5_ablations.txt 6)
	GlobalF4 trip_woop = (GlobalF4)a.woop;
	GlobalU4 trip_nodes = (GlobalU4)a.nodes;
	asm volatile("" : "+s"(trip_woop), "+s"(trip_nodes));

	// Shape of the loop: NO `continue`.  Every iteration runs top to bottom — start rays, exchange, trip — and a wave-uniform flag skips the
	// trip when the exchange handed out rays (they are started first) or the wave holds none.  With `continue` edges around the trip the
	// loop-carried ray state lived in two register sets, one for the blocks in front of the trip and one for the trip, with 10 copies at the
	// loop's latch and more at the trip's head: ~17 of a trip's ~340 vector instructions (profiles/history/r5_trip_budget.json).
	bool any_setup = __ballot(setup) != 0ull; // wave-uniform: some lane is about to start a ray (kept instead of a vote per iteration)
	for(;;)
	{
		// ---------------- start the rays of the slots the lanes have just taken (traversal.glsl:16-35) ----------------
		if(any_setup)
		{
			const unsigned long long starting = __ballot(setup);
			any_setup = false;
			asm volatile("; ADYPT_MARK setup_begin"); // (comments in the assembly: tools/instruction_mix.py weights the blocks of the loop by how often they run)
			if(setup)
			{
				aim();
				// The hit of the ray in flight lives in the origin fields of its slot (the origin is in registers from here on): written when a triangle
				// is accepted (ADYPT_TRIP_TAKE_HIT below), read by the shading round — never held in registers, never parked, nothing to deposit.  "No hit"
				// (traversal.glsl:30) is hit_t still at its start value when the ray is deposited: a bit of the ring entry, and the fields keep the origin
				if(SUN) shadow = ((tab[T_PW * kPathSlots + ray] >> kPwBounceShift) & 31u) == kPwShadow;
				hit_t = 1e9f; hit_idx = -1;
				sp = 0;
				ng_x = 0; ng_y = 0x80000000u; tg_x = 0; tg_y = 0;
				if(STATS) { n_nodes = 0; n_tris = 0; hash = 0x811c9dc5u; max_depth = 0; }
				overflow = false; pending = false; push_overflow = false; depth_after_push = 0;
				active = true;
				setup = false;
			}
			wave_rays += (uint32_t)__popcll(starting);
			asm volatile("; ADYPT_MARK setup_end");
		}

		// ---------------- exchange with the workgroup: deposit finished rays, shade a batch, take ready rays ----------------
		bool skip_trip = false; // wave-uniform
		const unsigned long long idle = __ballot(!active);
		const uint32_t n_idle = (uint32_t)__popcll(idle);
		if(n_idle >= a.refill_min)
		{
			const unsigned long long fl = __ballot(flush);
			const uint32_t n_flush = (uint32_t)__popcll(fl);
			// a look at the lists without the lock (a hint: everything is decided again under it)
			asm volatile("" ::: "memory"); // (read the control block afresh)
			const uint4 pk = *(const uint4 *)&ctl->n_shade;
			const uint32_t pk_shade = uni(pk.x), pk_trace = uni(pk.y), pk_live = uni(pk.z), pk_busy = uni(pk.w) & 1u, pk_rare = uni(pk.w) >> 16;
			const uint32_t pk_thr = min(a.shade_min, max(1u, pk_live >> 2));
			const bool pk_full = pk_thr == a.shade_min && pk_live >= a.shade_min + a.rare_min; // (the rule under the lock, below)
			const bool pk_ready = pk_full ? (pk_shade + n_flush >= pk_thr || (pk_rare != 0u && pk_rare >= a.rare_min)) : pk_shade + pk_rare + n_flush >= pk_thr;
			if(n_flush != 0u || pk_trace != 0u || (pk_ready && !pk_busy))
			{
				// Everything in this block runs at raised issue priority: while a wave is in here its rays do not advance, and at the fair share
				// of a SIMD's issue slots (1 / 6) the block's few hundred instructions would keep it away from them several times longer
				__builtin_amdgcn_s_setprio(ADYPT_PATH_PRIO);
				asm volatile("; ADYPT_MARK exchange_begin");
				// (deposit: the hit, by reference index, is in the slot already)
				const uint32_t fl_rank = lane_rank(fl), idle_rank = lane_rank(idle);
				auto ring = [](uint32_t i) { return i >= (uint32_t)kPathSlots ? i - (uint32_t)kPathSlots : i; };
				wg_lock(ctl, lane);
				uint32_t n_s = uni(ctl->n_shade), h_s = uni(ctl->h_shade), n_t = uni(ctl->n_trace), h_t = uni(ctl->h_trace);
				const uint32_t lv = uni(ctl->live);
				if(flush) to_shade[ring(h_s + n_s + fl_rank)] = (uint16_t)(ray | (hit_t == 1e9f ? kRingMiss : 0u)); // (an accepted triangle is nearer than the start value)
				n_s += n_flush;
				flush = false;
				const uint32_t thr = min(a.shade_min, max(1u, lv >> 2)); // fewer than 4 batches of paths left: smaller batches, down to single paths
				// Which hits the round takes.  While the workgroup holds 4 full batches or more (thr == shade_min): the deferred ones once rare_min of
				// them wait, else the oldest 64 of the to-shade ring.  Below that (the launch is ending): whatever waits in either ring, deferred first.
				// (live >= shade_min + rare_min: with every path of the workgroup waiting in the two rings, one of them has reached its threshold)
				const uint32_t bw = uni(ctl->busy);
				uint32_t n_r = bw >> 16, h_r = uni(ctl->h_rare);
				auto ring_r = [](uint32_t i) { return i >= (uint32_t)kRareCap ? i - (uint32_t)kRareCap : i; };
				const bool full_thr = thr == a.shade_min && lv >= a.shade_min + a.rare_min;
				uint32_t take_r = 0, take_s = 0;
				if((bw & 1u) == 0u) // one shading wave per workgroup at a time: one parking area
				{
					if(full_thr)
					{
						if(n_r != 0u && n_r >= a.rare_min) take_r = min(64u, n_r);
						else if(n_s >= thr) take_s = min(64u, n_s);
					}
					else if(n_s + n_r >= thr && n_s + n_r != 0u) { take_r = min(64u, n_r); take_s = min(64u - take_r, n_s); }
				}
				const uint32_t take = take_r + take_s;
				const bool do_shade = take != 0u;
				const bool may_defer = full_thr && take_r == 0u && a.rare_min != 0u; // (wave-uniform) a round of the to-shade ring's hits, not at the launch's end
				uint32_t sslot = 0;
				if(do_shade)
				{
					if((uint32_t)lane < take_r) sslot = to_rare[ring_r(h_r + (uint32_t)lane)]; // (deferred hits: never a miss)
					else if((uint32_t)lane < take) sslot = to_shade[ring(h_s + (uint32_t)lane - take_r)];
					h_s = ring(h_s + take_s); n_s -= take_s;
					h_r = ring_r(h_r + take_r); n_r -= take_r;
					if(lane == 0) { ctl->h_shade = h_s; ctl->h_rare = h_r; ctl->busy = 1u | (n_r << 16); }
				}
				else
				{
					const uint32_t got = min(n_idle, n_t);
					if(!active && idle_rank < got) { ray = to_trace[ring(h_t + idle_rank)]; setup = true; }
					if(lane == 0 && got) { ctl->n_trace = n_t - got; ctl->h_trace = ring(h_t + got); }
				}
				if(lane == 0) ctl->n_shade = n_s;
				wg_unlock(ctl, lane);

				if(do_shade)
				{
					asm volatile("; ADYPT_MARK shade_begin");
					const PathKernArgs &R = rare_args();
					const PathArgs &a = R.a; const FrameArgs &f = R.f; const SceneArgs &sc = R.sc; const PixelArgs &px = R.px; const int count_stats = R.count_stats;
					// ---------------- one iteration of Render()'s loop (pathtracer.glsl:107-202) for `take` paths, one per lane ----------------
					// The wave's own rays wait.  What of their state is not a function of the path table is parked in LDS for the round: the shading
					// code then has the registers the traversal loop lives in, and the loop itself stays register-allocated as in k_trace.
					// (an address that belongs to this rare block is computed from a value the compiler cannot see through, or it computes it once
					// before the persistent loop and keeps it in a register the traversal loop needs)
					uint32_t lane_here = (uint32_t)lane;
					asm volatile("" : "+v"(lane_here));
					uint32_t *pk_lane = park + lane_here * 4;
					*(uint4 *)(pk_lane + 0 * 256) = make_uint4(ng_x, ng_y, tg_x, tg_y);
					*(uint4 *)(pk_lane + 1 * 256) = make_uint4(__float_as_uint(hit_t), node, ray | ((uint32_t)sp << 16), STATS ? (uint32_t)hit_idx : 0u);
					asm volatile("" ::: "memory");
					bool have = (uint32_t)lane < take;
					const bool miss = (sslot & kRingMiss) != 0u;
					sslot &= kRingMiss - 1u;
					uint32_t pw = 0;
					F3 dir = f3(0, 0, 1), color = f3(0, 0, 0), origin = f3(0, 0, 0), ret = f3(0, 0, 0), ret_in = f3(0, 0, 0);
					int32_t tri_idx = -1;
					float tu = 0.0f, tv = 0.0f;
					if(have)
					{
						pw = tab[T_PW * kPathSlots + sslot];
						dir = f3(__uint_as_float(tab[T_DX * kPathSlots + sslot]), __uint_as_float(tab[T_DY * kPathSlots + sslot]), __uint_as_float(tab[T_DZ * kPathSlots + sslot]));
						color = f3(__uint_as_float(tab[T_CX * kPathSlots + sslot]), __uint_as_float(tab[T_CY * kPathSlots + sslot]), __uint_as_float(tab[T_CZ * kPathSlots + sslot]));
						if(!miss) { tri_idx = (int32_t)tab[T_OX * kPathSlots + sslot]; tu = __uint_as_float(tab[T_OY * kPathSlots + sslot]); tv = __uint_as_float(tab[T_OZ * kPathSlots + sslot]); }
					}
					const int b = (int)((pw >> kPwBounceShift) & 31u);
					const int pi = (int)(pw & kPwIdMask);
					bool parked = (pw & kPathParked) != 0u;
					const bool last = b + 1 >= f.max_bounce; // the path ends with this iteration whatever it hits: respond() stops after the emission (the switch's outputs have no reader)
					// SUN: what came back is the path's sun-visibility query (it ends the path either way: `last` holds, kPwShadow + 1 >= max_bounce) | the path escapes
					// and its query is still to be traced (it keeps its slot)
					const bool query_back = SUN && (uint32_t)b == kPwShadow, query_next = SUN && f.sun_query != 0 && !query_back && have && tri_idx == -1;
					// Everything a round waits for that does not depend on another fetch is issued up front, back to back — the hit's triangle record, the
					// reservation of the paths that replace those certain to end here (miss, or last bounce), the radiance parked so far, the pixel's shift
					// bytes and the frame's Sobol point of this bounce — and lands behind ONE wait; the material and the texels follow (two more).  A shading
					// wave is the one serial resource of its workgroup (64 paths every ~13 us at the bench rate): as the code stood, eight waits in a row.
					// (the destinations are "defined" by an empty asm, not zero-filled: a zero fill is a copy after the conditional load, and the copy waits)
					const int frame = (int)((uint32_t)pi / (uint32_t)f.n_local_px);
					const int L = pi - frame * f.n_local_px;
					TriCore tc;
#pragma unroll
					for(int i = 0; i < 20; ++i) asm volatile("" : "=v"(tc.v[i]));
					if(have && tri_idx != -1 && !query_back)
					{
						if(a.tri_remap) tri_idx = a.tri_remap[tri_idx]; // (contexts without the per-reference copy of the triangle records)
						tc = load_tri_core(sc, tri_idx);
					}
					// (a path certain to end is never deferred — on its last bounce it does not run the illum switch at all — so the count below holds after the deferral)
					const bool sure = have && !query_next && (tri_idx == -1 || last); // (an escaped path whose query is still to be traced stays)
					const uint32_t n_sure = (uint32_t)__popcll(__ballot(sure));
					const bool early = n_sure != 0u && !((seg_done >> home) & 1u);
					uint32_t rel;
					asm volatile("" : "=v"(rel));
					{
						// (the index passes through a vector register the compiler cannot see through: for a uniform address its atomic optimizer wraps the
						// one-lane atomic in a lane scan whose broadcast waits for the result on the spot — the latency this early issue is there to hide)
						uint32_t cur = (uint32_t)home * (uint32_t)kCursorStride;
						asm volatile("" : "+v"(cur));
						if(early && lane == 0) rel = atomicAdd(&a.cursor[cur], n_sure);
					}
					float4 r4;
					ADYPT_DEF4(r4);
					uint32_t shift16;
					float sob_x, sob_y;
					asm volatile("" : "=v"(shift16), "=v"(sob_x), "=v"(sob_y));
					if(have)
					{
						if(parked) r4 = f.done[pi];
						shift16 = *(const uint16_t *)(px.shift + (size_t)L * 2);
						const float2 sp2 = *(const float2 *)(f.sobol + frame * 64 + 2 * b);
						sob_x = sp2.x; sob_y = sp2.y;
					}
					if(may_defer)
					{
						const bool rare = have && tri_idx != -1 && !last && __float_as_uint(tc.v[19]) != 0u; // glossy lobe or dielectric (tracer.hip: the record's class word)
						const unsigned long long rm = __ballot(rare);
						// (where most hits are of that kind — a room of glossy walls — moving them only empties this round and overflows the ring: -31 % measured)
						if(rm != 0ull && (uint32_t)__popcll(rm) <= a.defer_max)
						{
							const uint32_t room = (uint32_t)kRareCap - n_r, rr = lane_rank(rm);
							const bool defer = rare && rr < room;
							if(defer) to_rare[ring_r(h_r + n_r + rr)] = (uint16_t)sslot; // (this wave owns the ring while it is the shading wave; the count is published under the lock below)
							n_r += min((uint32_t)__popcll(rm), room);
							have = have && !defer;
						}
					}

					bool alive = have, shaded = false, bad_mat = false;
					if(have)
					{
						if(parked) ret = f3(r4.x, r4.y, r4.z);
						ret_in = ret;
						if(query_next)
						{
							// pathtracer.glsl:132 with the occlusion query enabled: from the position the path escapes from (its slot's origin fields: nothing
							// was accepted, so nothing overwrote them) towards the sun; throughput and radiance wait in the slot
							dir = f3(f.sun_query_dir[0], f.sun_query_dir[1], f.sun_query_dir[2]);
							origin = f3(__uint_as_float(tab[T_OX * kPathSlots + sslot]), __uint_as_float(tab[T_OY * kPathSlots + sslot]), __uint_as_float(tab[T_OZ * kPathSlots + sslot]));
						}
						else if(tri_idx == -1)
						{
							ret = fma3(color, f3(f.sun[0], f.sun[1], f.sun[2]), ret); // pathtracer.glsl:130-135 (a query that came back empty: the sun is visible)
							alive = false;
						}
						else if(query_back) alive = false; // the query hit something: no sun term
						else
						{
							const SurfaceInfo si = fetch_info(f, sc, tc, tri_idx, tu, tv);
							origin = si.origin;
							if(si.bad_mat) { alive = false; bad_mat = true; }
							else
							{
								shaded = true;
								const RngPoint rng{unorm8_to_float(shift16 & 0xffu), unorm8_to_float(shift16 >> 8), sob_x, sob_y};
								alive = respond(f, si, rng, b, dir, color, ret);
							}
						}
						if(!alive) // main()'s clamp (pathtracer.glsl:224); the running mean is k_resolve's, in frame order (a k_path pass is always batched)
							f.done[pi] = make_float4(gl_min(ret.x, f.clamp), gl_min(ret.y, f.clamp), gl_min(ret.z, f.clamp), 1.0f);
					}
					wave_bad += (uint32_t)__popcll(__ballot(bad_mat));
					if(count_stats) wave_shaded += (uint32_t)__popcll(__ballot(shaded));
					if(alive)
					{
						if(__float_as_uint(ret.x) != __float_as_uint(ret_in.x) || __float_as_uint(ret.y) != __float_as_uint(ret_in.y) ||
						   __float_as_uint(ret.z) != __float_as_uint(ret_in.z))
						{
							f.done[pi] = make_float4(ret.x, ret.y, ret.z, 0.0f); // radiance picked up on the way: parked per path (shade.hpp)
							parked = true;
						}
						tab[T_PW * kPathSlots + sslot] = (uint32_t)pi | ((query_next ? kPwShadow : (uint32_t)(b + 1)) << kPwBounceShift) | (parked ? kPathParked : 0u);
						tab[T_DX * kPathSlots + sslot] = __float_as_uint(dir.x); tab[T_DY * kPathSlots + sslot] = __float_as_uint(dir.y); tab[T_DZ * kPathSlots + sslot] = __float_as_uint(dir.z);
						tab[T_CX * kPathSlots + sslot] = __float_as_uint(color.x); tab[T_CY * kPathSlots + sslot] = __float_as_uint(color.y); tab[T_CZ * kPathSlots + sslot] = __float_as_uint(color.z);
						tab[T_OX * kPathSlots + sslot] = __float_as_uint(origin.x); tab[T_OY * kPathSlots + sslot] = __float_as_uint(origin.y); tab[T_OZ * kPathSlots + sslot] = __float_as_uint(origin.z);
					}
					// ---------------- paths that ended: their slots take the next paths of the global queue ----------------
					const bool dead = have && !alive;
					const unsigned long long dl = __ballot(dead);
					const uint32_t n_dead = (uint32_t)__popcll(dl), dead_rank = lane_rank(dl);
					bool repl = false;
					uint32_t idx = 0;
					{
						uint32_t served = 0;
						if(early)
						{
							rel = uni(rel);
							const uint32_t seg_len = (uint32_t)__builtin_amdgcn_readlane((int)seg_len_lanes, home);
							if(rel < seg_len)
							{
								served = min(n_sure, seg_len - rel);
								if(dead && dead_rank < served) { idx = (uint32_t)home * a.seg_cap + rel + dead_rank; repl = true; }
							}
							if(rel + n_sure >= seg_len) seg_done |= 1u << home;
						}
						while(served < n_dead) // (wave-uniform) the rest: paths that ended unexpectedly, or the home segment has run out
						{
							uint32_t gb = 0, left = 0;
							const uint32_t gn = fetch_rays(seg_len_lanes, seg_done, a.cursor, a.seg_cap, home, n_dead - served, &gb, &left);
							if(gn == 0) break;
							if(dead && dead_rank >= served && dead_rank < served + gn) { idx = gb + (dead_rank - served); repl = true; }
							served += gn;
						}
					}
					if(repl) load_path(a, idx, sslot);
					// every global store of this round (finished samples, parked radiance) has left before another wave of the workgroup can
					// shade these paths again (same CU, same L1: no more is needed inside a workgroup)
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					const bool push = have && (alive || repl);
					const unsigned long long pm = __ballot(push);
					const uint32_t n_push = (uint32_t)__popcll(pm), push_rank = lane_rank(pm);
					const uint32_t n_lost = n_dead - (uint32_t)__popcll(__ballot(repl));
					// the wave's own rays come back: the parked registers, and origin / direction / octant from the table as at the ray's start
					{
						const uint4 p0 = *(const uint4 *)(pk_lane + 0 * 256), p1 = *(const uint4 *)(pk_lane + 1 * 256);
						ng_x = p0.x; ng_y = p0.y; tg_x = p0.z; tg_y = p0.w;
						hit_t = __uint_as_float(p1.x); node = p1.y; ray = p1.z & 0xffffu; sp = (int)(p1.z >> 16);
						if(STATS) hit_idx = (int32_t)p1.w;
						// (origin, direction, inverse direction and octant of the wave's own rays stayed in their registers: recomputing them from the table after
						// every round — 77 vector instructions, a square root and four reciprocals among them — cost 0.9 %; the price is one register pair the
						// round spills to scratch)
					}
					wg_lock(ctl, lane);
					n_t = uni(ctl->n_trace); h_t = uni(ctl->h_trace);
					if(push) to_trace[ring(h_t + n_t + push_rank)] = (uint16_t)sslot;
					n_t += n_push;
					const uint32_t got = min(n_idle, n_t); // and the wave's own idle lanes take the oldest ready rays
					if(!active && idle_rank < got) { ray = to_trace[ring(h_t + idle_rank)]; setup = true; }
					if(lane == 0) { ctl->n_trace = n_t - got; ctl->h_trace = ring(h_t + got); if(n_lost) ctl->live = ctl->live - n_lost; ctl->busy = n_r << 16; }
					wg_unlock(ctl, lane);
					asm volatile("; ADYPT_MARK shade_end");
				}
				asm volatile("; ADYPT_MARK exchange_end");
				__builtin_amdgcn_s_setprio(0);
				any_setup = __ballot(setup) != 0ull;
				skip_trip = any_setup; // the rays just taken are started first
			}
		}
		const unsigned long long live = ~idle; // (every lane of the wave runs this loop; `active` has not changed since the vote above)
		if(!skip_trip)
		{
			if(STATS && lane == 0) { wp[0] += 1; wp[1] += (unsigned long long)__popcll(live); }
			if(live == 0ull)
			{
				if(STATS && lane == 0) wp[7] += 1;
				if(uni(__hip_atomic_load(&ctl->live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0u) break; // the global queue is dry and the workgroup's last path has ended
				__builtin_amdgcn_s_sleep(8);
				skip_trip = true;
			}
		}
#define ADYPT_TRIP_SHADOW shadow
#define ADYPT_TRIP_TAKE_HIT(u, v, idx) { tab[T_OX * kPathSlots + ray] = (idx); tab[T_OY * kPathSlots + ray] = __float_as_uint(u); tab[T_OZ * kPathSlots + ray] = __float_as_uint(v); if(STATS) hit_idx = (int32_t)(idx); }
		if(!skip_trip)
#include "traverse_trip.inc"
#undef ADYPT_TRIP_TAKE_HIT
#undef ADYPT_TRIP_SHADOW
	}

	// ---------------- totals: per wave -> per workgroup (LDS) -> one device atomic per workgroup ----------------
	const PathKernArgs &E = rare_args();
	const PixelArgs &px = E.px;
	if(lane == 0) { atomicAdd(&ctl->rays, wave_rays); if(wave_shaded) atomicAdd(&ctl->shaded, wave_shaded); if(wave_bad) atomicAdd(&px.stats->bad_materials, (unsigned long long)wave_bad); }
	if(any_overflow) report_overflow(E.a.stats);
	__syncthreads();
	if(threadIdx.x == 0)
	{
		atomicAdd(&E.a.stats->rays, (unsigned long long)ctl->rays);
		atomicAdd(&E.a.stats->path_rays, (unsigned long long)ctl->rays);
		if(ctl->shaded) { atomicAdd(&px.stats->shaded, (unsigned long long)ctl->shaded); atomicAdd(&px.stats->path_shaded, (unsigned long long)ctl->shaded); }
		if(blockIdx.x == 0)
		{
			atomicAdd(&E.a.stats->clock_cycles, __builtin_readcyclecounter() - clk_c0);
			atomicAdd(&E.a.stats->clock_ticks, __builtin_amdgcn_s_memrealtime() - clk_r0);
		}
	}
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
			for(int i = 0; i < 8; ++i) atomicAdd(&E.a.stats->wave_profile[i], wp[i]);
			atomicAdd(&E.a.stats->nodes, st_nodes); atomicAdd(&E.a.stats->path_nodes, st_nodes);
			atomicAdd(&E.a.stats->tris, st_tris); atomicAdd(&E.a.stats->path_tris, st_tris);
			atomicAdd(&E.a.stats->hits, st_hits); atomicAdd(&E.a.stats->path_hits, st_hits);
			atomicMax(&E.a.stats->max_stack, st_maxdepth);
		}
	}
}

}  // namespace adypt
