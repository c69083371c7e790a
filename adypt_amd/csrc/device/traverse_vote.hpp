// k_trace_vote — variant of k_trace (traverse.hpp) with wave-voted phase scheduling.
//
// Same per-ray semantics and arithmetic as k_trace (and therefore as shaders/traversal.glsl:14-255): every lane
// performs for its ray exactly the sequence  slab test of node 1, triangles of node 1, slab test of node 2, ... .
// What changes is WHEN the wave executes which step: each trip of the main loop the wave votes (__ballot/__popcll)
// and runs either the triangle phase (lanes holding a fetched triangle pair) or the node phase (lanes holding a
// fetched node and no pending triangles); lanes in the other state wait for their phase.  In k_trace both phases
// run every trip with whatever lanes happen to need them (triangle block at ~30 % lane utilisation).
// The loads of the next step are issued at the end of the current one (node: 5 x dwordx4, first triangle pair:
// 6 x dwordx4) and are consumed one or more trips later, so they are in flight while other lanes compute.
#pragma once
#include "traverse.hpp"

namespace adypt {

template <bool STATS>
__global__ __launch_bounds__(kTraceThreads) void k_trace_vote(TraceArgs a)
{
	extern __shared__ uint2 lds_stack[]; // [waves][lds_depth][64]
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	uint2 *my_stack = lds_stack + (size_t)wave * a.lds_depth * 64 + lane;
	const uint32_t total_lanes = gridDim.x * blockDim.x;
	uint2 *my_spill = a.spill + (blockIdx.x * blockDim.x + threadIdx.x);
	const int home = blockIdx.x & (kNumSegments - 1);

	if(blockIdx.x == 0 && threadIdx.x == 0)
	{
		unsigned long long total = 0;
		for(int s = 0; s < kNumSegments; ++s) total += a.count[s * kCursorStride];
		atomicAdd(&a.stats->rays, total);
	}

	bool active = false, node_valid = false, tri_valid = false;
	uint32_t ray = 0;
	F3 origin = f3(0, 0, 0), dir = f3(0, 0, 1), idir = f3(0, 0, 1);
	bool nx = false, ny = false, nz = false;
	uint32_t octinv = 7u;
	float tmin = 0.0f, hit_t = 1e9f, hit_u = 0.0f, hit_v = 0.0f;
	int32_t hit_idx = -1;
	int sp = 0;
	uint32_t ng_x = 0, ng_y = 0, tg_x = 0, tg_y = 0;
	uint32_t n_nodes = 0, n_tris = 0, hash = 0, max_depth = 0;
	bool overflow = false;
	// pipeline registers: the fetched-but-not-yet-tested node and triangle pair of this lane
	uint32_t node = 0;
	uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0, n2 = n0, n3 = n0, n4 = n0;
	float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, p2 = p0, q0 = p0, q1 = p0, q2 = p0;
	uint32_t tri0 = 0, tri1 = 0;
	bool two = false;

	unsigned long long st_nodes = 0, st_tris = 0, st_hits = 0;
	uint32_t st_maxdepth = 0;
	bool any_overflow = false, exhausted = false;
	uint32_t loc_next = 0, loc_end = 0;

	// choose the next node of this lane's ray (traversal.glsl:47-66 / 245-250) and issue its fetch
	auto fetch_next_node = [&]() {
		node_valid = true;
		if(ng_y <= 0x00ffffffu)
		{
			if(sp == 0) { node_valid = false; return; }
			--sp;
			const uint2 g = sp < a.lds_depth ? my_stack[sp * 64] : my_spill[(size_t)(sp - a.lds_depth) * total_lanes];
			ng_x = g.x; ng_y = g.y;
			asm volatile("" : "+v"(ng_x), "+v"(ng_y)); // complete the pop before the fetches below are issued
		}
		const uint32_t imask = ng_y;
		const uint32_t bit = 31u - (uint32_t)__builtin_clz(ng_y);
		ng_y &= ~(1u << bit);
		if(ng_y > 0x00ffffffu)
		{
			if(sp < a.stack_size)
			{
				if(sp < a.lds_depth) my_stack[sp * 64] = make_uint2(ng_x, ng_y);
				else my_spill[(size_t)(sp - a.lds_depth) * total_lanes] = make_uint2(ng_x, ng_y);
				++sp;
				if(STATS) max_depth = max(max_depth, (uint32_t)sp);
			}
			else overflow = true;
		}
		const uint32_t slot = (bit - 24u) ^ octinv;
		node = ng_x + (uint32_t)__builtin_popcount(imask & ~(0xffffffffu << slot));
		const uint4 *np = a.nodes + (size_t)node * 5;
		n0 = np[0]; n1 = np[1]; n2 = np[2]; n3 = np[3]; n4 = np[4];
	};
	auto fetch_tri_pair = [&]() {
		const uint32_t b0 = (uint32_t)__builtin_ctz(tg_y);
		tg_y &= tg_y - 1u;
		two = tg_y != 0;
		const uint32_t b1 = two ? (uint32_t)__builtin_ctz(tg_y) : b0;
		tg_y &= tg_y - 1u;
		tri0 = tg_x + b0; tri1 = tg_x + b1;
		const float4 *w0 = a.woop + (size_t)tri0 * 3, *w1 = a.woop + (size_t)tri1 * 3;
		p0 = w0[0]; p1 = w0[1]; p2 = w0[2];
		q0 = w1[0]; q1 = w1[1]; q2 = w1[2];
		tri_valid = true;
	};
	auto test_tri = [&](const float4 &m0, const float4 &m1, const float4 &m2, uint32_t tri) {
		const float toz = m0.w - dot3(origin, f3(m0.x, m0.y, m0.z));
		const float tidz = 1.0f / dot3(dir, f3(m0.x, m0.y, m0.z));
		const float tt = toz * tidz;
		const float tu = fmaf(tt, dot3(dir, f3(m1.x, m1.y, m1.z)), m1.w + dot3(origin, f3(m1.x, m1.y, m1.z)));
		const float tv = fmaf(tt, dot3(dir, f3(m2.x, m2.y, m2.z)), m2.w + dot3(origin, f3(m2.x, m2.y, m2.z)));
		if(tt > tmin && tt < hit_t && tu >= 0.0f && tu <= 1.0f && tv >= 0.0f && tu + tv <= 1.0f)
		{
			hit_t = tt; hit_u = tu; hit_v = tv; hit_idx = (int32_t)tri;
		}
	};

	for(;;)
	{
		bool want_node = false, want_tri = false;
		// ---------------- retire finished rays (traversal.glsl:247-254) ----------------
		if(active && !tri_valid && !node_valid)
		{
			const int32_t tri_id = hit_idx != -1 ? a.tri_indices[hit_idx] : -1;
			a.hit[ray] = make_float4(__int_as_float(tri_id), hit_u, hit_v, hit_t);
			any_overflow |= overflow;
			if(STATS)
			{
				if(a.ray_stats)
				{
					RayStats rs;
					rs.ref_idx = hit_idx; rs.nodes = n_nodes; rs.tris = n_tris; rs.hash = hash;
					rs.max_depth = overflow ? 0xffffffffu : max_depth; rs.pad0 = rs.pad1 = rs.pad2 = 0;
					a.ray_stats[ray] = rs;
				}
				st_nodes += n_nodes; st_tris += n_tris; st_hits += hit_idx != -1 ? 1 : 0;
				st_maxdepth = max(st_maxdepth, max_depth);
			}
			active = false;
		}
		// ---------------- refill idle lanes ----------------
		const unsigned long long idle = __ballot(!active);
		const uint32_t n_idle = (uint32_t)__popcll(idle);
		if((!exhausted || loc_next < loc_end) && (n_idle >= a.refill_min))
		{
			if(loc_next == loc_end)
			{
				uint32_t cb = 0, cn = 0;
				if(lane == 0) cn = fetch_rays(a.count, a.cursor, a.seg_cap, home, a.chunk, &cb);
				cn = __builtin_amdgcn_readfirstlane(cn);
				cb = __builtin_amdgcn_readfirstlane(cb);
				loc_next = cb; loc_end = cb + cn;
				if(cn == 0) exhausted = true;
			}
			const uint32_t begin = loc_next;
			const uint32_t got = min(n_idle, loc_end - loc_next);
			loc_next += got;
			const uint32_t my_rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
			if(!active && my_rank < got)
			{
				ray = begin + my_rank;
				const float4 ro = a.ray_o[ray];
				const float4 rd = a.ray_d[ray];
				const float ooeps = __uint_as_float((127u - 64u) << 23);
				dir = f3(rd.x, rd.y, rd.z);
				dir.x = fabsf(dir.x) > ooeps ? dir.x : (dir.x >= 0 ? ooeps : -ooeps);
				dir.y = fabsf(dir.y) > ooeps ? dir.y : (dir.y >= 0 ? ooeps : -ooeps);
				dir.z = fabsf(dir.z) > ooeps ? dir.z : (dir.z >= 0 ? ooeps : -ooeps);
				dir = normalize3(dir);
				idir = f3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
				nx = dir.x < 0; ny = dir.y < 0; nz = dir.z < 0;
				octinv = 7u - ((nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u));
				origin = f3(ro.x, ro.y, ro.z);
				tmin = ro.w;
				asm volatile("" : "+v"(origin.x), "+v"(origin.y), "+v"(origin.z), "+v"(tmin));
				hit_t = 1e9f; hit_u = 0.0f; hit_v = 0.0f; hit_idx = -1;
				sp = 0;
				ng_x = 0; ng_y = 0x80000000u; tg_x = 0; tg_y = 0;
				if(STATS) { n_nodes = 0; n_tris = 0; hash = 0x811c9dc5u; max_depth = 0; }
				overflow = false;
				active = true; tri_valid = false; node_valid = false;
				want_node = true; // the root
			}
		}
		const unsigned long long tri_mask = __ballot(active && tri_valid);
		const unsigned long long node_mask = __ballot(active && !tri_valid && node_valid);
		// nothing fetched, nothing to retire, nothing left in the queue: done.  (Lanes refilled this trip are active with
		// want_node set; lanes that just finished are active and get retired at the top of the next trip.)
		if((tri_mask | node_mask) == 0ull && __ballot(active) == 0ull && exhausted && loc_next == loc_end) break;
		const uint32_t n_tri = (uint32_t)__popcll(tri_mask);
		const bool tri_phase = n_tri != 0 && (n_tri >= a.tri_min || node_mask == 0ull);

		if(tri_phase)
		{
			// ---------------- triangle phase (traversal.glsl:213-243), one pair per trip ----------------
			if(active && tri_valid)
			{
				test_tri(p0, p1, p2, tri0);
				if(two) test_tri(q0, q1, q2, tri1);
				if(STATS) n_tris += two ? 2u : 1u;
				tri_valid = false;
				want_tri = tg_y != 0;
			}
		}
		else if(active && !tri_valid && node_valid)
		{
			// ---------------- node phase: slab tests of the fetched node (traversal.glsl:69-205) ----------------
			if(STATS) { ++n_nodes; hash = (hash * 0x01000193u) ^ node; }
			const uint32_t octinv4 = octinv * 0x01010101u;
			const uint32_t head_w = n0.w;
			const float aix = __uint_as_float((head_w & 0xffu) << 23) * idir.x;
			const float aiy = __uint_as_float(((head_w >> 8) & 0xffu) << 23) * idir.y;
			const float aiz = __uint_as_float(((head_w >> 16) & 0xffu) << 23) * idir.z;
			const float aox = (__uint_as_float(n0.x) - origin.x) * idir.x;
			const float aoy = (__uint_as_float(n0.y) - origin.y) * idir.y;
			const float aoz = (__uint_as_float(n0.z) - origin.z) * idir.z;
			ng_x = n1.x;
			tg_x = n1.y;
			uint32_t hitmask = 0;
#pragma unroll
			for(int g = 0; g < 2; ++g)
			{
				const uint32_t meta4 = g ? n1.w : n1.z;
				const uint32_t is_inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
				const uint32_t bit_index4 = (meta4 ^ (octinv4 & ((is_inner4 >> 4) * 0xffu))) & 0x1f1f1f1fu;
				const uint32_t child_bits4 = (meta4 >> 5) & 0x07070707u;
				const uint32_t qlox = g ? n2.y : n2.x, qloy = g ? n2.w : n2.z, qloz = g ? n3.y : n3.x;
				const uint32_t qhix = g ? n3.w : n3.z, qhiy = g ? n4.y : n4.x, qhiz = g ? n4.w : n4.z;
				const uint32_t slox = nx ? qhix : qlox, shix = nx ? qlox : qhix;
				const uint32_t sloy = ny ? qhiy : qloy, shiy = ny ? qloy : qhiy;
				const uint32_t sloz = nz ? qhiz : qloz, shiz = nz ? qloz : qhiz;
#pragma unroll
				for(int j = 0; j < 4; ++j)
				{
					const int sh = 8 * j;
					const float txmin = fmaf((float)((slox >> sh) & 0xffu), aix, aox);
					const float tymin = fmaf((float)((sloy >> sh) & 0xffu), aiy, aoy);
					const float tzmin = fmaf((float)((sloz >> sh) & 0xffu), aiz, aoz);
					const float txmax = fmaf((float)((shix >> sh) & 0xffu), aix, aox);
					const float tymax = fmaf((float)((shiy >> sh) & 0xffu), aiy, aoy);
					const float tzmax = fmaf((float)((shiz >> sh) & 0xffu), aiz, aoz);
					const float cmin = fmaxf(fmaxf(txmin, tymin), fmaxf(tzmin, tmin));
					const float cmax = fminf(fminf(txmax, tymax), fminf(tzmax, hit_t));
					if(cmin <= cmax) hitmask |= ((child_bits4 >> sh) & 0xffu) << ((bit_index4 >> sh) & 0xffu);
				}
			}
			ng_y = (hitmask & 0xff000000u) | (head_w >> 24);
			tg_y = hitmask & 0x00ffffffu;
			node_valid = false;
			want_tri = tg_y != 0;
			want_node = true;
		}
		// ---------------- the ONLY fetch site: issue this trip's loads back to back (consumed in a later trip) -----------
		if(want_tri) fetch_tri_pair();
		if(want_node) fetch_next_node();
	}

	if(any_overflow) atomicAdd(&a.stats->overflows, 1ull);
	if(STATS)
	{
		for(int off = 32; off > 0; off >>= 1)
		{
			st_nodes += __shfl_down(st_nodes, off);
			st_tris += __shfl_down(st_tris, off);
			st_hits += __shfl_down(st_hits, off);
			st_maxdepth = max(st_maxdepth, (uint32_t)__shfl_down((int)st_maxdepth, off));
		}
		if(lane == 0)
		{
			atomicAdd(&a.stats->nodes, st_nodes);
			atomicAdd(&a.stats->tris, st_tris);
			atomicAdd(&a.stats->hits, st_hits);
			atomicMax(&a.stats->max_stack, st_maxdepth);
		}
	}
}

}  // namespace adypt
