"""Measurement variant of k_path (NOT product code): rewrites a scratch copy of device/path.hpp so that every wave accumulates the shader
cycles it spends in the exchange block, in shading rounds, asleep and waiting for the workgroup lock, and counts its trips / exchanges /
rounds; the sums land in DeviceStats::wave_profile (read by tools/path_profile.py).  Applied by tools/build_variant.sh:
    tools/build_variant.sh prof --transform adypt_amd/csrc/measure/k_path_profile.py
The timers themselves cost ~10 % of the kernel's speed: read the shares, not the absolute time."""
import sys
p = sys.argv[1] + "/path.hpp"
s = open(p).read()


def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old[:80])
    s = s.replace(old, new)


rep("""	uint32_t wave_rays = 0, wave_shaded = 0, wave_bad = 0; // wave-uniform totals, added up per workgroup at the end
""", """	uint32_t wave_rays = 0, wave_shaded = 0, wave_bad = 0; // wave-uniform totals, added up per workgroup at the end
	unsigned long long pf_t0 = __builtin_readcyclecounter(), pf_total0 = pf_t0, pf_exch = 0, pf_shade = 0, pf_idle = 0, pf_lock = 0;
	uint32_t pf_rounds = 0, pf_take = 0, pf_exchanges = 0, pf_trips = 0, pf_trip_lanes = 0, pf_busy = 0, pf_backlog = 0;
""")
rep("""		if(n_idle >= a.refill_min)
		{
			const unsigned long long fl = __ballot(flush);
""", """		if(n_idle >= a.refill_min)
		{
			pf_t0 = __builtin_readcyclecounter();
			const unsigned long long fl = __ballot(flush);
""")
rep("""				wg_lock(ctl, lane);
				uint32_t n_s = uni(ctl->n_shade)""", """				pf_exchanges += 1;
				{ const unsigned long long l0 = __builtin_readcyclecounter();
				wg_lock(ctl, lane);
				pf_lock += __builtin_readcyclecounter() - l0; }
				uint32_t n_s = uni(ctl->n_shade)""")
rep("""				const uint32_t take = take_r + take_s;
				const bool do_shade = take != 0u;
""", """				const uint32_t take = take_r + take_s;
				const bool do_shade = take != 0u;
				if(!do_shade && (bw & 1u) != 0u && n_s + n_r >= thr && n_s + n_r != 0u) pf_busy += 1; // would have shaded, but another wave of the workgroup is
				pf_backlog += n_s + n_r;
""")
rep("""				if(do_shade)
				{
					asm volatile("; ADYPT_MARK shade_begin");""", """				unsigned long long pf_s0 = __builtin_readcyclecounter();
				if(do_shade)
				{
					pf_rounds += 1; pf_take += take;
					asm volatile("; ADYPT_MARK shade_begin");""")
rep("""				asm volatile("; ADYPT_MARK exchange_end");
""", """				asm volatile("; ADYPT_MARK exchange_end");
				if(do_shade) pf_shade += __builtin_readcyclecounter() - pf_s0;
				pf_exch += __builtin_readcyclecounter() - pf_t0;
""")
rep("""				__builtin_amdgcn_s_sleep(8);
				skip_trip = true;
""", """				{ const unsigned long long i0 = __builtin_readcyclecounter();
				__builtin_amdgcn_s_sleep(8);
				pf_idle += __builtin_readcyclecounter() - i0; }
				skip_trip = true;
""")
rep("""				if(gn == 0) break;
				for(uint32_t i = (uint32_t)lane; i < gn; i += 64u) init_idx[have + i] = gb + i;
""", """				if(gn == 0) break;
				for(uint32_t i = (uint32_t)lane; i < gn; i += 64u) init_idx[have + i] = gb + i;
""")
# the tail: 100 MHz ticks of the launch's start, of the moment the FIRST wave finds the global queue dry, and of the last wave's end — in the
# otherwise unused path_hits / path_nodes / path_tris counters of the un-instrumented kernel (tools/path_profile.py reads them)
rep("""							if(gn == 0) break;
							if(dead && dead_rank >= served""", """							if(gn == 0) { if(lane == 0) atomicCAS(&a.stats->path_nodes, 0ull, (unsigned long long)__builtin_amdgcn_s_memrealtime()); break; }
							if(dead && dead_rank >= served""")
rep("""	const unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_r0 = __builtin_amdgcn_s_memrealtime();
""", """	const unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_r0 = __builtin_amdgcn_s_memrealtime();
	if(threadIdx.x == 0) atomicCAS(&a.stats->path_hits, 0ull, clk_r0);
""")
rep("""		if(!skip_trip)
#include "traverse_trip.inc"
#undef ADYPT_TRIP_TAKE_HIT
	}
""", """		if(!skip_trip)
		{
		pf_trips += 1; pf_trip_lanes += (uint32_t)__popcll(live);
#include "traverse_trip.inc"
		}
#undef ADYPT_TRIP_TAKE_HIT
	}
	if(lane == 0)
	{
		const unsigned long long total = __builtin_readcyclecounter() - pf_total0;
		atomicAdd(&a.stats->wave_profile[0], total); atomicAdd(&a.stats->wave_profile[1], pf_exch); atomicAdd(&a.stats->wave_profile[2], pf_shade);
		atomicAdd(&a.stats->wave_profile[3], pf_idle); atomicAdd(&a.stats->wave_profile[4], ((unsigned long long)pf_busy << 40) | (unsigned long long)pf_backlog);
		atomicAdd(&a.stats->wave_profile[5], ((unsigned long long)pf_rounds << 32) | pf_take);
		atomicAdd(&a.stats->wave_profile[6], ((unsigned long long)pf_exchanges << 32) | (pf_lock >> 8));
		atomicAdd(&a.stats->wave_profile[7], ((unsigned long long)pf_trips << 32) | (pf_trip_lanes >> 6));
		atomicMax(&a.stats->path_tris, __builtin_amdgcn_s_memrealtime());
	}
""")
open(p, "w").write(s)
