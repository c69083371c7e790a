"""Measurement variant of k_path (NOT product code): "drain mode" for the end of a launch (profiles/r4_ablations_k_path.txt items 15, 16, 20).
Once the global queue is dry (PathCtl::busy bit 1) and the workgroup holds at most ADYPT_DRAIN_LIVE paths, waves 2-3 take no more rays; when
their own have ended they shade what waits in the to-shade ring WITHOUT parking (nothing to park), beside each other and beside a parking wave,
at a threshold of max(1, live >> ADYPT_DRAIN_SHIFT) paths per round; waves 0-1 trace.  -DADYPT_DRAIN_BY_SLOT: the quiet waves are chosen by their
slot on the SIMD instead (see below).
    tools/build_variant.sh drain --transform adypt_amd/csrc/measure/k_path_drain_mode.py -DADYPT_DRAIN_LIVE=128 -DADYPT_DRAIN_SHIFT=3"""
import sys
p = sys.argv[1] + "/path.hpp"
s = open(p).read()


def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old[:80])
    s = s.replace(old, new)


rep("#ifndef ADYPT_PATH_WAVES", """#ifndef ADYPT_DRAIN_LIVE
#define ADYPT_DRAIN_LIVE 128
#endif
#ifndef ADYPT_DRAIN_SHIFT
#define ADYPT_DRAIN_SHIFT 3
#endif
#ifndef ADYPT_PATH_WAVES""")
rep("struct PathArgs {", "constexpr uint32_t kBusyShading = 1u, kBusyDry = 2u; // PathCtl::busy\nstruct PathArgs {")
# which waves go quiet: waves 2-3 of every workgroup (default), or — -DADYPT_DRAIN_BY_SLOT — every wave but the workgroup's first whose slot on its
# SIMD (HW_REG_HW_ID.WAVE_ID) is odd: if wave w of a workgroup always sits on SIMD w, quieting waves 2-3 empties two SIMDs and leaves the other two as
# crowded as before
rep("\tconst float tmin = a.tmin;\n", """\tconst float tmin = a.tmin;
#ifdef ADYPT_DRAIN_BY_SLOT
\tconst bool quiet_wave = uni(threadIdx.x) >= 64u && (__builtin_amdgcn_s_getreg((4 /* HW_REG_HW_ID */) | (0 << 6) | ((4 - 1) << 11)) & 1u) != 0u;
#else
\tconst bool quiet_wave = uni(threadIdx.x) >= 128u;
#endif
""")
rep("ctl->n_trace = have > (uint32_t)kTraceThreads ? have - (uint32_t)kTraceThreads : 0u; }",
    "ctl->n_trace = have > (uint32_t)kTraceThreads ? have - (uint32_t)kTraceThreads : 0u; if(have < (uint32_t)kPathSlots) ctl->busy = kBusyDry; }")
rep("""			if(n_flush != 0u || pk_trace != 0u || (pk_shade + n_flush >= pk_thr && !pk_busy))""",
    """			const bool drain = (pk_busy & kBusyDry) != 0u && pk_live <= (uint32_t)ADYPT_DRAIN_LIVE && quiet_wave;
			const uint32_t pk_free_thr = max(1u, pk_live >> ADYPT_DRAIN_SHIFT);
			if(n_flush != 0u || (pk_trace != 0u && !drain) || (pk_shade + n_flush >= pk_thr && !(pk_busy & kBusyShading)) || (drain && n_idle == 64u && pk_shade >= pk_free_thr))""")
rep("""				const bool do_shade = n_s >= thr && n_s != 0u && uni(ctl->busy) == 0u; // one shading wave per workgroup at a time: one parking area""",
    """				const bool free_hands = drain && n_idle == 64u; // (no rays of its own: nothing to park)
				const bool do_shade = n_s != 0u && ((free_hands && n_s >= max(1u, lv >> ADYPT_DRAIN_SHIFT)) || (n_s >= thr && (uni(ctl->busy) & kBusyShading) == 0u));""")
rep("if(lane == 0) { ctl->h_shade = h_s; ctl->busy = 1u; }", "if(lane == 0) { ctl->h_shade = h_s; if(!free_hands) ctl->busy = ctl->busy | kBusyShading; }")
rep("""					const uint32_t got = min(n_idle, n_t);
					if(!active && idle_rank < got) { ray = to_trace[ring(h_t + idle_rank)]; setup = true; }
					if(lane == 0 && got) { ctl->n_trace = n_t - got; ctl->h_trace = ring(h_t + got); }""",
    """					const uint32_t got = drain ? 0u : min(n_idle, n_t);
					if(!active && idle_rank < got) { ray = to_trace[ring(h_t + idle_rank)]; setup = true; }
					if(lane == 0 && got) { ctl->n_trace = n_t - got; ctl->h_trace = ring(h_t + got); }""")
rep("""					*(uint4 *)(pk_lane + 0 * 256) = make_uint4(__float_as_uint(hit_t), __float_as_uint(hit_u), __float_as_uint(hit_v), (uint32_t)hit_idx);
					*(uint4 *)(pk_lane + 1 * 256) = make_uint4(ng_x, ng_y, tg_x, tg_y);
					*(uint2 *)(park + 2 * 256 + lane_here * 2) = make_uint2(node, ray | ((uint32_t)sp << 16));
""", """					if(!free_hands)
					{
						*(uint4 *)(pk_lane + 0 * 256) = make_uint4(__float_as_uint(hit_t), __float_as_uint(hit_u), __float_as_uint(hit_v), (uint32_t)hit_idx);
						*(uint4 *)(pk_lane + 1 * 256) = make_uint4(ng_x, ng_y, tg_x, tg_y);
						*(uint2 *)(park + 2 * 256 + lane_here * 2) = make_uint2(node, ray | ((uint32_t)sp << 16));
					}
""")
rep("""					{
						const uint4 p0 = *(const uint4 *)(pk_lane + 0 * 256), p1 = *(const uint4 *)(pk_lane + 1 * 256);""",
    """					if(!free_hands)
					{
						const uint4 p0 = *(const uint4 *)(pk_lane + 0 * 256), p1 = *(const uint4 *)(pk_lane + 1 * 256);""")
rep("""						node = p2.x; ray = p2.y & 0xffffu; sp = (int)(p2.y >> 16);
						aim();
					}
""", """						node = p2.x; ray = p2.y & 0xffffu; sp = (int)(p2.y >> 16);
						aim();
					}
					else // (every lane is idle: its ray state is dead, and saying so keeps it out of registers during the round)
					{
						hit_t = 1e9f; hit_u = 0.0f; hit_v = 0.0f; hit_idx = -1; ng_x = 0; ng_y = 0; tg_x = 0; tg_y = 0; node = 0; ray = lane_here; sp = 0;
						od_x = v2(0, 0); od_y = v2(0, 0); od_z = v2(0, 1); idir = f3(0, 0, 1); nx = false; ny = false; nz = false; octinv = 7u;
					}
""")
rep("""					const uint32_t got = min(n_idle, n_t); // and the wave's own idle lanes take the oldest ready rays
					if(!active && idle_rank < got) { ray = to_trace[ring(h_t + idle_rank)]; setup = true; }
					if(lane == 0) { ctl->n_trace = n_t - got; ctl->h_trace = ring(h_t + got); if(n_lost) ctl->live = ctl->live - n_lost; ctl->busy = 0u; }""",
    """					const uint32_t got = drain ? 0u : min(n_idle, n_t); // and the wave's own idle lanes take the oldest ready rays
					if(!active && idle_rank < got) { ray = to_trace[ring(h_t + idle_rank)]; setup = true; }
					if(lane == 0)
					{
						ctl->n_trace = n_t - got; ctl->h_trace = ring(h_t + got);
						if(n_lost) ctl->live = ctl->live - n_lost; // (a path without a replacement: the global queue is dry)
						ctl->busy = (ctl->busy & (free_hands ? (kBusyShading | kBusyDry) : kBusyDry)) | (n_lost ? kBusyDry : 0u);
					}""")
open(p, "w").write(s)
