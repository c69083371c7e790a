"""Measurement variant of k_path (NOT product code): where a wave's TIME goes inside one iteration of the persistent loop, by dependent section.
Seven shader-clock stamps per trip (s_memtime + s_waitcnt lgkmcnt(0), between scheduling barriers); the delta since the previous stamp is added to the
section's 32-bit accumulator, which lives in a fixed SGPR the compiler is kept away from (k_path is held to 88 SGPRs; s88 .. s101 belong to the stamps).
At the end of the launch lane 0 of every wave adds its sums to DeviceStats::wave_profile (read by tools/path_timeline.py):
    [0] before the trip : everything between the end of a trip and the start of the next — ray setup, exchange, shading rounds, sleeping
    [1] A               : pop / choose / push (one LDS round trip when a group is popped)
    [2] B match         : ballots, ranks, the hand-out table written and read back (one LDS round trip)
    [3] B pull + issue  : the ray pulled through the crossbar (one LDS round trip), triangle index, all eight loads issued
    [4] triangle fetch  : what is left of the triangle loads' latency when the Woop arithmetic wants them (s_waitcnt vmcnt(5))
    [5] C               : Woop test, verdicts back, winner's (u, v, index) back (two LDS round trips), hit stored
    [6] D + E           : the rest of the node loads' latency, the slab test, the finished-ray check
    [7] trips << 36 | cost of one stamp, summed (a back-to-back stamp at the end of every trip: every section above contains one such cost)
    tools/build_variant.sh timeline --transform adypt_amd/csrc/measure/k_path_timeline.py [--transform adypt_amd/csrc/measure/k_path_init_cap.py]"""
import sys
d = sys.argv[1]


def edit(name, pairs):
    p = d + "/" + name
    s = open(p).read()
    for old, new in pairs:
        assert s.count(old) == 1, (name, s.count(old), old[:70])
        s = s.replace(old, new)
    open(p, "w").write(s)


CLOB = '"s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "s100", "s101", "scc", "memory"'


def stamp(i, pre=""):  # s98 = previous stamp (low word), s100:101 = this stamp, s99 = delta; s88 + i = accumulator of section i
    return ('__builtin_amdgcn_sched_barrier(0); asm volatile("%ss_memtime s[100:101]\\n\\ts_waitcnt lgkmcnt(0)\\n\\ts_sub_u32 s99, s100, s98\\n\\ts_add_u32 s%d, s%d, s99\\n\\ts_mov_b32 s98, s100" ::: %s); __builtin_amdgcn_sched_barrier(0);'
            % (pre, 88 + i, 88 + i, CLOB))


edit("traverse_trip.inc", [
    ("			const bool can_pop = !pending && ng_y <= 0x00ffffffu && sp != 0;", "			" + stamp(0) + "\n			const bool can_pop = !pending && ng_y <= 0x00ffffffu && sp != 0;"),
    ("			auto pull = [](uint32_t lane4, uint32_t v)", "			" + stamp(1) + "\n			auto pull = [](uint32_t lane4, uint32_t v)"),
    ("			const uint32_t src4 = ent << 2;", "			" + stamp(2) + "\n			const uint32_t src4 = ent << 2;"),
    ("			__builtin_amdgcn_sched_barrier(0); // keep the compiler from hoisting arithmetic between the load issues\n", "			" + stamp(3) + "\n"),
    ("			float tt, tu, tv;\n", "			" + stamp(4, "s_waitcnt vmcnt(5)\\n\\t") + "\n			float tt, tu, tv;\n"),
    ("			if(tg_y != 0)\n			{\n				// more triangles of this node", "			" + stamp(5) + "\n			if(tg_y != 0)\n			{\n				// more triangles of this node"),
    ("				active = false;\n			}\n		}", "				active = false;\n			}\n			" + stamp(6) + "\n			" + stamp(8) + "\n			asm volatile(\"s_add_u32 s95, s95, 1\" ::: \"s95\", \"scc\");\n		}"),
])
zero = " ".join('asm volatile("s_mov_b32 s%d, 0" ::: "s%d");' % (r, r) for r in range(88, 98)) + ' asm volatile("s_memtime s[100:101]\\n\\ts_waitcnt lgkmcnt(0)\\n\\ts_mov_b32 s98, s100" ::: "s98", "s100", "s101");'
read = " ".join('asm volatile("s_mov_b32 %%0, s%d" : "=s"(tl[%d]));' % (88 + i, i) for i in range(9))
edit("path.hpp", [
    ("template <bool STATS, bool SUN>\n__global__ __launch_bounds__(kTraceThreads, STATS ? 4 : ADYPT_PATH_WAVES) void k_path(PathKernArgs K)\n{",
     "template <bool STATS, bool SUN>\n__global__ __launch_bounds__(kTraceThreads, STATS ? 4 : ADYPT_PATH_WAVES) __attribute__((amdgpu_num_sgpr(88))) void k_path(PathKernArgs K)\n{\n	" + zero),
    ("	// ---------------- totals: per wave -> per workgroup (LDS) -> one device atomic per workgroup ----------------",
     "	{ uint32_t tl[9]; " + read + "\n	if(lane == 0) { for(int i = 0; i < 7; ++i) atomicAdd(&a.stats->wave_profile[i], (unsigned long long)tl[i]);\n"
     "		atomicAdd(&a.stats->wave_profile[7], ((unsigned long long)tl[7] << 36) | (unsigned long long)tl[8]); } } // (k_path<false> leaves wave_profile alone)\n"
     "	// ---------------- totals: per wave -> per workgroup (LDS) -> one device atomic per workgroup ----------------"),
])
# (k_trace includes the trip too: its stamps write the same fixed registers, which that kernel never reads — it is not held to 88 SGPRs, so it must not be run
# from this variant: tools/path_timeline.py uses the one-launch pipeline only)
