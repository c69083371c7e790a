"""Measurement variant of k_path (NOT product code): when do the workgroups of a launch finish, relative to the moment the global queue ran
dry?  Histogram in DeviceStats::wave_profile: bucket i (16 buckets of 100 us, two per 64-bit word) counts the workgroups that ended i x 100 us
after the first wave found the queue dry; workgroups that end before that count in bucket 0.  tools/path_tail_hist.py prints it.
    tools/build_variant.sh tailhist --transform adypt_amd/csrc/measure/k_path_tail_hist.py"""
import sys
p = sys.argv[1] + "/path.hpp"
s = open(p).read()


def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old[:80])
    s = s.replace(old, new)


rep("""							if(gn == 0) break;
							if(dead && dead_rank >= served""", """							if(gn == 0) { if(lane == 0) atomicCAS(&a.stats->path_nodes, 0ull, (unsigned long long)__builtin_amdgcn_s_memrealtime()); break; }
							if(dead && dead_rank >= served""")
rep("""	if(threadIdx.x == 0)
	{
		atomicAdd(&E.a.stats->rays, (unsigned long long)ctl->rays);
""", """	if(threadIdx.x == 0)
	{
		{
			const unsigned long long dry = __hip_atomic_load(&E.a.stats->path_nodes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), now = __builtin_amdgcn_s_memrealtime();
			const unsigned long long d = dry && now > dry ? (now - dry) / 10000ull : 0ull; // 100 MHz ticks -> 100 us buckets
			const uint32_t b = (uint32_t)(d > 15ull ? 15ull : d);
			atomicAdd(&E.a.stats->wave_profile[b >> 1], 1ull << ((b & 1u) * 32u));
			atomicAdd(&E.a.stats->path_tris, dry && now > dry ? now - dry : 0ull); // sum of the end times (ticks): the mean
		}
		atomicAdd(&E.a.stats->rays, (unsigned long long)ctl->rays);
""")
open(p, "w").write(s)
