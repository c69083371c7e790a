"""Measurement variant of k_path (NOT product code): a trace of the shading rounds of workgroups 0..7 — 100 MHz timestamp at the start and end of every
round, paths taken, paths alive in the workgroup, deposited hits left waiting, ready rays waiting, rays the shading wave itself holds — plus the
moment the first wave of the launch found the global queue dry.  tools/path_drain_trace.py reads it through adypt_debug_read_drain.
    tools/build_variant.sh drain --transform adypt_amd/csrc/measure/k_path_drain_trace.py"""
import sys
d = sys.argv[1]
p = d + "/path.hpp"
s = open(p).read()


def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old[:80])
    s = s.replace(old, new)


rep("""namespace adypt {

#ifndef ADYPT_PATH_SLOTS""", """namespace adypt {
constexpr int kDrainWgs = 8, kDrainEvents = 8192;
struct DrainEvent { unsigned long long t0, t1; uint32_t take, live, left_shade, left_trace, own_rays, pad; };
__device__ DrainEvent g_drain[kDrainWgs][kDrainEvents];
__device__ uint32_t g_drain_n[kDrainWgs];
__device__ unsigned long long g_drain_dry;

#ifndef ADYPT_PATH_SLOTS""")
rep("""							if(gn == 0) break;
							if(dead && dead_rank >= served""", """							if(gn == 0) { if(lane == 0) atomicCAS(&g_drain_dry, 0ull, (unsigned long long)__builtin_amdgcn_s_memrealtime()); break; }
							if(dead && dead_rank >= served""")
rep("""					asm volatile("; ADYPT_MARK shade_begin");""", """					asm volatile("; ADYPT_MARK shade_begin");
					const unsigned long long dr_t0 = __builtin_amdgcn_s_memrealtime();
					const uint32_t dr_take = take, dr_live = lv, dr_ls = n_s, dr_lt = n_t, dr_own = 64u - n_idle;""")
rep("""					asm volatile("; ADYPT_MARK shade_end");""", """					if(lane == 0 && blockIdx.x < kDrainWgs)
					{
						const uint32_t e = atomicAdd(&g_drain_n[blockIdx.x], 1u);
						if(e < kDrainEvents) g_drain[blockIdx.x][e] = DrainEvent{dr_t0, (unsigned long long)__builtin_amdgcn_s_memrealtime(), dr_take, dr_live, dr_ls, dr_lt, dr_own, 0u};
					}
					asm volatile("; ADYPT_MARK shade_end");""")
open(p, "w").write(s)
t = d + "/tracer.hip"
s = open(t).read()
rep("""extern "C" {

int adypt_abi_version(void)""", """extern "C" int adypt_debug_read_drain(void *events, uint32_t *counts, unsigned long long *dry, int reset)
{
	(void)hipDeviceSynchronize();
	if(events && hipMemcpyFromSymbol(events, HIP_SYMBOL(adypt::g_drain), sizeof(adypt::DrainEvent) * adypt::kDrainWgs * adypt::kDrainEvents) != hipSuccess) return -3;
	if(counts && hipMemcpyFromSymbol(counts, HIP_SYMBOL(adypt::g_drain_n), sizeof(uint32_t) * adypt::kDrainWgs) != hipSuccess) return -3;
	if(dry && hipMemcpyFromSymbol(dry, HIP_SYMBOL(adypt::g_drain_dry), 8) != hipSuccess) return -3;
	if(reset)
	{
		static uint32_t zero[adypt::kDrainWgs] = {0}; unsigned long long z = 0;
		(void)hipMemcpyToSymbol(HIP_SYMBOL(adypt::g_drain_n), zero, sizeof(zero)); (void)hipMemcpyToSymbol(HIP_SYMBOL(adypt::g_drain_dry), &z, 8);
	}
	return 0;
}

extern "C" {

int adypt_abi_version(void)""")
open(t, "w").write(s)
