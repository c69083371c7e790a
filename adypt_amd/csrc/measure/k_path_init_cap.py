"""Measurement variant of k_path (NOT product code): a workgroup holds at most ADYPT_PATH_INIT_CAP paths (default 64) instead of kPathSlots — the state of
a workgroup at the END of a product launch (a few dozen paths, each with its remaining bounces in sequence), reproduced from the first instruction of a
launch so that it can be timed alone: 8 such workgroups with nothing else on the chip, or 6 per CU everywhere (tools/path_floor.py).
    tools/build_variant.sh cap64 --transform adypt_amd/csrc/measure/k_path_init_cap.py [--transform adypt_amd/csrc/measure/k_path_drain_trace.py] -DADYPT_PATH_INIT_CAP=64"""
import sys
d = sys.argv[1]
p = d + "/path.hpp"
s = open(p).read()


def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old[:80])
    s = s.replace(old, new)


rep("""			while(have < (uint32_t)kPathSlots)
			{
				uint32_t gb = 0, left = 0;
				const uint32_t gn = fetch_rays(seg_len_lanes, seg_done, a.cursor, a.seg_cap, home, (uint32_t)kPathSlots - have, &gb, &left);""",
    """#ifndef ADYPT_PATH_INIT_CAP
#define ADYPT_PATH_INIT_CAP 64
#endif
			while(have < (uint32_t)ADYPT_PATH_INIT_CAP)
			{
				uint32_t gb = 0, left = 0;
				const uint32_t gn = fetch_rays(seg_len_lanes, seg_done, a.cursor, a.seg_cap, home, (uint32_t)ADYPT_PATH_INIT_CAP - have, &gb, &left);""")
open(p, "w").write(s)
