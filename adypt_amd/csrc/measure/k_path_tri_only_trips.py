"""Measurement variant of k_path<true> (NOT product code): how many lane-trips end with triangles still pending (the lane could not slab-test its
next node this trip), and — -DADYPT_COUNT_ABSORBABLE — how many of those had a lane of the OTHER pair of their quad with nothing to test this
trip (neither a triangle of its own nor its pair-neighbour's second): the trips a four-lane instead of a two-lane hand-over could save.
Counted in wave_profile slot 6 ("refills", unused by k_path); read with adypt_get_wave_profile after an instrumented batch.
    tools/build_variant.sh trionly --transform adypt_amd/csrc/measure/k_path_tri_only_trips.py [-DADYPT_COUNT_ABSORBABLE]"""
import sys
p = sys.argv[1] + "/traverse_trip.inc"
s = open(p).read()
old = "\t\t\tif(tg_y != 0)\n\t\t\t{\n\t\t\t\t// more triangles of this node: next trip (the pending node is fetched in the trip that consumes the last of them)\n\t\t\t}\n"
assert s.count(old) == 1
new = """\t\t\t{
\t\t\t\t// a lane with nothing to test this trip; is there one in the other pair of my quad?
\t\t\t\tconst uint32_t free_me = (!do_test) ? 1u : 0u;
\t\t\t\tconst uint32_t free_other_pair = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)free_me, 0x4E, 0xF, 0xF, true)   // quad_perm [2,3,0,1]
\t\t\t\t                               | (uint32_t)__builtin_amdgcn_update_dpp(0, (int)free_me, 0x1B, 0xF, 0xF, true);  // quad_perm [3,2,1,0]
#ifdef ADYPT_COUNT_ABSORBABLE
\t\t\t\tconst bool counted = active && tg_y != 0 && free_other_pair != 0u;
#else
\t\t\t\tconst bool counted = active && tg_y != 0; (void)free_other_pair;
#endif
\t\t\t\tif(STATS) { const unsigned long long m = __ballot(counted); if(lane == 0) wp[6] += (unsigned long long)__popcll(m); }
\t\t\t}
""" + old
s = s.replace(old, new)
open(p, "w").write(s)
