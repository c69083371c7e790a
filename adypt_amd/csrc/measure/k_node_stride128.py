"""Measurement variant (NOT product code): the 80-byte CWBVH8 nodes at a stride of 128 bytes — one cache line each, no node straddling two lines (VERDICT r5 task 3:
tools/microbench/gather_roof.hip gathers 128-byte-aligned records 1.49 x as fast as 80-byte ones at 2.7 GiB).  The kernels read 5 x 16 bytes per node as before; the upload
pads.  tools/build_variant.sh node128 --transform adypt_amd/csrc/measure/k_node_stride128.py"""
import sys
d = sys.argv[1]


def edit(name, pairs):
    p = d + "/" + name
    s = open(p).read()
    for old, new in pairs:
        assert s.count(old) == 1, (name, s.count(old), old[:70])
        s = s.replace(old, new)
    open(p, "w").write(s)


edit("traverse.hpp", [("constexpr int kNodeUint4 = 5;", "constexpr int kNodeUint4 = 8;")])
edit("tracer.hip", [("	TRY_CREATE(upload(c, &c->d_nodes, (const uint8_t *)d->nodes, (size_t)d->n_nodes * 80));",
                     "	{ std::vector<uint8_t> padded((size_t)d->n_nodes * 128, 0);\n"
                     "	  for(int64_t i = 0; i < d->n_nodes; ++i) memcpy(&padded[(size_t)i * 128], (const uint8_t *)d->nodes + (size_t)i * 80, 80);\n"
                     "	  TRY_CREATE(upload(c, &c->d_nodes, padded.data(), padded.size())); }")])
