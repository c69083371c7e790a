"""Measurement variant of k_path (NOT product code): node prefetch without destination registers — global_load_lds_dword of the first and last
dword of an 80-byte node into a 256-byte LDS area nobody reads (the kernel has no VGPR to spare for an early fetch) — at two points:
  -DADYPT_PF_PUSH  when a node group is pushed: the child a later pop of that group visits first ("the popped stack top's node
                   fetched early");
  -DADYPT_PF_NEXT  after a slab test: the closest hit inner child, i.e. the node the next trip chooses and fetches anyway (a head start of the
                   loop's back edge, the exchange check and section A).
-DADYPT_PF_ONE: only the first dword (one line; a node straddles two 128-byte lines half the time).
    tools/build_variant.sh pfpush --transform adypt_amd/csrc/measure/k_path_prefetch.py -DADYPT_PF_PUSH
Numbers: profiles/r4_ablations_k_path.txt item 18."""
import sys
d = sys.argv[1]


def edit(name, pairs):
    s = open(d + "/" + name).read()
    for old, new in pairs:
        assert s.count(old) == 1, (name, s.count(old), old[:70])
        s = s.replace(old, new)
    open(d + "/" + name, "w").write(s)


edit("path.hpp", [
    ("\t       sizeof(PathCtl);\n}", "\t       sizeof(PathCtl) + 256; // + the prefetch sink\n}"),
    ("\tconst float tmin = a.tmin;\n",
     "\tconst float tmin = a.tmin;\n"
     "\tconst uint32_t pf_sink = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)(ctl + 1);\n"
     "\tauto prefetch_node = [&](uint32_t n) {\n"
     "\t\tconst char *p = (const char *)(a.nodes + (size_t)n * kNodeUint4);\n"
     "\t\tuint32_t keep;\n"
     "#ifdef ADYPT_PF_ONE\n"
     "\t\tasm volatile(\"s_mov_b32 %0, m0\\n\\ts_mov_b32 m0, %2\\n\\tglobal_load_lds_dword %1, off\\n\\ts_mov_b32 m0, %0\" : \"=&s\"(keep) : \"v\"(p), \"s\"(pf_sink));\n"
     "#else\n"
     "\t\tasm volatile(\"s_mov_b32 %0, m0\\n\\ts_mov_b32 m0, %2\\n\\tglobal_load_lds_dword %1, off\\n\\tglobal_load_lds_dword %1, off offset:76\\n\\ts_mov_b32 m0, %0\" : \"=&s\"(keep) : \"v\"(p), \"s\"(pf_sink));\n"
     "#endif\n"
     "\t};\n"
     "#define ADYPT_PREFETCH_NODE(n) prefetch_node(n)\n"),
])
edit("traverse_trip.inc", [
    ("\t\t\t\t++sp;\n\t\t\t\tif(STATS) depth_after_push = (uint32_t)sp;\n",
     "\t\t\t\t++sp;\n\t\t\t\tif(STATS) depth_after_push = (uint32_t)sp;\n"
     "#if defined(ADYPT_PF_PUSH) && defined(ADYPT_PREFETCH_NODE)\n"
     "\t\t\t\t{ const uint32_t pslot = ((31u - (uint32_t)__builtin_clz(ng_y)) - 24u) ^ octinv; ADYPT_PREFETCH_NODE(ng_x + (uint32_t)__builtin_popcount(ng_y & ~(0xffffffffu << pslot))); }\n"
     "#endif\n"),
    ("\t\t\t\ttg_y = hitmask & 0x00ffffffu;\n\t\t\t}\n",
     "\t\t\t\ttg_y = hitmask & 0x00ffffffu;\n"
     "#if defined(ADYPT_PF_NEXT) && defined(ADYPT_PREFETCH_NODE)\n"
     "\t\t\t\tif(ng_y > 0x00ffffffu) { const uint32_t pslot = ((31u - (uint32_t)__builtin_clz(ng_y)) - 24u) ^ octinv; ADYPT_PREFETCH_NODE(ng_x + (uint32_t)__builtin_popcount(ng_y & ~(0xffffffffu << pslot))); }\n"
     "#endif\n"
     "\t\t\t}\n"),
])
