"""Measurement variants of k_trace / k_shade (NOT product code): puts the hook points of csrc/measure/k_trace_ablations.hpp into a scratch copy
of the device sources — the product sources carry none.  Which hook does anything is chosen by -DADYPT_ABLATE_<what> (see the header):
    tools/build_variant.sh loads2 --transform adypt_amd/csrc/measure/k_trace_hooks.py -DADYPT_ABLATE_EXTRA_LOADS
    tools/build_variant.sh fp16   --transform adypt_amd/csrc/measure/k_trace_hooks.py -DADYPT_ABLATE_FP16_NODES
    tools/build_variant.sh tl     --transform adypt_amd/csrc/measure/k_trace_hooks.py -DADYPT_ABLATE_WAVE_TIMELINE      (tools/wave_timeline.py)
    tools/build_variant.sh tril2  --transform adypt_amd/csrc/measure/k_trace_hooks.py -DADYPT_ABLATE_SHADE_TRI_L2
and compared with tools/ab.py / tools/path_sweep.py (ADYPT_LIB).  Numbers: profiles/r2_ablations_k_trace.txt, profiles/r3_ablations_k_trace.txt."""
import sys
d = sys.argv[1]


def edit(name, pairs):
    s = open(d + "/" + name).read()
    for old, new in pairs:
        assert s.count(old) == 1, (name, s.count(old), old[:70])
        s = s.replace(old, new)
    open(d + "/" + name, "w").write(s)


DEFAULTS = """#define ADYPT_MEASUREMENT_BUILD
#include "../measure/k_trace_ablations.hpp"
#ifndef ADYPT_MEASURE_FP16_NODES
#define ADYPT_MEASURE_MORE_NODE_REGS()
#define ADYPT_MEASURE_LOAD_MORE_NODE(np)
#endif
#ifndef ADYPT_MEASURE_WAVE_TIMELINE
#define ADYPT_MEASURE_WAVE_BEGIN()
#define ADYPT_MEASURE_WAVE_FIRST_RAYS()
#define ADYPT_MEASURE_WAVE_QUEUE_DRY()
#define ADYPT_MEASURE_WAVE_END(stats)
#endif
#if defined(ADYPT_ABLATE_SHADE_TRI_L2)  // k_shade's triangle gather folded onto the first 16384 records (2 MB: resident in every XCD's L2): wrong images,
#define ADYPT_MEASURE_SHADE_GATHER_INDEX(i) ((i) & 16383)  // but the kernel's time then says what the gathers' misses cost
#else
#define ADYPT_MEASURE_SHADE_GATHER_INDEX(i) (i)
#endif
"""
edit("canon_math.hpp", [("namespace adypt {\n\nstruct F3", DEFAULTS + "\nnamespace adypt {\n\nstruct F3")])
edit("shade.hpp", [("const TriCore tc = load_tri_core(sc, tri_idx);\n\tconst float *tri = tc.v;\n\tconst int matid = __float_as_int(tri[18]);\n\t// the hit's geometry",
                    "const TriCore tc = load_tri_core(sc, ADYPT_MEASURE_SHADE_GATHER_INDEX(tri_idx));\n\tconst float *tri = tc.v;\n\tconst int matid = __float_as_int(tri[18]);\n\t// the hit's geometry")])
edit("traverse.hpp", [
    ("constexpr int kNodeUint4 = 5;", "#ifndef ADYPT_MEASURE_FP16_NODES\nconstexpr int kNodeUint4 = 5;\n#endif //"),
    ("\tconst unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_r0 = __builtin_amdgcn_s_memrealtime();\n\tif(blockIdx.x == 0 && threadIdx.x == 0)\n\t{\n\t\tunsigned long long total = CAMERA",
     "\tADYPT_MEASURE_WAVE_BEGIN();\n\tconst unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_r0 = __builtin_amdgcn_s_memrealtime();\n\tif(blockIdx.x == 0 && threadIdx.x == 0)\n\t{\n\t\tunsigned long long total = CAMERA"),
    ("\t\t\t\tif(cn == 0 && dry) exhausted = true;\n", "\t\t\t\tif(cn == 0 && dry) { exhausted = true; ADYPT_MEASURE_WAVE_QUEUE_DRY(); }\n\t\t\t\telse if(cn) { ADYPT_MEASURE_WAVE_FIRST_RAYS(); }\n"),
    ("\tif(blockIdx.x == 0 && threadIdx.x == 0)\n\t{\n\t\tatomicAdd(&a.stats->clock_cycles,", "\tADYPT_MEASURE_WAVE_END(a.stats);\n\tif(blockIdx.x == 0 && threadIdx.x == 0)\n\t{\n\t\tatomicAdd(&a.stats->clock_cycles,"),
])
edit("traverse_trip.inc", [
    ("\t\t\t\twp0 = w0[0]; wp1 = w0[1]; wp2 = w0[2];\n", "\t\t\t\twp0 = w0[0]; wp1 = w0[1]; wp2 = w0[2];\n\t\t\t\tADYPT_MEASURE_AFTER_TRI_LOADS(w0);\n"),
    ("ADYPT_DEF4(n3); ADYPT_DEF4(n4);\n", "ADYPT_DEF4(n3); ADYPT_DEF4(n4);\n\t\t\tADYPT_MEASURE_MORE_NODE_REGS();\n"),
    ("n3 = np[3]; n4 = np[4];\n", "n3 = np[3]; n4 = np[4];\n\t\t\t\tADYPT_MEASURE_LOAD_MORE_NODE(np);\n\t\t\t\tADYPT_MEASURE_AFTER_NODE_LOADS(np, lane);\n"),
    ("\t\t\t\tuint32_t hitmask = 0;\n#pragma unroll\n",
     "\t\t\t\tuint32_t hitmask = 0;\n#ifdef ADYPT_MEASURE_FP16_NODES\n\t\t\t\thitmask = slab_test_fp16_nodes(n1, n2, n3, n4, n5, n6, n7, nx, ny, nz, octinv4, aix, aiy, aiz, aox, aoy, aoz, tmin, hit_t);\n#else\n#pragma unroll\n"),
    ("\t\t\t\tng_y = (hitmask & 0xff000000u) | (head_w >> 24);\n", "#endif\n\t\t\t\tADYPT_MEASURE_AFTER_SLAB_TEST(aox, aoy, aoz, aix, aiy);\n\t\t\t\tng_y = (hitmask & 0xff000000u) | (head_w >> 24);\n"),
])
edit("tracer.hip", [
    ('#include "traverse.hpp"\n', '#define ADYPT_TRACER_TU // (measure/k_trace_ablations.hpp defines its read-back entry points in this translation unit only)\n#include "traverse.hpp"\n'),
    ("\tTRY_CREATE(upload(c, &c->d_nodes, (const uint8_t *)d->nodes, (size_t)d->n_nodes * 80));\n",
     "#ifdef ADYPT_MEASURE_FP16_NODES // 128-byte nodes with binary16 bounds\n\t{\n\t\tconst std::vector<uint8_t> wide = adypt::nodes_as_fp16((const uint8_t *)d->nodes, (size_t)d->n_nodes);\n"
     "\t\tTRY_CREATE(upload(c, &c->d_nodes, wide.data(), wide.size()));\n\t}\n#else\n\tTRY_CREATE(upload(c, &c->d_nodes, (const uint8_t *)d->nodes, (size_t)d->n_nodes * 80));\n#endif\n"),
])
