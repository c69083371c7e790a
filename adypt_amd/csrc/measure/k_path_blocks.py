"""Measurement variant of k_path (NOT product code): every block of the persistent loop that the hardware may skip — the blocks behind a wave-level
branch — is bracketed by `; ADYPT_MARK <name>_begin / _end` comments in the assembly and counts, in an SGPR, how often a wave ENTERS it (an s_add in the
block itself: it runs exactly when the block's vector instructions are issued, whatever the lanes' masks).  tools/trip_budget.py reads the static
instruction counts between the marks (no GPU needed); tools/path_block_counts.py reads the entry counts on the GPU box; together they are the EXECUTED
vector instructions per trip (profiles/r6_trip_budget.json, r6_k_path_block_counts.json), which tools/valu_issue_model.py checks against SQ_INSTS_VALU.
    tools/build_variant.sh blocks --transform adypt_amd/csrc/measure/k_path_blocks.py
Three counting passes (twelve counters each: ADYPT_BLOCKS_SET = trip | shade | rare); every mark is inserted in every pass.  Blocks: setup | exchange | shade | trip;
inside the trip A_pop (A_pop_spill) A_choose A_push (A_push_spill) B_tri_load B_node_load C_woop D_slab E_flush; inside a shading round S_parked S_miss S_surface
(S_textured S_glossy S_diffuse S_mirror S_dielectric) S_dead S_alive S_replace S_early S_fetch_more S_defer (a round moves glossy / dielectric hits to the deferred ring); inside the exchange X_pick | X_take (its two exclusive branches: a round follows, or rays are taken); anywhere div_slow (the division sequence behind rcp_ieee's
range test), F_try (one try of fetch_rays at one queue segment) and X_lock_spin.  What lies inside a block but outside its sub-blocks runs whenever the block does."""
import sys
d = sys.argv[1]
import os
COUNT = os.environ.get("ADYPT_BLOCKS_COUNT", "0") != "0"   # 0: marks only (static counts: tools/trip_budget.py); 1: marks + entry counters (tools/path_block_counts.py)
SET = os.environ.get("ADYPT_BLOCKS_SET", "trip")           # which twelve blocks get the counters (every mark is always inserted): trip | shade | rare
SETS = {"trip": ["setup", "exchange", "shade", "trip", "A_pop", "A_choose", "A_push", "B_tri_load", "B_node_load", "C_woop", "D_slab", "E_flush"],
        "shade": ["shade", "S_parked", "S_miss", "S_surface", "S_textured", "S_glossy", "S_diffuse", "S_mirror", "S_dielectric", "S_dead", "S_alive", "S_replace"],
        # blocks expected to run (almost) never: stack entries beyond the LDS part, the division sequence behind rcp_ieee's range test (every site one
        # counter), the shading round's second and later reservations of replacement paths, its early reservation, the lock's spin
        "rare": ["trip", "shade", "A_pop_spill", "A_push_spill", "div_slow", "S_fetch_more", "S_early", "X_lock_spin", "exchange", "setup", "F_try", "S_defer"]}
LANES = os.environ.get("ADYPT_BLOCKS_LANES", "0") != "0"   # 1: the counters add up the ACTIVE LANES at each entry (s_bcnt1 of exec) instead of the entries: five 32-bit counters per pass
SETS_LANES = {"trip": ["trip", "A_choose", "C_woop", "D_slab", "E_flush"], "shade": ["S_surface", "S_textured", "S_glossy", "S_diffuse", "S_dielectric"],
              "rare": ["S_miss", "S_mirror", "S_dead", "S_alive", "S_replace"],
              # why lanes sit out a trip's slab test (probe blocks that exist in this pass only): no ray | triangles of the last node left after this trip's pair;
              # and the triangle list: lanes with two or more, three or more, four or more triangles at hand
              "wait": ["W_idle", "W_wait", "W_two", "W_three", "W_four"]}
NAMES = SETS_LANES[SET] if LANES else SETS[SET]
# Counters: a value that is modified inside a divergent block cannot live in an SGPR the compiler allocates (the merge after the block is per lane).
# So k_path is held to 96 SGPRs (amdgpu_num_sgpr) and the counters live in s96 .. s101, touched only by inline assembly: two 16-bit counters per
# register (a wave makes < 65536 trips per launch at the batch sizes measured).


def edit(name, pairs):
    p = d + "/" + name
    s = open(p).read()
    for old, new in pairs:
        assert s.count(old) == 1, (name, s.count(old), old[:70])
        s = s.replace(old, new)
    open(p, "w").write(s)


def enter(n):  # the counter of block n goes up once per wave that enters the block (SALU: not a vector instruction, blind to the exec mask)
    # (sched_barrier: without it the scheduler is free to move a block's arithmetic across the comment, and both marks end up at the block's top)
    if n not in NAMES or not COUNT:
        return '__builtin_amdgcn_sched_barrier(0); asm volatile("; ADYPT_MARK %s_begin"); __builtin_amdgcn_sched_barrier(0);' % n
    i = NAMES.index(n)
    if LANES:  # s96 .. s100 = lanes that entered block i, s101 = scratch
        return '__builtin_amdgcn_sched_barrier(0); asm volatile("s_bcnt1_i32_b64 s101, exec\\n\\ts_add_u32 s%d, s%d, s101 ; ADYPT_MARK %s_begin" ::: "s%d", "s101", "scc"); __builtin_amdgcn_sched_barrier(0);' % (96 + i, 96 + i, n, 96 + i)
    return '__builtin_amdgcn_sched_barrier(0); asm volatile("s_add_u32 s%d, s%d, %s ; ADYPT_MARK %s_begin" ::: "s%d"); __builtin_amdgcn_sched_barrier(0);' % (96 + i // 2, 96 + i // 2, "0x10000" if i & 1 else "1", n, 96 + i // 2)


def leave(n):
    return '__builtin_amdgcn_sched_barrier(0); asm volatile("; ADYPT_MARK %s_end"); __builtin_amdgcn_sched_barrier(0);' % n


TRIP_EDITS = [
    ("			const bool can_pop = !pending && ng_y <= 0x00ffffffu && sp != 0;", "			asm volatile(\"; ADYPT_MARK sec_A\");\n			const bool can_pop = !pending && ng_y <= 0x00ffffffu && sp != 0;"),
    ("			auto pull = [](uint32_t lane4, uint32_t v)", "			asm volatile(\"; ADYPT_MARK sec_B\");\n			auto pull = [](uint32_t lane4, uint32_t v)"),
    ("			float tt, tu, tv;\n", "			asm volatile(\"; ADYPT_MARK sec_C\");\n			float tt, tu, tv;\n"),
    ("			if(tg_y != 0)\n			{\n				// more triangles of this node", "			asm volatile(\"; ADYPT_MARK sec_D\");\n			if(tg_y != 0)\n			{\n				// more triangles of this node"),
    ("			if(active && tg_y == 0 && !pending && ((ng_y <= 0x00ffffffu && sp == 0) || ((ANY || kTripShadowRays) && lane_any && won)))", "			asm volatile(\"; ADYPT_MARK sec_E\");\n			if(active && tg_y == 0 && !pending && ((ng_y <= 0x00ffffffu && sp == 0) || ((ANY || kTripShadowRays) && lane_any && won)))"),
    ("			if(can_pop)\n			{\n				--sp;", "			if(can_pop)\n			{\n				" + enter("A_pop") + "\n				--sp;"),
    ("					const uint2 g = my_spill[(size_t)(sp - a.lds_depth) * total_lanes];\n					ng_x = g.x; ng_y = g.y;\n					asm volatile(\"\" : \"+v\"(ng_x), \"+v\"(ng_y));\n				}\n			}",
     "					" + enter("A_pop_spill") + "\n					const uint2 g = my_spill[(size_t)(sp - a.lds_depth) * total_lanes];\n					ng_x = g.x; ng_y = g.y;\n					asm volatile(\"\" : \"+v\"(ng_x), \"+v\"(ng_y));\n					" + leave("A_pop_spill") + "\n				}\n				" + leave("A_pop") + "\n			}"),
    ("			if(choose)\n			{\n				const uint32_t imask = ng_y;", "			if(choose)\n			{\n				" + enter("A_choose") + "\n				const uint32_t imask = ng_y;"),
    ("				pending = true;\n			}", "				pending = true;\n				" + leave("A_choose") + "\n			}"),
    ("			if(push_ok)\n			{\n				if(sp < a.lds_depth) my_stack[sp * 64] = make_uint2(ng_x, ng_y);\n				else my_spill[(size_t)(sp - a.lds_depth) * total_lanes] = make_uint2(ng_x, ng_y);\n				++sp;",
     "			if(push_ok)\n			{\n				" + enter("A_push") + "\n				if(sp < a.lds_depth) my_stack[sp * 64] = make_uint2(ng_x, ng_y);\n				else { " + enter("A_push_spill") + " my_spill[(size_t)(sp - a.lds_depth) * total_lanes] = make_uint2(ng_x, ng_y); " + leave("A_push_spill") + " }\n				++sp;"),
    ("				if(STATS) depth_after_push = (uint32_t)sp;\n			}", "				if(STATS) depth_after_push = (uint32_t)sp;\n				" + leave("A_push") + "\n			}"),
    ("			if(do_test)\n			{\n				const auto w0 = trip_woop", "			if(do_test)\n			{\n				" + enter("B_tri_load") + "\n				const auto w0 = trip_woop"),
    ("				wp0 = trip_ld(w0); wp1 = trip_ld(w0 + 1); wp2 = trip_ld(w0 + 2);\n			}", "				wp0 = trip_ld(w0); wp1 = trip_ld(w0 + 1); wp2 = trip_ld(w0 + 2);\n				" + leave("B_tri_load") + "\n			}"),
    ("			if(pending && tg_y == 0)\n			{\n				const auto np = trip_nodes", "			if(pending && tg_y == 0)\n			{\n				" + enter("B_node_load") + "\n				const auto np = trip_nodes"),
    ("				n0 = trip_ld(np); n1 = trip_ld(np + 1); n2 = trip_ld(np + 2); n3 = trip_ld(np + 3); n4 = trip_ld(np + 4);\n			}", "				n0 = trip_ld(np); n1 = trip_ld(np + 1); n2 = trip_ld(np + 2); n3 = trip_ld(np + 3); n4 = trip_ld(np + 4);\n				" + leave("B_node_load") + "\n			}"),
    ("			if(do_test)\n			{\n				if(STATS) wave_event(2);", "			if(do_test)\n			{\n				" + enter("C_woop") + "\n				if(STATS) wave_event(2);"),
    ("				geom_ok = tt > t_tmin && tu >= 0.0f && tu <= 1.0f && tv >= 0.0f && tu + tv <= 1.0f;\n			}", "				geom_ok = tt > t_tmin && tu >= 0.0f && tu <= 1.0f && tv >= 0.0f && tu + tv <= 1.0f;\n				" + leave("C_woop") + "\n			}"),
    ("				pending = false;\n				if(ANY || kTripShadowRays) overflow |= push_overflow;", "				" + enter("D_slab") + "\n				pending = false;\n				if(ANY || kTripShadowRays) overflow |= push_overflow;"),
    ("				tg_y = hitmask & 0x00ffffffu;\n			}", "				tg_y = hitmask & 0x00ffffffu;\n				" + leave("D_slab") + "\n			}"),
    ("				if(ANY || kTripShadowRays) { if(lane_any) { ng_y = 0; sp = 0; } } // an any-hit ray ends with work left: make the lane inert", "				" + enter("E_flush") + "\n				if(ANY || kTripShadowRays) { if(lane_any) { ng_y = 0; sp = 0; } } // an any-hit ray ends with work left: make the lane inert"),
    ("				active = false;\n			}\n		}", "				active = false;\n				" + leave("E_flush") + "\n			}\n		}"),
]
if LANES and SET == "wait":
    TRIP_EDITS += [
        ("			if(tg_y != 0)\n			{\n				" if False else "				// more triangles of this node: next trip (the pending node is fetched in the trip that consumes the last of them)\n", "				" + enter("W_wait") + "\n"),
        ("			const bool do_test = (uint32_t)lane < n_tests;", "			if(!active) { " + enter("W_idle") + " }\n			if(tg1 != 0) { " + enter("W_two") + " }\n			if(tg2 != 0) { " + enter("W_three") + " }\n			if(tg3 != 0) { " + enter("W_four") + " }\n			const bool do_test = (uint32_t)lane < n_tests;"),
    ]
edit("traverse_trip.inc", TRIP_EDITS)
# (fetch_rays' loop over the 8 queue segments is unrolled: 8 equal instances of the block, one counter = tries in all)
edit("traverse.hpp", [("		if((seg_done >> s) & 1u) continue;\n		const uint32_t seg_len = (uint32_t)__builtin_amdgcn_readlane((int)seg_len_lanes, s);",
                       "		if((seg_done >> s) & 1u) continue;\n		" + enter("F_try") + "\n		const uint32_t seg_len = (uint32_t)__builtin_amdgcn_readlane((int)seg_len_lanes, s);"),
                      ("		seg_done |= 1u << s;\n	}\n	*left = 0;", "		seg_done |= 1u << s;\n		" + leave("F_try") + "\n	}\n	*left = 0;")])
edit("canon_math.hpp", [("	return 1.0f / x;\n}", "	" + enter("div_slow") + "\n	const float q_slow = 1.0f / x;\n	" + leave("div_slow") + "\n	return q_slow;\n}")])
# (k_trace includes the trip too: it gets a dummy counter array)
zero = " ".join('asm volatile("s_mov_b32 s%d, 0" ::: "s%d");' % (r, r) for r in range(96, 102))
read = " ".join('asm volatile("s_mov_b32 %%0, s%d" : "=s"(bc[%d]));' % (96 + i, i) for i in range(6))
SHADE_PATH = [
    ("						if(parked) r4 = f.done[pi];", "						if(parked) { " + enter("S_parked") + " r4 = f.done[pi]; " + leave("S_parked") + " }"),
    ("							ret = fma3(color, f3(f.sun[0], f.sun[1], f.sun[2]), ret); // pathtracer.glsl:130-135 (a query that came back empty: the sun is visible)\n							alive = false;",
     "							" + enter("S_miss") + "\n							ret = fma3(color, f3(f.sun[0], f.sun[1], f.sun[2]), ret); // pathtracer.glsl:130-135 (a query that came back empty: the sun is visible)\n							alive = false;\n							" + leave("S_miss")),
    ("							const SurfaceInfo si = fetch_info(f, sc, tc, tri_idx, tu, tv);", "							" + enter("S_surface") + "\n							const SurfaceInfo si = fetch_info(f, sc, tc, tri_idx, tu, tv);"),
    ("						if(rm != 0ull && (uint32_t)__popcll(rm) <= a.defer_max)\n						{", "						if(rm != 0ull && (uint32_t)__popcll(rm) <= a.defer_max)\n						{\n							" + enter("S_defer")),
    ("							have = have && !defer;\n						}", "							have = have && !defer;\n							" + leave("S_defer") + "\n						}"),
    ("								alive = respond(f, si, rng, b, dir, color, ret);\n							}\n						}", "								alive = respond(f, si, rng, b, dir, color, ret);\n							}\n							" + leave("S_surface") + "\n						}"),
    ("						if(!alive) // main()'s clamp (pathtracer.glsl:224); the running mean is k_resolve's, in frame order (a k_path pass is always batched)\n							f.done[pi] = make_float4(gl_min(ret.x, f.clamp), gl_min(ret.y, f.clamp), gl_min(ret.z, f.clamp), 1.0f);",
     "						if(!alive) { " + enter("S_dead") + "\n							f.done[pi] = make_float4(gl_min(ret.x, f.clamp), gl_min(ret.y, f.clamp), gl_min(ret.z, f.clamp), 1.0f); " + leave("S_dead") + " }"),
    ("					if(alive)\n					{\n						if(__float_as_uint(ret.x) != __float_as_uint(ret_in.x)", "					if(alive)\n					{\n						" + enter("S_alive") + "\n						if(__float_as_uint(ret.x) != __float_as_uint(ret_in.x)"),
    ("						tab[T_OX * kPathSlots + sslot] = __float_as_uint(origin.x); tab[T_OY * kPathSlots + sslot] = __float_as_uint(origin.y); tab[T_OZ * kPathSlots + sslot] = __float_as_uint(origin.z);\n					}\n					// ---------------- paths that ended",
     "						tab[T_OX * kPathSlots + sslot] = __float_as_uint(origin.x); tab[T_OY * kPathSlots + sslot] = __float_as_uint(origin.y); tab[T_OZ * kPathSlots + sslot] = __float_as_uint(origin.z);\n						" + leave("S_alive") + "\n					}\n					// ---------------- paths that ended"),
    ("					if(repl) load_path(a, idx, sslot);", "					if(repl) { " + enter("S_replace") + " load_path(a, idx, sslot); " + leave("S_replace") + " }"),
]
SHADE_HPP = [
    ("		if(f.n_tex != 0 && dtex != -1 && dtex >= 0 && dtex < f.n_tex) s.diffuse = textured_diffuse(sc, tri_idx, tex_desc, tu, tv, w);\n		else s.diffuse = f3(md.y, md.z, md.w);\n		s.specular = f3(ms.y, ms.z, ms.w);",
     "		if(f.n_tex != 0 && dtex != -1 && dtex >= 0 && dtex < f.n_tex) { " + enter("S_textured") + " s.diffuse = textured_diffuse(sc, tri_idx, tex_desc, tu, tv, w); " + leave("S_textured") + " }\n		else s.diffuse = f3(md.y, md.z, md.w);\n		s.specular = f3(ms.y, ms.z, ms.w);"),
    ("		if(e > 0.3f)\n		{\n			const F3 r = reflect3(dir, normal), shv = sample_hemisphere(rng, b, e);", "		if(e > 0.3f)\n		{\n			" + enter("S_glossy") + "\n			const F3 r = reflect3(dir, normal), shv = sample_hemisphere(rng, b, e);"),
    ("			done = true;\n		}\n		else illum = 1;", "			done = true;\n			" + leave("S_glossy") + "\n		}\n		else illum = 1;"),
    ("		if(illum == 1)\n		{\n			dir = align_direction(sample_hemisphere(rng, b, 0.0f), normal);\n			color = color * diffuse;\n		}",
     "		if(illum == 1)\n		{\n			" + enter("S_diffuse") + "\n			dir = align_direction(sample_hemisphere(rng, b, 0.0f), normal);\n			color = color * diffuse;\n			" + leave("S_diffuse") + "\n		}"),
    ("		else if(illum >= 3 && illum <= 5)\n		{\n			color = color * specular;\n			dir = reflect3(dir, normal);\n		}",
     "		else if(illum >= 3 && illum <= 5)\n		{\n			" + enter("S_mirror") + "\n			color = color * specular;\n			dir = reflect3(dir, normal);\n			" + leave("S_mirror") + "\n		}"),
    ("		else if(illum == 6 || illum == 7)\n		{\n			float eta = ior;", "		else if(illum == 6 || illum == 7)\n		{\n			" + enter("S_dielectric") + "\n			float eta = ior;"),
    ("			else dir = reflect3(dir, normal);\n		}\n	}\n	return alive;", "			else dir = reflect3(dir, normal);\n			" + leave("S_dielectric") + "\n		}\n	}\n	return alive;"),
]
edit("shade.hpp", SHADE_HPP)
pairs = SHADE_PATH + [
    ("						if(early && lane == 0) rel = atomicAdd(&a.cursor[cur], n_sure);", "						if(early && lane == 0) { " + enter("S_early") + " rel = atomicAdd(&a.cursor[cur], n_sure); " + leave("S_early") + " }"),
    ("						while(served < n_dead) // (wave-uniform) the rest: paths that ended unexpectedly, or the home segment has run out\n						{", "						while(served < n_dead) // (wave-uniform) the rest: paths that ended unexpectedly, or the home segment has run out\n						{\n							" + enter("S_fetch_more")),
    ("							served += gn;\n						}", "							served += gn;\n							" + leave("S_fetch_more") + "\n						}"),
    ("			expect = 0u;\n			__builtin_amdgcn_s_sleep(1);", "			" + enter("X_lock_spin") + "\n			expect = 0u;\n			__builtin_amdgcn_s_sleep(1);\n			" + leave("X_lock_spin")),
    ('				asm volatile("; ADYPT_MARK exchange_begin");', "				" + enter("exchange")),
    # the two exclusive branches under the lock (marks only: X_pick is entered exactly when a round follows = `shade`, X_take otherwise = `exchange` - `shade`)
    ("				if(do_shade)\n				{\n					if((uint32_t)lane < take_r)", "				if(do_shade)\n				{\n					" + enter("X_pick") + "\n					if((uint32_t)lane < take_r)"),
    ("					if(lane == 0) { ctl->h_shade = h_s; ctl->h_rare = h_r; ctl->busy = 1u | (n_r << 16); }\n				}\n				else\n				{",
     "					if(lane == 0) { ctl->h_shade = h_s; ctl->h_rare = h_r; ctl->busy = 1u | (n_r << 16); }\n					" + leave("X_pick") + "\n				}\n				else\n				{\n					" + enter("X_take")),
    ("					if(lane == 0 && got) { ctl->n_trace = n_t - got; ctl->h_trace = ring(h_t + got); }\n				}", "					if(lane == 0 && got) { ctl->n_trace = n_t - got; ctl->h_trace = ring(h_t + got); }\n					" + leave("X_take") + "\n				}"),
    ('					asm volatile("; ADYPT_MARK shade_begin");', "					" + enter("shade")),
    ('			asm volatile("; ADYPT_MARK setup_begin");', "			" + enter("setup")),
    ("		if(!skip_trip)\n#include \"traverse_trip.inc\"", "		if(!skip_trip)\n		{\n		" + enter("trip") + "\n#include \"traverse_trip.inc\"\n		" + leave("trip") + "\n		}"),
    ('			asm volatile("; ADYPT_MARK setup_end");', "			" + leave("setup")),
    ('					asm volatile("; ADYPT_MARK shade_end");', "					" + leave("shade")),
    ('				asm volatile("; ADYPT_MARK exchange_end");', "				" + leave("exchange")),
]
if COUNT:
    pairs += [
        ("template <bool STATS, bool SUN>\n__global__ __launch_bounds__(kTraceThreads, STATS ? 4 : ADYPT_PATH_WAVES) void k_path(PathKernArgs K)\n{",
         "template <bool STATS, bool SUN>\n__global__ __launch_bounds__(kTraceThreads, STATS ? 4 : ADYPT_PATH_WAVES) __attribute__((amdgpu_num_sgpr(96))) void k_path(PathKernArgs K)\n{\n	" + zero),
        ("	// ---------------- totals: per wave -> per workgroup (LDS) -> one device atomic per workgroup ----------------",
         "	{ uint32_t bc[6]; " + read + "\n	if(lane == 0) for(int i = 0; i < 6; ++i) atomicAdd(&a.stats->wave_profile[i], " + ("(unsigned long long)bc[i]" if LANES else "(((unsigned long long)(bc[i] >> 16)) << 32) | (unsigned long long)(bc[i] & 0xffffu)") + "); } // (k_path<false> leaves wave_profile alone)\n"
         "	// ---------------- totals: per wave -> per workgroup (LDS) -> one device atomic per workgroup ----------------"),
    ]
edit("path.hpp", pairs)
