"""Developer tool (GPU box): how often a wave of k_path ENTERS each block of its persistent loop (the blocks behind a wave-level branch), on the bench
workload.  Needs the counting variant:
    ADYPT_BLOCKS_COUNT=1 tools/build_variant.sh blockcnt --transform adypt_amd/csrc/measure/k_path_blocks.py
    ADYPT_LIB=adypt_amd/libadypt_blockcnt.so python tools/path_block_counts.py > profiles/r6_k_path_block_counts.json
    ADYPT_BLOCKS_COUNT=1 ADYPT_BLOCKS_SET=shade tools/build_variant.sh shadecnt --transform adypt_amd/csrc/measure/k_path_blocks.py
    ADYPT_BLOCKS_SET=shade ADYPT_LIB=adypt_amd/libadypt_shadecnt.so python tools/path_block_counts.py > profiles/r6_k_path_shade_block_counts.json
    (and the same with `rare` / rarecnt -> profiles/r6_k_path_rare_block_counts.json)
With tools/trip_budget.py's static counts these are the EXECUTED vector instructions per trip, which tools/valu_issue_model.py checks against SQ_INSTS_VALU."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes, _native as N
SET = os.environ.get("ADYPT_BLOCKS_SET", "trip")  # which counting variant ADYPT_LIB is: trip | shade | rare
SETS = {"trip": ["setup", "exchange", "shade", "trip", "A_pop", "A_choose", "A_push", "B_tri_load", "B_node_load", "C_woop", "D_slab", "E_flush"],
        "shade": ["shade", "S_parked", "S_miss", "S_surface", "S_textured", "S_glossy", "S_diffuse", "S_mirror", "S_dielectric", "S_dead", "S_alive", "S_replace"],
        "rare": ["trip", "shade", "A_pop_spill", "A_push_spill", "div_slow", "S_fetch_more", "S_early", "X_lock_spin", "exchange", "setup", "F_try", "S_defer"]}  # = adypt_amd/csrc/measure/k_path_blocks.py
LANES = os.environ.get("ADYPT_BLOCKS_LANES", "0") != "0"  # the variant was built with ADYPT_BLOCKS_LANES=1: the counters hold active lanes, five per pass
SETS_LANES = {"trip": ["trip", "A_choose", "C_woop", "D_slab", "E_flush"], "shade": ["S_surface", "S_textured", "S_glossy", "S_diffuse", "S_dielectric"],
              "rare": ["S_miss", "S_mirror", "S_dead", "S_alive", "S_replace"], "wait": ["W_idle", "W_wait", "W_two", "W_three", "W_four"]}
NAMES = SETS_LANES[SET] if LANES else SETS[SET]
scene = os.environ.get("SWEEP_SCENE", "sponza"); fr = int(os.environ.get("SWEEP_FRAMES", "20")); warm = int(os.environ.get("SWEEP_WARMUP", "5"))
spec = scenes.make_scene(scene, os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080,
                         pt={"maxBounce": 8, "tmpLifetime": 16, "stackSize": 24, "subpixel": 8, "clamp": 4.0, "sun": [12.0, 11.0, 10.0]})
inst = api.Instance(); assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
p = inst.m_path_tracer
p.SetInstrumentation(timing=True)
p.Trace(True, warm); p.DeviceSynchronize(); p.ResetStats()
p.Trace(True, fr); p.DeviceSynchronize()
s = p.GetStats(); w = list(p.GetWaveProfile().values())
if LANES:
    lanes = {n: int(w[i]) for i, n in enumerate(NAMES)}
    print(json.dumps({"lib": os.path.basename(N.LIB_PATH), "set": SET, "scene": scene, "frames": fr, "warmup": warm, "k_path_rays": int(s["path_rays"]), "shaded": int(s.get("shaded", 0)),
                      "lanes_entered": lanes, "lanes_per_ray": {n: round(c / max(1, s["path_rays"]), 5) for n, c in lanes.items()},
                      "note": "lanes_entered[b] = active lanes (s_bcnt1 of exec) summed over every entry of a wave into block b"}))
    sys.exit(0)
counts = {}
for i, n in enumerate(NAMES):
    v = w[i // 2]
    counts[n] = int((v >> 32) if (i & 1) else (v & 0xffffffff))
trips = counts.get("trip", 0)
if SET != "trip":
    print(json.dumps({"lib": os.path.basename(N.LIB_PATH), "set": SET, "scene": scene, "frames": fr, "warmup": warm, "k_path_rays": int(s["path_rays"]), "wave_entries": counts,
                      "entries_per_round": {n: round(c / max(1, counts["shade"]), 4) for n, c in counts.items()}}))
    sys.exit(0)
print(json.dumps({"lib": os.path.basename(N.LIB_PATH), "scene": scene, "frames": fr, "warmup": warm, "k_path_launches": s["path_launches"], "k_path_rays": int(s["path_rays"]),
                  "wave_entries": counts, "entries_per_trip": {n: round(c / max(1, trips), 4) for n, c in counts.items()},
                  "wave_trips_per_ray": round(trips / max(1, s["path_rays"]), 5),
                  "note": "wave_entries[b] = waves that entered block b, summed over the launch(es): the block's vector instructions were issued that often, whatever the lanes' masks"}))
