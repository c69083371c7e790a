"""Developer tool (CPU): profiles/r6_k_path_lane_counts.json from what tools/full_cycle.sh left under gpurun_out/r6/counts — the ACTIVE LANES summed over every entry of a
wave into a block (lanes_*.json: the four lane-counting variants) over the block's wave entries (trip.json, shade.json, rare.json) = the block's lane occupancy; and why
lanes sit out a trip's slab test (the `wait` pass).    python tools/lane_counts_profile.py [dir] > profiles/r6_k_path_lane_counts.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r6", "counts")
ld = lambda n: json.loads([l for l in open(os.path.join(C, n)).read().strip().splitlines() if l.startswith("{")][-1])
entries = {}
for n in ("trip.json", "shade.json", "rare.json"):
    entries.update(ld(n)["wave_entries"])
trip = ld("trip.json")
rays = trip["k_path_rays"]
sets = {}
for n in ("lanes_trip.json", "lanes_shade.json", "lanes_rare.json"):
    for b, lanes in ld(n)["lanes_entered"].items():
        e = entries.get(b)
        sets[b] = {"lanes": lanes, "lanes_per_ray": round(lanes / rays, 5), "wave_entries": e, "lane_occupancy": round(lanes / (64.0 * e), 4) if e else None}
lane_trips = sets["trip"]["lanes"]
WHAT = {"W_idle": "lanes without a ray (finished, waiting for the wave's next exchange)", "W_wait": "lanes whose node still has triangles after what this trip's list took (their pending node waits a trip)",
        "W_two": "lanes with two or more triangles at hand", "W_three": "... three or more", "W_four": "... four or more (the fourth waits: a lane offers three per trip)"}
wait = {b: {"lanes": v, "lanes_per_ray": round(v / rays, 5), "share_of_lane_trips": round(v / lane_trips, 4), "what": WHAT.get(b, "")} for b, v in ld("lanes_wait.json")["lanes_entered"].items()}
print(json.dumps({"what": "k_path<false, false> on the bench workload (20 frames after 5 warm-up): ACTIVE LANES summed over every entry of a wave into a block (s_bcnt1 of exec; counting variants of "
                          "adypt_amd/csrc/measure/k_path_blocks.py with ADYPT_BLOCKS_LANES=1, tools/path_block_counts.py).  lanes / (64 x the block's wave entries) = the block's lane occupancy.",
                  "sets": sets, "k_path_rays": rays, "wave_trips_per_ray": trip["wave_trips_per_ray"], "lane_trips_per_ray": round(lane_trips / rays, 4),
                  "why_lanes_sit_out_the_slab_test": {"what": "probe blocks of the counting variant ADYPT_BLOCKS_SET=wait (exist in that pass only); lane-trips per ray", "sets": wait},
                  "round_5": {"lane_trips_per_ray": 13.5186, "wave_trips_per_ray": 0.21119, "W_idle": 1.5931, "W_wait": 1.35747, "C_woop_lane_occupancy": 0.3751, "D_slab_lane_occupancy": 0.7469,
                              "source": "profiles/history/r5_k_path_lane_counts.json"}}, indent=1))
