"""SIMD-occupancy profile of the traversal kernel on the bench workload (instrumented k_trace<true> launches):
how many loop trips, triangle-pair iterations and slab-test phases the waves execute and how many lanes are live in
each — the numbers that price divergence.  python tools/wave_profile.py [--scene sponza] [--frames 4]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--cache", default="/tmp/adypt_scenes")
    a = ap.parse_args()
    from adypt_amd import api, scenes
    os.makedirs(a.cache, exist_ok=True)
    spec = scenes.make_scene(a.scene, a.cache, width=a.width, height=a.height)
    inst = api.Instance()
    assert inst.InitializeFromFile(spec.config_path)
    pt = inst.m_path_tracer
    pt.Trace(True, 1)  # the retrace frame
    pt.ReadResult()
    pt.SetInstrumentation(counters=True)
    pt.ResetStats()
    pt.Trace(True, a.frames)
    pt.ReadResult()
    st, wp = pt.GetStats(), pt.GetWaveProfile()
    rep = dict(wp)
    rep["rays"] = st["rays"]
    rep["nodes_per_ray"] = st["nodes_visited"] / st["rays"]
    rep["tris_per_ray"] = st["tris_tested"] / st["rays"]
    rep["lanes_per_trip"] = wp["trip_lanes"] / max(1, wp["trips"])
    rep["tri_iters_per_trip"] = wp["tri_iters"] / max(1, wp["trips"])
    rep["lanes_per_tri_iter"] = wp["tri_lanes"] / max(1, wp["tri_iters"])
    rep["node_phases_per_trip"] = wp["node_phases"] / max(1, wp["trips"])
    rep["lanes_per_node_phase"] = wp["node_lanes"] / max(1, wp["node_phases"])
    rep["trips_per_ray_x64"] = wp["trips"] * 64 / st["rays"]
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
