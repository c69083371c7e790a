"""The committed measurement files bench.py builds its roofline from must be reproducible from each other (CPU only): the issue model is a pure
function of the counter calibration, the static budget of the loop's blocks (compiled from THIS tree) and the block-entry counts; its executed-instruction
total reproduces what the hardware counted (SQ_INSTS_VALU) within 3 %; and a counter profile that claims to belong to this tree carries every field
bench.py reads."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_issue_model_is_what_its_generator_prints():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "valu_issue_model.py")], stdout=subprocess.PIPE, check=True).stdout
    assert json.loads(out) == json.load(open(os.path.join(ROOT, "profiles", "r6_valu_issue_model.json")))
    m = json.loads(out)
    shares = sum(c["share"] for c in m["classes"])
    assert abs(shares - 1.0) < 0.01
    assert 2.0 < m["avg_issue_cycles_per_inst_architectural"] < m["avg_issue_cycles_per_inst_single_class_loops"] < 8.0
    # the mix is weighted by EXECUTED instructions: blocks x entries must add up to what SQ_INSTS_VALU counted per wave-trip (round 4's static
    # weighting was off by 65 %)
    assert abs(m["model_over_measured"] - 1.0) < 0.03, (m["executed_valu_per_wave_trip_model"], m["executed_valu_per_wave_trip_measured"])


def test_trip_budget_is_of_this_tree():
    """profiles/r6_trip_budget.json = tools/trip_budget.py on the device sources as they are (two gfx950 compiles, no GPU)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trip_budget.py")], stdout=subprocess.PIPE, check=True).stdout
    have = json.load(open(os.path.join(ROOT, "profiles", "r6_trip_budget.json")))
    assert json.loads(out) == have
    assert abs(have["trip_valu_static_marked_build"] - have["trip_valu_static_product_build"]) <= 12  # (the marks are scheduling barriers)


def test_counter_profiles_have_what_bench_reads():
    import bench
    from tools.source_hash import device_source_hash
    for name in (bench.PMC_BENCH, bench.PMC_SANMIGUEL):
        p = json.load(open(os.path.join(ROOT, "profiles", name)))
        for key in ("source_hash", "valu_insts_per_ray", "lane_util", "traffic_bytes_per_ray", "traffic_bytes_per_ray_uncorrected", "TCC_hit_rate",
                    "vmem_rd_insts_per_ray", "rays", "launches", "command"):
            assert key in p, (name, key)
        assert p["kernel"].startswith("k_path<false")
        assert 0.3 < p["lane_util"] <= 1.0 and 20 < p["valu_insts_per_ray"] < 400
        pmc, why = bench.counter_figures(name)
        # either the profile belongs to this tree, or bench.py says why not (and then reports pmc_stale instead of using it)
        assert (pmc is not None) == (p["source_hash"] == device_source_hash()), why
    model = bench.load_profile(bench.ISSUE_MODEL)
    pmc = json.load(open(os.path.join(ROOT, "profiles", bench.PMC_BENCH)))
    r = bench.valu_roofline(pmc, model, 7.0e9, live_clock_ghz=2.25)
    assert r["bound"] == "valu_issue" and r["clock_source"].startswith("live") and 0.5 < r["frac"] < 1.05
    assert abs(r["peak"] - 1024 * 2.25) < 1e-6 and r["frac"] < r["frac_at_single_class_loop_rates"]
