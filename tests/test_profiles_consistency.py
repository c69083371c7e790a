"""The committed measurement files bench.py builds its roofline from must be reproducible from each other (CPU only, no compiler):
the issue model is a pure function of the counter calibration and the instruction mix, and a counter profile that claims to belong to this
tree carries every field bench.py reads."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_issue_model_is_what_its_generator_prints():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "valu_issue_model.py")], stdout=subprocess.PIPE, check=True).stdout
    assert json.loads(out) == json.load(open(os.path.join(ROOT, "profiles", "r4_valu_issue_model.json")))
    m = json.loads(out)
    shares = sum(c["share"] for c in m["classes"])
    assert abs(shares - 1.0) < 0.01
    assert 2.0 < m["avg_issue_cycles_per_inst_architectural"] < m["avg_issue_cycles_per_inst_single_class_loops"] < 8.0


def test_counter_profiles_have_what_bench_reads():
    import bench
    from tools.source_hash import device_source_hash
    for name in (bench.PMC_BENCH, bench.PMC_SANMIGUEL):
        p = json.load(open(os.path.join(ROOT, "profiles", name)))
        for key in ("source_hash", "valu_insts_per_ray", "lane_util", "traffic_bytes_per_ray", "traffic_bytes_per_ray_uncorrected", "TCC_hit_rate",
                    "vmem_rd_insts_per_ray", "rays", "launches", "command"):
            assert key in p, (name, key)
        assert p["kernel"] == "k_path<false>"
        assert 0.3 < p["lane_util"] <= 1.0 and 20 < p["valu_insts_per_ray"] < 400
        pmc, why = bench.counter_figures(name)
        # either the profile belongs to this tree, or bench.py says why not (and then reports pmc_stale instead of using it)
        assert (pmc is not None) == (p["source_hash"] == device_source_hash()), why
    model = bench.load_profile(bench.ISSUE_MODEL)
    pmc = json.load(open(os.path.join(ROOT, "profiles", bench.PMC_BENCH)))
    r = bench.valu_roofline(pmc, model, 7.0e9, live_clock_ghz=2.25)
    assert r["bound"] == "valu_issue" and r["clock_source"].startswith("live") and 0.5 < r["frac"] < 1.05
    assert abs(r["peak"] - 1024 * 2.25) < 1e-6 and r["frac"] < r["frac_at_single_class_loop_rates"]
