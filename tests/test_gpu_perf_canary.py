"""A performance canary (VERDICT r5 task 6): the product depends on compiler behaviour it does not control — two -mllvm / -f switches, empty-asm register pinning, a kernel
tuned to its last VGPR — so a toolchain change must show up as a RED TEST, not as a slower BENCH line months later.
  * CPU half: the register budget of the hot kernels as compiled from THIS tree with the Makefile's own flags (6 waves per SIMD = at most 80 VGPRs; the traversal
    loop itself must not spill: the scratch of k_path is the shading round's one register pair).
  * GPU half: k_path's in-kernel rate on the bench scene against the committed line of the driver's command (profiles/r6_bench_json_driver_command.json):
    at least 0.9 of it (the pool's boxes differ by +-1.5 %)."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "adypt_amd", "csrc")


def _resources():
    subprocess.check_call(["make", "-s", "-C", CSRC, "asm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(os.path.join(CSRC, "build", "tracer.s")).read()
    out = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n((?:.*\n)*?)\s+\.wavefront_size:", text):
        body = m.group(2)
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", body).group(1))
        out[m.group(1)] = {"vgpr": g("vgpr_count"), "sgpr_spill": g("sgpr_spill_count"), "vgpr_spill": g("vgpr_spill_count"), "scratch": g("private_segment_fixed_size")}
    return out


def test_register_budget_of_the_hot_kernels():
    r = _resources()
    hot = {"_ZN5adypt6k_pathILb0ELb0EEEvNS_12PathKernArgsE": 24, "_ZN5adypt6k_pathILb0ELb1EEEvNS_12PathKernArgsE": 24,      # (scratch allowed: the shading round's spilled pair(s))
           "_ZN5adypt7k_traceILb0ELb0EEEvNS_9TraceArgsE": 0, "_ZN5adypt7k_traceILb0ELb1EEEvNS_9TraceArgsE": 0,
           "_ZN5adypt14k_trace_cameraILb0ELb0EEEvNS_15TraceCameraArgsE": 0, "_ZN5adypt14k_trace_cameraILb0ELb1EEEvNS_15TraceCameraArgsE": 0}
    for name, scratch in hot.items():
        assert name in r, name
        assert r[name]["vgpr"] <= 80, (name, r[name])            # 6 waves per SIMD
        assert r[name]["scratch"] <= scratch, (name, r[name])
    assert r["_ZN5adypt6k_pathILb0ELb0EEEvNS_12PathKernArgsE"]["sgpr_spill"] <= 32


@pytest.mark.gpu
def test_k_path_rate_against_the_committed_line(scene_cache):
    from adypt_amd import api, scenes
    path = os.path.join(ROOT, "profiles", "r6_bench_json_driver_command.json")
    if not os.path.exists(path):
        pytest.skip("no committed line of the driver's command for this round yet (tools/full_cycle.sh + tools/finalize_bench_profiles.py write it)")
    line = json.loads(open(path).read().strip().splitlines()[-1])
    want = line["roofline"]["kernel_Mrays_s"]
    spec = scenes.make_scene("sponza", scene_cache, width=1920, height=1080, pt={"maxBounce": 8, "tmpLifetime": 16, "stackSize": 24, "subpixel": 8, "clamp": 4.0, "sun": [12.0, 11.0, 10.0]})
    inst = api.Instance()
    assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
    p = inst.m_path_tracer
    p.SetInstrumentation(timing=True)
    p.Trace(True, 5)
    best = 0.0
    for _ in range(3):
        p.Reset(); p.Trace(True, 5); p.ResetStats()
        p.Trace(True, 20)
        s = p.GetStats()
        assert s["path_launches"] == 1
        best = max(best, s["path_rays"] / s["path_ms"] / 1e3)
    assert best >= 0.9 * want, "k_path %.0f Mrays/s in-kernel, the committed line has %.0f: a toolchain or source change cost more than 10 %%" % (best, want)
