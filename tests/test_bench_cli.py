"""bench.py's --gpus contract: N GPUs however it is started — a launcher's WORLD_SIZE, or none (one process drives N devices through
adypt_create_multi) — and never a silent fallback to fewer devices than asked for."""
import json
import time
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--scene", "tiny0", "--width", "192", "--height", "128", "--no-cpu-baseline", "--no-hbm-block", "--no-single-frame", "--no-extra-blocks"]


def _bench(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, cwd=ROOT)


def test_more_gpus_than_the_box_has_exits_non_zero(tmp_path):
    """No launcher, --gpus 8: on this box (no GPU at all, or fewer than 8) the run must fail loudly — not print an n_gpus 1 line."""
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("the box has 8 devices")
    r = _bench(["--gpus", "8", "--steps", "2", "--warmup", "1", "--cache", str(tmp_path)] + SMALL)
    assert r.returncode != 0
    assert b"never runs on fewer GPUs" in r.stderr and not r.stdout.strip()


def test_gpus_must_match_a_launchers_world_size(tmp_path):
    r = _bench(["--gpus", "4", "--steps", "2", "--warmup", "1", "--cache", str(tmp_path)] + SMALL, env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and b"WORLD_SIZE=2" in r.stderr


@pytest.mark.gpu
def test_single_process_multi_device_path_on_one_card(tmp_path):
    """The one-process path of `bench.py --gpus 3` with the three tile shards on ONE device (ADYPT_MULTI_SHARED_DEVICE, a test hook: the line
    says so and is not a measurement): three contexts, the same image as the 1-GPU run, per-device statistics."""
    one = _bench(["--gpus", "1", "--steps", "4", "--warmup", "2", "--repeats", "1", "--cache", str(tmp_path)] + SMALL)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    three = _bench(["--gpus", "3", "--steps", "4", "--warmup", "2", "--repeats", "2", "--rehearsal", "--selfcheck", "--cache", str(tmp_path)] + SMALL, env={"ADYPT_MULTI_SHARED_DEVICE": "1"})
    assert three.returncode == 0, three.stderr.decode()[-2000:]
    a, b = json.loads(one.stdout.decode().strip().splitlines()[-1]), json.loads(three.stdout.decode().strip().splitlines()[-1])
    assert a["n_gpus"] == 1 and b["n_gpus"] == 3
    assert b["image_mean"] == a["image_mean"]
    assert "not a measurement" in b["config"]["comm"]
    pr = b["per_rank"]
    assert len(pr["rays"]) == 3 and sum(pr["rays"]) == a["config"]["rays_per_step"] * 4 and all(r > 0 for r in pr["rays"])
    assert b["repeats"] == 2 and len(b["ms_per_step_all"]) == 2 and b["value_min"] <= b["value"] <= b["value_max"]
    assert b["config"]["selfcheck"] == {"frames": 2, "words_differing": 0, "image_mean": b["config"]["selfcheck"]["image_mean"]} and b["config"]["rehearsal"] is True
    assert len(b["config"]["setup_s_per_device"]) == 3 and all(t >= 0 for t in b["config"]["setup_s_per_device"]) and b["config"]["setup_s_per_device"][0] > 0
    # without the hook the same command must refuse: the box has one device
    real = _bench(["--gpus", "3", "--steps", "4", "--warmup", "2", "--cache", str(tmp_path)] + SMALL)
    assert real.returncode != 0 and not real.stdout.strip()
    # ... and the ENVIRONMENT ALONE must not switch the hook on (the shipped library ignores it without adypt_enable_test_hooks)
    env_only = _bench(["--gpus", "3", "--steps", "4", "--warmup", "2", "--cache", str(tmp_path)] + SMALL, env={"ADYPT_MULTI_SHARED_DEVICE": "1"})
    assert env_only.returncode != 0 and not env_only.stdout.strip()


@pytest.mark.gpu
def test_a_gather_that_never_finishes_ends_the_process(tmp_path):
    """The gather watchdog (multi.hip): with the stall hook the gather of `bench.py --gpus 2` never proceeds; after ADYPT_GATHER_TIMEOUT seconds the
    process must print where it stands and exit with code 86 — not hang, not re-exec."""
    t0 = time.time()
    r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--repeats", "1", "--rehearsal", "--cache", str(tmp_path)] + SMALL,
               env={"ADYPT_MULTI_SHARED_DEVICE": "1", "ADYPT_GATHER_STALL_TEST": "1", "ADYPT_GATHER_TIMEOUT": "3"}, timeout=300)
    assert r.returncode == 86, (r.returncode, r.stderr.decode()[-1500:])
    err = r.stderr.decode()
    assert "gather watchdog" in err and "rank 0" in err and "rank 1" in err and "stalled by ADYPT_GATHER_STALL_TEST" in err
    assert not r.stdout.strip() and time.time() - t0 < 280


@pytest.mark.gpu
def test_process_per_gpu_path_of_the_bench_with_two_ranks_on_one_card(tmp_path):
    """What the driver's launcher starts — one bench.py process per rank, RANK / WORLD_SIZE in the environment — with the two ranks sharing the one
    card through the host-staged TEST transport (RCCL refuses two ranks on one device): the repeats, the barriers, the gather and the per-rank
    table run with world 2, and the image is the 1-GPU image."""
    one = _bench(["--gpus", "1", "--steps", "4", "--warmup", "2", "--repeats", "1", "--cache", str(tmp_path / "one")] + SMALL)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    env = {"WORLD_SIZE": "2", "ADYPT_COMM_TRANSPORT": "host", "ADYPT_BENCH_DEVICE": "0", "MASTER_PORT": str(20000 + os.getpid() % 20000), "ADYPT_RUN_ID": "bench%d" % os.getpid()}
    procs = []
    for r in range(2):
        e = dict(os.environ)
        e.update(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--repeats", "3", "--rehearsal", "--selfcheck", "--cache", str(tmp_path / "two")] + SMALL,
                                      env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT))
    outs = [p.communicate(timeout=600) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], (outs[0][1].decode()[-1500:], outs[1][1].decode()[-1500:])
    a, b = json.loads(one.stdout.decode().strip().splitlines()[-1]), json.loads(outs[0][0].decode().strip().splitlines()[-1])
    assert not outs[1][0].strip()  # only rank 0 prints
    assert b["n_gpus"] == 2 and b["image_mean"] == a["image_mean"] and "not a measurement" in b["config"]["comm"]
    assert len(b["per_rank"]["rays"]) == 2 and sum(b["per_rank"]["rays"]) == a["config"]["rays_per_step"] * 4
    assert b["repeats"] == 3 and len(b["ms_per_step_all"]) == 3
    assert b["config"]["selfcheck"]["words_differing"] == 0
