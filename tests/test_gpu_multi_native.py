"""-m gpu: the library's own multi-GPU boundary (include/adypt_hip.h "native multi-GPU", csrc/device/multi.hip).  A GPU box
has ONE card, and RCCL refuses two ranks on one device, so what can run here is the n_dev = 1 / world = 1 case of every entry
point — RCCL is loaded, a communicator is created on the device, the gather / un-tiling / read-back path and the control-plane
helpers run — plus the argument checking.  The N > 1 shard geometry the gather relies on is covered by the tile-shard tests
(test_gpu_parity.py, test_gpu_large.py) and the world-2 CPU test (test_distributed_cpu.py)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

from adypt_amd import api, distributed as D, _native as N  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from tests.helpers import bits, oracle_params_from_config, oracle_scene_from_instance  # noqa: E402
from tests.test_gpu_parity import make_instance  # noqa: E402


def _multi(inst, devices=(0,), seed=99):
    c = inst.m_config
    m = api.MultiPathTracer()
    m.Initialize(c.pt_params(seed), inst.m_hipscene, c.m_width, c.m_height, devices)
    ip, iv = inst.m_camera.matrices()
    m.SetCamera(ip, iv, inst.m_camera.position)
    return m


def test_multi_single_device_with_rccl_communicator(scene_cache, sobol_matrices):
    inst = make_instance(scene_cache, "sponza", 320, 200, pt={"maxBounce": 6, "stackSize": 24, "tmpLifetime": 4})
    c = inst.m_config.c
    m = _multi(inst)
    assert m.DeviceCount() == 1
    m.CommInit()  # ncclCommInitAll on one device: RCCL loads and initialises on this box
    osc, P = oracle_scene_from_instance(inst), oracle_params_from_config(c)
    m.Trace(False)
    rgba, _, _ = O.primary_frame(osc, P, 0)
    assert np.array_equal(bits(m.ReadResult()), bits(rgba[..., :3]))
    m.Trace(True, 9)
    st = O.PathTracerState(c.width, c.height)
    ost = O.pt_frames(osc, P, O.shift_bytes(99, c.width, c.height), sobol_matrices, st, 9).as_dict()
    img = m.ReadResult()
    assert np.array_equal(bits(img), bits(st.accum[..., :3])) and m.GetSPP() == 9
    assert m.ContextStats(0)["rays"] >= ost["rays"]
    # the same frames through a plain context
    inst.m_path_tracer.Trace(True, 9)
    assert np.array_equal(bits(inst.m_path_tracer.ReadResult()), bits(img))
    # the assembled image stays resident in HBM for callers that want it there (adypt_multi_read_radiance copies this buffer)
    assert m.GatherDevice()
    # one frame per call with look-ahead, like the C++ binding
    m.Reset()
    m.SetLookahead(True)
    st2 = O.PathTracerState(c.width, c.height)
    for _ in range(5):
        m.Trace(True, 1)
        O.pt_frames(osc, P, O.shift_bytes(99, c.width, c.height), sobol_matrices, st2, 1)
        assert np.array_equal(bits(m.ReadResult()), bits(st2.accum[..., :3]))
    m.destroy()


def test_multi_without_communicator_and_argument_errors(scene_cache):
    inst = make_instance(scene_cache, "tiny0", 96, 64)
    m = _multi(inst)
    m.Trace(True, 3)          # n_dev = 1: the gather needs no communicator and does not create one
    inst.m_path_tracer.Trace(True, 3)
    assert np.array_equal(bits(m.ReadResult()), bits(inst.m_path_tracer.ReadResult()))
    m.destroy()
    with pytest.raises(N.AdyptError) as e:
        _multi(inst, devices=(0, 0))  # RCCL needs distinct devices
    assert e.value.code == N.E_INVALID and "twice" in str(e.value)
    with pytest.raises(N.AdyptError):
        _multi(inst, devices=(0, 99))


def test_comm_api_single_rank(scene_cache):
    inst = make_instance(scene_cache, "tiny0", 100, 75)
    pt = inst.m_path_tracer
    pt.Trace(True, 4)
    ref = pt.ReadResult()
    uid = D.exchange_unique_id(0, 1)
    assert len(uid) == 128 and any(uid)
    pt.CommInit(uid)          # ncclCommInitRank(nranks = 1)
    assert np.array_equal(bits(pt.CommReadResult()), bits(ref))
    assert pt.CommGatherDevice()
    assert pt.CommAllReduce([3.0, -5.0], "max") == [3.0, -5.0] and pt.CommAllReduce([2.5], "sum") == [2.5]
    pt.CommBarrier()
    pt.DeviceSynchronize()
    # a sharded context without a communicator must say so instead of hanging
    part = make_instance(scene_cache, "tiny0", 100, 75, rank=1, world=2)
    part.m_path_tracer.Trace(True, 1)
    with pytest.raises(N.AdyptError) as e:
        part.m_path_tracer.CommGatherDevice()
    assert e.value.code == N.E_STATE


@pytest.mark.parametrize("name,w,h,n_dev", [("tiny0", 200, 120, 3), ("sponza", 320, 200, 2), ("tiny0", 64, 36, 4), ("sibenik", 160, 90, 8)])
def test_multi_fan_out_on_one_device(name, w, h, n_dev, scene_cache, sobol_matrices, monkeypatch):
    """Everything of adypt_multi except the RCCL transport, with N > 1, on the one GPU a test box has (ADYPT_MULTI_SHARED_DEVICE:
    the shards share the device, the peer -> root transfers are device copies): the fan-out of every call, the tile shards
    (including shards that own no block: 64x36 on 4), the gather layout and the un-tiling must reproduce the 1-context frames."""
    monkeypatch.setenv("ADYPT_MULTI_SHARED_DEVICE", "1")
    inst = make_instance(scene_cache, name, w, h, pt={"maxBounce": 5, "stackSize": 24, "tmpLifetime": 3})
    c = inst.m_config.c
    m = _multi(inst, devices=(0,) * n_dev)
    assert m.DeviceCount() == n_dev
    osc, P = oracle_scene_from_instance(inst), oracle_params_from_config(c)
    m.m_viewer_type = 4
    m.Trace(False)
    rgba, _, _ = O.primary_frame(osc, P, 4)
    assert np.array_equal(bits(m.ReadResult()), bits(rgba[..., :3]))
    m.Trace(True, 7)
    st = O.PathTracerState(c.width, c.height)
    ost = O.pt_frames(osc, P, O.shift_bytes(99, c.width, c.height), sobol_matrices, st, 7).as_dict()
    assert np.array_equal(bits(m.ReadResult()), bits(st.accum[..., :3])) and m.GetSPP() == 7
    assert sum(m.ContextStats(i)["rays"] for i in range(n_dev)) == ost["rays"] + c.width * c.height  # + the viewer frame's primaries
    assert m.GetStats()["rays"] == ost["rays"] + c.width * c.height   # adypt_multi_get_stats: counts summed over the devices
    assert np.array_equal(m.ReadDisplay(), O.display(st.accum, 3))         # every shard converts its own tiles: together the window
    # one frame per call with look-ahead on every shard
    m.Reset()
    m.SetLookahead(True)
    st2 = O.PathTracerState(c.width, c.height)
    for _ in range(4):
        m.Trace(True, 1)
        O.pt_frames(osc, P, O.shift_bytes(99, c.width, c.height), sobol_matrices, st2, 1)
        assert np.array_equal(bits(m.ReadResult()), bits(st2.accum[..., :3]))
    with pytest.raises(N.AdyptError):
        m.CommInit()  # no communicator in this mode
    m.destroy()


@pytest.mark.parametrize("world,w,h", [(2, 160, 96), (3, 100, 75), (4, 64, 36)])
def test_process_per_gpu_gather_with_world_greater_than_one(world, w, h):
    """adypt_comm_* with world = 2, 3, 4 PROCESSES on the one device: the same code path as over RCCL (counts per rank, strides, grouped
    send / receive enqueued on the contexts' streams behind the tracing kernels, un-tiling on the root, all-reduce, barrier), the bytes
    carried by the host-staged test transport instead (ADYPT_COMM_TRANSPORT=host: RCCL refuses two ranks on one device).  Rank 0's
    assembled image must equal the one-context image bit for bit, twice in a row; (4, 64 x 36) includes a rank that owns no block.
    The ranks are started by a launcher process that never touches the GPU (tools/comm_world.py)."""
    import json, subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "comm_world.py"), "launch", str(world), "tiny0", str(w), str(h), "5"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert out["ok"] and out["world"] == world and out["pixels_that_differ_from_the_one_context_image"] == [0, 0], out
