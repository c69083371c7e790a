"""-m gpu: the slot-claim audit (adypt_set_instrumentation flag 4) over thousands of launches of the kernels that compact their survivors with
append_slot — k_gen_primary, k_shade, k_shade_first — in the conditions round 3's queue corruption appeared in: several sub-batch chains on
separate streams (adypt_set_pipeline(4)), 8+ frames in flight, launch after launch.  Every queue those kernels append to is poisoned before the
launch and checked after it: each slot below the segment's counter written exactly once by a distinct path, nothing above it.  One error fails
the test; the images must also equal an un-audited run's."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from adypt_amd import api, scenes  # noqa: E402
from tests.helpers import bits  # noqa: E402
from tests.test_gpu_fused_bounces import environment  # noqa: E402

PT = {"tmpLifetime": 4, "maxBounce": 8, "subpixel": 2, "stackSize": 24}


def _tracer(cache, env, pipeline, fused, fif=8):
    spec = scenes.make_scene("tiny0", cache, width=256, height=160, pt=PT)
    with environment(**env):
        inst = api.Instance()
        assert inst.InitializeFromFile(spec.config_path, shift_seed=9)
    p = inst.m_path_tracer
    p.SetFramesInFlight(fif)
    p.SetPipeline(pipeline)
    p.SetFusedBounces(fused)
    return inst, p


@pytest.mark.parametrize("what,env,pipeline,fused,launches_per_batch", [
    ("k_shade_first + k_shade on 4 chains", {}, 4, False, 4 * 8),
    ("k_gen_primary + k_shade on 4 chains", {"ADYPT_FIRST_FUSED": "0"}, 4, False, 4 * 9),
    ("k_shade_first feeding k_path", {}, 1, True, 1),
    ("k_gen_primary + k_shade, one chain, 13 frames in flight", {"ADYPT_FIRST_FUSED": "0"}, 1, False, 9),
])
def test_every_queue_slot_is_claimed_exactly_once(what, env, pipeline, fused, launches_per_batch, scene_cache):
    fif = 13 if "13 frames" in what else 8
    inst, p = _tracer(scene_cache, env, pipeline, fused, fif)
    batches = max(8, (2000 + launches_per_batch - 1) // launches_per_batch) if launches_per_batch > 1 else 600
    p.SetInstrumentation(audit=True)
    for _ in range(batches):
        p.Trace(True, fif)          # one batch of `fif` frames: one appending launch per chain and bounce (+ the re-tracing frames' camera rays)
    st = p.GetStats()
    assert st["audit_errors"] == 0, what
    audited = p.ReadResult()
    # the same frames without the audit: the audit only reads and poisons what the next launch overwrites
    inst2, q = _tracer(scene_cache, env, pipeline, fused, fif)
    q.Trace(True, batches * fif)
    assert np.array_equal(bits(audited), bits(q.ReadResult()))


def test_the_audit_sees_a_planted_double_claim(scene_cache):
    """The detector detects: with ADYPT_AUDIT_SELFTEST=1 the check kernel is handed a queue in which two slots hold the same path."""
    inst, p = _tracer(scene_cache, {"ADYPT_AUDIT_SELFTEST": "1"}, 1, False)
    p.SetInstrumentation(audit=True)
    p.Trace(True, 8)
    assert p.GetStats()["audit_errors"] > 0
