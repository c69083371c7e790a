"""The any-hit overload of the reference (shaders/traversal.glsl:257-494): oracle
consistency on CPU, bit-exact GPU parity with -m gpu."""
import numpy as np
import pytest

from oracle import oracle_py as O
from tests.helpers import bits, golden_scene, oracle_scene_from_golden, random_rays


@pytest.mark.parametrize("name", ["tiny0", "tiny1"])
def test_oracle_anyhit_is_consistent_with_closest_hit(name):
    _, _, _, tris, _, _ = golden_scene(name)
    sc = oracle_scene_from_golden(name)
    rays = random_rays(tris, 6000, 21)
    closest, anyh = O.trace(sc, rays, 32), O.trace(sc, rays, 32, any_hit=True)
    # occluded <=> a closest hit exists; the first accepted triangle is never nearer than the closest one
    assert np.array_equal(anyh["tri_id"] >= 0, closest["tri_id"] >= 0)
    occ = anyh["tri_id"] >= 0
    assert (anyh["t"][occ] >= closest["t"][occ]).all()
    # early exit: never more work than the full traversal, identical visit prefix
    assert (anyh["nodes"] <= closest["nodes"]).all() and (anyh["tris"] <= closest["tris"]).all()
    assert (anyh["nodes"][occ] < closest["nodes"][occ]).any() or (anyh["tris"][occ] < closest["tris"][occ]).any()
    miss = ~occ
    assert anyh[miss].tobytes() == closest[miss].tobytes()  # a ray that hits nothing does exactly the same work
    # accepted triangles really are hits: barycentrics in range
    assert ((anyh["u"][occ] >= 0) & (anyh["v"][occ] >= 0) & (anyh["u"][occ] + anyh["v"][occ] <= 1)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h", [("tiny0", 64, 48), ("sibenik", 64, 36), ("sponza", 64, 36)])
def test_gpu_anyhit_bit_exact(name, w, h, scene_cache):
    from adypt_amd import api, scenes
    from tests.helpers import oracle_scene_from_instance
    spec = scenes.make_scene(name, scene_cache, width=w, height=h)
    inst = api.Instance()
    assert inst.InitializeFromFile(spec.config_path, shift_seed=1)
    osc = oracle_scene_from_instance(inst)
    rays = random_rays(inst.scene.triangles, 50000, 8)
    g = inst.m_path_tracer.TraceRays(rays, with_stats=True, any_hit=True)
    o = O.trace(osc, rays, inst.m_config.c.stack_size, any_hit=True)
    assert g.tobytes() == o.tobytes()  # first accepted triangle, u/v/t bits, nodes, tris, visit hash, max depth
    g2 = inst.m_path_tracer.TraceRays(rays, with_stats=False, any_hit=True)
    assert np.array_equal(g2["tri_id"], o["tri_id"]) and np.array_equal(bits(g2["t"]), bits(o["t"]))
    # closest-hit queries are unaffected by the any-hit instantiation
    assert inst.m_path_tracer.TraceRays(rays, with_stats=True).tobytes() == O.trace(osc, rays, inst.m_config.c.stack_size).tobytes()
