"""One process per rank of a tile shard, all on ONE device, through the library's process-per-GPU entry points (adypt_comm_*) with the
host-staged transport (ADYPT_COMM_TRANSPORT=host, adypt_amd/csrc/device/host_transport.hpp) in place of RCCL, which refuses two ranks
on one device.  Exercises with world > 1 what a 1-GPU box otherwise cannot: id exchange, ncclCommInitRank-shaped init, the per-rank
counts / strides of the gather, stream ordering of the sends / receives against the tracing kernels, the un-tiling on the root, the
all-reduce and the barrier.

    python tools/comm_world.py launch <world> <scene> <width> <height> <spp>     (GPU-less parent: starts the ranks, prints rank 0's JSON)
    python tools/comm_world.py rank                                              (started by `launch`; RANK / WORLD_SIZE in the environment)
"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def launch(world, scene, w, h, spp):
    # this process never touches the GPU: it only starts the ranks and relays their exit codes
    env = dict(os.environ, ADYPT_COMM_TRANSPORT="host", WORLD_SIZE=str(world), MASTER_PORT=str(20000 + os.getpid() % 20000), ADYPT_RUN_ID="cw%d" % os.getpid())
    env.setdefault("ADYPT_CACHE", os.path.join(tempfile.gettempdir(), "adypt_cache"))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "rank", scene, str(w), str(h), str(spp)], env=dict(env, RANK=str(r), ADYPT_CACHE=os.path.join(env["ADYPT_CACHE"], "cw_rank%d" % r)),
                              stdout=subprocess.PIPE if r == 0 else None) for r in range(world)]
    out = procs[0].communicate()[0].decode()
    codes = [p.wait() for p in procs]
    print(out.strip().splitlines()[-1] if out.strip() else "{}")
    return 0 if all(c == 0 for c in codes) else 1


def rank_main(scene, w, h, spp):
    import numpy as np
    from adypt_amd import api, distributed as D, scenes
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    spec = scenes.make_scene(scene, os.environ["ADYPT_CACHE"], width=w, height=h, pt={"maxBounce": 5, "tmpLifetime": 4, "subpixel": 3})
    api.enable_test_hooks()  # the host-staged transport is a test hook: ignored by the library unless asked for explicitly
    inst = api.Instance()
    assert inst.InitializeFromFile(spec.config_path, shift_seed=4711, device=0, tile_rank=rank, tile_nranks=world)
    pt = inst.m_path_tracer
    pt.CommInit(D.exchange_unique_id(rank, world))
    pt.CommBarrier()
    red = pt.CommAllReduce([float(rank + 1), 10.0 * rank], "sum"), pt.CommAllReduce([float(rank + 1), -float(rank)], "max")
    assert red[0] == [world * (world + 1) / 2.0, 10.0 * world * (world - 1) / 2.0] and red[1] == [float(world), 0.0], red
    results = {}
    for label, n in (("first", spp), ("more", 3)):     # two gathers: the second one re-uses communicator, counts and staging
        pt.TraceAsync(n)                                 # no wait: the gather is ordered behind the frames on the context's stream
        img = pt.CommReadResult()
        results[label] = img
    pt.CommBarrier()
    if rank == 0:
        full = api.Instance()
        assert full.InitializeFromFile(spec.config_path, shift_seed=4711, device=0)
        ok = True
        diffs = []
        fp = full.m_path_tracer
        for label, n in (("first", spp), ("more", 3)):
            fp.Trace(True, n)
            ref = fp.ReadResult()
            d = int((ref.view(np.uint32) != results[label].view(np.uint32)).any(-1).sum())
            diffs.append(d)
            ok = ok and d == 0
        print(json.dumps({"world": world, "scene": scene, "size": [w, h], "spp": [spp, 3], "pixels_that_differ_from_the_one_context_image": diffs,
                          "blocks_per_rank": [int(api.shard_block_count(w, h, r, world)) if hasattr(api, "shard_block_count") else None for r in range(world)], "ok": ok}))
        sys.stdout.flush()
        return 0 if ok else 1
    return 0


if __name__ == "__main__":
    if sys.argv[1] == "launch":
        sys.exit(launch(int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])))
    sys.exit(rank_main(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])))
