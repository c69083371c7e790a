"""N>1 path on CPU: world_size-2 gloo processes shard the frame by pixel tile (same ownership function as the HIP
contexts), render their tiles with the oracle, and assemble the image with the single gather of
adypt_amd.distributed — the code path bench.py --gpus N uses with the nccl (RCCL) backend."""
import os
import socket
import sys
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, spp, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adypt_amd import distributed as D, scenes
    from oracle import oracle_py as O
    from tests.helpers import GOLDEN, oracle_scene_from_golden
    sc = oracle_scene_from_golden("tiny0")
    cam = scenes._SCENE_TABLE["tiny0"][3]
    ip, iv = O.camera(cam["fov"], cam["yaw"], cam["pitch"], w, h)
    P = O.make_params(w, h, cam["position"], ip, iv, stack_size=16, max_bounce=4)
    sm = np.fromfile(os.path.join(GOLDEN, "sobol_matrices_64x32.u32"), dtype=np.uint32).reshape(64, 32)
    st = O.PathTracerState(w, h)
    stats = O.pt_frames(sc, P, O.shift_bytes(11, w, h), sm, st, spp, mask=D.owner_mask(w, h, rank, world), n_threads=2)
    local = D.tile_from_image(st.accum, rank, world).reshape(-1)
    assert local.size == D.block_count(w, h, rank, world) * D.BLOCK_PIXELS * 4
    pad = np.zeros(D.max_block_count(w, h, world) * D.BLOCK_PIXELS * 4, dtype=np.float32)
    pad[:local.size] = local
    rgb = D.gather_radiance(torch.from_numpy(pad), w, h, rank, world)
    rays = torch.tensor([stats.rays], dtype=torch.int64)
    dist.all_reduce(rays)  # bench.py sums the units all ranks processed
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # ... and takes the MAX time over ranks
    assert t.item() == float(world)
    if rank == 0:
        np.save(out_path, rgb)
        np.save(out_path + ".rays.npy", rays.numpy())
    else:
        assert rgb is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h", [(2, 100, 75), (2, 64, 64)])
def test_tile_shard_gather_world2_gloo(world, w, h, tmp_path, sobol_matrices):
    from adypt_amd import distributed as D, scenes
    from oracle import oracle_py as O
    from tests.helpers import oracle_scene_from_golden
    out = str(tmp_path / "img.npy")
    spp = 3
    mp.spawn(_worker, args=(world, _free_port(), w, h, spp, out), nprocs=world, join=True)
    got = np.load(out)
    sc = oracle_scene_from_golden("tiny0")
    cam = scenes._SCENE_TABLE["tiny0"][3]
    ip, iv = O.camera(cam["fov"], cam["yaw"], cam["pitch"], w, h)
    P = O.make_params(w, h, cam["position"], ip, iv, stack_size=16, max_bounce=4)
    st = O.PathTracerState(w, h)
    stats = O.pt_frames(sc, P, O.shift_bytes(11, w, h), sobol_matrices, st, spp)
    assert np.array_equal(got.view(np.uint32), st.accum[..., :3].view(np.uint32))  # sharded == single process, bit for bit
    assert int(np.load(out + ".rays.npy")[0]) == stats.rays


def test_ownership_partitions_the_image():
    from adypt_amd import distributed as D
    for (w, h, n) in [(1920, 1080, 8), (4096, 4096, 8), (100, 75, 3), (33, 31, 2), (64, 64, 1)]:
        total = sum(D.owner_mask(w, h, r, n).astype(np.int64) for r in range(n))
        assert (total == 1).all()
        counts = [D.block_count(w, h, r, n) for r in range(n)]
        assert sum(counts) == ((w + 31) // 32) * ((h + 31) // 32)
        assert max(counts) - min(counts) <= max(2, n // 2)  # balanced
        img = np.random.RandomState(0).rand(h, w, 4).astype(np.float32)
        rgb = np.zeros((h, w, 3), np.float32)
        for r in range(n):
            D.untile(w, h, r, n, D.tile_from_image(img, r, n), rgb)
        assert np.array_equal(rgb, img[..., :3])


def _rendezvous_worker(rank, world, path, q):
    from adypt_amd import distributed as D
    uid = D.exchange_unique_id(rank, world, path=path, make_id=lambda: bytes(range(128)), timeout_s=30.0)
    q.put((rank, uid))


def test_native_rendezvous_file_carries_the_id_between_processes(tmp_path):
    """The native (torch-free) multi-process path exchanges ONE thing outside RCCL: the 128-byte communicator id, through a
    file written atomically by rank 0.  RCCL itself needs GPUs; the exchange is checked here with a stand-in id."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    path = str(tmp_path / "id")
    procs = [ctx.Process(target=_rendezvous_worker, args=(r, 3, path, q)) for r in (2, 1, 0)]  # readers start first
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(30)
    assert got == {0: bytes(range(128)), 1: bytes(range(128)), 2: bytes(range(128))}


def test_native_rendezvous_tells_waiting_ranks_when_rank0_cannot_make_an_id(tmp_path):
    """bench.py falls back to torch.distributed when the library cannot bring RCCL up: the waiting ranks must learn that at
    once (not after the timeout), so every rank takes the same branch."""
    from adypt_amd import distributed as D
    path = str(tmp_path / "id")

    def broken():
        raise OSError("librccl.so not found")

    with pytest.raises(OSError):
        D.exchange_unique_id(0, 2, path=path, make_id=broken)
    t0 = time.time()
    with pytest.raises(RuntimeError):
        D.exchange_unique_id(1, 2, path=path, timeout_s=30.0)
    assert time.time() - t0 < 5.0


def test_rendezvous_file_lives_in_a_private_directory_and_is_never_followed_through_a_symlink(tmp_path, monkeypatch):
    """the id file used to sit in shared /tmp under a predictable name, opened with plain open()."""
    import stat
    from adypt_amd import distributed as D
    monkeypatch.delenv("ADYPT_RENDEZVOUS_DIR", raising=False)
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    import tempfile
    tempfile.tempdir = None  # re-read TMPDIR
    try:
        monkeypatch.setenv("MASTER_PORT", "29999")
        monkeypatch.setenv("ADYPT_RUN_ID", "job/../7")  # hostile characters are flattened, the parent pid is not part of the name
        p = D.rendezvous_path()
        d = os.path.dirname(p)
        assert os.path.dirname(d) == str(tmp_path) and stat.S_IMODE(os.lstat(d).st_mode) == 0o700
        assert "/" not in os.path.basename(p) and ".." not in os.path.basename(p).replace("_.._", "") and str(os.getppid()) not in os.path.basename(p)
        # a symlink planted at the file's path is neither written through (rank 0) nor read through (other ranks)
        target = tmp_path / "victim"
        target.write_bytes(b"x" * 128)
        os.symlink(str(target), p)
        with pytest.raises(TimeoutError):
            D.exchange_unique_id(1, 2, path=p, timeout_s=0.3)
        os.unlink(p)
        os.symlink(str(target), p + ".tmp%d" % os.getpid())
        assert D.exchange_unique_id(0, 2, path=p, make_id=lambda: bytes(range(128))) == bytes(range(128))
        assert target.read_bytes() == b"x" * 128 and not os.path.islink(p)
    finally:
        tempfile.tempdir = None


def test_a_stale_id_of_an_earlier_job_is_not_taken_for_this_jobs(tmp_path):
    """Back-to-back jobs of one launcher port (the driver runs N = 1, 2, 4, 8 in a row): an id file a dead job left behind must be ignored by the
    waiting ranks until rank 0 of THIS job has written its own."""
    from adypt_amd import distributed as D
    path = str(tmp_path / "id")
    with open(path, "wb") as f:
        f.write(b"\xee" * 128)
    old = time.time() - 30 * 86400  # (older than this test's parent process, the launcher stand-in)
    os.utime(path, (old, old))
    with pytest.raises(TimeoutError):
        D.exchange_unique_id(1, 2, path=path, timeout_s=0.5)         # the stale id is not returned
    assert D.exchange_unique_id(0, 2, path=path, make_id=lambda: bytes(range(128))) == bytes(range(128))
    assert D.exchange_unique_id(1, 2, path=path, timeout_s=5.0) == bytes(range(128))
    # a file this module wrote names its writer: an id whose writer is gone is stale however young the file is (a rank started by a wrapper of its
    # own, whose parent is younger than rank 0's file, must not reject a LIVE writer's id either: no time stamp is consulted for such files)
    with open(path, "wb") as f:
        f.write(b"\xee" * 128 + b"|writer=%d:%d\n" % (2 ** 22 - 3, 12345))       # no such process
    with pytest.raises(TimeoutError):
        D.exchange_unique_id(1, 2, path=path, timeout_s=0.5)
    with open(path, "wb") as f:
        f.write(b"\xee" * 128 + b"|writer=%d:%d\n" % (os.getpid(), D._proc_start_ticks(os.getpid()) + 1))   # this pid, an earlier incarnation
    with pytest.raises(TimeoutError):
        D.exchange_unique_id(1, 2, path=path, timeout_s=0.5)
    with open(path, "wb") as f:
        f.write(b"\xaa" * 128 + b"|writer=%d:%d\n" % (os.getpid(), D._proc_start_ticks(os.getpid())))
    old = time.time() - 30 * 86400
    os.utime(path, (old, old))                                                   # ... old time stamp, live writer: taken
    assert D.exchange_unique_id(1, 2, path=path, timeout_s=5.0) == b"\xaa" * 128
