"""Pixel-tile sharding across GPUs: one process per GPU, zero communication while rendering, ONE gather of the fp32
radiance per output frame.

The image is cut into 32x32-pixel blocks; block (bx, by) belongs to rank (bx + by) mod world (diagonal interleave:
sky, floor and geometry-heavy regions are spread evenly).  Every rank renders only its blocks into a compact
block-major RGBA buffer; pixels are independent (pathtracer.glsl:220-227 has no cross-pixel reduction, the Sobol
point is per frame, the shift per pixel), so the assembled image is bit-identical to a single-GPU render.

`gather_radiance` works on any torch.distributed backend: "nccl" (= RCCL over xGMI) on GPUs with device tensors,
"gloo" on CPU (used by the world_size-2 CPU tests).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _native as N

BLOCK = 32
BLOCK_PIXELS = BLOCK * BLOCK
import time as _time


def _launcher_start_time() -> float:
    """Wall-clock time at which this process's PARENT started (the launcher all ranks of a torch.distributed.run job share), 0.0 when /proc does not say.
    An id file older than that cannot belong to this job — whereas a rank's own start time would not do: the ranks of one job reach this point seconds apart
    (a fresh box pages the libraries in for the first rank), and a slow rank must not take rank 0's fresh file for a stale one."""
    import os
    try:
        with open("/proc/%d/stat" % os.getppid()) as f:
            fields = f.read().rsplit(")", 1)[1].split()
        ticks = int(fields[19])  # starttime: field 22 of /proc/pid/stat, in clock ticks since boot
        with open("/proc/uptime") as f:
            uptime = float(f.read().split()[0])
        return _time.time() - uptime + ticks / float(os.sysconf("SC_CLK_TCK"))
    except (OSError, ValueError, IndexError):
        return 0.0


def _proc_start_ticks(pid: int) -> int:
    """starttime of process `pid` (field 22 of /proc/pid/stat, clock ticks since boot); -1 when there is no such process, -2 when /proc does not say."""
    try:
        with open("/proc/%d/stat" % pid) as f:
            return int(f.read().rsplit(")", 1)[1].split()[19])
    except FileNotFoundError:
        return -1
    except (OSError, ValueError, IndexError):
        return -2


def block_count(width: int, height: int, rank: int, world: int) -> int:
    n = N.lib.adypt_shard_block_count(width, height, rank, world)
    if n < 0:
        raise ValueError("bad shard geometry")
    return int(n)


def max_block_count(width: int, height: int, world: int) -> int:
    return max(block_count(width, height, r, world) for r in range(world))


def owner_mask(width: int, height: int, rank: int, world: int) -> np.ndarray:
    """uint8 H x W mask of the pixels rank `rank` renders."""
    by, bx = np.mgrid[0:height, 0:width]
    return (((bx // BLOCK) + (by // BLOCK)) % world == rank).astype(np.uint8)


def tile_from_image(rgba: np.ndarray, rank: int, world: int) -> np.ndarray:
    """H x W x 4 image -> compact block-major float4 buffer of `rank` (inverse of untile; used by CPU tests)."""
    h, w = rgba.shape[:2]
    nbx, nby = (w + BLOCK - 1) // BLOCK, (h + BLOCK - 1) // BLOCK
    out = []
    for by in range(nby):
        for bx in range(nbx):
            if (bx + by) % world != rank:
                continue
            blk = np.zeros((BLOCK, BLOCK, 4), dtype=np.float32)
            src = rgba[by * BLOCK:(by + 1) * BLOCK, bx * BLOCK:(bx + 1) * BLOCK]
            blk[:src.shape[0], :src.shape[1]] = src
            # 16 wave tiles of 8x8 (4 per row), each row-major
            out.append(blk.reshape(4, 8, 4, 8, 4).transpose(0, 2, 1, 3, 4).reshape(BLOCK_PIXELS, 4))
    return np.concatenate(out, axis=0) if out else np.zeros((0, 4), dtype=np.float32)


def untile(width: int, height: int, rank: int, world: int, local_rgba: np.ndarray, rgb: np.ndarray) -> None:
    local_rgba = np.ascontiguousarray(local_rgba, dtype=np.float32)
    assert rgb.dtype == np.float32 and rgb.flags.c_contiguous and rgb.shape == (height, width, 3)
    assert local_rgba.size >= block_count(width, height, rank, world) * BLOCK_PIXELS * 4
    N.check_host(N.lib.adypt_untile_host(width, height, rank, world, local_rgba.ctypes.data, rgb.ctypes.data))


def gather_radiance(local, width: int, height: int, rank: int, world: int, group=None) -> Optional[np.ndarray]:
    """`local`: 1-D float32 torch tensor (device or CPU) holding this rank's compact buffer padded with zeros to
    max_block_count * 1024 * 4 floats.  Returns the assembled H x W x 3 image on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    n = max_block_count(width, height, world) * BLOCK_PIXELS * 4
    assert local.numel() == n and local.dtype == torch.float32
    if world == 1:
        parts = [local]
    else:
        if local.is_cuda and dist.get_backend(group) != "nccl":
            local = local.cpu()  # gloo rehearsal of the N>1 flow: gloo gathers host tensors
        parts = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
        dist.gather(local, gather_list=parts, dst=0, group=group)  # the single collective of the data path
    if rank != 0:
        return None
    rgb = np.zeros((height, width, 3), dtype=np.float32)
    for r in range(world):
        untile(width, height, r, world, parts[r].detach().cpu().numpy(), rgb)
    return rgb


def gather_radiance_device(local, tracer, width: int, height: int, rank: int, world: int, group=None):
    """GPU-resident variant of `gather_radiance` (backend "nccl" = RCCL, or world == 1): the one gather lands in a
    contiguous device buffer on rank 0, which un-tiles all ranks' blocks on the device (`adypt_assemble_radiance`).
    Returns the assembled H x W x 3 float32 *device* tensor on rank 0, None elsewhere — nothing crosses PCIe."""
    import torch
    import torch.distributed as dist
    n = max_block_count(width, height, world) * BLOCK_PIXELS * 4
    assert local.is_cuda and local.numel() == n and local.dtype == torch.float32
    if world == 1:
        gathered = local
    else:
        gathered = torch.empty(world * n, dtype=torch.float32, device=local.device) if rank == 0 else None
        parts = list(gathered.view(world, n).unbind(0)) if rank == 0 else None
        dist.gather(local, gather_list=parts, dst=0, group=group)  # the single collective of the data path
    if rank != 0:
        return None
    rgb = torch.empty((height, width, 3), dtype=torch.float32, device=local.device)
    torch.cuda.current_stream(local.device).synchronize()  # the gather ran on torch's stream, the un-tiling runs on the tracer's
    tracer.assemble_radiance(gathered.data_ptr(), n // 4, rgb.data_ptr())
    return rgb


# ---------------------------------------------------------------------------------------------------------------------
# Native path (no torch): the library's own RCCL communicator.  One process per GPU, launched by any launcher that sets
# RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT (torch.distributed.run does); rank 0 makes the 128-byte RCCL id and the ranks
# of the node exchange it through a file — the only thing the launcher's channel has to carry.
# ---------------------------------------------------------------------------------------------------------------------
def comm_unique_id() -> bytes:
    import ctypes as C
    buf = C.create_string_buffer(128)
    N.check(N.lib.adypt_comm_unique_id(buf))
    return buf.raw


def rendezvous_path() -> str:
    """One file per job, inside a directory only this user can enter (mode 0700, ownership checked): keyed by the launcher's port and
    run id so concurrent or repeated jobs never read a stale id.  Without a run id the launcher's pid (our parent) stands in for it —
    ranks of one torch.distributed.run share it; launchers that start every rank from a different parent must set ADYPT_RUN_ID."""
    import os
    import stat
    import tempfile
    base = os.environ.get("ADYPT_RENDEZVOUS_DIR") or os.path.join(tempfile.gettempdir(), "adypt_rccl_%d" % os.getuid())
    os.makedirs(base, mode=0o700, exist_ok=True)
    st = os.lstat(base)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077 and not os.environ.get("ADYPT_RENDEZVOUS_DIR")):
        raise RuntimeError("rendezvous directory %s is not a private directory of this user" % base)
    run_id = os.environ.get("TORCHELASTIC_RUN_ID") or os.environ.get("ADYPT_RUN_ID")
    key = "%s_%s" % (os.environ.get("MASTER_PORT", "0"), run_id if run_id else "ppid%d" % os.getppid())
    return os.path.join(base, "id_" + "".join(ch if ch.isalnum() or ch in "-_." else "_" for ch in key))


def _write_new(path: str, data: bytes) -> None:
    """Create `path` (it must not exist, symlinks are not followed), mode 0600."""
    import os
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
    with os.fdopen(fd, "wb") as f:
        f.write(data)


def exchange_unique_id(rank: int, world: int, path: Optional[str] = None, make_id=comm_unique_id, timeout_s: float = 120.0) -> bytes:
    """Rank 0 writes the id (exclusive create + atomic rename), the others wait for the file.  Single node only (shared /tmp)."""
    import os
    import time
    path = path or rendezvous_path()
    if world == 1:
        return make_id()
    if rank == 0:
        try:
            os.unlink(path)  # (a stale id of an earlier job: see below)
        except OSError:
            pass
        try:
            uid = make_id()
        except Exception:
            try:
                _write_new(path, b"failed")  # tell the waiting ranks at once instead of letting them run into the timeout
            except OSError:
                pass
            raise
        tmp = path + ".tmp%d" % os.getpid()
        try:
            os.unlink(tmp)
        except OSError:
            pass
        # the id, and who wrote it: "this file belongs to a job that is running" is then a fact a waiting rank can check (the writer — rank 0, blocked in
        # its communicator's creation until the others arrive — is alive, and is the same incarnation of that pid), not an inference from time stamps
        _write_new(tmp, uid + b"|writer=%d:%d\n" % (os.getpid(), _proc_start_ticks(os.getpid())))
        os.replace(tmp, path)
        return uid
    t0 = time.time()
    while True:
        try:
            fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
            with os.fdopen(fd, "rb") as f:
                st = os.fstat(f.fileno())
                if st.st_uid != os.getuid():
                    raise RuntimeError("rendezvous file %s belongs to another user" % path)
                uid = f.read()
            # an id left behind by an EARLIER job that died before rank 0 could remove it (same launcher port, back-to-back runs) must not be taken
            # for this job's.  A file this module wrote names its writer: stale = that process is gone (or its pid belongs to a later process).  Only a
            # bare 128-byte file (written by something else) falls back to the time stamp: it counts if it is younger than this rank's PARENT — which is
            # the job's launcher when one parent starts every rank (torch.distributed.run, tools/comm_world.py), the case that heuristic was made for.
            tag = uid[128:]
            uid = uid[:128]
            if tag.startswith(b"|writer=") and tag.endswith(b"\n"):
                try:
                    wpid, wticks = (int(x) for x in tag[8:-1].split(b":"))
                    now = _proc_start_ticks(wpid)
                    if now == -1 or (now >= 0 and wticks >= 0 and now != wticks):
                        uid = b""
                except ValueError:
                    uid = b""
            elif tag:
                uid = b"" if uid + tag != b"failed" else b"failed"
            elif st.st_mtime < _launcher_start_time() - 2.0:
                uid = b""
            if len(uid) == 128:
                return uid
            if uid == b"failed":
                raise RuntimeError("rank 0 could not create an RCCL id (%s)" % path)
        except OSError:
            pass
        if time.time() - t0 > timeout_s:
            raise TimeoutError("no RCCL id from rank 0 at %s after %.0f s" % (path, timeout_s))
        time.sleep(0.01)
