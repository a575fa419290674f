"""The N>1 path on CPU: world_size-2 gloo processes run the sharding, the ONE weight-arena broadcast and the result
gather.  The arena is packed by the C-ABI library's host-only entry points (no GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from meshanything_amd import dp                                   # noqa: E402
from meshanything_amd.checkpoint import synthetic_items           # noqa: E402
from meshanything_amd.config import MAConfig, DTYPE_BF16          # noqa: E402


def test_shard_indices_partition():
    for n in (0, 1, 7, 8, 64, 513):
        for world in (1, 2, 3, 8):
            owned = [dp.shard_indices(n, r, world) for r in range(world)]
            flat = sorted(i for o in owned for i in o)
            assert flat == list(range(n))                                  # every shape exactly once
            assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1
    with pytest.raises(ValueError):
        dp.shard_indices(4, 2, 2)
    assert dp.batches([0, 2, 4, 6, 8], 2) == [[0, 2], [4, 6], [8]]


def test_merge_sharded_detects_loss_and_duplicates():
    a, b = {0: "x", 2: "z"}, {1: "y"}
    assert dp.merge_sharded([a, b], 3) == ["x", "y", "z"]
    with pytest.raises(ValueError):
        dp.merge_sharded([a, {}], 3)
    with pytest.raises(ValueError):
        dp.merge_sharded([a, {0: "dup", 1: "y"}], 3)


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank: int, world: int, port: int, n_items: int, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = dp.init_process_group("gloo")
    assert (r, w) == (rank, world)
    cfg = MAConfig.tiny(dtype=DTYPE_BF16)
    # rank 0 reads the checkpoint and packs the arena; the other rank gets it in ONE broadcast
    local = dp.pack_host_arena(cfg, synthetic_items(cfg))          # what this rank WOULD have packed (for the check only)
    got = dp.broadcast_host_arena(local if rank == 0 else None, local.nbytes, src=0)
    same = bool(np.array_equal(got, local))
    # each rank "generates" its shard; results are gathered on the host and merged back into input order
    mine = {i: np.full(3, i, dtype=np.int64) for i in dp.shard_indices(n_items, rank, world)}
    allr = dp.gather_to_rank0(mine, rank, world)
    if rank == 0:
        merged = dp.merge_sharded(allr, n_items)
        ret["ok"] = same and all(int(m[0]) == i for i, m in enumerate(merged)) and len(merged) == n_items
        ret["sizes"] = [len(d) for d in allr]
    else:
        ret[f"same{rank}"] = same
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_broadcast_and_gather():
    world, n_items = 2, 7
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    assert ret["ok"] is True
    assert ret["same1"] is True                  # the broadcast arena is byte-identical to a local pack
    assert list(ret["sizes"]) == [4, 3]
