"""bench.py's rank -> shapes plan: no 8-GPU node is available to the builder, so the N > 1 layout of the
benchmark is pinned on the CPU -- every rank of an N-GPU run takes 8 distinct shapes, the N ranks cover 0 .. 8N-1 exactly once
(reference: accelerate's batch-sampler sharding, /root/reference/main.py:137-146), and the JSON names "batch=8xN"."""
import importlib.util
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(REPO, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("world", [2, 4, 8])
def test_every_rank_gets_8_distinct_shapes(world):
    b = _bench()
    seen = []
    for rank in range(world):
        pl = b.plan(gpus=world, batch=0, rank=rank, world=world)
        assert pl["batch"] == 8 and len(pl["shapes"]) == 8 and pl["global_batch"] == 8 * world
        assert f"batch=8x{world}" in pl["workload"] and pl["parallelism"].startswith(f"dp{world} ")
        assert pl["scaling"] == "weak"
        seen += pl["shapes"]
    assert sorted(seen) == list(range(8 * world)), "the ranks must cover global shape indices 0 .. 8N-1 exactly once"


def test_single_gpu_is_configs_1():
    b = _bench()
    pl = b.plan(gpus=1, batch=0, rank=0, world=1)
    assert pl["batch"] == 1 and pl["shapes"] == [0] and "configs[1]" in pl["workload"] and "mouse.npy" in pl["workload"]
    pl = b.plan(gpus=1, batch=64, rank=0, world=1, sampling=True)
    assert pl["shapes"] == list(range(64)) and "configs[2]" in pl["workload"] and "top-k 50" in pl["workload"]


def test_world_size_mismatch_is_refused():
    b = _bench()
    with pytest.raises(SystemExit, match="WORLD_SIZE is 8"):
        b.plan(gpus=4, batch=0, rank=0, world=8)
    # and through the command line, with the environment torch.distributed.run would set (no GPU is touched before the check)
    env = dict(os.environ, RANK="3", WORLD_SIZE="8", LOCAL_RANK="3")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE is 8" in (r.stderr + r.stdout)


def test_config_4_is_512_shapes_each_once_with_its_own_uniform_stream():
    """BASELINE.json configs[3] ("batch 512 synthetic clouds sharded data-parallel over 8 x MI355X") = `bench.py --gpus 8 --batch 64 --sampling`:
    every rank steps 64 shapes together with top-k / top-p sampling, rank r owns the contiguous block 64 r .. 64 r + 63 (whole batches
    round-robin over the processes, as accelerate's sampler sharding of main.py:137-146 hands them out), the 8 ranks cover 0 .. 511 exactly
    once, and every (rank, row) pair has its own uniform stream (the in-kernel stream is keyed by (seed, row, step); the seed is per rank)."""
    b = _bench()
    seen, streams = [], set()
    for rank in range(8):
        pl = b.plan(gpus=8, batch=64, rank=rank, world=8, sampling=True)
        assert pl["batch"] == 64 and pl["shapes"] == list(range(64 * rank, 64 * rank + 64)) and pl["global_batch"] == 512
        assert "configs[3]" in pl["workload"] and "batch 512" in pl["workload"] and "top-k 50 / top-p 0.95" in pl["workload"] and "proper" not in pl["workload"]
        assert pl["scaling"] == "weak" and pl["parallelism"].startswith("dp8 ")
        seen += pl["shapes"]
        streams |= {(pl["seed"], row) for row in range(64)}
    assert sorted(seen) == list(range(512))
    assert len(streams) == 512, "two rows of the job would draw from the same uniform stream"
    # the greedy 8 x N default keeps the same seed plumbing
    assert len({b.plan(gpus=4, batch=0, rank=r, world=4)["seed"] for r in range(4)}) == 4
