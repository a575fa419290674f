"""Data-parallel host logic: one process per GPU, shapes sharded over ranks, weights broadcast once.

Reference analogue: `accelerate launch main.py` -> `accelerator.prepare(dataloader, model)` (main.py:113-118,137-146): the
dataloader is sharded over processes and DDP broadcasts rank 0's parameters when it wraps the model; nothing else is ever
communicated.  Here the same three things are explicit:

  * `shard_indices`     which shapes a rank owns (round-robin, no wrap-around padding: the reference's `even_batches`
                        padding generates duplicate shapes that overwrite the same `{uid}_gen.obj`);
  * `load_weights_dp`   rank 0 packs the checkpoint into the engine's arena, then ONE broadcast of that arena
                        (`torch.distributed` backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests);
  * `gather_to_rank0`   results come home as Python objects on the host (the reference writes per-rank files).

No collective runs per decode step.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .config import MAConfig


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun / accelerate environment (defaults: single process)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Join the job if it has more than one process.  backend None -> "nccl" (RCCL) when a GPU is visible, else "gloo"."""
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Shapes owned by `rank`: i % world == rank.  Every shape is owned exactly once; ranks beyond n_items own none."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, n_items, world))


def batches(indices: Sequence[int], batch_size: int) -> List[List[int]]:
    """DataLoader(batch_size, drop_last=False, shuffle=False) over a rank's shard (main.py:137-142)."""
    return [list(indices[i:i + batch_size]) for i in range(0, len(indices), batch_size)]


def pack_host_arena(cfg: MAConfig, items: Iterable[Tuple[str, object]]) -> np.ndarray:
    """Checkpoint tensors -> the engine's packed weight arena in HOST memory (no GPU needed): what rank 0 broadcasts."""
    lib = _lib.load()
    c = cfg.to_c()
    nbytes = lib.ma_arena_bytes(C.byref(c))
    if nbytes < 0:
        raise _lib.MAError(int(nbytes), "ma_arena_bytes failed")
    arena = np.zeros(int(nbytes), dtype=np.uint8)
    descs, keep = [], []
    from .engine import Engine
    for name, arr in items:
        d, a = Engine._desc(name, arr)
        descs.append(d); keep.append(a)
    arr_t = (_lib.TensorDesc * len(descs))(*descs)
    err = C.create_string_buffer(512)
    rc = lib.ma_pack_weights_host(C.byref(c), arr_t, len(descs), C.c_void_p(arena.ctypes.data), err, 512)
    if rc != 0:
        raise _lib.MAError(rc, err.value.decode())
    return arena


def broadcast_host_arena(arena: Optional[np.ndarray], nbytes: int, src: int = 0) -> np.ndarray:
    """Broadcast a host arena (gloo path / CPU tests).  Ranks != src pass arena=None and receive a fresh array."""
    t = torch.from_numpy(arena) if arena is not None else torch.empty(nbytes, dtype=torch.uint8)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return t.numpy()


def load_weights_dp(engine, items_fn: Callable[[], Iterable[Tuple[str, object]]], rank: int, world: int, src: int = 0,
                    force_broadcast: bool = False) -> None:
    """Rank `src` reads the checkpoint (items_fn is only called there) and packs it into its device arena; every other rank
    receives the arena in ONE broadcast on the device (RCCL over xGMI) -- the DDP initial parameter broadcast of the
    reference, main.py:146 -- and marks its weights loaded."""
    if rank == src:
        engine.load_weights(items_fn())
    if world > 1 or (force_broadcast and dist.is_initialized()):
        arena = engine.arena_tensor()                # uint8 view of the device arena
        dist.broadcast(arena, src=src)
        torch.cuda.synchronize()
        if rank != src:
            engine.mark_weights_loaded()


def gather_to_rank0(obj, rank: int, world: int) -> Optional[List]:
    """Host-side gather of per-rank results (lists / dicts of numpy arrays).  Returns the list on rank 0, None elsewhere."""
    if world == 1:
        return [obj]
    out = [None] * world if rank == 0 else None
    dist.gather_object(obj, out, dst=0)
    return out


def merge_sharded(per_rank: Sequence[Dict[int, object]], n_items: int) -> List[object]:
    """Undo `shard_indices`: per-rank {global index: result} dicts -> one list in input order."""
    merged: Dict[int, object] = {}
    for d in per_rank:
        for k, v in d.items():
            if k in merged:
                raise ValueError(f"shape {k} was produced by two ranks")
            merged[k] = v
    missing = [i for i in range(n_items) if i not in merged]
    if missing:
        raise ValueError(f"shapes {missing[:8]}... were produced by no rank")
    return [merged[i] for i in range(n_items)]
