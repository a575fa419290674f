"""ma_engine_broadcast_weights (include/meshanything_amd.h; replaces accelerate's initial parameter broadcast, main.py:113-118,146)
with a real RCCL communicator.  One GPU is all a gpurun box has, so the communicator has one rank: the test covers the whole
call path -- librccl resolved at run time, ncclBroadcast on the engine's arena, weights marked loaded -- and that the arena
bytes survive it; the multi-rank data path is covered on CPU by tests/test_dp_gloo.py (gloo, world size 2)."""
import ctypes as C

import numpy as np
import pytest
import torch

from meshanything_amd.config import MAConfig, DTYPE_BF16
from conftest import cached_state_dict

pytestmark = pytest.mark.gpu


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _rccl():
    for name in ("librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"):
        try:
            return C.CDLL(name, mode=C.RTLD_GLOBAL)
        except OSError:
            continue
    pytest.skip("librccl.so not found")


def test_rccl_broadcast_entry_point():
    from meshanything_amd import _lib, dp
    from meshanything_amd.engine import Engine
    rccl = _rccl()
    uid = _UniqueId()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    torch.cuda.set_device(0)
    comm = C.c_void_p()
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0 and comm.value
    try:
        cfg = MAConfig.tiny(dtype=DTYPE_BF16)
        sd = cached_state_dict(cfg)
        host = dp.pack_host_arena(cfg, sd.items())
        eng = Engine(cfg)
        with pytest.raises(_lib.MAError, match="MA_ERR_STATE"):                    # nothing loaded yet
            eng.encode(torch.zeros(1, cfg.n_points, 6).cuda())
        arena = eng.arena_tensor()
        arena.copy_(torch.from_numpy(host).cuda())                                  # rank 0 of a DP job packs, then everyone receives
        torch.cuda.synchronize()
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(eng.lib.ma_engine_broadcast_weights(eng.h, comm, 0, C.c_void_p(stream)), eng.h)
        torch.cuda.synchronize()
        assert np.array_equal(eng.arena_tensor().cpu().numpy(), host)               # root's bytes are what every rank ends up with
        ref = Engine(cfg)
        ref.load_weights(sd.items())
        x = torch.randn(1, cfg.n_points, 6)
        x[..., 3:] /= x[..., 3:].norm(dim=-1, keepdim=True)
        a, _ = eng.encode(x.cuda())
        b, _ = ref.encode(x.cuda())
        assert torch.equal(a, b)                                                    # the broadcast arena is a loaded engine
        assert eng.lib.ma_engine_broadcast_weights(eng.h, None, 0, None) == -1      # null communicator: MA_ERR_INVALID
    finally:
        rccl.ncclCommDestroy(comm)
