import os
import sys


def _cpu_budget() -> int:
    """Host threads the CPU oracle may use: the affinity mask, capped by a cgroup CPU quota when there is one, never more
    than 8.  (PyTorch defaults to one intra-op thread per visible core; on a 256-core GPU box -- worse, on one whose cgroup
    only grants a few of them -- that turns every small oracle GEMM into a thread-barrier storm: round 1's driver run of
    this suite was killed at 1200 s for exactly that reason.  bench.py pins its CPU leg the same way.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(8, n))


_THREADS = _cpu_budget()
for _k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ[_k] = str(_THREADS)                    # before torch / numpy spin up their pools
os.environ.setdefault("TOKENIZERS_PARALLELISM", "false")

import pytest  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")
    import torch
    torch.set_num_threads(_THREADS)
    try:
        torch.set_num_interop_threads(max(1, min(4, _THREADS)))
    except RuntimeError:
        pass


# Run order of the GPU suite: the BASELINE.json-shape parity tests first (350M shape, batched MFMA decode, 1600 faces), so that a
# run cut short still carries the evidence that matters; cheap kernel-level tests last.  Stable within a class, so the
# module-scoped engine fixtures (one per precision policy) are still built once.
_FIRST = ("test_full_", "test_v2_scale", "test_fidelity", "test_persistent", "test_batched_mfma_decode", "test_large_batches",
          "test_rccl_")


def pytest_collection_modifyitems(config, items):
    def prio(item):
        name = item.name
        for i, p in enumerate(_FIRST):
            if name.startswith(p):
                return i
        return len(_FIRST)
    items.sort(key=prio)


GOLDEN = os.path.join(REPO, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_SD_CACHE = {}


def cached_state_dict(cfg, **kw):
    """Seeded synthetic checkpoint, built once per session and shape (the 350M layout is 2.4 GB of fp32 and ~13 s of RNG).
    Treat the arrays as read-only."""
    from meshanything_amd.checkpoint import state_dict_spec, synthetic_state_dict
    key = (tuple((k, v[0]) for k, v in state_dict_spec(cfg, kw.get("include_unused", False), kw.get("bert_fused", False)).items()),
           tuple(sorted(kw.items())))
    sd = _SD_CACHE.get(key)
    if sd is None:
        sd = _SD_CACHE[key] = synthetic_state_dict(cfg, **kw)
    return sd


@pytest.fixture(scope="session")
def state_dicts():
    return cached_state_dict
