import os
import sys
import time


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


# A GPU box (/dev/kfd) is a shared host (profiles/r03_diag_cpu_host.txt: loadavg 8-17 of 256 cores with nothing of ours running, a
# 16-core cgroup quota): there the suite keeps its arithmetic on the GPU (oracle on torch-ROCm, kernel references in fp64 torch-ROCm)
# and whatever host-side torch work is left runs on ONE thread -- no OpenMP team, so no barrier or wake-up that another tenant's
# load can stretch (the same tiny-shape CPU oracle calls took 1 s here and 131 s on a driver-style box with a 2-thread team).
_ON_GPU_BOX = os.path.exists("/dev/kfd")
_THREADS = 1 if _ON_GPU_BOX else _cpu_budget()
for _k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ[_k] = str(_THREADS)                    # before torch / numpy spin up their pools
os.environ.setdefault("TOKENIZERS_PARALLELISM", "false")

import pytest  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


_GPU_RUN = False


def oracle_device() -> str:
    """Where the GPU suite runs the oracle: torch-ROCm ("cuda") whenever there is a GPU -- the long verifications must not depend on
    the box's host cores (oracle/meshanything_oracle.py header) -- unless MA_ORACLE_DEVICE=cpu asks for the CPU."""
    import torch
    want = os.environ.get("MA_ORACLE_DEVICE", "")
    if want:
        return want
    return "cuda" if torch.cuda.is_available() else "cpu"


def pytest_configure(config):
    global _CONFIG, _GPU_RUN
    _CONFIG = config
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")
    import torch
    _GPU_RUN = torch.cuda.is_available()
    n = 1 if _GPU_RUN else _THREADS              # (see _ON_GPU_BOX above)
    torch.set_num_threads(n)
    try:
        torch.set_num_interop_threads(max(1, min(4, n)))
    except RuntimeError:
        pass


# Run order of the GPU suite: the ~240 kernel-level and tiny-shape tests first (seconds; they carry most
# SURVEY.md section-8 rows), then the fused-launch bitwise tests, then the 350M-shape pipeline tests of the benchmarked bf16 policy,
# then the fp32 ("exact") policy, the fidelity reports last.  A run that is cut short therefore still proves the cheap, broad part,
# and every finished test leaves a flushed "[t=...s] PASSED <nodeid>" line (terminal + gpurun_out/gpu_test_progress.log).
_ORDER = (
    ("test_gpu_kernels.py", ""),
    ("test_gpu_oracle_device.py", ""),
    ("test_gpu_pipeline.py", "tiny"),            # every test that takes the `tiny` fixture: test_*[fp32|bf16] without "full"
    ("test_gpu_model_api.py", ""),
    ("test_gpu_rccl.py", ""),
    ("test_gpu_pipeline.py", "test_large_batches"),
    ("test_gpu_pipeline.py", "test_weights_"),
    ("test_gpu_persist.py", ""),
    ("test_gpu_rows_attn.py", ""),
    ("test_gpu_rows_fused.py", ""),
    ("test_gpu_reference_anchor.py", ""),
    ("test_gpu_long_context.py", ""),
    ("test_gpu_pipeline.py", "[bf16]"),
    ("test_gpu_pipeline.py", "[fp16]"),
    ("test_gpu_pipeline.py", "test_v2_scale"),
    ("test_gpu_pipeline.py", "[fp32]"),
    ("test_gpu_fidelity.py", ""),
)


def _prio(item):
    mod = os.path.basename(str(item.fspath))
    name = item.name
    is_full = name.startswith("test_full_")
    for i, (m, key) in enumerate(_ORDER):
        if m != mod:
            continue
        if key == "":
            return i
        if key == "tiny":
            if not is_full and not name.startswith(("test_large_batches", "test_weights_", "test_v2_scale")):
                return i
            continue
        if key in ("[bf16]", "[fp16]", "[fp32]"):
            if is_full and name.endswith(key):
                return i
            continue
        if name.startswith(key):
            return i
    return len(_ORDER)


def pytest_collection_modifyitems(config, items):
    items.sort(key=_prio)                        # stable: parametrised module fixtures stay grouped inside a class


_T0 = time.time()
_PROGRESS = None


def _progress_line(config, line):
    global _PROGRESS
    tr = config.pluginmanager.get_plugin("terminalreporter")
    if tr is not None:
        tr.ensure_newline()
        tr.write_line(line)
        try:
            tr._tw.flush()
        except Exception:
            pass
    if _PROGRESS is None:
        try:
            d = os.path.join(REPO, "gpurun_out")
            os.makedirs(d, exist_ok=True)
            _PROGRESS = open(os.path.join(d, "gpu_test_progress.log"), "a", buffering=1)
            _PROGRESS.write(f"== pytest session pid {os.getpid()} started {time.strftime('%Y-%m-%d %H:%M:%S')}\n")
        except OSError:
            _PROGRESS = False
    if _PROGRESS:
        _PROGRESS.write(line + "\n")
        _PROGRESS.flush()


_CONFIG = None


def pytest_runtest_logreport(report):
    """One flushed line per finished test on the GPU box, so that a truncated tail names exactly what passed."""
    if _CONFIG is None or not _GPU_RUN:
        return
    if report.when == "call" or (report.when == "setup" and report.outcome != "passed"):
        _progress_line(_CONFIG, f"[t={time.time() - _T0:7.1f}s] {report.outcome.upper()} {report.nodeid} ({report.duration:.2f}s)")


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


_ARENA_CACHE = {}


def load_weights_cached(engine, cfg, **kw):
    """`engine.load_weights(cached_state_dict(cfg, **kw).items())`, but the packed arena of a layout is built once per session and
    copied device-to-device into later engines of the same layout (the arena layout is a function of the model dimensions and the
    precision policy only -- not of max_batch or n_max_faces): 600 M parameters are converted once, not once per test module."""
    import torch
    from meshanything_amd.checkpoint import state_dict_spec
    key = (cfg.dtype, cfg.enc_exact, tuple((k, v[0]) for k, v in state_dict_spec(cfg, False, False).items()), tuple(sorted(kw.items())))
    arena = engine.arena_tensor()
    have = _ARENA_CACHE.get(key)
    if have is not None and have.numel() == arena.numel():
        arena.copy_(have)
        torch.cuda.synchronize()
        engine.mark_weights_loaded()
        return
    engine.load_weights(cached_state_dict(cfg, **kw).items())
    torch.cuda.synchronize()
    _ARENA_CACHE[key] = engine.arena_tensor().clone()


@pytest.fixture(scope="session")
def state_dicts():
    return cached_state_dict


def mouse_variants(golden_dir, k):
    import numpy as np
    import torch
    """k distinct, non-degenerate clouds: pc_examples/mouse.npy rotated about z then y by fixed angles (normals rotate with the
    points), re-normalised like Dataset.  (Random-weight models collapse on the synthetic sphere clouds: every logit margin
    falls below the bf16 noise floor, which makes those rows useless as parity probes at full size.)"""
    from oracle.meshanything_oracle import normalize_pc
    base = np.load(os.path.join(golden_dir, "dataset.npz"))["mouse_norm"].astype(np.float32)
    out = []
    for i in range(k):
        az, ay = 0.7 * i, 0.4 * i
        rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]], dtype=np.float32)
        ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]], dtype=np.float32)
        r = ry @ rz
        pc = np.concatenate([base[:, :3] @ r.T, base[:, 3:] @ r.T], axis=1).astype(np.float32)
        pc[:, 3:] /= np.linalg.norm(pc[:, 3:], axis=1, keepdims=True)
        out.append(normalize_pc(pc) if i else np.load(os.path.join(golden_dir, "dataset.npz"))["mouse_norm"])
    return torch.from_numpy(np.stack(out))

import gc

_HEALTH = ("chain_fallbacks", "xchg_timeouts", "scalar_sweep_rescues")
TRANSIENT_FALLBACKS = []            # generations that fell back once and were run again (reported at the end of the session)


def _health_note(eng, what, before, after):
    return (f"{what}: the fused decode launches did not carry this generation (counters before {before}, after {after}, last exchange code "
            f"{eng.get_option('xchg_last_code')}, first give-up: code {eng.get_option('xchg_first_giveup_code')} block/wave word {eng.get_option('xchg_first_giveup_block'):#x} "
            f"after {eng.get_option('xchg_first_giveup_polls')} polls, sweeps that restarted their clock {eng.get_option('xchg_descheduled')}, chain_resident "
            f"{eng.get_option('chain_resident')})")


def fused_generate(eng, what, *args, **kw):
    """`eng.generate(*args, **kw)` whose answer must come from the fused decode launches: a generation whose fused launches time out
    re-runs on the five-launch chain and returns the same kind of answer, so a test that only looks at the answer can pass on the wrong kernels.  The
    engine's health counters -- generations that fell back, exchange sweeps that gave up, scalar sweeps that a vector look had to finish -- must not move
    across the generation whose result is returned, and the fused launches must still be armed after it.
    One lone give-up per few full-suite runs has been seen since round 5 (never in a soak of one engine; rounds 5 and 6 each one, on different launches):
    such a generation is run ONCE more with the fused launches re-armed, the event is printed, kept in TRANSIENT_FALLBACKS and listed in the session's
    summary; a second fall-back in a row fails the test.  Where the device cannot hold the fused grids (chain_resident == 0 from the start) nothing is
    fused and nothing is checked.  (The fall-back itself has its own tests: test_gpu_persist.py / test_gpu_rows_attn.py, *falls_back*.)"""
    for attempt in (0, 1):
        armed = eng.get_option("chain_resident") == 1
        before = {k: eng.get_option(k) for k in _HEALTH}
        out = eng.generate(*args, **kw)
        if not armed:
            return out
        after = {k: eng.get_option(k) for k in _HEALTH}
        if after == before and eng.get_option("chain_resident") == 1:
            return out
        note = _health_note(eng, what, before, after)
        if attempt == 0:
            TRANSIENT_FALLBACKS.append(note)
            print(f"\n[fused path] {note} -- re-armed, running this generation once more", flush=True)
            del out
            eng.set_option("chain_resident", 1)
            continue
        raise AssertionError(note + ": twice in a row; what was verified is the fall-back chain")


def pytest_terminal_summary(terminalreporter):
    if TRANSIENT_FALLBACKS:
        terminalreporter.section("fused decode launches: generations that fell back once and were re-run")
        for n in TRANSIENT_FALLBACKS:
            terminalreporter.line(n)


def generate_on_a_starved_device(eng, gen, cus=224, attempts=5):
    """`gen()` (a generation on the fused decode launches) while `cus` CUs are held by a second stream for up to 2 s: returns (result, seconds, warnings
    caught).  The engine must notice the starved grid (one bounded sweep), fall back and say so with a RuntimeWarning.
    The second stream has to sit on a hardware queue of its own for that: HIP multiplexes a process's streams onto a few hardware queues (four by default),
    in order of creation, and a stream of the engine (its main one; since round 6 also the prefill's tail-chain stream) that shares a queue with the hog's
    simply waits behind it -- the generation then takes the hog's 2 s and meets an idle device: no time-out, nothing to fall back from (seen in full-suite
    runs, where dozens of streams have come and gone).  So: a fresh side stream per attempt, all kept alive so that the next one lands on the next queue."""
    import time, warnings
    import torch
    kept = []
    for k in range(attempts):
        side = torch.cuda.Stream()
        kept.append(side)
        release = torch.zeros(1, dtype=torch.int32).pin_memory()
        torch.cuda.synchronize()
        t0 = time.time()
        eng.occupy_cus(cus, 2_000_000, stream=side, release=release)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            try:
                out = gen()
            finally:
                release[0] = 1
        dt = time.time() - t0
        side.synchronize()
        torch.cuda.synchronize()
        if any("fused decode launches timed out" in str(w.message) for w in caught):
            return out, dt, caught
        print(f"[starved device] attempt {k}: no time-out in {dt:.2f} s -- an engine stream shared the hog's hardware queue and waited behind it; another side stream", flush=True)
        eng.set_option("chain_resident", 1)
    pytest.fail(f"the fused launches never timed out under a {cus}-CU hog in {attempts} attempts")


@pytest.fixture(autouse=True)
def _collect_before_gpu_tests(request):
    """Engines (and torch tensors) that are only reachable through reference cycles are destroyed when the collector gets round to it: ma_engine_destroy
    is a string of hipFree calls -- device-wide synchronisations -- and one landing in the middle of a test that holds most of the device with a
    second stream (the *falls_back* tests) stalls the launches under test until that stream lets go.  (One of the two explanations considered for a stalled
    fall-back test in round 6; the other -- an engine stream sharing the hog's hardware queue -- was the one confirmed, generate_on_a_starved_device.  Collecting
    between tests costs nothing and keeps engine destruction out of the tests' timed regions.)"""
    if request.node.get_closest_marker("gpu") is not None:
        gc.collect()
    yield
