#!/usr/bin/env python3
"""Benchmark of the MeshAnything hot path on MI355X.

A "step" = one pass of the hot path over one batch: point cloud -> encode -> prefill -> 7202 decode steps (800-face cap,
eos suppressed: random-init weights never emit a natural eos pattern) -> detokenize.  At N=1 the workload is
BASELINE.json configs[1] (single shape pc_examples/mouse.npy, 350M shape, bf16, greedy, KV-cache decode); at N>1 every
rank runs the same per-GPU work on its own shape (weak scaling, weights broadcast once over RCCL, no per-step collective).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `value` = face-tokens/s over all ranks (generated tokens incl. the dropped first and the
last slot) with the input cloud already resident in HBM.
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL / tensor sharing fail with the legacy mode); keep a caller's value
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from meshanything_amd.config import MAConfig, DTYPE_BF16, DTYPE_F16, DTYPE_F32  # noqa: E402
from meshanything_amd.checkpoint import synthetic_state_dict          # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s measured achievable
# health counters of the fused decode launches (ma_engine_get_option): generations that fell back to the launch chain, sweeps that gave up, the error word of
# the last fall-back, scalar sweeps a vector look had to finish (rows_attn.hpp), long-lived blocks
HEALTH_KEYS = ("chain_resident", "chain_fallbacks", "xchg_timeouts", "xchg_last_code", "scalar_sweep_rescues", "slow_blocks", "slow_block_max_us", "xchg_descheduled")


def synth_cloud(seed: int, n: int) -> np.ndarray:
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(n, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    r = 0.3 + 0.7 * torch.rand(n, 1, generator=g)
    return torch.cat([d * r, d], dim=-1).numpy().astype(np.float32)


def normalize_pc(pc_normal: np.ndarray) -> np.ndarray:
    """Dataset.__getitem__ normalisation (main.py:45-58) -- product-side copy lives in meshanything_amd/data.py."""
    from meshanything_amd.data import normalize_pc as f
    return f(pc_normal)


def gemv_bytes_per_step(cfg: MAConfig, esz: int):
    H, F, V = cfg.hidden, cfg.ffn, cfg.vocab
    elems = cfg.layers * (3 * H * H + H * H + F * H + H * F) + V * H + H * cfg.codebook_dim
    launches = cfg.layers * 4 + 2
    return elems * esz, launches


def kv_bytes_per_step(cfg: MAConfig, length: int, esz: int) -> int:
    return cfg.layers * 2 * cfg.hidden * esz * length


# algorithmic GFLOP per shape of the dense phases: encoder 108.5 + prefix projections 0.8, prefill 158.5, detokenizer 115.6
DENSE_GFLOP = {"encode_prefix": 109.3, "prefill": 158.5, "detokenize": 115.6}
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TFLOPS = 157.3  # f32-input MFMA = the fp32 vector rate (MI355X_MICROARCH.md)


def dense_phase_table(eng, cfg: MAConfig, batches=(16, 64), iters: int = 3):
    """Encoder + prefix, prefill (generate with one new token) and detokenizer for a batch of synthetic clouds: ms, TFLOP/s and
    fraction of the dense bf16 MFMA peak (time-derived; the PMC MFMA-busy view of the same phases is under profiles/)."""
    out = []
    g = torch.Generator().manual_seed(0)
    for B in batches:
        if B > cfg.max_batch:
            continue
        d = torch.randn(B, cfg.n_points, 3, generator=g)
        d = d / d.norm(dim=-1, keepdim=True)
        x = torch.cat([d * 0.9, d], dim=-1).half().cuda()
        ids = torch.randint(0, cfg.codebook_size, (B, cfg.n_max_faces * 9), generator=g).cuda()
        acc = {k: [] for k in DENSE_GFLOP}
        for it in range(iters + 1):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
            lat, prefix = eng.encode(x)
            ev[1].record()
            eng.generate(prefix, max_new_tokens=1, suppress_eos=True)
            ev[2].record()
            eng.detokenize(ids, lat)
            ev[3].record()
            torch.cuda.synchronize()
            if it:
                for i, k in enumerate(DENSE_GFLOP):
                    acc[k].append(ev[i].elapsed_time(ev[i + 1]))
        ms = {k: float(np.median(v)) for k, v in acc.items()}
        # the point encoder runs on the fp32 matrix path under cfg.enc_exact (1e-5 on its activations): it is rated against the fp32 MFMA
        # peak, the 16-bit phases (prefill, detokenizer) against the bf16 one
        exact_enc = bool(cfg.enc_exact)
        k16 = [k for k in ms if not (exact_enc and k == "encode_prefix")]
        tot_ms = sum(ms[k] for k in k16)
        tot_gf = sum(DENSE_GFLOP[k] for k in k16) * B
        row = {"batch": B, "ms": {k: round(v, 3) for k, v in ms.items()}, "TFLOPs": {k: round(DENSE_GFLOP[k] * B / ms[k], 1) for k in ms},
               "phases_16bit": k16, "all_ms": round(tot_ms, 3), "all_TFLOPs": round(tot_gf / tot_ms, 1), "frac_of_bf16_mfma_peak": round(tot_gf / tot_ms / MFMA_PEAK_TFLOPS, 4)}
        # ... and every phase together (encoder included, whatever its precision) against the bf16 peak: the like-for-like figure across rounds
        # (rounds 1-3 ran a 16-bit encoder)
        all_ms, all_gf = sum(ms.values()), sum(DENSE_GFLOP.values()) * B
        row["all_phases_incl_encoder"] = {"ms": round(all_ms, 3), "TFLOPs": round(all_gf / all_ms, 1), "frac_of_bf16_mfma_peak": round(all_gf / all_ms / MFMA_PEAK_TFLOPS, 4)}
        row["enc_exact"] = int(exact_enc)
        if exact_enc:
            row["encoder_fp32_frac_of_fp32_mfma_peak"] = round(DENSE_GFLOP["encode_prefix"] * B / ms["encode_prefix"] / FP32_MFMA_PEAK_TFLOPS, 4)
        out.append(row)
    return out


def measured_peaks(eng):
    """What THIS box reaches on the two roofline denominators, next to the vendor numbers the fractions are
    quoted against: a 16-byte-per-lane streaming copy of 2 x 1 GiB (ma_op_stream_copy, the fastest of its three forms; read + write bytes) and the library's dense bf16
    GEMM on 8192^3 uniform random [-1, 1) operands (ma_op_gemm_bf16)."""
    import ctypes as C
    lib = eng.lib
    cur = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
    b = torch.empty(n, dtype=torch.uint8, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(reps):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / reps * 1e-3
    t_copy = min(timed(lambda: lib.ma_op_stream_copy(C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), C.c_size_t(n), mode, cur), 10) for mode in (0, 1, 2))
    del a, b
    M = 8192
    A = (torch.rand(M, M, device="cuda") * 2 - 1).to(torch.bfloat16)
    W = (torch.rand(M, M, device="cuda") * 2 - 1).to(torch.bfloat16)
    Cb = torch.empty(M, M, dtype=torch.bfloat16, device="cuda")
    z = C.c_void_p(0)
    t_gemm = timed(lambda: lib.ma_op_gemm_bf16(C.c_void_p(A.data_ptr()), M, C.c_void_p(W.data_ptr()), z, z, 0, z, 0, C.c_void_p(Cb.data_ptr()), M, M, M, M, 0, cur), 5)
    del A, W, Cb
    torch.cuda.empty_cache()
    return {"stream_copy_GBps": round(2 * n / t_copy / 1e9, 1), "stream_copy_frac_of_8TBps": round(2 * n / t_copy / 1e9 / HBM_PEAK_GBS, 4),
            "gemm_bf16_8192_TFLOPs": round(2 * M ** 3 / t_gemm / 1e12, 1), "gemm_bf16_frac_of_2500TF": round(2 * M ** 3 / t_gemm / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "note": "measured by this run on this box (HIP events); the roofline fractions in this line stay quoted against the vendor peaks 8000 GB/s / 2500 TFLOP/s"}


def plan(gpus: int, batch: int, rank: int, world: int, faces: int = 800, dtype: str = "bf16", sampling: bool = False, tokens_per_shape: int = 7202):
    """What this rank runs (pure: no GPU, no process group -- tests/test_bench_plan.py checks it for world sizes 1..8).
    BASELINE.json's metric is "batch=1 and batch=8xN shapes": `--gpus 1` decodes ONE shape (configs[1]); `--gpus N > 1` gives every
    rank 8 shapes that step together (weak scaling: per-GPU work fixed), global shape index g = rank * batch + j, so the N ranks
    cover shapes 0 .. 8N-1 exactly once (reference: accelerate's batch-sampler sharding, main.py:137-146).  Shape 0 is
    pc_examples/mouse.npy after Dataset normalisation, every other one a seeded synthetic cloud."""
    if world != gpus:
        raise SystemExit(f"--gpus {gpus} but WORLD_SIZE is {world}: launch with torch.distributed.run --nproc-per-node {gpus}")
    if not 0 <= rank < world:
        raise SystemExit(f"RANK {rank} outside [0, {world})")
    if batch <= 0:
        batch = 1 if gpus == 1 else 8
    shapes = [rank * batch + j for j in range(batch)]
    if world > 1 and batch == 8:
        head = f"BASELINE.json metric 'batch=8xN shapes': batch={batch}x{world} ({world * batch} shapes over {world} GPUs, the layout of configs[3] at 8 per GPU)"
    elif world > 1 and faces == 800 and sampling:
        # configs[3] itself is `--gpus 8 --batch 64 --sampling`: 512 clouds, rank r owns the contiguous block 64 r .. 64 r + 63 (what
        # accelerate's BatchSamplerShard hands process r of 8 at batch size 64: whole batches, round-robin -- main.py:137-146)
        head = f"BASELINE.json configs[3]: batch {world * batch} sharded data-parallel over {world} GPUs ({batch} per GPU)" + ("" if (world, batch) == (8, 64) else " [configs[3] proper is 64 x 8]")
    else:
        head = (f"BASELINE.json configs[{1 if batch == 1 else (2 if faces == 800 else 4)}]"
                + (" (per-GPU share of configs[3] when launched on 8 GPUs)" if batch > 1 and faces == 800 else ""))
    what = ("single shape pc_examples/mouse.npy (Dataset-normalised, seed 0)" if batch == 1
            else f"batch {batch} per GPU (mouse.npy + seeded synthetic 4096-pt clouds)")
    workload = (f"{head}: {what}, 350M shape, {dtype}, 1xMI355X per rank, {'top-k 50 / top-p 0.95 sampling' if sampling else 'greedy'}, "
                f"{faces}-face cap ({tokens_per_shape} tokens/shape, eos suppressed), KV-cache decode, hipGraph")
    # sampling: the in-kernel uniform stream is keyed by (seed, row, step) -- the seed differs per rank, so no two of the world x batch rows
    # of a job draw from the same stream
    return {"batch": batch, "shapes": shapes, "global_batch": world * batch, "workload": workload, "seed": 1234 + 1000003 * rank,
            "parallelism": f"dp{world} (independent shapes, weights broadcast once)", "scaling": "weak"}


def cpu_baseline(cfg: MAConfig, sd, x: torch.Tensor, decode_steps: int = 384, budget_s: float = 20.0):
    """The oracle (a CPU port of the reference arithmetic, fp32, PyTorch threads = host cores) on a bounded sample of the
    same workload: encode + prefill + up to `decode_steps` greedy KV-cache steps (at most `budget_s` seconds of them) for the same
    cloud and weights."""
    from oracle.meshanything_oracle import Oracle
    # more threads than ~16 make PyTorch's batch-1 GEMVs slower (256-core box: 15 s/step with 256 threads), so the
    # baseline uses 16 threads and says so in `cores`
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    o = Oracle(cfg, sd, "fp32")
    t0 = time.time()
    lat = o.encode_latents(x)
    prefix = o.process_point_feature(lat)
    t_enc = time.time() - t0
    t0 = time.time()
    cache = [None] * cfg.layers
    h = o.opt_layers(o.embed_prefix(prefix), cache)
    tok = int(torch.argmax(o.lm_head(h[0, -1])))
    t_prefill = time.time() - t0
    t0 = time.time()
    done = 0
    for n in range(1, decode_steps + 1):                    # bounded twice: `decode_steps` steps or `budget_s` seconds, whichever comes first
        e = o.embed_tokens(torch.tensor([tok]), torch.tensor([n]))
        h = o.opt_layers(e[None], cache)
        lg = o.lm_head(h[0, -1])
        lg[1] = float("-inf")
        tok = int(torch.argmax(lg))
        done = n
        if n >= 16 and time.time() - t0 > budget_s:         # a host shared with other tenants can be 10x slower than an idle one
            break
    decode_steps = done
    t_dec = time.time() - t0
    tps = decode_steps / t_dec
    est_mesh = t_enc + t_prefill + cfg.max_new_tokens / tps
    return {"value": round(tps, 2), "unit": "face-tokens/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 (torch CPU, {cores} threads): encode {t_enc:.2f}s + prefill {t_prefill:.2f}s + {decode_steps} greedy "
                      f"KV-cache decode steps at context {cfg.cond_length}..{cfg.cond_length + decode_steps} ({t_dec:.2f}s); "
                      f"extrapolated >= {est_mesh:.0f} s/mesh at 800 faces (context grows to {cfg.max_seq}); the FULL 7202-token run of the same oracle on a gpurun "
                      f"box's host (16 threads) is committed as profiles/r05_cpu_baseline_full.json: 722.7 s/mesh = 9.96 face-tokens/s, step 49.7 / 65.2 / 303 ms at "
                      f"context 300 / 3800 / 7400 (the per-step torch.cat of the cache, as transformers 4.39.3 does it, dominates late)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--faces", type=int, default=800)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=4)
    ap.add_argument("--batch", type=int, default=0, help="shapes per GPU decoded together; default: 1 at --gpus 1 (BASELINE.json configs[1]), "
                                                         "8 at --gpus N > 1 (the metric's 'batch=8xN shapes'); configs 3 / 5 use 64 / 8")
    ap.add_argument("--sampling", action="store_true", help="top-k 50 / top-p 0.95 sampling (configs 3-4) instead of greedy")
    ap.add_argument("--no-batched-table", action="store_true", help="skip the batch 8 / 64 decode-step table of the default run")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    pl = plan(args.gpus, args.batch, rank, world, args.faces, args.dtype, args.sampling, args.faces * 9 + 2)
    args.batch = pl["batch"]                                # BASELINE.json metric: batch 1, and batch 8 x N shapes
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ          # under torchrun the RCCL path runs even with one rank
    real_stdout = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # RCCL prints its version banner with C stdio on stdout at communicator init; stdout must carry exactly ONE JSON line,
        # so fd 1 points at stderr for the rest of the run and the JSON line is written to the saved descriptor
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    from meshanything_amd.engine import Engine
    from meshanything_amd import dp
    cfg = MAConfig.full(dtype={"bf16": DTYPE_BF16, "fp16": DTYPE_F16, "fp32": DTYPE_F32}[args.dtype], n_max_faces=args.faces, max_batch=args.batch)
    eng = Engine(cfg, local_rank)
    sd = {}
    t_load = time.time()

    def items():                                            # only called on rank 0
        # random-init weights in the reference key layout (no network).  init="diverse": the default init's greedy stream is ONE token
        # for ever (checkpoint.py); this one depends on its own tokens and positions -- `tokens_distinct` in the JSON line shows it
        sd.update(synthetic_state_dict(cfg, init="diverse"))
        return sd.items()
    # weights travel once, rank 0 -> all, as ONE RCCL broadcast of the packed arena over xGMI
    dp.load_weights_dp(eng, items, rank, world, force_broadcast=use_dist)
    t_load = time.time() - t_load

    # shapes of this rank: global shape index g = rank * batch + j; shape 0 is pc_examples/mouse.npy after Dataset
    # normalisation (seed 0), every other one a seeded synthetic cloud
    rows = []
    for gidx in pl["shapes"]:
        if gidx == 0:
            rows.append(np.load(os.path.join(REPO, "tests", "golden", "dataset.npz"))["mouse_norm"])
        else:
            rows.append(normalize_pc(synth_cloud(gidx, cfg.n_points)))
    pc = np.stack(rows)
    x = torch.from_numpy(pc).cuda()

    def step():
        return eng.forward(x, suppress_eos=True, sampling=args.sampling, seed=pl["seed"])

    for _ in range(args.warmup):
        out = step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tokens_per_step = cfg.max_new_tokens * args.batch
    assert tuple(out["tokens"].shape) == (args.batch, cfg.max_new_tokens)
    tokens_distinct = [len(set(r.tolist())) for r in out["tokens"].cpu()]
    # health of the fused decode launches over everything this engine ran so far (a generation that lost them was re-run on the
    # five-launch chain: correct, slower, and visible here)
    health = {k: eng.get_option(k) for k in HEALTH_KEYS}
    # encoder activations of shape 0 (mouse.npy) against the reference's own perceiver (tests/golden/full.npz; the encoder weights of
    # init="diverse" are the default ones): the north star's 1e-5
    enc_err = None
    if rank == 0 and pl["shapes"][0] == 0:
        gold = np.load(os.path.join(REPO, "tests", "golden", "full.npz"))
        lat0 = out["latents"][0].cpu().numpy()
        enc_err = float(max(np.abs(lat0[gold["full_rows"]] - gold["full_latents_rows"]).max(), np.abs(lat0[:, :8] - gold["full_latents_cols8"]).max()))

    if rank == 0:
        esz = 4 if args.dtype == "fp32" else 2
        # ---- per-phase times (one extra pass, HIP events on the current stream) ----
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        lat, prefix = eng.encode(x)
        ev[1].record()
        toks, _ = eng.generate(prefix, suppress_eos=True, sampling=args.sampling, seed=pl["seed"])
        ev[2].record()
        ids = eng.postprocess_tokens(toks)
        coords = eng.detokenize(ids, lat)
        ev[3].record()
        torch.cuda.synchronize()
        phases = {"encode_ms": ev[0].elapsed_time(ev[1]), "generate_ms": ev[1].elapsed_time(ev[2]), "detokenize_ms": ev[2].elapsed_time(ev[3])}
        # ---- roofline of the dominant kernel: the weight-streaming GEMV.  HIP events (on the stream the kernels run on)
        # around `profile_steps` steps with ONLY the GEMV launches enqueued: elapsed / launches = average launch duration,
        # boundary to the next launch included -- the number a rocprofv3 kernel trace of the same command reports
        # (profiles/).  Algorithmic bytes per launch = weight bytes of one step / GEMV launches of one step.
        mid = cfg.cond_length + cfg.max_new_tokens // 2
        eng.set_option("profile_batch", args.batch)
        prof = eng.profile_decode(mid, args.profile_steps)
        wbytes, _ = gemv_bytes_per_step(cfg, esz)
        kvbytes = args.batch * kv_bytes_per_step(cfg, mid, esz)
        mfma_path = args.batch >= 4 and args.dtype != "fp32"
        fused_qkv = bool(eng.get_option("fuse_qkv_attn")) and not mfma_path
        fused_o1 = bool(eng.get_option("fuse_oproj_fc1")) and not mfma_path
        fused_f2 = fused_o1 and bool(eng.get_option("fuse_fc2"))
        qkv_w = cfg.layers * 3 * cfg.hidden * cfg.hidden * esz
        # 8 rows on the matrix-core path: the two-launch layer -- rows_attn_kernel streams q/k/v + out_proj weights AND the cache,
        # rows_mlp_kernel fc1 + fc2 (csrc/rows_attn.hpp, rows_mlp.hpp); both counted in the class their launch is timed in
        rows_attn = mfma_path and args.batch == 8 and bool(eng.get_option("fuse_rows_attn")) and bool(eng.get_option("chain_resident"))
        rows_mlp = mfma_path and args.batch == 8 and bool(eng.get_option("fuse_rows_mlp")) and bool(eng.get_option("chain_resident"))
        attn_w = cfg.layers * 4 * cfg.hidden * cfg.hidden * esz if rows_attn else (qkv_w if fused_qkv else 0)      # matrices streamed by the cache-class launches
        # the launches of a decode step fall in two classes (ma_profile_decode times each class alone, HIP events on the launch
        # stream around `profile_steps` steps; elapsed / launches = average launch duration, boundary to the next launch included --
        # what a rocprofv3 kernel trace of the same command reports, profiles/):
        #   weights : every launch that streams weight matrices only (gemv_kernel, oproj_fc1_kernel; batched: gemm_dec + prologues)
        #   cache   : the launches that stream the KV cache (attn_decode_kernel, or qkv_attn_kernel = q/k/v weights + cache)
        classes = {}
        for key, name, byts in (("gemv", "weights", wbytes - attn_w), ("attn_decode", "cache", kvbytes + attn_w)):
            n_l = prof["launches"][key]
            per_step = n_l // args.profile_steps
            ms = prof["ms"][key]
            classes[name] = {"launches_per_step": per_step, "launches_timed": n_l, "avg_launch_us": round(ms / max(1, n_l) * 1e3, 3),
                             "bytes_per_launch": int(byts / max(1, per_step)), "GBps": round(byts / max(1, per_step) / (ms / max(1, n_l) * 1e-3) / 1e9, 1),
                             "us_per_step": round(ms / args.profile_steps * 1e3, 1)}
        kern = {"weights": ("rows_mlp_kernel (LayerNorm 1 + fc1 + fc2 + LayerNorm 2, 8 rows) + gemv / gemm_dec (embed, lm_head)" if rows_mlp else
                            "gemm_dec_kernel + rows_prologue_kernel (batched decode weight stream)" if mfma_path else
                            (("oproj_fc1_kernel (out_proj + LN + fc1 + fc2) + gemv_kernel (embed, lm_head)" if fused_f2 else "oproj_fc1_kernel + gemv_kernel (fc2, embed, lm_head)") if fused_o1 else "gemv_kernel") + " -- the launches that stream weight matrices only"),
                "cache": ("rows_attn_kernel (8 rows: q/k/v projection + two-block attention + out_proj: their weights + the KV cache)" if rows_attn else
                          "qkv_attn_kernel (q/k/v projection + split-KV attention: q/k/v weights + the KV cache)" if fused_qkv else "attn_decode_kernel (KV cache)")}
        dom = max(classes, key=lambda k: classes[k]["us_per_step"])          # the class the step spends most of its time in
        dc = classes[dom]
        step_ms = prof["step_ms_graph"] or prof["step_ms_eager"]
        step_bytes = wbytes + kvbytes
        # HBM bytes per launch of the dominant class from the PMC counters (scripts/gpu_r6.sh, stage pmc: separate rocprofv3 --pmc passes,
        # corrected as the MI355X guide prescribes; committed under profiles/): None when no such file is present
        # (a --pmc pass cannot run inside this process: the figure is IMPORTED from the committed profile of the same kernels and labelled so)
        traffic, traffic_source = None, None
        for name in ("r06_pmc_decode_traffic.json", "r05_pmc_decode_traffic.json", "r04_pmc_decode_traffic.json", "r03_pmc_decode_traffic.json", "r02_pmc_decode_traffic.json"):
            f = os.path.join(REPO, "profiles", name)
            if traffic is None and os.path.exists(f) and args.batch == 1 and args.dtype == "bf16" and args.faces == 800:
                traffic = json.load(open(f)).get("hbm_bytes_per_launch", {}).get(dom)
                if traffic is not None:
                    traffic_source = f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; imported, NOT measured by this run)"
        # 8 rows: the PMC pass of the two fused launches at THIS cache length (profile_decode steps only, scripts/archive/gpu_pmc_b8_mid.sh)
        f8 = os.path.join(REPO, "profiles", "r05_pmc_decode_traffic_b8_kv3858.json")
        if traffic is None and rows_attn and rows_mlp and args.dtype == "bf16" and mid == 3858 and os.path.exists(f8):
            traffic = json.load(open(f8)).get("hbm_bytes_per_launch", {}).get(dom)
            if traffic is not None:
                traffic_source = "profiles/r05_pmc_decode_traffic_b8_kv3858.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over decode steps at this cache length; imported, NOT measured by this run)"
        roofline = {"bound": "hbm", "kernel": f"{kern[dom]}, {dc['launches_per_step']} launches per step", "achieved": dc["GBps"],
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(dc["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "bytes_per_launch": dc["bytes_per_launch"], "avg_launch_us": dc["avg_launch_us"], "launches_timed": dc["launches_timed"],
                    "kv_len": mid, "classes": {k: dict(v, kernel=kern[k]) for k, v in classes.items()},
                    "decode_step_ms_graph": round(prof["step_ms_graph"], 4), "decode_step_ms_eager": round(prof["step_ms_eager"], 4),
                    "decode_step_GBps_at_mid_context": round(step_bytes / (step_ms * 1e-3) / 1e9, 1),
                    "decode_step_frac_of_peak": round(step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        health_after = {k: eng.get_option(k) for k in HEALTH_KEYS}
        peaks = measured_peaks(eng)
        cpu = None
        if not args.no_cpu_baseline and world == 1:          # the CPU leg is reported at N=1 only
            cpu = cpu_baseline(cfg, sd, torch.from_numpy(pc[:1]))
        batched = None
        batch8_generation = None
        dense = None
        fp32_exact = None
        fp16_policy = None
        if args.batch == 1 and not args.no_batched_table and args.dtype == "bf16" and world == 1:
            # configs 3-5 in brief: the decode step when 8 / 64 shapes share the weight stream (mid context, graph replay)
            eng.close()
            torch.cuda.empty_cache()
            cfg_b = MAConfig.full(dtype=DTYPE_BF16, n_max_faces=args.faces, max_batch=64)
            eng_b = Engine(cfg_b, local_rank)
            eng_b.load_weights(sd.items())
            batched = []
            for Bb in (8, 64):
                eng_b.set_option("profile_batch", Bb)
                pb = eng_b.profile_decode(mid, 2)
                sb = pb["step_ms_graph"]
                byts = wbytes + Bb * kv_bytes_per_step(cfg_b, mid, 2)
                batched.append({"batch": Bb, "kv_len": mid, "decode_step_ms": round(sb, 4), "face_tokens_per_s": round(Bb / (sb * 1e-3), 1),
                                "GBps": round(byts / (sb * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(byts / (sb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
            # ... and ONE whole generation of 8 shapes stepping together (the per-GPU share of the metric's "batch = 8 x N"): 7202 steps, cache to the end
            x8 = torch.from_numpy(np.stack([pc[0]] + [normalize_pc(synth_cloud(j, cfg_b.n_points)) for j in range(1, 8)])).cuda()
            _, prefix8 = eng_b.encode(x8)
            eng_b.generate(prefix8, suppress_eos=True, max_new_tokens=64)          # warm: graph capture for 8 rows
            h8 = {k: eng_b.get_option(k) for k in HEALTH_KEYS}
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            t8, _ = eng_b.generate(prefix8, suppress_eos=True)
            torch.cuda.synchronize()
            t_8 = time.perf_counter() - t1
            assert tuple(t8.shape) == (8, cfg_b.max_new_tokens)
            batch8_generation = {"batch": 8, "tokens": 8 * cfg_b.max_new_tokens, "seconds": round(t_8, 3), "face_tokens_per_s": round(8 * cfg_b.max_new_tokens / t_8, 1),
                                 "sec_per_mesh": round(t_8 / 8, 4), "ms_per_step": round(t_8 / cfg_b.max_new_tokens * 1e3, 4),
                                 "fused_rows_attn": int(eng_b.get_option("fuse_rows_attn")), "fused_rows_mlp": int(eng_b.get_option("fuse_rows_mlp")),
                                 "fused_launch_health_before": h8, "fused_launch_health_after": {k: eng_b.get_option(k) for k in HEALTH_KEYS},
                                 "note": "ma_generate only (prefill + 7202 decode steps of 8 rows, eos suppressed), one warm generation, wall clock around the call"}
            dense = dense_phase_table(eng_b, cfg_b)
            eng_b.close()
            torch.cuda.empty_cache()
            # the parity ("exact") mode's throughput: the same mesh with fp32 weights, fp32 KV cache and no activation rounding -- the mode the
            # token-identity / 1e-5 gates of tests/ run in (round 6: on the same two fused launches per layer as the 16-bit policies, with the
            # five-launch chain's fp32 summation order kept bit for bit)
            cfg_x = MAConfig.full(dtype=DTYPE_F32, n_max_faces=args.faces, max_batch=1)
            eng_x = Engine(cfg_x, local_rank)
            eng_x.load_weights(sd.items())
            eng_x.forward(x[:1], suppress_eos=True, max_new_tokens=64)          # warm: graph capture, clocks
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ox = eng_x.forward(x[:1], suppress_eos=True)
            torch.cuda.synchronize()
            t_x = time.perf_counter() - t1
            assert tuple(ox["tokens"].shape) == (1, cfg_x.max_new_tokens)
            fp32_exact = {"face_tokens_per_s": round(cfg_x.max_new_tokens / t_x, 1), "sec_per_mesh": round(t_x, 3), "meshes_timed": 1,
                          "fused_qkv_attn": int(eng_x.get_option("fuse_qkv_attn")), "fused_oproj_fc1": int(eng_x.get_option("fuse_oproj_fc1")),
                          "chain_fallbacks": int(eng_x.get_option("chain_fallbacks")),
                          "note": "MA_DTYPE_F32 policy, same cloud and weights, one warm mesh; streams 2x the bytes of the bf16 policy"}
            eng_x.close()
            # the reference's own arithmetic class (fp16 autocast, main.py:114-118,149): the same mesh under MA_DTYPE_F16 -- same kernels, IEEE half
            # instead of bf16 in weights / KV / GEMM inputs
            cfg_h = MAConfig.full(dtype=DTYPE_F16, n_max_faces=args.faces, max_batch=1)
            eng_h = Engine(cfg_h, local_rank)
            eng_h.load_weights(sd.items())
            eng_h.forward(x[:1], suppress_eos=True, max_new_tokens=64)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            oh = eng_h.forward(x[:1], suppress_eos=True)
            torch.cuda.synchronize()
            t_h = time.perf_counter() - t1
            fp16_policy = {"face_tokens_per_s": round(cfg_h.max_new_tokens / t_h, 1), "sec_per_mesh": round(t_h, 3), "meshes_timed": 1,
                           "tokens_distinct": len(set(oh["tokens"][0].tolist())),
                           "tokens_equal_to_the_bf16_run": int((oh["tokens"][0].cpu() == out["tokens"][0].cpu()).sum()),
                           "note": "MA_DTYPE_F16 policy (the reference's fp16-autocast precision class), same cloud and weights, one warm mesh"}
            eng_h.close()
        total_tokens = world * args.steps * tokens_per_step
        res = {
            "metric": f"face-tokens/sec ({args.faces}-face cap, batch {args.batch} per GPU) + sec/mesh", "value": round(total_tokens / dt, 2), "unit": "face-tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": pl["workload"],
                       "global_batch": pl["global_batch"], "tokens_per_mesh": cfg.max_new_tokens, "parallelism": pl["parallelism"],
                       "weights": "seeded random init in the reference key layout (no checkpoint available offline), init=diverse: checkpoint.py"},
            "sec_per_mesh": round(dt / args.steps / args.batch, 4), "batched_decode_steps": batched, "batch8_generation": batch8_generation, "dense_phases": dense, "phases_ms": {k: round(v, 3) for k, v in phases.items()},
            "weights_load_s": round(t_load, 2), "fp32_exact": fp32_exact, "fp16_policy": fp16_policy, "roofline": roofline, "cpu_baseline": cpu,
            "tokens_distinct": tokens_distinct, "encoder_max_abs_err": enc_err, "fused_launch_health": {"timed_region": health, "after_profiling": health_after},
            "measured_peaks": peaks,
        }
        if real_stdout is not None:
            os.write(real_stdout, (json.dumps(res) + "\n").encode())
        else:
            print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
