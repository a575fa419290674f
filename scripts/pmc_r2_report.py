#!/usr/bin/env python3
"""Turn the per-kernel PMC means of scripts/archive/gpu_pmc_r2.sh (round 6: scripts/gpu_r6.sh, stage pmc) into the two round-2 summaries:
 r02_pmc_decode_traffic.json : HBM bytes per launch of the two decode launch classes (FETCH_SIZE doubled: gfx950 tallies the 128-byte
                               requests of a wide coalesced stream at 64 bytes, MI355X_MICROARCH.md, HBM; + WRITE_SIZE; units KiB)
 r02_pmc_dense_mfma.json     : matrix-core busy fraction of the dense-phase kernels = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) /
                               (GRBM_GUI_ACTIVE / 8 x 1024): rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (a 190 us GEMM reads 3.7 M
                               cycles = 8 x 466 K at 2.4 GHz), so the per-dispatch active time is a eighth of it
usage: pmc_r2_report.py decode_raw.json dense_raw.json outdir"""
import json, os, sys
dec, den, out = json.load(open(sys.argv[1])), json.load(open(sys.argv[2])), sys.argv[3]
RND = sys.argv[4] if len(sys.argv) > 4 else "r02"

def cls(name):
    if "qkv_attn_kernel" in name or "attn_decode_kernel" in name or "rows_attn_kernel" in name or "attn_decode_final_kernel" in name:
        return "cache"
    if "oproj_fc1_kernel" in name or "gemv_kernel" in name or "rows_mlp_kernel" in name or "gemm_dec_kernel" in name or "gemm_dec_ln_kernel" in name or "rows_prologue_kernel" in name:
        return "weights"
    return None

acc = {"weights": [0.0, 0], "cache": [0.0, 0]}
per_kernel = {}
for k, cs in dec.items():
    c = cls(k)
    if not c or "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
        continue
    n = cs["FETCH_SIZE"]["dispatches"]
    byts = (2.0 * cs["FETCH_SIZE"]["mean"] + cs["WRITE_SIZE"]["mean"]) * 1024.0
    acc[c][0] += byts * n; acc[c][1] += n
    per_kernel[k] = {"class": c, "dispatches": n, "FETCH_SIZE_KiB_raw": round(cs["FETCH_SIZE"]["mean"], 1), "WRITE_SIZE_KiB": round(cs["WRITE_SIZE"]["mean"], 1),
                     "hbm_bytes_per_launch": int(byts)}
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on scripts/prof_step.py --options use_graph=0 --gen 96 "
                 "(shipped defaults: fused q/k/v + attention, fused out_proj + fc1); FETCH_SIZE doubled (gfx950 counts the 128-B requests of a wide stream at 64 B)",
       "hbm_bytes_per_launch": {c: int(v[0] / max(1, v[1])) for c, v in acc.items()},
       "dispatches": {c: v[1] for c, v in acc.items()}, "per_kernel": per_kernel,
       "note": "the 'cache' class mixes cache lengths 257..~450 of the profiled run: compare with algorithmic bytes at that length, not at mid context"}
json.dump(res, open(os.path.join(out, RND + "_pmc_decode_traffic.json"), "w"), indent=1, sort_keys=True)

rows = {}
for k, cs in den.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in cs or "GRBM_GUI_ACTIVE" not in cs:
        continue
    mf, ga = cs["SQ_VALU_MFMA_BUSY_CYCLES"], cs["GRBM_GUI_ACTIVE"]
    if mf["mean"] <= 0:
        continue
    rows[k] = {"dispatches": mf["dispatches"], "SQ_VALU_MFMA_BUSY_CYCLES": round(mf["mean"], 1), "GRBM_GUI_ACTIVE": round(ga["mean"], 1),
               "mfma_busy_frac": round(mf["mean"] / (ga["mean"] / 8.0 * 1024), 4)}
DENSE = ("gemm_tile", "gemm256", "gemm_dec", "gemm_mfma_f32", "attention", "vt_pack", "ln_rows2", "kv_fill2", "kv_fill_rows", "cvt_rows", "add_rows2", "codes_gather2", "fourier2", "coords_argmax")
def is_f32(k):                       # kernels of the exact (fp32) point encoder: fp32 matrix path, fp32 attention, fp32 row kernels
    return "gemm_mfma_f32" in k or "attention_f32" in k or ("<float>" in k and "unsigned short" not in k and "_Float16" not in k)
def busy(pred):
    m = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] * v["dispatches"] for k, v in rows.items() if pred(k))
    a = sum(cs["GRBM_GUI_ACTIVE"]["mean"] * cs["GRBM_GUI_ACTIVE"]["dispatches"] for k, cs in den.items()
            if "GRBM_GUI_ACTIVE" in cs and any(t in k for t in DENSE) and pred(k))
    return round(m / max(1.0, a / 8.0 * 1024), 4)
json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace on scripts/prof_dense.py --batches 64 --iters 1",
           "formula": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); a phase = MFMA cycles of its kernels / active cycles of ALL its kernels (GEMM, attention, LayerNorm, gathers, casts)",
           "dense_16bit_phases_mfma_busy_frac": busy(lambda k: not is_f32(k)), "fp32_encoder_mfma_busy_frac": busy(is_f32),
           "dense_phase_mfma_busy_frac": busy(lambda k: True), "per_kernel": rows},
          open(os.path.join(out, RND + "_pmc_dense_mfma.json"), "w"), indent=1, sort_keys=True)
