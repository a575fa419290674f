// Weight-streaming GEMV: y[N] = epilogue(W[N,K] . prologue(x)[K]).
//
// This is the decode step's dominant kernel (reference call sites: the nn.Linear calls inside [3p] OPTDecoderLayer
// reached from shape_opt.py:403-410, `input_layer` shape_opt.py:243, `lm_head` shape_opt.py:155).  At batch 1 every
// weight byte is used once, so the kernel is a pure HBM stream and -- because one launch only moves 2-16 MB -- a
// latency chain: launch ramp -> operands arrive -> reduce -> epilogue.  Everything is arranged to keep that chain short
// and the traffic next to the weights small:
//   * the input vector is produced ONCE per block (not once per wave: with ~4000 waves per launch, per-wave copies of a
//     4-16 KB vector plus LayerNorm affine cost several times the weight bytes in L2 traffic): 256 threads load it,
//     apply the prologue (nothing | LayerNorm of the post-LN residual stream | merge of the split-KV attention
//     partials), round it to the policy dtype and park it in LDS; waves then read their K-slices with ds_read_b128;
//   * all global loads are issued in the first instructions, in the order they are consumed (vmcnt returns in order):
//     prologue operands, this wave's weight pieces (non-temporal), epilogue operands (bias / residual / tables);
//     the decode state is ONE scalar load;
//   * one wave owns RPW whole rows or a K-slice of them; a lane holds <= 8 16-byte weight pieces, a wave reads 1 KiB
//     contiguous per instruction;
//   * reductions run on DPP (common.hpp), not on the LDS crossbar.
//
// Algorithmic bytes per launch: N*K*sizeof(WT) (+ N*4 bias, negligible).
#pragma once
#include "attn_decode.hpp"
#include "common.hpp"
#include "state.hpp"

namespace ma {

enum GemvEpi { EPI_PLAIN = 0, EPI_QKV = 1, EPI_EMBED = 2, EPI_LMHEAD = 3 };
enum GemvPro { PRO_PLAIN = 0, PRO_LN = 1, PRO_ATTN = 2 };

struct GemvArgs {
    const void* W;          // [N][K] row-major, WT
    const float* bias;      // [N] or null
    const float* x;         // [K] (ignored by EPI_EMBED, which reads the codebook row of the current token, and by PRO_ATTN)
    const float* ln_g;      // LayerNorm prologue on x when non-null (PRO_LN)
    const float* ln_b;
    float ln_eps;
    float* xn_out;          // prologue(x) before rounding, written once by block 0 when non-null: the residual for a later epilogue
    const float* res;       // [N] residual added after the activation, or null
    float* y;               // [N]
    void* yb; int yb_stride; // EPI_EMBED, 16-bit policies: the same output rounded to WT as well (the batched step's first q/k/v GEMM operand), or null
    int N, K, act, round_x, epi;
    // EPI_QKV: rows [0,H) -> y (q, fp32); [H,2H) -> K cache; [2H,3H) -> V cache at position st->pos
    void* kcache; void* vcache; int H; int max_seq;
    // EPI_EMBED (shape_opt.py:237-245, 323-328, 359-364): see the epilogue
    const float* codebook; const float* extra; const float* tokpos; const float* cond; const float* postab; int T;
    // EPI_LMHEAD: per-block argmax partials (greedy pick), optional eos suppression
    float* part_val; int* part_idx;
    const DecState* st;
    // PRO_ATTN: x = merge of the split-KV attention partials (attn_decode.hpp), K = attn_heads * 64
    const float* attn_ws; int attn_heads;
    // batch: grid.y = batch row b.  Per-row operands advance by these element strides; the weights are shared (rows after
    // the first find them in L2 / Infinity Cache).  kv_row_stride: elements between two rows' cache planes.
    int x_stride, y_stride, res_stride, xn_stride, part_stride; size_t kv_row_stride; size_t attn_ws_stride;
    // diagnostics (ma_trace_decode): wave 0 of block b stores the 100 MHz real-time counter at four points into trace[b*4..]
    unsigned long long* trace;
};
#define MA_TRACE(tr, slot) do { if ((tr) && threadIdx.x == 0) (tr)[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)

template <typename WT> struct WTraits;
template <> struct WTraits<float>  { static constexpr int VEC = 4; };
template <> struct WTraits<bf16_t> { static constexpr int VEC = 8; };
template <> struct WTraits<f16_t> { static constexpr int VEC = 8; };

template <typename WT>
__device__ inline void unpack16(const u32x4& w, float* f) {
    if constexpr (sizeof(WT) == 4) {
        f[0] = __uint_as_float(w.x); f[1] = __uint_as_float(w.y); f[2] = __uint_as_float(w.z); f[3] = __uint_as_float(w.w);
    } else unpack8<WT>(w, f);
}

template <typename KT> __device__ inline void store_kv(KT* p, float v) {
    if constexpr (sizeof(KT) == 4) *p = v;
    else *reinterpret_cast<uint16_t*>(p) = H16<KT>::bits(v);
}

// Work decomposition: one block = 4 waves = RPB = (4/KSPLIT)*RPW output rows; KSPLIT waves share a group of RPW rows,
// each owning LPL 16-byte pieces per lane and row (K = KC = KSPLIT * LPL * 64 * VEC).  Lane j of the first wave of a
// group finishes row j of the group.
// LPL == 0: generic path for small / odd K (K % VEC == 0): KSPLIT = RPW = 1, per-wave prologue, no LDS staging.
template <typename WT, int KSPLIT, int LPL, int RPW, int PRO>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs a) {
    constexpr int VEC = WTraits<WT>::VEC;
    constexpr int RPB = (4 / KSPLIT) * RPW;
    constexpr int KC = LPL > 0 ? KSPLIT * LPL * 64 * VEC : 4;
    __shared__ __attribute__((aligned(16))) float xl[KC];
    __shared__ float red[8];
    __shared__ float lv[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int grp = w / KSPLIT, wk = w % KSPLIT;
    const int row0 = blockIdx.x * RPB + grp * RPW;       // first row of this wave's group
    const WT* W = reinterpret_cast<const WT*>(a.W);
    const int K = a.K, N = a.N;
    const int my_row = row0 + lane;                      // row finished by this lane (lanes < RPW of the group's first wave)
    const bool fin = wk == 0 && lane < RPW && my_row < N;
    MA_TRACE(a.trace, 0);
    // batch row: per-row operands (blockIdx.y = 0 and zero strides at batch 1)
    const int brow = blockIdx.y;
    a.y += (size_t)brow * a.y_stride;
    if (a.x) a.x += (size_t)brow * a.x_stride;
    if (a.res) a.res += (size_t)brow * a.res_stride;
    if (a.xn_out) a.xn_out += (size_t)brow * a.xn_stride;
    if (a.attn_ws) a.attn_ws += (size_t)brow * a.attn_ws_stride;

    // ---- (0) decode state: ONE scalar load of the whole record (no dependent scalar round trips) ------------------
    const float* x = a.x;
    int tok = 0, tstep = 0, pos = 0, skip = -1;
    bool skip_dot = false;
    if (a.epi != EPI_PLAIN) {
        const DecState sv = a.st[brow];
        tok = sv.cur_tok; tstep = sv.t; pos = sv.pos;
        if (a.epi == EPI_LMHEAD) skip = sv.suppress_eos ? 1 : -1;      // eos = 1 (meshanything.py:103)
        if (a.epi == EPI_EMBED) {
            skip_dot = tok < 3;                  // bos/eos/pad use extra_embeds, no Linear (shape_opt.py:240-241)
            x = a.codebook + (size_t)(skip_dot ? 0 : tok - 3) * K;
        }
    }
    float e_bias = 0.f, e_res = 0.f, e_t0 = 0.f, e_t1 = 0.f, e_t2 = 0.f, e_t3 = 0.f;
    // epilogue operands of the finishing lanes: issued right behind the weight loads (they come back with them)
    auto load_epilogue_operands = [&]() {
        if (fin) {
            if (a.bias) e_bias = a.bias[my_row];
            if (a.res) e_res = a.res[my_row];
            if (a.epi == EPI_EMBED) {
                int m = (tstep - 2) % 9; if (m < 0) m += 9;            // python modulo (shape_opt.py:457)
                const int slot = skip_dot ? tok : m + 3;
                e_t0 = skip_dot ? a.extra[(size_t)tok * N + my_row] : 0.f;
                e_t1 = a.tokpos[(size_t)slot * N + my_row];
                e_t2 = a.cond[N + my_row];                             // cond_embed row 1 (generated tokens)
                e_t3 = a.postab[(size_t)(a.T + tstep - 1 + 2) * N + my_row];
            }
        }
    };

    float acc[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) acc[j] = 0.f;

    if constexpr (LPL > 0) {
        // ---- (1) prologue operands: thread t owns the float4 chunks t, t+256, ... of the input vector ---------------
        constexpr int NCH = (KC / 4 + 255) / 256;
        f32x4 xv[NCH], gv[PRO == PRO_LN ? NCH : 1], bv[PRO == PRO_LN ? NCH : 1];
        f32x4 pml[PRO == PRO_ATTN ? ATTN_NCHUNK / 2 : 1], po[PRO == PRO_ATTN ? ATTN_NCHUNK : 1];
        float x0 = 0.f;                      // PRO_LN: shift of the one-pass statistics (element 0 of the row, a broadcast load)
        if constexpr (PRO == PRO_LN) x0 = x[0];
        if constexpr (PRO == PRO_ATTN) {
            static_assert(PRO != PRO_ATTN || NCH == 1, "PRO_ATTN fast path: K <= 1024");
            const int k = tid * 4, h = k >> 6, d0 = k & 63;
            if (k < KC) attn_partials_load(a.attn_ws, a.attn_heads, h, d0, pml, po);
        } else {
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int idx = tid + 256 * j;
                if (idx < KC / 4) {
                    xv[j] = *reinterpret_cast<const f32x4*>(x + idx * 4);
                    if constexpr (PRO == PRO_LN) {
                        gv[j] = *reinterpret_cast<const f32x4*>(a.ln_g + idx * 4);
                        bv[j] = *reinterpret_cast<const f32x4*>(a.ln_b + idx * 4);
                    }
                } else {
                    xv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (PRO == PRO_LN) { gv[j] = xv[j]; bv[j] = xv[j]; }
                }
            }
        }
        // ---- (2) this wave's weight pieces, then the epilogue operands ------------------------------------------------
        u32x4 wv[RPW][LPL];
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const WT* wr = W + (size_t)min(row0 + j, N - 1) * K;         // tail rows re-read the last row; result discarded
#pragma unroll
            for (int i = 0; i < LPL; ++i) wv[j][i] = ld_stream16(wr + ((wk * LPL + i) * 64 + lane) * VEC);
        }
        load_epilogue_operands();
        // ---- (3) prologue math, result parked in LDS (rounded to the policy dtype) ------------------------------------
        if constexpr (PRO == PRO_LN) {
            ln_block_onepass<NCH>(xv, gv, bv, x0, tid, KC / 4, K, a.ln_eps, red);
        } else if constexpr (PRO == PRO_ATTN) {
            xv[0] = attn_partials_merge(pml, po);
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int idx = tid + 256 * j;
            if (idx < KC / 4) {
                if (a.xn_out && blockIdx.x == 0) *reinterpret_cast<f32x4*>(a.xn_out + idx * 4) = xv[j];
                f32x4 r = xv[j];
                if (a.round_x) { r.x = H16<WT>::round(r.x); r.y = H16<WT>::round(r.y); r.z = H16<WT>::round(r.z); r.w = H16<WT>::round(r.w); }
                *reinterpret_cast<f32x4*>(&xl[idx * 4]) = r;
            }
        }
        __syncthreads();
        MA_TRACE(a.trace, 1);
        // ---- (4) dot products ---------------------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < LPL; ++i) {
            const int k0 = ((wk * LPL + i) * 64 + lane) * VEC;
            float xs[VEC];
#pragma unroll
            for (int v = 0; v < VEC; v += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(&xl[k0 + v]);
                xs[v] = t.x; xs[v + 1] = t.y; xs[v + 2] = t.z; xs[v + 3] = t.w;
            }
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                float wf[VEC];
                unpack16<WT>(wv[j][i], wf);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[j] = fmaf(wf[v], xs[v], acc[j]);
            }
        }
    } else {
        // generic path (tiny / odd shapes): every wave runs the prologue for itself, straight from global memory
        static_assert(LPL > 0 || (KSPLIT == 1 && RPW == 1), "generic path: one row per wave");
        load_epilogue_operands();
        constexpr int CH = 64 * VEC;
        float mean = 0.f, rstd = 1.f;
        const int nc = (K + CH - 1) / CH;
        const int rowc = min(row0, N - 1);
        if constexpr (PRO == PRO_LN) {
            float s = 0.f;
            for (int k = lane; k < K; k += 64) s += x[k];
            mean = wave_sum(s) / (float)K;
            float q = 0.f;
            for (int k = lane; k < K; k += 64) { float d = x[k] - mean; q += d * d; }
            rstd = 1.0f / sqrtf(wave_sum(q) / (float)K + a.ln_eps);
        }
        for (int c = 0; c < nc; ++c) {
            const int k0 = (c * 64 + lane) * VEC;
            if (k0 < K) {
                float xv[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    float t;
                    if constexpr (PRO == PRO_ATTN) t = attn_partials_merge_one(a.attn_ws, a.attn_heads, (k0 + v) >> 6, (k0 + v) & 63);
                    else t = x[k0 + v];
                    if constexpr (PRO == PRO_LN) t = (t - mean) * rstd * a.ln_g[k0 + v] + a.ln_b[k0 + v];
                    if (a.xn_out && blockIdx.x == 0 && w == 0) a.xn_out[k0 + v] = t;
                    xv[v] = a.round_x ? H16<WT>::round(t) : t;
                }
                u32x4 wq = ld_stream16(W + (size_t)rowc * K + k0);
                float wf[VEC];
                unpack16<WT>(wq, wf);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[0] = fmaf(wf[v], xv[v], acc[0]);
            }
        }
    }

    // ---- reduce: lane j of the group's first wave ends up with row j's dot product -----------------------------------
    MA_TRACE(a.trace, 2);
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const float t = wave_sum(acc[j]);
        if constexpr (KSPLIT > 1) { if (lane == 0) red[(grp * RPW + j) * KSPLIT + wk] = t; }
        else { if (lane == j) v = t; }
    }
    if constexpr (KSPLIT > 1) {
        __syncthreads();
        if (fin) {
#pragma unroll
            for (int k = 0; k < KSPLIT; ++k) v += red[(grp * RPW + lane) * KSPLIT + k];
        }
    }

    // ---- epilogue: operands already in registers ----------------------------------------------------------------------
    if (a.epi == EPI_LMHEAD) {
        // logits + one argmax partial per block
        if (fin) { a.y[my_row] = v; lv[grp * RPW + lane] = v; }
        __syncthreads();
        if (tid == 0) {
            float bvv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < RPB; ++r) {
                const int n = blockIdx.x * RPB + r;
                if (n < N && n != skip && arg_better(lv[r], n, bvv, bi)) { bvv = lv[r]; bi = n; }
            }
            a.part_val[(size_t)brow * a.part_stride + blockIdx.x] = bvv; a.part_idx[(size_t)brow * a.part_stride + blockIdx.x] = bi;
        }
    } else if (fin) {
        if (a.epi == EPI_EMBED) {
            // e = (extra[tok] | input_layer(codebook[tok-3])) + token_embed_positions[slot] + cond_embed[1] + embed_positions[T+t-1+2]
            float e = skip_dot ? e_t0 : v + e_bias;
            e += e_t1;
            e += e_t2;
            e += e_t3;
            a.y[my_row] = e;
            if constexpr (sizeof(WT) == 2) { if (a.yb) store_kv<WT>(reinterpret_cast<WT*>(a.yb) + (size_t)brow * a.yb_stride + my_row, e); }
        } else {
            v += e_bias;                                   // e_bias / e_res are 0 when absent
            v = apply_act(v, a.act);
            v += e_res;
            if (a.epi == EPI_QKV) {
                const int part = my_row / a.H, c = my_row - part * a.H;
                if (part == 0) a.y[c] = v;
                else {
                    const int head = c >> 6, d = c & 63;
                    const size_t off = (size_t)brow * a.kv_row_stride + ((size_t)head * a.max_seq + pos) * 64 + d;
                    store_kv<WT>(reinterpret_cast<WT*>(part == 1 ? a.kcache : a.vcache) + off, v);
                }
            } else {
                a.y[my_row] = v;
            }
        }
    }
    MA_TRACE(a.trace, 3);
}

// ---- host-side dispatch ------------------------------------------------------------------------------------------
struct GemvShape { int ksplit, lpl, rpw; };
// rows per wave for the big matrices (engine option "gemv_rpw"): 1 = most blocks ... 4 = a quarter of the blocks / input copies
// (default 4: measured 1.4 % (2 vs 1) + 0.5 % (4 vs 2) faster per decode step, profiles/r01_ab_rows_per_wave.txt)
inline int& gemv_rpw_big() { static int v = 4; return v; }
// N <= 2048, K = 2 pieces (out_proj, embed): 0 = two waves split K, one row per group (512 blocks at N = 1024);
// r > 0 = one wave per r whole rows (4 r rows per block: fewer blocks repeat the prologue -- out_proj's merges 67 KB of partials).
// Default 1 (256 blocks): decode step 519.6 -> 491.7 us at kv 300 (profiles/r02_ab_oproj_block_shape.txt)
inline int& gemv_small_rows() { static int v = 1; return v; }
// K = 8 pieces (fc2): waves that split K (4 = 1024 blocks at N = 1024, each staging the 16 KB input; 2; 1 = one wave per row, 256 blocks).
// Default 1: 494.2 -> 483.0 us (profiles/r02_ab_fc2_block_shape.txt)
inline int& gemv_k8_ksplit() { static int v = 1; return v; }

template <typename WT>
inline GemvShape gemv_shape(int N, int K) {
    constexpr int VEC = WTraits<WT>::VEC;
    if (K % (64 * VEC) != 0) return {1, 0, 1};
    const int nc = K / (64 * VEC);                       // 16-byte pieces per lane for one row
    const int rpw = N >= 2048 ? gemv_rpw_big() : 1;
    switch (nc) {
        case 1: return {1, 1, rpw > 2 ? 2 : rpw};
        // > 1024 blocks do not fit the chip at once (8195 lm_head rows: 2049 blocks start over 2.9 us): two rows per wave
        case 2: return N <= 2048 ? (gemv_small_rows() > 0 ? GemvShape{1, 2, gemv_small_rows()} : GemvShape{2, 1, 1}) : GemvShape{1, 2, N > 4096 ? (rpw > 2 ? rpw : 2) : rpw};
        case 4: return {2, 2, N > 4096 ? 2 : (rpw > 2 ? 2 : rpw)};
        case 8: return gemv_k8_ksplit() == 1 ? GemvShape{1, 8, 1} : gemv_k8_ksplit() == 2 ? GemvShape{2, 4, 1} : GemvShape{4, 2, 1};
        case 16: return {4, 4, 1};
        default: return {1, 0, 1};
    }
}
template <typename WT>
inline int gemv_rows_per_block(int N, int K) { const GemvShape g = gemv_shape<WT>(N, K); return (4 / g.ksplit) * g.rpw; }
template <typename WT>
inline int gemv_num_blocks(int N, int K) { const int rpb = gemv_rows_per_block<WT>(N, K); return (N + rpb - 1) / rpb; }

template <typename WT, int KS, int LP, int RW>
inline void launch_gemv_pro(const GemvArgs& a, int pro, dim3 grid, hipStream_t s) {
    if (pro == PRO_LN) hipLaunchKernelGGL((gemv_kernel<WT, KS, LP, RW, PRO_LN>), grid, dim3(256), 0, s, a);
    else if (pro == PRO_ATTN) {
        if constexpr (LP == 0 || KS * LP * 64 * WTraits<WT>::VEC <= 1024) hipLaunchKernelGGL((gemv_kernel<WT, KS, LP, RW, PRO_ATTN>), grid, dim3(256), 0, s, a);
    } else hipLaunchKernelGGL((gemv_kernel<WT, KS, LP, RW, PRO_PLAIN>), grid, dim3(256), 0, s, a);
}

template <typename WT>
inline hipError_t launch_gemv(const GemvArgs& a, hipStream_t s, int batch = 1) {
    constexpr int VEC = WTraits<WT>::VEC;
    if (a.K % VEC != 0) return hipErrorInvalidValue;
    const int pro = a.attn_ws ? PRO_ATTN : (a.ln_g ? PRO_LN : PRO_PLAIN);
    if (pro == PRO_ATTN && (a.K != a.attn_heads * 64)) return hipErrorInvalidValue;
    GemvShape g = gemv_shape<WT>(a.N, a.K);
    if (pro == PRO_ATTN && g.lpl > 0 && a.K > 1024) g = GemvShape{1, 0, 1};       // wide merges take the generic path
    const int rpb = (4 / g.ksplit) * g.rpw;
    const dim3 grid((a.N + rpb - 1) / rpb, batch);
#define MA_GEMV_CASE(KS, LP, RW) if (g.ksplit == KS && g.lpl == LP && g.rpw == RW) { launch_gemv_pro<WT, KS, LP, RW>(a, pro, grid, s); return hipGetLastError(); }
    MA_GEMV_CASE(1, 1, 1) MA_GEMV_CASE(1, 1, 2) MA_GEMV_CASE(2, 1, 1) MA_GEMV_CASE(1, 2, 1) MA_GEMV_CASE(1, 2, 2) MA_GEMV_CASE(1, 2, 4) MA_GEMV_CASE(2, 2, 1) MA_GEMV_CASE(2, 2, 2)
    MA_GEMV_CASE(4, 2, 1) MA_GEMV_CASE(2, 4, 1) MA_GEMV_CASE(1, 8, 1) MA_GEMV_CASE(4, 4, 1) MA_GEMV_CASE(1, 0, 1)
#undef MA_GEMV_CASE
    return hipErrorInvalidValue;
}

}  // namespace ma
