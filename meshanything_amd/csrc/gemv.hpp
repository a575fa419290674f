// Weight-streaming GEMV: y[N] = epilogue(W[N,K] . prologue(x)[K]).
//
// This is the decode step's dominant kernel (reference call sites: the nn.Linear calls inside [3p] OPTDecoderLayer
// reached from shape_opt.py:403-410, `input_layer` shape_opt.py:243, `lm_head` shape_opt.py:155).  At batch 1 every
// weight byte is used once, so the kernel is a pure HBM stream: one wave owns RPW output rows, each lane loads
// 16-byte pieces of the rows it owns (a wave reads 1 KiB contiguous per instruction), all loads of a wave are issued
// before anything waits, the x vector lives in registers (K/64 values per lane), the LayerNorm of the post-LN
// residual stream is recomputed in the prologue by every wave (wave-local: no LDS, no barrier) and the reduction is
// a 6-step butterfly.  No LDS is used at all: the operand is streamed once and not shared (guide: GEMV / M<=16 rule).
//
// Algorithmic bytes per launch: N*K*sizeof(WT) (+ N*4 bias, negligible).
#pragma once
#include "common.hpp"
#include "state.hpp"

namespace ma {

enum GemvEpi { EPI_PLAIN = 0, EPI_QKV = 1, EPI_EMBED = 2, EPI_LMHEAD = 3 };

struct GemvArgs {
    const void* W;          // [N][K] row-major, WT
    const float* bias;      // [N] or null
    const float* x;         // [K] (ignored by EPI_EMBED, which reads the codebook row of the current token)
    const float* ln_g;      // LayerNorm prologue on x when non-null
    const float* ln_b;
    float ln_eps;
    float* xn_out;          // LN(x) written once (by wave 0 of block 0) when non-null: the residual for a later epilogue
    const float* res;       // [N] residual added after the activation, or null
    float* y;               // [N]
    int N, K, act, round_x, epi;
    // EPI_QKV: rows [0,H) -> y (q, fp32); [H,2H) -> K cache; [2H,3H) -> V cache at position st->pos
    void* kcache; void* vcache; int H; int max_seq;
    // EPI_EMBED (shape_opt.py:237-245, 323-328, 359-364): see embed_epilogue()
    const float* codebook; const float* extra; const float* tokpos; const float* cond; const float* postab; int T;
    // EPI_LMHEAD: per-wave argmax partials (greedy pick), optional eos suppression
    float* part_val; int* part_idx;
    const DecState* st;
};

template <typename WT> struct WTraits;
template <> struct WTraits<float>  { static constexpr int VEC = 4; };
template <> struct WTraits<bf16_t> { static constexpr int VEC = 8; };

template <typename WT>
__device__ inline void unpack16(const u32x4& w, float* f) {
    if constexpr (sizeof(WT) == 4) {
        f[0] = __uint_as_float(w.x); f[1] = __uint_as_float(w.y); f[2] = __uint_as_float(w.z); f[3] = __uint_as_float(w.w);
    } else {
        f[0] = bf_lo(w.x); f[1] = bf_hi(w.x); f[2] = bf_lo(w.y); f[3] = bf_hi(w.y);
        f[4] = bf_lo(w.z); f[5] = bf_hi(w.z); f[6] = bf_lo(w.w); f[7] = bf_hi(w.w);
    }
}

template <typename KT> __device__ inline void store_kv(KT* p, float v);
template <> __device__ inline void store_kv<float>(float* p, float v) { *p = v; }
template <> __device__ inline void store_kv<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// Work decomposition (v2, after the launch-floor microbenchmark scripts/ubench_launch.hip: a streaming kernel needs >= ~1000
// blocks and <= 2-4 loads per lane to reach the ~2.7-3.6 us floor for 2-8 MB; 256 fat blocks cost up to 2x that):
//   one block = 4 waves = RPB = 4/KSPLIT output rows; KSPLIT waves share one row, each owning LPL 16-byte pieces per lane
//   (K = KSPLIT * LPL * 64 * VEC); partial sums meet in LDS (one barrier), thread r finishes row r.
// LPL == 0: generic path for small / odd K (K % VEC == 0), KSPLIT = 1.
template <typename WT, int KSPLIT, int LPL>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs a) {
    constexpr int VEC = WTraits<WT>::VEC;
    constexpr int RPB = 4 / KSPLIT;
    __shared__ float red[4];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int row_in_block = w / KSPLIT, wk = w % KSPLIT;
    const int row = blockIdx.x * RPB + row_in_block;
    const int rowc = min(row, a.N - 1);                  // tail rows re-read the last row; result discarded
    const WT* W = reinterpret_cast<const WT*>(a.W);
    const int K = a.K;

    // ---- input vector selection -------------------------------------------------------------------------------
    const float* x = a.x;
    int tok = 0, tstep = 0;
    bool skip_dot = false;
    if (a.epi == EPI_EMBED) {
        tok = a.st->cur_tok;
        tstep = a.st->t;
        skip_dot = tok < 3;                      // bos/eos/pad use extra_embeds, no Linear (shape_opt.py:240-241)
        x = a.codebook + (size_t)(skip_dot ? 0 : tok - 3) * K;
    }
    const bool has_ln = a.ln_g != nullptr;
    float acc = 0.f;

    if constexpr (LPL > 0) {
        // (1) everything small first (x slices, LN affine slices, then the full x for the statistics): these return
        //     first (vmcnt is in-order) and come from L2; (2) then this wave's weight pieces; (3) LN math overlaps (2).
        float xs[LPL][VEC], gs[LPL][VEC], bs[LPL][VEC];
#pragma unroll
        for (int i = 0; i < LPL; ++i) {
            const int k0 = ((wk * LPL + i) * 64 + lane) * VEC;
#pragma unroll
            for (int v = 0; v < VEC; v += 4) {
                f32x4 t = *reinterpret_cast<const f32x4*>(x + k0 + v);
                xs[i][v] = t.x; xs[i][v + 1] = t.y; xs[i][v + 2] = t.z; xs[i][v + 3] = t.w;
                if (has_ln) {
                    f32x4 g = *reinterpret_cast<const f32x4*>(a.ln_g + k0 + v);
                    f32x4 b = *reinterpret_cast<const f32x4*>(a.ln_b + k0 + v);
                    gs[i][v] = g.x; gs[i][v + 1] = g.y; gs[i][v + 2] = g.z; gs[i][v + 3] = g.w;
                    bs[i][v] = b.x; bs[i][v + 1] = b.y; bs[i][v + 2] = b.z; bs[i][v + 3] = b.w;
                }
            }
        }
        constexpr int XALL = KSPLIT * LPL * VEC;         // K / 64 values of x per lane
        float xall[XALL];
        if (has_ln) {
#pragma unroll
            for (int j = 0; j < XALL; j += 4) {
                f32x4 t = *reinterpret_cast<const f32x4*>(x + ((j / 4) * 64 + lane) * 4);
                xall[j] = t.x; xall[j + 1] = t.y; xall[j + 2] = t.z; xall[j + 3] = t.w;
            }
        }
        u32x4 wv[LPL];
        const WT* wr = W + (size_t)rowc * K;
#pragma unroll
        for (int i = 0; i < LPL; ++i) wv[i] = ld_stream16(wr + ((wk * LPL + i) * 64 + lane) * VEC);
        if (has_ln) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < XALL; ++j) s += xall[j];
            const float mean = wave_sum(s) / (float)K;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < XALL; ++j) { const float d = xall[j] - mean; q += d * d; }
            const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)K + a.ln_eps);
#pragma unroll
            for (int i = 0; i < LPL; ++i)
#pragma unroll
                for (int v = 0; v < VEC; ++v) xs[i][v] = (xs[i][v] - mean) * rstd * gs[i][v] + bs[i][v];
            if (a.xn_out && blockIdx.x == 0 && row_in_block == 0) {     // the KSPLIT waves of row 0 cover all of x once
#pragma unroll
                for (int i = 0; i < LPL; ++i) {
                    const int k0 = ((wk * LPL + i) * 64 + lane) * VEC;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) a.xn_out[k0 + v] = xs[i][v];
                }
            }
        }
        if (a.round_x) {
#pragma unroll
            for (int i = 0; i < LPL; ++i)
#pragma unroll
                for (int v = 0; v < VEC; ++v) xs[i][v] = round_bf16(xs[i][v]);
        }
#pragma unroll
        for (int i = 0; i < LPL; ++i) {
            float wf[VEC];
            unpack16<WT>(wv[i], wf);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc = fmaf(wf[v], xs[i][v], acc);
        }
    } else {
        constexpr int CH = 64 * VEC;
        float mean = 0.f, rstd = 1.f;
        const int nc = (K + CH - 1) / CH;
        if (has_ln) {
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
                    float t = x[k0 + v];
                    if (has_ln) {
                        t = (t - mean) * rstd * a.ln_g[k0 + v] + a.ln_b[k0 + v];
                        if (a.xn_out && blockIdx.x == 0 && w == 0) a.xn_out[k0 + v] = t;
                    }
                    xv[v] = a.round_x ? round_bf16(t) : t;
                }
                u32x4 wq = ld_stream16(W + (size_t)rowc * K + k0);
                float wf[VEC];
                unpack16<WT>(wq, wf);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc = fmaf(wf[v], xv[v], acc);
            }
        }
    }

    acc = wave_sum(acc);
    if (lane == 0) red[w] = acc;
    __syncthreads();

    // ---- epilogue: thread r finishes row r of the block ---------------------------------------------------------
    if (a.epi == EPI_LMHEAD) {
        if (threadIdx.x == 0) {
            const int skip = a.st->suppress_eos ? 1 : -1;     // eos = 1 (meshanything.py:103)
            float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < RPB; ++r) {
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < KSPLIT; ++k) v += red[r * KSPLIT + k];
                const int n = blockIdx.x * RPB + r;
                if (n < a.N) {
                    a.y[n] = v;
                    if (n != skip && arg_better(v, n, bv, bi)) { bv = v; bi = n; }
                }
            }
            a.part_val[blockIdx.x] = bv; a.part_idx[blockIdx.x] = bi;
        }
        return;
    }
    const int r = threadIdx.x;
    const int n = blockIdx.x * RPB + r;
    if (r >= RPB || n >= a.N) return;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < KSPLIT; ++k) v += red[r * KSPLIT + k];
    if (a.epi == EPI_EMBED) {
        // e = (extra[tok] | input_layer(codebook[tok-3])) + token_embed_positions[slot] + cond_embed[1] + embed_positions[T+t-1+2]
        float e = skip_dot ? a.extra[(size_t)tok * a.N + n] : v + a.bias[n];
        int m = (tstep - 2) % 9; if (m < 0) m += 9;            // python modulo (shape_opt.py:457)
        const int slot = skip_dot ? tok : m + 3;
        e += a.tokpos[(size_t)slot * a.N + n];
        e += a.cond[a.N + n];                                  // cond_embed row 1 (generated tokens)
        e += a.postab[(size_t)(a.T + tstep - 1 + 2) * a.N + n];
        a.y[n] = e;
        return;
    }
    if (a.bias) v += a.bias[n];
    v = apply_act(v, a.act);
    if (a.res) v += a.res[n];
    if (a.epi == EPI_QKV) {
        const int part = n / a.H, c = n - part * a.H;
        if (part == 0) { a.y[c] = v; return; }
        const int head = c >> 6, d = c & 63;
        const size_t off = ((size_t)head * a.max_seq + a.st->pos) * 64 + d;
        store_kv<WT>(reinterpret_cast<WT*>(part == 1 ? a.kcache : a.vcache) + off, v);
        return;
    }
    a.y[n] = v;
}

// ---- host-side dispatch ------------------------------------------------------------------------------------------
struct GemvShape { int ksplit, lpl; };
template <typename WT>
inline GemvShape gemv_shape(int N, int K) {
    constexpr int VEC = WTraits<WT>::VEC;
    if (K % (64 * VEC) != 0) return {1, 0};
    const int nc = K / (64 * VEC);                       // 16-byte pieces per lane for one row
    switch (nc) {
        case 1: return {1, 1};
        case 2: return N <= 2048 ? GemvShape{2, 1} : GemvShape{1, 2};
        case 4: return {2, 2};
        case 8: return {4, 2};
        case 16: return {4, 4};
        default: return {1, 0};
    }
}
template <typename WT>
inline int gemv_num_blocks(int N, int K) { const GemvShape g = gemv_shape<WT>(N, K); return (N + 4 / g.ksplit - 1) / (4 / g.ksplit); }

template <typename WT>
inline hipError_t launch_gemv(const GemvArgs& a, hipStream_t s) {
    if (a.K % WTraits<WT>::VEC != 0) return hipErrorInvalidValue;
    const GemvShape g = gemv_shape<WT>(a.N, a.K);
    const dim3 grid(gemv_num_blocks<WT>(a.N, a.K)), block(256);
#define MA_GEMV_CASE(KS, LP) if (g.ksplit == KS && g.lpl == LP) { hipLaunchKernelGGL((gemv_kernel<WT, KS, LP>), grid, block, 0, s, a); return hipGetLastError(); }
    MA_GEMV_CASE(1, 1) MA_GEMV_CASE(2, 1) MA_GEMV_CASE(1, 2) MA_GEMV_CASE(2, 2) MA_GEMV_CASE(4, 2) MA_GEMV_CASE(4, 4) MA_GEMV_CASE(1, 0)
#undef MA_GEMV_CASE
    return hipErrorInvalidValue;
}

}  // namespace ma
