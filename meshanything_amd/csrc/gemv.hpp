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

// NC > 0: K == NC*64*VEC exactly, x and all weight pieces of the wave are held in registers (fast path).
// NC == 0: any K with K % VEC == 0 (small test shapes): chunks are walked with x re-read from L1/L2.
template <typename WT, int RPW, int NC>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs a) {
    constexpr int VEC = WTraits<WT>::VEC;
    constexpr int CH = 64 * VEC;
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    const int gwave = blockIdx.x * 4 + wave_in_block;
    const int row0 = gwave * RPW;
    if (row0 >= a.N) {
        // whole wave idle (tail); an LMHEAD partial slot still has to be defined
        if (a.epi == EPI_LMHEAD && lane == 0) { a.part_val[gwave] = -INFINITY; a.part_idx[gwave] = 0x7fffffff; }
        return;
    }
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

    float acc[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) acc[r] = 0.f;

    if constexpr (NC > 0) {
        // (1) x slices first (they return first: vmcnt is in-order), then every weight piece of this wave
        float xs[NC][VEC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float* xp = x + (c * 64 + lane) * VEC;
#pragma unroll
            for (int v = 0; v < VEC; v += 4) {
                f32x4 t = *reinterpret_cast<const f32x4*>(xp + v);
                xs[c][v] = t.x; xs[c][v + 1] = t.y; xs[c][v + 2] = t.z; xs[c][v + 3] = t.w;
            }
        }
        u32x4 wv[RPW][NC];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = min(row0 + r, a.N - 1);          // clamp: tail rows re-read the last row, result discarded
            const WT* wr = W + (size_t)row * K;
#pragma unroll
            for (int c = 0; c < NC; ++c) wv[r][c] = ld_stream16(wr + (c * 64 + lane) * VEC);
        }
        // (2) LayerNorm prologue while the weights are in flight
        if (a.ln_g) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int v = 0; v < VEC; ++v) s += xs[c][v];
            const float mean = wave_sum(s) / (float)K;
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int v = 0; v < VEC; ++v) { float d = xs[c][v] - mean; q += d * d; }
            const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)K + a.ln_eps);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int k0 = (c * 64 + lane) * VEC;
#pragma unroll
                for (int v = 0; v < VEC; ++v) xs[c][v] = (xs[c][v] - mean) * rstd * a.ln_g[k0 + v] + a.ln_b[k0 + v];
            }
            if (a.xn_out && gwave == 0) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int k0 = (c * 64 + lane) * VEC;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) a.xn_out[k0 + v] = xs[c][v];
                }
            }
        }
        if (a.round_x) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int v = 0; v < VEC; ++v) xs[c][v] = round_bf16(xs[c][v]);
        }
        // (3) FMA
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float wf[VEC];
                unpack16<WT>(wv[r][c], wf);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[r] = fmaf(wf[v], xs[c][v], acc[r]);
            }
        }
    } else {
        float mean = 0.f, rstd = 1.f;
        const int nc = (K + CH - 1) / CH;
        if (a.ln_g) {
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
                    if (a.ln_g) {
                        t = (t - mean) * rstd * a.ln_g[k0 + v] + a.ln_b[k0 + v];
                        if (a.xn_out && gwave == 0) a.xn_out[k0 + v] = t;
                    }
                    xv[v] = a.round_x ? round_bf16(t) : t;
                }
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int row = min(row0 + r, a.N - 1);
                    u32x4 w = ld_stream16(W + (size_t)row * K + k0);
                    float wf[VEC];
                    unpack16<WT>(w, wf);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[r] = fmaf(wf[v], xv[v], acc[r]);
                }
            }
        }
    }

#pragma unroll
    for (int r = 0; r < RPW; ++r) acc[r] = wave_sum(acc[r]);

    // ---- epilogue: lane r finishes row r ----------------------------------------------------------------------
    if (a.epi == EPI_LMHEAD) {
        if (lane == 0) {
            const int skip = a.st->suppress_eos ? 1 : -1;     // eos = 1 (meshanything.py:103)
            float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int n = row0 + r;
                if (n < a.N) {
                    a.y[n] = acc[r];
                    if (n != skip && arg_better(acc[r], n, bv, bi)) { bv = acc[r]; bi = n; }
                }
            }
            a.part_val[gwave] = bv; a.part_idx[gwave] = bi;
        }
        return;
    }
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < RPW; ++r) if (lane == r) v = acc[r];
    const int n = row0 + lane;
    if (lane >= RPW || n >= a.N) return;
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
template <typename WT, int RPW>
inline hipError_t launch_gemv_rpw(const GemvArgs& a, hipStream_t s) {
    constexpr int VEC = WTraits<WT>::VEC;
    const int waves = (a.N + RPW - 1) / RPW;
    const int blocks = (waves + 3) / 4;
    const int K = a.K;
    // register fast paths: K = NC * 64 lanes * VEC elements (bf16: 1024 -> 2, 4096 -> 8; fp32: 1024 -> 4, 4096 -> 16)
    if (K == 2 * 64 * VEC)       hipLaunchKernelGGL((gemv_kernel<WT, RPW, 2>), dim3(blocks), dim3(256), 0, s, a);
    else if (K == 4 * 64 * VEC)  hipLaunchKernelGGL((gemv_kernel<WT, RPW, 4>), dim3(blocks), dim3(256), 0, s, a);
    else if (K == 8 * 64 * VEC && RPW <= 2)  hipLaunchKernelGGL((gemv_kernel<WT, (RPW <= 2 ? RPW : 1), 8>), dim3(blocks), dim3(256), 0, s, a);
    else if (K == 16 * 64 * VEC && RPW == 1) hipLaunchKernelGGL((gemv_kernel<WT, 1, 16>), dim3(blocks), dim3(256), 0, s, a);
    else                         hipLaunchKernelGGL((gemv_kernel<WT, RPW, 0>), dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

// number of per-wave partials an EPI_LMHEAD launch writes (must match launch_gemv's RPW choice)
inline int gemv_rpw_for(int N) { return N >= 6144 ? 4 : (N >= 2048 ? 2 : 1); }
inline int gemv_num_waves(int N) { int r = gemv_rpw_for(N); int w = (N + r - 1) / r; return ((w + 3) / 4) * 4; }

template <typename WT>
inline hipError_t launch_gemv(const GemvArgs& a, hipStream_t s) {
    if (a.K % WTraits<WT>::VEC != 0) return hipErrorInvalidValue;
    // rows per wave: keep >= ~1000 waves in flight (4+/CU) while amortising the x prologue
    const int rpw = gemv_rpw_for(a.N);
    if (rpw == 4) return launch_gemv_rpw<WT, 4>(a, s);
    if (rpw == 2) return launch_gemv_rpw<WT, 2>(a, s);
    return launch_gemv_rpw<WT, 1>(a, s);
}

}  // namespace ma
