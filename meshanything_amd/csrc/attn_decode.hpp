// Single-query (decode-step) attention over the in-place KV cache, split-KV ("flash-decoding") form.
//
// Replaces [3p] OptFlashAttention2 -> flash_attn_func with q_len 1 and the per-step torch.cat KV growth
// (SURVEY.md 2.1).  Cache layout per layer: K and V each (heads, max_seq, 64) of KT, so one head's positions are a
// dense (len x 128 B | 256 B) stream.  grid = (splits, heads): split s owns a contiguous range of positions; inside a
// block each wave reads 1 KiB per instruction (PPW positions x 64 dims), LPP lanes share one position, and every
// lane group runs its own online softmax, merged once at the end (LDS) -> partial (m, l, o[64]) per (split, head).
// A second tiny kernel merges the splits.  HBM-bound: algorithmic bytes = 2 * len * 64 * sizeof(KT) per head.
#pragma once
#include "common.hpp"
#include "state.hpp"

namespace ma {

constexpr int ATTN_PART_STRIDE = 66;   // m, l, o[64]

template <typename KT>
__global__ __launch_bounds__(256) void attn_decode_kernel(const float* __restrict__ q, const KT* __restrict__ kc,
                                                          const KT* __restrict__ vc, int max_seq, const DecState* st,
                                                          int len_override, int round_q, float* __restrict__ part) {
    constexpr int EPL = 16 / sizeof(KT);     // elements per lane per 16-byte load
    constexpr int LPP = 64 / EPL;            // lanes per position
    constexpr int PPW = 64 / LPP;            // positions per wave-load
    constexpr int U = 4;                     // position groups in flight per wave
    const int s = blockIdx.x, S = gridDim.x, h = blockIdx.y, H = gridDim.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int len = len_override >= 0 ? len_override : st->pos + 1;
    const int chunk = (len + S - 1) / S;
    const int start = s * chunk;
    const int end = min(len, start + chunk);
    const int slot = lane / LPP, dsub = lane % LPP;

    float qv[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        float t = q[h * 64 + dsub * EPL + e];
        qv[e] = round_q ? round_bf16(t) : t;
    }
    float m = -1e30f, l = 0.f, o[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] = 0.f;

    const KT* kh = kc + (size_t)h * max_seq * 64;
    const KT* vh = vc + (size_t)h * max_seq * 64;
    for (int base = start + w * PPW; base < end; base += 4 * PPW * U) {
        u32x4 kr[U], vr[U];
        bool valid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = base + u * 4 * PPW + slot;
            valid[u] = p < end;
            const size_t off = (size_t)(valid[u] ? p : start) * 64 + dsub * EPL;
            kr[u] = ld_stream16(kh + off);
            vr[u] = ld_stream16(vh + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float kf[EPL], vf[EPL];
            if constexpr (sizeof(KT) == 4) {
                kf[0] = __uint_as_float(kr[u].x); kf[1] = __uint_as_float(kr[u].y); kf[2] = __uint_as_float(kr[u].z); kf[3] = __uint_as_float(kr[u].w);
                vf[0] = __uint_as_float(vr[u].x); vf[1] = __uint_as_float(vr[u].y); vf[2] = __uint_as_float(vr[u].z); vf[3] = __uint_as_float(vr[u].w);
            } else {
                kf[0] = bf_lo(kr[u].x); kf[1] = bf_hi(kr[u].x); kf[2] = bf_lo(kr[u].y); kf[3] = bf_hi(kr[u].y);
                kf[4] = bf_lo(kr[u].z); kf[5] = bf_hi(kr[u].z); kf[6] = bf_lo(kr[u].w); kf[7] = bf_hi(kr[u].w);
                vf[0] = bf_lo(vr[u].x); vf[1] = bf_hi(vr[u].x); vf[2] = bf_lo(vr[u].y); vf[3] = bf_hi(vr[u].y);
                vf[4] = bf_lo(vr[u].z); vf[5] = bf_hi(vr[u].z); vf[6] = bf_lo(vr[u].w); vf[7] = bf_hi(vr[u].w);
            }
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) d = fmaf(qv[e], kf[e], d);
            d = group_sum<LPP>(d) * 0.125f;                 // 1/sqrt(64)
            if (valid[u]) {
                const float mn = fmaxf(m, d);
                const float alpha = expf(m - mn);
                const float pexp = expf(d - mn);
                l = l * alpha + pexp;
#pragma unroll
                for (int e = 0; e < EPL; ++e) o[e] = fmaf(pexp, vf[e], o[e] * alpha);
                m = mn;
            }
        }
    }

    // merge the 4 * PPW per-slot states of this block
    __shared__ float sm[4 * PPW], sl[4 * PPW], so[4 * PPW][64];
    const int gs = w * PPW + slot;
    if (dsub == 0) { sm[gs] = m; sl[gs] = l; }
#pragma unroll
    for (int e = 0; e < EPL; ++e) so[gs][dsub * EPL + e] = o[e];
    __syncthreads();
    if (w == 0) {
        float M = -1e30f;
#pragma unroll
        for (int i = 0; i < 4 * PPW; ++i) M = fmaxf(M, sm[i]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int i = 0; i < 4 * PPW; ++i) {
            const float f = expf(sm[i] - M);
            L = fmaf(sl[i], f, L);
            O = fmaf(so[i][lane], f, O);
        }
        float* pp = part + ((size_t)s * H + h) * ATTN_PART_STRIDE;
        if (lane == 0) { pp[0] = M; pp[1] = L; }
        pp[2 + lane] = O;
    }
}

// merge the splits: out[h*64+d] = sum_s O_s e^(M_s-M) / sum_s L_s e^(M_s-M).  grid = (H), block = 64.
// All S partial records are loaded before anything is used: one memory round trip instead of S dependent ones
// (the partials were written by other CUs, i.e. they come from L2 / Infinity Cache, ~1 us each).
constexpr int ATTN_MAX_SPLITS = 64;
__global__ __launch_bounds__(64) void attn_combine_kernel(const float* __restrict__ part, int S, float* __restrict__ out) {
    const int h = blockIdx.x, H = gridDim.x, lane = threadIdx.x;
    float ms[ATTN_MAX_SPLITS / 1], ls[ATTN_MAX_SPLITS], os[ATTN_MAX_SPLITS];
#pragma unroll
    for (int s = 0; s < ATTN_MAX_SPLITS; ++s) {
        const bool ok = s < S;
        const float* pp = part + ((size_t)(ok ? s : 0) * H + h) * ATTN_PART_STRIDE;
        ms[s] = ok ? pp[0] : -1e30f;
        ls[s] = ok ? pp[1] : 0.f;
        os[s] = ok ? pp[2 + lane] : 0.f;
    }
    float M = -1e30f;
#pragma unroll
    for (int s = 0; s < ATTN_MAX_SPLITS; ++s) M = fmaxf(M, ms[s]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int s = 0; s < ATTN_MAX_SPLITS; ++s) {
        const float f = expf(ms[s] - M);
        L = fmaf(ls[s], f, L);
        O = fmaf(os[s], f, O);
    }
    out[h * 64 + lane] = O / L;
}

// fill the KV cache from prefill projections: src (rows, ld) fp32 with K at column koff + h*64 + d, V at voff + ...
template <typename KT>
__global__ void kv_fill_kernel(const float* __restrict__ src, int ld, int koff, int voff, int rows, int H, int max_seq,
                               KT* __restrict__ kc, KT* __restrict__ vc) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = rows * H * 64;
    if (idx >= total) return;
    const int d = idx & 63, h = (idx >> 6) % H, r = idx / (64 * H);
    const size_t dst = ((size_t)h * max_seq + r) * 64 + d;
    const float k = src[(size_t)r * ld + koff + h * 64 + d];
    const float v = src[(size_t)r * ld + voff + h * 64 + d];
    if constexpr (sizeof(KT) == 4) { kc[dst] = k; vc[dst] = v; }
    else { kc[dst] = f2bf(k); vc[dst] = f2bf(v); }
}

}  // namespace ma
