// Single-query (decode-step) attention over the in-place KV cache, split-KV ("flash-decoding") form.
//
// Replaces [3p] OptFlashAttention2 -> flash_attn_func with q_len 1 and the per-step torch.cat KV growth
// (SURVEY.md 2.1).  Cache layout per layer: K and V each (heads, max_seq, 64) of KT, so one head's positions are a
// dense (len x 128 B | 256 B) stream.  grid = (ATTN_NCHUNK, heads) = 256 blocks for the 350M shape, one per CU: the
// cached positions of a head are cut into ATTN_NCHUNK equal chunks whatever the length, so the grid, and therefore the
// captured graph, never changes.  Inside a block each wave owns 32 positions per round of 128, reads 1 KiB per
// instruction (PPW positions x 64 dims, LPP lanes per position); the loads of round r+1 are in flight while round r is
// reduced; scores go through a 3/4-step DPP reduction; the 4 x PPW online-softmax states of the block meet once in LDS
// -> one partial (m, l, o[64]) per (head, chunk).
// There is NO merge kernel: the consumer of the attention output is the out_proj GEMV, and its prologue merges the
// ATTN_NCHUNK partials of every head while its weight loads are in flight (gemv.hpp, PRO_ATTN) -- one launch boundary
// less per layer and no cross-block exchange inside a launch.
// HBM-bound: algorithmic bytes = 2 * len * 64 * sizeof(KT) per head.
#pragma once
#include "common.hpp"
#include "state.hpp"

namespace ma {

constexpr int ATTN_NCHUNK = 16;        // split factor (partials per head)

// workspace: ML[heads][NCHUNK][2] (m, l) followed by O[heads][NCHUNK][64]
__host__ __device__ inline size_t attn_workspace_floats(int heads) { return (size_t)heads * ATTN_NCHUNK * (2 + 64); }

// ---- consumer side: merge the partials of head h for the four dims d0..d0+3 ----------------------------------------
__device__ inline void attn_partials_load(const float* ws, int H, int h, int d0, f32x4 (&pml)[ATTN_NCHUNK / 2], f32x4 (&po)[ATTN_NCHUNK]) {
    const float* ml = ws + (size_t)h * ATTN_NCHUNK * 2;
    const float* o = ws + (size_t)H * ATTN_NCHUNK * 2 + (size_t)h * ATTN_NCHUNK * 64 + d0;
#pragma unroll
    for (int i = 0; i < ATTN_NCHUNK / 2; ++i) pml[i] = *reinterpret_cast<const f32x4*>(ml + 4 * i);
#pragma unroll
    for (int c = 0; c < ATTN_NCHUNK; ++c) po[c] = *reinterpret_cast<const f32x4*>(o + c * 64);
}
__device__ inline f32x4 attn_partials_merge(const f32x4 (&pml)[ATTN_NCHUNK / 2], const f32x4 (&po)[ATTN_NCHUNK]) {
    float m[ATTN_NCHUNK], l[ATTN_NCHUNK];
#pragma unroll
    for (int i = 0; i < ATTN_NCHUNK / 2; ++i) { m[2 * i] = pml[i].x; l[2 * i] = pml[i].y; m[2 * i + 1] = pml[i].z; l[2 * i + 1] = pml[i].w; }
    float M = m[0];
#pragma unroll
    for (int c = 1; c < ATTN_NCHUNK; ++c) M = fmaxf(M, m[c]);
    float L = 0.f;
    f32x4 O = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < ATTN_NCHUNK; ++c) {
        const float f = expf(m[c] - M);
        L = fmaf(l[c], f, L);
        O.x = fmaf(po[c].x, f, O.x); O.y = fmaf(po[c].y, f, O.y); O.z = fmaf(po[c].z, f, O.z); O.w = fmaf(po[c].w, f, O.w);
    }
    const float inv = 1.0f / L;
    return f32x4{O.x * inv, O.y * inv, O.z * inv, O.w * inv};
}
// one element (generic / tiny shapes): same arithmetic, same order
__device__ inline float attn_partials_merge_one(const float* ws, int H, int h, int d) {
    const float* ml = ws + (size_t)h * ATTN_NCHUNK * 2;
    const float* o = ws + (size_t)H * ATTN_NCHUNK * 2 + (size_t)h * ATTN_NCHUNK * 64 + d;
    float M = ml[0];
    for (int c = 1; c < ATTN_NCHUNK; ++c) M = fmaxf(M, ml[2 * c]);
    float L = 0.f, O = 0.f;
    for (int c = 0; c < ATTN_NCHUNK; ++c) {
        const float f = expf(ml[2 * c] - M);
        L = fmaf(ml[2 * c + 1], f, L);
        O = fmaf(o[c * 64], f, O);
    }
    return O * (1.0f / L);
}
// standalone merge (kernel-level entry point ma_op_decode_attention and its test): out[h*64 + d]
__global__ __launch_bounds__(256) void attn_merge_kernel(const float* __restrict__ ws, int H, float* __restrict__ out) {
    const int k = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (k >= H * 64) return;
    f32x4 pml[ATTN_NCHUNK / 2], po[ATTN_NCHUNK];
    attn_partials_load(ws, H, k >> 6, k & 63, pml, po);
    *reinterpret_cast<f32x4*>(out + k) = attn_partials_merge(pml, po);
}

// ---- producer side ---------------------------------------------------------------------------------------------------
// ROWWAVE = false: the 4 waves of a block share one (row, head, chunk) and split its positions (batch 1: every CU works on
// the one sequence).  ROWWAVE = true (batches of >= 4 rows): each wave owns its own batch row's (head, chunk) -- a quarter
// of the blocks, no block-level merge, and short caches (a chunk of <= 32 positions) keep all four waves busy.
template <typename KT, bool ROWWAVE>
__global__ __launch_bounds__(256) void attn_decode_kernel(const float* __restrict__ q, const KT* __restrict__ kc,
                                                          const KT* __restrict__ vc, int max_seq, const DecState* st,
                                                          int len_override, int round_q, float* __restrict__ ws,
                                                          unsigned long long* trace, int q_stride, size_t kv_row_stride, int batch) {
    constexpr int EPL = 16 / sizeof(KT);     // elements per lane per 16-byte load
    constexpr int LPP = 64 / EPL;            // lanes per position
    constexpr int PPW = 64 / LPP;            // positions per wave-load
    constexpr int U = 32 / PPW;              // loads per lane per operand and round: 32 positions per wave
    constexpr int NS = 4 * PPW;              // softmax states per block
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x, h = blockIdx.y, H = gridDim.y;
    const int brow_raw = ROWWAVE ? blockIdx.z * 4 + w : blockIdx.z;
    const bool row_ok = brow_raw < batch;
    const int brow = row_ok ? brow_raw : batch - 1;       // surplus waves shadow the last row and store nothing
    constexpr int RSH = ROWWAVE ? 5 : 7;                  // positions per round: 32 per wave | 128 per block
    const int woff = ROWWAVE ? 0 : w * 32;
    const int slot = lane / LPP, dsub = lane % LPP;
    if (trace && threadIdx.x == 0) trace[(blockIdx.y * ATTN_NCHUNK + blockIdx.x) * 4 + 0] = __builtin_amdgcn_s_memrealtime();
    // batch row (grid.z): its query, its cache planes, its partials, its state
    q += (size_t)brow * q_stride;
    kc += (size_t)brow * kv_row_stride;
    vc += (size_t)brow * kv_row_stride;
    ws += (size_t)brow * attn_workspace_floats(H);

    // q does not depend on the length: its load goes out together with the state's
    float qv[EPL];
    {
        const float* qp = q + h * 64 + dsub * EPL;
#pragma unroll
        for (int e = 0; e < EPL; e += 4) {
            f32x4 t = *reinterpret_cast<const f32x4*>(qp + e);
            qv[e] = t.x; qv[e + 1] = t.y; qv[e + 2] = t.z; qv[e + 3] = t.w;
        }
    }
    const int len = len_override >= 0 ? len_override : st[brow].pos + 1;
    const int per = (len + ATTN_NCHUNK - 1) / ATTN_NCHUNK;
    const int start = c * per;
    const int end = min(len, start + per);
    const int nround = (max(end - start, 0) + (1 << RSH) - 1) >> RSH;

    const KT* kh = kc + (size_t)h * max_seq * 64 + dsub * EPL;
    const KT* vh = vc + (size_t)h * max_seq * 64 + dsub * EPL;
    u32x4 kA[U], vA[U], kB[U], vB[U];
    auto issue = [&](int r, u32x4 (&kr)[U], u32x4 (&vr)[U]) {
        const int base = start + (r << RSH) + woff + slot;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = base + u * PPW;
            kr[u] = ld_stream16(kh + (size_t)(p < end ? p : start) * 64);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = base + u * PPW;
            vr[u] = ld_stream16(vh + (size_t)(p < end ? p : start) * 64);
        }
    };
    float m = -1e30f, l = 0.f, o[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] = 0.f;
    auto reduce = [&](int r, const u32x4 (&kr)[U], const u32x4 (&vr)[U]) {
        const int base = start + (r << RSH) + woff + slot;
        float d[U];
        float mr = m;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float kf[EPL];
            if constexpr (sizeof(KT) == 4) {
                kf[0] = __uint_as_float(kr[u].x); kf[1] = __uint_as_float(kr[u].y); kf[2] = __uint_as_float(kr[u].z); kf[3] = __uint_as_float(kr[u].w);
            } else {
                kf[0] = bf_lo(kr[u].x); kf[1] = bf_hi(kr[u].x); kf[2] = bf_lo(kr[u].y); kf[3] = bf_hi(kr[u].y);
                kf[4] = bf_lo(kr[u].z); kf[5] = bf_hi(kr[u].z); kf[6] = bf_lo(kr[u].w); kf[7] = bf_hi(kr[u].w);
            }
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) t = fmaf(qv[e], kf[e], t);
            t = group_sum<LPP>(t) * 0.125f;                 // 1/sqrt(64)
            d[u] = (base + u * PPW < end) ? t : -1e30f;
            mr = fmaxf(mr, d[u]);
        }
        const float alpha = expf(m - mr);                   // one rescale per round
        l *= alpha;
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] *= alpha;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float vf[EPL];
            if constexpr (sizeof(KT) == 4) {
                vf[0] = __uint_as_float(vr[u].x); vf[1] = __uint_as_float(vr[u].y); vf[2] = __uint_as_float(vr[u].z); vf[3] = __uint_as_float(vr[u].w);
            } else {
                vf[0] = bf_lo(vr[u].x); vf[1] = bf_hi(vr[u].x); vf[2] = bf_lo(vr[u].y); vf[3] = bf_hi(vr[u].y);
                vf[4] = bf_lo(vr[u].z); vf[5] = bf_hi(vr[u].z); vf[6] = bf_lo(vr[u].w); vf[7] = bf_hi(vr[u].w);
            }
            const float pexp = (base + u * PPW < end) ? expf(d[u] - mr) : 0.f;
            l += pexp;
#pragma unroll
            for (int e = 0; e < EPL; ++e) o[e] = fmaf(pexp, vf[e], o[e]);
        }
        m = mr;
    };
    if (round_q) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) qv[e] = round_bf16(qv[e]);
    }
    if (nround > 0) issue(0, kA, vA);
    for (int r = 0; r < nround; r += 2) {
        if (r + 1 < nround) issue(r + 1, kB, vB);
        reduce(r, kA, vA);
        if (r + 1 < nround) {
            if (r + 2 < nround) issue(r + 2, kA, vA);
            reduce(r + 1, kB, vB);
        }
    }

    if (trace && threadIdx.x == 0) trace[(blockIdx.y * ATTN_NCHUNK + blockIdx.x) * 4 + 2] = __builtin_amdgcn_s_memrealtime();
    // merge the NS per-slot states of this block -> one partial per (head, chunk); an empty chunk publishes (m=-1e30, l=0, o=0).
    // Two levels, all 256 threads: thread (quarter qd, dim) folds NS/4 states, then wave 0 folds the four quarters.
    __shared__ float sm[NS], sl[NS], so[NS][64];
    __shared__ float qm[4], ql[4], qo[4][64];
    const int gs = w * PPW + slot;
    if (dsub == 0) { sm[gs] = m; sl[gs] = l; }
#pragma unroll
    for (int e = 0; e < EPL; ++e) so[gs][dsub * EPL + e] = o[e];
    __syncthreads();
    {
        constexpr int NQ = NS / 4;
        float M = -1e30f;
#pragma unroll
        for (int i = 0; i < NQ; ++i) M = fmaxf(M, sm[w * NQ + i]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const float f = expf(sm[w * NQ + i] - M);
            L = fmaf(sl[w * NQ + i], f, L);
            O = fmaf(so[w * NQ + i][lane], f, O);
        }
        if constexpr (ROWWAVE) {                            // this wave's partial is final: one (row, head, chunk) per wave
            if (row_ok) {
                float* ml = ws + ((size_t)h * ATTN_NCHUNK + c) * 2;
                float* op = ws + (size_t)H * ATTN_NCHUNK * 2 + ((size_t)h * ATTN_NCHUNK + c) * 64;
                if (lane == 0) { ml[0] = M; ml[1] = L; }
                op[lane] = O;
            }
            return;
        }
        if (lane == 0) { qm[w] = M; ql[w] = L; }
        qo[w][lane] = O;
    }
    __syncthreads();
    if (w == 0) {
        const float M = fmaxf(fmaxf(qm[0], qm[1]), fmaxf(qm[2], qm[3]));
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float f = expf(qm[i] - M);
            L = fmaf(ql[i], f, L);
            O = fmaf(qo[i][lane], f, O);
        }
        float* ml = ws + ((size_t)h * ATTN_NCHUNK + c) * 2;
        float* op = ws + (size_t)H * ATTN_NCHUNK * 2 + ((size_t)h * ATTN_NCHUNK + c) * 64;
        if (lane == 0) { ml[0] = M; ml[1] = L; }
        op[lane] = O;
        if (trace && threadIdx.x == 0) trace[(blockIdx.y * ATTN_NCHUNK + blockIdx.x) * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

template <typename KT>
inline hipError_t launch_attn_decode(const float* q, const void* kc, const void* vc, int H, int max_seq, const DecState* st, int len_override,
                                     int round_q, float* workspace, hipStream_t s, unsigned long long* trace = nullptr, int batch = 1,
                                     int q_stride = 0, size_t kv_row_stride = 0, bool rowwave = false) {
    // rowwave: only with the batched MFMA decode path (the row-parallel GEMV path keeps every row's arithmetic identical
    // to a batch-1 run, including the attention's merge order)
    if (rowwave && batch >= 4)
        hipLaunchKernelGGL((attn_decode_kernel<KT, true>), dim3(ATTN_NCHUNK, H, (batch + 3) / 4), dim3(256), 0, s, q, reinterpret_cast<const KT*>(kc),
                           reinterpret_cast<const KT*>(vc), max_seq, st, len_override, round_q, workspace, trace, q_stride, kv_row_stride, batch);
    else
        hipLaunchKernelGGL((attn_decode_kernel<KT, false>), dim3(ATTN_NCHUNK, H, batch), dim3(256), 0, s, q, reinterpret_cast<const KT*>(kc),
                           reinterpret_cast<const KT*>(vc), max_seq, st, len_override, round_q, workspace, trace, q_stride, kv_row_stride, batch);
    return hipGetLastError();
}

// fill the KV cache from prefill projections: src (B * rows, ld) fp32 with K at column koff + h*64 + d, V at voff + ...;
// grid.y = batch row b: its `rows` source rows start at b * rows, its planes at b * kv_row_stride elements
template <typename KT>
__global__ void kv_fill_kernel(const float* __restrict__ src, int ld, int koff, int voff, int rows, int H, int max_seq,
                               KT* __restrict__ kc, KT* __restrict__ vc, size_t kv_row_stride) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = rows * H * 64;
    if (idx >= total) return;
    const int b = blockIdx.y;
    const int d = idx & 63, h = (idx >> 6) % H, r = idx / (64 * H);
    const size_t dst = (size_t)b * kv_row_stride + ((size_t)h * max_seq + r) * 64 + d;
    const float* sp = src + ((size_t)b * rows + r) * ld;
    const float k = sp[koff + h * 64 + d];
    const float v = sp[voff + h * 64 + d];
    if constexpr (sizeof(KT) == 4) { kc[dst] = k; vc[dst] = v; }
    else { kc[dst] = f2bf(k); vc[dst] = f2bf(v); }
}

}  // namespace ma
