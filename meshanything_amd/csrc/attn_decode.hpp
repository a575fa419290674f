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
// Geometry of one wave-load (shared by the launch-chain kernel below and the persistent decode kernel, persist.hpp): a lane
// holds EPL elements (16 bytes) of one position, LPP lanes share a position, a wave-load covers PPW positions, a wave owns
// 32 positions per round (U loads per operand).
template <typename KT> struct AttnGeom {
    static constexpr int EPL = 16 / sizeof(KT), LPP = 64 / EPL, PPW = 64 / LPP, U = 32 / PPW, NS = 4 * PPW;
};
template <typename KT> __device__ __forceinline__ void attn_unpack(const u32x4& r, float (&f)[16 / sizeof(KT)]) {
    if constexpr (sizeof(KT) == 4) {
        f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
    } else unpack8<KT>(r, f);
}
// online-softmax state of one position slot (the LPP lanes of a slot hold the same m, l and their own EPL dims of o)
template <typename KT> struct AttnSlotState { float m, l, o[16 / sizeof(KT)]; };

// one round: the U positions base + u * PPW of this lane's slot (positions >= end are masked), one rescale per round.
// OVR: position ovr_pos takes its K / V rows from (ovr_k, ovr_v) instead of the loaded registers (persistent kernel: the
// newest position has not reached the cache yet).
// FULL: the caller knows that every position of the round is below `end` (and, with OVR false, that none is the overridden one): the masks
// are dropped -- the same arithmetic on the same values, about a quarter fewer instructions (selects, compares) per round.
template <typename KT, bool OVR, bool FULL = false>
__device__ __forceinline__ void attn_round_reduce(AttnSlotState<KT>& st, const float (&qv)[16 / sizeof(KT)], const u32x4 (&kr)[AttnGeom<KT>::U],
                                                  const u32x4 (&vr)[AttnGeom<KT>::U], int base, int end, int ovr_pos, const u32x4& ovr_k,
                                                  const u32x4& ovr_v) {
    using G = AttnGeom<KT>;
    float d[G::U];
    float mr = st.m;
#pragma unroll
    for (int u = 0; u < G::U; ++u) {
        float kf[G::EPL];
        if constexpr (OVR) attn_unpack<KT>(base + u * G::PPW == ovr_pos ? ovr_k : kr[u], kf);
        else attn_unpack<KT>(kr[u], kf);
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < G::EPL; ++e) t = fmaf(qv[e], kf[e], t);
        t = group_sum<G::LPP>(t) * 0.125f;                 // 1/sqrt(64)
        d[u] = (FULL || base + u * G::PPW < end) ? t : -1e30f;
        mr = fmaxf(mr, d[u]);
    }
    const float alpha = expf(st.m - mr);                   // one rescale per round
    st.l *= alpha;
#pragma unroll
    for (int e = 0; e < G::EPL; ++e) st.o[e] *= alpha;
#pragma unroll
    for (int u = 0; u < G::U; ++u) {
        float vf[G::EPL];
        if constexpr (OVR) attn_unpack<KT>(base + u * G::PPW == ovr_pos ? ovr_v : vr[u], vf);
        else attn_unpack<KT>(vr[u], vf);
        const float pexp = (FULL || base + u * G::PPW < end) ? expf(d[u] - mr) : 0.f;
        st.l += pexp;
#pragma unroll
        for (int e = 0; e < G::EPL; ++e) st.o[e] = fmaf(pexp, vf[e], st.o[e]);
    }
    st.m = mr;
}

// LDS of the block-level merge: NS slot states, then one state per wave ("quarter")
template <typename KT> struct AttnMergeLds {
    float sm[AttnGeom<KT>::NS], sl[AttnGeom<KT>::NS], so[AttnGeom<KT>::NS][64];
    float qm[4], ql[4], qo[4][64];
};
// level 1: wave w folds the NS/4 slot states [w * NS/4, ...) (lane = dim).  Result: (M, L, O[lane]) of the wave's quarter.
template <typename KT>
__device__ __forceinline__ void attn_fold_quarter(const AttnMergeLds<KT>& S, int w, int lane, float& M, float& L, float& O) {
    constexpr int NQ = AttnGeom<KT>::NS / 4;
    M = -1e30f;
#pragma unroll
    for (int i = 0; i < NQ; ++i) M = fmaxf(M, S.sm[w * NQ + i]);
    L = 0.f; O = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const float f = expf(S.sm[w * NQ + i] - M);
        L = fmaf(S.sl[w * NQ + i], f, L);
        O = fmaf(S.so[w * NQ + i][lane], f, O);
    }
}
// level 2 (one wave): fold the four quarters
template <typename KT>
__device__ __forceinline__ void attn_fold_block(const AttnMergeLds<KT>& S, int lane, float& M, float& L, float& O) {
    M = fmaxf(fmaxf(S.qm[0], S.qm[1]), fmaxf(S.qm[2], S.qm[3]));
    L = 0.f; O = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float f = expf(S.qm[i] - M);
        L = fmaf(S.ql[i], f, L);
        O = fmaf(S.qo[i][lane], f, O);
    }
}

// ROWWAVE = false: the 4 waves of a block share one (row, head, chunk) and split its positions (batch 1: every CU works on
// the one sequence).  ROWWAVE = true (batches of >= 4 rows): each wave owns its own batch row's (head, chunk) -- a quarter
// of the blocks, no block-level merge, and short caches (a chunk of <= 32 positions) keep all four waves busy.
template <typename KT, bool ROWWAVE>
__global__ __launch_bounds__(256) void attn_decode_kernel(const float* __restrict__ q, const KT* __restrict__ kc,
                                                          const KT* __restrict__ vc, int max_seq, const DecState* st,
                                                          int len_override, int round_q, float* __restrict__ ws,
                                                          unsigned long long* trace, int q_stride, size_t kv_row_stride, int batch) {
    using G = AttnGeom<KT>;
    constexpr int EPL = G::EPL, LPP = G::LPP, PPW = G::PPW, U = G::U;
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
    AttnSlotState<KT> ss;
    ss.m = -1e30f; ss.l = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) ss.o[e] = 0.f;
    const u32x4 none = {0u, 0u, 0u, 0u};
    auto reduce = [&](int r, const u32x4 (&kr)[U], const u32x4 (&vr)[U]) {
        attn_round_reduce<KT, false>(ss, qv, kr, vr, start + (r << RSH) + woff + slot, end, -1, none, none);
    };
    if (round_q) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) qv[e] = H16<KT>::round(qv[e]);
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
    __shared__ AttnMergeLds<KT> S;
    const int gs = w * PPW + slot;
    if (dsub == 0) { S.sm[gs] = ss.m; S.sl[gs] = ss.l; }
#pragma unroll
    for (int e = 0; e < EPL; ++e) S.so[gs][dsub * EPL + e] = ss.o[e];
    __syncthreads();
    {
        float M, L, O;
        attn_fold_quarter<KT>(S, w, lane, M, L, O);
        if constexpr (ROWWAVE) {                            // this wave's partial is final: one (row, head, chunk) per wave
            if (row_ok) {
                float* ml = ws + ((size_t)h * ATTN_NCHUNK + c) * 2;
                float* op = ws + (size_t)H * ATTN_NCHUNK * 2 + ((size_t)h * ATTN_NCHUNK + c) * 64;
                if (lane == 0) { ml[0] = M; ml[1] = L; }
                op[lane] = O;
            }
            return;
        }
        if (lane == 0) { S.qm[w] = M; S.ql[w] = L; }
        S.qo[w][lane] = O;
    }
    __syncthreads();
    if (w == 0) {
        float M, L, O;
        attn_fold_block<KT>(S, lane, M, L, O);
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

// FINAL form for batches whose (row, head) pairs give the chip enough blocks: one block per (row, head), its NW waves split ALL
// cached positions of the head (rounds of NW x 32), the block-level merge finishes the softmax and the normalised output goes
// straight to the out_proj GEMM's bf16 activation buffer -- no partials in HBM, no merge launch ([3p] flash_attn_func with
// q_len 1, one call per layer).  NW = 4 from 12 rows on (>= 192 blocks); 8 <= rows < 12 give 128-176 blocks, which stream
// with NW = 8 waves each; below 8 rows the split form wins (too few blocks).  Same per-slot arithmetic
// (attn_round_reduce) as the split form; only the grouping of positions into partial states differs.
// SPLIT2 (8..11 rows: 128-176 (row, head) pairs leave half of the CUs without a block): TWO blocks per pair, each over every other
// round; block z = 0 hands its (m, l, o[64]) to block z = 1 inside the launch as 66 tagged 8-byte granules (the protocol of the
// batch-1 chain, qkv_attn.hpp: epoch = position * 32 + layer + 1, bounded sweep, error word), which merges the two states in a fixed
// order and writes the output.  The publishing blocks are dispatched first and never wait.
constexpr int ATTN_PAIR_GRANULES = 72;                   // 64 x o + m + l, padded
constexpr unsigned ATTN_ERR_PAIR = 64;
template <typename KT, int NW, bool SPLIT2>
__global__ __launch_bounds__(NW * 64) void attn_decode_final_kernel(const float* __restrict__ q, const KT* __restrict__ kc, const KT* __restrict__ vc,
                                                                    int max_seq, const DecState* st, int len_override, int round_q,
                                                                    bf16_t* __restrict__ out, int out_stride, int q_stride, size_t kv_row_stride,
                                                                    unsigned long long* pair_gran, unsigned* err, int layer) {
    using G = AttnGeom<KT>;
    constexpr int EPL = G::EPL, LPP = G::LPP, PPW = G::PPW, U = G::U;
    constexpr int RPOS = NW * 32;                           // positions per round of the block
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int h = blockIdx.x, brow = blockIdx.y;
    const int slot = lane / LPP, dsub = lane % LPP;
    q += (size_t)brow * q_stride;
    kc += (size_t)brow * kv_row_stride;
    vc += (size_t)brow * kv_row_stride;
    float qv[EPL];
    {
        const float* qp = q + h * 64 + dsub * EPL;
#pragma unroll
        for (int e = 0; e < EPL; e += 4) {
            f32x4 t = *reinterpret_cast<const f32x4*>(qp + e);
            qv[e] = t.x; qv[e + 1] = t.y; qv[e + 2] = t.z; qv[e + 3] = t.w;
        }
    }
    // SPLIT2: the rounds alternate between the two blocks -- the merging block (z = 1) takes the even rounds (ceil(n / 2) of them), the
    // publisher (z = 0) the odd ones -- so each block's FIRST round is known before the row's length is: its keys and values are
    // requested together with q, ahead of the (dependent) read of the write position.  Positions past the end are masked by the
    // reduction and zeroed before it (below); the address is clamped to the plane.
    const int g0 = SPLIT2 ? (blockIdx.z == 0 ? 1 : 0) : 0;  // first round of this block; round r of the block = round g0 + GS r of the row
    constexpr int GS = SPLIT2 ? 2 : 1;
    const KT* kh = kc + (size_t)h * max_seq * 64 + dsub * EPL;
    const KT* vh = vc + (size_t)h * max_seq * 64 + dsub * EPL;
    u32x4 kA[U], vA[U], kB[U], vB[U];
    {
        const int base = g0 * RPOS + w * 32 + slot;
#pragma unroll
        for (int u = 0; u < U; ++u) kA[u] = ld_stream16(kh + (size_t)min(base + u * PPW, max_seq - 1) * 64);
#pragma unroll
        for (int u = 0; u < U; ++u) vA[u] = ld_stream16(vh + (size_t)min(base + u * PPW, max_seq - 1) * 64);
    }
    const int end = len_override >= 0 ? len_override : st[brow].pos + 1;
    const int nr_all = (max(end, 0) + RPOS - 1) / RPOS;
    const int nround = SPLIT2 ? (blockIdx.z == 0 ? nr_all / 2 : (nr_all + 1) / 2) : nr_all;
    {   // the early round's slots past the end: whatever the plane held there (a caller's buffer need not be finite) becomes zero --
        // the reduction masks the score, but 0 x NaN in the value sum would not be 0
        const u32x4 zero = {0u, 0u, 0u, 0u};
        const int base = g0 * RPOS + w * 32 + slot;
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + u * PPW >= end) { kA[u] = zero; vA[u] = zero; }
    }
    // (a slot past the end asks for the position of slot 0 of its own wave-load -- the eight positions of a wave-load are one request each, so the
    //  dead slot's lanes fold into a live request -- and only if that is dead too for position 0: these are non-temporal loads, a line asked for
    //  again is fetched again; the last, partial round of every (row, head) was up to a round of such requests, rows_attn.hpp)
    auto issue = [&](int r, u32x4 (&kr)[U], u32x4 (&vr)[U]) {
        const int base = (g0 + GS * r) * RPOS + w * 32 + slot;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = base + u * PPW, p0 = p - slot;
            kr[u] = ld_stream16(kh + (size_t)(p < end ? p : p0 < end ? p0 : 0) * 64);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = base + u * PPW, p0 = p - slot;
            vr[u] = ld_stream16(vh + (size_t)(p < end ? p : p0 < end ? p0 : 0) * 64);
        }
    };
    AttnSlotState<KT> ss;
    ss.m = -1e30f; ss.l = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) ss.o[e] = 0.f;
    const u32x4 none = {0u, 0u, 0u, 0u};
    auto reduce = [&](int r, const u32x4 (&kr)[U], const u32x4 (&vr)[U]) {
        attn_round_reduce<KT, false>(ss, qv, kr, vr, (g0 + GS * r) * RPOS + w * 32 + slot, end, -1, none, none);
    };
    if (round_q) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) qv[e] = H16<KT>::round(qv[e]);
    }
    for (int r = 0; r < nround; r += 2) {                  // (round 0 is already on its way)
        if (r + 1 < nround) issue(r + 1, kB, vB);
        reduce(r, kA, vA);
        if (r + 1 < nround) {
            if (r + 2 < nround) issue(r + 2, kA, vA);
            reduce(r + 1, kB, vB);
        }
    }
    // merge: wave w folds its own PPW slot states (lane = dim), then wave 0 folds the NW wave states (for NW = 4 this is
    // attn_fold_quarter / attn_fold_block of the split form)
    __shared__ float sm[NW * PPW], sl[NW * PPW], so[NW * PPW][64];
    __shared__ float wm_[NW], wl_[NW], wo_[NW][64];
    const int gs = w * PPW + slot;
    if (dsub == 0) { sm[gs] = ss.m; sl[gs] = ss.l; }
#pragma unroll
    for (int e = 0; e < EPL; ++e) so[gs][dsub * EPL + e] = ss.o[e];
    __syncthreads();
    {
        float M = -1e30f;
#pragma unroll
        for (int i = 0; i < PPW; ++i) M = fmaxf(M, sm[w * PPW + i]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const float f = expf(sm[w * PPW + i] - M);
            L = fmaf(sl[w * PPW + i], f, L);
            O = fmaf(so[w * PPW + i][lane], f, O);
        }
        if (lane == 0) { wm_[w] = M; wl_[w] = L; }
        wo_[w][lane] = O;
    }
    __syncthreads();
    if (w == 0) {
        float M = wm_[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) M = fmaxf(M, wm_[i]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const float f = expf(wm_[i] - M);
            L = fmaf(wl_[i], f, L);
            O = fmaf(wo_[i][lane], f, O);
        }
        if constexpr (SPLIT2) {
            typedef __attribute__((address_space(1))) unsigned long long gull;
            gull* g = (gull*)(pair_gran + ((size_t)brow * gridDim.x + h) * ATTN_PAIR_GRANULES);
            const unsigned epoch = (unsigned)(end - 1) * 32u + (unsigned)layer + 1u;
            if (blockIdx.z == 0) {                           // publisher: 64 x o, then m and l
                __hip_atomic_store(g + lane, ((unsigned long long)epoch << 32) | __float_as_uint(O), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane < 2) __hip_atomic_store(g + 64 + lane, ((unsigned long long)epoch << 32) | __float_as_uint(lane == 0 ? M : L), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            unsigned long long vo = 0, vm = (unsigned long long)epoch << 32;
            u64 t0 = __builtin_amdgcn_s_memrealtime();
            unsigned spins = 0;
            for (;;) {
                vo = __hip_atomic_load(g + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane < 2) vm = __hip_atomic_load(g + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((unsigned)(vo >> 32) == epoch && (unsigned)(vm >> 32) == epoch)) break;
                __builtin_amdgcn_s_sleep(1);
                if (xchg_expired(spins, t0, err)) {
                    if (lane == 0) xchg_raise(err, ATTN_ERR_PAIR, spins);
                    vo = 0; vm = 0;
                    break;
                }
            }
            const float O2 = __uint_as_float((unsigned)vo);
            const float M2 = __shfl(__uint_as_float((unsigned)vm), 0, 64), L2 = __shfl(__uint_as_float((unsigned)vm), 1, 64);
            const float Mx = fmaxf(M, M2), f1 = expf(M - Mx), f2 = expf(M2 - Mx);      // this block's rounds first, then the partner's
            L = fmaf(L2, f2, L * f1);
            O = fmaf(O2, f2, O * f1);
        }
        out[(size_t)brow * out_stride + h * 64 + lane] = H16<KT>::bits(O * (1.0f / L));     // position 0 always exists: L > 0
    }
}

template <typename KT>
inline hipError_t launch_attn_decode_final(const float* q, const void* kc, const void* vc, int H, int max_seq, const DecState* st, int len_override,
                                           int round_q, bf16_t* out, int out_stride, hipStream_t s, int batch, int q_stride, size_t kv_row_stride,
                                           int waves = 0, unsigned long long* pair_gran = nullptr, unsigned* err = nullptr, int layer = 0) {
    if (waves == 0) waves = batch >= 12 ? 4 : 8;          // measured: profiles/r02_ab_batched_attention_forms.txt
    const KT* k = reinterpret_cast<const KT*>(kc); const KT* v = reinterpret_cast<const KT*>(vc);
    if (pair_gran) {                                       // two blocks per (row, head), 8 waves each
        if (!err || waves != 8) return hipErrorInvalidValue;
        hipLaunchKernelGGL((attn_decode_final_kernel<KT, 8, true>), dim3(H, batch, 2), dim3(512), 0, s, q, k, v, max_seq, st, len_override, round_q, out, out_stride, q_stride, kv_row_stride, pair_gran, err, layer);
        return hipGetLastError();
    }
    unsigned long long* ng = nullptr; unsigned* ne = nullptr;
    if (waves == 4) hipLaunchKernelGGL((attn_decode_final_kernel<KT, 4, false>), dim3(H, batch), dim3(256), 0, s, q, k, v, max_seq, st, len_override, round_q, out, out_stride, q_stride, kv_row_stride, ng, ne, 0);
    else if (waves == 8) hipLaunchKernelGGL((attn_decode_final_kernel<KT, 8, false>), dim3(H, batch), dim3(512), 0, s, q, k, v, max_seq, st, len_override, round_q, out, out_stride, q_stride, kv_row_stride, ng, ne, 0);
    else if (waves == 16) hipLaunchKernelGGL((attn_decode_final_kernel<KT, 16, false>), dim3(H, batch), dim3(1024), 0, s, q, k, v, max_seq, st, len_override, round_q, out, out_stride, q_stride, kv_row_stride, ng, ne, 0);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace ma
