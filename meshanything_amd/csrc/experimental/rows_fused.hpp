// The two-launch decoder layer of the batch-1 chain (qkv_attn.hpp, oproj_fc1.hpp) for SMALL BATCHES: 2 .. 8 rows that step together
// (BASELINE.json's "batch = 8 x N shapes": the per-rank workload of `bench.py --gpus N`, N > 1), bf16 policy, hidden 1024, ffn 4096.
//
// Why: the matrix-core decode path (gemm_decode.hpp) needs 4-5 dependent launches per layer; at 8 rows each streams 2-8 MB of weights
// in 7-9 us, i.e. the weight stream runs at 0.9 TB/s -- launch latency, not bandwidth (profiles/r02_bench_batch8_kernel_stats.csv).
// Putting the rows into the grid of the batch-1 fused launches does not work either: 256 x B blocks are not co-resident.  Here the
// GRID stays 256 blocks and every block LOOPS over the rows: its weight rows (24 KB in the first launch, 72 KB in the second) are
// loaded ONCE into registers and reused for all rows, and the all-gathers inside the launches carry B rows at once (the polling sweep
// is the expensive part of an exchange: it grows with the bytes, not with the number of exchanges).
//   1. qkv_attn_rows_kernel<NB>: block (chunk c, head h): LayerNorm + the 12 q/k/v rows of its head slice for every row -> granules;
//      the q (k, v) of the head gathered per row; split-KV attention of chunk c for every row back to back (the K/V loads of the
//      next row's first round fly under the current row's last); the (m, l, o[64]) partials of the head's 16 chunks meet INSIDE the
//      launch (66 granules per partial) and block c merges row c -> the attention output row, rounded to bf16, in HBM (2 KB per row:
//      the second launch would otherwise re-read B x 67 KB of partials in each of its 256 blocks).
//   2. oproj_fc1_rows_kernel<NB>: block b: out_proj rows 4b .. 4b + 3 for every row -> y1 granules, gathered for all rows at once ->
//      LayerNorm 1 + fc1 rows 16b .. 16b + 15 + ReLU -> bf16-pair granules, gathered for all rows -> fc2 rows 4b .. 4b + 3.
// ARITHMETIC: per row exactly the batch-1 chain's (same helpers, same fmaf order, same merge order): row b of a batch produces the
// bits of its batch-1 run -- the criterion of tests/test_gpu_rows_fused.py -- and, transitively, of the five-launch chain.
// Protocol: the tagged-granule exchange of common.hpp (epoch = position * 32 + layer + 1, rows share the position), bounded sweeps, the
// engine's error word; all 256 blocks must be resident (second launch: 130 KB of LDS, one block per CU) -- a starved grid times out
// once and the engine re-runs the generation on the launch chain that needs no co-residency (engine.hip generate_batch).
// Reference: [3p] OPTDecoderLayer + OptFlashAttention2 reached from shape_opt.py:403-410 with a batch of rows (meshanything.py:143-162).
#pragma once
#include "../attn_decode.hpp"
#include "../common.hpp"
#include "../gemv.hpp"
#include "../oproj_fc1.hpp"
#include "../qkv_attn.hpp"
#include "../state.hpp"

namespace ma {

constexpr int RF_MAX_ROWS = 8;
constexpr int RF_PART = 66;                              // granules of one split-KV partial: 64 x o, m, l
constexpr unsigned RF_ERR_GATHER = 128;

struct RowsFusedArgs {
    QkvAttnArgs q;                                       // first launch (per-row strides as in the rows-in-grid form); q.ws unused
    OprojFc1Args o;                                      // second launch; o.attn_ws unused
    u64* part_gran;                                      // [batch][heads][ATTN_NCHUNK][RF_PART] granules: the in-launch partial exchange
    bf16_t* attn_out;                                    // [batch][hidden] merged attention output, bf16 (the value out_proj's GEMV rounds to)
    int attn_out_stride;
    int B;
};

// ---------------------------------------------------------------------------------------------------------------- first launch
template <int NB, int PRO>
__global__ __launch_bounds__(256) void qkv_attn_rows_kernel(RowsFusedArgs A) {
    typedef AttnGeom<bf16_t> G;
    constexpr int KC = 1024;
    const QkvAttnArgs& a = A.q;
    __shared__ __attribute__((aligned(16))) float xl[KC];
    __shared__ float red[8];
    __shared__ __attribute__((aligned(16))) float qg[NB][64];
    __shared__ __attribute__((aligned(16))) bf16_t kvg[NB][128];
    __shared__ AttnMergeLds<bf16_t> S;
    __shared__ __attribute__((aligned(16))) float pl[ATTN_NCHUNK][RF_PART + 2];      // the gathered partials of the row this block merges
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int c, h;
    qkv_block_role(a, c, h);
    const int nheads = gridDim.y, Hd = a.hidden, B = A.B;
    const DecState sv = a.st[0];                         // rows step together: one position for all
    const int len = sv.pos + 1, pos = len - 1;
    const unsigned epoch = (unsigned)pos * 32u + (unsigned)a.layer + 1u;
    const int per = (len + ATTN_NCHUNK - 1) / ATTN_NCHUNK, c_last = (len - 1) / per;
    const int start = c * per, end = min(len, start + per);
    const int nround = (max(end - start, 0) + 127) >> 7;
    unsigned long long* tr = a.trace ? a.trace + (h * ATTN_NCHUNK + c) * 4 : nullptr;      // diagnostics: start | rows published | q/k/v gathered | attention done
    if (tr && tid == 0) tr[0] = __builtin_amdgcn_s_memrealtime();

    // ---- (1) operands: this block's 12 weight rows, once; every row's input vector ----------------------------------------------------
    QkvOperands op;
    qkv_load_operands<PRO>(a, c, h, op);
    f32x4 xin[NB];
    float x0[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        const int rr = r < B ? r : 0;
        const float* x = a.x + (size_t)rr * a.x_stride;
        xin[r] = *reinterpret_cast<const f32x4*>(x + tid * 4);
        x0[r] = PRO == PRO_LN ? x[0] : 0.f;
    }
    const int row = 64 * h + 4 * c + w;
    const int slot = lane / G::LPP, dsub = lane % G::LPP, woff = w * 32;
    asm volatile("" ::: "memory");

    // ---- (2) per row: LayerNorm, the 12 dot products (gemv_kernel<bf16_t, 1, 2, *, PRO>, bit for bit), publish -----------------------------
#pragma unroll 1
    for (int r = 0; r < B; ++r) {
        f32x4 xv[1] = {xin[0]};
#pragma unroll
        for (int i = 1; i < NB; ++i) if (i == r) xv[0] = xin[i];
        float x0r = x0[0];
#pragma unroll
        for (int i = 1; i < NB; ++i) if (i == r) x0r = x0[i];
        f32x4 gv[1] = {op.gv}, bv[1] = {op.bv};
        if constexpr (PRO == PRO_LN) ln_block_onepass<1>(xv, gv, bv, x0r, tid, KC / 4, KC, a.ln_eps, red);
        else __syncthreads();                            // the previous row's dot products have read xl
        if (a.xn_out && c == 0 && h == 0) *reinterpret_cast<f32x4*>(a.xn_out + (size_t)r * a.xn_stride + tid * 4) = xv[0];
        {
            f32x4 t = xv[0];
            t.x = round_bf16(t.x); t.y = round_bf16(t.y); t.z = round_bf16(t.z); t.w = round_bf16(t.w);
            *reinterpret_cast<f32x4*>(&xl[tid * 4]) = t;
        }
        __syncthreads();
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k0 = (i * 64 + lane) * 8;
            float xs[8];
#pragma unroll
            for (int v = 0; v < 8; v += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(&xl[k0 + v]);
                xs[v] = t.x; xs[v + 1] = t.y; xs[v + 2] = t.z; xs[v + 3] = t.w;
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                float wf[8];
                unpack16<bf16_t>(op.wv[p][i], wf);
#pragma unroll
                for (int v = 0; v < 8; ++v) acc[p] = fmaf(wf[v], xs[v], acc[p]);
            }
        }
        float out3[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) { float v = wave_sum(acc[p]); v += op.bq[p]; out3[p] = v; }
        u64* gran = a.gran + (size_t)r * 3 * Hd;
        if (lane < 3) ps_publish(gran, lane * Hd + row, epoch, __float_as_uint(lane == 0 ? out3[0] : lane == 1 ? out3[1] : out3[2]));
    }

    if (tr && tid == 0) tr[1] = __builtin_amdgcn_s_memrealtime();
    // ---- (3) the K / V stream of row 0 starts now; wave w gathers q (k, v) of rows w and w + 4 ---------------------------------------------
    u32x4 kA[G::U], vA[G::U], kB[G::U], vB[G::U];
    auto issue = [&](int r, int rd, u32x4 (&kr)[G::U], u32x4 (&vr)[G::U]) {
        const bf16_t* kh = a.kcache + (size_t)r * a.kv_row_stride + (size_t)h * a.max_seq * 64 + dsub * G::EPL;
        const bf16_t* vh = a.vcache + (size_t)r * a.kv_row_stride + (size_t)h * a.max_seq * 64 + dsub * G::EPL;
        const int base = start + (rd << 7) + woff + slot;
#pragma unroll
        for (int u = 0; u < G::U; ++u) { const int p = base + u * G::PPW; kr[u] = ld_stream16(kh + (size_t)(p < end ? p : start) * 64); }
#pragma unroll
        for (int u = 0; u < G::U; ++u) { const int p = base + u * G::PPW; vr[u] = ld_stream16(vh + (size_t)(p < end ? p : start) * 64); }
    };
    if (nround > 0) issue(0, 0, kA, vA);
    asm volatile("" ::: "memory");
    {
        const int nparts = c == c_last ? 3 : 1;
        u64 t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        bool done[2] = {w >= B, w + 4 >= B};
        for (;;) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = w + 4 * j;
                if (done[j]) continue;
                const gu64* g64 = (const gu64*)(a.gran + (size_t)r * 3 * Hd);
                u64 v[3];
                bool ok = true;
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    v[p] = (u64)epoch << 32;
                    if (p < nparts) v[p] = __hip_atomic_load(g64 + p * Hd + 64 * h + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (unsigned)(v[p] >> 32) == epoch;
                }
                if (__all(ok)) {
                    qg[r][lane] = round_bf16(__uint_as_float((unsigned)v[0]));
                    kvg[r][lane] = f2bf(__uint_as_float((unsigned)v[1]));
                    kvg[r][64 + lane] = f2bf(__uint_as_float((unsigned)v[2]));
                    done[j] = true;
                }
            }
            if (done[0] && done[1]) break;
            __builtin_amdgcn_s_sleep(1);
            if (xchg_expired(spins, t0, a.err)) {
                if (lane == 0) xchg_raise(a.err, QA_ERR_GATHER, spins);
#pragma unroll
                for (int j = 0; j < 2; ++j) if (!done[j]) { const int r = w + 4 * j; qg[r][lane] = 0.f; kvg[r][lane] = 0; kvg[r][64 + lane] = 0; }
                break;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = w + 4 * j;
            if (r < B && c == c_last && lane < 32) {     // the newest position joins row r's cache
                const u32x2 pk = *reinterpret_cast<const u32x2*>(&kvg[r][(lane >> 4) * 64 + (lane & 15) * 4]);
                bf16_t* plane = (lane >> 4) ? a.vcache : a.kcache;
                *reinterpret_cast<u32x2*>(plane + (size_t)r * a.kv_row_stride + ((size_t)h * a.max_seq + pos) * 64 + (lane & 15) * 4) = pk;
            }
        }
    }
    __syncthreads();
    if (tr && tid == 0) tr[2] = __builtin_amdgcn_s_memrealtime();

    // ---- (4) attention over chunk c, row after row: (row, round) is one flat sequence through the two register sets -----------------------
    const int ovr = c == c_last ? pos : -1;
    const int total = B * nround;
    auto seq_issue = [&](int s, u32x4 (&kr)[G::U], u32x4 (&vr)[G::U]) { issue(s / nround, s % nround, kr, vr); };
    int s = 0;
#pragma unroll 1
    for (int r = 0; r < B; ++r) {
        float qv[G::EPL];
        {
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(&qg[r][dsub * G::EPL]), q1 = *reinterpret_cast<const f32x4*>(&qg[r][dsub * G::EPL + 4]);
            qv[0] = q0.x; qv[1] = q0.y; qv[2] = q0.z; qv[3] = q0.w; qv[4] = q1.x; qv[5] = q1.y; qv[6] = q1.z; qv[7] = q1.w;
        }
        const u32x4 ok4 = *reinterpret_cast<const u32x4*>(&kvg[r][dsub * G::EPL]), ov4 = *reinterpret_cast<const u32x4*>(&kvg[r][64 + dsub * G::EPL]);
        AttnSlotState<bf16_t> ss;
        ss.m = -1e30f; ss.l = 0.f;
#pragma unroll
        for (int e = 0; e < G::EPL; ++e) ss.o[e] = 0.f;
#pragma unroll 1
        for (int rd = 0; rd < nround; ++rd, ++s) {
            if ((s & 1) == 0) {
                if (s + 1 < total) seq_issue(s + 1, kB, vB);
                attn_round_reduce<bf16_t, true>(ss, qv, kA, vA, start + (rd << 7) + woff + slot, end, ovr, ok4, ov4);
            } else {
                if (s + 1 < total) seq_issue(s + 1, kA, vA);
                attn_round_reduce<bf16_t, true>(ss, qv, kB, vB, start + (rd << 7) + woff + slot, end, ovr, ok4, ov4);
            }
        }
        const int gs = w * G::PPW + slot;
        if (dsub == 0) { S.sm[gs] = ss.m; S.sl[gs] = ss.l; }
#pragma unroll
        for (int e = 0; e < G::EPL; ++e) S.so[gs][dsub * G::EPL + e] = ss.o[e];
        __syncthreads();
        {
            float M, L, O;
            attn_fold_quarter<bf16_t>(S, w, lane, M, L, O);
            if (lane == 0) { S.qm[w] = M; S.ql[w] = L; }
            S.qo[w][lane] = O;
        }
        __syncthreads();
        if (w == 0) {                                    // the partial of (row r, head h, chunk c) -> 66 granules for the head's 16 blocks
            float M, L, O;
            attn_fold_block<bf16_t>(S, lane, M, L, O);
            u64* pg = A.part_gran + (((size_t)r * nheads + h) * ATTN_NCHUNK + c) * RF_PART;
            ps_publish(pg, lane, epoch, __float_as_uint(O));
            if (lane < 2) ps_publish(pg, 64 + lane, epoch, __float_as_uint(lane == 0 ? M : L));
        }
    }

    if (tr && tid == 0) tr[3] = __builtin_amdgcn_s_memrealtime();
    // ---- (5) block c merges row c: the 16 partials of (row c, head h) gathered by wave 0, merged in the chain's order -------------------------
    if (c < B && w == 0) {
        const int r = c;
        const gu64* pg = (const gu64*)(A.part_gran + ((size_t)r * nheads + h) * ATTN_NCHUNK * RF_PART);
        constexpr int NG = ATTN_NCHUNK * RF_PART, NL = (NG + 63) / 64;                  // 1056 granules, 17 per lane
        u64 t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0, pend = (1u << NL) - 1u;
        // a chunk's (m, l) granules are published after its 64 o granules: poll those 32 alone until they are there (see oproj_fc1_rows (3))
        for (;;) {
            u64 v = (u64)epoch << 32;
            if (lane < 2 * ATTN_NCHUNK) v = __hip_atomic_load(pg + (lane >> 1) * RF_PART + 64 + (lane & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((unsigned)(v >> 32) == epoch)) break;
            __builtin_amdgcn_s_sleep(2);
            if (xchg_expired(spins, t0, a.err)) break;         // the full sweep below raises the error word
        }
        for (;;) {
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                if (!((pend >> k) & 1u)) continue;
                const int id = k * 64 + lane;
                u64 v = (u64)epoch << 32;
                if (id < NG) v = __hip_atomic_load(pg + id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = (unsigned)(v >> 32) == epoch;
                if (ok && id < NG) pl[id / RF_PART][id % RF_PART] = __uint_as_float((unsigned)v);
                if (__all(ok)) pend &= ~(1u << k);
            }
            if (!pend) break;
            __builtin_amdgcn_s_sleep(1);
            if (xchg_expired(spins, t0, a.err)) {
                if (lane == 0) xchg_raise(a.err, RF_ERR_GATHER, spins);
                for (int id = lane; id < NG; id += 64) pl[id / RF_PART][id % RF_PART] = id % RF_PART == 65 ? 1.f : 0.f;      // l = 1: no division by zero
                break;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < 16) {                                 // 4 dims per lane: attn_partials_merge on the same values in the same order
            f32x4 pml[ATTN_NCHUNK / 2], po[ATTN_NCHUNK];
#pragma unroll
            for (int i = 0; i < ATTN_NCHUNK / 2; ++i) pml[i] = f32x4{pl[2 * i][64], pl[2 * i][65], pl[2 * i + 1][64], pl[2 * i + 1][65]};
#pragma unroll
            for (int cc = 0; cc < ATTN_NCHUNK; ++cc) po[cc] = f32x4{pl[cc][4 * lane], pl[cc][4 * lane + 1], pl[cc][4 * lane + 2], pl[cc][4 * lane + 3]};
            const f32x4 m4 = attn_partials_merge(pml, po);
            u32x2 pk;
            pk.x = (uint32_t)f2bf(m4.x) | ((uint32_t)f2bf(m4.y) << 16);
            pk.y = (uint32_t)f2bf(m4.z) | ((uint32_t)f2bf(m4.w) << 16);
            *reinterpret_cast<u32x2*>(A.attn_out + (size_t)r * A.attn_out_stride + h * 64 + 4 * lane) = pk;
        }
    }
}

// --------------------------------------------------------------------------------------------------------------- second launch
template <int NB>
__global__ __launch_bounds__(256) void oproj_fc1_rows_kernel(RowsFusedArgs A) {
    constexpr int KC = 1024, KF = 4096;
    const OprojFc1Args& a = A.o;
    extern __shared__ __attribute__((aligned(16))) char rf_smem[];
    float* xl = reinterpret_cast<float*>(rf_smem);                                   // [NB][KC]: out_proj input, then LN1 output (rounded)
    float* yraw = xl + NB * KC;                                                       // [NB][KC]: gathered y1
    bf16_t* ffl = reinterpret_cast<bf16_t*>(yraw + NB * KC);                          // [NB][KF]: gathered relu(fc1), bf16
    float* h1l = reinterpret_cast<float*>(ffl + NB * KF);                             // [NB][4]
    float* red = h1l + NB * 4;                                                        // [2][8]: alternating per row (no barrier between two rows' LayerNorms)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b = blockIdx.x, B = A.B;
    const unsigned epoch = (unsigned)a.st[0].pos * 32u + (unsigned)a.layer + 1u;
    unsigned long long* tr = a.trace ? a.trace + b * 4 : nullptr;                     // diagnostics: start | y1 published | y1 gathered | relu(fc1) gathered
    if (tr && tid == 0) tr[0] = __builtin_amdgcn_s_memrealtime();

    // ---- (1) operands, once: out_proj row, 4 fc1 rows, fc2 row (72 KB per block); every row's attention output -----------------------------------
    const int orow = 4 * b + w;
    u32x4 wo[2], w1[4][2], w2[8];
#pragma unroll
    for (int i = 0; i < 2; ++i) wo[i] = ld_stream16(a.Wo + (size_t)orow * KC + (i * 64 + lane) * 8);
    const float e_bo = a.bo[orow];
    u32x2 ain[NB];
    float e_res[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        const int rr = r < B ? r : 0;
        ain[r] = *reinterpret_cast<const u32x2*>(A.attn_out + (size_t)rr * A.attn_out_stride + tid * 4);
        e_res[r] = a.res[(size_t)rr * a.res_stride + orow];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) w1[j][i] = ld_stream16(a.W1 + (size_t)(16 * b + 4 * w + j) * KC + (i * 64 + lane) * 8);
    float e_b1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) e_b1[j] = a.b1[16 * b + 4 * w + j];
    const f32x4 gv0 = *reinterpret_cast<const f32x4*>(a.ln_g + tid * 4), bv0 = *reinterpret_cast<const f32x4*>(a.ln_b + tid * 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) w2[i] = ld_stream16(a.W2 + (size_t)orow * KF + (i * 64 + lane) * 8);
    const float e_b2 = a.b2[orow];
    asm volatile("" ::: "memory");

    // ---- (2) out_proj for every row (gemv_kernel<bf16_t, 1, 2, 1, PRO_ATTN> on the merged, rounded attention output) ------------------------
#pragma unroll
    for (int r = 0; r < NB; ++r)
        if (r < B) *reinterpret_cast<f32x4*>(&xl[r * KC + tid * 4]) = f32x4{bf_lo(ain[r].x), bf_hi(ain[r].x), bf_lo(ain[r].y), bf_hi(ain[r].y)};
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < B; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k0 = (i * 64 + lane) * 8;
            float xs[8], wf[8];
#pragma unroll
            for (int v = 0; v < 8; v += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(&xl[r * KC + k0 + v]);
                xs[v] = t.x; xs[v + 1] = t.y; xs[v + 2] = t.z; xs[v + 3] = t.w;
            }
            unpack16<bf16_t>(wo[i], wf);
#pragma unroll
            for (int v = 0; v < 8; ++v) acc = fmaf(wf[v], xs[v], acc);
        }
        float v = wave_sum(acc);
        v += e_bo;
        float er = e_res[0];
#pragma unroll
        for (int i = 1; i < NB; ++i) if (i == r) er = e_res[i];
        v += er;
        if (lane == 0) ps_publish(a.gran + (size_t)r * KC, orow, epoch, __float_as_uint(v));
    }

    if (tr && tid == 0) tr[1] = __builtin_amdgcn_s_memrealtime();
    // ---- (3) all-gather of y1 for all rows: wave w sweeps its quarter of every row.  What an exchange costs is its POLLING (256 blocks x
    //      64 KB per pass at 8 rows, re-read until the slowest producer is through, starved every other stream: the first version of these
    //      launches ran 1.4-1.8x SLOWER than the launch chain, profiles/r03_rows_fused_v1_first_correct.txt).  A producer wave publishes its
    //      rows in order, so the LAST row is polled alone (the traffic of a batch-1 exchange); once it is complete the earlier rows are, in
    //      practice, too: one pass over them -- still verified tag by tag, nothing relies on the order in which stores become visible ----------
    {
        u64 t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        bool failed = false;
#pragma unroll 1
        for (int stage = 0; stage < 2 && !failed; ++stage) {
            unsigned pend = stage == 0 ? (0xfu << (4 * (B - 1))) : (B >= 2 ? ((1u << (4 * (B - 1))) - 1u) : 0u);      // bit 4 r + k: granules [w * 256 + k * 64, + 64) of row r
            while (pend) {
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    if (!((pend >> (4 * r)) & 0xfu)) continue;
                    const gu64* g64 = (const gu64*)(a.gran + (size_t)r * KC) + w * 256;
                    u64 v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[k] = (u64)epoch << 32;
                        if ((pend >> (4 * r + k)) & 1u) v[k] = __hip_atomic_load(g64 + k * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if ((pend >> (4 * r + k)) & 1u) {
                            const bool ok = (unsigned)(v[k] >> 32) == epoch;
                            if (ok) yraw[r * KC + w * 256 + k * 64 + lane] = __uint_as_float((unsigned)v[k]);
                            if (__all(ok)) pend &= ~(1u << (4 * r + k));
                        }
                    }
                }
                if (!pend) break;
                __builtin_amdgcn_s_sleep(2);
                if (xchg_expired(spins, t0, a.err)) {
                    if (lane == 0) xchg_raise(a.err, OF_ERR_GATHER, spins);
                    for (int r = 0; r < B; ++r)
#pragma unroll
                        for (int k = 0; k < 4; ++k) yraw[r * KC + w * 256 + k * 64 + lane] = 0.f;
                    failed = true;
                    break;
                }
            }
        }
    }
    __syncthreads();
    if (tr && tid == 0) tr[2] = __builtin_amdgcn_s_memrealtime();

    // ---- (4) per row: LayerNorm 1, fc1 rows (gemv_kernel<bf16_t, 1, 2, 4, PRO_LN>), ReLU, publish as bf16 pairs ------------------------------
#pragma unroll 1
    for (int r = 0; r < B; ++r) {
        f32x4 xv[1], gv[1] = {gv0}, bv[1] = {bv0};
        xv[0] = *reinterpret_cast<const f32x4*>(&yraw[r * KC + tid * 4]);
        const float x0 = yraw[r * KC];
        ln_block_onepass<1>(xv, gv, bv, x0, tid, KC / 4, KC, a.ln_eps, red + 8 * (r & 1));
        if (tid == b) { h1l[r * 4 + 0] = xv[0].x; h1l[r * 4 + 1] = xv[0].y; h1l[r * 4 + 2] = xv[0].z; h1l[r * 4 + 3] = xv[0].w; }
        f32x4 t = xv[0];
        t.x = round_bf16(t.x); t.y = round_bf16(t.y); t.z = round_bf16(t.z); t.w = round_bf16(t.w);
        *reinterpret_cast<f32x4*>(&xl[r * KC + tid * 4]) = t;
    }
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < B; ++r) {
        float acc4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k0 = (i * 64 + lane) * 8;
            float xs[8];
#pragma unroll
            for (int v = 0; v < 8; v += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(&xl[r * KC + k0 + v]);
                xs[v] = t.x; xs[v + 1] = t.y; xs[v + 2] = t.z; xs[v + 3] = t.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float wf[8];
                unpack16<bf16_t>(w1[j][i], wf);
#pragma unroll
                for (int v = 0; v < 8; ++v) acc4[j] = fmaf(wf[v], xs[v], acc4[j]);
            }
        }
        float outv = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = wave_sum(acc4[j]);
            v += e_b1[j];
            v = fmaxf(v, 0.0f);
            if (lane == j) outv = v;
        }
        const float nb = __shfl_down(outv, 1, 64);
        if (lane == 0 || lane == 2) ps_publish(a.gran2 + (size_t)r * KF, 8 * b + 2 * w + (lane >> 1), epoch, (unsigned)f2bf(outv) | ((unsigned)f2bf(nb) << 16));
    }

    // ---- (5) all-gather of relu(fc1) for all rows (2048 bf16-pair granules per row, wave w: [512 w, 512 w + 512)): the last row alone first,
    //      then the earlier rows four at a time (see (3)) ---------------------------------------------------------------------------------------
    {
        u64 t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        bool failed = false;
#pragma unroll 1
        for (int pass = 0; pass < 3 && !failed; ++pass) {        // pass 0: row B - 1 | pass 1: rows 0 .. 3 | pass 2: rows 4 .. 7 (each without row B - 1)
            const int r0 = pass == 0 ? B - 1 : (pass - 1) * 4;
            unsigned pend = 0;                                     // bit 8 j + k: granules [w * 512 + k * 64, + 64) of row r0 + j
            if (pass == 0) pend = 0xffu;
            else for (int j = 0; j < 4; ++j) if (r0 + j < B - 1) pend |= 0xffu << (8 * j);
            while (pend) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!((pend >> (8 * j)) & 0xffu)) continue;
                    const int r = r0 + j;
                    const gu64* g64 = (const gu64*)(a.gran2 + (size_t)r * KF) + w * 512;
                    uint32_t* fr = reinterpret_cast<uint32_t*>(ffl + (size_t)r * KF) + w * 512;
                    u64 v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        v[k] = (u64)epoch << 32;
                        if ((pend >> (8 * j + k)) & 1u) v[k] = __hip_atomic_load(g64 + k * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if ((pend >> (8 * j + k)) & 1u) {
                            const bool ok = (unsigned)(v[k] >> 32) == epoch;
                            if (ok) fr[k * 64 + lane] = (unsigned)v[k];
                            if (__all(ok)) pend &= ~(1u << (8 * j + k));
                        }
                    }
                }
                if (!pend) break;
                __builtin_amdgcn_s_sleep(2);
                if (xchg_expired(spins, t0, a.err)) {
                    if (lane == 0) xchg_raise(a.err, OF_ERR_GATHER, spins);
                    for (int r = 0; r < B; ++r)
#pragma unroll
                        for (int k = 0; k < 8; ++k) reinterpret_cast<uint32_t*>(ffl + (size_t)r * KF)[w * 512 + k * 64 + lane] = 0u;
                    failed = true;
                    break;
                }
            }
        }
    }
    __syncthreads();
    if (tr && tid == 0) tr[3] = __builtin_amdgcn_s_memrealtime();

    // ---- (6) fc2 (gemv_kernel<bf16_t, 1, 8, 1, PRO_PLAIN>), row 4b + w, for every row -----------------------------------------------------------
#pragma unroll 1
    for (int r = 0; r < B; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k0 = (i * 64 + lane) * 8;
            float xs[8], wf[8];
            const u32x4 t = *reinterpret_cast<const u32x4*>(ffl + (size_t)r * KF + k0);
            xs[0] = bf_lo(t.x); xs[1] = bf_hi(t.x); xs[2] = bf_lo(t.y); xs[3] = bf_hi(t.y);
            xs[4] = bf_lo(t.z); xs[5] = bf_hi(t.z); xs[6] = bf_lo(t.w); xs[7] = bf_hi(t.w);
            unpack16<bf16_t>(w2[i], wf);
#pragma unroll
            for (int v = 0; v < 8; ++v) acc = fmaf(wf[v], xs[v], acc);
        }
        float v = wave_sum(acc);
        v += e_b2;
        v += h1l[r * 4 + w];
        if (lane == 0) a.y2_out[(size_t)r * a.y2_stride + orow] = v;
    }
}

constexpr size_t rf_oproj_lds(int nb) { return (size_t)nb * (1024 * 4 + 1024 * 4 + 4096 * 2 + 16) + 128; }

inline hipError_t launch_qkv_attn_rows(const RowsFusedArgs& A, int heads, hipStream_t s) {
    if (A.q.hidden != 1024 || heads * 64 != A.q.hidden || A.B < 1 || A.B > RF_MAX_ROWS) return hipErrorInvalidValue;
    const dim3 grid(ATTN_NCHUNK, heads, 1);
    if (A.B <= 4) {
        if (A.q.ln_g) hipLaunchKernelGGL((qkv_attn_rows_kernel<4, PRO_LN>), grid, dim3(256), 0, s, A);
        else hipLaunchKernelGGL((qkv_attn_rows_kernel<4, PRO_PLAIN>), grid, dim3(256), 0, s, A);
    } else {
        if (A.q.ln_g) hipLaunchKernelGGL((qkv_attn_rows_kernel<8, PRO_LN>), grid, dim3(256), 0, s, A);
        else hipLaunchKernelGGL((qkv_attn_rows_kernel<8, PRO_PLAIN>), grid, dim3(256), 0, s, A);
    }
    return hipGetLastError();
}

inline hipError_t rf_prepare() {
    hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(oproj_fc1_rows_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rf_oproj_lds(4));
    if (r != hipSuccess) return r;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(oproj_fc1_rows_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rf_oproj_lds(8));
}

inline hipError_t launch_oproj_fc1_rows(const RowsFusedArgs& A, int hidden, int ffn, hipStream_t s) {
    if (hidden != 1024 || ffn != 4096 || !A.o.W2 || A.B < 1 || A.B > RF_MAX_ROWS) return hipErrorInvalidValue;
    if (A.B <= 4) hipLaunchKernelGGL((oproj_fc1_rows_kernel<4>), dim3(hidden / 4), dim3(256), rf_oproj_lds(4), s, A);
    else hipLaunchKernelGGL((oproj_fc1_rows_kernel<8>), dim3(hidden / 4), dim3(256), rf_oproj_lds(8), s, A);
    return hipGetLastError();
}

}  // namespace ma
