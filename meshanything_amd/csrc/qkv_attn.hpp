// Fused q/k/v projection + split-KV decode attention of one layer, batch-1 launch chain (any policy: templated on the storage format; hidden = 1024).
//
// Replaces two launches of the chain -- the fused-QKV GEMV (gemv.hpp, EPI_QKV; [3p] OPTAttention q/k/v projections reached from
// shape_opt.py:403-410) and the split-KV attention (attn_decode.hpp; [3p] OptFlashAttention2 + the per-step torch.cat) -- by one:
// attention of head h needs only the 192 q/k/v outputs of head h, so the exchange between the two is not all-to-all.  Block
// (chunk c, head h) of the 16 x heads grid
//   1. issues, in its first instructions: the input vector + LayerNorm parameters and its 12 weight rows (q, k, v rows
//      64 h + 4 c + wave of the fused [3H][H] matrix: 24 KB) -- NOT yet the cache rows: they would compete with these on the way to
//      the publish (step 3) and are requested right after it, riding under the exchange (-2 % per step, measured);
//   2. runs the GEMV prologue (LayerNorm of the post-LN stream, common.hpp) and its 12 dot products -- the arithmetic of
//      gemv_kernel<bf16, 1, 2, *, *>, bit for bit;
//   3. publishes the 12 values as 8-byte {epoch, value} granules (one sc1 store each; MI355X guide, Guideline 16 R2) and one
//      wave sweeps the 64 granules of q_h (the block holding the newest position also k_h, v_h: 192) until every tag carries this
//      launch's epoch -- the exchange stays among the 16 blocks of a head.  Epoch = position * 32 + layer + 1: strictly
//      increasing within a generation, the buffer is zeroed when the position restarts;
//   4. attends over its chunk with the shared round / merge code of attn_decode.hpp (the newest position's K / V rows come from the
//      granules, and are written to the cache for later steps) and writes the (m, l, o[64]) partial for the out_proj launch.
// One launch boundary and one dependent-vector round trip less per layer; the K / V stream is in flight during the exchange.
// Every block of the grid must be resident (256 x batch blocks of 256 threads: any device with >= 32 CUs); the sweep is bounded
// (20 ms) and raises the engine's error word instead of hanging.
#pragma once
#include "attn_decode.hpp"
#include "common.hpp"
#include "gemv.hpp"
#include "state.hpp"

namespace ma {

struct QkvAttnArgs {
    const bf16_t* W; const float* bias;                  // fused [3 hidden][hidden] projection, [3 hidden] bias
    const float* x; const float* ln_g; const float* ln_b; float ln_eps; float* xn_out;      // as GemvArgs (ln_g == null: no LayerNorm)
    bf16_t* kcache; bf16_t* vcache; int max_seq; int hidden;
    const DecState* st; int len_override; int layer;
    float* ws;                                           // split-KV partials (attn_workspace_floats per batch row)
    u64* gran;                                           // [batch][3 hidden] granules
    unsigned* err;
    int x_stride, xn_stride; size_t kv_row_stride;
    unsigned long long* trace;
    int xcd_local;                                       // 1: the 16 blocks of a head are dispatched to ONE XCD (see qkv_block_role)
};
constexpr unsigned QA_ERR_GATHER = 16;

// the operands of a block that do not depend on the input vector: its 12 weight rows (4 waves x q, k, v), their biases, the LayerNorm
// parameters.  A caller that runs other work first (layer_fused.hpp) requests them early and hands them over.
// NP = 16-byte pieces per lane and weight row: 2 in the 16-bit policies (8 elements each), 4 in the fp32 policy (4 elements each; round 6: the fp32
// policy on the fused chain -- the struct's bf16_t pointers are then plain addresses of fp32 storage, reinterpreted here)
template <typename HT> struct QkvStore { typedef HT T; typedef uint16_t Word; static constexpr int VEC = 8, NP = 2; };
template <> struct QkvStore<float> { typedef float T; typedef float Word; static constexpr int VEC = 4, NP = 4; };
template <int NP> struct QkvOperandsT { u32x4 wv[3][NP]; float bq[3]; f32x4 gv, bv; };
typedef QkvOperandsT<2> QkvOperands;
template <int PRO, typename HT = bf16_t>
__device__ __forceinline__ void qkv_load_operands(const QkvAttnArgs& a, const int c, const int h, QkvOperandsT<QkvStore<HT>::NP>& op) {
    typedef typename QkvStore<HT>::T ST;
    constexpr int VEC = QkvStore<HT>::VEC, NP = QkvStore<HT>::NP;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if constexpr (PRO == PRO_LN) {
        op.gv = *reinterpret_cast<const f32x4*>(a.ln_g + tid * 4);
        op.bv = *reinterpret_cast<const f32x4*>(a.ln_b + tid * 4);
    }
    const int row = 64 * h + 4 * c + w;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const ST* wr = reinterpret_cast<const ST*>(a.W) + (size_t)(p * a.hidden + row) * 1024;
#pragma unroll
        for (int i = 0; i < NP; ++i) op.wv[p][i] = ld_stream16(wr + (i * 64 + lane) * VEC);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) op.bq[p] = a.bias[p * a.hidden + row];
}

// XLDS: the input vector is already in LDS (`xlds`, 1024 floats; layer_fused.hpp) instead of global memory, and the operands were
// requested by the caller.
// HT: the 16-bit format of the weights and the cache (bf16_t | f16_t, common.hpp H16)
template <int PRO, bool XLDS, typename HT = bf16_t>
__device__ __forceinline__ void qkv_attn_body(QkvAttnArgs a, const int c, const int h, const int nheads, const int brow, const float* xlds, QkvOperandsT<QkvStore<HT>::NP>& op) {
    typedef typename QkvStore<HT>::T ST;                             // storage element of the weights and the cache
    constexpr int VEC = QkvStore<HT>::VEC, NP = QkvStore<HT>::NP;
    typedef AttnGeom<ST> G;
    constexpr int KC = 1024;
    __shared__ __attribute__((aligned(16))) float xl[KC];
    __shared__ float red[8];
    __shared__ __attribute__((aligned(16))) float qg[64];            // q_h, already rounded to the cache format (one conversion per lane, by the sweeping wave)
    typedef typename QkvStore<HT>::Word KW;                          // the cache's element as a plain word (uint16_t | float)
    __shared__ __attribute__((aligned(16))) KW kvg[128];            // newest position's k_h | v_h in the cache's format (what the cache holds)
    __shared__ AttnMergeLds<ST> S;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int Hd = a.hidden;
    if (a.trace && tid == 0) a.trace[(h * ATTN_NCHUNK + c) * 4 + 0] = __builtin_amdgcn_s_memrealtime();
    const float* x = a.x + (size_t)brow * a.x_stride;
    const DecState sv = a.st[brow];
    const int len = a.len_override >= 0 ? a.len_override : sv.pos + 1;
    const int pos = len - 1;                             // the newest position: its K / V rows are produced by this launch
    const unsigned epoch = (unsigned)pos * 32u + (unsigned)a.layer + 1u;
    const int per = (len + ATTN_NCHUNK - 1) / ATTN_NCHUNK, c_last = (len - 1) / per;
    const int start = c * per, end = min(len, start + per);
    const int nround = (max(end - start, 0) + 127) >> 7;

    // ---- (1) loads, in consumption order: input vector (+ LayerNorm parameters), weight rows, bias ------------------------------------
    f32x4 xv[1];
    float x0 = 0.f;
    if constexpr (XLDS) xv[0] = *reinterpret_cast<const f32x4*>(xlds + tid * 4);
    else xv[0] = *reinterpret_cast<const f32x4*>(x + tid * 4);
    if constexpr (PRO == PRO_LN) x0 = XLDS ? xlds[0] : x[0];
    if constexpr (!XLDS) qkv_load_operands<PRO, HT>(a, c, h, op);
    const int row = 64 * h + 4 * c + w;                  // this wave's row inside each of the q, k, v blocks of the fused matrix
    f32x4 gv[1] = {op.gv}, bv[1] = {op.bv};
    const int slot = lane / G::LPP, dsub = lane % G::LPP, woff = w * 32;
    const ST* kh = reinterpret_cast<const ST*>(a.kcache) + (size_t)brow * a.kv_row_stride + (size_t)h * a.max_seq * 64 + dsub * G::EPL;
    const ST* vh = reinterpret_cast<const ST*>(a.vcache) + (size_t)brow * a.kv_row_stride + (size_t)h * a.max_seq * 64 + dsub * G::EPL;
    u32x4 kA[G::U], vA[G::U], kB[G::U], vB[G::U];
    auto issue = [&](int r, u32x4 (&kr)[G::U], u32x4 (&vr)[G::U]) {
        const int base = start + (r << 7) + woff + slot;
#pragma unroll
        for (int u = 0; u < G::U; ++u) { const int p = base + u * G::PPW; kr[u] = ld_stream16(kh + (size_t)(p < end ? p : start) * 64); }
#pragma unroll
        for (int u = 0; u < G::U; ++u) { const int p = base + u * G::PPW; vr[u] = ld_stream16(vh + (size_t)(p < end ? p : start) * 64); }
    };
    asm volatile("" ::: "memory");                       // pin the weight / bias loads HERE

    // ---- (2) prologue + dot products: gemv_kernel<bf16_t, 1, 2, *, PRO> for the rows of this block --------------------------------
    if constexpr (PRO == PRO_LN) ln_block_onepass<1>(xv, gv, bv, x0, tid, KC / 4, KC, a.ln_eps, red);
    if (a.xn_out && c == 0 && h == 0) *reinterpret_cast<f32x4*>(a.xn_out + (size_t)brow * a.xn_stride + tid * 4) = xv[0];
    {
        f32x4 r = xv[0];
        r.x = H16<HT>::round(r.x); r.y = H16<HT>::round(r.y); r.z = H16<HT>::round(r.z); r.w = H16<HT>::round(r.w);
        *reinterpret_cast<f32x4*>(&xl[tid * 4]) = r;
    }
    __syncthreads();
    // fp32: gemv_kernel<float, 2, 2, *> splits K between two waves (pieces 0, 1 | 2, 3) and adds the halves in LDS: the same two half-sums here, in one wave
    constexpr int NH = NP / 2;                                       // halves of K that are summed apart (1: the 16-bit policies' one sum)
    float acc[NH][3];
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) { acc[hf][0] = 0.f; acc[hf][1] = 0.f; acc[hf][2] = 0.f; }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int k0 = (i * 64 + lane) * VEC;
        float xs[VEC];
#pragma unroll
        for (int v = 0; v < VEC; v += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(&xl[k0 + v]);
            xs[v] = t.x; xs[v + 1] = t.y; xs[v + 2] = t.z; xs[v + 3] = t.w;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            float wf[VEC];
            unpack16<ST>(op.wv[p][i], wf);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[i / 2 % NH][p] = fmaf(wf[v], xs[v], acc[i / 2 % NH][p]);
        }
    }
    float out3[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        float v = wave_sum(acc[0][p]);
        if constexpr (NH == 2) { float t = 0.f; t += v; t += wave_sum(acc[1][p]); v = t; }      // (v = 0 + half 0 + half 1: gemv_kernel's order)
        v += op.bq[p]; out3[p] = v;
    }
    if (a.trace && tid == 0) a.trace[(h * ATTN_NCHUNK + c) * 4 + 1] = __builtin_amdgcn_s_memrealtime();

    // ---- (3) exchange inside the head: publish 3 values per wave, wave 0 sweeps what this block needs -----------------------------
    u64* gran = a.gran + (size_t)brow * 3 * Hd;
    if (lane < 3) ps_publish(gran, lane * Hd + row, epoch, __float_as_uint(lane == 0 ? out3[0] : lane == 1 ? out3[1] : out3[2]));
    // cache rows are requested once this block's q/k/v rows are out: they ride under the exchange and do not compete with the input
    // vector and the weight rows on the way to the publish (-2 % per step against requesting them in the first instructions; a
    // second round in flight buys nothing: profiles/r02_ab_exchange_and_load_placement.txt)
    if (nround > 0) issue(0, kA, vA);
    asm volatile("" ::: "memory");                       // pin them HERE (hipcc would sink them below the sweep)
    if (w == 0) {
        const gu64* g64 = (const gu64*)gran;
        const int nparts = c == c_last ? 3 : 1;          // q for everyone; k, v for the block that holds the newest position
        u64 t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        for (;;) {
            u64 v[3];
            bool ok = true;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                v[p] = (u64)epoch << 32;
                if (p < nparts) v[p] = __hip_atomic_load(g64 + p * Hd + 64 * h + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (unsigned)(v[p] >> 32) == epoch;
            }
            if (__all(ok)) {
                qg[lane] = H16<HT>::round(__uint_as_float((unsigned)v[0]));
                if constexpr (sizeof(ST) == 4) { kvg[lane] = __uint_as_float((unsigned)v[1]); kvg[64 + lane] = __uint_as_float((unsigned)v[2]); }
                else { kvg[lane] = H16<HT>::bits(__uint_as_float((unsigned)v[1])); kvg[64 + lane] = H16<HT>::bits(__uint_as_float((unsigned)v[2])); }
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            if (xchg_expired(spins, t0, a.err)) {
                if (lane == 0) xchg_raise(a.err, QA_ERR_GATHER, spins);
                qg[lane] = 0.f; kvg[lane] = 0; kvg[64 + lane] = 0;
                break;
            }
        }
        if (lane == 0) xchg_note_slow(a.err, spins, t0);
        if (c == c_last && lane < 32) {                  // the newest position joins the cache (read by later steps' launches)
            KW* plane = reinterpret_cast<KW*>((lane >> 4) ? a.vcache : a.kcache) + (size_t)brow * a.kv_row_stride + ((size_t)h * a.max_seq + pos) * 64 + (lane & 15) * 4;
            const KW* src = kvg + (lane >> 4) * 64 + (lane & 15) * 4;      // four elements per lane
            if constexpr (sizeof(ST) == 4) *reinterpret_cast<f32x4*>(plane) = *reinterpret_cast<const f32x4*>(src);
            else *reinterpret_cast<u32x2*>(plane) = *reinterpret_cast<const u32x2*>(src);
        }
    }
    __syncthreads();
    if (a.trace && tid == 0) a.trace[(h * ATTN_NCHUNK + c) * 4 + 2] = __builtin_amdgcn_s_memrealtime();

    // ---- (4) attention over chunk c (attn_decode.hpp, the launch chain's arithmetic; the newest position from the granules) ----------
    float qv[G::EPL];
#pragma unroll
    for (int e = 0; e < G::EPL; e += 4) {
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(qg + dsub * G::EPL + e);
        qv[e] = q0.x; qv[e + 1] = q0.y; qv[e + 2] = q0.z; qv[e + 3] = q0.w;
    }
    const u32x4 ok4 = *reinterpret_cast<const u32x4*>(kvg + dsub * G::EPL), ov4 = *reinterpret_cast<const u32x4*>(kvg + 64 + dsub * G::EPL);
    const int ovr = c == c_last ? pos : -1;
    AttnSlotState<ST> ss;
    ss.m = -1e30f; ss.l = 0.f;
#pragma unroll
    for (int e = 0; e < G::EPL; ++e) ss.o[e] = 0.f;
    for (int r = 0; r < nround; r += 2) {
        if (r + 1 < nround) issue(r + 1, kB, vB);
        attn_round_reduce<ST, true>(ss, qv, kA, vA, start + (r << 7) + woff + slot, end, ovr, ok4, ov4);
        if (r + 1 < nround) {
            if (r + 2 < nround) issue(r + 2, kA, vA);
            attn_round_reduce<ST, true>(ss, qv, kB, vB, start + ((r + 1) << 7) + woff + slot, end, ovr, ok4, ov4);
        }
    }
    const int gs = w * G::PPW + slot;
    if (dsub == 0) { S.sm[gs] = ss.m; S.sl[gs] = ss.l; }
#pragma unroll
    for (int e = 0; e < G::EPL; ++e) S.so[gs][dsub * G::EPL + e] = ss.o[e];
    __syncthreads();
    {
        float M, L, O;
        attn_fold_quarter<ST>(S, w, lane, M, L, O);
        if (lane == 0) { S.qm[w] = M; S.ql[w] = L; }
        S.qo[w][lane] = O;
    }
    __syncthreads();
    if (w == 0) {
        float M, L, O;
        attn_fold_block<ST>(S, lane, M, L, O);
        float* ws = a.ws + (size_t)brow * attn_workspace_floats(nheads);
        float* ml = ws + ((size_t)h * ATTN_NCHUNK + c) * 2;
        float* op = ws + (size_t)nheads * ATTN_NCHUNK * 2 + ((size_t)h * ATTN_NCHUNK + c) * 64;
        if (lane == 0) { ml[0] = M; ml[1] = L; }
        op[lane] = O;
        if (a.trace && tid == 0) a.trace[(h * ATTN_NCHUNK + c) * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

// (chunk, head) of a block.  Workgroups go to the 8 XCDs round-robin in dispatch order (observed, never relied upon: MI355X guide,
// "Workgroup dispatch"), i.e. block L = x + 16 y lands on XCD L % 8.  With the plain role (chunk x, head y) the 16 blocks of a head sit
// on all eight XCDs and the q/k/v exchange among them crosses the fabric; remapped, XCD i runs heads i and i + 8 (two times 16 blocks):
// publisher and poller of a granule share an L2 (hand-off +0.1-0.3 us cheaper per hop, guide row handoff-1to1).  A speed choice only:
// the exchange protocol does not care where a block runs.
__device__ __forceinline__ void qkv_block_role(const QkvAttnArgs& a, int& c, int& h) {
    c = blockIdx.x; h = blockIdx.y;
    if (a.xcd_local && (gridDim.y & 7) == 0) {
        const int L = blockIdx.x + ATTN_NCHUNK * blockIdx.y, xcd = L & 7, j = L >> 3;
        h = xcd + 8 * (j / ATTN_NCHUNK); c = j % ATTN_NCHUNK;
    }
}

template <int PRO, typename HT = bf16_t>
__global__ __launch_bounds__(256) void qkv_attn_kernel(QkvAttnArgs a) {
    QkvOperandsT<QkvStore<HT>::NP> op;
    int c, h;
    qkv_block_role(a, c, h);
    qkv_attn_body<PRO, false, HT>(a, c, h, gridDim.y, blockIdx.z, nullptr, op);
}

template <typename HT>
inline hipError_t launch_qkv_attn(const QkvAttnArgs& a, int heads, int batch, hipStream_t s) {
    if (a.hidden != 1024 || heads * 64 != a.hidden) return hipErrorInvalidValue;
    const dim3 grid(ATTN_NCHUNK, heads, batch);
    if (a.ln_g) hipLaunchKernelGGL((qkv_attn_kernel<PRO_LN, HT>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((qkv_attn_kernel<PRO_PLAIN, HT>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace ma
