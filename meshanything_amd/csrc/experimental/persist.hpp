// Persistent decode step: the whole batch-1 step (embedding -> 24 OPT layers -> lm_head -> greedy pick) as ONE launch.
//
// Reference path: the same 123 reference calls as the launch chain (shape_opt.py:318-364, 403-410, 155; [3p] OPTDecoderLayer,
// GenerationMixin greedy).  The launch chain (engine.hip enqueue_decode_step) pays, per dependent op, a kernel boundary
// (~1.35 us), a launch ramp, and a memory round trip for the input vector that queues behind the op's own weight stream.
// Here one workgroup per CU stays resident for the whole step (recipe: MI355X guide, section "Persistent kernels" price list:
// engine-vs-launches, prefetch-credit, allgather, nt-weights):
//   wave 0    LOADER   streams this CU's weight rows of every op, in consumption order, with LDS-DMA (global_load_lds
//                      dwordx4 nt) into a ring of 16 x 8 KiB units.  Weights depend on nothing, so the loader runs AHEAD
//                      across every dependency edge (up to 1.3 layers): when an op's input arrives its weights are in LDS.
//   wave 1    COMM     gathers each op's input vector: one relaxed sc1 sweep over 8-byte {tag, value} granules (the data is
//                      the flag: no fence, no separate flag word), re-polled with s_sleep until every tag carries the edge's
//                      epoch; parks the values in LDS and bumps an LDS counter.  It starts polling an edge only after its
//                      own CU has published its share (no polling under the local compute).
//   waves 2-5 COMPUTE  the four waves of the launch-chain kernels: same lane -> element mapping, same fmaf chains, same DPP
//                      reductions, same summation order -> the step's logits are bit-identical to the launch chain's.  Each
//                      wave publishes its own output rows as granules (one sc1 8-byte store per value, or per bf16 pair).
// CU roles: block b sits (observed, not relied upon) on XCD b % 8; head h = 2 (b % 8) + (b / 8) / 16, chunk c = (b / 8) % 16:
// the 16 CUs of a head share an XCD, so the two per-head exchanges stay inside one L2.
// Edges per layer (6): y2 -> all (fp32) | q,k,v -> the head's 16 CUs | split-KV partials -> the head's 16 CUs | merged
// attention output -> all (bf16 pairs) | y1 -> all (fp32) | relu(fc1) -> all (bf16 pairs); per step one more for the
// per-wave argmax partials.  Epoch of an edge = serial * 32 + k (serial: device word bumped once per launch, never reset), so
// granule buffers are never cleared and a stale tag can never match.
// Every wait is bounded (PS_TIMEOUT_TICKS of the 100 MHz real-time counter): on expiry the wave raises the block's abort word
// and the engine's error word and leaves; the host turns that into an error after the burst.
// Eligibility (engine.hip persist_eligible): bf16 policy, batch 1, greedy, hidden 1024 / ffn 4096 / 16 heads, 256 CUs.
#pragma once
#include "../attn_decode.hpp"
#include "../common.hpp"
#include "../gemv.hpp"
#include "../misc.hpp"
#include "../state.hpp"

namespace ma {

constexpr int PS_CUS = 256, PS_H = 1024, PS_F = 4096, PS_HEADS = 16, PS_THREADS = 384;
constexpr int PS_UNIT = 8192, PS_RING = 16, PS_UNITS_LAYER = 12, PS_UNITS_HEAD = 9;

// granule buffer (8-byte words)
constexpr int PG_Y2 = 0, PG_Y1 = 1024, PG_QKV = 2048, PG_PART = 5120, PG_A = PG_PART + PS_HEADS * ATTN_NCHUNK * 66, PG_FFN = PG_A + 512,
              PG_ARG = PG_FFN + 2048, PG_TOTAL = PG_ARG + 2048;
// LDS (bytes)
constexpr unsigned PL_RING = 0, PL_XRAW = PS_RING * PS_UNIT, PL_XB = PL_XRAW + 4096, PL_QKVG = PL_XB + 8192, PL_PART = PL_QKVG + 768,
                   PL_MERGE = PL_PART + ATTN_NCHUNK * 66 * 4, PL_CTRL = PL_MERGE + (unsigned)sizeof(AttnMergeLds<bf16_t>), PL_TOTAL = PL_CTRL + 64;
static_assert(PL_TOTAL <= 160 * 1024, "persistent kernel LDS budget");
enum { PC_LANDED = 0, PC_SIG = 1, PC_GATHERED = 2, PC_PUBLISHED = 3, PC_CBAR = 4, PC_ABORT = 5 };
enum { PS_ERR_LOADER = 1, PS_ERR_COMM = 2, PS_ERR_COMPUTE = 4, PS_ERR_GATHER = 8 };
constexpr int PS_TRACE_EVENTS = 320;                           // per block, comm wave: 2 stamps per edge + 2
constexpr int PS_TRACE2_EVENTS = 512;                          // per block, compute wave 0: 4 per weight op + 3 per attention

struct PersistArgs {
    const DecLayerPtrs* layers; int L;
    const bf16_t* lm_head; int V;
    const float *embtab, *extra, *tokpos, *cond, *postab; int T;
    bf16_t* kv; size_t kv_plane; int max_seq;                  // plane (2 l) = K of layer l, (2 l + 1) = V; elements
    DecState* st; long long* tokens_out;
    float* logits;
    u64* gran; unsigned* serial; unsigned* err;
    u64* trace;                                                // optional [PS_CUS][PS_TRACE_EVENTS] then [PS_CUS][PS_TRACE2_EVENTS]
};

typedef const __attribute__((address_space(1))) float* gcf;
typedef const __attribute__((address_space(1))) f32x4* gcf4;
#define PS_G(p) ((gcf)(p))

// ---- small primitives ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned ps_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ps_fail(unsigned* ctrl, unsigned* gerr, unsigned code) {
    __hip_atomic_store(ctrl + PC_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_or(gerr, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until ctrl[idx] >= target (LDS poll).  false: abort raised or timeout.
__device__ __forceinline__ bool ps_wait(unsigned* ctrl, int idx, unsigned target, u64 t0, unsigned* gerr, unsigned code) {
    unsigned spins = 0;
    while (ps_ld(ctrl + idx) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 127u) == 0) {
            if (ps_ld(ctrl + PC_ABORT)) return false;
            if (__builtin_amdgcn_s_memrealtime() - t0 > PS_TIMEOUT_TICKS) { ps_fail(ctrl, gerr, code); return false; }
        }
    }
    return true;
}
// LDS counter += 1 by one lane, after this wave's earlier LDS traffic (in-order per wave; the asm keeps the compiler from moving it)
__device__ __forceinline__ void ps_bump(unsigned* ctrl, int idx, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(ctrl + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void ps_glds16(const void* gsrc, unsigned lds_dst) {      // one 1 KiB LDS-DMA piece (MI355X guide 5.7)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned ps_pack2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }

// units freed once `ops` weight ops of the step are complete (4 per layer: qkv 3 units, out_proj 1, fc1 4, fc2 4; then lm_head 9)
__device__ __forceinline__ int ps_units_freed(int ops, int L) {
    const int full = ops >> 2, r = ops & 3;
    if (full >= L) return L * PS_UNITS_LAYER + (ops > 4 * L ? PS_UNITS_HEAD : 0);
    return full * PS_UNITS_LAYER + (r == 0 ? 0 : r == 1 ? 3 : r == 2 ? 4 : 8);
}

// ---- wave 0: loader -----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ps_loader(const PersistArgs& a, unsigned* ctrl, unsigned lds0, int b, int h, int c, int lane, u64 t0) {
    const int units = a.L * PS_UNITS_LAYER + PS_UNITS_HEAD;
    const int extra_rows = a.V - PS_CUS * 32;
    unsigned landed = 0;
    for (int u = 0; u < units; ++u) {
        if (u >= PS_RING) {                                   // ring space: unit u reuses the slot of unit u - 16
            unsigned spins = 0;
            while (u - ps_units_freed((int)(ps_ld(ctrl + PC_SIG) >> 2), a.L) >= PS_RING) {
                if (landed < (unsigned)u) {                   // blocked: everything issued so far may as well be published
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    landed = (unsigned)u;
                    if (lane == 0) __hip_atomic_store(ctrl + PC_LANDED, landed, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                __builtin_amdgcn_s_sleep(2);
                if ((++spins & 127u) == 0) {
                    if (ps_ld(ctrl + PC_ABORT)) return;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > PS_TIMEOUT_TICKS) { ps_fail(ctrl, a.err, PS_ERR_LOADER); return; }
                }
            }
        }
        // source of unit u: 8 KiB of contiguous rows of one matrix
        const char* src;
        int valid = PS_UNIT;
        if (u < a.L * PS_UNITS_LAYER) {
            const int l = u / PS_UNITS_LAYER, k = u - l * PS_UNITS_LAYER;
            const DecLayerPtrs& w = a.layers[l];
            if (k < 3) src = reinterpret_cast<const char*>(w.qkv_w) + (size_t)(k * PS_H + 64 * h + 4 * c) * PS_H * 2;
            else if (k == 3) src = reinterpret_cast<const char*>(w.o_w) + (size_t)(4 * b) * PS_H * 2;
            else if (k < 8) src = reinterpret_cast<const char*>(w.fc1_w) + (size_t)(16 * b + 4 * (k - 4)) * PS_H * 2;
            else src = reinterpret_cast<const char*>(w.fc2_w) + (size_t)(4 * b + (k - 8)) * PS_F * 2;
        } else {
            const int k = u - a.L * PS_UNITS_LAYER;
            if (k < 8) src = reinterpret_cast<const char*>(a.lm_head) + (size_t)(32 * b + 4 * k) * PS_H * 2;
            else { src = reinterpret_cast<const char*>(a.lm_head) + (size_t)(b < extra_rows ? PS_CUS * 32 + b : 0) * PS_H * 2; valid = PS_H * 2; }
        }
        const unsigned dst = lds0 + PL_RING + (unsigned)(u % PS_RING) * PS_UNIT;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int off = p * 1024 < valid ? p * 1024 : (p & 1) * 1024;     // the one-row unit re-reads its row (never past the matrix)
            ps_glds16(src + off + lane * 16, __builtin_amdgcn_readfirstlane(dst + p * 1024));
        }
        if (u >= 2) {                                          // three units in flight: unit u - 2 has landed
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            if ((unsigned)(u - 1) > landed) {
                landed = (unsigned)(u - 1);
                if (lane == 0) __hip_atomic_store(ctrl + PC_LANDED, landed, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(ctrl + PC_LANDED, (unsigned)units, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- wave 1: comm -------------------------------------------------------------------------------------------------------------
// sweep n <= NK * 64 granules (granule g of the edge lives at gran[src(g)]) until every tag == epoch; value g -> dst[g] (LDS
// words).  A pass issues the loads of every still-pending 64-granule group before it looks at any of them (one memory round
// trip per pass, however long the edge is); groups that are complete are not read again.
template <int NK, typename SrcF>
__device__ __forceinline__ bool ps_gather(const u64* gran, SrcF src, int n, unsigned epoch, unsigned* dst, int lane, unsigned* ctrl, u64 t0,
                                          unsigned* gerr) {
    const gu64* g64 = (const gu64*)gran;
    unsigned pend = NK >= 32 ? 0xffffffffu : ((1u << NK) - 1u);
    unsigned spins = 0;
    for (;;) {
        u64 v[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int g = k * 64 + lane;
            v[k] = (u64)epoch << 32;
            if (((pend >> k) & 1u) && g < n) v[k] = __hip_atomic_load(g64 + src(g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            if ((pend >> k) & 1u) {
                const int g = k * 64 + lane;
                const bool ok = (unsigned)(v[k] >> 32) == epoch;
                if (ok && g < n) dst[g] = (unsigned)v[k];
                if (__all(ok)) pend &= ~(1u << k);
            }
        }
        if (!pend) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 31u) == 0) {
            if (ps_ld(ctrl + PC_ABORT)) return false;
            if (__builtin_amdgcn_s_memrealtime() - t0 > PS_TIMEOUT_TICKS) { ps_fail(ctrl, gerr, PS_ERR_GATHER); return false; }
        }
    }
}

struct PsTrace {
    u64* p; int n; int cap;
    __device__ __forceinline__ void stamp(int lane) { if (p && lane == 0 && n < cap) p[n] = __builtin_amdgcn_s_memrealtime(); ++n; }
};

__device__ __forceinline__ void ps_comm(const PersistArgs& a, char* smem, unsigned* ctrl, int b, int h, int c, int lane, u64 t0, const DecState& sv,
                        unsigned serial) {
    unsigned* xraw = reinterpret_cast<unsigned*>(smem + PL_XRAW);
    unsigned* xb = reinterpret_cast<unsigned*>(smem + PL_XB);
    unsigned* qkvg = reinterpret_cast<unsigned*>(smem + PL_QKVG);
    unsigned* part = reinterpret_cast<unsigned*>(smem + PL_PART);
    PsTrace tr{a.trace ? a.trace + (size_t)b * PS_TRACE_EVENTS : nullptr, 0, PS_TRACE_EVENTS};
    tr.stamp(lane);
    unsigned pub = 0, gathered = 0;
    const int pos = sv.pos, len = pos + 1, per = (len + ATTN_NCHUNK - 1) / ATTN_NCHUNK, c_last = (len - 1) / per;
    // one edge: wait for the local share, sweep, announce
#define PS_EDGE(PUBS, SRC, N, EPOCH, DST)                                                                                 \
    do {                                                                                                                       \
        pub += (PUBS);                                                                                                         \
        if (!ps_wait(ctrl, PC_PUBLISHED, pub, t0, a.err, PS_ERR_COMM)) return;                                                 \
        tr.stamp(lane);                                                                                                        \
        if (!ps_gather<((N) + 63) / 64>(a.gran, SRC, N, EPOCH, DST, lane, ctrl, t0, a.err)) return;                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                     \
        ++gathered;                                                                                                            \
        if (lane == 0) __hip_atomic_store(ctrl + PC_GATHERED, gathered, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);       \
        tr.stamp(lane);                                                                                                        \
    } while (0)
    for (int l = 0; l < a.L; ++l) {
        const unsigned ep = serial * 32u + (unsigned)l + 1u;
        if (l > 0) PS_EDGE(4, [](int g) { return PG_Y2 + g; }, PS_H, ep, xraw);
        PS_EDGE(4, [h](int g) { return PG_QKV + (g >> 6) * PS_H + 64 * h + (g & 63); }, 192, ep, qkvg);
        if (c == c_last && lane < 32) {
            // the newest position's K / V rows of this head go to the cache for later steps (write-through: a later step of a
            // multi-step launch would read them from another CU); this step's attention takes them from LDS
            const float* src = reinterpret_cast<const float*>(qkvg) + 64 + (lane >> 4) * 64 + (lane & 15) * 4;
            const u64 v = (u64)ps_pack2(src[0], src[1]) | ((u64)ps_pack2(src[2], src[3]) << 32);
            bf16_t* plane = a.kv + (size_t)(2 * l + (lane >> 4)) * a.kv_plane + ((size_t)h * a.max_seq + pos) * 64 + (lane & 15) * 4;
            __hip_atomic_store((gu64*)plane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        PS_EDGE(1, [h](int g) { return PG_PART + h * ATTN_NCHUNK * 66 + g; }, ATTN_NCHUNK * 66, ep, part);
        PS_EDGE(1, [](int g) { return PG_A + g; }, 512, ep, xb);
        PS_EDGE(4, [](int g) { return PG_Y1 + g; }, PS_H, ep, xraw);
        PS_EDGE(4, [](int g) { return PG_FFN + g; }, 2048, ep, xb);
    }
    {
        const unsigned ep = serial * 32u + (unsigned)a.L + 1u;
        PS_EDGE(4, [](int g) { return PG_Y2 + g; }, PS_H, ep, xraw);
        PS_EDGE(4, [](int g) { return PG_ARG + g; }, 2048, ep + 1u, xb);
    }
#undef PS_EDGE
    // greedy pick over the 1024 per-wave partials (value, index): larger value wins, ties -> lower index
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int i = lane; i < 1024; i += 64) {
        const float v = __uint_as_float(xb[2 * i]); const int ix = (int)xb[2 * i + 1];
        if (arg_better(v, ix, bv, bi)) { bv = v; bi = ix; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (arg_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (b == 0 && lane == 0) {                                 // generate() bookkeeping, as pick_kernel
        const int t = sv.t;
        int tok = bi;
        if (sv.finished) tok = TOK_PAD;
        if (t < sv.max_new) a.tokens_out[t] = tok;
        if (tok == TOK_EOS) a.st->finished = 1;
        a.st->cur_tok = tok;
        a.st->t = t + 1;
        a.st->pos = a.T + t;
        *a.serial = serial + 1u;
    }
    tr.stamp(lane);
}

// ---- waves 2-5: compute -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ps_unpack8(const u32x4& r, float (&f)[8]) {
    f[0] = bf_lo(r.x); f[1] = bf_hi(r.x); f[2] = bf_lo(r.y); f[3] = bf_hi(r.y); f[4] = bf_lo(r.z); f[5] = bf_hi(r.z); f[6] = bf_lo(r.w); f[7] = bf_hi(r.w);
}
// lane partial of one 512-element piece: weights (16 bytes at row + (piece * 64 + lane) * 16) . x (8 floats), fmaf chain as gemv_kernel
__device__ __forceinline__ float ps_piece(const char* row, int piece, int lane, const float* xs, float acc) {
    const u32x4 w = *reinterpret_cast<const u32x4*>(row + (piece * 64 + lane) * 16);
    float wf[8];
    unpack16<bf16_t>(w, wf);
#pragma unroll
    for (int v = 0; v < 8; ++v) acc = fmaf(wf[v], xs[v], acc);
    return acc;
}
// K = 1024, x in registers, KSPLIT = 1 / LPL = 2 shape (qkv, fc1, lm_head): wave_sum(pieces 0 and 1 in one chain)
__device__ __forceinline__ float ps_row_k1024(const char* row, int lane, const float (&xs)[16]) {
    float acc = ps_piece(row, 0, lane, xs, 0.f);
    acc = ps_piece(row, 1, lane, xs + 8, acc);
    return wave_sum(acc);
}

// LayerNorm of the gathered row in xraw (4 KB fp32) as the block prologue of gemv_kernel computes it (common.hpp pieces):
// statistics over the four "virtual waves" of 64 float4 chunks, then this lane's 16 elements ((i * 64 + lane) * 8 + v)
// normalised and rounded to bf16; *res = the fp32 normalised element `rn` (a later epilogue's residual).
// the LayerNorm affine parameters this lane needs (its 16 elements + element rn), fetched BEFORE the wave waits for the row
struct PsLnPar { f32x4 g[4], b[4]; float g_rn, b_rn; };
__device__ __forceinline__ void ps_ln_prefetch(const float* g, const float* bta, int lane, int rn, PsLnPar& P) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int idx = 2 * (i * 64 + lane) + hh;
            P.g[i * 2 + hh] = *(gcf4)(g + idx * 4); P.b[i * 2 + hh] = *(gcf4)(bta + idx * 4);
        }
    P.g_rn = PS_G(g)[rn]; P.b_rn = PS_G(bta)[rn];
}
__device__ __forceinline__ void ps_layernorm(const char* smem, const PsLnPar& P, int lane, int rn, float (&xs)[16], float& res) {
    const f32x4* x4 = reinterpret_cast<const f32x4*>(smem + PL_XRAW);
    const float* x1 = reinterpret_cast<const float*>(smem + PL_XRAW);
    const float x0 = x1[0];
    float s[4], q[4];
#pragma unroll
    for (int vw = 0; vw < 4; ++vw) {
        f32x4 x = x4[vw * 64 + lane];
        float ss = 0.f, qq = 0.f;
        ln_chunk_moments(x, x0, ss, qq);
        s[vw] = wave_sum(ss); q[vw] = wave_sum(qq);
    }
    float md, rstd;
    ln_finish(s[0], s[1], s[2], s[3], q[0], q[1], q[2], q[3], PS_H, 1e-5f, md, rstd);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int idx = 2 * (i * 64 + lane) + hh;
            f32x4 x = x4[idx];
            x.x -= x0; x.y -= x0; x.z -= x0; x.w -= x0;
            ln_apply(x, md, rstd, P.g[i * 2 + hh], P.b[i * 2 + hh]);
            xs[i * 8 + hh * 4 + 0] = round_bf16(x.x); xs[i * 8 + hh * 4 + 1] = round_bf16(x.y);
            xs[i * 8 + hh * 4 + 2] = round_bf16(x.z); xs[i * 8 + hh * 4 + 3] = round_bf16(x.w);
        }
    res = ln_apply1(x1[rn] - x0, md, rstd, P.g_rn, P.b_rn);
}


typedef __attribute__((address_space(3))) char* lptr;
typedef const __attribute__((address_space(1))) bf16_t* gkv;

// The attention section of one layer for one compute wave, one function (inlined: hipcc only honours register budgets on kernels, and an
// out-of-line copy took 248 VGPRs + scratch).  Issues round 0 before q arrives, waits for the
// q / k / v gather, reduces this wave's positions, takes part in the two-level block merge (two compute-wave barriers) and,
// on wave 0, publishes the (m, l, o[64]) partial of (head, chunk) as 66 granules at g0.  false: a bounded wait expired.
__device__ __forceinline__ bool ps_attention(lptr lds, gkv kplane_h, gkv vplane_h, gu64* gran_g, unsigned* gerr, unsigned ep, int g0, int c, int pos,
                                          int cw, int lane, unsigned need_g, unsigned need_bar, u64 t0) {
    typedef AttnGeom<bf16_t> G;
    char* smem = (char*)lds;
    unsigned* ctrl = reinterpret_cast<unsigned*>(smem + PL_CTRL);
    u64* gran = (u64*)gran_g;
    const int len = pos + 1, per = (len + ATTN_NCHUNK - 1) / ATTN_NCHUNK, start = c * per, end = min(len, start + per);
    const int nround = (max(end - start, 0) + 127) >> 7;
    const int slot = lane / G::LPP, dsub = lane % G::LPP, woff = cw * 32;
    const bf16_t* kh = (const bf16_t*)kplane_h + dsub * G::EPL;
    const bf16_t* vh = (const bf16_t*)vplane_h + dsub * G::EPL;
    u32x4 kA[G::U], vA[G::U], kB[G::U], vB[G::U];
    auto issue = [&](int r, u32x4 (&kr)[G::U], u32x4 (&vr)[G::U]) {
        const int base = start + (r << 7) + woff + slot;
#pragma unroll
        for (int u = 0; u < G::U; ++u) { const int p = base + u * G::PPW; kr[u] = ld_stream16(kh + (size_t)(p < end ? p : start) * 64); }
#pragma unroll
        for (int u = 0; u < G::U; ++u) { const int p = base + u * G::PPW; vr[u] = ld_stream16(vh + (size_t)(p < end ? p : start) * 64); }
    };
    if (nround > 0) issue(0, kA, vA);                 // the cached rows do not depend on q: in flight while q travels
    asm volatile("" ::: "memory");                    // (keeps hipcc from sinking the loads below the wait)
    if (!ps_wait(ctrl, PC_GATHERED, need_g + 1, t0, gerr, PS_ERR_COMPUTE)) return false;
    const float* qg = reinterpret_cast<const float*>(smem + PL_QKVG);
    float qv[G::EPL];
#pragma unroll
    for (int e = 0; e < G::EPL; ++e) qv[e] = round_bf16(qg[dsub * G::EPL + e]);
    u32x4 ok, ov;                                      // the newest position's rows (not in the cache yet), this lane's 8 dims
    {
        const float* kn = qg + 64 + dsub * G::EPL; const float* vn = qg + 128 + dsub * G::EPL;
        ok = u32x4{ps_pack2(kn[0], kn[1]), ps_pack2(kn[2], kn[3]), ps_pack2(kn[4], kn[5]), ps_pack2(kn[6], kn[7])};
        ov = u32x4{ps_pack2(vn[0], vn[1]), ps_pack2(vn[2], vn[3]), ps_pack2(vn[4], vn[5]), ps_pack2(vn[6], vn[7])};
    }
    AttnSlotState<bf16_t> ss;
    ss.m = -1e30f; ss.l = 0.f;
#pragma unroll
    for (int e = 0; e < G::EPL; ++e) ss.o[e] = 0.f;
    for (int r = 0; r < nround; r += 2) {
        if (r + 1 < nround) issue(r + 1, kB, vB);
        attn_round_reduce<bf16_t, true>(ss, qv, kA, vA, start + (r << 7) + woff + slot, end, pos, ok, ov);
        if (r + 1 < nround) {
            if (r + 2 < nround) issue(r + 2, kA, vA);
            attn_round_reduce<bf16_t, true>(ss, qv, kB, vB, start + ((r + 1) << 7) + woff + slot, end, pos, ok, ov);
        }
    }
    AttnMergeLds<bf16_t>& S = *reinterpret_cast<AttnMergeLds<bf16_t>*>(smem + PL_MERGE);
    const int gs = cw * G::PPW + slot;
    if (dsub == 0) { S.sm[gs] = ss.m; S.sl[gs] = ss.l; }
#pragma unroll
    for (int e = 0; e < G::EPL; ++e) S.so[gs][dsub * G::EPL + e] = ss.o[e];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(ctrl + PC_CBAR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (!ps_wait(ctrl, PC_CBAR, need_bar + 4, t0, gerr, PS_ERR_COMPUTE)) return false;
    {
        float M, Lq, O;
        attn_fold_quarter<bf16_t>(S, cw, lane, M, Lq, O);
        if (lane == 0) { S.qm[cw] = M; S.ql[cw] = Lq; }
        S.qo[cw][lane] = O;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(ctrl + PC_CBAR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (!ps_wait(ctrl, PC_CBAR, need_bar + 8, t0, gerr, PS_ERR_COMPUTE)) return false;
    if (cw == 0) {
        float M, Lq, O;
        attn_fold_block<bf16_t>(S, lane, M, Lq, O);
        ps_publish(gran, g0 + 2 + lane, ep, __float_as_uint(O));
        if (lane < 2) ps_publish(gran, g0 + lane, ep, __float_as_uint(lane == 0 ? M : Lq));
    }
    return true;
}

__device__ __forceinline__ void ps_compute(const PersistArgs& a, char* smem, unsigned* ctrl, int b, int h, int c, int cw, int lane, u64 t0, const DecState& sv,
                           unsigned serial) {
    typedef AttnGeom<bf16_t> G;
    const int pos = sv.pos, tstep = sv.t, tok = sv.cur_tok;
    const int rn = 4 * b + cw;                                // the output row this wave owns in the N = hidden ops
    unsigned need_g = 0, need_bar = 0;                        // gathers consumed / compute barriers passed so far
    PsTrace tr{a.trace && cw == 0 ? a.trace + (size_t)PS_CUS * PS_TRACE_EVENTS + (size_t)b * PS_TRACE2_EVENTS : nullptr, 0, PS_TRACE2_EVENTS};
    PsLnPar lnp;
    int ubase = 0;                                             // first ring unit of the current layer
    float xs[16], resid;
    auto ring = [&](int u) -> const char* { return smem + PL_RING + (u % PS_RING) * PS_UNIT; };
#define PS_WAIT_G() do { if (!ps_wait(ctrl, PC_GATHERED, ++need_g, t0, a.err, PS_ERR_COMPUTE)) return; } while (0)
#define PS_WAIT_U(U) do { if (!ps_wait(ctrl, PC_LANDED, (unsigned)(U) + 1u, t0, a.err, PS_ERR_COMPUTE)) return; } while (0)
#define PS_CBAR() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
                       if (lane == 0) __hip_atomic_fetch_add(ctrl + PC_CBAR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);       \
                       need_bar += 4; if (!ps_wait(ctrl, PC_CBAR, need_bar, t0, a.err, PS_ERR_COMPUTE)) return; } while (0)

    // ---- step input: the token embedding (shape_opt.py:237-245, 323-328, 359-364, 453-460), this lane's 16 elements ----------
    {
        const bool special = tok < 3;
        int m = (tstep - 2) % 9; if (m < 0) m += 9;
        const int slot = special ? tok : m + 3;
        const float* base = special ? a.extra + (size_t)tok * PS_H : a.embtab + (size_t)(tok - 3) * PS_H;
        const float* tp = a.tokpos + (size_t)slot * PS_H;
        const float* cd = a.cond + PS_H;
        const float* pt = a.postab + (size_t)(a.T + tstep - 1 + 2) * PS_H;
        auto emb = [&](int k) { float e = base[k]; e += tp[k]; e += cd[k]; e += pt[k]; return e; };
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int v = 0; v < 8; ++v) xs[i * 8 + v] = round_bf16(emb((i * 64 + lane) * 8 + v));
        resid = emb(rn);
    }

    for (int l = 0; l < a.L; ++l, ubase += PS_UNITS_LAYER) {
        const DecLayerPtrs& w = a.layers[l];
        const unsigned ep = serial * 32u + (unsigned)l + 1u;
        // ---- q, k, v rows 64 h + 4 c + cw of the fused [3H][H] matrix ([3p] OPTAttention projections) ---------------------
        {
            const int row = 64 * h + 4 * c + cw;
            float bq[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) bq[p] = PS_G(w.qkv_b)[p * PS_H + row];
            if (l > 0) {
                ps_ln_prefetch(a.layers[l - 1].ln2_g, a.layers[l - 1].ln2_b, lane, rn, lnp);
                PS_WAIT_G();
                ps_layernorm(smem, lnp, lane, rn, xs, resid);
            }
            tr.stamp(lane);
            float v3[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                PS_WAIT_U(ubase + p);
                if (p == 0) tr.stamp(lane);
                v3[p] = ps_row_k1024(ring(ubase + p) + cw * 2048, lane, xs) + bq[p];
            }
            ps_bump(ctrl, PC_SIG, lane);
            tr.stamp(lane);
            if (lane < 3) ps_publish(a.gran, PG_QKV + lane * PS_H + row, ep, __float_as_uint(lane == 0 ? v3[0] : lane == 1 ? v3[1] : v3[2]));
            ps_bump(ctrl, PC_PUBLISHED, lane);
            tr.stamp(lane);
        }
        // ---- attention over the cache, split-KV chunk c of head h (attn_decode.hpp; the 4 compute waves = its 4 waves) ------
        {
            const bf16_t* kh = a.kv + (size_t)(2 * l) * a.kv_plane + (size_t)h * a.max_seq * 64;
            const bf16_t* vh = a.kv + (size_t)(2 * l + 1) * a.kv_plane + (size_t)h * a.max_seq * 64;
            if (!ps_attention((lptr)smem, (gkv)kh, (gkv)vh, (gu64*)a.gran, a.err, ep, PG_PART + (h * ATTN_NCHUNK + c) * 66, c, pos, cw, lane, need_g, need_bar, t0)) return;
            need_g += 1; need_bar += 8;
            tr.stamp(lane);
            if (cw == 0) {
                ps_bump(ctrl, PC_PUBLISHED, lane);
                // ---- merge of the head's 16 partials (gemv.hpp PRO_ATTN arithmetic), elements 4 c .. 4 c + 3 of head h -----------
                PS_WAIT_G();
                tr.stamp(lane);
                {   // merge of elements 4 c .. 4 c + 3 (lane d = lane & 3 owns element 4 c + d): the arithmetic and ORDER of
                    // attn_partials_merge -- max over chunks, then L and O as fmaf chains over chunk 0..15 -- with the sixteen
                    // exp() and all LDS reads issued side by side (lane cc < 16 holds chunk cc's m, l and weight)
                    const float* pw = reinterpret_cast<const float*>(smem + PL_PART);
                    const int cc = lane & 15, d = 4 * c + (lane & 3);
                    const float mc = pw[cc * 66], lc = pw[cc * 66 + 1];
                    float oc[ATTN_NCHUNK];
#pragma unroll
                    for (int k = 0; k < ATTN_NCHUNK; ++k) oc[k] = pw[k * 66 + 2 + d];
                    const float M = row_max16(mc);                       // the 16 lanes of a DPP row hold the 16 chunks: exact, order-free
                    const float fc = expf(mc - M);
                    float Ls = 0.f, Os = 0.f;
#pragma unroll
                    for (int k = 0; k < ATTN_NCHUNK; ++k) {
                        const float f = readlane_f(fc, k);
                        Ls = fmaf(readlane_f(lc, k), f, Ls);
                        Os = fmaf(oc[k], f, Os);
                    }
                    const float inv = 1.0f / Ls;
                    const float av = Os * inv;
                    const float a0 = readlane_f(av, 0), a1 = readlane_f(av, 1), a2 = readlane_f(av, 2), a3 = readlane_f(av, 3);
                    if (lane < 2) ps_publish(a.gran, PG_A + (64 * h + 4 * c) / 2 + lane, ep, lane == 0 ? ps_pack2(a0, a1) : ps_pack2(a2, a3));
                }
                ps_bump(ctrl, PC_PUBLISHED, lane);
                tr.stamp(lane);
            } else {
                ++need_g;                                      // the partials' gather is consumed by wave 0 only
            }
        }
        // ---- y1 = resid + Wo a + bo, row rn (gemv shape {1, 2, 1}: both 512-element pieces in one chain, one reduction) --------------
        {
            const float bo = PS_G(w.o_b)[rn];
            PS_WAIT_G();
            tr.stamp(lane);
            PS_WAIT_U(ubase + 3);
            tr.stamp(lane);
            const char* row = ring(ubase + 3) + cw * 2048;
            float acc = 0.f;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float xa[8];
                ps_unpack8(*reinterpret_cast<const u32x4*>(smem + PL_XB + (p * 64 + lane) * 16), xa);
                acc = ps_piece(row, p, lane, xa, acc);
            }
            const float t0s = wave_sum(acc);
            ps_bump(ctrl, PC_SIG, lane);
            tr.stamp(lane);
            float v = t0s;
            v += bo;
            v += resid;
            if (lane == 0) ps_publish(a.gran, PG_Y1 + rn, ep, __float_as_uint(v));
            ps_bump(ctrl, PC_PUBLISHED, lane);
            tr.stamp(lane);
        }
        // ---- f = relu(W1 LN1(y1) + b1), rows 16 b + 4 cw .. + 3 -------------------------------------------------------------------
        {
            float b1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b1[j] = PS_G(w.fc1_b)[16 * b + 4 * cw + j];
            ps_ln_prefetch(w.ln1_g, w.ln1_b, lane, rn, lnp);
            PS_WAIT_G();
            ps_layernorm(smem, lnp, lane, rn, xs, resid);
            tr.stamp(lane);
            PS_WAIT_U(ubase + 4 + cw);
            tr.stamp(lane);
            const char* u4 = ring(ubase + 4 + cw);
            float f[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = fmaxf(ps_row_k1024(u4 + j * 2048, lane, xs) + b1[j], 0.f);
            ps_bump(ctrl, PC_SIG, lane);
            tr.stamp(lane);
            if (lane < 2) ps_publish(a.gran, PG_FFN + (16 * b + 4 * cw) / 2 + lane, ep, lane == 0 ? ps_pack2(f[0], f[1]) : ps_pack2(f[2], f[3]));
            ps_bump(ctrl, PC_PUBLISHED, lane);
            tr.stamp(lane);
        }
        // ---- y2 = h1 + W2 f + b2, row rn (gemv shape {1, 8, 1}: the eight 512-element pieces in one chain, one reduction) ------------
        {
            const float b2 = PS_G(w.fc2_b)[rn];
            PS_WAIT_G();
            tr.stamp(lane);
            PS_WAIT_U(ubase + 8 + cw);
            tr.stamp(lane);
            const char* row = ring(ubase + 8 + cw);
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float xa[8];
                ps_unpack8(*reinterpret_cast<const u32x4*>(smem + PL_XB + (i * 64 + lane) * 16), xa);
                acc = ps_piece(row, i, lane, xa, acc);
            }
            float v = wave_sum(acc);
            ps_bump(ctrl, PC_SIG, lane);
            tr.stamp(lane);
            v += b2;
            v += resid;
            if (lane == 0) ps_publish(a.gran, PG_Y2 + rn, serial * 32u + (unsigned)l + 2u, __float_as_uint(v));
            ps_bump(ctrl, PC_PUBLISHED, lane);
            tr.stamp(lane);
        }
    }
    // ---- lm_head on LN2_{L-1}(y2) (shape_opt.py:155): rows 32 b + 8 cw .. + 7, plus row 8192 + b on the first V - 8192 CUs ---------
    {
        ps_ln_prefetch(a.layers[a.L - 1].ln2_g, a.layers[a.L - 1].ln2_b, lane, rn, lnp);
        PS_WAIT_G();
        ps_layernorm(smem, lnp, lane, rn, xs, resid);
        tr.stamp(lane);
        const int skip = sv.suppress_eos ? TOK_EOS : -1;
        float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
            PS_WAIT_U(ubase + 2 * cw + uu);
            const char* u4 = ring(ubase + 2 * cw + uu);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = 32 * b + 8 * cw + 4 * uu + j;
                const float v = ps_row_k1024(u4 + j * 2048, lane, xs);
                if (lane == 0) a.logits[n] = v;
                if (n != skip && arg_better(v, n, bv, bi)) { bv = v; bi = n; }
            }
        }
        if (cw == 0 && b < a.V - PS_CUS * 32) {
            PS_WAIT_U(ubase + 8);
            const int n = PS_CUS * 32 + b;
            const float v = ps_row_k1024(ring(ubase + 8), lane, xs);
            if (lane == 0) a.logits[n] = v;
            if (n != skip && arg_better(v, n, bv, bi)) { bv = v; bi = n; }
        }
        ps_bump(ctrl, PC_SIG, lane);
        if (lane < 2) ps_publish(a.gran, PG_ARG + 2 * (4 * b + cw) + lane, serial * 32u + (unsigned)a.L + 2u, lane == 0 ? __float_as_uint(bv) : (unsigned)bi);
        ps_bump(ctrl, PC_PUBLISHED, lane);
    }
#undef PS_WAIT_G
#undef PS_WAIT_U
#undef PS_CBAR
}

__global__ __launch_bounds__(PS_THREADS) void persist_decode_kernel(PersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) char ps_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int h = 2 * (b & 7) + ((b >> 3) >> 4), c = (b >> 3) & 15;
    unsigned* ctrl = reinterpret_cast<unsigned*>(ps_smem + PL_CTRL);
    if (tid < 16) ctrl[tid] = 0u;
    __syncthreads();
    u64 t0 = __builtin_amdgcn_s_memrealtime();
    const DecState sv = *a.st;
    const unsigned serial = *a.serial;
#ifndef ROLE_ONLY
#define ROLE_ONLY -1
#endif
    if (wv == 0) { if (ROLE_ONLY < 0 || ROLE_ONLY == 0) ps_loader(a, ctrl, (unsigned)(size_t)ps_smem, b, h, c, lane, t0); }
    else if (wv == 1) { if (ROLE_ONLY < 0 || ROLE_ONLY == 1) ps_comm(a, ps_smem, ctrl, b, h, c, lane, t0, sv, serial); }
    else { if (ROLE_ONLY < 0 || ROLE_ONLY == 2) ps_compute(a, ps_smem, ctrl, b, h, c, wv - 2, lane, t0, sv, serial); }
}

// once per process and device, outside any stream capture: the kernel asks for more than the default 64 KB of dynamic LDS
inline hipError_t persist_prepare() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(persist_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PL_TOTAL);
}
inline hipError_t launch_persist_decode(const PersistArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(persist_decode_kernel, dim3(PS_CUS), dim3(PS_THREADS), PL_TOTAL, s, a);
    return hipGetLastError();
}

}  // namespace ma
