// The first half of a decoder layer for EIGHT rows stepping together, in ONE launch: LayerNorm 2 of the previous layer (folded prologue) ->
// q/k/v projection on the matrix cores -> single-query attention over the KV cache (two blocks per (row, head)) -> out_proj + bias +
// residual.  Replaces three launches of the batched matrix-core chain (gemm_dec_ln_kernel with EPI_QKV | attn_decode_final_kernel<8, true> |
// gemm_dec_kernel for out_proj; [3p] OPTDecoderLayer.self_attn reached from shape_opt.py:403-410, OptFlashAttention2 with q_len 1 and the
// per-step torch.cat) -- the two seams of the layer whose payload is small: q/k/v -> attention is 192 values per
// (row, head) and never leaves the 16 blocks of a head; attention -> out_proj is 64 values per (row, head), 16 KB for all rows.
//
// grid (16 heads, 8 rows, 2 halves) = 256 blocks of 8 waves, one per CU (every block must be resident: the engine's `chain_resident`
// gate, bounded sweeps, the error word and the fall-back to the three-launch form are those of the other fused launches, engine.hip).
// Block (h, b, z):
//   A. requests, in consumption order: LayerNorm parameters, the vectors of batch row `wave` (the producer's split-K partials + deferred
//      bias + residual), its 16-row q/k/v weight tile (rows 4 j .. 4 j + 3 of q, k and v of head h, j = 2 b + z: the 16 blocks of a head
//      cut its 192 rows into 12 each), and ALREADY the first two rounds of its half of the (b, h) cache stream -- they do not depend on
//      anything this launch computes, so 2 x 64 KB per block are on their way while the prologue runs;
//   B. normalises the eight rows (one per wave; gemm_dec_ln_kernel's arithmetic, bit for bit), parks them in LDS as 16-bit;
//   C. 16 x 16 x 32 MFMAs: A = the weight tile straight from HBM, B = the rows from LDS, the 8 waves split K and meet in LDS (the order of
//      gemm_dec_ln_kernel<.., 8>); wave 0 adds the bias and publishes its 4 x 3 outputs of every row as {epoch, two 16-bit values} granules
//      (q rounded as the attention kernel rounds it, k / v as the cache holds them; k / v also go to the cache for later steps);
//   D. wave 0 sweeps the 96 granules of (b, h); all waves attend over the block's rounds (even / odd rounds of 256 positions for z = 1 / 0,
//      attn_round_reduce with the newest position taken from the granules), fold their states in LDS; block z = 0 hands (m, l, o[64]) to
//      block z = 1 (attn_decode_final_kernel's protocol and merge order: the attention output has the same bits as the launch it replaces);
//   E. block z = 1 publishes the normalised output of (b, h) as 32 granules; the 64 blocks (h, b < 4, z = 0) -- free since their hand-over --
//      each own one 16-row tile of out_proj (requested in step B): every wave sweeps one row's 512 granules into LDS, four waves run
//      gemm_dec_kernel<1, 8>'s K split, wave 0 adds bias + residual (the LayerNorm output of step B, kept in LDS) and writes y1.
// Epoch = position * 32 + layer + 1 (strictly increasing within a generation; the buffers are zeroed when the position restarts).
// HBM-bound: algorithmic bytes = the layer's q/k/v + out_proj matrices (8 MB) + 8 rows x 2 x len x 2 KB of cache.
#pragma once
#include "attn_decode.hpp"
#include "common.hpp"
#include "gemm_decode.hpp"
#include "state.hpp"

namespace ma {

constexpr unsigned RA_ERR_QKV = 256, RA_ERR_OUT = 512;
constexpr int RA_ROWS = 8;                                 // rows of the launch (the MFMA B tile holds 16; the grid's pairs fill 256 CUs at 8)
constexpr int RA_QKV_GRANULES = 3 * 16 * 32;               // per row: [q | k | v][head][32 pairs]
constexpr int RA_OUT_GRANULES = 16 * 32;                   // per row: [head][32 pairs]

struct RowsAttnArgs {
    const bf16_t* Wqkv; const float* bqkv;                 // fused [3 x 1024][1024] projection, [3 x 1024] bias
    const bf16_t* xb; int xb_stride;                       // layer 0 (ln_g == null): the rows as 16-bit, written by the embedding launch
    const float* pin; int pin_stride; int pin_parts; const float* pbias; const float* pres; int pres_stride;     // as GemmDecArgs (gemm_dec_ln_kernel)
    const float* ln_g; const float* ln_b; float ln_eps;
    float* xn_out; int xn_stride;                          // LayerNorm output fp32 (diagnostics / the three-launch form's residual), or null
    bf16_t* kcache; bf16_t* vcache; size_t kv_row_stride; int max_seq;
    const DecState* st; int len_override; int layer;
    u64* qkv_gran;                                         // [8][RA_QKV_GRANULES]
    u64* pair_gran;                                        // [8][16][ATTN_PAIR_GRANULES]
    u64* out_gran;                                         // [8][RA_OUT_GRANULES]
    unsigned* err;
    const bf16_t* Wo; const float* bo;                     // out_proj [1024][1024], [1024]
    const float* res; int res_stride;                      // layer 0: the residual (the embedding, fp32); otherwise the LayerNorm output of step B
    float* y1; int y1_stride;                              // out: residual + Wo a + bo, fp32 [8][1024]
    unsigned long long* trace;                             // 4 stamps per block (ma_trace_decode) or null
};

// HASLN: the rows are LayerNorm(sum of PARTS partial buffers [+ pbias + pres when DEFER]); otherwise they come as 16-bit from `xb`
// EARLY: when the first cache rounds (64 KB per block and round, 16 MB per launch) are requested.  A CU takes only ~20-25 GB/s from HBM and its
// requests leave in order, so whatever is asked for in front of the q/k/v exchange delays it by bytes / 20 GB/s: in-kernel stamps show every
// wave held at the ISSUE of its requests (profiles/r05_decode_step_timeline_b8_*.txt, r05_ab_rows_attn_request_placement.txt; step at 8 rows,
// kv 3858, same box):
//   1  both rounds in the kernel's first instructions ............ q/k/v published 8.4-10 us after the block's start, step 1067 us
//   0  once the rows' vectors have arrived ........................ the same (the 6 MB of q/k/v weights share the memory system with 32 MB of cache), 1065 us
//   2  both rounds + the out_proj tile behind the q/k/v MFMAs ..... MFMAs done at 1.4 us, but wave 0 reaches its publish 5.5 us later, 1056 us
//   4  nothing before the exchange is over ....................... exchange over at 6.0 us, the attention starts on a cold stream, 1054 us
//   3  ONE round behind the MFMAs, by the seven waves that do not publish and sweep; wave 0's own round behind its sweep; the
//      second round and the out_proj tile once the exchange is over ........................................................ 1043 us
//   5  the sweep by SCALAR loads of the waves 0 .. 3 (their own path to L2: not behind the stream) while the waves 4 .. 7 request
//      both rounds; the sweeping waves' rounds behind the exchange barrier; the out_proj tile by the 64 blocks that use it, behind the
//      stream: exchange over at 4.2 us instead of 7.7 ........................................ 1029 us where 3 gives 1051 (another box)
//   6  (default) 5 + FAST: a round that lies wholly below the newest position is reduced without masks, override and clamps
//      (~100 of ~380 instructions per round fewer, the same bits) ............. -5 .. -10 us at kv 3858, -40 us at kv 7300 against 5
// (steps A - E above describe placement 1 / the vector sweep; with 5, step D's sweep is the scalar one and the out_proj tile of step E is
//  requested after the block's last round)
template <bool HASLN, int PARTS, bool DEFER, int EARLY, int QW, typename HT>
__global__ __launch_bounds__(512) void rows_attn_kernel(RowsAttnArgs a) {
    using G = AttnGeom<HT>;
    // QW = waves that split K in the q/k/v MFMAs = the split of the launch replaced, so that the sums have the same bits: 8 x 128 where the
    // launch chain folds LayerNorm 2 into the GEMM (gemm_dec_ln_kernel<.., 8>: every layer but the first), 4 x 256 for layer 0, whose rows
    // come as 16-bit from the embedding launch (gemm_dec_kernel<1, 8>; waves 4 .. 7 shadow 0 .. 3)
    static_assert(QW == 8 || (QW == 4 && !HASLN), "K split");
    constexpr int K = 1024, NW = 8, KW = K / QW, CH = KW / 32, XS = K + 16;
    constexpr int EPL = G::EPL, LPP = G::LPP, PPW = G::PPW, U = G::U, RPOS = NW * 32;
    static_assert(EPL == 8 && U == 4, "16-bit cache only");
    __shared__ __attribute__((aligned(16))) bf16_t xl[RA_ROWS * XS];       // the eight rows as 16-bit (step C); later the attention output of all rows (step E)
    __shared__ __attribute__((aligned(16))) float red[NW][64][4];
    __shared__ __attribute__((aligned(16))) float resl[RA_ROWS][16];       // LayerNorm output of the block's out_proj columns (its residual)
    __shared__ __attribute__((aligned(16))) float qg[64];
    __shared__ __attribute__((aligned(16))) unsigned kvg[64];              // newest position: k (32 pairs) | v (32 pairs)
    __shared__ float sm[NW * PPW], sl[NW * PPW], so[NW * PPW][64];
    __shared__ float wm_[NW], wl_[NW], wo_[NW][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (wave-uniform: branches on w are scalar branches)
    const int m = lane & 15, kg = lane >> 4;
    const int h = blockIdx.x, b = blockIdx.y, z = blockIdx.z;
    const int j = 2 * b + z;                                // this block's slice of head h: dims 4 j .. 4 j + 3 of q, k and v
    const int slot = lane / LPP, dsub = lane % LPP;
    const bool oproj = z == 0 && b < 4;                     // 64 out_proj tiles: tile 4 h + b
    const int n0 = (4 * h + (b & 3)) * 16;
    // the row's length first, in front of every store of this kernel: a scalar load (behind a store hipcc reads it through the vector
    // memory path and, being the youngest request, waits for it with vmcnt(0) -- i.e. for the whole first cache round)
    const int end = a.len_override >= 0 ? a.len_override : a.st[b].pos + 1;
    const int pos = end - 1;                                // the newest position: its k / v rows are produced by this launch
    const unsigned epoch = (unsigned)pos * 32u + (unsigned)a.layer + 1u;
    unsigned long long* tr = a.trace ? a.trace + (size_t)((z * RA_ROWS + b) * 16 + h) * 4 : nullptr;
    if (tr && threadIdx.x == 0) tr[0] = __builtin_amdgcn_s_memrealtime();

    // ---- A: requests ------------------------------------------------------------------------------------------------------------------
    f32x4 gv[4], bv[4], pb[4];
    f32x4 xa[PARTS][4], va[4];
    u32x4 x16[2];
    if constexpr (HASLN) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = (lane + 64 * i) * 4;
            gv[i] = *reinterpret_cast<const f32x4*>(a.ln_g + idx);
            bv[i] = *reinterpret_cast<const f32x4*>(a.ln_b + idx);
            if constexpr (DEFER) pb[i] = *reinterpret_cast<const f32x4*>(a.pbias + idx);
        }
        const float* x = a.pin + (size_t)w * a.pin_stride;
#pragma unroll
        for (int p = 0; p < PARTS; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) xa[p][i] = *reinterpret_cast<const f32x4*>(x + (size_t)p * RA_ROWS * a.pin_stride + (lane + 64 * i) * 4);
        if constexpr (DEFER) {
#pragma unroll
            for (int i = 0; i < 4; ++i) va[i] = *reinterpret_cast<const f32x4*>(a.pres + (size_t)w * a.pres_stride + (lane + 64 * i) * 4);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) x16[i] = *reinterpret_cast<const u32x4*>(a.xb + (size_t)w * a.xb_stride + (lane + 64 * i) * 8);
    }
    asm volatile("" ::: "memory");
    // the q/k/v tile: tile row mm -> row 4 j + (mm & 3) of part mm >> 2 of head h (rows 12 .. 15 repeat the v rows; their outputs are dropped)
    const int kbase = (w % QW) * KW + kg * 8;
    const bf16_t* wrow = a.Wqkv + (size_t)(min(m >> 2, 2) * K + 64 * h + 4 * j + (m & 3)) * K + kbase;
    u32x4 wv[CH];
#pragma unroll
    for (int s = 0; s < CH; ++s) wv[s] = ld_stream16(wrow + s * 32);
    const f32x4 qb = *reinterpret_cast<const f32x4*>(a.bqkv + min(kg, 2) * K + 64 * h + 4 * j);
    // the cache stream of (b, h): rounds of 256 positions alternate between the two blocks -- z = 1 takes the even rounds and merges,
    // z = 0 the odd ones and hands over (attn_decode_final_kernel<.., 8, true>)
    const int g0 = z == 0 ? 1 : 0;
    const bf16_t* kh = a.kcache + (size_t)b * a.kv_row_stride + (size_t)h * a.max_seq * 64 + dsub * EPL;
    const bf16_t* vh = a.vcache + (size_t)b * a.kv_row_stride + (size_t)h * a.max_seq * 64 + dsub * EPL;
    constexpr bool SCAL = EARLY >= 5;                       // the q/k/v sweep by scalar loads
    constexpr bool FAST = EARLY == 6;                       // rounds that lie wholly below the newest position: no masks, no override, no clamps (same bits)
    u32x4 kA[U], vA[U], kB[U], vB[U];
    // Slots past the end: their scores are masked, but their REQUESTS are real.  The first version clamped them to the plane and so streamed
    // two or three useless rounds per block at short caches (step at kv 600: 717 us; 668 us since).  A slot past the end now asks for the position
    // of slot 0 of ITS OWN wave-load (the eight positions of a wave-load are one request each: the lanes of a dead slot fold into a live one), for
    // the newest position if that is dead too (one request for the whole instruction), and a round entirely past the end is not requested at all.
    // PMC, 256 generated steps at 8 rows (cache 257 .. 513): 23.0 MB per launch against 21.0 MB algorithmic (8.4 MB of q/k/v + out_proj weights
    // + 8 rows x 385 positions x 4 KB) = 1.10x, profiles/r05_pmc_decode_traffic_b8.json.
    auto kv_pos = [&](int base, int u) -> size_t {
        const int p = base + u * PPW, p0 = p - slot;          // this slot's position; slot 0's of the same wave-load
        return (size_t)(p < end ? p : p0 < end ? p0 : pos);
    };
    auto early = [&](int rr, u32x4 (&kr)[U], u32x4 (&vr)[U]) {
        if ((g0 + 2 * rr) * RPOS >= end) return;              // (block-uniform: the whole round is past the end; its registers are never reduced)
        const int base = (g0 + 2 * rr) * RPOS + w * 32 + slot;
#pragma unroll
        for (int u = 0; u < U; ++u) kr[u] = ld_stream16(kh + kv_pos(base, u) * 64);
#pragma unroll
        for (int u = 0; u < U; ++u) vr[u] = ld_stream16(vh + kv_pos(base, u) * 64);
    };
    if constexpr (EARLY == 1) early(0, kA, vA);
    asm volatile("" ::: "memory");
    // (a memory clobber orders requests, not arithmetic: hipcc hoists the sums of step B above the weight / cache requests and waits for the row
    //  in front of them.  Passing the row's registers through an empty asm HERE pins their first use behind every request above.)
    if constexpr (HASLN) {
#pragma unroll
        for (int p = 0; p < PARTS; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(xa[p][i]));
    }

    // ---- B: the rows ------------------------------------------------------------------------------------------------------------------
    if constexpr (HASLN) {
        f32x4 s[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                        // the partial buffers in their order, + bias, + residual (gemm_dec_ln_kernel::sum_row)
            s[i] = xa[0][i];
#pragma unroll
            for (int p = 1; p < PARTS; ++p) { s[i].x += xa[p][i].x; s[i].y += xa[p][i].y; s[i].z += xa[p][i].z; s[i].w += xa[p][i].w; }
            if constexpr (DEFER) {
                s[i].x += pb[i].x; s[i].y += pb[i].y; s[i].z += pb[i].z; s[i].w += pb[i].w;
                s[i].x += va[i].x; s[i].y += va[i].y; s[i].z += va[i].z; s[i].w += va[i].w;
            }
        }
        if constexpr (EARLY == 1) early(1, kB, vB);         // (the row's vectors have collapsed: registers for the second round)
        else if constexpr (EARLY == 0) early(0, kA, vA);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(s[i]));
        const float x0 = readlane_f(s[0].x, 0);
        float sm1 = 0.f, sq1 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) ln_chunk_moments(s[i], x0, sm1, sq1);
        sm1 = wave_sum(sm1); sq1 = wave_sum(sq1);
        float md, rstd;
        ln_finish(sm1, 0.f, 0.f, 0.f, sq1, 0.f, 0.f, 0.f, K, a.ln_eps, md, rstd);
        const bool writer = a.xn_out && h == 0 && b == 0 && z == 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = (lane + 64 * i) * 4;
            ln_apply(s[i], md, rstd, gv[i], bv[i]);
            if (writer) *reinterpret_cast<f32x4*>(a.xn_out + (size_t)w * a.xn_stride + idx) = s[i];
            *reinterpret_cast<u32x2*>(&xl[w * XS + idx]) = pack4<HT>(s[i]);
            if (idx >= n0 && idx < n0 + 16) *reinterpret_cast<f32x4*>(&resl[w][idx - n0]) = s[i];
        }
        if constexpr (EARLY == 0) early(1, kB, vB);
    } else {
        if constexpr (EARLY == 0) early(0, kA, vA);
        if constexpr (EARLY < 2) early(1, kB, vB);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(&xl[w * XS + (lane + 64 * i) * 8]) = x16[i];
    }
    // the out_proj tile of this block (used by 64 of the 256 blocks; 32 KB each): rows n0 + m, the K split of gemm_dec_kernel<1, 8>
    // (four waves x 256; waves 4 .. 7 shadow waves 0 .. 3)
    const bf16_t* worow = a.Wo + (size_t)(n0 + m) * K + (w & 3) * 256 + kg * 8;
    u32x4 wo[8];
    if constexpr (EARLY < 2) {
#pragma unroll
        for (int s = 0; s < 8; ++s) wo[s] = ld_stream16(worow + s * 32);
    }
    const f32x4 ob = *reinterpret_cast<const f32x4*>(a.bo + n0 + kg * 4);
    f32x4 rs = {0.f, 0.f, 0.f, 0.f};
    if constexpr (!HASLN) rs = *reinterpret_cast<const f32x4*>(a.res + (size_t)min(m, RA_ROWS - 1) * a.res_stride + n0 + kg * 4);
    asm volatile("" ::: "memory");
    __syncthreads();

    // ---- C: q/k/v of the block's 12 rows for all eight batch rows ----------------------------------------------------------------------
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const bf16_t* xr = xl + min(m, RA_ROWS - 1) * XS + kbase;
#pragma unroll
        for (int s = 0; s < CH; ++s) acc = H16<HT>::mfma16(wv[s], *reinterpret_cast<const u32x4*>(xr + s * 32), acc);
        *reinterpret_cast<f32x4*>(&red[w][lane][0]) = acc;
    }
    if constexpr (EARLY == 2) {                             // the weights have arrived: now the cache stream (and behind it the out_proj tile)
        asm volatile("" ::: "memory");
        early(0, kA, vA);
        early(1, kB, vB);
#pragma unroll
        for (int s = 0; s < 8; ++s) wo[s] = ld_stream16(worow + s * 32);
        asm volatile("" ::: "memory");
    }
    if constexpr (EARLY == 3) {                             // ... one round, and not by the wave that publishes and sweeps
        asm volatile("" ::: "memory");
        if (w != 0) early(0, kA, vA);
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    if (tr && threadIdx.x == 0) tr[1] = __builtin_amdgcn_s_memrealtime();          // q/k/v MFMAs done
    if (w == 0) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][lane][0]);
#pragma unroll
        for (int i = 1; i < QW; ++i) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(&red[i][lane][0]);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        v.x += qb.x; v.y += qb.y; v.z += qb.z; v.w += qb.w;
        // lane (m, kg): batch row m, part kg (q, k, v), dims 4 j .. 4 j + 3 of head h
        if (m < RA_ROWS && kg < 3) {
            const unsigned p0 = H16<HT>::pack2(v.x, v.y), p1 = H16<HT>::pack2(v.z, v.w);
            u64* g = a.qkv_gran + (size_t)m * RA_QKV_GRANULES + (kg * 16 + h) * 32 + 2 * j;
            ps_publish(g, 0, epoch, p0);
            ps_publish(g, 1, epoch, p1);
            if (kg > 0) {                                   // the newest position joins the cache (read by later steps' launches)
                bf16_t* plane = kg == 1 ? a.kcache : a.vcache;
                u32x2 pk; pk.x = p0; pk.y = p1;
                *reinterpret_cast<u32x2*>(plane + (size_t)m * a.kv_row_stride + ((size_t)h * a.max_seq + pos) * 64 + 4 * j) = pk;
            }
        }
        // ---- D.1: the 96 granules of (b, h): lanes 0 .. 31 q and v, lanes 32 .. 63 k --------------------------------------------------
        if constexpr (!SCAL) {
        const gu64* gq = (const gu64*)(a.qkv_gran + (size_t)b * RA_QKV_GRANULES + h * 32);
        const gu64* p1 = gq + (lane < 32 ? lane : 16 * 32 + (lane - 32));
        const gu64* p2 = gq + 2 * 16 * 32 + (lane & 31);
        u64 t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        u64 v1, v2;
        for (;;) {
            v1 = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v2 = __hip_atomic_load(p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((unsigned)(v1 >> 32) == epoch && (unsigned)(v2 >> 32) == epoch)) break;
            __builtin_amdgcn_s_sleep(1);
            if (xchg_expired(spins, t0, a.err)) {
                if (lane == 0) xchg_raise(a.err, RA_ERR_QKV, spins);
                v1 = 0; v2 = 0;
                break;
            }
        }
        if (lane == 0) xchg_note_slow(a.err, spins, t0);
        if (lane < 32) {
            qg[2 * lane] = H16<HT>::lo((unsigned)v1); qg[2 * lane + 1] = H16<HT>::hi((unsigned)v1);
            kvg[32 + lane] = (unsigned)v2;
        } else kvg[lane - 32] = (unsigned)v1;
        if constexpr (EARLY == 3) early(0, kA, vA);
        }
    }
    if constexpr (SCAL) {
        // The sweep by SCALAR loads, spread over the waves 0 .. 3 (three 64-byte pieces each: q = pieces 0 .. 3, k = 4 .. 7, v = 8 .. 11), while
        // the waves 4 .. 7 request their first two cache rounds; the sweeping waves request theirs when they have what they swept for.  The
        // barrier in front: wave 0's granule stores have been issued before the stream takes the vector-memory path.
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // (raw: the stores have been issued; their acknowledgement is not waited for)
        asm volatile("" ::: "memory");
        if (w >= 4) {
            early(0, kA, vA);
            early(1, kB, vB);
        } else {
            const u64* gq = a.qkv_gran + (size_t)b * RA_QKV_GRANULES + h * 32;
            const int c0 = 3 * w;
            const u64* pp[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) { const int c = c0 + i; pp[i] = gq + (size_t)(c >> 2) * 16 * 32 + (c & 3) * 8; }
            u64 t0 = __builtin_amdgcn_s_memrealtime();
            unsigned spins = 0;
            u32x16 g[3];
            for (;;) {
                sload3x64_glc(pp[0], pp[1], pp[2], g[0], g[1], g[2]);
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) ok = ok && g[i][2 * e + 1] == epoch;
                if (ok) break;
                if (spins >= 48u && (spins & 15u) == 0u) {
                    // Insurance: the protocol of every other sweep (agent-scope vector loads, which the memory model defines) looks at the same 24
                    // granules; if THEY carry the epoch while the scalar reads do not, the scalar path served stale bytes -- take the vector
                    // values and count the event (err[4], ma_engine_get_option "scalar_sweep_rescues"; never seen in 10^6 sweeps of the tests).
                    const int li = lane < 24 ? lane : 0;
                    const u64 vv = __hip_atomic_load((const gu64*)pp[li >> 3] + (li & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__all((unsigned)(vv >> 32) == epoch)) {
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const u64 t = __shfl(vv, i * 8 + e);
                                g[i][2 * e] = __builtin_amdgcn_readfirstlane((unsigned)t); g[i][2 * e + 1] = epoch;
                            }
                        if (lane == 0) __hip_atomic_fetch_add(a.err + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                __builtin_amdgcn_s_sleep(1);
                if (xchg_expired(spins, t0, a.err)) {
                    if (lane == 0) xchg_raise(a.err, RA_ERR_QKV, spins);
#pragma unroll
                    for (int i = 0; i < 3; ++i) g[i] = u32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                    break;
                }
            }
            if (lane == 0) xchg_note_slow(a.err, spins, t0);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int c = c0 + i, region = c >> 2, gi = (c & 3) * 8 + lane;      // lane e < 8: granule e of the piece
                unsigned val = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) val = lane == e ? g[i][2 * e] : val;
                if (lane < 8) {
                    if (region == 0) { qg[2 * gi] = H16<HT>::lo(val); qg[2 * gi + 1] = H16<HT>::hi(val); }
                    else kvg[(region - 1) * 32 + gi] = val;
                }
            }
        }
    }
    __syncthreads();
    if constexpr (SCAL) {                             // (a wave stalls at the issue of a request the CU's path has no room for: behind the barrier)
        if (tr && threadIdx.x == 0) tr[2] = __builtin_amdgcn_s_memrealtime();
        if (w < 4) { early(0, kA, vA); early(1, kB, vB); }
        asm volatile("" ::: "memory");
    }
    if constexpr (EARLY >= 3) {                             // the exchange is over: the rest of the prefetch depth, and the out_proj tile
        if constexpr (EARLY == 4) early(0, kA, vA);
        if constexpr (!SCAL) {
            early(1, kB, vB);
#pragma unroll
            for (int s = 0; s < 8; ++s) wo[s] = ld_stream16(worow + s * 32);
        }
        asm volatile("" ::: "memory");
    }
    if constexpr (!SCAL) { if (tr && threadIdx.x == 0) tr[2] = __builtin_amdgcn_s_memrealtime(); }

    // ---- D.2: attention over this block's rounds ---------------------------------------------------------------------------------------
    const int nr_all = (max(end, 0) + RPOS - 1) / RPOS;
    const int nround = z == 0 ? nr_all / 2 : (nr_all + 1) / 2;
    auto issue = [&](int r, u32x4 (&kr)[U], u32x4 (&vr)[U]) {
        const int base = (g0 + 2 * r) * RPOS + w * 32 + slot;
#pragma unroll
        for (int u = 0; u < U; ++u) kr[u] = ld_stream16(kh + kv_pos(base, u) * 64);
#pragma unroll
        for (int u = 0; u < U; ++u) vr[u] = ld_stream16(vh + kv_pos(base, u) * 64);
    };
    float qv[EPL];
    {
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(qg + dsub * EPL), q1 = *reinterpret_cast<const f32x4*>(qg + dsub * EPL + 4);
        qv[0] = q0.x; qv[1] = q0.y; qv[2] = q0.z; qv[3] = q0.w; qv[4] = q1.x; qv[5] = q1.y; qv[6] = q1.z; qv[7] = q1.w;
    }
    const u32x4 ok4 = *reinterpret_cast<const u32x4*>(kvg + dsub * 4), ov4 = *reinterpret_cast<const u32x4*>(kvg + 32 + dsub * 4);
    AttnSlotState<HT> ss;
    ss.m = -1e30f; ss.l = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) ss.o[e] = 0.f;
    // a round whose 256 positions all lie below the newest one (block-uniform) needs neither the masks nor the override of the newest position,
    // and its requests no clamps: the same arithmetic with ~100 of ~380 instructions per round fewer (the round loop is not hidden under the
    // stream: the fp16 instantiation, 64 conversions per round cheaper, steps 2 us per layer faster at kv 3858)
    auto inner = [&](int rr) { return FAST && (g0 + 2 * rr + 1) * RPOS < end; };
    auto reduce = [&](int rr, u32x4 (&kr)[U], u32x4 (&vr)[U]) {
        const int rb = (g0 + 2 * rr) * RPOS + w * 32 + slot;
        if (inner(rr)) attn_round_reduce<HT, false, true>(ss, qv, kr, vr, rb, end, -1, ok4, ov4);
        else attn_round_reduce<HT, true>(ss, qv, kr, vr, rb, end, pos, ok4, ov4);
    };
    auto issue_f = [&](int rr, u32x4 (&kr)[U], u32x4 (&vr)[U]) {
        if (!inner(rr)) { issue(rr, kr, vr); return; }
        const int base = (g0 + 2 * rr) * RPOS + w * 32 + slot;
#pragma unroll
        for (int u = 0; u < U; ++u) kr[u] = ld_stream16(kh + (size_t)(base + u * PPW) * 64);
#pragma unroll
        for (int u = 0; u < U; ++u) vr[u] = ld_stream16(vh + (size_t)(base + u * PPW) * 64);
    };
    for (int r = 0; r < nround; r += 2) {                   // (rounds 0 and 1 are already on their way)
        reduce(r, kA, vA);
        if (r + 2 < nround) issue_f(r + 2, kA, vA);
        if (r + 1 < nround) {
            reduce(r + 1, kB, vB);
            if (r + 3 < nround) issue_f(r + 3, kB, vB);
        }
    }
    if constexpr (SCAL) {
        // the out_proj tile: only the 64 blocks that use it ask for it (64 KB of requests per block otherwise, a cache round's worth on the CU's
        // path), and behind the stream: the merges, the hand-over and the sweep of step E lie between here and its first use
        if (oproj) {
#pragma unroll
            for (int s = 0; s < 8; ++s) wo[s] = ld_stream16(worow + s * 32);
        }
        asm volatile("" ::: "memory");
    }
    // block-level merge (attn_decode_final_kernel: wave w folds its PPW slot states, wave 0 the NW wave states)
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
        gu64* g = (gu64*)(a.pair_gran + ((size_t)b * 16 + h) * ATTN_PAIR_GRANULES);
        if (z == 0) {                                       // publisher: 64 x o, then m and l
            __hip_atomic_store(g + lane, ((u64)epoch << 32) | __float_as_uint(O), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane < 2) __hip_atomic_store(g + 64 + lane, ((u64)epoch << 32) | __float_as_uint(lane == 0 ? M : L), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            u64 vo = 0, vm = (u64)epoch << 32;
            u64 t0 = __builtin_amdgcn_s_memrealtime();
            unsigned spins = 0;
            for (;;) {
                vo = __hip_atomic_load(g + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                vm = __hip_atomic_load(g + 64 + (lane & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((unsigned)(vo >> 32) == epoch && (unsigned)(vm >> 32) == epoch)) break;
                __builtin_amdgcn_s_sleep(1);
                if (xchg_expired(spins, t0, a.err)) {
                    if (lane == 0) xchg_raise(a.err, ATTN_ERR_PAIR, spins);
                    vo = 0; vm = 0;
                    break;
                }
            }
            if (lane == 0) xchg_note_slow(a.err, spins, t0);
            const float O2 = __uint_as_float((unsigned)vo);
            const float M2 = readlane_f(__uint_as_float((unsigned)vm), 0), L2 = readlane_f(__uint_as_float((unsigned)vm), 1);
            const float Mx = fmaxf(M, M2), f1 = expf(M - Mx), f2 = expf(M2 - Mx);      // this block's rounds first, then the partner's
            L = fmaf(L2, f2, L * f1);
            O = fmaf(O2, f2, O * f1);
            // ---- E.1: the attention output of (b, h), rounded as the out_proj GEMM's operand, two dims per granule ------------------------
            // (position 0 always exists: L > 0.)  The product is rounded to 16 bits in the SAME expression as in attn_decode_final_kernel: for fp16
            // hipcc fuses the multiplication and the conversion into one instruction with ONE rounding (v_fma_mixlo_f16); a product kept in fp32
            // first (to hand it to the neighbour lane) is rounded twice and differs in about one value of 8 000
            const unsigned ob = H16<HT>::bits(O * (1.0f / L));
            const unsigned nb = __shfl_down(ob, 1, 64);
            if (!(lane & 1)) ps_publish(a.out_gran + (size_t)b * RA_OUT_GRANULES + h * 32, lane >> 1, epoch, ob | (nb << 16));
        }
    }
    if (tr && threadIdx.x == 0) tr[3] = __builtin_amdgcn_s_memrealtime();
    if (!oproj) return;

    // ---- E.2: out_proj tile n0 .. n0 + 15 for all eight rows ---------------------------------------------------------------------------
    __syncthreads();                                        // (xl: every wave is past step C; wave 0 past its hand-over)
    {
        const gu64* ga = (const gu64*)(a.out_gran + (size_t)w * RA_OUT_GRANULES);
        u64 t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        u64 v[8];
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                v[c] = __hip_atomic_load(ga + lane + 64 * c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (unsigned)(v[c] >> 32) == epoch;
            }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if (xchg_expired(spins, t0, a.err)) {
                if (lane == 0) xchg_raise(a.err, RA_ERR_OUT, spins);
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = 0;
                break;
            }
        }
        if (lane == 0) xchg_note_slow(a.err, spins, t0);
        unsigned* al = reinterpret_cast<unsigned*>(xl + w * XS);
#pragma unroll
        for (int c = 0; c < 8; ++c) al[lane + 64 * c] = (unsigned)v[c];
    }
    __syncthreads();
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const bf16_t* xr = xl + min(m, RA_ROWS - 1) * XS + (w & 3) * 256 + kg * 8;
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = H16<HT>::mfma16(wo[s], *reinterpret_cast<const u32x4*>(xr + s * 32), acc);
        *reinterpret_cast<f32x4*>(&red[w][lane][0]) = acc;
    }
    __syncthreads();
    if (w == 0 && m < RA_ROWS) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][lane][0]);
#pragma unroll
        for (int i = 1; i < 4; ++i) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(&red[i][lane][0]);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        if constexpr (HASLN) rs = *reinterpret_cast<const f32x4*>(&resl[m][kg * 4]);
        v.x += ob.x; v.y += ob.y; v.z += ob.z; v.w += ob.w;        // gd_epi_store: + bias, then + residual
        v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w;
        *reinterpret_cast<f32x4*>(a.y1 + (size_t)m * a.y1_stride + n0 + kg * 4) = v;
        if (tr && threadIdx.x == 0) tr[3] = __builtin_amdgcn_s_memrealtime();       // (the out_proj blocks: their end, not their hand-over)
    }
}

template <typename HT>
inline hipError_t launch_rows_attn(const RowsAttnArgs& a, int heads, int rows, hipStream_t s, int early_kv = 2, int q_waves = 4) {
    if (heads != 16 || rows != RA_ROWS || !a.Wqkv || !a.Wo || !a.qkv_gran || !a.pair_gran || !a.out_gran || !a.err || !a.y1 || a.y1_stride % 4) return hipErrorInvalidValue;
    const dim3 grid(16, RA_ROWS, 2), block(512);
    // placements: 6 (the default) and its two controls -- 5 (6 without the mask-free rounds) and 3 (the q/k/v sweep by vector loads) -- are
    // product code; 0, 1, 2 and 4 were measured and not kept (profiles/r05_ab_rows_attn_request_placement.txt) and are compiled in with
    // MA_EXPERIMENTAL=1 only (84 instantiations of this kernel sat in the product library)
#ifdef MA_EXPERIMENTAL
    if (early_kv < 0 || early_kv > 6) return hipErrorInvalidValue;
#define MA_RA(L, P, D, Q) do { if (early_kv == 2) hipLaunchKernelGGL((rows_attn_kernel<L, P, D, 2, Q, HT>), grid, block, 0, s, a); \
                               else if (early_kv == 5) hipLaunchKernelGGL((rows_attn_kernel<L, P, D, 5, Q, HT>), grid, block, 0, s, a); \
                               else if (early_kv == 6) hipLaunchKernelGGL((rows_attn_kernel<L, P, D, 6, Q, HT>), grid, block, 0, s, a); \
                               else if (early_kv == 3) hipLaunchKernelGGL((rows_attn_kernel<L, P, D, 3, Q, HT>), grid, block, 0, s, a); \
                               else if (early_kv == 4) hipLaunchKernelGGL((rows_attn_kernel<L, P, D, 4, Q, HT>), grid, block, 0, s, a); \
                               else if (early_kv == 1) hipLaunchKernelGGL((rows_attn_kernel<L, P, D, 1, Q, HT>), grid, block, 0, s, a); \
                               else hipLaunchKernelGGL((rows_attn_kernel<L, P, D, 0, Q, HT>), grid, block, 0, s, a); } while (0)
#else
    if (early_kv != 3 && early_kv != 5 && early_kv != 6) return hipErrorInvalidValue;
#define MA_RA(L, P, D, Q) do { if (early_kv == 5) hipLaunchKernelGGL((rows_attn_kernel<L, P, D, 5, Q, HT>), grid, block, 0, s, a); \
                               else if (early_kv == 6) hipLaunchKernelGGL((rows_attn_kernel<L, P, D, 6, Q, HT>), grid, block, 0, s, a); \
                               else hipLaunchKernelGGL((rows_attn_kernel<L, P, D, 3, Q, HT>), grid, block, 0, s, a); } while (0)
#endif
    if (!a.ln_g) {
        if (!a.xb || !a.res || a.xb_stride % 8 || a.res_stride % 4) return hipErrorInvalidValue;
        if (q_waves == 8) MA_RA(false, 1, false, 8);
        else if (q_waves == 4) MA_RA(false, 1, false, 4);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (!a.pin || !a.ln_b || a.pin_stride % 4 || (a.pres && a.pres_stride % 4) || (a.pbias != nullptr) != (a.pres != nullptr)) return hipErrorInvalidValue;
    const bool d = a.pres != nullptr;
    if (a.pin_parts == 1 && !d) MA_RA(true, 1, false, 8);
    else if (a.pin_parts == 2 && d) MA_RA(true, 2, true, 8);
    else if (a.pin_parts == 4 && d) MA_RA(true, 4, true, 8);
    else return hipErrorInvalidValue;
#undef MA_RA
    return hipGetLastError();
}

}  // namespace ma
