// Dense attention of the bf16 policy, second generation: O = softmax_fp32(Q K^T * scale) V on the matrix cores with the
// "swapped" formulation, head_dim 64.  Same reference call sites as attn.hpp (QKVMultiheadAttention / QKVMultiheadCrossAttention,
// transformer_blocks.py:56-74, 166-185; [3p] flash_attn prefill, shape_opt.py:403-410; [3p] BERT self-attention, meshanything.py:62-64).
//
// Why a second kernel: attention_mfma_kernel (attn.hpp) spent the dense phases' time, not the GEMM -- 52 % of it at 8.6 % matrix-core
// busy at batch 64 (profiles/r02_pmc_dense_mfma.json): V was transposed through LDS with 2-byte stores every tile, P went through
// LDS, three block barriers per 64-key tile, no load under compute, 64-row blocks that each re-staged the same K / V.
//
// Structure (MI355X guide, "Fused attention prefill"):
//   * S^T = K Q^T with v_mfma_f32_32x32x16_bf16: A = a K tile (keys x d) read from LDS with ds_read_b128, B = the wave's 32 query
//     rows, held in registers for the whole block.  In the accumulator layout a LANE owns ONE query (column lane & 31) and 32 of the
//     tile's 64 keys, its partner lane ^ 32 the other 32: the online softmax is 32 register operations plus ONE cross-lane exchange
//     per tile (the row maximum); the row sum stays a per-lane partial until the end.
//   * O^T = V^T P^T: P^T is already in the B-operand register layout (column = query) -- converted to bf16 in place, never touching
//     LDS; the k-index of the contraction is permuted to the order in which the accumulator holds the keys ({0-3, 8-11 | 4-7, 12-15}
//     of every 16), and V^T is STORED in that order by the kernel that produces it (vt_pack_kernel below: V arrives row-major from
//     the q/k/v GEMM; the transposition costs one pass over 2-byte data instead of 16 two-byte LDS stores per thread and tile).
//     A = the V^T tile (d x keys) from LDS, again one ds_read_b128 per MFMA.
//   * one block = NW waves x 32 query rows (NW = 3 for the 257-row sequences: 96-row blocks waste 11 %, 128-row blocks a third of their
//     staging), K / V^T tiles of 64 keys in a THREE-stage LDS ring filled by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, the
//     bank swizzle on the SOURCE address: the DMA writes lane-linear): the pieces of tile t + 2 are requested before the 16 MFMAs of tile t, a
//     counted vmcnt leaves them in flight across the tile's ONE barrier.  (Round 5's form staged tile t + 1 through registers -- requested before
//     the MFMAs, written to LDS after them -- so every tile ended on the full latency of its successor's loads: a block was a chain of round
//     trips, 9 % matrix-core busy on the 257-row sequences.)
//   * both tiles are [64 rows][128 bytes] with the 16-byte chunk index XOR (row >> 1) & 7: conflict-free for the hardware's 16-lane
//     ds_read_b128 groups when the 32 lanes of a half-wave read 32 different rows (checked by enumeration);
//   * the output leaves through a per-wave LDS patch so that a row is stored as whole 128-byte lines.
// Arithmetic: scores, softmax statistics and the output accumulate in fp32; Q, K, V and the probabilities that multiply V are bf16 (the
// policy's rounding points, oracle/meshanything_oracle.py `attention(dense=True)`); exp is v_exp_f32 on log2-scaled scores.
// Algorithmic FLOPs: 4 Sq Sk 64 per head.
#pragma once
#include "attn.hpp"
#include "common.hpp"
#include "gemm.hpp"
#include "gemm_tile.hpp"

namespace ma {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Attn2Args {
    const bf16_t* Q; int q_rs, q_hs;
    const bf16_t* K; int k_rs, k_hs;
    const bf16_t* VT;                   // packed V^T: [batch][H][64][skp] (vt_pack_kernel), skp = Sk rounded up to 64
    bf16_t* O; int o_rs;
    int Sq, Sk, skp, H;
    float scale;
    int causal_offset;                  // < 0: full attention; else query i sees keys <= causal_offset + i
    size_t q_bs, k_bs, o_bs;            // element strides between the samples of a batch (grid.z)
};

__device__ __forceinline__ int a2_slot(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
// two floats -> packed bf16 (round to nearest even; v_cvt_pk_bf16_f32 on gfx950), element 0 in the low half
typedef float a2_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 a2_bf16x2 __attribute__((ext_vector_type(2)));
template <typename HT> __device__ __forceinline__ uint32_t a2_pack(float lo, float hi_) {
    if constexpr (__is_same(HT, f16_t)) return H16<f16_t>::pack2(lo, hi_);
    else return __builtin_bit_cast(uint32_t, __builtin_convertvector(a2_f32x2{lo, hi_}, a2_bf16x2));
}

// V (Sk rows x 64 d per head, row stride v_rs, head offset h * v_hs) -> V^T [batch][H][64][skp], zero beyond Sk, the keys of every
// aligned 16 stored in the order {0-3, 8-11, 4-7, 12-15}.  One block = one 64-key tile of one (sample, head).
__global__ __launch_bounds__(256) void vt_pack_kernel(const bf16_t* __restrict__ V, int v_rs, int v_hs, size_t v_bs, bf16_t* __restrict__ VT, int Sk, int skp, int H) {
    __shared__ bf16_t t[64][66];        // [key][d], padded: the column reads below walk the keys
    const int tid = threadIdx.x, h = blockIdx.y, b = blockIdx.z, k0 = blockIdx.x * 64;
    const bf16_t* vp = V + (size_t)b * v_bs + (size_t)h * v_hs;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + 256 * i, kr = id >> 3, c = id & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (k0 + kr < Sk) v = *reinterpret_cast<const u32x4*>(vp + (size_t)(k0 + kr) * v_rs + c * 8);
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { t[kr][c * 8 + 2 * j] = (bf16_t)(w4[j] & 0xffffu); t[kr][c * 8 + 2 * j + 1] = (bf16_t)(w4[j] >> 16); }
    }
    __syncthreads();
    bf16_t* op = VT + ((size_t)b * H + h) * 64 * skp + k0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + 256 * i, d = id >> 3, c = id & 7;       // output chunk c of row d: positions 8c .. 8c + 7 of the tile
        const int g = c >> 1, half = c & 1;                          // 16-key group, which half of its permuted order
        uint32_t w4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p0 = 2 * j, p1 = 2 * j + 1;                    // positions inside the half -> keys {0-3, 8-11} | {4-7, 12-15}
            const int ka = 16 * g + 4 * half + (p0 & 3) + 8 * (p0 >> 2), kb = 16 * g + 4 * half + (p1 & 3) + 8 * (p1 >> 2);
            w4[j] = (uint32_t)t[ka][d] | ((uint32_t)t[kb][d] << 16);
        }
        *reinterpret_cast<u32x4*>(op + (size_t)d * skp + c * 8) = u32x4{w4[0], w4[1], w4[2], w4[3]};
    }
}

// HT: the 16-bit format of Q / K / V^T / O (bf16_t | f16_t, common.hpp H16)
// Three waves per SIMD (launch bound): the 96-row form came out at 169 registers -- ONE over the step to three waves per SIMD -- so a CU held two
// blocks of it, and a block is a chain of latencies (Q and the first tile, then one round trip per tile: 9 % matrix-core busy, r04_pmc_dense_mfma).
// With the bound a dozen loop-invariant addresses go to scratch around the loop (none inside it) and a CU holds four blocks.
template <int NW, typename HT = bf16_t>
__global__ __launch_bounds__(NW * 64, 3) void attention_mfma2_kernel(Attn2Args a) {
    constexpr int NT = NW * 64, RB = NW * 32, TILE = 64 * 128;         // threads, query rows per block, bytes of one K or V^T tile
    constexpr int NST = 3;                                              // ring stages
    __shared__ __attribute__((aligned(16))) char smem[NST * 2 * TILE]; // ring of three stages x {K, V^T}; reused for the output patches
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), ln = lane & 31, hi = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z, q0 = blockIdx.x * RB;
    const bf16_t* Qp = a.Q + (size_t)b * a.q_bs + (size_t)h * a.q_hs;
    const bf16_t* Kp = a.K + (size_t)b * a.k_bs + (size_t)h * a.k_hs;
    const bf16_t* Vp = a.VT + ((size_t)b * a.H + h) * 64 * a.skp;
    const int qrow = q0 + w * 32 + ln;                                  // this lane's query
    const bool qok = qrow < a.Sq;

    // Q^T fragments: B[k = d][n = query]: lane (query ln, k group hi) holds d = 16 s + 8 hi .. + 7 for the four 16-deep steps
    u32x4 qf[4];
    {
        // (rows past Sq read row 0 and are never stored: no lane-dependent branch around a request -- hipcc waits for the data where such
        //  a branch ends)
        const bf16_t* qp = Qp + (size_t)(qok ? qrow : 0) * a.q_rs + 8 * hi;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const u32x4*>(qp + 16 * s);
    }
    int kv_end = a.Sk;
    if (a.causal_offset >= 0) kv_end = min(a.Sk, a.causal_offset + min(q0 + RB - 1, a.Sq - 1) + 1);
    const int nt = (kv_end + 63) >> 6;

    // LDS-DMA of tile t into ring stage t % 3: 16 pieces of 1 KiB (8 K pieces = 8 key rows each, 8 V^T pieces = 8 d rows each), piece p by wave p % NW.
    // Lane l of a piece lands at byte 16 l = row l >> 3, slot l & 7 of the piece: it fetches the chunk whose swizzled slot that is (a2_slot is an
    // XOR: its own inverse).  Key rows past Sk repeat the last row: their scores are masked below (key <= klimit); V^T is zero-padded by
    // vt_pack_kernel.  No lane-dependent branch around a request.
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int drow = lane >> 3, dslot = lane & 7;
    auto stage = [&](int t) {
        const int kv0 = t << 6;
        const unsigned base = lds0 + (unsigned)(t % NST) * (2u * TILE);
#pragma unroll
        for (int i = 0; i < (16 + NW - 1) / NW; ++i) {
            const int p = w + NW * i;                                    // (wave-uniform)
            if (NW * i + NW > 16 && p >= 16) continue;
            const int r = (p & 7) * 8 + drow, c = dslot ^ ((r >> 1) & 7);
            const bf16_t* ks = Kp + (size_t)min(kv0 + r, a.Sk - 1) * a.k_rs + c * 8;
            const bf16_t* vs = Vp + (size_t)r * a.skp + kv0 + c * 8;
            gt_glds16(p < 8 ? ks : vs, __builtin_amdgcn_readfirstlane(base + (unsigned)p * 1024u));
        }
    };
    // pieces this wave requests per tile: the counted wait at a tile's end leaves exactly one tile's worth in flight
    const int my_pieces = (16 - w + NW - 1) / NW;

    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }
    float mrun = -1e30f, lsum = 0.f;
    const float sc2 = a.scale * 1.44269504088896340736f;               // scores in the log2 domain: exp(x) = exp2(x log2 e)

    if (nt > 0) stage(0);
    // every request made so far (the Q fragments, tile 0) is retired HERE, on every path: hipcc's wait-count pass must not carry "Q may still be in
    // flight" into the loop (it would guard the first MFMAs of every tile with vmcnt(3..0)); the DMA requests are invisible to it and counted by hand
    __builtin_amdgcn_s_waitcnt(0x0F70);                                 // vmcnt(0) -- the BUILTIN: the wait-count pass must see it
    if (nt > 1) stage(1);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        if (t + 2 < nt) stage(t + 2);                                   // ring stage (t + 2) % 3 = (t - 1) % 3: last read during tile t - 1, before the previous barrier
        const char* kb = smem + (t % NST) * 2 * TILE;
        const char* vb = kb + TILE;
        // ---- S^T = K Q^T: two 32-key blocks x four 16-deep steps ---------------------------------------------------------------------
        f32x16 sacc[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { sacc[0][i] = 0.f; sacc[1][i] = 0.f; }
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
            const int r = kbk * 32 + ln;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(kb + r * 128 + a2_slot(r, 2 * s + hi) * 16);
                sacc[kbk] = H16<HT>::mfma32(kf, qf[s], sacc[kbk]);
            }
        }
        // ---- online softmax on the lane's 32 keys of its query: key = 64 t + 32 kbk + (i & 3) + 8 (i >> 2) + 4 hi ----------------------
        // (round 6: the softmax's vector arithmetic, not the matrix pipe, set this kernel's pace -- ~300 vector instructions against 16 MFMAs per tile and
        //  wave, matrix cores 0.12 - 0.32 busy.  Three cuts: tiles that lie wholly below every row's last visible key skip the mask (a wave-uniform test:
        //  all but the last tile of an unmasked sequence, all but the diagonal tiles of a causal one); the scale rides in the exponent's fused multiply-add
        //  (scores stay raw, the running maximum is kept in the scaled domain: one multiply per tile instead of 32); the output accumulators are rescaled
        //  only when some row's maximum moved.)
        const int kbase = (t << 6) + 4 * hi;
        const int klimit = a.causal_offset >= 0 ? min(a.Sk - 1, a.causal_offset + qrow) : a.Sk - 1;       // last visible key of this lane's query
        const int klimit_wave = a.causal_offset >= 0 ? min(a.Sk - 1, a.causal_offset + q0 + w * 32) : a.Sk - 1;      // ... of the wave's first row: the smallest
        float mx = -INFINITY;
        if ((t << 6) + 63 <= klimit_wave) {
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[kbk][i]);
        } else {
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kbase + 32 * kbk + (i & 3) + 8 * (i >> 2);
                    const float v = key <= klimit ? sacc[kbk][i] : -INFINITY;
                    sacc[kbk][i] = v;
                    mx = fmaxf(mx, v);
                }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * sc2;                   // the partner lane holds the query's other 32 keys; sc2 > 0: the maximum commutes with the scale
        const float mnew = fmaxf(mrun, mx);
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
        const bool moved = mnew != mrun;
        mrun = mnew;
        float ps = 0.f;
        u32x4 pf[4];                                                    // P^T as B fragments: slice s = keys 16 s .. 16 s + 15 of the tile
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint32_t w4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e0 = __builtin_amdgcn_exp2f(fmaf(sacc[s >> 1][8 * (s & 1) + 2 * j], sc2, -mnew));        // exp2(-inf) = 0: masked keys drop out
                const float e1 = __builtin_amdgcn_exp2f(fmaf(sacc[s >> 1][8 * (s & 1) + 2 * j + 1], sc2, -mnew));
                ps += e0 + e1;
                w4[j] = a2_pack<HT>(e0, e1);
            }
            pf[s] = u32x4{w4[0], w4[1], w4[2], w4[3]};
        }
        if (__any(moved)) {                                             // (alpha == 1 exactly for every row otherwise)
            lsum = lsum * alpha + ps;
#pragma unroll
            for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; }
        } else lsum += ps;
        // ---- O^T += V^T P^T: four 16-key slices x two 32-row d blocks ------------------------------------------------------------------
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int r = db * 32 + ln;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const u32x4 vf = *reinterpret_cast<const u32x4*>(vb + r * 128 + a2_slot(r, 2 * s + hi) * 16);
                oacc[db] = H16<HT>::mfma32(vf, pf[s], oacc[db]);
            }
        }
        // tile t + 1 has landed (this wave's pieces: requested one tile ago; tile t + 2's stay in flight), then everyone's: ONE barrier per tile.
        // (the barrier is a bare s_barrier behind lgkmcnt(0): hipcc does not know of the DMA requests, so __syncthreads() carries no vmcnt(0))
        if (t + 2 < nt) {
            if (my_pieces == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (my_pieces == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // ---- epilogue: normalise, round, and store whole rows through this wave's LDS patch (32 rows x 128 bytes) ----------------------------
    const float ltot = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = ltot > 0.f ? 1.0f / ltot : 0.f;
    char* patch = smem + w * 4096;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int j = 0; j < 4; ++j) {                                   // registers 4 j .. 4 j + 3: d = 32 db + 8 j + 4 hi + 0 .. 3
            u32x2 pk;
            pk.x = a2_pack<HT>(oacc[db][4 * j] * inv, oacc[db][4 * j + 1] * inv);
            pk.y = a2_pack<HT>(oacc[db][4 * j + 2] * inv, oacc[db][4 * j + 3] * inv);
            *reinterpret_cast<u32x2*>(patch + ln * 128 + a2_slot(ln, 4 * db + j) * 16 + 8 * hi) = pk;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // the patch is private to the wave: its LDS queue is in order, no barrier needed
    bf16_t* Op = a.O + (size_t)b * a.o_bs + (size_t)h * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = lane + 64 * i, r = id >> 3, c = id & 7, row = q0 + w * 32 + r;
        if (row < a.Sq) *reinterpret_cast<u32x4*>(Op + (size_t)row * a.o_rs + c * 8) = *reinterpret_cast<const u32x4*>(patch + r * 128 + a2_slot(r, c) * 16);
    }
}

inline size_t attn2_vt_elems(int Sk, int H, int batch) { return (size_t)batch * H * 64 * ((Sk + 63) & ~63); }

// V^T packing + attention.  `vt` = workspace of attn2_vt_elems(Sk, H, batch) bf16 elements.
template <typename HT>
inline hipError_t launch_attention2(const AttnArgs& a, bf16_t* vt, hipStream_t s) {
    if (a.Sq <= 0 || a.Sk <= 0) return hipSuccess;
    const int skp = (a.Sk + 63) & ~63;
    if ((a.q_rs | a.k_rs | a.v_rs | a.o_rs | a.q_hs | a.k_hs | a.v_hs) % 8) return hipErrorInvalidValue;      // 16-byte vector accesses
    hipLaunchKernelGGL(vt_pack_kernel, dim3(skp / 64, a.H, a.batch), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(a.V), a.v_rs, a.v_hs, a.v_bs, vt, a.Sk, skp, a.H);
    Attn2Args g{reinterpret_cast<const bf16_t*>(a.Q), a.q_rs, a.q_hs, reinterpret_cast<const bf16_t*>(a.K), a.k_rs, a.k_hs, vt, reinterpret_cast<bf16_t*>(a.O), a.o_rs,
                a.Sq, a.Sk, skp, a.H, a.scale, a.causal_offset, a.q_bs, a.k_bs, a.o_bs};
    // 96-row blocks when they waste fewer rows than 128-row blocks (257 rows: 288 vs 384)
    const int pad3 = (a.Sq + 95) / 96 * 96, pad4 = (a.Sq + 127) / 128 * 128;
    if (pad3 < pad4) hipLaunchKernelGGL((attention_mfma2_kernel<3, HT>), dim3(pad3 / 96, a.H, a.batch), dim3(192), 0, s, g);
    else hipLaunchKernelGGL((attention_mfma2_kernel<4, HT>), dim3(pad4 / 128, a.H, a.batch), dim3(256), 0, s, g);
    return hipGetLastError();
}

}  // namespace ma
