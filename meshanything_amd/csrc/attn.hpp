// Dense attention for the encoder (257 x 4096 cross, 257 x 257 self), the decoder prefill (257 x 257 causal) and the
// detokenizer (1057 x 1057): O = softmax_fp32(Q K^T * scale) V, head_dim 64.
// Reference: QKVMultiheadAttention / QKVMultiheadCrossAttention (transformer_blocks.py:56-74, 166-185),
// [3p] flash_attn prefill (shape_opt.py:403-410) and [3p] BERT self-attention (meshanything.py:62-64).
//
// Two kernels here, both flash-style (online softmax, K/V tiles staged in LDS, nothing S x S ever materialised):
//  * attention_f32_kernel (exact fp32: the encoder under cfg.enc_exact, everything under the fp32 policy): fp32 matrix cores, exact
//    fp32 probabilities;
//  * attention_mfma_kernel (first-generation bf16 kernel, kept as a cross-check: the engine's bf16 kernel is attn2.hpp).
#pragma once
#include "common.hpp"

namespace ma {

struct AttnArgs {
    // element type of Q / K / V / O: float for attention_kernel and attention_mfma_kernel<float>, bf16 for
    // attention_mfma_kernel<bf16_t> (the engine's bf16 policy: activations are produced and consumed as bf16)
    const void* Q; int q_rs, q_hs;      // element strides: row (sequence position), head
    const void* K; int k_rs, k_hs;
    const void* V; int v_rs, v_hs;
    void* O; int o_rs;                  // O[q * o_rs + h*64 + d]
    int Sq, Sk, H;
    float scale;
    int causal_offset;                  // < 0: full attention; else query i sees keys <= causal_offset + i
    int round_bf16;
    // batch of independent samples: grid.z = sample, element strides between samples (0 / 1 launch at batch 1)
    size_t q_bs = 0, k_bs = 0, v_bs = 0, o_bs = 0; int batch = 1;
};

typedef float attn_f32x16 __attribute__((ext_vector_type(16)));

// fp32 "exact" attention on the fp32 matrix path (v_mfma_f32_32x32x2_f32: an exact fp32 FMA chain at the fp32 vector rate, MI355X guide
// section 3) -- the encoder's kernel under every policy with cfg.enc_exact, and prefill / detokenizer under the fp32 policy.  It replaces
// the VALU kernel of rounds 1-3, whose 64-float query and output rows per thread spilled 3 KB per lane.
// "Swapped" formulation (as attn2.hpp): S^T = K Q^T, so a lane owns ONE query (column lane & 31) and 16 of a tile's 32 keys, its partner
// lane ^ 32 the other 16 -- the online softmax is register arithmetic plus one cross-lane exchange; P^T is then already the B operand of
// O^T = V^T P^T when the contraction walks the keys in accumulator order (step r <-> key (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).
//   * one block = NW waves x 32 query rows of one head; K / V tiles of 32 keys staged through registers into LDS (the loads of tile t + 1
//     fly under the 64 MFMAs of tile t), K stored transposed [64 d][33] so that the A-operand reads (32 keys of one d) are conflict-free,
//     V row-major [32][64] (A operand of the second product: 32 consecutive d of one key);
//   * Q^T fragments (32 registers) live in registers for the whole block.
// 64 MFMAs of 64 cycles per wave and tile = the fp32 matrix rate whenever the LDS reads hide under them.  ~110 VGPRs, no scratch.
constexpr int AF_KT_LD = 33;

template <int NW>
__global__ __launch_bounds__(NW * 64) void attention_f32_kernel(AttnArgs a) {
    constexpr int NT = NW * 64, RB = NW * 32;
    constexpr int NCH = (512 + NT - 1) / NT;                       // float4 chunks per thread, tile and operand (32 keys x 16 chunks)
    __shared__ __attribute__((aligned(16))) float KT[64 * AF_KT_LD];
    __shared__ __attribute__((aligned(16))) float Vs[32 * 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ln = lane & 31, hi = lane >> 5;
    const int h = blockIdx.y, q0 = blockIdx.x * RB;
    const float* Qf = reinterpret_cast<const float*>(a.Q) + blockIdx.z * a.q_bs + (size_t)h * a.q_hs;
    const float* Kf = reinterpret_cast<const float*>(a.K) + blockIdx.z * a.k_bs + (size_t)h * a.k_hs;
    const float* Vf = reinterpret_cast<const float*>(a.V) + blockIdx.z * a.v_bs + (size_t)h * a.v_hs;
    float* Of = reinterpret_cast<float*>(a.O) + blockIdx.z * a.o_bs + (size_t)h * 64;
    const int qrow = q0 + w * 32 + ln;
    const bool qok = qrow < a.Sq;
    auto rnd = [&](float v) { return a.round_bf16 ? round_bf16(v) : v; };

    // Q^T as B operand of S^T = K Q^T: lane (query ln, half hi) holds Q[q][2 s + hi], s = 0 .. 31
    float qf[32];
    {
        const float* qp = Qf + (size_t)(qok ? qrow : 0) * a.q_rs;
        // (rows past Sq read row 0 and are never stored.  A guard around each request -- `if (qok) t = load` -- made hipcc wait for every one of
        //  the 16 before issuing the next: sixteen dependent round trips in front of a block's first tile.)
        f32x4 t[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) t[c] = *reinterpret_cast<const f32x4*>(qp + 4 * c);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            qf[2 * c] = rnd(hi ? t[c].y : t[c].x);
            qf[2 * c + 1] = rnd(hi ? t[c].w : t[c].z);
        }
    }
    int kv_end = a.Sk;
    if (a.causal_offset >= 0) kv_end = min(a.Sk, a.causal_offset + min(q0 + RB - 1, a.Sq - 1) + 1);
    const int nt = (kv_end + 31) >> 5;

    f32x4 kreg[NCH], vreg[NCH];
    auto gload = [&](int t) {
        const int kv0 = t << 5;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            // clamped, not guarded (see above): chunks past 512 repeat chunk 511 and are not stored; key rows past Sk repeat the last row
            // and are masked by the softmax (key <= klimit), their probabilities are exact zeros in front of finite values
            const int id = min(tid + NT * i, 511), r = id >> 4, c = id & 15;
            const size_t kr = (size_t)min(kv0 + r, a.Sk - 1);
            kreg[i] = *reinterpret_cast<const f32x4*>(Kf + kr * a.k_rs + 4 * c);
            vreg[i] = *reinterpret_cast<const f32x4*>(Vf + kr * a.v_rs + 4 * c);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = tid + NT * i, r = id >> 4, c = id & 15;
            if (NT * NCH == 512 || id < 512) {
                KT[(4 * c + 0) * AF_KT_LD + r] = rnd(kreg[i].x); KT[(4 * c + 1) * AF_KT_LD + r] = rnd(kreg[i].y);
                KT[(4 * c + 2) * AF_KT_LD + r] = rnd(kreg[i].z); KT[(4 * c + 3) * AF_KT_LD + r] = rnd(kreg[i].w);
                *reinterpret_cast<f32x4*>(&Vs[r * 64 + 4 * c]) = f32x4{rnd(vreg[i].x), rnd(vreg[i].y), rnd(vreg[i].z), rnd(vreg[i].w)};
            }
        }
    };

    attn_f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }
    float mrun = -1e30f, lsum = 0.f;
    const int klimit = a.causal_offset >= 0 ? min(a.Sk - 1, a.causal_offset + qrow) : a.Sk - 1;       // last visible key of this lane's query

    if (nt > 0) gload(0);
    for (int t = 0; t < nt; ++t) {
        __syncthreads();                                            // the previous tile is consumed
        lstore();
        __syncthreads();
        if (t + 1 < nt) gload(t + 1);                               // flies under this tile's MFMAs
        // ---- S^T = K Q^T: 32 steps of two dims --------------------------------------------------------------------------------------------
        attn_f32x16 sacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 32; ++s2) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(KT[(2 * s2 + hi) * AF_KT_LD + ln], qf[s2], sacc, 0, 0, 0);
        // ---- online softmax on the lane's 16 keys of its query: key = 32 t + (i & 3) + 8 (i >> 2) + 4 hi -----------------------------------
        const int kbase = (t << 5) + 4 * hi;
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = kbase + (i & 3) + 8 * (i >> 2);
            const float v = key <= klimit ? sacc[i] * a.scale : -INFINITY;
            sacc[i] = v;
            mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx);
        const float alpha = expf(mrun - mnew);
        mrun = mnew;
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float e = expf(sacc[i] - mnew); sacc[i] = e; ps += e; }      // exp(-inf) = 0: masked keys drop out
        lsum = lsum * alpha + ps;
#pragma unroll
        for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; }
        // ---- O^T += V^T P^T: step r contracts key (r & 3) + 8 (r >> 2) + 4 hi, which is accumulator register r of S^T ----------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * hi;
            oacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[key * 64 + ln], sacc[r], oacc[0], 0, 0, 0);
            oacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[key * 64 + 32 + ln], sacc[r], oacc[1], 0, 0, 0);
        }
    }
    const float ltot = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = ltot > 0.f ? 1.0f / ltot : 0.f;
    if (qok) {
        float* op = Of + (size_t)qrow * a.o_rs;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int j = 0; j < 4; ++j)                              // registers 4 j .. 4 j + 3: d = 32 db + 8 j + 4 hi + 0 .. 3
                *reinterpret_cast<f32x4*>(op + 32 * db + 8 * j + 4 * hi) = f32x4{oacc[db][4 * j] * inv, oacc[db][4 * j + 1] * inv, oacc[db][4 * j + 2] * inv, oacc[db][4 * j + 3] * inv};
    }
}

// ---- bf16 policy: the same flash structure on the matrix cores ------------------------------------------------------------
// One block = 64 query rows of one head, one wave = 16 of them.  Per 64-key tile: K (row-major) and V (transposed) are
// staged in LDS as bf16 (the policy's rounding point); S = Q K^T with v_mfma_f32_16x16x32_bf16 (A = Q fragment from
// registers, B = K rows straight out of LDS with ds_read_b128); the softmax runs in registers on the accumulator layout
// (row = (lane >> 4) * 4 + r, key = lane & 15: the 16 lanes of a DPP row share a query row, so row max / row sum are DPP
// reductions); P is rounded to bf16, turned into A fragments through a per-wave LDS patch, and O += P V uses the transposed
// V tile as the B operand.  Scores, softmax statistics and the output accumulate in fp32.
typedef __bf16 attn_bf16x8_t __attribute__((ext_vector_type(8)));
constexpr int AM_LD = 72;                // bf16 elements per LDS row: 64 + 8 pad (144-byte rows keep ds_read_b128 aligned)


// IT = element type of Q / K / V / O in HBM: float (rounded to bf16 while staging) or bf16_t (already rounded by the producer)
template <typename IT>
__global__ __launch_bounds__(256) void attention_mfma_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * AM_LD];
    __shared__ __attribute__((aligned(16))) bf16_t Vt[64 * AM_LD];
    __shared__ __attribute__((aligned(16))) bf16_t Ps[4][16 * AM_LD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m16 = lane & 15, kg = lane >> 4;
    const int h = blockIdx.y, q0 = blockIdx.x * 64;
    const IT* Qp = reinterpret_cast<const IT*>(a.Q) + blockIdx.z * a.q_bs; const IT* Kp = reinterpret_cast<const IT*>(a.K) + blockIdx.z * a.k_bs;
    const IT* Vp = reinterpret_cast<const IT*>(a.V) + blockIdx.z * a.v_bs; IT* Op = reinterpret_cast<IT*>(a.O) + blockIdx.z * a.o_bs;
    // 8 consecutive elements -> packed bf16 (fp32 input: rounded here; bf16 input: one 16-byte load)
    auto load8 = [](const IT* p, bool ok) -> u32x4 {
        u32x4 pk = {0u, 0u, 0u, 0u};
        if (ok) {
            if constexpr (sizeof(IT) == 4) {
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(p), t1 = *reinterpret_cast<const f32x4*>(p + 4);
                pk.x = (uint32_t)f2bf(t0.x) | ((uint32_t)f2bf(t0.y) << 16); pk.y = (uint32_t)f2bf(t0.z) | ((uint32_t)f2bf(t0.w) << 16);
                pk.z = (uint32_t)f2bf(t1.x) | ((uint32_t)f2bf(t1.y) << 16); pk.w = (uint32_t)f2bf(t1.z) | ((uint32_t)f2bf(t1.w) << 16);
            } else pk = *reinterpret_cast<const u32x4*>(p);
        }
        return pk;
    };

    // Q fragments of this wave's 16 rows: lane (row m16, dims s*32 + kg*8 .. +8)
    attn_bf16x8_t qa[2];
    {
        const int qrow = q0 + w * 16 + m16;
        const bool ok = qrow < a.Sq;
        const IT* qp = Qp + (size_t)(ok ? qrow : 0) * a.q_rs + (size_t)h * a.q_hs + kg * 8;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) qa[s2] = __builtin_bit_cast(attn_bf16x8_t, load8(qp + s2 * 32, ok));
    }
    f32x4 o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrun[4] = {-1e30f, -1e30f, -1e30f, -1e30f}, lsum[4] = {0.f, 0.f, 0.f, 0.f};

    int kv_end = a.Sk;
    if (a.causal_offset >= 0) kv_end = min(a.Sk, a.causal_offset + min(q0 + 63, a.Sq - 1) + 1);
    const int skey = tid >> 2, sd = (tid & 3) * 16;       // staging: 4 threads per key row, 16 dims each
    const int qrow_base = q0 + w * 16 + kg * 4;           // accumulator rows of this lane: qrow_base + r

    for (int kv0 = 0; kv0 < kv_end; kv0 += 64) {
        {
            const int kp = kv0 + skey;
            const bool kok = kp < a.Sk;
            const IT* kptr = Kp + (size_t)(kok ? kp : 0) * a.k_rs + (size_t)h * a.k_hs + sd;
            const IT* vptr = Vp + (size_t)(kok ? kp : 0) * a.v_rs + (size_t)h * a.v_hs + sd;
            u32x4 kk[2], vv[2];                          // this thread's 16 dims of key row skey, packed bf16
#pragma unroll
            for (int i = 0; i < 2; ++i) { kk[i] = load8(kptr + 8 * i, kok); vv[i] = load8(vptr + 8 * i, kok); }
            __syncthreads();                             // previous tile fully consumed
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(&Ks[skey * AM_LD + sd + 8 * i]) = kk[i];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t wds[4] = {vv[i].x, vv[i].y, vv[i].z, vv[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    Vt[(sd + 8 * i + 2 * j + 0) * AM_LD + skey] = (bf16_t)(wds[j] & 0xffffu);
                    Vt[(sd + 8 * i + 2 * j + 1) * AM_LD + skey] = (bf16_t)(wds[j] >> 16);
                }
            }
            __syncthreads();
        }
        // S = Q K^T for this wave's 16 rows x 64 keys
        f32x4 sc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const u32x4 kb = *reinterpret_cast<const u32x4*>(&Ks[(j * 16 + m16) * AM_LD + s2 * 32 + kg * 8]);
                sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[s2], __builtin_bit_cast(attn_bf16x8_t, kb), sc[j], 0, 0, 0);
            }
        }
        // scale + mask, online softmax per accumulator row r
        float p[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = kv0 + j * 16 + m16;
                const bool valid = key < a.Sk && (a.causal_offset < 0 || key <= a.causal_offset + qrow_base + r);
                const float v = valid ? sc[j][r] * a.scale : -INFINITY;
                p[j][r] = v;
                mx = fmaxf(mx, v);
            }
            mx = row_max16(mx);
            const float mnew = fmaxf(mrun[r], mx);
            const float alpha = expf(mrun[r] - mnew);
            mrun[r] = mnew;
            float ps = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e = (p[j][r] == -INFINITY) ? 0.f : expf(p[j][r] - mnew);
                p[j][r] = e;
                ps += e;
            }
            lsum[r] = lsum[r] * alpha + ps;               // per-lane partial of the row sum (alpha is row-uniform)
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t][r] *= alpha;
        }
        // P (bf16) -> this wave's LDS patch in row-major [row][key], then back as A fragments
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ps[w][(kg * 4 + r) * AM_LD + j * 16 + m16] = f2bf(p[j][r]);
        __syncthreads();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const u32x4 pa = *reinterpret_cast<const u32x4*>(&Ps[w][m16 * AM_LD + s2 * 32 + kg * 8]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const u32x4 vb = *reinterpret_cast<const u32x4*>(&Vt[(t * 16 + m16) * AM_LD + s2 * 32 + kg * 8]);
                o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(attn_bf16x8_t, pa), __builtin_bit_cast(attn_bf16x8_t, vb), o[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qrow = qrow_base + r;
        const float l = row_sum16(lsum[r]);
        if (qrow < a.Sq) {
            const float inv = 1.0f / l;
            IT* op = Op + (size_t)qrow * a.o_rs + h * 64 + m16;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (sizeof(IT) == 4) op[t * 16] = o[t][r] * inv;
                else op[t * 16] = f2bf(o[t][r] * inv);
            }
        }
    }
}

// round_bf16 == 1 (bf16 policy): matrix cores, fp32 tensors (kernel-level entry point) | == 3: matrix cores, bf16 tensors (the
// engine's bf16 policy) | == 0 (fp32 "exact" policy) and 2: the exact-fp32 VALU kernel above
inline hipError_t launch_attention(const AttnArgs& a, hipStream_t s) {
    if (a.Sq <= 0) return hipSuccess;
    if (a.round_bf16 == 1) hipLaunchKernelGGL(attention_mfma_kernel<float>, dim3((a.Sq + 63) / 64, a.H, a.batch), dim3(256), 0, s, a);
    else if (a.round_bf16 == 3) hipLaunchKernelGGL(attention_mfma_kernel<bf16_t>, dim3((a.Sq + 63) / 64, a.H, a.batch), dim3(256), 0, s, a);
    else {
        // 16-byte row accesses: every stride a multiple of 4 elements.  96-row blocks when they waste fewer rows than 128-row ones (257 rows)
        if ((a.q_rs | a.k_rs | a.v_rs | a.o_rs | a.q_hs | a.k_hs | a.v_hs) % 4) return hipErrorInvalidValue;
        const int pad3 = (a.Sq + 95) / 96 * 96, pad4 = (a.Sq + 127) / 128 * 128;
        if (pad3 < pad4) hipLaunchKernelGGL(attention_f32_kernel<3>, dim3(pad3 / 96, a.H, a.batch), dim3(192), 0, s, a);
        else hipLaunchKernelGGL(attention_f32_kernel<4>, dim3(pad4 / 128, a.H, a.batch), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

}  // namespace ma
