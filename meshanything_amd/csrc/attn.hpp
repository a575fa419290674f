// Dense attention for the encoder (257 x 4096 cross, 257 x 257 self), the decoder prefill (257 x 257 causal) and the
// detokenizer (1057 x 1057): O = softmax_fp32(Q K^T * scale) V, head_dim 64.
// Reference: QKVMultiheadAttention / QKVMultiheadCrossAttention (transformer_blocks.py:56-74, 166-185),
// [3p] flash_attn prefill (shape_opt.py:403-410) and [3p] BERT self-attention (meshanything.py:62-64).
//
// Two kernels, both flash-style (online softmax, K/V tiles staged in LDS, nothing S x S ever materialised):
//  * attention_kernel (fp32 "exact" policy): fp32 VALU with exact fp32 probabilities.  One block = 64 query rows of one
//    head; thread (r, c) owns query row r and the keys j = 4*jj + c of each 64-key tile, keeps a partial (l, o[64]) for
//    them, and the four threads of a row are merged once at the end -- no P exchange.  LDS rows are padded to 68 floats:
//    the four key rows a wave reads per instruction fall on disjoint banks.
//  * attention_mfma_kernel (bf16 policy, further down): the same structure on the matrix cores.
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

constexpr int ATT_LD = 68;

__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Ks[64 * ATT_LD];
    __shared__ __attribute__((aligned(16))) float Vs[64 * ATT_LD];
    const int tid = threadIdx.x, r = tid >> 2, c = tid & 3;
    const int h = blockIdx.y;
    const int q0 = blockIdx.x * 64;
    const float* Qf = reinterpret_cast<const float*>(a.Q) + blockIdx.z * a.q_bs; const float* Kf = reinterpret_cast<const float*>(a.K) + blockIdx.z * a.k_bs;
    const float* Vf = reinterpret_cast<const float*>(a.V) + blockIdx.z * a.v_bs; float* Of = reinterpret_cast<float*>(a.O) + blockIdx.z * a.o_bs;
    const int m = q0 + r;
    const bool row_ok = m < a.Sq;

    float q[64];
    {
        const float* qp = Qf + (size_t)(row_ok ? m : 0) * a.q_rs + (size_t)h * a.q_hs;
#pragma unroll
        for (int d = 0; d < 64; d += 4) {
            f32x4 t = *reinterpret_cast<const f32x4*>(qp + d);
            q[d] = t.x; q[d + 1] = t.y; q[d + 2] = t.z; q[d + 3] = t.w;
        }
        if (a.round_bf16) {
#pragma unroll
            for (int d = 0; d < 64; ++d) q[d] = round_bf16(q[d]);
        }
    }
    float o[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) o[d] = 0.f;
    float mrun = -1e30f, l = 0.f;

    int kv_end = a.Sk;
    if (a.causal_offset >= 0) kv_end = min(a.Sk, a.causal_offset + min(q0 + 63, a.Sq - 1) + 1);

    const int spos = tid >> 2, sd = (tid & 3) * 16;     // staging: 4 threads per key row, 16 dims each
    for (int kv0 = 0; kv0 < kv_end; kv0 += 64) {
        {
            const int kp = kv0 + spos;
            f32x4 kk[4], vv[4];
            if (kp < a.Sk) {
                const float* kptr = Kf + (size_t)kp * a.k_rs + (size_t)h * a.k_hs + sd;
                const float* vptr = Vf + (size_t)kp * a.v_rs + (size_t)h * a.v_hs + sd;
#pragma unroll
                for (int i = 0; i < 4; ++i) { kk[i] = *reinterpret_cast<const f32x4*>(kptr + 4 * i); vv[i] = *reinterpret_cast<const f32x4*>(vptr + 4 * i); }
                if (a.round_bf16) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        kk[i].x = round_bf16(kk[i].x); kk[i].y = round_bf16(kk[i].y); kk[i].z = round_bf16(kk[i].z); kk[i].w = round_bf16(kk[i].w);
                        vv[i].x = round_bf16(vv[i].x); vv[i].y = round_bf16(vv[i].y); vv[i].z = round_bf16(vv[i].z); vv[i].w = round_bf16(vv[i].w);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) { kk[i] = f32x4{0, 0, 0, 0}; vv[i] = f32x4{0, 0, 0, 0}; }
            }
            __syncthreads();                             // previous tile fully consumed
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<f32x4*>(&Ks[spos * ATT_LD + sd + 4 * i]) = kk[i];
                *reinterpret_cast<f32x4*>(&Vs[spos * ATT_LD + sd + 4 * i]) = vv[i];
            }
            __syncthreads();
        }
        // scores for keys j = 4*jj + c
        float s[16];
        float tmax = -1e30f;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int j = jj * 4 + c;
            const float* kr = &Ks[j * ATT_LD];
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < 64; d += 4) {
                f32x4 t = *reinterpret_cast<const f32x4*>(kr + d);
                acc = fmaf(q[d], t.x, acc); acc = fmaf(q[d + 1], t.y, acc); acc = fmaf(q[d + 2], t.z, acc); acc = fmaf(q[d + 3], t.w, acc);
            }
            const int kp = kv0 + j;
            const bool valid = kp < a.Sk && (a.causal_offset < 0 || kp <= a.causal_offset + m);
            s[jj] = valid ? acc * a.scale : -INFINITY;
            tmax = fmaxf(tmax, s[jj]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 1, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 2, 64));
        const float mnew = fmaxf(mrun, tmax);
        const float alpha = expf(mrun - mnew);
        mrun = mnew;
        l *= alpha;
#pragma unroll
        for (int d = 0; d < 64; ++d) o[d] *= alpha;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const float p = (s[jj] == -INFINITY) ? 0.f : expf(s[jj] - mnew);
            l += p;
            const float* vr = &Vs[(jj * 4 + c) * ATT_LD];
#pragma unroll
            for (int d = 0; d < 64; d += 4) {
                f32x4 t = *reinterpret_cast<const f32x4*>(vr + d);
                o[d] = fmaf(p, t.x, o[d]); o[d + 1] = fmaf(p, t.y, o[d + 1]); o[d + 2] = fmaf(p, t.z, o[d + 2]); o[d + 3] = fmaf(p, t.w, o[d + 3]);
            }
        }
    }
    // merge the four key-subsets of each query row (they share mrun by construction)
    l += __shfl_xor(l, 1, 64);
    l += __shfl_xor(l, 2, 64);
    const float inv = 1.0f / l;
#pragma unroll
    for (int d = 0; d < 64; ++d) {
        float t = o[d];
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        o[d] = t * inv;
    }
    if (row_ok) {
        float* op = Of + (size_t)m * a.o_rs + h * 64;
#pragma unroll
        for (int d = 0; d < 64; ++d)
            if ((d >> 4) == c) op[d] = o[d];
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
    else hipLaunchKernelGGL(attention_kernel, dim3((a.Sq + 63) / 64, a.H, a.batch), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace ma
