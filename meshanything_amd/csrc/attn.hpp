// Dense attention for the encoder (257 x 4096 cross, 257 x 257 self), the decoder prefill (257 x 257 causal) and the
// detokenizer (1057 x 1057): O = softmax_fp32(Q K^T * scale) V, head_dim 64.
// Reference: QKVMultiheadAttention / QKVMultiheadCrossAttention (transformer_blocks.py:56-74, 166-185),
// [3p] flash_attn prefill (shape_opt.py:403-410) and [3p] BERT self-attention (meshanything.py:62-64).
//
// Round-1 kernel: flash-style (online softmax, K/V tiles staged in LDS, nothing S x S ever materialised) on the fp32
// VALU with exact fp32 probabilities, so that both precision policies have oracle-reproducible rounding points
// (q, k, v optionally rounded to bf16; P stays fp32).  One block = 64 query rows of one head; thread (r, c) owns query
// row r and the keys j = 4*jj + c of each 64-key tile, keeps a partial (l, o[64]) for them, and the four threads of a
// row are merged once at the end -- no P exchange.  LDS rows are padded to 68 floats: the four key rows a wave reads
// per instruction fall on disjoint banks.  An MFMA version (swapped QK^T, in-register softmax) is the planned upgrade.
#pragma once
#include "common.hpp"

namespace ma {

struct AttnArgs {
    const float* Q; int q_rs, q_hs;     // element strides: row (sequence position), head
    const float* K; int k_rs, k_hs;
    const float* V; int v_rs, v_hs;
    float* O; int o_rs;                 // O[q * o_rs + h*64 + d]
    int Sq, Sk, H;
    float scale;
    int causal_offset;                  // < 0: full attention; else query i sees keys <= causal_offset + i
    int round_bf16;
};

constexpr int ATT_LD = 68;

__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Ks[64 * ATT_LD];
    __shared__ __attribute__((aligned(16))) float Vs[64 * ATT_LD];
    const int tid = threadIdx.x, r = tid >> 2, c = tid & 3;
    const int h = blockIdx.y;
    const int q0 = blockIdx.x * 64;
    const int m = q0 + r;
    const bool row_ok = m < a.Sq;

    float q[64];
    {
        const float* qp = a.Q + (size_t)(row_ok ? m : 0) * a.q_rs + (size_t)h * a.q_hs;
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
                const float* kptr = a.K + (size_t)kp * a.k_rs + (size_t)h * a.k_hs + sd;
                const float* vptr = a.V + (size_t)kp * a.v_rs + (size_t)h * a.v_hs + sd;
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
        float* op = a.O + (size_t)m * a.o_rs + h * 64;
#pragma unroll
        for (int d = 0; d < 64; ++d)
            if ((d >> 4) == c) op[d] = o[d];
    }
}

inline hipError_t launch_attention(const AttnArgs& a, hipStream_t s) {
    if (a.Sq <= 0) return hipSuccess;
    hipLaunchKernelGGL(attention_kernel, dim3((a.Sq + 63) / 64, a.H), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace ma
