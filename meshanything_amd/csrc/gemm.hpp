// C[M,N] = act(A[M,K] . W[N,K]^T + bias) + R, fp32 "exact" policy -- the dense (M = B x 257 / B x 1057 / B x 4096 rows)
// projections of the point encoder, the decoder prefill and the detokenizer (reference: every nn.Linear in
// transformer_blocks.py, sal_perceiver.py:45,90,383-396,273-275, meshanything.py:42-80,125-132, and [3p] OPT/BERT layers at
// prefill).  The bf16 policy's GEMM is gemm_tile.hpp; this file keeps the exact-fp32 matrix-core kernel and a scalar
// cross-check kernel.
//
// Both operands are K-contiguous ("NT"), which is exactly what MFMA fragments want.  64x64 block tile, 4 waves as
// 2x2, each wave a 32x32 sub-tile = 2x2 MFMA 16x16 tiles, v_mfma_f32_16x16x4_f32 (an exact fp32 FMA chain), BK = 16.
// Operands are staged global -> registers -> LDS with the next tile's global loads issued before the current tile's MFMAs.
// C/D fragment: col = lane & 15, row = (lane >> 4) * 4 + reg  (MI355X guide section 3).
#pragma once
#include "common.hpp"

namespace ma {

struct GemmArgs {
    const float* A; int lda;
    const void* W;              // [N][K]
    const float* bias;          // [N] or null
    const float* R; int ldr;    // residual [M][N] or null (may alias C)
    float* C; int ldc;
    int M, N, K, act;
    int r_mod;                  // > 0: residual row = m % r_mod (a per-sample table broadcast over the batch: the encoder's query)
    RowMap cmap;                // output row of logical row m (rows of a batch landing inside larger per-sample blocks)
};

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ inline void gemm_epilogue(const GemmArgs& g, const f32x4 (&acc)[2][2], int bm, int bn, int wm, int wn, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = bn + wn * 32 + j * 16 + (lane & 15);
            if (n >= g.N) continue;
            const float b = g.bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = bm + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
                if (m >= g.M) continue;
                float v = apply_act(acc[i][j][r] + b, g.act);
                if (g.R) v += g.R[(size_t)(g.r_mod > 0 ? m % g.r_mod : m) * g.ldr + n];
                g.C[g.cmap(m) * g.ldc + n] = v;
            }
        }
}

__global__ __launch_bounds__(256) void gemm_mfma_f32_kernel(GemmArgs g) {
    constexpr int BK = 16, LDS_LD = 20;                 // 80-byte rows
    __shared__ __attribute__((aligned(16))) float As[64 * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[64 * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    const int bm = blockIdx.y * 64, bn = blockIdx.x * 64;
    const float* W = reinterpret_cast<const float*>(g.W);
    const int srow = tid >> 2, skc = (tid & 3) * 4;
    const int am = bm + srow, wnrow = bn + srow;
    const bool a_ok = am < g.M, w_ok = wnrow < g.N;
    const float* ap = g.A + (size_t)(a_ok ? am : 0) * g.lda + skc;
    const float* wp = W + (size_t)(w_ok ? wnrow : 0) * g.K + skc;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 ar = {0, 0, 0, 0}, wr = {0, 0, 0, 0};
    if (a_ok) ar = *reinterpret_cast<const f32x4*>(ap);
    if (w_ok) wr = *reinterpret_cast<const f32x4*>(wp);

    for (int k0 = 0; k0 < g.K; k0 += BK) {
        *reinterpret_cast<f32x4*>(&As[srow * LDS_LD + skc]) = ar;
        *reinterpret_cast<f32x4*>(&Bs[srow * LDS_LD + skc]) = wr;
        __syncthreads();
        const int kn = k0 + BK;
        if (kn < g.K) {
            if (a_ok) ar = *reinterpret_cast<const f32x4*>(ap + kn);
            if (w_ok) wr = *reinterpret_cast<const f32x4*>(wp + kn);
        }
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            float af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = As[(wm * 32 + i * 16 + (lane & 15)) * LDS_LD + ks * 4 + (lane >> 4)];
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = Bs[(wn * 32 + j * 16 + (lane & 15)) * LDS_LD + ks * 4 + (lane >> 4)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    gemm_epilogue(g, acc, bm, bn, wm, wn, lane);
}

// Plain one-thread-per-output kernel with the same rounding points: a cross-check for the MFMA kernels
// (ma_op_gemm impl=1) and an emergency switch (`gemm_impl` option); never the default.
template <typename WT>
__global__ __launch_bounds__(256) void gemm_ref_kernel(GemmArgs g) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int m = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (m >= g.M || n >= g.N) return;
    const WT* W = reinterpret_cast<const WT*>(g.W) + (size_t)n * g.K;
    const float* a = g.A + (size_t)m * g.lda;
    float acc = 0.f;
    for (int k = 0; k < g.K; ++k) {
        float av = a[k], wv;
        if constexpr (sizeof(WT) == 4) wv = W[k];
        else { av = round_bf16(av); wv = bf2f(W[k]); }
        acc = fmaf(av, wv, acc);
    }
    float v = apply_act(acc + (g.bias ? g.bias[n] : 0.f), g.act);
    if (g.R) v += g.R[(size_t)(g.r_mod > 0 ? m % g.r_mod : m) * g.ldr + n];
    g.C[g.cmap(m) * g.ldc + n] = v;
}

template <typename WT>
inline hipError_t launch_gemm(const GemmArgs& g, int impl, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if (impl == 1) {
        hipLaunchKernelGGL((gemm_ref_kernel<WT>), dim3((g.N + 63) / 64, (g.M + 3) / 4), dim3(256), 0, s, g);
        return hipGetLastError();
    }
    if (g.K % 32 != 0 || g.lda % 4 != 0) return hipErrorInvalidValue;
    dim3 grid((g.N + 63) / 64, (g.M + 63) / 64);
    if constexpr (sizeof(WT) == 4) hipLaunchKernelGGL(gemm_mfma_f32_kernel, grid, dim3(256), 0, s, g);
    else return hipErrorInvalidValue;                       // bf16 weights: gemm_tile.hpp (bf16 activations)
    return hipGetLastError();
}

}  // namespace ma
