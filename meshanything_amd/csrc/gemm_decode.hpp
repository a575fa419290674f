// Batched decode step: the weight-streaming "GEMV" when B > 1 rows step together -- a skinny GEMM
// Y[B,N] = epilogue(X[B,K] . W[N,K]^T) on the bf16 matrix cores (same reference call sites as gemv.hpp).
//
// The operand that matters is still W (2-16 MB per launch, read once from HBM whatever B is); X is B x K bf16 (<= 512 KB)
// and lives in L2.  v_mfma_f32_16x16x32_bf16 with A = a 16-row tile of W (lane: row = lane & 15, 8 consecutive k at
// (lane >> 4) * 8 -- exactly one 16-byte load from the row-major weight matrix, no LDS staging, no transpose) and
// B = 16 batch rows of X (same lane -> k mapping, one 16-byte load from the bf16 activation buffer).  D[m][n]: lane holds
// weight rows (lane >> 4) * 4 + r of batch row lane & 15, i.e. four consecutive outputs of one activation row -> one
// 16-byte store.  One block = 16 weight rows; its 4 waves split K (each streams 16 x K/4 weights, 8 loads in flight per
// lane and chunk) and meet once in LDS (fixed summation order: deterministic).  grid = (N / 16, KS): matrices with few
// rows (out_proj, fc2: N = 1024 -> 64 row tiles) are also split along K over KS blocks so that every CU streams; a split
// launch writes raw fp32 partials [KS][B][N] and its bias / residual / LayerNorm happen in the next `rows_prologue_kernel`,
// which sums the KS partials in fixed order first.  All loads of a chunk (weights AND activations) are issued before its
// first MFMA.
//
// The prologues that the batch-1 GEMV runs per block (LayerNorm of the post-LN residual stream, merge of the split-KV
// attention partials) would be repeated per block for every batch row here, so they run once per row in
// `rows_prologue_kernel` (same arithmetic, same summation order as gemv.hpp) and hand the GEMM a bf16 activation buffer.
// fp32 "exact" policy: no MFMA path; the engine uses the row-parallel GEMV (grid.y = batch row) instead.
#pragma once
#include "attn_decode.hpp"
#include "common.hpp"
#include "gemm.hpp"
#include "gemv.hpp"
#include "state.hpp"

namespace ma {

struct GemmDecArgs {
    const bf16_t* W;            // [N][K] bf16
    const float* bias;          // [N] or null
    const bf16_t* xb; int xb_stride;     // activations [B][K] bf16 (already rounded by their producer)
    const float* res; int res_stride;    // residual [B][N] fp32 or null
    float* y; int y_stride;              // fp32 output [B][N] or null; with ksplit > 1: partials [ksplit][B][y_stride] (raw sums)
    bf16_t* yb; int yb_stride;           // bf16 output [B][N] or null (feeds the next GEMM directly)
    int N, K, B, act, epi;               // epi: EPI_PLAIN | EPI_QKV
    int ksplit;                          // grid.y: blocks along K (1 = whole K in one block, epilogue applied here)
    void* kcache; void* vcache; size_t kv_row_stride; int H; int max_seq;
    const DecState* st;
    // gemm_dec_ln_kernel (B <= 16, K = 1024): the LayerNorm prologue of rows_prologue_kernel<PRO_LN> inside the GEMM -- xb is not read;
    // the activation rows are LN(sum of pin_parts partial buffers [parts][B][pin_stride] + pbias + pres), kept in LDS as bf16
    const float* pin; int pin_stride; int pin_parts; const float* pbias; const float* pres; int pres_stride;
    const float* ln_g; const float* ln_b; float ln_eps;
    float* xn_out; int xn_stride;        // LN output fp32 (a later residual), written by block (0, 0); may be null
};

// HT (all kernels of this file): the 16-bit format of weights, activations and cache (bf16_t | f16_t, common.hpp H16)
template <int MT, int CH, typename HT = bf16_t>
__global__ __launch_bounds__(256) void gemm_dec_kernel(GemmDecArgs a) {
    __shared__ __attribute__((aligned(16))) float red[4][MT][64][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int K = a.K, Kw = K / (4 * a.ksplit);            // k-range of one wave
    const int kbase = (blockIdx.y * 4 + w) * Kw + kg * 8;
    const bf16_t* wrow = a.W + (size_t)min(n0 + m, a.N - 1) * K + kbase;
    const bf16_t* xrow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) xrow[t] = a.xb + (size_t)min(t * 16 + m, a.B - 1) * a.xb_stride + kbase;

    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < Kw; k0 += CH * 32) {
        u32x4 wv[CH], xv[CH][MT];
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const bool ok = k0 + s * 32 < Kw;
            wv[s] = ok ? ld_stream16(wrow + k0 + s * 32) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int t = 0; t < MT; ++t) xv[s][t] = ok ? *reinterpret_cast<const u32x4*>(xrow[t] + k0 + s * 32) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int s = 0; s < CH; ++s)
#pragma unroll
            for (int t = 0; t < MT; ++t)
                acc[t] = H16<HT>::mfma16(wv[s], xv[s][t], acc[t]);
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) *reinterpret_cast<f32x4*>(&red[w][t][lane][0]) = acc[t];
    __syncthreads();
    // epilogue: wave t finishes batch tile t (MT <= 4)
    if (w >= MT) return;
    const int t = w;
    f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][t][lane][0]);
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(&red[i][t][lane][0]);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    const int b = t * 16 + m;                             // batch row of this lane
    if (b >= a.B) return;
    const int r0 = n0 + kg * 4;                           // first of this lane's four consecutive output rows
    float o[4] = {v.x, v.y, v.z, v.w};
    if (a.ksplit > 1) {                                   // raw partial sums; bias / residual / norm happen in the consumer's prologue
        float* yp = a.y + ((size_t)blockIdx.y * a.B + b) * a.y_stride;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (r0 + r < a.N) yp[r0 + r] = o[r];
        return;
    }
    const int pos = a.epi == EPI_QKV ? a.st[b].pos : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = r0 + r;
        if (n >= a.N) continue;
        float x = o[r];
        if (a.bias) x += a.bias[n];
        x = apply_act(x, a.act);
        if (a.res) x += a.res[(size_t)b * a.res_stride + n];
        if (a.epi == EPI_QKV) {
            const int part = n / a.H, c = n - part * a.H;
            if (part == 0) a.y[(size_t)b * a.y_stride + c] = x;
            else {
                const int head = c >> 6, d = c & 63;
                const size_t off = (size_t)b * a.kv_row_stride + ((size_t)head * a.max_seq + pos) * 64 + d;
                reinterpret_cast<bf16_t*>(part == 1 ? a.kcache : a.vcache)[off] = H16<HT>::bits(x);
            }
        } else {
            if (a.y) a.y[(size_t)b * a.y_stride + n] = x;
            if (a.yb) a.yb[(size_t)b * a.yb_stride + n] = H16<HT>::bits(x);
        }
    }
}

// Small batches (B <= 16: one MFMA batch tile): the per-row LayerNorm prologue runs INSIDE the consuming GEMM instead of in a launch
// of its own (a decode step of 4-16 rows is launch-latency bound: 5 us per dependent launch).  Every block normalises all B rows itself
// -- one wave per row, rows w, w + 4, ...: sum of the producer's partial buffers in their fixed order + bias + residual, one-pass
// shifted statistics (common.hpp), wave-level reduction -- and parks them in LDS as bf16 (row stride padded by 32 bytes: the 16 rows
// of an MFMA B fragment land in different banks).  The weight rows of the block (its whole K range: 8 loads per lane) are requested
// before the prologue.  Same MFMA mapping and epilogues as gemm_dec_kernel.  K = 1024 (the hidden size).
template <typename HT = bf16_t>
__global__ __launch_bounds__(256) void gemm_dec_ln_kernel(GemmDecArgs a) {
    constexpr int K = 1024, CH = 8, XS = K + 16;             // XS: LDS row stride in bf16 elements (32 bytes of padding: conflict-free fragments)
    __shared__ __attribute__((aligned(16))) float red[4][64][4];
    __shared__ __attribute__((aligned(16))) bf16_t xl[16 * XS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int kbase = w * 256 + kg * 8;
    const bf16_t* wrow = a.W + (size_t)min(n0 + m, a.N - 1) * K + kbase;
    u32x4 wv[CH];
#pragma unroll
    for (int s = 0; s < CH; ++s) wv[s] = ld_stream16(wrow + s * 32);
    asm volatile("" ::: "memory");

    // ---- prologue: rows w, w + 4, ...; two rows per pass so that the second row's loads fly under the first row's arithmetic -------------
    const bool writer = blockIdx.x == 0 && a.xn_out;
    auto load_row = [&](int r, f32x4 (&xv)[4], float& x0) {          // sum of the partial buffers in their order, + bias, + residual
        const float* x = a.pin + (size_t)r * a.pin_stride;
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = *reinterpret_cast<const f32x4*>(x + (lane + 64 * j) * 4);
        x0 = x[0];
        for (int p = 1; p < a.pin_parts; ++p) {
            const float* xp = x + (size_t)p * a.B * a.pin_stride;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const f32x4 t = *reinterpret_cast<const f32x4*>(xp + (lane + 64 * j) * 4); xv[j].x += t.x; xv[j].y += t.y; xv[j].z += t.z; xv[j].w += t.w; }
            x0 += xp[0];
        }
        if (a.pbias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const f32x4 t = *reinterpret_cast<const f32x4*>(a.pbias + (lane + 64 * j) * 4); xv[j].x += t.x; xv[j].y += t.y; xv[j].z += t.z; xv[j].w += t.w; }
            x0 += a.pbias[0];
        }
        if (a.pres) {
            const float* rp = a.pres + (size_t)r * a.pres_stride;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const f32x4 t = *reinterpret_cast<const f32x4*>(rp + (lane + 64 * j) * 4); xv[j].x += t.x; xv[j].y += t.y; xv[j].z += t.z; xv[j].w += t.w; }
            x0 += rp[0];
        }
    };
    auto finish_row = [&](int r, f32x4 (&xv)[4], float x0) {
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) ln_chunk_moments(xv[j], x0, sm, sq);
        sm = wave_sum(sm); sq = wave_sum(sq);
        float md, rstd;
        ln_finish(sm, 0.f, 0.f, 0.f, sq, 0.f, 0.f, 0.f, K, a.ln_eps, md, rstd);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = (lane + 64 * j) * 4;
            const f32x4 g = *reinterpret_cast<const f32x4*>(a.ln_g + idx), bb = *reinterpret_cast<const f32x4*>(a.ln_b + idx);
            ln_apply(xv[j], md, rstd, g, bb);
            if (writer) *reinterpret_cast<f32x4*>(a.xn_out + (size_t)r * a.xn_stride + idx) = xv[j];
            *reinterpret_cast<u32x2*>(&xl[r * XS + idx]) = pack4<HT>(xv[j]);
        }
    };
    for (int r = w; r < a.B; r += 8) {
        f32x4 xa[4], xb2[4];
        float x0a, x0b = 0.f;
        const bool two = r + 4 < a.B;
        load_row(r, xa, x0a);
        if (two) load_row(r + 4, xb2, x0b);
        finish_row(r, xa, x0a);
        if (two) finish_row(r + 4, xb2, x0b);
    }
    __syncthreads();

    // ---- MFMA: A = the block's 16 weight rows, B = the 16 (<= B valid) activation rows from LDS ---------------------------------------
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bf16_t* xr = xl + min(m, a.B - 1) * XS + kbase;
#pragma unroll
    for (int s = 0; s < CH; ++s) {
        const u32x4 xv = *reinterpret_cast<const u32x4*>(xr + s * 32);
        acc = H16<HT>::mfma16(wv[s], xv, acc);
    }
    *reinterpret_cast<f32x4*>(&red[w][lane][0]) = acc;
    __syncthreads();
    if (w != 0) return;
    f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][lane][0]);
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(&red[i][lane][0]);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    const int b = m;
    if (b >= a.B) return;
    const int r0 = n0 + kg * 4;
    float o[4] = {v.x, v.y, v.z, v.w};
    const int pos = a.epi == EPI_QKV ? a.st[b].pos : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = r0 + r;
        if (n >= a.N) continue;
        float x = o[r];
        if (a.bias) x += a.bias[n];
        x = apply_act(x, a.act);
        if (a.res) x += a.res[(size_t)b * a.res_stride + n];
        if (a.epi == EPI_QKV) {
            const int part = n / a.H, c = n - part * a.H;
            if (part == 0) a.y[(size_t)b * a.y_stride + c] = x;
            else {
                const int head = c >> 6, d = c & 63;
                const size_t off = (size_t)b * a.kv_row_stride + ((size_t)head * a.max_seq + pos) * 64 + d;
                reinterpret_cast<bf16_t*>(part == 1 ? a.kcache : a.vcache)[off] = H16<HT>::bits(x);
            }
        } else {
            if (a.y) a.y[(size_t)b * a.y_stride + n] = x;
            if (a.yb) a.yb[(size_t)b * a.yb_stride + n] = H16<HT>::bits(x);
        }
    }
}

template <typename HT>
inline hipError_t launch_gemm_dec_ln(const GemmDecArgs& a, hipStream_t s) {
    if (a.K != 1024 || a.B < 1 || a.B > 16 || a.ksplit != 1 || !a.pin || a.pin_parts < 1 || !a.ln_g || !a.ln_b || a.pin_stride % 4 || (a.pres && a.pres_stride % 4) ||
        (a.xn_out && a.xn_stride % 4)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((gemm_dec_ln_kernel<HT>), dim3((a.N + 15) / 16), dim3(256), 0, s, a);
    return hipGetLastError();
}

// K split over blocks for matrices with few row tiles (every CU should stream): up to 4 when N <= 2048 and K allows it
inline int gemm_dec_ksplit(int N, int K) {
    if (N > 2048) return 1;
    if (K % (4 * 4 * 32) == 0) return 4;
    if (K % (4 * 2 * 32) == 0) return 2;
    return 1;
}

template <typename HT>
inline hipError_t launch_gemm_dec(const GemmDecArgs& a, hipStream_t s) {
    if (a.ksplit < 1 || a.K % (4 * a.ksplit * 32) != 0 || a.B < 1 || a.B > 64) return hipErrorInvalidValue;
    if (a.ksplit > 1 && (!a.y || a.epi != EPI_PLAIN)) return hipErrorInvalidValue;
    const dim3 grid((a.N + 15) / 16, a.ksplit), block(256);
    const int mt = (a.B + 15) / 16;
    if (mt == 1) hipLaunchKernelGGL((gemm_dec_kernel<1, 8, HT>), grid, block, 0, s, a);
    else if (mt == 2) hipLaunchKernelGGL((gemm_dec_kernel<2, 8, HT>), grid, block, 0, s, a);
    else if (mt == 3) hipLaunchKernelGGL((gemm_dec_kernel<3, 4, HT>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_dec_kernel<4, 4, HT>), grid, block, 0, s, a);
    return hipGetLastError();
}

// ---- per-row prologue: one block per batch row --------------------------------------------------------------------------
struct RowsProArgs {
    const float* x; int x_stride;        // [B][K] fp32 (PRO_PLAIN / PRO_LN); with nparts > 1: partials [nparts][B][x_stride]
    int nparts; int B;                   // x = sum of nparts partial buffers (fixed order) + bias + res when nparts >= 1
    const float* bias;                   // [K] or null, added to the summed input
    const float* res; int res_stride;    // [B][K] fp32 or null, added to the summed input
    const float* ln_g; const float* ln_b; float ln_eps;
    const float* attn_ws; size_t attn_ws_stride; int attn_heads;      // PRO_ATTN: K = heads * 64
    float* xn_out; int xn_stride;        // prologue(x) fp32 (the residual of a later epilogue) or null
    bf16_t* xb; int xb_stride;           // prologue(x) rounded to bf16: the GEMM's activation operand
    int K;
};

// thread t owns the float4 chunks t, t+256, ... of its row: the same partition, arithmetic and summation order as the
// block-level prologue of gemv_kernel, so a batched row sees the same activation bits as a batch-1 run
template <int PRO, typename HT = bf16_t>
__global__ __launch_bounds__(256) void rows_prologue_kernel(RowsProArgs a) {
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b = blockIdx.x, K = a.K;
    constexpr int NCH = 4;                               // K <= 4096
    f32x4 xv[NCH];
    const int nq = K / 4;
    if constexpr (PRO == PRO_ATTN) {
        const int k = tid * 4;
        if (k < K) {
            f32x4 pml[ATTN_NCHUNK / 2], po[ATTN_NCHUNK];
            attn_partials_load(a.attn_ws + (size_t)b * a.attn_ws_stride, a.attn_heads, k >> 6, k & 63, pml, po);
            xv[0] = attn_partials_merge(pml, po);
        }
    } else {
        const float* x = a.x + (size_t)b * a.x_stride;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int idx = tid + 256 * j;
            xv[j] = idx < nq ? *reinterpret_cast<const f32x4*>(x + idx * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // split-K producer: add the other partials in order, then the deferred epilogue (bias, residual)
        for (int p = 1; p < a.nparts; ++p) {
            const float* xp = x + (size_t)p * a.B * a.x_stride;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int idx = tid + 256 * j;
                if (idx < nq) { const f32x4 t = *reinterpret_cast<const f32x4*>(xp + idx * 4); xv[j].x += t.x; xv[j].y += t.y; xv[j].z += t.z; xv[j].w += t.w; }
            }
        }
        if (a.bias) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int idx = tid + 256 * j;
                if (idx < nq) { const f32x4 t = *reinterpret_cast<const f32x4*>(a.bias + idx * 4); xv[j].x += t.x; xv[j].y += t.y; xv[j].z += t.z; xv[j].w += t.w; }
            }
        }
        if (a.res) {
            const float* rp = a.res + (size_t)b * a.res_stride;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int idx = tid + 256 * j;
                if (idx < nq) { const f32x4 t = *reinterpret_cast<const f32x4*>(rp + idx * 4); xv[j].x += t.x; xv[j].y += t.y; xv[j].z += t.z; xv[j].w += t.w; }
            }
        }
    }
    if constexpr (PRO == PRO_LN) {
        // element 0 of the (summed) row, computed by every thread exactly as thread 0 computes it: the statistics' shift
        const float* x = a.x + (size_t)b * a.x_stride;
        float x0 = x[0];
        for (int p = 1; p < a.nparts; ++p) x0 += x[(size_t)p * a.B * a.x_stride];
        if (a.bias) x0 += a.bias[0];
        if (a.res) x0 += a.res[(size_t)b * a.res_stride];
        f32x4 gv[NCH], bv[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int idx = tid + 256 * j;
            gv[j] = idx < nq ? *reinterpret_cast<const f32x4*>(a.ln_g + idx * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            bv[j] = idx < nq ? *reinterpret_cast<const f32x4*>(a.ln_b + idx * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        ln_block_onepass<NCH>(xv, gv, bv, x0, tid, nq, K, a.ln_eps, red);
    }
    constexpr int NJ = PRO == PRO_ATTN ? 1 : NCH;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int idx = tid + 256 * j;
        if (idx < nq) {
            if (a.xn_out) *reinterpret_cast<f32x4*>(a.xn_out + (size_t)b * a.xn_stride + idx * 4) = xv[j];
            *reinterpret_cast<u32x2*>(a.xb + (size_t)b * a.xb_stride + idx * 4) = pack4<HT>(xv[j]);
        }
    }
}

template <typename HT>
inline hipError_t launch_rows_prologue(const RowsProArgs& a, int pro, int B, hipStream_t s) {
    if (a.K % 4 != 0 || a.K > 4096 || (pro == PRO_ATTN && (a.K > 1024 || a.K != a.attn_heads * 64))) return hipErrorInvalidValue;
    if (pro == PRO_LN) hipLaunchKernelGGL((rows_prologue_kernel<PRO_LN, HT>), dim3(B), dim3(256), 0, s, a);
    else if (pro == PRO_ATTN) hipLaunchKernelGGL((rows_prologue_kernel<PRO_ATTN, HT>), dim3(B), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((rows_prologue_kernel<PRO_PLAIN, HT>), dim3(B), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace ma
