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
// first MFMA; the epilogue's operands (bias, residual, write position) are requested in front of them (gd_epi_request).
// Every kernel of this file asks for ALL it reads before it uses any of it, with clamped indices and template parameters instead of
// branches around the requests: hipcc waits for a request made inside a lane-dependent branch where the branch ends.
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
    unsigned long long* trace;           // diagnostics (ma_trace_decode): 4 stamps of the 100 MHz counter per block, or null
};
__device__ __forceinline__ void gd_stamp(const GemmDecArgs& a, int i) {
    if (a.trace && threadIdx.x == 0) a.trace[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + i] = __builtin_amdgcn_s_memrealtime();
}

// ---- epilogue of one lane: four consecutive outputs n = r0 .. r0 + 3 of batch row b ------------------------------------------------
// Its operands (bias, residual, the row's write position) are REQUESTED before the weight stream and only consumed here: read where
// they are used, each was a dependent L2 round trip of its own behind the matrix product (8 of them in a q/k/v epilogue: 1.5 us of a
// 4 us launch, profiles/r04_decode_step_timeline_b8.txt).  `vec`: the four outputs are inside N and every row stride is a multiple of
// four elements -- one 16-byte access per operand; otherwise element by element (the odd-sized lm_head).
struct GdEpi { f32x4 bias, res; int pos; };
__device__ __forceinline__ bool gd_vec(const GemmDecArgs& a, int n0) {
    return n0 + 16 <= a.N && ((a.res_stride | a.y_stride | a.yb_stride) & 3) == 0;
}
__device__ __forceinline__ GdEpi gd_epi_request(const GemmDecArgs& a, int b, int r0) {
    // element by element with clamped indices, under block-uniform conditions only: a lane-dependent branch around a request (or two
    // request forms merging into one variable) makes hipcc wait for the data where the branch ends -- in front of the weight stream
    GdEpi e; e.bias = f32x4{0.f, 0.f, 0.f, 0.f}; e.res = e.bias; e.pos = 0;
    if (a.ksplit > 1) return e;
    if (a.epi == EPI_QKV) e.pos = a.st[b].pos;
    const int n1 = a.N - 1;
    if (a.bias) e.bias = f32x4{a.bias[min(r0, n1)], a.bias[min(r0 + 1, n1)], a.bias[min(r0 + 2, n1)], a.bias[min(r0 + 3, n1)]};
    if (a.res) {
        const float* rp = a.res + (size_t)b * a.res_stride;
        e.res = f32x4{rp[min(r0, n1)], rp[min(r0 + 1, n1)], rp[min(r0 + 2, n1)], rp[min(r0 + 3, n1)]};
    }
    return e;
}
template <typename HT>
__device__ __forceinline__ void gd_epi_store(const GemmDecArgs& a, const GdEpi& e, const f32x4& v, int b, int r0, bool vec) {
    if (a.ksplit > 1) {                                   // raw partial sums; bias / residual / norm happen in the consumer's prologue
        float* yp = a.y + ((size_t)blockIdx.y * a.B + b) * a.y_stride + r0;
        if (vec) *reinterpret_cast<f32x4*>(yp) = v;
        else {
            const float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) if (r0 + r < a.N) yp[r] = o[r];
        }
        return;
    }
    float x[4] = {v.x, v.y, v.z, v.w};
    const float bb[4] = {e.bias.x, e.bias.y, e.bias.z, e.bias.w}, rr[4] = {e.res.x, e.res.y, e.res.z, e.res.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (a.bias) x[r] += bb[r];
        x[r] = apply_act(x[r], a.act);
        if (a.res) x[r] += rr[r];
    }
    if (a.epi == EPI_QKV) {                               // r0 is a multiple of 4 and H of 64: the four outputs share their part and head
        const int part = r0 / a.H, c = r0 - part * a.H;
        if (part == 0) {
            float* q = a.y + (size_t)b * a.y_stride + c;
            if (vec) *reinterpret_cast<f32x4*>(q) = f32x4{x[0], x[1], x[2], x[3]};
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (r0 + r < a.N) q[r] = x[r];
            }
        } else {
            const int head = c >> 6, d = c & 63;
            const size_t off = (size_t)b * a.kv_row_stride + ((size_t)head * a.max_seq + e.pos) * 64 + d;
            bf16_t* dst = reinterpret_cast<bf16_t*>(part == 1 ? a.kcache : a.vcache) + off;
            if (vec) *reinterpret_cast<u32x2*>(dst) = pack4<HT>(f32x4{x[0], x[1], x[2], x[3]});
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (r0 + r < a.N) dst[r] = H16<HT>::bits(x[r]);
            }
        }
        return;
    }
    if (vec) {
        if (a.y) *reinterpret_cast<f32x4*>(a.y + (size_t)b * a.y_stride + r0) = f32x4{x[0], x[1], x[2], x[3]};
        if (a.yb) *reinterpret_cast<u32x2*>(a.yb + (size_t)b * a.yb_stride + r0) = pack4<HT>(f32x4{x[0], x[1], x[2], x[3]});
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (r0 + r < a.N) {
            if (a.y) a.y[(size_t)b * a.y_stride + r0 + r] = x[r];
            if (a.yb) a.yb[(size_t)b * a.yb_stride + r0 + r] = H16<HT>::bits(x[r]);
        }
    }
}

// HT (all kernels of this file): the 16-bit format of weights, activations and cache (bf16_t | f16_t, common.hpp H16)
template <int MT, int CH, typename HT = bf16_t>
__global__ __launch_bounds__(256) void gemm_dec_kernel(GemmDecArgs a) {
    __shared__ __attribute__((aligned(16))) float red[4][MT][64][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    gd_stamp(a, 0); gd_stamp(a, 1);
    const int K = a.K, Kw = K / (4 * a.ksplit);            // k-range of one wave
    const int kbase = (blockIdx.y * 4 + w) * Kw + kg * 8;
    const bf16_t* wrow = a.W + (size_t)min(n0 + m, a.N - 1) * K + kbase;
    const bf16_t* xrow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) xrow[t] = a.xb + (size_t)min(t * 16 + m, a.B - 1) * a.xb_stride + kbase;
    // wave t finishes batch tile t (MT <= 4): its epilogue operands go out in front of the weight stream
    const int eb = w * 16 + m, r0 = n0 + kg * 4;           // batch row / first of the four consecutive outputs of this lane
    const bool e_on = w < MT && eb < a.B, vec = gd_vec(a, n0);
    const GdEpi ep = gd_epi_request(a, min(eb, a.B - 1), r0);      // every lane asks (clamped row): a lane-dependent branch around the
                                                                         // requests would make hipcc wait for them at its end

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
    gd_stamp(a, 2);
    if (!e_on) return;
    f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][w][lane][0]);
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(&red[i][w][lane][0]);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    gd_epi_store<HT>(a, ep, v, eb, r0, vec);
    gd_stamp(a, 3);
}

// Small batches (B <= 16: one MFMA batch tile): the per-row LayerNorm prologue runs INSIDE the consuming GEMM instead of in a launch
// of its own (a decode step of 4-16 rows is launch-latency bound: 5 us per dependent launch).  Every block normalises all B rows itself
// -- one wave per row, rows w, w + 4, ...: sum of the producer's PARTS partial buffers in their fixed order + bias + residual, one-pass
// shifted statistics (common.hpp), wave-level reduction -- and parks them in LDS as bf16 (row stride padded by 32 bytes: the 16 rows
// of an MFMA B fragment land in different banks).  Same MFMA mapping and epilogues as gemm_dec_kernel.  K = 1024 (the hidden size).
// Everything the prologue reads is requested in as few dependent steps as the registers allow: the block's weight rows (whole K range:
// 8 loads per lane), LayerNorm parameters, deferred bias and the epilogue's operands first; then ALL vectors of a row together (PARTS
// partials + residual), two rows at a time when PARTS <= 2.  Summing partial by partial, row by row, parameters by chunk -- the first
// form of this kernel -- was ~20 dependent L2 round trips: 6.8 us of prologue in front of q/k/v, 4.6 us in front of fc1, against 2.3 us
// for the whole weight stream of a launch without prologue (profiles/r04_decode_step_timeline_b8.txt).
// DEFER: the producer's deferred epilogue (bias `pbias` + residual `pres`, both or neither) is added in front of the LayerNorm -- a
// template flag, not a branch: requests inside a branch make hipcc wait for them where the branch ends
// NW waves per block (4 | 8): with 8, eight rows are normalised one per wave and ALL their vectors are in flight at once, and each wave
// streams an eighth of the block's weight rows.
template <int PARTS, bool DEFER, int NW, typename HT = bf16_t>
__global__ __launch_bounds__(NW * 64) void gemm_dec_ln_kernel(GemmDecArgs a) {
    constexpr int K = 1024, KW = K / NW, CH = KW / 32, XS = K + 16;   // XS: LDS row stride in bf16 elements (32 bytes of padding: conflict-free fragments)
    __shared__ __attribute__((aligned(16))) float red[NW][64][4];
    __shared__ __attribute__((aligned(16))) bf16_t xl[16 * XS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    gd_stamp(a, 0);
    const int kbase = w * KW + kg * 8;
    const bf16_t* wrow = a.W + (size_t)min(n0 + m, a.N - 1) * K + kbase;
    // parameters every row shares, and the epilogue's operands (wave 0 finishes the block; lane: batch row m, outputs n0 + 4 kg ..)
    f32x4 gv[4], bv[4], pb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = (lane + 64 * j) * 4;
        gv[j] = *reinterpret_cast<const f32x4*>(a.ln_g + idx);
        bv[j] = *reinterpret_cast<const f32x4*>(a.ln_b + idx);
        if constexpr (DEFER) pb[j] = *reinterpret_cast<const f32x4*>(a.pbias + idx);
    }
    const int r0 = n0 + kg * 4;
    const bool e_on = w == 0 && m < a.B;
    const bool vec = gd_vec(a, n0);
    const GdEpi ep = gd_epi_request(a, min(m, a.B - 1), r0);       // (every lane asks, clamped row: no lane-dependent branch around the requests)
    asm volatile("" ::: "memory");

    // ---- prologue: rows w, w + 4, ... ------------------------------------------------------------------------------------------------
    const bool writer = blockIdx.x == 0 && a.xn_out;
    auto request_row = [&](int r, f32x4 (&xv)[PARTS][4], f32x4 (&rv)[4]) {
        const float* x = a.pin + (size_t)r * a.pin_stride;
#pragma unroll
        for (int p = 0; p < PARTS; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[p][j] = *reinterpret_cast<const f32x4*>(x + (size_t)p * a.B * a.pin_stride + (lane + 64 * j) * 4);
        if constexpr (DEFER) {
#pragma unroll
            for (int j = 0; j < 4; ++j) rv[j] = *reinterpret_cast<const f32x4*>(a.pres + (size_t)r * a.pres_stride + (lane + 64 * j) * 4);
        }
        asm volatile("" ::: "memory");
    };
    auto sum_row = [&](f32x4 (&xv)[PARTS][4], f32x4 (&rv)[4], f32x4 (&s)[4]) {      // the partial buffers in their order, + bias, + residual
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[j] = xv[0][j];
#pragma unroll
            for (int p = 1; p < PARTS; ++p) { s[j].x += xv[p][j].x; s[j].y += xv[p][j].y; s[j].z += xv[p][j].z; s[j].w += xv[p][j].w; }
            if constexpr (DEFER) {
                s[j].x += pb[j].x; s[j].y += pb[j].y; s[j].z += pb[j].z; s[j].w += pb[j].w;
                s[j].x += rv[j].x; s[j].y += rv[j].y; s[j].z += rv[j].z; s[j].w += rv[j].w;
            }
        }
    };
    auto norm_row = [&](int r, f32x4 (&s)[4]) {
        const float x0 = readlane_f(s[0].x, 0);            // element 0 of the summed row: the statistics' shift (lane 0 holds it)
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) ln_chunk_moments(s[j], x0, sm, sq);
        sm = wave_sum(sm); sq = wave_sum(sq);
        float md, rstd;
        ln_finish(sm, 0.f, 0.f, 0.f, sq, 0.f, 0.f, 0.f, K, a.ln_eps, md, rstd);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = (lane + 64 * j) * 4;
            ln_apply(s[j], md, rstd, gv[j], bv[j]);
            if (writer) *reinterpret_cast<f32x4*>(a.xn_out + (size_t)r * a.xn_stride + idx) = s[j];
            *reinterpret_cast<u32x2*>(&xl[r * XS + idx]) = pack4<HT>(s[j]);
        }
    };
    // Loads come back in the order they were asked for (one counter), so the activation rows -- L2 hits -- are asked for BEFORE the block's
    // weight rows -- an HBM stream: the LayerNorm arithmetic then runs while the weights are on their way.
    u32x4 wv[CH];
    int next;                                              // first row of the one-at-a-time tail
    if constexpr (NW == 8) {                               // one row per wave in the first pass (B <= 8: the only one)
        const bool one = w < a.B;
        f32x4 xa[PARTS][4], va[4], sa[4];
        if (one) request_row(w, xa, va);
#pragma unroll
        for (int s = 0; s < CH; ++s) wv[s] = ld_stream16(wrow + s * 32);
        asm volatile("" ::: "memory");
        if (one) { sum_row(xa, va, sa); norm_row(w, sa); }
        next = w + NW;
    } else {                                               // rows w and w + 4 in the first pass
        const int ra_ = w, rb_ = w + NW;
        const bool one = ra_ < a.B, two = rb_ < a.B;
        f32x4 xa[PARTS][4], xb2[PARTS][4], va[4], vb[4], sa[4], sb[4];
        if (one) request_row(ra_, xa, va);
        if constexpr (PARTS > 2) { if (one) sum_row(xa, va, sa); }     // 4 partials + residual = 80 registers: the first row collapses before
        if (two) request_row(rb_, xb2, vb);                              // the second is asked for
#pragma unroll
        for (int s = 0; s < CH; ++s) wv[s] = ld_stream16(wrow + s * 32);
        asm volatile("" ::: "memory");
        if constexpr (PARTS <= 2) { if (one) sum_row(xa, va, sa); }
        if (one) norm_row(ra_, sa);
        if (two) { sum_row(xb2, vb, sb); norm_row(rb_, sb); }
        next = w + 2 * NW;
    }
    for (int r = next; r < a.B; r += NW) {                 // more rows: one at a time
        f32x4 xa[PARTS][4], va[4], sa[4];
        request_row(r, xa, va);
        sum_row(xa, va, sa);
        norm_row(r, sa);
    }
    __syncthreads();
    gd_stamp(a, 1);

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
    gd_stamp(a, 2);
    if (!e_on) return;
    f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][lane][0]);
#pragma unroll
    for (int i = 1; i < NW; ++i) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(&red[i][lane][0]);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    gd_epi_store<HT>(a, ep, v, m, r0, vec);
    gd_stamp(a, 3);
}

template <typename HT>
inline hipError_t launch_gemm_dec_ln(const GemmDecArgs& a, hipStream_t s, int waves = 0) {
    if (a.K != 1024 || a.B < 1 || a.B > 16 || a.ksplit != 1 || !a.pin || !a.ln_g || !a.ln_b || a.pin_stride % 4 || (a.pres && a.pres_stride % 4) ||
        (a.xn_out && a.xn_stride % 4)) return hipErrorInvalidValue;
    if ((a.pbias != nullptr) != (a.pres != nullptr)) return hipErrorInvalidValue;          // the deferred epilogue comes whole
    const dim3 grid((a.N + 15) / 16);
    const bool d = a.pres != nullptr;
    // up to 8 rows: 8 waves, one row each (every vector of every row in flight together); more: 4 waves, rows w, w + 4, ...
    const bool w8 = waves == 8 || (waves == 0 && a.B <= 8);
#define MA_GDLN(P, D) do { if (w8) hipLaunchKernelGGL((gemm_dec_ln_kernel<P, D, 8, HT>), grid, dim3(512), 0, s, a); \
                           else hipLaunchKernelGGL((gemm_dec_ln_kernel<P, D, 4, HT>), grid, dim3(256), 0, s, a); } while (0)
    if (a.pin_parts == 1 && !d) MA_GDLN(1, false);
    else if (a.pin_parts == 1) MA_GDLN(1, true);
    else if (a.pin_parts == 2 && d) MA_GDLN(2, true);
    else if (a.pin_parts == 4 && d) MA_GDLN(4, true);
#undef MA_GDLN
    else return hipErrorInvalidValue;                  // (gemm_dec_ksplit gives 1, 2 or 4)
    return hipGetLastError();
}

// K split over blocks for matrices with few row tiles (every CU should stream): up to 4 when N <= 2048 and K allows it
inline int gemm_dec_ksplit(int N, int K) {
    if (N > 2048) return 1;
    if (K % (4 * 4 * 32) == 0) return 4;
    if (K % (4 * 2 * 32) == 0) return 2;
    return 1;
}

inline int& gemm_dec_chunks() { static int v = 8; return v; }      // chunks in flight per wave at 33 .. 64 rows: 8 | 4 (engine option mfma_chunks)

template <typename HT>
inline hipError_t launch_gemm_dec(const GemmDecArgs& a, hipStream_t s) {
    if (a.ksplit < 1 || a.K % (4 * a.ksplit * 32) != 0 || a.B < 1 || a.B > 64) return hipErrorInvalidValue;
    if (a.ksplit > 1 && (!a.y || a.epi != EPI_PLAIN)) return hipErrorInvalidValue;
    const dim3 grid((a.N + 15) / 16, a.ksplit), block(256);
    const int mt = (a.B + 15) / 16;
    if (mt == 1) hipLaunchKernelGGL((gemm_dec_kernel<1, 8, HT>), grid, block, 0, s, a);
    else if (mt == 2) hipLaunchKernelGGL((gemm_dec_kernel<2, 8, HT>), grid, block, 0, s, a);
    // 33 .. 64 rows: 8 chunks as well (24 / 32 activation requests per lane besides the 8 weight requests, 154 / 190 registers): the k-range
    // of a wave (256 in every decode GEMM) is then ONE round trip; with 4 chunks it was two (6.2 us to the last MFMA instead of ~4,
    // profiles/r04_decode_step_timeline_b64.txt).  gemm_dec_chunks() = 4 restores the shallow form (A/B).
    else if (mt == 3 && gemm_dec_chunks() == 4) hipLaunchKernelGGL((gemm_dec_kernel<3, 4, HT>), grid, block, 0, s, a);
    else if (mt == 3) hipLaunchKernelGGL((gemm_dec_kernel<3, 8, HT>), grid, block, 0, s, a);
    else if (gemm_dec_chunks() == 4) hipLaunchKernelGGL((gemm_dec_kernel<4, 4, HT>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_dec_kernel<4, 8, HT>), grid, block, 0, s, a);
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
// PARTS partial buffers (1 | 2 | 4), DEFER: the producer's deferred bias + residual (both or neither), NCH float4 chunks per thread
// (1: K <= 1024, 4: K <= 4096) -- template parameters, not branches: every vector of the row is requested before the first is used
// (one round trip instead of one per vector, profiles/r04_decode_step_timeline_b8.txt), and hipcc waits for the requests made inside a
// branch where the branch ends.  Chunk indices past the row are clamped for the request and masked at the use.
template <int PRO, int PARTS, bool DEFER, int NCH, typename HT = bf16_t>
__global__ __launch_bounds__(256) void rows_prologue_kernel(RowsProArgs a) {
    __shared__ float red[8];
    __shared__ float x0s;
    const int tid = threadIdx.x, b = blockIdx.x, K = a.K;
    f32x4 xv[NCH], gv[NCH], bv[NCH];
    const int nq = K / 4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (PRO == PRO_ATTN) {
        const int k = tid * 4;
        if (k < K) {
            f32x4 pml[ATTN_NCHUNK / 2], po[ATTN_NCHUNK];
            attn_partials_load(a.attn_ws + (size_t)b * a.attn_ws_stride, a.attn_heads, k >> 6, k & 63, pml, po);
            xv[0] = attn_partials_merge(pml, po);
        }
    } else {
        const float* x = a.x + (size_t)b * a.x_stride;
        f32x4 pv[PARTS][NCH], bz[NCH], rz[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int ic = min(tid + 256 * j, nq - 1) * 4;
#pragma unroll
            for (int p = 0; p < PARTS; ++p) pv[p][j] = *reinterpret_cast<const f32x4*>(x + (size_t)p * a.B * a.x_stride + ic);
            if constexpr (DEFER) {
                bz[j] = *reinterpret_cast<const f32x4*>(a.bias + ic);
                rz[j] = *reinterpret_cast<const f32x4*>(a.res + (size_t)b * a.res_stride + ic);
            }
            if constexpr (PRO == PRO_LN) {
                gv[j] = *reinterpret_cast<const f32x4*>(a.ln_g + ic);
                bv[j] = *reinterpret_cast<const f32x4*>(a.ln_b + ic);
            }
        }
        asm volatile("" ::: "memory");
        // split-K producer: the partials in their order, then the deferred epilogue (bias, residual)
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            f32x4 v = pv[0][j];
#pragma unroll
            for (int p = 1; p < PARTS; ++p) { v.x += pv[p][j].x; v.y += pv[p][j].y; v.z += pv[p][j].z; v.w += pv[p][j].w; }
            if constexpr (DEFER) {
                v.x += bz[j].x; v.y += bz[j].y; v.z += bz[j].z; v.w += bz[j].w;
                v.x += rz[j].x; v.y += rz[j].y; v.z += rz[j].z; v.w += rz[j].w;
            }
            const bool in = tid + 256 * j < nq;
            xv[j] = in ? v : z4;
            if constexpr (PRO == PRO_LN) { gv[j] = in ? gv[j] : z4; bv[j] = in ? bv[j] : z4; }
        }
    }
    if constexpr (PRO == PRO_LN) {
        // element 0 of the (summed) row -- thread 0 holds it -- is the statistics' shift
        if (tid == 0) x0s = xv[0].x;
        __syncthreads();
        ln_block_onepass<NCH>(xv, gv, bv, x0s, tid, nq, K, a.ln_eps, red);
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int idx = tid + 256 * j;
        if (idx < nq) {
            if (a.xn_out) *reinterpret_cast<f32x4*>(a.xn_out + (size_t)b * a.xn_stride + idx * 4) = xv[j];
            *reinterpret_cast<u32x2*>(a.xb + (size_t)b * a.xb_stride + idx * 4) = pack4<HT>(xv[j]);
        }
    }
}

template <int PRO, int NCH, typename HT>
inline hipError_t launch_rows_prologue_v(const RowsProArgs& a, int B, hipStream_t s) {
    const bool d = a.res != nullptr;
    const int np = a.nparts < 1 ? 1 : a.nparts;
    if ((a.bias != nullptr) != d) return hipErrorInvalidValue;                    // the deferred epilogue comes whole
#define MA_RP(P, D) hipLaunchKernelGGL((rows_prologue_kernel<PRO, P, D, NCH, HT>), dim3(B), dim3(256), 0, s, a)
    if (np == 1 && !d) MA_RP(1, false);
    else if (np == 1) MA_RP(1, true);
    else if (np == 2 && !d) MA_RP(2, false);
    else if (np == 2) MA_RP(2, true);
    else if (np == 4 && !d) MA_RP(4, false);
    else if (np == 4) MA_RP(4, true);
    else return hipErrorInvalidValue;
#undef MA_RP
    return hipGetLastError();
}

template <typename HT>
inline hipError_t launch_rows_prologue(const RowsProArgs& a, int pro, int B, hipStream_t s) {
    if (a.K % 4 != 0 || a.K > 4096 || (pro == PRO_ATTN && (a.K > 1024 || a.K != a.attn_heads * 64))) return hipErrorInvalidValue;
    if (pro == PRO_ATTN) { hipLaunchKernelGGL((rows_prologue_kernel<PRO_ATTN, 1, false, 1, HT>), dim3(B), dim3(256), 0, s, a); return hipGetLastError(); }
    if (pro == PRO_LN) {
        if (!a.ln_g || !a.ln_b) return hipErrorInvalidValue;
        return a.K <= 1024 ? launch_rows_prologue_v<PRO_LN, 1, HT>(a, B, s) : launch_rows_prologue_v<PRO_LN, 4, HT>(a, B, s);
    }
    return a.K <= 1024 ? launch_rows_prologue_v<PRO_PLAIN, 1, HT>(a, B, s) : launch_rows_prologue_v<PRO_PLAIN, 4, HT>(a, B, s);
}

}  // namespace ma
