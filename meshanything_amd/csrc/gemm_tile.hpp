// Dense GEMM of the bf16 policy: C[M,N] = act(A[M,K] . W[N,K]^T + bias) + R with BOTH operands bf16 in HBM.
//
// Reference call sites: every nn.Linear of the point encoder (transformer_blocks.py, sal_perceiver.py:45,90,273-275,383-396),
// the decoder prefill ([3p] OPTDecoderLayer at S = 257, shape_opt.py:403-410) and the detokenizer (meshanything.py:42-80),
// with the samples of a batch stacked along M (M = B x 257 | B x 4096 | B x 1057).
//
// Structure (MI355X guide section 5, the 128 x 128 LDS-staged tile with global_load_lds):
//   * block tile 128 (M) x 128 (N) x 64 (K), 4 waves as 2 x 2, each wave a 64 x 64 sub-tile = 4 x 4 v_mfma_f32_16x16x32_bf16
//     tiles (16 fp32x4 accumulators); a 64 x 64 x 32 variant serves the small problems (M <= 64, pre_kl N = 64, tiny shapes);
//   * both operands go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPR round trip) into a
//     ring of NS LDS stages: the loads of K-tiles t + 1 .. t + NS - 1 fly under the MFMAs of K-tile t (s_waitcnt vmcnt(n) lets
//     the younger tiles stay in flight), one barrier per K-tile.  What bounds the kernel is the bytes in flight per CU against
//     the ~1.5 us loaded memory latency (dense phases), hence the ring;
//   * the LDS image is lane-linear (the DMA cannot scatter), so the bank-conflict swizzle is applied to the SOURCE address:
//     16-byte chunk c of tile row r is fetched into slot c ^ swz(r) and read back from there by ds_read_b128 (swz = r & 7 for
//     128-byte tile rows, (r >> 1) & 3 for 64-byte rows: conflict-free for the hardware's 16-lane ds_read_b128 groups);
//   * the W tile is the MFMA A operand and the activation tile the B operand, so a lane ends up with FOUR CONSECUTIVE n of one
//     output row m: bias / residual / outputs move as 16-byte (fp32) or 8-byte (bf16) vectors;
//   * epilogue fused: bias, ReLU / GELU(erf), fp32 residual; the result is stored as fp32 (residual stream, LayerNorm input)
//     or as bf16 (the next GEMM's / attention's operand -- the policy's rounding point moves from that consumer's load to here,
//     same value), so activations cross HBM once, in 2 bytes.
// Accumulation order along K is k-ascending in 32-wide MFMA steps, exactly as the 64 x 64 kernel of round 1 (gemm.hpp).
// Algorithmic FLOPs: 2 M N K.
#pragma once
#include "common.hpp"
#include "gemm.hpp"

namespace ma {

struct GemmTArgs {
    const bf16_t* A; int lda;       // [M][lda] bf16, K contiguous
    const bf16_t* W;                // [N][K] bf16
    const float* bias;              // [N] or null
    const float* R; int ldr;        // fp32 residual [M][ldr] or null (may alias C)
    float* C; int ldc;              // fp32 output or null
    bf16_t* Cb; int ldcb;           // bf16 output or null
    int M, N, K, act;
    int r_mod;                      // > 0: residual row = m % r_mod (a per-sample table broadcast over the batch)
    RowMap cmap;                    // output row of logical row m
    int xcd_swizzle;                // 1: tiles handed out so that one XCD works on consecutive tiles (see the kernel)
    // optional, honoured by the persistent 256 x 256 kernel only (gemm256.hpp; launch_gemm_dense reports how many rows it covered): the prefill's
    // fused q|k|v projection writes its K and V columns -- [kv_col0, 2 kv_col0) and [2 kv_col0, 3 kv_col0), heads of 64 -- straight into the KV-cache
    // planes instead of Cb: logical row m = sample m / kv_T, position m % kv_T -> plane[sample * kv_row_stride + (head * kv_max_seq + position) * 64 + d]
    bf16_t* kv_k; bf16_t* kv_v; size_t kv_row_stride; int kv_max_seq, kv_T, kv_col0;
    // launch_gemm_dense only: which rows this call computes -- 0 = all; 1 = rows [0, M - M % 256); 2 = the M % 256 rows behind them.  The kernels are chosen
    // as for the whole problem, so 1 and 2 together write exactly what 0 writes (the prefill runs its last rows as a chain of their own on a second stream)
    int part;
};

// MA_NO_ASAN: the LDS-DMA kernels stay uninstrumented in the sanitizer build (MA_DEBUG=asan): device ASan lowers a kernel's LDS to global
// memory, where a `global_load ... lds` has no destination (hipcc then fails on the M0 operand of this statement)
#define MA_NO_ASAN __attribute__((no_sanitize("address")))
MA_NO_ASAN __device__ __forceinline__ void gt_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");      // (wave-uniform by contract; the
                                                                                                                 //  explicit readfirstlane keeps it in an SGPR in unoptimised / sanitizer builds too)
}

template <int N> __device__ __forceinline__ void gt_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// BM x BN x BK block tile, 4 waves as 2 x 2.  LDS: NS stages x (BM + BN) rows x BK bf16.
// RAW: the K-loop's barrier is a bare s_barrier behind `s_waitcnt lgkmcnt(0)` instead of __syncthreads().  With an LDS-DMA in flight
// __syncthreads() carries `vmcnt(0)` (the DMA is a pending LDS write on the VM counter; MI355X guide section 5, "Pipelining across
// barriers"), which drains the ring at every K-tile -- the reason the deeper rings of round 2 measured slower, not faster.  The raw
// form keeps the younger tiles flying across the barrier; what it must still order is covered explicitly: this wave's own pieces of
// tile kt by the counted vmcnt, everyone's by the barrier behind it, and the stage that is re-filled was read one iteration earlier
// (its ds_reads retired by the lgkmcnt(0) in front of this barrier).
// WM x WN waves (default 2 x 2): each wave a (BM / WM) x (BN / WN) sub-tile.  The 4 x 2 form carries a 256 x 128 block tile on 8 waves: a
// third less LDS fill per FLOP than 128 x 128 (the loop is bound by the LDS-DMA issue and fill rate, not by the matrix cores).
// HT: the 16-bit format of both operands and of the 16-bit output (bf16_t | f16_t, common.hpp H16)
template <typename HT, int BM, int BN, int BK, int NS, bool RAW = false, int WM = 2, int WN = 2>
MA_NO_ASAN __global__ __launch_bounds__(WM * WN * 64) void gemm_tile_kernel(GemmTArgs g) {
    constexpr int NW = WM * WN;
    constexpr int CH = BK / 8;                        // 16-byte chunks per tile row
    auto swz = [](int r) { return CH == 8 ? (r & 7) : ((r >> 1) & 3); };
    constexpr int RPI = 64 / CH;                      // tile rows one DMA instruction covers
    constexpr int ROWB = BK * 2;                      // bytes per tile row
    constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;   // 16 x 16 MFMA tiles per wave along m / n
    extern __shared__ __attribute__((aligned(16))) char gt_smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w / WN, wn = w % WN;
    // Workgroups go to the 8 XCDs round-robin in dispatch order (x fastest), and every XCD has its own L2.  With the plain
    // (x = N tile, y = M tile) order and 8 N tiles, XCD i would own N-tile column i and stream EVERY activation tile through
    // its L2: 8 x the activation bytes over the fabric.  Remapped, XCD i owns the i-th eighth of the tile list in (m, n)
    // order: the N tiles of one M tile run back to back on ONE XCD, so an activation tile crosses the fabric once and the
    // weight matrix (<= 8 MB) is the operand that is shared through L2 / Infinity Cache.
    int tile_x = blockIdx.x, tile_y = blockIdx.y;
    if (g.xcd_swizzle) {
        const int NT = gridDim.x, tiles = NT * gridDim.y;
        const int L = blockIdx.y * NT + blockIdx.x;
        const int xcd = L & 7, i = L >> 3, lo = tiles >> 3, rem = tiles & 7;
        const int t = xcd * lo + min(xcd, rem) + i;
        tile_y = t / NT; tile_x = t - tile_y * NT;
    }
    const int bm = tile_y * BM, bn = tile_x * BN;
    const unsigned lds0 = (unsigned)(size_t)gt_smem;

    // ---- DMA addressing: instruction p of an operand covers tile rows p * RPI .. + RPI - 1; lane -> (row, slot) -----------
    const int drow = lane / CH, dslot = lane % CH;
    auto issue = [&](int kt, int stage) {
        const int k0 = kt * BK;
        const unsigned sbase = lds0 + (unsigned)stage * STAGE;
#pragma unroll
        for (int p = w; p < BM / RPI; p += NW) {
            const int r = p * RPI + drow;
            const int m = min(bm + r, g.M - 1);
            const bf16_t* src = g.A + (size_t)m * g.lda + k0 + ((dslot ^ swz(r)) * 8);
            gt_glds16(src, __builtin_amdgcn_readfirstlane(sbase + (unsigned)p * 1024u));
        }
#pragma unroll
        for (int p = w; p < BN / RPI; p += NW) {
            const int r = p * RPI + drow;
            const int n = min(bn + r, g.N - 1);
            const bf16_t* src = g.W + (size_t)n * g.K + k0 + ((dslot ^ swz(r)) * 8);
            gt_glds16(src, __builtin_amdgcn_readfirstlane(sbase + (unsigned)A_BYTES + (unsigned)p * 1024u));
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, kg = lane >> 4;          // fragment row, 8-element k group
    const int nk = g.K / BK;
    constexpr int PF = NS - 1;                         // K-tiles in flight ahead of the one being multiplied
    constexpr int IPW = (BM / RPI + BN / RPI) / NW;    // DMA instructions per wave and K-tile
    static_assert((BM / RPI) % NW == 0 && (BN / RPI) % NW == 0 && (PF - 1) * IPW < 64, "tile shape");
#pragma unroll
    for (int t = 0; t < PF; ++t) if (t < nk) issue(t, t);
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + PF <= nk) gt_wait_vm<(PF - 1) * IPW>();       // this wave's pieces of K-tile kt have landed (younger tiles still fly) ...
        else gt_wait_vm<0>();
        if constexpr (RAW) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        else __syncthreads();                                  // ... and everyone's; the stage read last iteration is free
        if (kt + PF < nk) issue(kt + PF, stage == 0 ? NS - 1 : stage - 1);
        const char* sa = gt_smem + stage * STAGE;
        const char* sw = sa + A_BYTES;
#pragma unroll
        for (int s = 0; s < BK / 32; ++s) {
            u32x4 wf[TN], xf[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int r = wn * (BN / WN) + i * 16 + fr;
                wf[i] = *reinterpret_cast<const u32x4*>(sw + r * ROWB + (((s * 4 + kg) ^ swz(r)) * 16));
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int r = wm * (BM / WM) + j * 16 + fr;
                xf[j] = *reinterpret_cast<const u32x4*>(sa + r * ROWB + (((s * 4 + kg) ^ swz(r)) * 16));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = H16<HT>::mfma16(wf[i], xf[j], acc[i][j]);
        }
        stage = stage + 1 == NS ? 0 : stage + 1;
    }

    // ---- epilogue: lane holds n = n0 .. n0 + 3 of row m ------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int n0 = bn + wn * (BN / WN) + i * 16 + kg * 4;
        if (n0 >= g.N) continue;
        const bool full = n0 + 3 < g.N;
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) {
            if (full) b4 = *reinterpret_cast<const f32x4*>(g.bias + n0);
            else { b4.x = g.bias[n0]; if (n0 + 1 < g.N) b4.y = g.bias[n0 + 1]; if (n0 + 2 < g.N) b4.z = g.bias[n0 + 2]; }
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = bm + wm * (BM / WM) + j * 16 + fr;
            if (m >= g.M) continue;
            const size_t mr = (size_t)(g.r_mod > 0 ? m % g.r_mod : m), mo = g.cmap(m);
            f32x4 v = acc[i][j];
            v.x = apply_act(v.x + b4.x, g.act); v.y = apply_act(v.y + b4.y, g.act);
            v.z = apply_act(v.z + b4.z, g.act); v.w = apply_act(v.w + b4.w, g.act);
            if (full) {
                if (g.R) { const f32x4 r4 = *reinterpret_cast<const f32x4*>(g.R + mr * g.ldr + n0); v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
                if (g.C) *reinterpret_cast<f32x4*>(g.C + mo * g.ldc + n0) = v;
                if (g.Cb) {
                    *reinterpret_cast<u32x2*>(g.Cb + mo * g.ldcb + n0) = pack4<HT>(v);
                }
            } else {
                const float vv[4] = {v.x, v.y, v.z, v.w};
                for (int r = 0; r < 4 && n0 + r < g.N; ++r) {
                    float t = vv[r];
                    if (g.R) t += g.R[mr * g.ldr + n0 + r];
                    if (g.C) g.C[mo * g.ldc + n0 + r] = t;
                    if (g.Cb) g.Cb[mo * g.ldcb + n0 + r] = H16<HT>::bits(t);
                }
            }
        }
    }
}

// variant knob (engine option gemm_variant; only libraries built with MA_EXPERIMENTAL=1 honour values other than the default 6; A/B in
// profiles/): 0 = K-tile 64, 2 stages | 1 = K-tile 32, 4 stages | 2 = K-tile 32, 5 stages |
// 3 = K-tile 64, 3 stages (one block per CU) | 4 = K-tile 64, 3 stages, raw barrier | 5 = K-tile 32, 4 stages, raw barrier | 6 = K-tile 64,
// 2 stages, raw barrier (default) | 7 = K-tile 32, 6 stages, raw barrier | 8 = the 128 x 64 tile for every shape (A/B of the tail rounds)
inline int& gemm_tile_variant() { static int v = 6; return v; }      // 6: +2 ... +13 % over 0 on the path's shapes (profiles/r03_ab_dense_attention_gemm_variants.txt)

template <typename HT, int BM, int BN, int BK, int NS, bool RAW = false, int WM = 2, int WN = 2>
inline hipError_t gt_launch(const GemmTArgs& g, hipStream_t s) {
    constexpr int LDS = NS * (BM + BN) * BK * 2;
    static bool attr = false;
    if (!attr) {
        hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tile_kernel<HT, BM, BN, BK, NS, RAW, WM, WN>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (r != hipSuccess) return r;
        attr = true;
    }
    hipLaunchKernelGGL((gemm_tile_kernel<HT, BM, BN, BK, NS, RAW, WM, WN>), dim3((g.N + BN - 1) / BN, (g.M + BM - 1) / BM), dim3(WM * WN * 64), LDS, s, g);
    return hipGetLastError();
}

// 16-byte DMA sources and vector epilogue accesses need: lda % 8 == 0, K % 32 == 0, ldc / ldr % 4 == 0, ldcb % 4 == 0
// choice_rows > 0: the tile shape is chosen as for a problem of that many rows (a stretch of a larger problem's rows computed by a call of its own gets the
// larger problem's kernel, hence its bits: launch_gemm_dense, GemmTArgs::part); the grid always comes from the real row count
template <typename HT>
inline hipError_t launch_gemm_tile(const GemmTArgs& g, hipStream_t s, int choice_rows = 0) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    const int Msel = choice_rows > 0 ? choice_rows : g.M;
    if (g.K % 32 != 0 || g.lda % 8 != 0 || (g.C && g.ldc % 4) || (g.R && g.ldr % 4) || (g.Cb && g.ldcb % 4) || (!g.C && !g.Cb)) return hipErrorInvalidValue;
    const long tiles128 = (long)((g.N + 127) / 128) * ((Msel + 127) / 128);
    const int v = gemm_tile_variant();
#ifdef MA_EXPERIMENTAL
    // the A/B variants of rounds 2-3 (profiles/r02_ab_gemm_tile_stages.txt, r03_ab_gemm_tile_occupancy_and_tail.txt): evidence, not product
    if (v == 8 && g.K % 64 == 0 && g.M > 64 && g.N > 32) return gt_launch<HT, 128, 64, 64, 2, true>(g, s);      // the half tile everywhere
    if (v == 12 && g.K % 64 == 0 && g.M > 128 && g.N > 64) return gt_launch<HT, 256, 128, 64, 2, true, 4, 2>(g, s);    // 256 x 128 tile, 8 waves, 96 KB
    if (v == 13 && g.K % 64 == 0 && g.M > 128 && g.N > 64) return gt_launch<HT, 256, 128, 64, 3, true, 4, 2>(g, s);    // ... three stages, 144 KB
    if (v == 14 && g.K % 64 == 0 && g.M > 128 && g.N > 64) return gt_launch<HT, 256, 128, 32, 4, true, 4, 2>(g, s);    // ... K-tile 32, four stages, 96 KB
    if (v == 9 && g.K % 64 == 0 && g.M > 64 && g.N > 64) return gt_launch<HT, 128, 128, 32, 3, true>(g, s);     // 48 KB of LDS: three blocks per CU
    if (v == 10 && g.K % 64 == 0 && g.M > 64 && g.N > 32) return gt_launch<HT, 128, 64, 64, 3, true>(g, s);     // half tile, two K-tiles in flight
    if (v == 11 && g.K % 64 == 0 && g.M > 64 && g.N > 64) return gt_launch<HT, 128, 128, 32, 2, true>(g, s);    // 32 KB of LDS: four-five blocks per CU
    if (g.K % 64 == 0 && g.M > 64 && g.N > 64 && tiles128 >= 160) {
        if (v == 1) return gt_launch<HT, 128, 128, 32, 4>(g, s);
        if (v == 2) return gt_launch<HT, 128, 128, 32, 5>(g, s);
        if (v == 3) return gt_launch<HT, 128, 128, 64, 3>(g, s);
        if (v == 4) return gt_launch<HT, 128, 128, 64, 3, true>(g, s);
        if (v == 5) return gt_launch<HT, 128, 128, 32, 4, true>(g, s);
        if (v == 7) return gt_launch<HT, 128, 128, 32, 6, true>(g, s);
        if (v == 0) return gt_launch<HT, 128, 128, 64, 2>(g, s);
    }
    if (g.K % 64 == 0 && g.M > 64 && g.N > 32 && tiles128 < 160) {
        if (v == 1 || v == 2) return gt_launch<HT, 128, 64, 32, 5>(g, s);
        if (v == 3) return gt_launch<HT, 128, 64, 64, 3>(g, s);
        if (v == 0) return gt_launch<HT, 128, 64, 64, 2>(g, s);
    }
#endif
    (void)v;
    // Problems with many tiles (the detokenizer's M = B x 1057, wide N): the 256 x 128 tile on 8 waves with three LDS stages (144 KB, one block
    // per CU) -- a third less LDS fill per FLOP.  Isolated: 569 vs 504 TFLOP/s (67648 x 3072 x 768), 602 vs 514 (262144 x 1536 x 768), 983 vs
    // 720-868 on 8192^3 (profiles/r03_ab_gemm_tile_occupancy_and_tail.txt); inside the pipeline the detokenizer gains 2.6 %, the encoder's
    // 262144-row GEMM loses (profiles/r03_ab_dense_gemm_selection_in_pipeline.txt), so it is kept to 2048 .. 16384 tiles of M <= 131072.
    if (g.K % 64 == 0 && Msel > 128 && Msel <= 131072 && g.N > 64 && tiles128 >= 2048) return gt_launch<HT, 256, 128, 64, 3, true, 4, 2>(g, s);
    if (g.K % 64 == 0 && Msel > 64 && g.N > 64 && tiles128 >= 160) return gt_launch<HT, 128, 128, 64, 2, true>(g, s);
    // the 128 x 128 grid would leave a third of the CUs idle: halve the tile along N
    if (g.K % 64 == 0 && Msel > 64 && g.N > 32) return gt_launch<HT, 128, 64, 64, 2, true>(g, s);
    return gt_launch<HT, 64, 64, 32, 2>(g, s);
}

// fp32 -> bf16 rows (kernel-level entry point ma_op_gemm with a bf16 weight and an fp32 activation matrix; small utility elsewhere)
__global__ void f32_to_bf16_rows_kernel(const float* __restrict__ src, int lds, bf16_t* __restrict__ dst, int ldd, int rows, int cols) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const int i = idx / cols, n = idx - i * cols;
    dst[(size_t)i * ldd + n] = f2bf(src[(size_t)i * lds + n]);
}

}  // namespace ma
