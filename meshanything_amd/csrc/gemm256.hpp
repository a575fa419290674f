// Dense GEMM of the 16-bit policies for the large dense-phase shapes: C[M,N] = act(A[M,K] . W[N,K]^T + bias) + R on a 256 x 256 x 64 block
// tile, 8 waves, with the staged "8-phase" K-loop of the MI355X guide (section 5, "The 256^2 8-phase template"; T2-T5).
//
// Same reference call sites, operands, epilogue and accumulation order (k ascending in 32-wide MFMA steps) as gemm_tile.hpp, whose
// 128 x 128 tile stays the kernel of the small problems and of the remainder tiles (see launch_gemm256).  Why a second kernel: the 128-row
// tile needs 62 B/clk/CU of LDS fill per MFMA-bound cycle and stalls on its one barrier per K-tile (15-18 % of the
// bf16 MFMA peak in the pipeline); a 256 x 256 tile halves the fill per FLOP, and the phase structure below keeps LDS reads, LDS-DMA and
// MFMAs of the two wave groups overlapped.
//
// Geometry: 8 waves as 2 (m) x 4 (n); a wave owns 128 x 64 outputs = 2 x 2 quadrants of 64 x 32 = 32 accumulator tiles of 16 x 16 (128
// accumulator registers).  A K-tile (64 deep) is FOUR 16-KB pieces in LDS, each 128 rows x 128 B:
//     X_s (s = 0, 1): activation rows wm * 128 + s * 64 + [0, 64) of both wave rows      W_s: weight rows wn * 64 + s * 32 + [0, 32) of all four wave columns
// i.e. piece s holds exactly what every wave reads for its quadrants (s, .) / (., s).  Two K-tile buffers = 128 KB, one block per CU.
// Pieces arrive by LDS-DMA (global_load_lds_dwordx4: 2 instructions per wave and piece); the bank-conflict swizzle sits on the SOURCE address
// (chunk c of piece row r lands in slot c ^ (r & 7), read back from there with ds_read_b128: guide rule 21).
//
// K-loop, per K-tile T four phases, each = { ds_read the phase's register sub-tile | issue ONE piece of a later K-tile | s_barrier |
// lgkmcnt(0) | 16 MFMAs (one quadrant x K = 64) | s_barrier }:
//     P0: read X_0 (8 x b128), W_0 (4)   stage X_1(T+1)   quadrant (0,0)
//     P1: read W_1 (4)                    stage W_0(T+1)   quadrant (0,1)
//     P2: read X_1 (8)                    stage X_0(T+2)   quadrant (1,1)
//     P3: read W_0 (4)                    stage W_1(T+2)   quadrant (1,0)      + the K-tile's only vmcnt
//   * WAR: a piece is re-staged TWO phases after its last ds_read (X_0: P0 -> P2, W_1: P1 -> P3, X_1: P2 -> next P0, W_0: P3 -> next P1),
//     so the reads were retired by an lgkmcnt(0) that every wave has passed (two barriers in between, also for the staggered group);
//   * RAW: DMA completes in issue order; at P3 the youngest piece the next K-tile needs (W_0(T+1), issued at P1) is followed by two
//     younger pieces (4 instructions): `s_waitcnt vmcnt(4)` before P3's first barrier, reads from the next phase on -- never a vmcnt(0) in
//     the steady state, loads stay in flight across the barriers (raw s_barrier: __syncthreads() would drain them);
//   * the two wave groups (waves 0-3 / 4-7 = wave rows 0 / 1, which share the four SIMDs pairwise) run staggered by ONE barrier: while
//     one group issues MFMAs the other reads LDS and issues DMA -- the role split that s_setprio(1) around the MFMA cluster arbitrates.
// Epilogue as gemm_tile.hpp (bias, ReLU / GELU, fp32 residual, fp32 and / or 16-bit output through the row map).
// Algorithmic FLOPs: 2 M N K.  Tiles are handed out XCD-aware (consecutive tiles of the (m, n) list on one XCD).
#pragma once
#include "common.hpp"
#include "gemm_tile.hpp"

namespace ma {

constexpr int G256_PIECE = 128 * 128;              // bytes of one piece (128 rows x 64 elements x 2 B)
constexpr int G256_LDS = 2 * 4 * G256_PIECE;       // two K-tile buffers x {X_0, X_1, W_0, W_1}

template <int N> __device__ __forceinline__ void g256_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// tile_lo: first tile (of the (m, n) tile list, n fastest) this launch computes; it computes gridDim.x of them
// ACT: the activation is a template parameter here (128 accumulators x an inlined erf would otherwise sit in every instantiation)
template <typename HT, int ACT>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmTArgs g, int tile_lo, int ntx) {
    extern __shared__ __attribute__((aligned(16))) char g256_smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 2, wn = w & 3;
    // XCD-aware hand-out: workgroups go to the XCDs round-robin; XCD i works on the i-th eighth of this launch's tile range
    int t = blockIdx.x;
    {
        const int tiles = gridDim.x, xcd = t & 7, i = t >> 3, lo = tiles >> 3, rem = tiles & 7;
        t = xcd * lo + min(xcd, rem) + i;
    }
    t += tile_lo;
    const int tile_y = t / ntx, tile_x = t - tile_y * ntx;
    const int bm = tile_y * 256, bn = tile_x * 256;
    const unsigned lds0 = (unsigned)(size_t)g256_smem;
    const int nk = g.K >> 6;

    // ---- LDS-DMA: instruction p (0 .. 15) of a piece covers piece rows 8 p .. 8 p + 7; this wave issues p = w and w + 8 ---------------------
    const int drow = lane >> 3, dslot = lane & 7;
    const bf16_t* xsrc[2][2]; const bf16_t* wsrc[2][2];             // [piece s][instruction i]: source of this lane, K-tile 0
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pr = (w + 8 * i) * 8 + drow;                   // piece row
            const int sw = (dslot ^ (pr & 7)) * 8;                   // swizzled 16-byte chunk of the source row
            const int m = min(bm + (pr >> 6) * 128 + s * 64 + (pr & 63), g.M - 1);
            const int n = min(bn + (pr >> 5) * 64 + s * 32 + (pr & 31), g.N - 1);
            xsrc[s][i] = g.A + (size_t)m * g.lda + sw;
            wsrc[s][i] = g.W + (size_t)n * g.K + sw;
        }
    auto stage_x = [&](int s, int kt) {
        const unsigned dst = lds0 + (unsigned)(kt & 1) * (4u * G256_PIECE) + (unsigned)s * G256_PIECE + (unsigned)w * 1024u;
        gt_glds16(xsrc[s][0] + (size_t)kt * 64, __builtin_amdgcn_readfirstlane(dst));
        gt_glds16(xsrc[s][1] + (size_t)kt * 64, __builtin_amdgcn_readfirstlane(dst + 8192u));
    };
    auto stage_w = [&](int s, int kt) {
        const unsigned dst = lds0 + (unsigned)(kt & 1) * (4u * G256_PIECE) + (unsigned)(2 + s) * G256_PIECE + (unsigned)w * 1024u;
        gt_glds16(wsrc[s][0] + (size_t)kt * 64, __builtin_amdgcn_readfirstlane(dst));
        gt_glds16(wsrc[s][1] + (size_t)kt * 64, __builtin_amdgcn_readfirstlane(dst + 8192u));
    };

    // ---- fragment addressing: lane (fr = row of a 16-row MFMA tile, kg = its 8-element k group) ----------------------------------------------
    const int fr = lane & 15, kg = lane >> 4;
    // byte offsets inside a piece: X sub-tile row wm * 64 + j * 16 + fr, W sub-tile row wn * 32 + i * 16 + fr; k step ks -> chunk ks * 4 + kg
    int xoff[4][2], woff[2][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { const int pr = wm * 64 + j * 16 + fr; xoff[j][ks] = pr * 128 + (((ks * 4 + kg) ^ (pr & 7)) * 16); }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { const int pr = wn * 32 + i * 16 + fr; woff[i][ks] = pr * 128 + (((ks * 4 + kg) ^ (pr & 7)) * 16); }

    f32x4 acc[2][2][2][4];                                           // [x sub][w sub][n tile i][m tile j]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 xf[4][2], wf[2][2];
    auto read_x = [&](int s, int kt) {
        const char* base = g256_smem + (kt & 1) * (4 * G256_PIECE) + s * G256_PIECE;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xf[j][ks] = *reinterpret_cast<const u32x4*>(base + xoff[j][ks]);
    };
    auto read_w = [&](int s, int kt) {
        const char* base = g256_smem + (kt & 1) * (4 * G256_PIECE) + (2 + s) * G256_PIECE;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wf[i][ks] = *reinterpret_cast<const u32x4*>(base + woff[i][ks]);
    };
    // the MFMA cluster of one phase: quadrant (a, b) x K = 64.  Between two raw barriers; the reads it consumes are retired first.
#define G256_COMPUTE(a, b)                                                                                       \
    do {                                                                                                         \
        __builtin_amdgcn_s_barrier();                                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                                           \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                         \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                        \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                    \
                    acc[a][b][i][j] = H16<HT>::mfma16(wf[i][ks], xf[j][ks], acc[a][b][i][j]);                    \
        __builtin_amdgcn_s_setprio(0);                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        __builtin_amdgcn_s_barrier();                                                                            \
    } while (0)

    // ---- prologue: K-tile 0 complete, the first two pieces of K-tile 1 ---------------------------------------------------------------------
    stage_x(0, 0); stage_w(0, 0); stage_w(1, 0); stage_x(1, 0);
    if (nk > 1) { stage_x(0, 1); stage_w(1, 1); g256_wait_vm<4>(); } else g256_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();                       // the second wave group runs one barrier behind the first

    for (int kt = 0; kt < nk; ++kt) {
        // P0
        read_w(0, kt); __builtin_amdgcn_sched_barrier(0); read_x(0, kt);
        if (kt + 1 < nk) stage_x(1, kt + 1);
        G256_COMPUTE(0, 0);
        // P1
        read_w(1, kt);
        if (kt + 1 < nk) stage_w(0, kt + 1);
        G256_COMPUTE(0, 1);
        // P2
        read_x(1, kt);
        if (kt + 2 < nk) stage_x(0, kt + 2);
        G256_COMPUTE(1, 1);
        // P3
        read_w(0, kt);
        if (kt + 2 < nk) { stage_w(1, kt + 2); g256_wait_vm<4>(); }  // W_0(kt + 1) and everything older have landed; two pieces stay in flight
        else g256_wait_vm<0>();
        G256_COMPUTE(1, 0);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();                       // (the barrier the first group is ahead by)
#undef G256_COMPUTE

    // ---- epilogue: lane holds n = n0 .. n0 + 3 of row m (gemm_tile.hpp's) --------------------------------------------------------------------
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n0 = bn + wn * 64 + b * 32 + i * 16 + kg * 4;
            if (n0 >= g.N) continue;
            const bool full = n0 + 3 < g.N;
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (g.bias) {
                if (full) b4 = *reinterpret_cast<const f32x4*>(g.bias + n0);
                else { b4.x = g.bias[n0]; if (n0 + 1 < g.N) b4.y = g.bias[n0 + 1]; if (n0 + 2 < g.N) b4.z = g.bias[n0 + 2]; }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = bm + wm * 128 + a * 64 + j * 16 + fr;
                    if (m >= g.M) continue;
                    const size_t mr = (size_t)(g.r_mod > 0 ? m % g.r_mod : m), mo = g.cmap(m);
                    f32x4 v = acc[a][b][i][j];
                    v.x = apply_act(v.x + b4.x, ACT); v.y = apply_act(v.y + b4.y, ACT);
                    v.z = apply_act(v.z + b4.z, ACT); v.w = apply_act(v.w + b4.w, ACT);
                    if (full) {
                        if (g.R) { const f32x4 r4 = *reinterpret_cast<const f32x4*>(g.R + mr * g.ldr + n0); v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
                        if (g.C) *reinterpret_cast<f32x4*>(g.C + mo * g.ldc + n0) = v;
                        if (g.Cb) *reinterpret_cast<u32x2*>(g.Cb + mo * g.ldcb + n0) = pack4<HT>(v);
                    } else {
                        const float vv[4] = {v.x, v.y, v.z, v.w};
                        for (int r = 0; r < 4 && n0 + r < g.N; ++r) {
                            float tt = vv[r];
                            if (g.R) tt += g.R[mr * g.ldr + n0 + r];
                            if (g.C) g.C[mo * g.ldc + n0 + r] = tt;
                            if (g.Cb) g.Cb[mo * g.ldcb + n0 + r] = H16<HT>::bits(tt);
                        }
                    }
                }
        }
}

// Whole rounds of the chip (one block per CU) run on the 256 x 256 tile; what is left of the tile list -- M = B x 257 gives 65 x N / 256
// tiles, a little more than a multiple of the CU count -- is covered by the 128 x 128 tile (four per big tile, lin_* mapping of
// gemm_tile.hpp), so the tail costs a quarter-tile's time on a few CUs instead of a whole extra round on all of them.
template <typename HT, int ACT>
inline hipError_t g256_launch(const GemmTArgs& g, int tiles, int ntx, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<HT, ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS);
        if (r != hipSuccess) return r;
        attr = true;
    }
    hipLaunchKernelGGL((gemm256_kernel<HT, ACT>), dim3(tiles), dim3(512), G256_LDS, s, g, 0, ntx);
    return hipGetLastError();
}

template <typename HT>
inline hipError_t launch_gemm256(const GemmTArgs& g, int n_cus, hipStream_t s) {
    const int ntx = (g.N + 255) / 256, nty = (g.M + 255) / 256, tiles = ntx * nty;
    const int full = n_cus > 0 ? tiles / n_cus * n_cus : tiles;
    // a remainder of more than 3/4 of a round runs as one more round of big tiles (the small tile would take longer than that round)
    const int big = (tiles - full) * 4 > 3 * n_cus ? tiles : full;
    if (big > 0) {
        hipError_t r = g.act == ACT_RELU ? g256_launch<HT, ACT_RELU>(g, big, ntx, s) : g.act == ACT_GELU ? g256_launch<HT, ACT_GELU>(g, big, ntx, s) : g256_launch<HT, ACT_NONE>(g, big, ntx, s);
        if (r != hipSuccess) return r;
    }
    if (big < tiles) {
        GemmTArgs t = g;
        t.lin_t0 = big; t.lin_ntx = ntx; t.lin_tiles = tiles - big;
        return gt_launch_lin<HT>(t, s);
    }
    return hipSuccess;
}

// engine option "gemm256" (default 1): A/B switch between this kernel and the 128-row tiles for the shapes it covers
inline int& gemm256_enabled() { static int v = 1; return v; }

// the dense GEMM of the 16-bit policies: the 256 x 256 kernel where the problem fills at least one round of the chip with its tiles,
// the tiles of gemm_tile.hpp otherwise (small M of batch-1 runs, N = 64 / 128 projections, K not a multiple of 64)
template <typename HT>
inline hipError_t launch_gemm_dense(const GemmTArgs& g, int n_cus, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if (g.K % 32 != 0 || g.lda % 8 != 0 || (g.C && g.ldc % 4) || (g.R && g.ldr % 4) || (g.Cb && g.ldcb % 4) || (!g.C && !g.Cb)) return hipErrorInvalidValue;
    const long tiles = (long)((g.N + 255) / 256) * ((g.M + 255) / 256);
    if (gemm256_enabled() && n_cus > 0 && g.K % 64 == 0 && g.K >= 128 && g.N >= 256 && g.M >= 256 && tiles >= n_cus && tiles < (1L << 24))
        return launch_gemm256<HT>(g, n_cus, s);
    return launch_gemm_tile<HT>(g, s);
}

}  // namespace ma
