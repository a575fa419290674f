// Dense GEMM of the 16-bit policies for the large dense-phase shapes: C[M,N] = act(A[M,K] . W[N,K]^T + bias) + R on a 256 x 256 x 64 block
// tile, 8 waves, with the staged "8-phase" K-loop of the MI355X guide (section 5, "The 256^2 8-phase template"; T2-T5).
//
// Same reference call sites, operands, epilogue and accumulation order (k ascending in 32-wide MFMA steps) as gemm_tile.hpp, whose
// 128 x 128 tile stays the kernel of the small and oddly sized problems (see launch_gemm_dense).  Why a second kernel: the 128-row
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
// Algorithmic FLOPs: 2 M N K.  Tiles are handed out XCD-aware (an 8 x 4 patch of tiles per XCD at a time: see the kernel).
#pragma once
#include "common.hpp"
#include "gemm_decode.hpp"
#include "gemm_tile.hpp"

namespace ma {

constexpr int G256_PIECE = 128 * 128;              // bytes of one piece (128 rows x 64 elements x 2 B)
constexpr int G256_LDS = 2 * 4 * G256_PIECE;       // two K-tile buffers x {X_0, X_1, W_0, W_1}

template <int N> __device__ __forceinline__ void g256_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// The launch computes the nty x ntx tiles of rows [0, 256 nty).  Tile order: column panels of four n-tiles, m fastest inside a panel, so that
// the 32 tiles an XCD works on at a time form an 8 (m) x 4 (n) patch -- 12 operand tiles through its L2 instead of 2 + 16 (or 1 + 32) with
// the n-fastest list.
// ACT: the activation is a template parameter here (128 accumulators x an inlined erf would otherwise sit in every instantiation)
// ABL (scripts/ubench_gemm256.hip only; 0 in the library): ablation bits -- 1 no LDS-DMA in the loop, 2 no ds_reads, 4 no MFMAs, 8 no stores
// EV (scripts/ubench_gemm256_epi.hip only; G256_EV in the library): epilogue form bits -- 2 the straight-line form of whole tiles (all LDS reads /
// residual loads of a pass issued first, then the stores); (bit 1, the hardware 16-bit conversion, became common.hpp's f2bf / pack2)
constexpr int G256_EV = 2;
// LNF: the LayerNorm that follows an N = hidden GEMM (post-LN layers: h = LN(h + W a + b), [3p] OPTDecoderLayer; shape_opt.py:403-410) finished INSIDE the
// GEMM's epilogue instead of by a row kernel that re-reads the sum (ln_rows2_kernel: 168 MB per call at 64 samples, 30 us of a 600-us layer, twice per
// layer).  A row spans the ntx = N / 256 tiles of its tile row, i.e. ntx workgroups: they exchange per-row moments through 8-byte {epoch, value} granules
// (common.hpp ps_publish; MI355X guide, Guideline 16 R2) in two rounds -- row sums -> mean, then sums of squared deviations -> variance: the exact
// two-pass form of ln_rows2_kernel -- and each normalises its own 256 columns: outputs h (fp32, in place over the residual it read) and its 16-bit copy.
// Order of every sum is fixed (4 columns of a lane, the DPP tree over 16 lanes, the four waves of a tile, the tiles in ascending order), so all tiles
// of a row compute the same mean / rstd bit for bit and the result does not depend on timing.  The workgroups of a tile row are consecutive in the
// tile list; a sweep is bounded like every in-launch exchange of this engine (20 ms, then the error word: the generation is re-run without the fused
// forms, engine.hip generate_batch).  Whole tiles only, K not split.
struct G256Ln {
    const float* gamma; const float* beta; float eps;
    u64* gran;                     // [2][nty * ntx][256]: row sums | sums of squared deviations, one granule per (tile, row)
    unsigned* err;                 // the engine's error words (common.hpp xchg_raise)
    unsigned epoch;                // unique per launch, never 0 (the buffer starts zeroed)
};
constexpr unsigned G256_ERR_LN = 2048;
constexpr int G256_LN_LDS = 8192;  // LDS behind the operand buffers: [4 waves][256] partial sums, [256] mean, [256] rstd
template <typename HT, int ACT, int ABL = 0, int EV = G256_EV, bool LNF = false>
MA_NO_ASAN __global__ __launch_bounds__(512) void gemm256_kernel(GemmTArgs g, int nty, int ntx, unsigned long long* trace = nullptr, int ks = 1, long part_stride = 0, G256Ln ln = G256Ln{}) {
    extern __shared__ __attribute__((aligned(16))) char g256_smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 2, wn = w & 3;
    // XCD-aware hand-out: workgroups go to the XCDs round-robin; XCD i works on the i-th eighth of this launch's tile range
    // ks > 1 (fp32 output only): split along K -- the grid is ks x the tile list, launch part `split` accumulates k in [split K / ks, (split + 1) K / ks)
    // into the fp32 buffer g.C + split * part_stride (bias and residual go into part 0; the consumer -- ln_rows2_kernel -- adds the parts up).  For
    // the N = hidden GEMMs of small batches, whose 256 x 256 tiles fill a fraction of the chip (fc2 at 16 samples: 64 tiles x 64 K-tiles)
    int t = blockIdx.x, split = 0;
    const int tiles = gridDim.x / ks;
    if (ks > 1) { split = t / tiles; t -= split * tiles; }
    {
        const int xcd = t & 7, i = t >> 3, lo = tiles >> 3, rem = tiles & 7;
        t = xcd * lo + min(xcd, rem) + i;
    }
    const int kbeg = split * (g.K / ks);
    const float* const biasp = split == 0 ? g.bias : nullptr;
    const float* const Rp = split == 0 ? g.R : nullptr;
    float* const Cp = g.C ? g.C + (size_t)split * part_stride : nullptr;
    int tile_y, tile_x;
    {
        const int fullp = ntx >> 2, tailw = ntx & 3, cut = fullp * 4 * nty;
        if (t < cut) { const int p = t / (4 * nty), r = t - p * 4 * nty; tile_y = r >> 2; tile_x = p * 4 + (r & 3); }
        else { const int r = t - cut; tile_y = r / tailw; tile_x = fullp * 4 + r - tile_y * tailw; }
    }
    const int bm = tile_y * 256, bn = tile_x * 256;
    const unsigned lds0 = (unsigned)(size_t)g256_smem;
    const int nk = (g.K / ks) >> 6;

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
            xsrc[s][i] = g.A + (size_t)m * g.lda + sw + kbeg;
            wsrc[s][i] = g.W + (size_t)n * g.K + sw + kbeg;
        }
    auto stage_x = [&](int s, int kt) {
        if constexpr (ABL & 1) { if (kt > 0) return; }
        const unsigned dst = lds0 + (unsigned)(kt & 1) * (4u * G256_PIECE) + (unsigned)s * G256_PIECE + (unsigned)w * 1024u;
        gt_glds16(xsrc[s][0] + (size_t)kt * 64, __builtin_amdgcn_readfirstlane(dst));
        gt_glds16(xsrc[s][1] + (size_t)kt * 64, __builtin_amdgcn_readfirstlane(dst + 8192u));
    };
    auto stage_w = [&](int s, int kt) {
        if constexpr (ABL & 1) { if (kt > 0) return; }
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
        if constexpr (ABL & 2) { if (kt > 0) return; }
        const char* base = g256_smem + (kt & 1) * (4 * G256_PIECE) + s * G256_PIECE;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xf[j][ks] = *reinterpret_cast<const u32x4*>(base + xoff[j][ks]);
    };
    auto read_w = [&](int s, int kt) {
        if constexpr (ABL & 2) { if (kt > 0) return; }
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
        if (!(ABL & 4) || kt == 0)                                                                               \
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
    if (trace && tid == 0) trace[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memrealtime();
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
    if (trace && tid == 0) trace[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
#undef G256_COMPUTE

    // ---- epilogue ----------------------------------------------------------------------------------------------------------------------------
    // A lane holds n = n0 .. n0 + 3 of 32 rows: stored from registers that is 32 narrow stores per lane into 64-byte row segments, and with ONE
    // block per CU nothing overlaps them -- measured 11.7 us per tile, as long as the whole K-loop at K = 512 (store-issue bound, guide T21).
    // Instead every wave parks its 128 x 64 sub-tile in its own 16 KB of the (now free) LDS -- bias and activation applied, 16-byte chunks
    // XOR-swizzled by the row -- and streams it out as whole row segments with 16-byte stores: 128 B (16-bit output: 8 rows per instruction) or
    // 256 B (fp32 output, 64 rows at a time: 4 rows per instruction); the fp32 residual is read the same way.  Only this wave touches its
    // region, so the hand-over is an lgkmcnt(0), not a barrier (every wave has passed the K-loop's last barrier: no one reads operands any more).
    char* patch = g256_smem + w * 16384;
    f32x4 bias4[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n0 = bn + wn * 64 + b * 32 + i * 16 + kg * 4;
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (biasp) {
                if (n0 + 3 < g.N) b4 = *reinterpret_cast<const f32x4*>(biasp + n0);
                else { if (n0 < g.N) b4.x = biasp[n0]; if (n0 + 1 < g.N) b4.y = biasp[n0 + 1]; if (n0 + 2 < g.N) b4.z = biasp[n0 + 2]; }
            }
            bias4[b][i] = b4;
        }
    const int nw0 = bn + wn * 64;                                    // first column of this wave's sub-tile
    // whole tiles with plain row addressing take the straight-line form: every LDS read (and residual load) of a pass is issued before the first
    // store, so a pass costs one LDS / memory round trip instead of one per store (the guarded loop below waits for each chunk in turn)
    const bool whole = (EV & 2) && bm + 256 <= g.M && bn + 256 <= g.N && g.cmap.grp == 0 && g.r_mod == 0;
    if (Cp == nullptr) {
        // 16-bit output only: patch rows of 128 B (64 elements), chunk = 8 elements
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        f32x4 v = acc[a][b][i][j];
                        const f32x4 b4 = bias4[b][i];
                        v.x = apply_act(v.x + b4.x, ACT); v.y = apply_act(v.y + b4.y, ACT); v.z = apply_act(v.z + b4.z, ACT); v.w = apply_act(v.w + b4.w, ACT);
                        const int rr = a * 64 + j * 16 + fr, chunk = b * 4 + i * 2 + (kg >> 1);
                        *reinterpret_cast<u32x2*>(patch + rr * 128 + ((chunk ^ (rr & 7)) * 16) + (kg & 1) * 8) = pack4<HT>(v);
                    }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int r8 = lane >> 3, c = lane & 7, ncol = nw0 + c * 8;
        if constexpr (!(ABL & 8)) {
        if (whole) {
            u32x4 q[16];
#pragma unroll
            for (int it = 0; it < 16; ++it) { const int row = it * 8 + r8; q[it] = *reinterpret_cast<const u32x4*>(patch + row * 128 + ((c ^ (row & 7)) * 16)); }
            bf16_t* dst = g.Cb + (size_t)(bm + wm * 128 + r8) * g.ldcb + ncol;
#pragma unroll
            for (int it = 0; it < 16; ++it) *reinterpret_cast<u32x4*>(dst + (size_t)(it * 8) * g.ldcb) = q[it];
        } else {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = it * 8 + r8, m = bm + wm * 128 + row;
            if (m >= g.M || ncol >= g.N) continue;
            const u32x4 q = *reinterpret_cast<const u32x4*>(patch + row * 128 + ((c ^ (row & 7)) * 16));
            bf16_t* dst = g.Cb + g.cmap(m) * g.ldcb + ncol;
            if (ncol + 7 < g.N) *reinterpret_cast<u32x4*>(dst) = q;
            else {                                                   // ragged right edge: element stores straight from the registers (no address of q taken)
                const unsigned qw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int r = 0; r < 8; ++r) if (ncol + r < g.N) dst[r] = (bf16_t)(qw[r >> 1] >> ((r & 1) * 16));
            }
        }
        }
        }
    } else {
        // fp32 output (+ residual, + optional 16-bit copy): 64 rows at a time, patch rows of 256 B (64 floats), chunk = 4 floats
        const int r4 = lane >> 4, c = lane & 15, ncol = nw0 + c * 4;
        // whole tiles: four passes of 32 rows (8 chunks per lane); the residual rows of pass p + 1 are requested before the stores of pass p go out
        f32x4 rv[8];
        auto load_res = [&](int pass) {
            const float* rp = Rp + (size_t)(bm + wm * 128 + pass * 32 + r4) * g.ldr + ncol;
#pragma unroll
            for (int it = 0; it < 8; ++it) rv[it] = *reinterpret_cast<const f32x4*>(rp + (size_t)(it * 4) * g.ldr);
        };
        if (whole && Rp) load_res(0);
        f32x4 yv[LNF ? 4 : 1][8];                                     // LNF: the wave's 128 x 64 sums stay in registers until the row moments are known
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        f32x4 v = acc[a][b][i][j];
                        const f32x4 b4 = bias4[b][i];
                        v.x = apply_act(v.x + b4.x, ACT); v.y = apply_act(v.y + b4.y, ACT); v.z = apply_act(v.z + b4.z, ACT); v.w = apply_act(v.w + b4.w, ACT);
                        const int rr = j * 16 + fr, chunk = b * 8 + i * 4 + kg;
                        *reinterpret_cast<f32x4*>(patch + rr * 256 + ((chunk ^ (rr & 15)) * 16)) = v;
                    }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (!(ABL & 8)) {
            if (whole) {
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    const int pass = a * 2 + hb;
                    f32x4 v[8];
#pragma unroll
                    for (int it = 0; it < 8; ++it) { const int row = hb * 32 + it * 4 + r4; v[it] = *reinterpret_cast<const f32x4*>(patch + row * 256 + ((c ^ (row & 15)) * 16)); }
                    if (Rp) {
#pragma unroll
                        for (int it = 0; it < 8; ++it) { v[it].x += rv[it].x; v[it].y += rv[it].y; v[it].z += rv[it].z; v[it].w += rv[it].w; }
                        if (pass < 3) load_res(pass + 1);
                    }
                    if constexpr (LNF) {
#pragma unroll
                        for (int it = 0; it < 8; ++it) yv[pass][it] = v[it];
                    } else {
                    const size_t m0 = (size_t)(bm + wm * 128 + pass * 32 + r4);
                    float* cp = Cp + m0 * g.ldc + ncol;
#pragma unroll
                    for (int it = 0; it < 8; ++it) *reinterpret_cast<f32x4*>(cp + (size_t)(it * 4) * g.ldc) = v[it];
                    if (g.Cb) {
                        bf16_t* cb = g.Cb + m0 * g.ldcb + ncol;
#pragma unroll
                        for (int it = 0; it < 8; ++it) *reinterpret_cast<u32x2*>(cb + (size_t)(it * 4) * g.ldcb) = pack4<HT>(v[it]);
                    }
                    }
                }
            } else {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = it * 4 + r4, m = bm + wm * 128 + a * 64 + row;
                if (m >= g.M || ncol >= g.N) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(patch + row * 256 + ((c ^ (row & 15)) * 16));
                const size_t mr = (size_t)(g.r_mod > 0 ? m % g.r_mod : m), mo = g.cmap(m);
                if (ncol + 3 < g.N) {
                    if (Rp) { const f32x4 r4v = *reinterpret_cast<const f32x4*>(Rp + mr * g.ldr + ncol); v.x += r4v.x; v.y += r4v.y; v.z += r4v.z; v.w += r4v.w; }
                    *reinterpret_cast<f32x4*>(Cp + mo * g.ldc + ncol) = v;
                    if (g.Cb) *reinterpret_cast<u32x2*>(g.Cb + mo * g.ldcb + ncol) = pack4<HT>(v);
                } else {
                    const float vv[4] = {v.x, v.y, v.z, v.w};
                    for (int r = 0; r < 4 && ncol + r < g.N; ++r) {
                        float tt = vv[r];
                        if (Rp) tt += Rp[mr * g.ldr + ncol + r];
                        Cp[mo * g.ldc + ncol + r] = tt;
                        if (g.Cb) g.Cb[mo * g.ldcb + ncol + r] = H16<HT>::bits(tt);
                    }
                }
            }
            }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the patch is re-written for the second half
        }
        if constexpr (LNF) {
            // (the launcher sends whole tiles only: `whole` holds for every workgroup of an LNF launch)
            float* lsum = reinterpret_cast<float*>(g256_smem + G256_LDS);      // [4][256]
            float* lmean = lsum + 1024; float* lrstd = lmean + 256;
            const int rloc = wm * 128 + r4;                               // block-local row of (pass 0, it 0) of this lane
            const size_t tiles_all = (size_t)nty * ntx;
            u64* const gs = ln.gran + (size_t)tile_y * ntx * 256;        // this tile row's granules: [tile_x][256]
            u64* const gq = gs + tiles_all * 256;
            // the two rounds share their second half: add the four waves' partial sums, publish, gather the other tiles' in ascending order
            auto exchange = [&](u64* gr) -> float {
                const float mine = (lsum[tid] + lsum[256 + tid]) + (lsum[512 + tid] + lsum[768 + tid]);
                ps_publish(gr + (size_t)tile_x * 256, tid, ln.epoch, __float_as_uint(mine));
                u64 t0 = __builtin_amdgcn_s_memrealtime();
                unsigned spins = 0;
                float tot = 0.f;
                for (int tx = 0; tx < ntx; ++tx) {
                    float val = mine;
                    if (tx != tile_x) {
                        const gu64* pp = (const gu64*)(gr + (size_t)tx * 256) + tid;
                        for (;;) {
                            const u64 v = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((unsigned)(v >> 32) == ln.epoch) { val = __uint_as_float((unsigned)v); break; }
                            __builtin_amdgcn_s_sleep(1);
                            if (xchg_expired(spins, t0, ln.err)) { xchg_raise(ln.err, G256_ERR_LN, spins); val = 0.f; break; }
                        }
                    }
                    tot += val;
                }
                return tot;
            };
            // ---- round 1: row sums -> mean
#pragma unroll
            for (int pass = 0; pass < 4; ++pass)
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const f32x4 y = yv[pass][it];
                    const float sw = row_sum16((y.x + y.y) + (y.z + y.w));
                    if (c == 0) lsum[wn * 256 + rloc + pass * 32 + it * 4] = sw;
                }
            __syncthreads();
            if (tid < 256) lmean[tid] = exchange(gs) / (float)g.N;
            __syncthreads();
            // ---- round 2: sums of squared deviations -> rstd  (yv <- y - mean)
#pragma unroll
            for (int pass = 0; pass < 4; ++pass)
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const float mu = lmean[rloc + pass * 32 + it * 4];
                    f32x4 d = yv[pass][it];
                    d.x -= mu; d.y -= mu; d.z -= mu; d.w -= mu;
                    yv[pass][it] = d;
                    const float qw = row_sum16((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w));
                    if (c == 0) lsum[wn * 256 + rloc + pass * 32 + it * 4] = qw;
                }
            __syncthreads();
            if (tid < 256) lrstd[tid] = 1.0f / sqrtf(exchange(gq) / (float)g.N + ln.eps);
            __syncthreads();
            // ---- normalise this tile's columns: h (fp32, over the residual this workgroup read) and its 16-bit copy
            const f32x4 ga = *reinterpret_cast<const f32x4*>(ln.gamma + ncol), be = *reinterpret_cast<const f32x4*>(ln.beta + ncol);
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const size_t m0 = (size_t)(bm + wm * 128 + pass * 32 + r4);
                float* cp = Cp + m0 * g.ldc + ncol;
                bf16_t* cb = g.Cb ? g.Cb + m0 * g.ldcb + ncol : nullptr;
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const float rs = lrstd[rloc + pass * 32 + it * 4];
                    const f32x4 d = yv[pass][it];
                    f32x4 o;
                    o.x = d.x * rs * ga.x + be.x; o.y = d.y * rs * ga.y + be.y; o.z = d.z * rs * ga.z + be.z; o.w = d.w * rs * ga.w + be.w;
                    *reinterpret_cast<f32x4*>(cp + (size_t)(it * 4) * g.ldc) = o;
                    if (cb) *reinterpret_cast<u32x2*>(cb + (size_t)(it * 4) * g.ldcb) = pack4<HT>(o);
                }
            }
        }
    }
    if (trace && tid == 0) trace[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();
}

// ---- the persistent form (round 6): one workgroup per CU walks the tile list --------------------------------------------------------------------
// What the one-tile kernel above leaves uncovered at K = 768 .. 1024 is fixed cost per tile, not the K-loop: ~2.5 us until the first K-tile has
// landed, the dispatch of a fresh 512-thread / 128-KB workgroup per tile, and the epilogue's stores, which the block can only END behind
// (profiles/r06_ubench_gemm256_epi.txt).  Here a workgroup keeps its CU for the whole launch (grid = CUs, tile = blockIdx + i * grid):
//   * the FIRST TWO K-tiles of the next tile are requested right after the K-loop's last barrier, BEFORE the epilogue of the tile just computed:
//     they land under the epilogue, and the next K-loop starts without a fill;
//   * the epilogue goes through a 4-KB patch per wave in the LDS the operand buffers do not use (128 + 8 x 4 = 160 KB, the whole LDS of a CU) --
//     four passes of 32 rows: 8 ds_write_b64, 4 ds_read_b128, 4 global_store_dwordx4 -- and its 16 stores are never waited for: they drain
//     under the next tile's first K-tiles.  The vector-memory counter completes in order, so the first wait that names a load younger than the
//     stores (P3 of K-tile 1) is also the point by which the stores must have left: ~2 K-tiles after they were issued;
//   * the bias cannot be loaded by ordinary loads without draining that queue (hipcc's own wait for a VGPR load counts only what it issued:
//     beside 16 LDS-DMA requests it would wait for nearly all of them): the wave's 64 bias values travel by ONE 4-byte LDS-DMA into the head of
//     its patch, requested after the previous epilogue, covered by the K-loop's closing vmcnt(0), read back by ds_read.
// Counted waits (vector-memory operations per wave, oldest first, when a tile starts): 16 LDS-DMA of K-tiles 0 / 1 | 16 epilogue stores | 1 bias DMA
// -> `vmcnt(17)` = both K-tiles landed (first tile: no stores yet -> vmcnt(1)); K-tile 0 stages nothing at P0 / P1 (already there) and waits for
// nothing; from K-tile 1 on the loop is the one above.  The count of 16 stores is load-bearing: tests/test_gemm256p_isa.py counts them in the ISA.
// Whole tiles only, 16-bit output only (launch_gemm_dense sends everything else to the one-tile kernel): M, N multiples of 256, plain row addressing.
// Same MFMA order per output as every other tile of this file: bit-identical results.
constexpr int G256P_PATCH = 4096;
constexpr int G256P_LDS = G256_LDS + 8 * G256P_PATCH;

MA_NO_ASAN __device__ __forceinline__ void gt_glds4(const void* gsrc, unsigned lds_dst) {       // 4 bytes per lane: LDS[dst + 4 lane] <- *gsrc
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}

// KV: the K / V column tiles of the prefill's q|k|v projection leave for the KV-cache planes (GemmTArgs::kv_*; a wave's 64 columns are one head, a
// lane's 16-byte chunk of a row is one 16-byte chunk of a cache row: the same 16 stores, other addresses -- kv_fill2_kernel's pass over the tensor goes)
template <typename HT, int ACT, int ABL = 0, bool KV = false>
MA_NO_ASAN __global__ __launch_bounds__(512) void gemm256p_kernel(GemmTArgs g, int nty, int ntx, unsigned long long* trace = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char g256_smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 2, wn = w & 3;
    const int tiles = nty * ntx;
    const unsigned lds0 = (unsigned)(size_t)g256_smem;
    const int nk = g.K >> 6;
    // tile list as in gemm256_kernel: XCD i works on the i-th eighth of the list (grid is a multiple of 8: a workgroup stays on its XCD's eighth),
    // inside it column panels of four n-tiles, m fastest
    auto tile_origin = [&](int t, int& bm, int& bn) {
        const int xcd = t & 7, i = t >> 3, lo = tiles >> 3, rem = tiles & 7;
        t = xcd * lo + min(xcd, rem) + i;
        const int fullp = ntx >> 2, tailw = ntx & 3, cut = fullp * 4 * nty;
        int tile_y, tile_x;
        if (t < cut) { const int p = t / (4 * nty), r = t - p * 4 * nty; tile_y = r >> 2; tile_x = p * 4 + (r & 3); }
        else { const int r = t - cut; tile_y = r / tailw; tile_x = fullp * 4 + r - tile_y * tailw; }
        bm = tile_y * 256; bn = tile_x * 256;
    };

    const int drow = lane >> 3, dslot = lane & 7;
    const bf16_t* xsrc[2][2]; const bf16_t* wsrc[2][2];             // [piece s][instruction i]: source of this lane, K-tile 0 of the tile being staged
    auto set_src = [&](int bm, int bn) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pr = (w + 8 * i) * 8 + drow;               // piece row
                const int sw = (dslot ^ (pr & 7)) * 8;               // swizzled 16-byte chunk of the source row
                const int m = bm + (pr >> 6) * 128 + s * 64 + (pr & 63);
                const int n = bn + (pr >> 5) * 64 + s * 32 + (pr & 31);
                xsrc[s][i] = g.A + (size_t)m * g.lda + sw;
                wsrc[s][i] = g.W + (size_t)n * g.K + sw;
            }
    };
    auto stage_x = [&](int s, int kt) {
        if constexpr (ABL & 1) { if (kt > 1) return; }
        const unsigned dst = lds0 + (unsigned)(kt & 1) * (4u * G256_PIECE) + (unsigned)s * G256_PIECE + (unsigned)w * 1024u;
        gt_glds16(xsrc[s][0] + (size_t)kt * 64, __builtin_amdgcn_readfirstlane(dst));
        gt_glds16(xsrc[s][1] + (size_t)kt * 64, __builtin_amdgcn_readfirstlane(dst + 8192u));
    };
    auto stage_w = [&](int s, int kt) {
        if constexpr (ABL & 1) { if (kt > 1) return; }
        const unsigned dst = lds0 + (unsigned)(kt & 1) * (4u * G256_PIECE) + (unsigned)(2 + s) * G256_PIECE + (unsigned)w * 1024u;
        gt_glds16(wsrc[s][0] + (size_t)kt * 64, __builtin_amdgcn_readfirstlane(dst));
        gt_glds16(wsrc[s][1] + (size_t)kt * 64, __builtin_amdgcn_readfirstlane(dst + 8192u));
    };
    auto stage_first_two = [&]() {                                   // 16 requests per wave, the order the K-loop would have issued them in
        stage_x(0, 0); stage_w(0, 0); stage_w(1, 0); stage_x(1, 0);
        stage_x(0, 1); stage_w(1, 1); stage_x(1, 1); stage_w(0, 1);
    };
    char* patch = g256_smem + G256_LDS + w * G256P_PATCH;
    // ALWAYS one request (the counted waits assume it): without a bias any readable word will do, the values are then not used
    auto stage_bias = [&](int bn) {
        const float* src = g.bias ? g.bias + bn + wn * 64 + lane : reinterpret_cast<const float*>(g.W) + lane;
        gt_glds4(src, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)G256_LDS + (unsigned)w * (unsigned)G256P_PATCH));
    };

    const int fr = lane & 15, kg = lane >> 4;
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
    u32x4 xf[4][2], wf[2][2];
    auto read_x = [&](int s, int kt) {
        if constexpr (ABL & 2) { if (kt > 0) return; }
        const char* base = g256_smem + (kt & 1) * (4 * G256_PIECE) + s * G256_PIECE;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xf[j][ks] = *reinterpret_cast<const u32x4*>(base + xoff[j][ks]);
    };
    auto read_w = [&](int s, int kt) {
        if constexpr (ABL & 2) { if (kt > 0) return; }
        const char* base = g256_smem + (kt & 1) * (4 * G256_PIECE) + (2 + s) * G256_PIECE;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wf[i][ks] = *reinterpret_cast<const u32x4*>(base + woff[i][ks]);
    };
#define G256_COMPUTE(a, b)                                                                                       \
    do {                                                                                                         \
        __builtin_amdgcn_s_barrier();                                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                                           \
        if (!(ABL & 4) || kt == 0)                                                                               \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                         \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                        \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                    \
                    acc[a][b][i][j] = H16<HT>::mfma16(wf[i][ks], xf[j][ks], acc[a][b][i][j]);                    \
        __builtin_amdgcn_s_setprio(0);                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        __builtin_amdgcn_s_barrier();                                                                            \
    } while (0)

    int tl = blockIdx.x;                                             // (grid <= tiles: every workgroup has a first tile)
    int bm, bn;
    tile_origin(tl, bm, bn);
    set_src(bm, bn);
    stage_first_two();
    stage_bias(bn);
    bool first = true;
    for (;;) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // both K-tiles of this tile have landed (younger: the previous tile's 16 stores and the bias request; first tile: the bias request)
        if (first) g256_wait_vm<1>(); else g256_wait_vm<17>();
        if (trace && tid == 0) trace[(size_t)tl * 4 + 0] = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();                   // the second wave group runs one barrier behind the first

        for (int kt = 0; kt < nk; ++kt) {
            // P0
            read_w(0, kt); __builtin_amdgcn_sched_barrier(0); read_x(0, kt);
            if (kt >= 1 && kt + 1 < nk) stage_x(1, kt + 1);
            G256_COMPUTE(0, 0);
            // P1
            read_w(1, kt);
            if (kt >= 1 && kt + 1 < nk) stage_w(0, kt + 1);
            G256_COMPUTE(0, 1);
            // P2
            read_x(1, kt);
            if (kt + 2 < nk) stage_x(0, kt + 2);
            G256_COMPUTE(1, 1);
            // P3
            read_w(0, kt);
            if (kt + 2 < nk) { stage_w(1, kt + 2); if (kt >= 1) g256_wait_vm<4>(); }     // K-tile 1 landed before the loop
            else g256_wait_vm<0>();                                  // (the last K-tile: also this tile's bias and, at the latest, the previous tile's stores)
            G256_COMPUTE(1, 0);
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();                   // (the barrier the first group is ahead by): every wave has read its last operands
        if (trace && tid == 0) trace[(size_t)tl * 4 + 1] = __builtin_amdgcn_s_memrealtime();

        // ---- next tile's first two K-tiles, then this tile's epilogue -----------------------------------------------------------------------
        const int bm_e = bm, bn_e = bn, tl_e = tl;
        tl += gridDim.x;
        const bool more = tl < tiles;
        if (more) {
            tile_origin(tl, bm, bn);
            set_src(bm, bn);
            stage_first_two();
        }
        f32x4 bias4[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bias4[b][i] = *reinterpret_cast<const f32x4*>(patch + (b * 32 + i * 16 + kg * 4) * 4);
                if (!g.bias) bias4[b][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        {
            const int r8 = lane >> 3, c = lane & 7;
            const int col0 = bn_e + wn * 64;
            bf16_t* dst = g.Cb + (size_t)(bm_e + wm * 128 + r8) * g.ldcb + (col0 + c * 8);
            const bool to_cache = KV && col0 >= g.kv_col0;           // (wave-uniform: a wave's 64 columns are one head of Q, K or V)
            int kv_b = 0, kv_pos = 0;                                 // sample / position of this lane's row of the coming store
            bf16_t* kv_plane = nullptr;
            if constexpr (KV) {
                if (to_cache) {
                    const int m0 = bm_e + wm * 128 + r8;
                    kv_b = m0 / g.kv_T; kv_pos = m0 - kv_b * g.kv_T;
                    const int cc = col0 - g.kv_col0, isv = cc >= g.kv_col0 ? 1 : 0, head = (cc - isv * g.kv_col0) >> 6;
                    kv_plane = (isv ? g.kv_v : g.kv_k) + (size_t)head * g.kv_max_seq * 64 + c * 8;
                }
            }
#pragma unroll
            for (int st = 0; st < 4; ++st) {                          // pass st: the wave's rows 32 st .. 32 st + 31
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            f32x4 v = acc[st >> 1][b][i][(st & 1) * 2 + jj];
                            const f32x4 b4 = bias4[b][i];
                            v.x = apply_act(v.x + b4.x, ACT); v.y = apply_act(v.y + b4.y, ACT); v.z = apply_act(v.z + b4.z, ACT); v.w = apply_act(v.w + b4.w, ACT);
                            const int rr = jj * 16 + fr, chunk = b * 4 + i * 2 + (kg >> 1);
                            *reinterpret_cast<u32x2*>(patch + rr * 128 + ((chunk ^ (rr & 7)) * 16) + (kg & 1) * 8) = pack4<HT>(v);
                        }
                u32x4 q[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) { const int row = it * 8 + r8; q[it] = *reinterpret_cast<const u32x4*>(patch + row * 128 + ((c ^ (row & 7)) * 16)); }
                if constexpr (!(ABL & 8)) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    bf16_t* p = dst + (size_t)(st * 32 + it * 8) * g.ldcb;
                    if constexpr (KV) {
                        if (to_cache) p = kv_plane + (size_t)kv_b * g.kv_row_stride + (size_t)kv_pos * 64;
                        kv_pos += 8;                                  // the next store's row is 8 further (kv_T >= 8: at most one sample boundary)
                        if (kv_pos >= g.kv_T) { kv_pos -= g.kv_T; ++kv_b; }
                    }
                    *reinterpret_cast<u32x4*>(p) = q[it];
                }
                } else { asm volatile("" :: "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3])); }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the patch's last read has returned: its head may take the next bias
        if (!more) {
            if (trace && tid == 0) trace[(size_t)tl_e * 4 + 2] = __builtin_amdgcn_s_memrealtime();
            break;
        }
        stage_bias(bn);
        if (trace && tid == 0) trace[(size_t)tl_e * 4 + 2] = __builtin_amdgcn_s_memrealtime();
        first = false;
    }
#undef G256_COMPUTE
}

// the LNF form of the one-tile kernel (LayerNorm finished in the epilogue)
template <typename HT>
inline hipError_t g256_launch_ln(const GemmTArgs& g, int nty, int ntx, hipStream_t s, const G256Ln& ln) {
    static bool attr = false;
    if (!attr) {
        hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<HT, ACT_NONE, 0, G256_EV, true>), hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS + G256_LN_LDS);
        if (r != hipSuccess) return r;
        attr = true;
    }
    hipLaunchKernelGGL((gemm256_kernel<HT, ACT_NONE, 0, G256_EV, true>), dim3(nty * ntx), dim3(512), G256_LDS + G256_LN_LDS, s, g, nty, ntx, (unsigned long long*)nullptr, 1, 0L, ln);
    return hipGetLastError();
}

template <typename HT, int ACT, bool KV = false>
inline hipError_t g256p_launch(const GemmTArgs& g, int nty, int ntx, int n_cus, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<HT, ACT, 0, KV>), hipFuncAttributeMaxDynamicSharedMemorySize, G256P_LDS);
        if (r != hipSuccess) return r;
        attr = true;
    }
    const int grid = nty * ntx <= n_cus ? nty * ntx : n_cus & ~7;      // (several rounds: a multiple of 8, so that a workgroup stays on its XCD's part of the tile list)
    hipLaunchKernelGGL((gemm256p_kernel<HT, ACT, 0, KV>), dim3(grid), dim3(512), G256P_LDS, s, g, nty, ntx, (unsigned long long*)nullptr);
    return hipGetLastError();
}

template <typename HT, int ACT>
inline hipError_t g256_launch(const GemmTArgs& g, int nty, int ntx, hipStream_t s, int ks = 1, long part_stride = 0) {
    static bool attr = false;
    if (!attr) {
        hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<HT, ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS);
        if (r != hipSuccess) return r;
        attr = true;
    }
    hipLaunchKernelGGL((gemm256_kernel<HT, ACT>), dim3(nty * ntx * ks), dim3(512), G256_LDS, s, g, nty, ntx, (unsigned long long*)nullptr, ks, part_stride);
    return hipGetLastError();
}

// engine option "gemm256" (default 2): 0 = the 128-row tiles only; 1 = the one-tile 256 x 256 kernel for the shapes it covers; 2 = 1 + the persistent
// form for whole-tile 16-bit-output problems
inline int& gemm256_enabled() { static int v = 2; return v; }

// The dense GEMM of the 16-bit policies.  The big tile wants rounds of the chip (one block per CU) that are reasonably full: the matrix is cut along M into
//   * the leading tile rows whose tiles fill their rounds to 60 % or more -- all whole tile rows when that holds (M = 64 x 257: 64 rows of 4 / 12 /
//     16 tiles = 1 / 3 / 4 rounds; M = 16 x 257, N = 3072: 192 tiles = 0.75 round), else the largest count that makes exact rounds;
//   * the rest: a tail of <= 64 rows (M = B x 257 and B x 1057 leave 64 at B = 64) goes to the skinny matrix-core GEMM of the batched decode
//     step (gemm_decode.hpp: 16 weight rows per block, the tail's rows as the B operand); anything larger to the tiles of gemm_tile.hpp.
// Problems that do not give the big tile one full round (small M of batch-1 runs, N = 64 / 128 projections, K not a multiple of 64) stay on
// gemm_tile.hpp entirely.  Outputs through a row map / a broadcast residual are not split (the encoder's, fp32 under enc_exact anyway).
inline int g256_gcd(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

// kv_rows (optional): with GemmTArgs::kv_k set, the number of leading rows whose K / V columns went to the cache planes (0: none -- the caller fills
// the cache from the output tensor, as it does for the rows behind that count)
// sk (optional): the caller can take the fp32 output as up to `max_parts` partial sums along K (buffers g.C + p * part_stride, to be added up by the
// consumer: ln_rows2_kernel's KS form): used when K is long and the 256 x 256 tiles would fill well under a round of the chip.  On return `parts` (1: not
// split) and `rows` = the leading rows that ARE split (the tail rows behind them are complete in part 0).
struct GemmSplitK { int max_parts = 1; long part_stride = 0; int parts = 1; int rows = 0; };
// lnf (optional): the caller wants LayerNorm(gamma, beta, eps) of the fp32 result instead of the result: C <- LN(A W^T + bias + R) (may be in place over R),
// Cb its 16-bit copy.  On return `rows` = the leading rows for which the GEMM did it (the one-tile kernel's LNF form; 0: none) -- the caller runs the
// row kernel over the rows behind them, whose plain sums were written to `tail_c` (same leading dimension as C) instead of C.
struct GemmLnFuse { G256Ln ln; float* tail_c = nullptr; int rows = 0; };
template <typename HT> inline hipError_t launch_gemm_dense_impl(const GemmTArgs& g, const GemmTArgs& g_ln, int n_cus, hipStream_t s, int* kv_rows, GemmSplitK* sk, GemmLnFuse* lnf);
template <typename HT>
inline hipError_t launch_gemm_dense(const GemmTArgs& g, int n_cus, hipStream_t s, int* kv_rows = nullptr, GemmSplitK* sk = nullptr, GemmLnFuse* lnf = nullptr) {
    if (lnf && lnf->tail_c) {
        // every launch that does NOT finish the LayerNorm writes plain sums to the caller's tail buffer and no 16-bit copy (the caller's row kernel follows)
        GemmTArgs plain = g;
        plain.C = lnf->tail_c; plain.Cb = nullptr;
        return launch_gemm_dense_impl<HT>(plain, g, n_cus, s, kv_rows, sk, lnf);
    }
    return launch_gemm_dense_impl<HT>(g, g, n_cus, s, kv_rows, sk, nullptr);
}
// g: the arguments of the plain launches; g_ln: those of the LNF launch (outputs = the LayerNorm's)
template <typename HT>
inline hipError_t launch_gemm_dense_impl(const GemmTArgs& g, const GemmTArgs& g_ln, int n_cus, hipStream_t s, int* kv_rows, GemmSplitK* sk, GemmLnFuse* lnf) {
    if (kv_rows) *kv_rows = 0;
    if (sk) { sk->parts = 1; sk->rows = 0; }
    if (lnf) lnf->rows = 0;
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if (g.K % 32 != 0 || g.lda % 8 != 0 || (g.C && g.ldc % 4) || (g.R && g.ldr % 4) || (g.Cb && g.ldcb % 4) || (!g.C && !g.Cb)) return hipErrorInvalidValue;
    // the 256-row tile's 16-bit-only epilogue (C == nullptr) carries no residual and stores 16 bytes to Cb: such calls take the 128-row tile
    const bool tile256_ok = !(g.R && !g.C) && (!g.Cb || g.ldcb % 8 == 0);
    // rows [r0, r0 + rows) on the small kernels: the skinny matrix-core GEMM of the batched decode step (`dec`), else the 128-row tiles in the shape that
    // `choice_rows` rows would get (the stretch these rows belong to in the call that computes everything)
    auto rows_on_small = [&](size_t r0, int rows, bool dec, int choice_rows) -> hipError_t {
        if (dec) {
            GemmDecArgs d{};
            d.W = g.W; d.bias = g.bias; d.xb = g.A + r0 * g.lda; d.xb_stride = g.lda; d.N = g.N; d.K = g.K; d.B = rows; d.act = g.act; d.epi = EPI_PLAIN; d.ksplit = 1;
            if (g.R) { d.res = g.R + r0 * g.ldr; d.res_stride = g.ldr; }
            if (g.C) { d.y = g.C + r0 * g.ldc; d.y_stride = g.ldc; }
            if (g.Cb) { d.yb = g.Cb + r0 * g.ldcb; d.yb_stride = g.ldcb; }
            return launch_gemm_dec<HT>(d, s);
        }
        GemmTArgs t = g;
        t.A = g.A + r0 * g.lda; t.M = rows;
        if (g.R) t.R = g.R + r0 * g.ldr;
        t.C = g.C ? g.C + r0 * g.ldc : nullptr;
        t.Cb = g.Cb ? g.Cb + r0 * g.ldcb : nullptr;
        return launch_gemm_tile<HT>(t, s, choice_rows);
    };
    const int part = g.part;
    const int Mm = g.M - g.M % 256;                                  // parts 1 | 2 meet here
    if (part != 0 && (g.cmap.grp != 0 || g.r_mod != 0 || lnf)) return hipErrorInvalidValue;
    if (tile256_ok && gemm256_enabled() && n_cus > 0 && g.K % 64 == 0 && g.K >= 128 && g.N >= 256 && g.M >= 256 && (long)((g.N + 255) / 256) * ((g.M + 255) / 256) < (1L << 24)) {
        const int ntx = (g.N + 255) / 256;
        const bool can_split = g.cmap.grp == 0 && g.r_mod == 0;
        // (round 6: 60 % instead of 88 % + at least one whole round.  A tile of the big kernel now runs at ~1.2 PFLOP/s per CU-round against ~0.6 for the
        //  128-row tiles at their best, so a launch whose last -- or only -- round is 60 % full still comes out ahead: 576 tiles = 2.25 rounds 934 vs 596
        //  TFLOP/s, 192 tiles = 0.75 round 911 vs ~450; profiles/r06_ubench_gemm256p.txt, r06_calib_library_gemm.txt)
        auto fills = [&](long tiles) { const long rounds = (tiles + n_cus - 1) / n_cus; return tiles * 100 >= rounds * n_cus * 60; };
        int nty = 0;                                                 // tile rows on the big tile
        int ks = 1;
        if (sk && sk->max_parts >= 2 && gemm256_enabled() >= 2 && can_split && g.C && !g.Cb && g.act == ACT_NONE && g.N % 256 == 0 && g.K >= 1024 && !fills((long)ntx * (g.M / 256)) &&
            (long)ntx * (g.M / 256) >= 8) {
            // few tiles, long K: 4 (or 2) parts along K so that the launch comes to about one round (fc2 at 16 samples: 64 tiles x 4 = one round of 16 K-tiles
            // instead of a quarter round of 64)
            const long tiles = (long)ntx * (g.M / 256);
            ks = (sk->max_parts >= 4 && tiles * 4 <= n_cus + n_cus / 8 && g.K % 256 == 0) ? 4 : ((tiles * 2 <= n_cus + n_cus / 8 && g.K % 128 == 0) ? 2 : 1);
            if (ks > 1) nty = g.M / 256;
        }
        if (ks > 1) {}
        else if (!can_split) { if (fills((long)ntx * ((g.M + 255) / 256))) nty = (g.M + 255) / 256; }      // ragged last tile row computed with clamped rows
        else {
            // all whole tile rows (when their rounds are reasonably full), or the largest count that makes EXACT rounds with the remaining rows on the 128-row
            // tiles -- whichever the estimate below puts first.  (The detokenizer at 64 samples: 264 tile rows x 3 tiles = 3.09 rounds; the fourth round
            // cost a whole tile time, 33 - 88 us, for 24 tiles, where the 2 112 remaining rows take 10 - 35 us on the small tiles.)
            const int nty_full = fills((long)ntx * (g.M / 256)) ? g.M / 256 : 0;
            const int q = n_cus / g256_gcd(ntx, n_cus), nty_exact = g.M / 256 / q * q;
            auto cost_us = [&](int n) -> double {
                if (n <= 0) return 1e30;
                const double t_tile = (g.K / 64) * 1.5 + (g.C ? 12.0 : 5.0) + 3.0;
                const long rounds = ((long)n * ntx + n_cus - 1) / n_cus;
                const long rest = g.M - (long)n * 256;
                const double t_rest = rest == 0 ? 0.0 : (rest <= 64 ? 11.0 : 15.0 + 2.0 * (double)rest * g.N * g.K / 2.5e8);      // (measured: 2 176 rows x 768 x 768 / 3072 on the small tiles 19 - 50 us)
                return rounds * t_tile + t_rest;
            };
            nty = cost_us(nty_exact) < cost_us(nty_full) ? nty_exact : nty_full;
        }
        if (nty > 0) {
            GemmTArgs m = g;
            m.M = can_split ? nty * 256 : g.M;
            // whole tiles with 16-bit output: the persistent form (next tile's operands requested before this tile's epilogue; its 4-KB-patch epilogue is
            // the faster one even where every workgroup has a single tile)
            const bool persist = gemm256_enabled() >= 2 && can_split && !g.C && g.Cb && !g.R && g.N % 256 == 0 && n_cus >= 8;
            // (measured and not kept, round 6: at 64 samples out_proj / fc2 with the LayerNorm inside took 79 / 158 us against 42 + 30 / 124 + 30 us for GEMM +
            //  row kernel -- the epilogue holds the 128 x 64 sums of a wave in registers across two exchanges (97 dwords per lane spilled) and every tile row
            //  waits for its slowest tile twice; profiles/r06_ab_layernorm_in_gemm.txt.  Compiled with MA_EXPERIMENTAL=1 only.)
#ifdef MA_EXPERIMENTAL
            const bool do_ln = lnf && ks == 1 && gemm256_enabled() >= 2 && can_split && g.C && g.act == ACT_NONE && g.N % 256 == 0 && lnf->ln.gran && lnf->ln.gamma && lnf->ln.beta &&
                               lnf->ln.err && lnf->ln.epoch != 0 && g_ln.C && g_ln.ldc % 4 == 0 && (!g_ln.Cb || g_ln.ldcb % 4 == 0);
            if (do_ln && part != 2) {
                lnf->rows = nty * 256;
                GemmTArgs ml = g_ln;
                ml.M = nty * 256;
                hipError_t r = g256_launch_ln<HT>(ml, nty, ntx, s, lnf->ln);
                if (r != hipSuccess) return r;
            }
#else
            const bool do_ln = false;
            (void)g_ln;
#endif
            if (ks > 1) {
                sk->parts = ks; sk->rows = nty * 256;
                hipError_t r = part == 2 ? hipSuccess : g256_launch<HT, ACT_NONE>(m, nty, ntx, s, ks, sk->part_stride);
                if (r != hipSuccess) return r;
            }
            const bool kv = ks == 1 && persist && kv_rows && g.kv_k && g.kv_v && g.act == ACT_NONE && g.kv_T >= 8 && g.kv_col0 % 64 == 0 && g.N == 3 * g.kv_col0;
            if (kv) {
                *kv_rows = nty * 256;
                hipError_t r = part == 2 ? hipSuccess : g256p_launch<HT, ACT_NONE, true>(m, nty, ntx, n_cus, s);
                if (r != hipSuccess) return r;
            }
            hipError_t r = (kv || ks > 1 || do_ln || part == 2) ? hipSuccess : persist ? (g.act == ACT_RELU ? g256p_launch<HT, ACT_RELU>(m, nty, ntx, n_cus, s) : g.act == ACT_GELU ? g256p_launch<HT, ACT_GELU>(m, nty, ntx, n_cus, s)
                                                                                                                             : g256p_launch<HT, ACT_NONE>(m, nty, ntx, n_cus, s))
                                   : (g.act == ACT_RELU ? g256_launch<HT, ACT_RELU>(m, nty, ntx, s) : g.act == ACT_GELU ? g256_launch<HT, ACT_GELU>(m, nty, ntx, s) : g256_launch<HT, ACT_NONE>(m, nty, ntx, s));
            const int rest = can_split ? g.M - nty * 256 : 0;
            if (r != hipSuccess || rest == 0) return r;
            // the rows behind the tiles: <= 64 of them on the skinny GEMM, more on the 128-row tiles (with lnf: g.C is the caller's tail buffer and g.Cb null -- they
            // leave as plain sums, their LayerNorm is the caller's row kernel); a call for part 1 | 2 runs the stretch of them on its side of Mm
            const bool dec = rest <= 64 && g.K % 128 == 0;
            const int rb = part == 2 ? std::max(nty * 256, Mm) : nty * 256, re = part == 1 ? Mm : g.M;
            return re > rb ? rows_on_small((size_t)rb, re - rb, dec, rest) : hipSuccess;
        }
    }
    if (part == 1) return Mm > 0 ? rows_on_small(0, Mm, false, g.M) : hipSuccess;
    if (part == 2) return g.M > Mm ? rows_on_small((size_t)Mm, g.M - Mm, false, g.M) : hipSuccess;
    return launch_gemm_tile<HT>(g, s);
}

}  // namespace ma
