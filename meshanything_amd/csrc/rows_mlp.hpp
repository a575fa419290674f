// The second half of a decoder layer for EIGHT rows stepping together, in ONE launch: LayerNorm 1 (folded prologue) -> fc1 + bias + ReLU on
// the matrix cores -> fc2 (split along K over four blocks per 16-row tile; raw partial sums, whose bias / residual / LayerNorm 2 run in the
// next layer's prologue).  Replaces two launches of the batched matrix-core chain (gemm_dec_ln_kernel<1, false, 8> | gemm_dec_kernel<1, 8>
// with ksplit 4; [3p] OPTDecoderLayer fc1 / activation_fn / fc2 reached from shape_opt.py:403-410).  The seam between them is an all-gather
// of relu(fc1): 8 rows x 4096 16-bit values -- but a K-split fc2 block needs only its quarter, 16 KB (scripts/ubench_allgather.hip prices
// that gather against the launch boundary it replaces: profiles/r05_ubench_allgather.txt).
//
// grid 256 blocks of 8 waves, one per CU (all resident: the gate, bounded sweeps and fall-back of the other fused launches).  Block i:
//   A. requests: LayerNorm parameters, row `wave` of y1, its fc1 tile (rows 16 i .. 16 i + 15, the 8 waves split K = 1024), then its fc2 tile
//      (rows 16 (i >> 2) .., K quarter i & 3: 32 KB, requested once the row has collapsed) -- both weight streams are under way before the
//      first LayerNorm instruction;
//   B. normalises the eight rows (one per wave), parks them in LDS as 16-bit; block 0 writes the fp32 rows (h1: the residual the next
//      layer's prologue adds);
//   C. fc1 MFMAs, the waves meet in LDS (gemm_dec_ln_kernel's order), wave 0 adds bias, ReLU, rounds and publishes 16 values per row as
//      {epoch, two 16-bit values} granules;
//   D. every wave sweeps one row's 512 granules of the block's K quarter into LDS; four waves run gemm_dec_kernel<1, 8>'s K split of the
//      fc2 tile; wave 0 stores the raw partial sums [quarter][row][1024] -- the bits the two-launch form produces;
//   F. (pf_k != null) L2 prefetch for the next launch by the 248 blocks that have nothing left to do while blocks 0 .. 7 run step E: workgroups go
//      to the XCDs round-robin (observed; only speed depends on it), so block i of this launch and block i of the next share an L2 -- it reads
//      what that block will ask for first: its 24 KB of q/k/v weights and its first cache round (64 KB; 16 MB per launch, which the idle HBM
//      delivers in the ~2.5 us step E takes).  A stream touched by the previous kernel comes back at 15 TB/s instead of 5.5
//      (profiles/r02_ubench_l2_mall_residency.txt).  MEASURED, NOT KEPT (engine option rows_mlp_prefetch, default 0): the next launch does start
//      faster (q/k/v MFMAs done 1.75 us after the block's start instead of 2.35, exchange over at 5.05 instead of 7.58, kernel 27.8 instead of
//      30.0 us), but the prefetch outlasts step E and the launch ends 1.7 us later: step 1063 vs 1055 us with one round, 1100 with two, 1052 vs
//      1052 with half a round (profiles/r05_ab_rows_mlp_prefetch.txt);
//   E. (ln2_g != null) LayerNorm 2 finished HERE instead of in every block of the next launch (where summing 4 partials + bias + residual of
//      8 rows was 160 KB of L2 reads per block, 41 MB per launch: 6 of the 10 us before the first q/k/v MFMA, profiles/r05_decode_step_
//      timeline_b8_*): wave 0 also publishes its partial sums as granules; block r < 8 gathers row r (4096 granules, 8 per thread), adds them in
//      the order of the folded prologue (quarters 0 .. 3, + bias, + residual: its own LayerNorm 1 output, kept in LDS), one wave normalises
//      (gemm_dec_ln_kernel::norm_row) and writes the fp32 row (the next out_proj's residual) and the 16-bit row (the next q/k/v operand).
// Epoch = position * 32 + layer + 1, as in rows_attn.hpp (its own buffer).  HBM-bound: 16 MB of fc1 + fc2 weights per launch.
#pragma once
#include "common.hpp"
#include "gemm_decode.hpp"
#include "rows_attn.hpp"
#include "state.hpp"

namespace ma {

constexpr unsigned RM_ERR_FFN = 1024;
constexpr int RM_FFN_GRANULES = 4096 / 2;                  // per row: relu(fc1) as pairs
constexpr int RM_Y2_GRANULES = 4 * 1024;                   // per row: fc2 partial sums [quarter][column], one fp32 each

struct RowsMlpArgs {
    const float* y1; int y1_stride;                        // [8][1024] fp32: residual + out_proj (rows_attn.hpp)
    const float* ln_g; const float* ln_b; float ln_eps;    // LayerNorm 1
    float* h1_out; int h1_stride;                          // its fp32 output (the residual of fc2, added by the next layer's prologue)
    const bf16_t* W1; const float* b1;                     // fc1 [4096][1024], [4096]
    const bf16_t* W2;                                      // fc2 [1024][4096]
    float* part; int part_stride;                          // out: raw fc2 partial sums [4][8][part_stride]
    const DecState* st; int layer;
    u64* ffn_gran;                                         // [8][RM_FFN_GRANULES]
    // step E (ln2_g != null): LayerNorm 2 of y2 = sum of the four partials + b2 + h1
    const float* b2; const float* ln2_g; const float* ln2_b;
    u64* y2_gran;                                          // [8][RM_Y2_GRANULES]
    float* x2_out; int x2_stride;                          // fp32 LayerNorm 2 output [8][1024]
    bf16_t* xb_out; int xb_stride;                         // ... and as 16-bit
    // step F (pf_k != null): blocks 8 .. 255, idle while blocks 0 .. 7 finish LayerNorm 2, pull the operands that block (i & 15, (i >> 4) & 7, i >> 7) of
    // the NEXT layer's rows_attn launch asks for first into this XCD's L2: its q/k/v weight tile and its first `pf_rounds` cache rounds
    const bf16_t* pf_k; const bf16_t* pf_v; size_t pf_row_stride; int pf_max_seq; int pf_rounds;
    const bf16_t* pf_wqkv;
    unsigned* pf_sink;
    unsigned* err;
    unsigned long long* trace;
};

template <typename HT>
__global__ __launch_bounds__(512) void rows_mlp_kernel(RowsMlpArgs a) {
    constexpr int K = 1024, F = 4096, NW = 8, KW = K / NW, CH = KW / 32, XS = K + 16;
    __shared__ __attribute__((aligned(16))) bf16_t xl[RA_ROWS * XS];       // LayerNorm output as 16-bit (step C); later the block's quarter of relu(fc1) (step D)
    __shared__ __attribute__((aligned(16))) float red[NW][64][4];
    __shared__ __attribute__((aligned(16))) float h1l[K];               // step E, blocks 0 .. 7: LayerNorm 1 output of row `block` (fp32); later y2 of that row
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (wave-uniform: branches on w are scalar branches)
    const int m = lane & 15, kg = lane >> 4;
    const int i = blockIdx.x;
    const int n1 = i * 16;                                  // fc1 rows of this block
    const int t2 = i >> 2, kq = i & 3, n2 = t2 * 16;        // fc2 tile and K quarter
    const unsigned epoch = (unsigned)a.st[0].pos * 32u + (unsigned)a.layer + 1u;       // (rows step together: one position; read before any store: a scalar load)
    unsigned long long* tr = a.trace ? a.trace + (size_t)i * 4 : nullptr;
    if (tr && threadIdx.x == 0) tr[0] = __builtin_amdgcn_s_memrealtime();

    // ---- A: requests ------------------------------------------------------------------------------------------------------------------
    f32x4 gv[4], bv[4], s[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int idx = (lane + 64 * c) * 4;
        gv[c] = *reinterpret_cast<const f32x4*>(a.ln_g + idx);
        bv[c] = *reinterpret_cast<const f32x4*>(a.ln_b + idx);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) s[c] = *reinterpret_cast<const f32x4*>(a.y1 + (size_t)w * a.y1_stride + (lane + 64 * c) * 4);
    asm volatile("" ::: "memory");
    const int kbase = w * KW + kg * 8;
    const bf16_t* w1row = a.W1 + (size_t)(n1 + m) * K + kbase;
    u32x4 wv[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) wv[c] = ld_stream16(w1row + c * 32);
    const f32x4 fb = *reinterpret_cast<const f32x4*>(a.b1 + n1 + kg * 4);
    asm volatile("" ::: "memory");
    // (pin the row's first use behind every request above: rows_attn.hpp, step A)
#pragma unroll
    for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(s[c]));

    // ---- B: LayerNorm 1 of row `w` (gemm_dec_ln_kernel::norm_row) -----------------------------------------------------------------------
    {
        const float x0 = readlane_f(s[0].x, 0);
        float sm1 = 0.f, sq1 = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) ln_chunk_moments(s[c], x0, sm1, sq1);
        sm1 = wave_sum(sm1); sq1 = wave_sum(sq1);
        float md, rstd;
        ln_finish(sm1, 0.f, 0.f, 0.f, sq1, 0.f, 0.f, 0.f, K, a.ln_eps, md, rstd);
        const bool writer = a.h1_out && i == 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int idx = (lane + 64 * c) * 4;
            ln_apply(s[c], md, rstd, gv[c], bv[c]);
            if (writer) *reinterpret_cast<f32x4*>(a.h1_out + (size_t)w * a.h1_stride + idx) = s[c];
            *reinterpret_cast<u32x2*>(&xl[w * XS + idx]) = pack4<HT>(s[c]);
            if (a.ln2_g && i == w) *reinterpret_cast<f32x4*>(&h1l[idx]) = s[c];        // block r keeps row r in fp32: the residual of step E
        }
    }
    asm volatile("" ::: "memory");
    __syncthreads();
    if (tr && threadIdx.x == 0) tr[1] = __builtin_amdgcn_s_memrealtime();          // rows normalised and staged

    // ---- C: fc1 tile for all eight rows -------------------------------------------------------------------------------------------------
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const bf16_t* xr = xl + min(m, RA_ROWS - 1) * XS + kbase;
#pragma unroll
        for (int c = 0; c < CH; ++c) acc = H16<HT>::mfma16(wv[c], *reinterpret_cast<const u32x4*>(xr + c * 32), acc);
        *reinterpret_cast<f32x4*>(&red[w][lane][0]) = acc;
    }
    __syncthreads();
    // fc2 tile: rows n2 + m, the K split of gemm_dec_kernel<1, 8> with ksplit 4 (four waves x 256 inside the quarter; waves 4 .. 7 shadow 0 .. 3).
    // Requested behind the fc1 reduction -- by wave 0 only after it has published: a CU takes ~20 GB/s from HBM, so 8 waves x 17 requests (136 KB)
    // in front of the LayerNorm barrier held every wave at the ISSUE of its requests for 4 us (rows staged 5.1 us after the block's start instead
    // of 1.5: in-kernel stamps, profiles/r05_decode_step_timeline_b8_*); now the tile streams under the publish and the sweep.
    const bf16_t* w2row = a.W2 + (size_t)(n2 + m) * F + kq * 1024 + (w & 3) * 256 + kg * 8;
    u32x4 w2[8];
    f32x4 b2v[1], g2v[4], be2v[4];                          // step E operands of the blocks that run it (rows' bias chunk of this thread; LayerNorm 2 parameters of wave 0)
    auto request_second = [&] {
#pragma unroll
        for (int c = 0; c < 8; ++c) w2[c] = ld_stream16(w2row + c * 32);
        if (a.ln2_g) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int idx = (lane + 64 * c) * 4;
            g2v[c] = *reinterpret_cast<const f32x4*>(a.ln2_g + idx);
            be2v[c] = *reinterpret_cast<const f32x4*>(a.ln2_b + idx);
        }
        b2v[0] = *reinterpret_cast<const f32x4*>(a.b2 + 128 * w + (lane >> 1) * 4);        // (only .xy / .zw of it are used: columns 128 w + 2 lane, + 1)
        }
        asm volatile("" ::: "memory");
    };
    if (w != 0) request_second();
    if (w == 0 && m < RA_ROWS) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][lane][0]);
#pragma unroll
        for (int c = 1; c < NW; ++c) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(&red[c][lane][0]);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        v.x = fmaxf(v.x + fb.x, 0.f); v.y = fmaxf(v.y + fb.y, 0.f); v.z = fmaxf(v.z + fb.z, 0.f); v.w = fmaxf(v.w + fb.w, 0.f);
        // lane (m, kg): row m, fc1 outputs n1 + 4 kg .. + 3
        u64* g = a.ffn_gran + (size_t)m * RM_FFN_GRANULES + (n1 + 4 * kg) / 2;
        ps_publish(g, 0, epoch, H16<HT>::pack2(v.x, v.y));
        ps_publish(g, 1, epoch, H16<HT>::pack2(v.z, v.w));
    }
    if (w == 0) request_second();
    if (tr && threadIdx.x == 0) tr[2] = __builtin_amdgcn_s_memrealtime();          // fc1 outputs published
    __syncthreads();                                        // (xl: every wave is past its step-C reads)

    // ---- D: the K quarter of relu(fc1), row `w`: 512 granules; then the fc2 tile ----------------------------------------------------------
    {
        const gu64* ga = (const gu64*)(a.ffn_gran + (size_t)w * RM_FFN_GRANULES + kq * 512);
        u64 t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        u64 v[8];
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                v[c] = __hip_atomic_load(ga + lane + 64 * c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (unsigned)(v[c] >> 32) == epoch;
            }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if (xchg_expired(spins, t0, a.err)) {
                if (lane == 0) xchg_raise(a.err, RM_ERR_FFN, spins);
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = 0;
                break;
            }
        }
        if (lane == 0) xchg_note_slow(a.err, spins, t0);
        unsigned* al = reinterpret_cast<unsigned*>(xl + w * XS);
#pragma unroll
        for (int c = 0; c < 8; ++c) al[lane + 64 * c] = (unsigned)v[c];
    }
    __syncthreads();
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const bf16_t* xr = xl + min(m, RA_ROWS - 1) * XS + (w & 3) * 256 + kg * 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) acc = H16<HT>::mfma16(w2[c], *reinterpret_cast<const u32x4*>(xr + c * 32), acc);
        *reinterpret_cast<f32x4*>(&red[w][lane][0]) = acc;
    }
    __syncthreads();
    if (w == 0 && m < RA_ROWS) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][lane][0]);
#pragma unroll
        for (int c = 1; c < 4; ++c) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(&red[c][lane][0]);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        if (a.part) *reinterpret_cast<f32x4*>(a.part + ((size_t)kq * RA_ROWS + m) * a.part_stride + n2 + kg * 4) = v;      // raw partial sums (gd_epi_store, ksplit > 1)
        if (a.ln2_g) {
            u64* g = a.y2_gran + (size_t)m * RM_Y2_GRANULES + kq * 1024 + n2 + kg * 4;
            ps_publish(g, 0, epoch, __float_as_uint(v.x)); ps_publish(g, 1, epoch, __float_as_uint(v.y));
            ps_publish(g, 2, epoch, __float_as_uint(v.z)); ps_publish(g, 3, epoch, __float_as_uint(v.w));
        }
        if (tr && threadIdx.x == 0) tr[3] = __builtin_amdgcn_s_memrealtime();
    }
    if (a.pf_k && i >= RA_ROWS) {
        // ---- F: the next launch's first operands -> this XCD's L2 -----------------------------------------------------------------------
        const int h = i & 15, b = (i >> 4) & 7, z = i >> 7, j = 2 * b + z, g0 = z == 0 ? 1 : 0;
        const int slot = lane >> 3, dsub = lane & 7;
        const int end = (int)(epoch - (unsigned)a.layer - 1u) / 32 + 1;          // position + 1 (the epoch is position * 32 + layer + 1)
        unsigned acc = 0;
        {   // rows_attn_kernel's weight tile: wave w, lane (m, kg): row 4 j + (m & 3) of part m >> 2 of head h, the eight waves' K split
            const bf16_t* wrow = a.pf_wqkv + (size_t)(min(m >> 2, 2) * K + 64 * h + 4 * j + (m & 3)) * K + w * KW + kg * 8;
#pragma unroll
            for (int c = 0; c < CH; ++c) { const u32x4 t = *reinterpret_cast<const u32x4*>(wrow + c * 32); acc ^= t.x ^ t.w; }
        }
        const bf16_t* kh = a.pf_k + (size_t)b * a.pf_row_stride + (size_t)h * a.pf_max_seq * 64 + dsub * 8;
        const bf16_t* vh = a.pf_v + (size_t)b * a.pf_row_stride + (size_t)h * a.pf_max_seq * 64 + dsub * 8;
        // pf_rounds: 1, 2 = that many whole rounds; 8 = the weight tile only; 9 = half of the first round (the waves 0 .. 3 of the block)
        const int nr = a.pf_rounds == 8 ? 0 : a.pf_rounds == 9 ? (w < 4 ? 1 : 0) : a.pf_rounds;
        for (int r = 0; r < nr; ++r) {
            const int base = (g0 + 2 * r) * 256 + w * 32 + slot;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = base + u * 8;
                const u32x4 tk = *reinterpret_cast<const u32x4*>(kh + (size_t)(p < end ? p : 0) * 64);
                const u32x4 tv = *reinterpret_cast<const u32x4*>(vh + (size_t)(p < end ? p : 0) * 64);
                acc ^= tk.x ^ tv.w;
            }
        }
        if (acc == 0x9e3779b9u && a.pf_sink) a.pf_sink[0] = acc;      // (keeps the requests: their data is not used here)
        return;
    }
    if (!a.ln2_g || i >= RA_ROWS) return;

    // ---- E: LayerNorm 2 of row i (blocks 0 .. 7) -------------------------------------------------------------------------------------------
    {   // thread (w, lane): columns 128 w + 2 lane, + 1 of the four quarters: 8 granules
        const int c0 = 128 * w + 2 * lane;
        const gu64* gy = (const gu64*)(a.y2_gran + (size_t)i * RM_Y2_GRANULES + c0);
        u64 t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        u64 v[8];
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    v[2 * q + e] = __hip_atomic_load(gy + q * 1024 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (unsigned)(v[2 * q + e] >> 32) == epoch;
                }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if (xchg_expired(spins, t0, a.err)) {
                if (lane == 0) xchg_raise(a.err, RM_ERR_FFN, spins);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = 0;
                break;
            }
        }
        if (lane == 0) xchg_note_slow(a.err, spins, t0);
        // the folded prologue's order (gemm_dec_ln_kernel::sum_row): quarters 0 .. 3, + bias, + residual
        const float bb[4] = {b2v[0].x, b2v[0].y, b2v[0].z, b2v[0].w};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float y = __uint_as_float((unsigned)v[e]);
            y += __uint_as_float((unsigned)v[2 + e]); y += __uint_as_float((unsigned)v[4 + e]); y += __uint_as_float((unsigned)v[6 + e]);
            y += bb[2 * (lane & 1) + e];
            y += h1l[c0 + e];
            h1l[c0 + e] = y;                                // (this thread's own two columns: read, then overwritten)
        }
    }
    __syncthreads();
    if (w == 0) {                                           // gemm_dec_ln_kernel::norm_row on one wave
        f32x4 y[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) y[c] = *reinterpret_cast<const f32x4*>(&h1l[(lane + 64 * c) * 4]);
        const float x0 = readlane_f(y[0].x, 0);
        float sm1 = 0.f, sq1 = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) ln_chunk_moments(y[c], x0, sm1, sq1);
        sm1 = wave_sum(sm1); sq1 = wave_sum(sq1);
        float md, rstd;
        ln_finish(sm1, 0.f, 0.f, 0.f, sq1, 0.f, 0.f, 0.f, K, a.ln_eps, md, rstd);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int idx = (lane + 64 * c) * 4;
            ln_apply(y[c], md, rstd, g2v[c], be2v[c]);
            *reinterpret_cast<f32x4*>(a.x2_out + (size_t)i * a.x2_stride + idx) = y[c];
            *reinterpret_cast<u32x2*>(a.xb_out + (size_t)i * a.xb_stride + idx) = pack4<HT>(y[c]);
        }
    }
}

template <typename HT>
inline hipError_t launch_rows_mlp(const RowsMlpArgs& a, int rows, int hidden, int ffn, hipStream_t s) {
    if (rows != RA_ROWS || hidden != 1024 || ffn != 4096 || !a.y1 || !a.ln_g || !a.ln_b || !a.W1 || !a.b1 || !a.W2 || (!a.part && !a.ln2_g) || !a.st || !a.ffn_gran || !a.err ||
        a.y1_stride % 4 || a.part_stride % 4 || (a.h1_out && a.h1_stride % 4)) return hipErrorInvalidValue;
    if (a.ln2_g && (!a.ln2_b || !a.b2 || !a.y2_gran || !a.x2_out || !a.xb_out || a.x2_stride % 4 || a.xb_stride % 4)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((rows_mlp_kernel<HT>), dim3(256), dim3(512), 0, s, a);
    return hipGetLastError();
}

}  // namespace ma
