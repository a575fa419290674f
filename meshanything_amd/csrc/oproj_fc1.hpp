// Second half of a decoder layer in one launch, batch-1 launch chain (any policy: templated on the storage format; hidden 1024, ffn 4096): merge of the split-KV
// partials + out_proj + residual, LayerNorm 1, fc1 + ReLU and (template FC2, the default) fc2 + residual.
//
// Replaces three launches of the five-launch chain ([3p] OPTDecoderLayer: out_proj + residual, self_attn_layer_norm, fc1 + ReLU, fc2 +
// residual; reached from shape_opt.py:403-410): all run on 256 blocks (out_proj and fc2: 4 rows per block, fc1: 16 rows per block), so
// block b of the fused launch does exactly the work of block b of each.  The dependencies between them -- every fc1 row needs all
// 1024 values of y1 = h + Wo a + bo, every fc2 row all 4096 values of relu(fc1) -- are all-gathers done inside the launch with
// tagged granules (MI355X guide, Guideline 16 R2) instead of a kernel boundary + launch ramp + a dependent reload of the vector:
//   * y1: one 8-byte {epoch, fp32} granule per wave out, the four waves sweep a quarter of the 1024 each;
//   * relu(fc1): fc2 consumes it rounded to bf16, so a granule carries two values ({epoch, bf16, bf16}): 2048 granules, a quarter
//     per wave.
// What such an exchange costs is its polling (profiles/r02_ab_exchange_and_load_placement.txt): one wave sweeping all of y1 took
// 2.2 us, four waves a quarter each 4 % of the step less; relu(fc1) with one value per granule 4 % more than with two.
// The attention partials, the out_proj row, the 4 fc1 rows (32 KB per block), biases and LayerNorm parameters are requested in the first
// instructions, the fc2 row under the first exchange.  The arithmetic is gemv_kernel's ({1, 2, 1}, {1, 2, 4}, {1, 8, 1} shapes), bit
// for bit (tests/test_gpu_persist.py).
// Epoch = position * 32 + layer + 1 (buffers zeroed when the position restarts); sweeps are bounded (20 ms) and raise the engine's
// error word instead of hanging; all 256 blocks of a batch row must be resident together (checked at engine creation).
// The body is a device function so that layer_fused.hpp can continue with the next layer's first half in the same launch.
#pragma once
#include "attn_decode.hpp"
#include "common.hpp"
#include "gemv.hpp"
#include "state.hpp"

namespace ma {

// pieces of 16 bytes per lane and 1024 elements of a weight row, and how gemv_kernel sums them, per storage format: the 16-bit policies run
// whole rows in one wave ({1, 2, *} and {1, 8, 1} shapes: ONE sum per row); the fp32 policy (round 6) splits K between waves ({2, 2, *} at
// K = 1024, {4, 4, 1} at K = 4096) and adds the per-wave sums in ascending order -- here one wave keeps the SAME partial sums apart.  In the
// fp32 policy the structs' bf16_t pointers are plain addresses of fp32 storage.
template <typename HT> struct OfStore { typedef HT T; static constexpr int VEC = 8, NP = 2, PARTS1 = 1, PARTS2 = 1; };
template <> struct OfStore<float> { typedef float T; static constexpr int VEC = 4, NP = 4, PARTS1 = 2, PARTS2 = 4; };
// sum of PARTS per-wave partials the way gemv_kernel's finishing lane adds them (v = 0; v += red[0]; v += red[1]; ...)
template <int PARTS>
__device__ __forceinline__ float of_sum_parts(const float (&acc)[PARTS]) {
    if constexpr (PARTS == 1) return wave_sum(acc[0]);
    else {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < PARTS; ++k) v += wave_sum(acc[k]);
        return v;
    }
}

struct OprojFc1Args {
    const bf16_t* Wo; const float* bo;                   // [hidden][hidden], [hidden]
    const bf16_t* W1; const float* b1;                   // [ffn][hidden], [ffn]
    const float* ln_g; const float* ln_b; float ln_eps;  // self_attn_layer_norm
    const float* attn_ws; int heads;                     // split-KV partials of this layer (attn_workspace_floats per batch row)
    const float* res;                                    // the layer input h (fp32, [hidden]): out_proj's residual
    float* h1_out;                                       // LN1(y1) fp32: fc2's residual
    float* ffn_out;                                      // relu(fc1) fp32 [ffn]
    const DecState* st; int layer;
    u64* gran;                                           // [batch][hidden] granules
    unsigned* err;
    int res_stride, h1_stride, ffn_stride;               // per batch row
    unsigned long long* trace;
    int sweep_waves;                                     // waves per block that poll the y1 granules: 1 | 2 | 4
    // fc2 in the same launch (template FC2): relu(fc1) is all-gathered too (4096 granules, a quarter per wave), then row 4b + w of fc2
    const bf16_t* W2; const float* b2; float* y2_out; int y2_stride; u64* gran2;
};
constexpr unsigned OF_ERR_GATHER = 32;

// NEXT: instead of storing y2, all-gather it (1024 granules in `gran3`, a quarter per wave) into `ynext` (LDS, 1024 floats): the next
// layer's q/k/v + attention continues in the same launch (layer_fused.hpp).  Arguments by value: a by-reference kernel-argument
// struct can end up in scratch.
// HT: the 16-bit format of the weights (bf16_t | f16_t, common.hpp H16)
template <int NSW, bool FC2, bool NEXT, typename HT, typename Hook>
__device__ __forceinline__ void oproj_fc1_body(OprojFc1Args a, const int b, const int brow, float* ynext, u64* gran3, Hook&& after_ffn_publish) {
    constexpr int KC = 1024, KF = 4096;
    typedef typename OfStore<HT>::T ST;
    constexpr int VEC = OfStore<HT>::VEC, NP = OfStore<HT>::NP, NP2 = 4 * NP, P1 = OfStore<HT>::PARTS1, P2 = OfStore<HT>::PARTS2;
    const ST* const Wo = reinterpret_cast<const ST*>(a.Wo);
    const ST* const W1 = reinterpret_cast<const ST*>(a.W1);
    const ST* const W2 = reinterpret_cast<const ST*>(a.W2);
    __shared__ __attribute__((aligned(16))) float ffl[FC2 ? KF : 4];      // relu(fc1), rounded to the weights' format (fc2's input)
    __shared__ float h1l[FC2 ? 4 : 1];                                    // LN1(y1)[4b + w]: fc2's residual
    __shared__ __attribute__((aligned(16))) float xl[KC];
    __shared__ __attribute__((aligned(16))) float yraw[KC];
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (a.trace && tid == 0) a.trace[b * 4 + 0] = __builtin_amdgcn_s_memrealtime();
    const float* ws = a.attn_ws + (size_t)brow * attn_workspace_floats(a.heads);
    const unsigned epoch = (unsigned)a.st[brow].pos * 32u + (unsigned)a.layer + 1u;

    // ---- (1) every load that does not depend on the exchange ---------------------------------------------------------------------
    f32x4 pml[ATTN_NCHUNK / 2], po[ATTN_NCHUNK];
    {
        const int k = tid * 4;
        attn_partials_load(ws, a.heads, k >> 6, k & 63, pml, po);
    }
    // fp16 instantiation: the 24 requests of the partials go out together, ahead of everything else -- hipcc interleaved them pairwise with
    // the merge arithmetic there (a wait per pair: +1.3 us per launch, 10.28 -> 9.58 us with this pin).  The bf16 instantiation keeps the
    // schedule the compiler gives it (measured best since round 2); forcing ALL requests, weights included, in front of the merge was 9 %
    // slower in both formats (profiles/r04_ab_load_pinning.txt).
    if constexpr (__is_same(HT, f16_t)) __builtin_amdgcn_sched_barrier(0);
    const int orow = 4 * b + w;                          // out_proj row of this wave
    u32x4 wo[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) wo[i] = ld_stream16(Wo + (size_t)orow * KC + (i * 64 + lane) * VEC);
    const float e_bo = a.bo[orow], e_res = a.res[(size_t)brow * a.res_stride + orow];
    u32x4 w1[4][NP];
    float e_b1[4];
    f32x4 gv[1], bv[1];
    auto load_fc1 = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NP; ++i) w1[j][i] = ld_stream16(W1 + (size_t)(16 * b + 4 * w + j) * KC + (i * 64 + lane) * VEC);
#pragma unroll
        for (int j = 0; j < 4; ++j) e_b1[j] = a.b1[16 * b + 4 * w + j];
        gv[0] = *reinterpret_cast<const f32x4*>(a.ln_g + tid * 4);
        bv[0] = *reinterpret_cast<const f32x4*>(a.ln_b + tid * 4);
    };
    load_fc1();              // in the first instructions: requesting the fc1 rows under the exchange instead is 2 % slower (profiles/r02_ab_exchange_and_load_placement.txt)
    asm volatile("" ::: "memory");

    // ---- (2) out_proj: gemv_kernel<bf16_t, 1, 2, 1, PRO_ATTN> ------------------------------------------------------------------------
    {
        f32x4 r = attn_partials_merge(pml, po);
        r.x = H16<HT>::round(r.x); r.y = H16<HT>::round(r.y); r.z = H16<HT>::round(r.z); r.w = H16<HT>::round(r.w);
        *reinterpret_cast<f32x4*>(&xl[tid * 4]) = r;
    }
    __syncthreads();
    float y1;
    {
        float acc[P1];
#pragma unroll
        for (int k = 0; k < P1; ++k) acc[k] = 0.f;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int k0 = (i * 64 + lane) * VEC;
            float xs[VEC], wf[VEC];
#pragma unroll
            for (int v = 0; v < VEC; v += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(&xl[k0 + v]);
                xs[v] = t.x; xs[v + 1] = t.y; xs[v + 2] = t.z; xs[v + 3] = t.w;
            }
            unpack16<ST>(wo[i], wf);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[i * P1 / NP] = fmaf(wf[v], xs[v], acc[i * P1 / NP]);
        }
        float v = of_sum_parts<P1>(acc);
        v += e_bo;
        v += e_res;
        y1 = v;
    }
    if (a.trace && tid == 0) a.trace[b * 4 + 1] = __builtin_amdgcn_s_memrealtime();

    // ---- (3) all-gather of y1: one granule per wave out, all 1024 in --------------------------------------------------
    u64* gran = a.gran + (size_t)brow * KC;
    if (lane == 0) ps_publish(gran, orow, epoch, __float_as_uint(y1));
    u32x4 w2[FC2 ? NP2 : 1];
    float e_b2 = 0.f;
    if constexpr (FC2) {                                 // fc2's row: not needed before the second exchange, requested under the first
#pragma unroll
        for (int i = 0; i < NP2; ++i) w2[i] = ld_stream16(W2 + (size_t)orow * KF + (i * 64 + lane) * VEC);
        e_b2 = a.b2[orow];
        asm volatile("" ::: "memory");
    }
    // NSW waves per block sweep, each its own 1024 / NSW granules (16 / NSW per lane and pass).  The polling traffic is what costs (MI355X
    // guide): four waves that each polled ALL granules lost 1.2 %; four waves polling a quarter each win 4 % over one wave polling all
    // (profiles/r02_ab_exchange_and_load_placement.txt)
    if (w < NSW) {
        constexpr int NK = 16 / NSW;
        const gu64* g64 = (const gu64*)gran + w * NK * 64;
        float* yr = yraw + w * NK * 64;
        u64 t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0, pend = (1u << NK) - 1u;
        for (;;) {
            u64 v[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                v[k] = (u64)epoch << 32;
                if ((pend >> k) & 1u) v[k] = __hip_atomic_load(g64 + k * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                if ((pend >> k) & 1u) {
                    const bool ok = (unsigned)(v[k] >> 32) == epoch;
                    if (ok) yr[k * 64 + lane] = __uint_as_float((unsigned)v[k]);
                    if (__all(ok)) pend &= ~(1u << k);
                }
            }
            if (!pend) break;
            __builtin_amdgcn_s_sleep(1);
            if (xchg_expired(spins, t0, a.err)) {
                if (lane == 0) xchg_raise(a.err, OF_ERR_GATHER, spins);
#pragma unroll
                for (int k = 0; k < NK; ++k) yr[k * 64 + lane] = 0.f;
                break;
            }
        }
        if (lane == 0) xchg_note_slow(a.err, spins, t0);
    }
    __syncthreads();
    if (a.trace && tid == 0) a.trace[b * 4 + 2] = __builtin_amdgcn_s_memrealtime();

    // ---- (4) fc1: gemv_kernel<bf16_t, 1, 2, 4, PRO_LN> on the gathered row ---------------------------------------------------------------
    {
        f32x4 xv[1];
        xv[0] = *reinterpret_cast<const f32x4*>(&yraw[tid * 4]);
        const float x0 = yraw[0];
        ln_block_onepass<1>(xv, gv, bv, x0, tid, KC / 4, KC, a.ln_eps, red);
        if constexpr (FC2) { if (tid == b) { h1l[0] = xv[0].x; h1l[1] = xv[0].y; h1l[2] = xv[0].z; h1l[3] = xv[0].w; } }   // elements 4b .. 4b + 3
        else if (b == 0) *reinterpret_cast<f32x4*>(a.h1_out + (size_t)brow * a.h1_stride + tid * 4) = xv[0];
        f32x4 r = xv[0];
        r.x = H16<HT>::round(r.x); r.y = H16<HT>::round(r.y); r.z = H16<HT>::round(r.z); r.w = H16<HT>::round(r.w);
        *reinterpret_cast<f32x4*>(&xl[tid * 4]) = r;
    }
    __syncthreads();
    float acc4[4][P1];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < P1; ++k) acc4[j][k] = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int k0 = (i * 64 + lane) * VEC;
        float xs[VEC];
#pragma unroll
        for (int v = 0; v < VEC; v += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(&xl[k0 + v]);
            xs[v] = t.x; xs[v + 1] = t.y; xs[v + 2] = t.z; xs[v + 3] = t.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float wf[VEC];
            unpack16<ST>(w1[j][i], wf);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc4[j][i * P1 / NP] = fmaf(wf[v], xs[v], acc4[j][i * P1 / NP]);
        }
    }
    float outv = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = of_sum_parts<P1>(acc4[j]);
        v += e_b1[j];
        v = fmaxf(v, 0.0f);
        if (lane == j) outv = v;
    }
    if constexpr (!FC2) {
        if (lane < 4) a.ffn_out[(size_t)brow * a.ffn_stride + 16 * b + 4 * w + lane] = outv;
        if (a.trace && tid == 0) a.trace[b * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    } else {
        // ---- (5) all-gather of relu(fc1).  fc2 consumes it rounded to bf16, so a granule carries TWO values (bf16 pair + epoch): 2 granules
        //      per wave out, 2048 in all, wave w sweeps granules [512 w, 512 w + 512) -- half the polling of one value per granule ------------
        //      fp32 policy: the value is consumed unrounded, one fp32 per granule (4096 granules, 1024 per wave in one pass of sixteen loads per lane)
        u64* g2 = a.gran2 + (size_t)brow * KF;
        if constexpr (sizeof(ST) == 4) {
            if (lane < 4) ps_publish(g2, 16 * b + 4 * w + lane, epoch, __float_as_uint(outv));
            after_ffn_publish();
            {   // (ONE pass of sixteen loads per lane: two passes of eight cost a second round trip, +0.5 us per launch)
                const gu64* g64 = (const gu64*)g2 + w * 1024;
                float* fr = ffl + w * 1024;
                u64 t0 = __builtin_amdgcn_s_memrealtime();
                unsigned spins = 0, pend = 0xffffu;
                for (;;) {
                    u64 v[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        v[k] = (u64)epoch << 32;
                        if ((pend >> k) & 1u) v[k] = __hip_atomic_load(g64 + k * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        if ((pend >> k) & 1u) {
                            const bool ok = (unsigned)(v[k] >> 32) == epoch;
                            if (ok) fr[k * 64 + lane] = __uint_as_float((unsigned)v[k]);
                            if (__all(ok)) pend &= ~(1u << k);
                        }
                    }
                    if (!pend) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (xchg_expired(spins, t0, a.err)) {
                        if (lane == 0) xchg_raise(a.err, OF_ERR_GATHER, spins);
#pragma unroll
                        for (int k = 0; k < 16; ++k) fr[k * 64 + lane] = 0.f;
                        break;
                    }
                }
                if (lane == 0) xchg_note_slow(a.err, spins, t0);
            }
        } else {
            const float nb = __shfl_down(outv, 1, 64);
            if (lane == 0 || lane == 2) ps_publish(g2, 8 * b + 2 * w + (lane >> 1), epoch, H16<HT>::pack2(outv, nb));
        after_ffn_publish();                             // the caller's next requests ride under this exchange
        {
            const gu64* g64 = (const gu64*)g2 + w * 512;
            float* fr = ffl + w * 1024;
            u64 t0 = __builtin_amdgcn_s_memrealtime();
            unsigned spins = 0, pend = 0xffu;
            for (;;) {
                u64 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v[k] = (u64)epoch << 32;
                    if ((pend >> k) & 1u) v[k] = __hip_atomic_load(g64 + k * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if ((pend >> k) & 1u) {
                        const bool ok = (unsigned)(v[k] >> 32) == epoch;
                        if (ok) { const unsigned pr = (unsigned)v[k]; fr[2 * (k * 64 + lane)] = H16<HT>::lo(pr); fr[2 * (k * 64 + lane) + 1] = H16<HT>::hi(pr); }
                        if (__all(ok)) pend &= ~(1u << k);
                    }
                }
                if (!pend) break;
                __builtin_amdgcn_s_sleep(1);
                if (xchg_expired(spins, t0, a.err)) {
                    if (lane == 0) xchg_raise(a.err, OF_ERR_GATHER, spins);
#pragma unroll
                    for (int k = 0; k < 8; ++k) { fr[2 * (k * 64 + lane)] = 0.f; fr[2 * (k * 64 + lane) + 1] = 0.f; }
                    break;
                }
            }
            if (lane == 0) xchg_note_slow(a.err, spins, t0);
        }
        }
        __syncthreads();
        if (a.trace && tid == 0) a.trace[b * 4 + 3] = __builtin_amdgcn_s_memrealtime();
        // ---- (6) fc2: gemv_kernel<bf16_t, 1, 8, 1, PRO_PLAIN> (fp32: <float, 4, 4, 1, PRO_PLAIN>), row 4b + w -------------------------------
        float acc[P2];
#pragma unroll
        for (int k = 0; k < P2; ++k) acc[k] = 0.f;
#pragma unroll
        for (int i = 0; i < NP2; ++i) {
            const int k0 = (i * 64 + lane) * VEC;
            float xs[VEC], wf[VEC];
#pragma unroll
            for (int v = 0; v < VEC; v += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(&ffl[k0 + v]);
                xs[v] = t.x; xs[v + 1] = t.y; xs[v + 2] = t.z; xs[v + 3] = t.w;
            }
            unpack16<ST>(w2[i], wf);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[i * P2 / NP2] = fmaf(wf[v], xs[v], acc[i * P2 / NP2]);
        }
        float v = of_sum_parts<P2>(acc);
        v += e_b2;
        v += h1l[w];
        if constexpr (!NEXT) {
            if (lane == 0) a.y2_out[(size_t)brow * a.y2_stride + orow] = v;
        } else {
            // ---- (7) all-gather of y2 for the next layer's LayerNorm: one granule per wave out, a quarter of the 1024 per wave in ----------
            u64* g3 = gran3 + (size_t)brow * KC;
            if (lane == 0) ps_publish(g3, orow, epoch, __float_as_uint(v));
            const gu64* g64 = (const gu64*)g3 + w * 256;
            float* yr = ynext + w * 256;
            u64 t0 = __builtin_amdgcn_s_memrealtime();
            unsigned spins = 0, pend = 0xfu;
            for (;;) {
                u64 q[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    q[k] = (u64)epoch << 32;
                    if ((pend >> k) & 1u) q[k] = __hip_atomic_load(g64 + k * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if ((pend >> k) & 1u) {
                        const bool ok = (unsigned)(q[k] >> 32) == epoch;
                        if (ok) yr[k * 64 + lane] = __uint_as_float((unsigned)q[k]);
                        if (__all(ok)) pend &= ~(1u << k);
                    }
                }
                if (!pend) break;
                __builtin_amdgcn_s_sleep(1);
                if (xchg_expired(spins, t0, a.err)) {
                    if (lane == 0) xchg_raise(a.err, OF_ERR_GATHER, spins);
#pragma unroll
                    for (int k = 0; k < 4; ++k) yr[k * 64 + lane] = 0.f;
                    break;
                }
            }
            __syncthreads();
        }
    }
}

template <int NSW, bool FC2, typename HT = bf16_t>
__global__ __launch_bounds__(256) void oproj_fc1_kernel(OprojFc1Args a) {
    oproj_fc1_body<NSW, FC2, false, HT>(a, blockIdx.x, blockIdx.y, nullptr, nullptr, [] {});
}

template <typename HT>
inline hipError_t launch_oproj_fc1(const OprojFc1Args& a, int hidden, int ffn, int batch, hipStream_t s) {
    if (hidden != 1024 || ffn != 4096 || a.heads * 64 != hidden) return hipErrorInvalidValue;
    if (a.W2) hipLaunchKernelGGL((oproj_fc1_kernel<4, true, HT>), dim3(hidden / 4, batch), dim3(256), 0, s, a);
    else if (a.sweep_waves == 2) hipLaunchKernelGGL((oproj_fc1_kernel<2, false, HT>), dim3(hidden / 4, batch), dim3(256), 0, s, a);
    else if (a.sweep_waves == 4) hipLaunchKernelGGL((oproj_fc1_kernel<4, false, HT>), dim3(hidden / 4, batch), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((oproj_fc1_kernel<1, false, HT>), dim3(hidden / 4, batch), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace ma
