// Shared device helpers for the gfx950 kernels (wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ma {

constexpr int WAVE = 64;
constexpr int HEAD_DIM = 64;

typedef uint16_t bf16_t;   // raw bf16 bits

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// fp32 -> bf16 bits, round-to-nearest-even (NaN kept quiet).  Matches torch .to(bfloat16) and checkpoint.bf16_round.
// Device code takes gfx950's conversion instruction (v_cvt_pk_bf16_f32 behind the __bf16 cast): the same bits as the integer form below for
// every non-NaN input (all 2^32 patterns compared on the GPU: scripts/ubench_gemm256_epi.hip, profiles/r06_ubench_gemm256_epi.txt) at one
// instruction per PAIR instead of ~10 with a divergent NaN branch per element -- the integer form was half of the 256 x 256 GEMM's epilogue.
__host__ __device__ inline bf16_t f2bf(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(bf16_t, (__bf16)f);
#else
    union { float f; uint32_t u; } v; v.f = f;
    uint32_t u = v.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
#endif
}
__host__ __device__ inline float bf2f(bf16_t b) {
    union { float f; uint32_t u; } v; v.u = ((uint32_t)b) << 16; return v.f;
}
// fp32 -> nearest bf16 value, returned as fp32
__device__ inline float round_bf16(float f) { return bf2f(f2bf(f)); }

__host__ inline float half2float_host(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, u;
    if (exp == 0) {
        if (man == 0) u = sign;
        else { int e = -1; do { e++; man <<= 1; } while (!(man & 0x400u)); man &= 0x3ffu; u = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13); }
    } else if (exp == 31) u = sign | 0x7f800000u | (man << 13);
    else u = sign | ((exp + 112u) << 23) | (man << 13);
    union { float f; uint32_t u; } v; v.u = u; return v.f;
}

// two bf16 packed in one dword -> two floats (element 0 in the low half)
__device__ inline float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ inline float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// fp32 -> IEEE half bits, round-to-nearest-even, overflow -> inf (torch .to(float16)); host side of the weight packer
__host__ inline uint16_t float2half_host(float f) {
    union { float f; uint32_t u; } v; v.f = f;
    const uint32_t sign = (v.u >> 16) & 0x8000u, abs = v.u & 0x7fffffffu;
    if (abs > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                       // NaN
    if (abs >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                      // >= 65520: rounds to inf
    if (abs < 0x38800000u) {                                                        // subnormal half (|f| < 2^-14) or zero
        if (abs < 0x33000000u) return (uint16_t)sign;                               // < 2^-25: rounds to zero
        const int shift = 126 - (int)(abs >> 23);                                   // 14 .. 24
        const uint32_t man = (abs & 0x7fffffu) | 0x800000u;
        uint32_t r = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((abs - 0x38000000u) >> 13);                                        // rebias the exponent, keep 10 mantissa bits
    const uint32_t rem = abs & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;                          // a mantissa carry moves into the exponent: still right
    return (uint16_t)(sign | r);
}

// ---- the engine's two 16-bit formats -------------------------------------------------------------------------------------------
// Storage is always raw 16-bit words (`bf16_t` pointers: weights, KV cache, activation tensors); the FORMAT is a compile-time tag of the
// kernels that touch it.  bf16 (MA_DTYPE_BF16: BASELINE.json's policy) and IEEE fp16 (MA_DTYPE_F16: the reference's own arithmetic --
// Accelerator(mixed_precision="fp16") + autocast, main.py:114-118,149) share every kernel; H16<T> holds what differs: the conversions
// and the matrix-core instruction.  `f16_t` is the tag (and the element type where a kernel is templated on its weight / cache type).
struct f16_t { uint16_t v; };
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename T> struct H16;
template <> struct H16<bf16_t> {
    static constexpr bool is16 = true;
    static __device__ __forceinline__ float lo(uint32_t w) { return bf_lo(w); }
    static __device__ __forceinline__ float hi(uint32_t w) { return bf_hi(w); }
    static __device__ __forceinline__ float one(uint16_t b) { return bf2f(b); }
    static __device__ __forceinline__ uint16_t bits(float f) { return f2bf(f); }
    static __device__ __forceinline__ float round(float f) { return bf2f(f2bf(f)); }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
    }
    static __device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct H16<f16_t> {
    static constexpr bool is16 = true;
    static __device__ __forceinline__ float lo(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[0]; }
    static __device__ __forceinline__ float hi(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[1]; }
    static __device__ __forceinline__ float one(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
    static __device__ __forceinline__ uint16_t bits(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
    static __device__ __forceinline__ float round(float f) { return (float)(_Float16)f; }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) { return __builtin_bit_cast(uint32_t, f16x2_t{(_Float16)a, (_Float16)b}); }
    static __device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct H16<float> {       // the fp32 "exact" policy: nothing is rounded
    static constexpr bool is16 = false;
    static __device__ __forceinline__ float round(float f) { return f; }
};
// eight 16-bit elements (one 16-byte load) -> eight floats
template <typename HT>
__device__ __forceinline__ void unpack8(const u32x4& w, float* f) {
    f[0] = H16<HT>::lo(w.x); f[1] = H16<HT>::hi(w.x); f[2] = H16<HT>::lo(w.y); f[3] = H16<HT>::hi(w.y);
    f[4] = H16<HT>::lo(w.z); f[5] = H16<HT>::hi(w.z); f[6] = H16<HT>::lo(w.w); f[7] = H16<HT>::hi(w.w);
}
// four floats -> four 16-bit elements
template <typename HT>
__device__ __forceinline__ u32x2 pack4(const f32x4& v) { u32x2 p; p.x = H16<HT>::pack2(v.x, v.y); p.y = H16<HT>::pack2(v.z, v.w); return p; }

// ---- wave reductions on the DPP path (no LDS crossbar: __shfl_xor lowers to ds_bpermute_b32, ~100 cycles per step
// and six dependent steps per reduction; a DPP step is one VALU op).  quad_perm xor 1, xor 2, then row_half_mirror and
// row_mirror leave every lane of a 16-lane row with the row total; the four row totals are fetched with v_readlane.
template <int CTRL>
__device__ inline float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;
__device__ inline float readlane_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

__device__ inline float row_sum16(float v) {
    v += dpp_mov<DPP_XOR1>(v);
    v += dpp_mov<DPP_XOR2>(v);
    v += dpp_mov<DPP_HALF_MIRROR>(v);
    v += dpp_mov<DPP_ROW_MIRROR>(v);
    return v;
}
__device__ inline float wave_sum(float v) {
    v = row_sum16(v);
    return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
// max over each aligned group of 16 lanes (a DPP row); every lane of the row gets it
__device__ inline float row_max16(float v) {
    v = fmaxf(v, dpp_mov<DPP_XOR1>(v));
    v = fmaxf(v, dpp_mov<DPP_XOR2>(v));
    v = fmaxf(v, dpp_mov<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_mov<DPP_ROW_MIRROR>(v));
    return v;
}
__device__ inline float wave_max(float v) {
    v = fmaxf(v, dpp_mov<DPP_XOR1>(v));
    v = fmaxf(v, dpp_mov<DPP_XOR2>(v));
    v = fmaxf(v, dpp_mov<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_mov<DPP_ROW_MIRROR>(v));
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}
// sum over aligned groups of G lanes (G power of two <= 16); every lane of the group gets the sum
template <int G>
__device__ inline float group_sum(float v) {
    static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16, "group_sum: G must be a power of two <= 16");
    if constexpr (G >= 2) v += dpp_mov<DPP_XOR1>(v);
    if constexpr (G >= 4) v += dpp_mov<DPP_XOR2>(v);
    if constexpr (G >= 8) v += dpp_mov<DPP_HALF_MIRROR>(v);
    if constexpr (G >= 16) v += dpp_mov<DPP_ROW_MIRROR>(v);
    return v;
}

// ---- LayerNorm of one row, one barrier -------------------------------------------------------------------------------------
// Both moments are accumulated on the shifted data d = x - x0 (x0 = element 0 of the row, the same value in every thread), so
// var = E[d^2] - E[d]^2 does not cancel when the row has a large MEAN (x0 is then close to it).  It is not immune to x0 itself being
// an outlier: a shift that sits D standard deviations off the mean costs about D^2 * 2^-22 of relative accuracy in the variance
// (D = 30: 2e-4, below the bf16 rounding of the consumer; D = 100: 2e-3; tests/test_gpu_kernels.py::test_layernorm_prologue_with_an_
// outlier_in_dim_0 measures it).  The streams normalised here are post-LN (OPT-350m: do_layer_norm_before = False): LayerNorm output
// plus one sub-layer's update, so a 100-sigma element would have to come out of a single out_proj / fc2 row.  The three pieces below are shared by
// the block-level prologues (gemv.hpp, gemm_decode.hpp) and the persistent decode kernel (persist.hpp), written with explicit
// fmaf so that every path produces the same bits: chunk moments -> per-wave sums (wave_sum) -> (red[0]+red[1])+(red[2]+red[3]).
__device__ __forceinline__ void ln_chunk_moments(f32x4& x, float x0, float& s, float& q) {      // x <- x - x0; adds the chunk's sums
    x.x -= x0; x.y -= x0; x.z -= x0; x.w -= x0;
    s += (x.x + x.y) + (x.z + x.w);
    q += fmaf(x.x, x.x, x.y * x.y) + fmaf(x.z, x.z, x.w * x.w);
}
__device__ __forceinline__ void ln_finish(float s0, float s1, float s2, float s3, float q0, float q1, float q2, float q3, int K, float eps,
                                          float& md, float& rstd) {
    md = ((s0 + s1) + (s2 + s3)) / (float)K;                                                  // mean of the shifted row
    const float var = fmaxf(fmaf(-md, md, ((q0 + q1) + (q2 + q3)) / (float)K), 0.f);
    rstd = 1.0f / sqrtf(var + eps);
}
__device__ __forceinline__ float ln_apply1(float d, float md, float rstd, float g, float b) { return fmaf((d - md) * rstd, g, b); }
__device__ __forceinline__ void ln_apply(f32x4& x, float md, float rstd, const f32x4& g, const f32x4& b) {   // x holds d = x - x0
    x.x = ln_apply1(x.x, md, rstd, g.x, b.x); x.y = ln_apply1(x.y, md, rstd, g.y, b.y);
    x.z = ln_apply1(x.z, md, rstd, g.z, b.z); x.w = ln_apply1(x.w, md, rstd, g.w, b.w);
}
// block form: thread t owns the float4 chunks t, t + 256, ... of the row (chunks >= nq are padding).  In place:
// xv <- (x - mean) * rstd * g + b.  `red` = 8 floats of LDS.
template <int NCH>
__device__ inline void ln_block_onepass(f32x4 (&xv)[NCH], const f32x4 (&gv)[NCH], const f32x4 (&bv)[NCH], float x0, int tid, int nq, int K,
                                        float eps, float* red) {
    const int lane = tid & 63, w = tid >> 6;
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if (tid + 256 * j < nq) ln_chunk_moments(xv[j], x0, s, q);
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (lane == 0) { red[w] = s; red[4 + w] = q; }
    __syncthreads();
    float md, rstd;
    ln_finish(red[0], red[1], red[2], red[3], red[4], red[5], red[6], red[7], K, eps, md, rstd);
#pragma unroll
    for (int j = 0; j < NCH; ++j) ln_apply(xv[j], md, rstd, gv[j], bv[j]);
}

// erf in fp32, both branches evaluated and selected (no divergence): N. Juffa's two minimax polynomials, < 1 ulp over the whole range
// (checked against scipy's fp64 erf on 3 M points: max abs error 5.8e-8, 0.97 ulp).  The library erff costs several times as much:
// a GELU epilogue spent as long on it as the tile's whole K loop (profiles/r04_ubench_gemm256_gelu.txt).
__device__ __forceinline__ float erf_fast(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    const float big = copysignf(1.0f - __expf(r), a);      // |a| > 475/512
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    const float small = fmaf(q, a, a);
    return t > 0.927734375f ? big : small;
}
// exact (erf) GELU, as torch.nn.GELU() / HF "gelu"
__device__ inline float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };
__device__ inline float apply_act(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.0f);
    if (act == ACT_GELU) return gelu_erf(v);
    return v;
}

// 16-byte streaming load (weights / KV are read once per launch: keep them out of the way of the small hot vectors)
__device__ inline u32x4 ld_stream16(const void* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
}

// logical row r of a batched tensor -> physical row: groups of `grp` rows sit `gstride` rows apart, shifted by `off`
// (e.g. the 256 latent rows of sample b inside its 257-row block: grp 256, gstride 257, off 1).  grp == 0: identity.
struct RowMap {
    int grp, gstride, off;
    __host__ __device__ inline size_t operator()(int r) const { return grp > 0 ? (size_t)(r / grp) * gstride + (r % grp) + off : (size_t)r; }
};

// ---- in-launch exchange between resident workgroups (MI355X guide, Guideline 16 R2) -------------------------------------------
// A value travels as ONE naturally aligned 8-byte {epoch, payload} granule written with an agent-scope (sc1) store and polled with
// relaxed agent-scope loads: the data is the flag, so no fence, no separate flag word and no ordering between the two are needed.
// Used by the fused launches of the decode chain (qkv_attn.hpp, oproj_fc1.hpp, layer_fused.hpp, attn_decode.hpp's two-block form)
// and by the persistent step (persist.hpp).
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
constexpr u64 PS_TIMEOUT_TICKS = 20ull * 100000ull;            // 20 ms of the 100 MHz real-time counter: bound of every sweep

__device__ __forceinline__ void ps_publish(u64* gran, int idx, unsigned epoch, unsigned value) {
    __hip_atomic_store((gu64*)gran + idx, ((u64)epoch << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Bounded sweep: true when the polling wave must give up -- its own deadline passed, or some block of this generation already raised
// the engine's error word (then every later sweep gives up after at most 64 polls instead of burning its own 20 ms: a launch whose
// blocks are not all resident costs one deadline, not one per launch).  Checked every 64th poll only: nothing on the fast path.
// The deadline is 20 ms OF POLLING, not of the wall clock alone (round 6): a wave that the clock says has been sweeping for 20 ms but that has made
// fewer than PS_MIN_POLLS polls (a healthy sweep ends within ~100; a starved one makes 1024 in 2 - 5 ms) was not running in between -- the driver
// took the process's queues off the device and put them back (waves saved and restored together, so co-residency holds; seen as 20 - 30 ms holes
// in profiled runs since round 3, and as a lone "time-out" of ~50 waves -- the ones inside a sweep at that instant -- in a generation that had
// every block resident).  Such a sweep restarts its clock, counts itself in err[8] ("xchg_descheduled") and goes on; it can do so only while
// its poll count is below the minimum, so a sweep that really waits for a block that is not there still ends: after 1024 polls + 20 ms at most.
constexpr unsigned PS_MIN_POLLS = 1024u;
__device__ __forceinline__ bool xchg_expired(unsigned& spins, u64& t0, unsigned* err) {
    if ((++spins & 63u) != 0) return false;
    const u64 now = __builtin_amdgcn_s_memrealtime();
    if (now - t0 > PS_TIMEOUT_TICKS) {
        if (spins >= PS_MIN_POLLS) return true;
        t0 = now;
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) __hip_atomic_fetch_add(err + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// a sweep gave up: raise the bit in the engine's error word (err[0]; read and cleared by the host after the first step and after every
// burst) and count the event in err[1], which is never cleared -- the total shows in ma_engine_get_option("xchg_timeouts") and in the
// bench line, so a run that lost its fused launches for a while cannot look like a clean one
// The FIRST give-up since the counters were last read out also leaves who it was (round 6; the one fall-back of round 5 could not be placed):
// err[5] = 0x80000000 | code, err[6] = blockIdx.x | y << 8 | z << 16 | wave << 24, err[7] = polls made; ma_engine_get_option("xchg_first_giveup_*").
__device__ __forceinline__ void xchg_raise(unsigned* err, unsigned code, unsigned spins = 0) {
    __hip_atomic_fetch_or(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(err + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned expect = 0;
    if (__hip_atomic_compare_exchange_strong(err + 5, &expect, 0x80000000u | code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(err + 6, (blockIdx.x & 255u) | ((blockIdx.y & 255u) << 8) | ((blockIdx.z & 255u) << 16) | ((threadIdx.x >> 6) << 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(err + 7, spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// diagnostics of the fused launches (a 20-30 ms dispatch per profiled run that is NOT a sweep time-out): a sweep that
// needed more than 256 polls leaves its duration (100 MHz ticks since its start) in err[2] if it exceeded 1 ms and counts itself in err[3];
// ma_engine_get_option("slow_blocks" / "slow_block_max_us") and the bench line report them.  Nothing on the normal path: the poll count is
// in a register, the clock is read only behind the branch.
__device__ __forceinline__ void xchg_note_slow(unsigned* err, unsigned spins, u64 t0) {
    if (spins <= 256u) return;
    const u64 dt = __builtin_amdgcn_s_memrealtime() - t0;
    if (dt > 100000ull) {
        __hip_atomic_fetch_max(err + 2, (unsigned)(dt > 0xffffffffull ? 0xffffffffull : dt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(err + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Three 64-byte scalar loads of granules, past the scalar cache (glc).  The scalar unit has its own path to L2: a poll made this way does not
// queue behind the CU's vector-memory stream (scripts/ubench_poll_under_stream.hip: a sweep by vector loads ends 1.4 - 2.0 us later when the
// block's other waves have 64 - 112 KB of cache lines requested; by scalar loads from a wave that has no vector requests of its own
// outstanding, 0.1 - 0.5 us later; a wave that HAS stalls at their issue first).  A granule is one aligned 8-byte word written by one store:
// a 64-byte read returns each of its eight granules whole.  The pointers must be wave-uniform.
// (Batch 1, qkv_attn.hpp: measured slower than the vector sweep -- nothing streams under that exchange at short caches, and a scalar round
// trip is the longer one: profiles/r05_ab_qkv_attn_scalar_sweep_batch1.txt.)
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ const void* sgpr_ptr(const void* p) {          // (an "s" operand must BE in scalar registers: not left to the optimiser -- the -O1 / ASan build)
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const void*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void sload3x64_glc(const void* p0, const void* p1, const void* p2, u32x16& a, u32x16& b, u32x16& c) {
    p0 = sgpr_ptr(p0); p1 = sgpr_ptr(p1); p2 = sgpr_ptr(p2);
    asm volatile("s_load_dwordx16 %0, %3, 0x0 glc\n\ts_load_dwordx16 %1, %4, 0x0 glc\n\ts_load_dwordx16 %2, %5, 0x0 glc\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b), "=&s"(c) : "s"(p0), "s"(p1), "s"(p2) : "memory");
}

// better-argmax: larger value wins, ties -> lower index (torch.argmax semantics)
__device__ inline bool arg_better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

}  // namespace ma
