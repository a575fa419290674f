// Token post-processing, coordinate argmax and the token pick / sampler (the batched row-wise kernels of the dense phases
// live in dense_ops.hpp).  Each cites the reference lines it implements.
#pragma once
#include "common.hpp"
#include "state.hpp"

namespace ma {

constexpr int TOK_BOS = 0, TOK_EOS = 1, TOK_PAD = 2;      // meshanything.py:102-104

// meshanything.py:141-142,163-172: eos-pad the generated tokens to 9F+2, drop first and last, specials -> -1, others -= 3
__global__ void postprocess_tokens_kernel(const long long* __restrict__ tokens, int ld_tokens, int n_generated, int max_new,
                                          long long* __restrict__ ids, int B) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int L = max_new - 2;
    if (idx >= B * L) return;
    const int b = idx / L, j = idx - b * L;
    const int src = j + 1;                                      // outputs[:, 1:-1]
    long long t = src < n_generated ? tokens[(size_t)b * ld_tokens + src] : (long long)TOK_EOS;
    ids[idx] = (t == TOK_BOS || t == TOK_EOS || t == TOK_PAD) ? -1 : t - 3;
}

// meshanything.py:69-78 + undiscretize (214-223): argmax over the discrete bins (lowest index wins ties),
// coord = idx / num_discrete * (0.5 - -0.5) + -0.5, NaN for masked faces.  One wave per (face, coordinate).
__global__ __launch_bounds__(256) void coords_argmax_kernel(const float* __restrict__ logits, int nf, int nd,
                                                            const unsigned char* __restrict__ mask, float* __restrict__ coords) {
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (item >= nf * 9) return;
    const float* lp = logits + (size_t)item * nd;
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int k = lane; k < nd; k += 64) { const float v = lp[k]; if (arg_better(v, k, bv, bi)) { bv = v; bi = k; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (arg_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) {
        const int f = item / 9;
        float t = (float)bi;
        t = t / (float)nd;
        t = t * (0.5f - (-0.5f)) + (-0.5f);
        coords[item] = mask[f] ? t : __builtin_nanf("");
    }
}

// ---- token pick: greedy argmax or top-k -> top-p -> inverse-CDF draw ([3p] GenerationMixin greedy / sample with
// TopKLogitsWarper + TopPLogitsWarper, call site meshanything.py:143-162), plus the generate() bookkeeping:
// a finished row emits pad, eos marks the row finished, the token becomes next step's input.
__device__ inline float hash_uniform(unsigned long long seed, int row, int t) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(((unsigned long long)row << 32) | (unsigned)t);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

constexpr int PICK_KMAX = 64;

// grid = batch rows: block b owns row b's logits, partials, state record and output row (nothing is shared between blocks).
__global__ __launch_bounds__(256) void pick_kernel(const float* __restrict__ logits, int V, const float* __restrict__ part_val,
                                                   const int* __restrict__ part_idx, int nparts, int part_stride, DecState* st,
                                                   long long* __restrict__ tokens_out, int tokens_stride, int T) {
    logits += (size_t)blockIdx.x * V;
    part_val += (size_t)blockIdx.x * part_stride;
    part_idx += (size_t)blockIdx.x * part_stride;
    st += blockIdx.x;
    tokens_out += (size_t)blockIdx.x * tokens_stride;
    extern __shared__ __attribute__((aligned(16))) float dyn[];      // V floats (sampling only)
    __shared__ float rv[4]; __shared__ int ri[4];
    __shared__ float cv[PICK_KMAX]; __shared__ int ci[PICK_KMAX];
    __shared__ int chosen;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // the state record and the argmax partials are requested together (the partials do not depend on the state: a sampled step
    // just drops them) -- one memory round trip instead of two in front of the greedy reduction
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int i = tid; i < nparts; i += 256) { const float v = part_val[i]; const int ix = part_idx[i]; if (arg_better(v, ix, bv, bi)) { bv = v; bi = ix; } }
    // batched MFMA lm_head (no partials): the first 8192 logits of the row are requested here as well, 32 per thread in one go (clamped
    // index, masked at the use) -- the greedy sweep below was five dependent round trips of 8 loads each
    constexpr int PICK_PRE = 32;
    float pre[PICK_PRE];
    const bool preloaded = nparts == 0;
#pragma unroll
    for (int j = 0; j < PICK_PRE; ++j) pre[j] = logits[preloaded ? min(tid + 256 * j, V - 1) : 0];
    const DecState sv = *st;
    const int do_sample = sv.do_sample;
    if (!do_sample) {
        if (nparts > 0) {            // per-block partials of the lm_head GEMV (eos already excluded there when suppressed): reduced above
        } else {                     // batched MFMA lm_head: plain logits
            const int skip = sv.suppress_eos ? TOK_EOS : -1;
#pragma unroll
            for (int j = 0; j < PICK_PRE; ++j) { const int i = tid + 256 * j; if (i < V && i != skip && arg_better(pre[j], i, bv, bi)) { bv = pre[j]; bi = i; } }
            for (int i = tid + 256 * PICK_PRE; i < V; i += 256) { const float v = logits[i]; if (i != skip && arg_better(v, i, bv, bi)) { bv = v; bi = i; } }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (arg_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { rv[w] = bv; ri[w] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int i = 1; i < 4; ++i) if (arg_better(rv[i], ri[i], bv, bi)) { bv = rv[i]; bi = ri[i]; }
            chosen = bi;
        }
    } else {
        // top-k by radix selection (4 passes of an 8-bit histogram over order-preserving integer keys) instead of k argmax
        // sweeps: ~10 us instead of ~190 us for k = 50 over 8195 logits.  Keeps every score >= the k-th largest
        // (TopKLogitsWarper semantics: ties at the threshold stay), up to PICK_KMAX candidates.
        __shared__ unsigned hist[16 * 256], wtot[4];      // 16 privatised copies: logits cluster in a few bins (same-address LDS atomics serialise)
        __shared__ unsigned sel_prefix, sel_krem;
        __shared__ int ncand;
        const int k = min(min(st->top_k, V), PICK_KMAX);
        for (int i = tid; i < V; i += 256) dyn[i] = (st->suppress_eos && i == TOK_EOS) ? -INFINITY : logits[i];
        if (tid == 0) { sel_prefix = 0u; sel_krem = (unsigned)k; ncand = 0; }
        __syncthreads();
        auto key_of = [](float f) -> unsigned { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
        for (int shift = 24; shift >= 0; shift -= 8) {
#pragma unroll
            for (int c = 0; c < 16; ++c) hist[c * 256 + tid] = 0u;
            __syncthreads();
            const unsigned prefix = sel_prefix;
            const unsigned himask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
            for (int i = tid; i < V; i += 256) {
                const unsigned kk = key_of(dyn[i]);
                if ((kk & himask) == (prefix & himask)) atomicAdd(&hist[(lane & 15) * 256 + ((kk >> shift) & 0xffu)], 1u);
            }
            __syncthreads();
            {   // suffix[t] = number of keys in bins >= t (parallel: in-wave shuffle scan, then the waves' totals through LDS);
                // the selected bin is the largest t with suffix[t] >= k_rem
                unsigned v = 0u;
#pragma unroll
                for (int c = 0; c < 16; ++c) v += hist[c * 256 + tid];
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_down(v, o, 64); if (lane + o < 64) v += up; }
                if (lane == 0) wtot[w] = v;
                __syncthreads();
                for (int ww = w + 1; ww < 4; ++ww) v += wtot[ww];
                const unsigned next = __shfl_down(v, 1, 64);                   // suffix[t + 1] inside the wave
                unsigned above = 0u;                                            // suffix[t + 1]: keys in strictly higher bins
                if (lane < 63) above = next; else for (int ww = w + 1; ww < 4; ++ww) above += wtot[ww];
                const unsigned krem = sel_krem;
                __syncthreads();                                                // everyone has read sel_krem / wtot
                if (v >= krem && above < krem) { sel_krem = krem - above; sel_prefix = prefix | ((unsigned)tid << shift); }
            }
            __syncthreads();
        }
        const unsigned thr = sel_prefix;                  // key of the k-th largest score
        for (int i = tid; i < V; i += 256) {
            if (key_of(dyn[i]) >= thr) {
                const int slot = atomicAdd(&ncand, 1);
                if (slot < PICK_KMAX) { cv[slot] = dyn[i]; ci[slot] = i; }
            }
        }
        __syncthreads();
        const int nc = min(ncand, PICK_KMAX);
        // order the candidates: descending score, ties by ascending index (rank by counting, one wave)
        __shared__ float sv[PICK_KMAX]; __shared__ int si[PICK_KMAX];
        if (tid < PICK_KMAX) {
            if (tid < nc) {
                const float v = cv[tid]; const int ix = ci[tid];
                int rank = 0;
                for (int j = 0; j < nc; ++j) if (arg_better(cv[j], ci[j], v, ix)) ++rank;
                sv[rank] = v; si[rank] = ix;
            }
        }
        __syncthreads();
        __shared__ float se[PICK_KMAX];
        if (tid < PICK_KMAX) se[tid] = tid < nc ? expf(sv[tid] - sv[0]) : 0.f;
        __syncthreads();
        if (tid == 0) {
            // candidates are in descending order.  top-p: drop the ascending prefix whose cumulative mass <= 1 - top_p
            // (the exponentials stay in LDS: a runtime-indexed local copy would live in scratch, guide rule 20)
            const int kk = nc;
            float sum = 0.f;
            for (int j = 0; j < kk; ++j) sum += se[j];
            const float thr_p = (float)(1.0 - (double)st->top_p);
            int keep = kk;
            float cum = 0.f;
            for (int j = kk - 1; j >= 1; --j) { cum += se[j] / sum; if (cum <= thr_p) keep = j; else break; }
            float sum2 = 0.f;
            for (int j = 0; j < keep; ++j) sum2 += se[j];
            const int t = st->t;
            const float u = st->uniforms ? st->uniforms[t] : hash_uniform(st->seed, st->row, t);
            int pick = keep - 1;
            float acc = 0.f;
            for (int j = 0; j < keep; ++j) { acc += se[j] / sum2; if (acc > u) { pick = j; break; } }
            chosen = si[pick];
        }
    }
    __syncthreads();
    // parity aid (ma_sample_cfg.logits_out): the distribution token t was picked from, kept for every step
    if (sv.logits_out && sv.t < sv.max_new && sv.t >= sv.logits_first) {
        float* lo = sv.logits_out + (size_t)(sv.t - sv.logits_first) * V;
        for (int i = tid; i < V; i += 256) lo[i] = logits[i];
    }
    if (tid == 0) {
        const int t = sv.t;
        int tok = chosen;
        if (sv.finished) tok = TOK_PAD;
        if (t < sv.max_new) tokens_out[t] = tok;
        // teacher forcing (ma_sample_cfg.forced_tokens): the pick is reported, the given token is fed
        // (an id outside [0, V) would index the codebook / extra-embed tables out of bounds in the next embedding launch: clamped here;
        //  Engine.generate refuses such a buffer on the host side)
        if (sv.forced && t < sv.max_new) { const long long f = sv.forced[t]; tok = (int)(f < 0 ? 0 : f >= V ? V - 1 : f); }
        if (tok == TOK_EOS) st->finished = 1;
        st->cur_tok = tok;
        st->t = t + 1;
        st->pos = T + t;          // next step feeds token t at cache row cond_length + (t+1) - 1
    }
}

// one state record per batch row; rows differ in `row` and in their slice of the injected uniforms
__global__ void init_state_kernel(DecState* st, DecState v, int B, int V) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    v.row += b;
    if (v.uniforms) v.uniforms += (size_t)b * v.max_new;
    if (v.forced) v.forced += (size_t)b * v.max_new;
    if (v.logits_out) v.logits_out += (size_t)b * (v.max_new - v.logits_first) * V;
    st[b] = v;
}
// used by stepwise prefill / profiling: set the fields one decode step reads (all rows)
__global__ void set_pos_kernel(DecState* st, int t, int pos, int cur_tok, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    st[b].t = t; st[b].pos = pos; st[b].cur_tok = cur_tok;
}
__global__ void fill_tokens_kernel(long long* p, long long v, int n) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) p[idx] = v;
}

// checkpoint tensor (fp32 | bf16 | fp16 as stored, rows x src_ld) -> its arena slot (fp32 or bf16, leading dimension dst_ld): the device
// half of ma_engine_load_weights.  Same rounding as the host packer (f2bf: round to nearest even).
// dst_dtype: MA_DTYPE_F32 | MA_DTYPE_BF16 | MA_DTYPE_F16 (the arena entry's type)
__global__ void cvt_weight_kernel(const void* __restrict__ src, int src_dtype, int src_ld, void* __restrict__ dst, int dst_dtype, int dst_ld, int rows, int cols) {
    const size_t total = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / cols, k = i - r * cols, si = r * src_ld + k;
        float v;
        if (src_dtype == MA_DTYPE_F32) v = reinterpret_cast<const float*>(src)[si];
        else if (src_dtype == MA_DTYPE_BF16) v = bf2f(reinterpret_cast<const uint16_t*>(src)[si]);
        else v = (float)reinterpret_cast<const _Float16*>(src)[si];
        if (dst_dtype == MA_DTYPE_F32) reinterpret_cast<float*>(dst)[r * dst_ld + k] = v;
        else if (dst_dtype == MA_DTYPE_F16) reinterpret_cast<uint16_t*>(dst)[r * dst_ld + k] = H16<f16_t>::bits(v);
        else reinterpret_cast<bf16_t*>(dst)[r * dst_ld + k] = f2bf(v);
    }
}

// measurement aid (ma_op_stream_copy): the 16-byte-per-lane streaming copy the MI355X guide quotes its achievable HBM rate on
// (6.29 TB/s of the 8 TB/s spec); bench.py times it on the box next to the vendor number
// mode 0: 2048 blocks, grid-stride, non-temporal loads and stores | 1: one 16-byte element per thread, plain loads and stores (the classic
// float4 copy) | 2: one element per thread, non-temporal.  bench.py reports the fastest.
template <int MODE>
__global__ __launch_bounds__(256) void stream_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    if constexpr (MODE == 0) {
        const size_t stride = (size_t)gridDim.x * 256;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
    } else {
        const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
        if (i >= n16) return;
        if constexpr (MODE == 1) dst[i] = src[i];
        else __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
    }
}

// test aid (ma_op_occupy_cus): a workgroup that holds its dynamic LDS allocation and sleeps until `ticks` of the 100 MHz counter passed
__global__ __launch_bounds__(64) void occupy_kernel(unsigned long long ticks, const int* release, unsigned* sink) {
    extern __shared__ char occupy_lds[];
    occupy_lds[threadIdx.x] = 1;
    u64 t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        if (release && __hip_atomic_load(release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;      // the host let go
        __builtin_amdgcn_s_sleep(64);
    }
    if (occupy_lds[threadIdx.x] == 2 && sink) *sink = 1;       // never true: keeps the allocation alive
}

}  // namespace ma
