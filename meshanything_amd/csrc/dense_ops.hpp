// Row-wise / gather kernels of the dense phases (point encoder, decoder prefill, detokenizer), for a BATCH of samples
// stacked along the rows.  Two buffer kinds: fp32 "stream" tensors (residual streams, LayerNorm inputs, user-visible
// outputs) and "activation" tensors of type AT that only ever feed a GEMM or the attention kernel -- AT = bf16_t under the
// bf16 policy (the policy's rounding point at a GEMM / attention input is applied once, by the producer, and the tensor
// crosses HBM in 2 bytes), AT = float under the fp32 "exact" policy.  Each kernel cites the reference lines it implements.
#pragma once
#include "common.hpp"
#include "misc.hpp"

namespace ma {

template <typename T> __device__ __forceinline__ void st_act(T* p, float v);
template <> __device__ __forceinline__ void st_act<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_act<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }
template <> __device__ __forceinline__ void st_act<f16_t>(f16_t* p, float v) { p->v = H16<f16_t>::bits(v); }

// FourierEmbedder.forward (embedder.py:87-105) + normals concat (sal_perceiver.py:87-89), rows = all points of the batch:
// out[i] = [x(3) | sin(x_d * 2^f) (d-major) | cos(...) | normal(3) | 0-pad to ld]
template <typename PT, typename AT>
__global__ void fourier2_kernel(const PT* __restrict__ pc, int n_rows, int F, AT* __restrict__ out, int ld) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * ld) return;
    const int i = idx / ld, col = idx - i * ld;
    const PT* p = pc + (size_t)i * 6;
    float v = 0.f;
    if (col < 3) v = (float)p[col];
    else if (col < 3 + 6 * F) {
        const int j = (col - 3) % (3 * F);
        const int dim = j / F, fr = j - dim * F;
        const float arg = (float)p[dim] * (float)(1 << fr);
        v = (col < 3 + 3 * F) ? sinf(arg) : cosf(arg);
    } else if (col < 6 + 6 * F) v = (float)p[3 + col - (3 + 6 * F)];
    st_act<AT>(out + idx, v);
}

// nn.LayerNorm over the last dim, one wave per row (two-pass mean / variance in fp32, the row held in registers: one global
// read, 16-byte accesses): x fp32 -> y32 (fp32, optional) and ya (AT, optional).  y32 may alias x.  D % 4 == 0, D <= 4096.
// NVT > 0: D == 256 NVT exactly -- every lane owns NVT chunks, no lane-dependent guard: the row's NVT requests and the 2 NVT parameter
// requests all go out before the first use (hipcc waits for a request made inside a lane-dependent branch where the branch ends: the
// guarded form, NVT == 0, is one round trip per chunk and streamed 1.8 TB/s; profiles/r04_dense_b64_kernel_stats.csv).
// KS > 1 (NVT > 0 only): the input of rows < split_rows is the sum of KS partial buffers x + p * part_stride (a GEMM split along K, gemm256.hpp
// GemmSplitK), added up in ascending p; rows behind split_rows are complete in part 0.  All KS x NVT requests of a row go out first.
template <typename AT, int NVT, int KS = 1>
__global__ __launch_bounds__(256) void ln_rows2_kernel(const float* __restrict__ x, int ldx, RowMap xin, const float* __restrict__ g,
                                                       const float* __restrict__ b, float eps, float* y32, int ld32, AT* __restrict__ ya,
                                                       int lda, RowMap yout, int rows, int D, long part_stride = 0, int split_rows = 0) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + xin(row) * ldx;
    constexpr int NV = NVT > 0 ? NVT : 16;                // float4 chunks per lane: D <= 64 * 4 * 16
    const int nq = D >> 2;
    auto in = [&](int j) { return NVT > 0 || lane + 64 * j < nq; };
    f32x4 v[NV], g4[NV], b4[NV];
    float s = 0.f;
    if constexpr (NVT > 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const f32x4*>(xr + 4 * (lane + 64 * j));
        if constexpr (KS > 1) {
            if (row < split_rows) {                               // (one row per wave: a wave-uniform branch)
                f32x4 pv[KS - 1][NV];
#pragma unroll
                for (int p = 1; p < KS; ++p)
#pragma unroll
                    for (int j = 0; j < NV; ++j) pv[p - 1][j] = *reinterpret_cast<const f32x4*>(xr + (size_t)p * part_stride + 4 * (lane + 64 * j));
#pragma unroll
                for (int p = 1; p < KS; ++p)
#pragma unroll
                    for (int j = 0; j < NV; ++j) { v[j].x += pv[p - 1][j].x; v[j].y += pv[p - 1][j].y; v[j].z += pv[p - 1][j].z; v[j].w += pv[p - 1][j].w; }
            }
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) { g4[j] = *reinterpret_cast<const f32x4*>(g + 4 * (lane + 64 * j)); b4[j] = *reinterpret_cast<const f32x4*>(b + 4 * (lane + 64 * j)); }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < NV; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int q = lane + 64 * j;
            if (q < nq) { v[j] = *reinterpret_cast<const f32x4*>(xr + 4 * q); s += (v[j].x + v[j].y) + (v[j].z + v[j].w); }
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float qq = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        if (in(j)) {
            const float d0 = v[j].x - mean, d1 = v[j].y - mean, d2 = v[j].z - mean, d3 = v[j].w - mean;
            qq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(qq) / (float)D + eps);
    const size_t orow = yout(row);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int q = lane + 64 * j;
        if (in(j)) {
            if constexpr (NVT == 0) { g4[j] = *reinterpret_cast<const f32x4*>(g + 4 * q); b4[j] = *reinterpret_cast<const f32x4*>(b + 4 * q); }
            f32x4 o;
            o.x = (v[j].x - mean) * rstd * g4[j].x + b4[j].x; o.y = (v[j].y - mean) * rstd * g4[j].y + b4[j].y;
            o.z = (v[j].z - mean) * rstd * g4[j].z + b4[j].z; o.w = (v[j].w - mean) * rstd * g4[j].w + b4[j].w;
            if (y32) *reinterpret_cast<f32x4*>(y32 + orow * ld32 + 4 * q) = o;
            if (ya) {
                if constexpr (sizeof(AT) == 4) *reinterpret_cast<f32x4*>(ya + orow * lda + 4 * q) = o;
                else {
                    *reinterpret_cast<u32x2*>(ya + orow * lda + 4 * q) = pack4<AT>(o);
                }
            }
        }
    }
}
template <typename AT>
inline void launch_ln_rows2(const float* x, int ldx, RowMap xin, const float* g, const float* b, float eps, float* y32, int ld32, AT* ya, int lda,
                            RowMap yout, int rows, int D, hipStream_t s, int parts = 1, long part_stride = 0, int split_rows = 0) {
    const dim3 grid((rows + 3) / 4), block(256);
    if (parts == 4 && D == 1024) { hipLaunchKernelGGL((ln_rows2_kernel<AT, 4, 4>), grid, block, 0, s, x, ldx, xin, g, b, eps, y32, ld32, ya, lda, yout, rows, D, part_stride, split_rows); return; }
    if (parts == 2 && D == 1024) { hipLaunchKernelGGL((ln_rows2_kernel<AT, 4, 2>), grid, block, 0, s, x, ldx, xin, g, b, eps, y32, ld32, ya, lda, yout, rows, D, part_stride, split_rows); return; }
    if (D == 1024) hipLaunchKernelGGL((ln_rows2_kernel<AT, 4>), grid, block, 0, s, x, ldx, xin, g, b, eps, y32, ld32, ya, lda, yout, rows, D);
    else if (D == 768) hipLaunchKernelGGL((ln_rows2_kernel<AT, 3>), grid, block, 0, s, x, ldx, xin, g, b, eps, y32, ld32, ya, lda, yout, rows, D);
    else if (D == 512) hipLaunchKernelGGL((ln_rows2_kernel<AT, 2>), grid, block, 0, s, x, ldx, xin, g, b, eps, y32, ld32, ya, lda, yout, rows, D);
    else if (D == 256) hipLaunchKernelGGL((ln_rows2_kernel<AT, 1>), grid, block, 0, s, x, ldx, xin, g, b, eps, y32, ld32, ya, lda, yout, rows, D);
    else hipLaunchKernelGGL((ln_rows2_kernel<AT, 0>), grid, block, 0, s, x, ldx, xin, g, b, eps, y32, ld32, ya, lda, yout, rows, D);
}

// out[i][n] = (mask == null || mask[i] ? in[i][n] : 0) + (t0 ? t0[n] : 0) + (tab ? tab[((tab_mod ? i % tab_mod : i) + row0) * ld_tab + n] : 0)
// -> fp32 (out32, may alias in) and AT copy (outa, optional):
//  - decoder prefill: prefix + cond_embed[0] + embed_positions[2 + i]         (shape_opt.py:331-337, 359-364)
//  - detokenizer:     point feature + point_pe[i]; masked face embeds + pos_embedding[i]   (meshanything.py:47, 58-60)
template <typename AT>
__global__ void add_rows2_kernel(const float* in, int ld_in, const unsigned char* __restrict__ mask, const float* __restrict__ t0,
                                 const float* __restrict__ tab, int ld_tab, int row0, float* out32, int ld_out, AT* __restrict__ outa,
                                 int ld_outa, int rows, int cols, int tab_mod) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const int i = idx / cols, n = idx - i * cols;
    float v = (mask == nullptr || mask[i]) ? in[(size_t)i * ld_in + n] : 0.f;
    if (t0) v += t0[n];
    if (tab) v += tab[(size_t)((tab_mod > 0 ? i % tab_mod : i) + row0) * ld_tab + n];
    if (out32) out32[(size_t)i * ld_out + n] = v;
    if (outa) st_act<AT>(outa + (size_t)i * ld_outa + n, v);
}

// dst[i][n] (AT) = mask == null || mask[i] ? src[in(i)][n] : 0   -- fp32 stream rows -> a GEMM operand (slices of the 257-row
// latent blocks, the cat([latents, shape_latents]) of meshanything.py:128-131, the masked decoded faces of :66-68)
template <typename AT>
__global__ void cvt_rows_kernel(const float* __restrict__ src, int lds, RowMap in, const unsigned char* __restrict__ mask, AT* __restrict__ dst,
                                int ldd, int rows, int cols) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const int i = idx / cols, n = idx - i * cols;
    const float v = (mask == nullptr || mask[i]) ? src[in(i) * lds + n] : 0.f;
    st_act<AT>(dst + (size_t)i * ldd + n, v);
}

// get_codes (meshanything.py:178-212) fused with the 'b (nf nv) d -> b nf (nv d)' rearrange (:53) and the face mask (:57), for
// all faces of the batch: out[f][v*D + d] = sum_{q<3} codebook[ids[f*9 + v*3 + q]][d] (pad -1 contributes 0);
// mask[f] = all nine ids != -1.  out32 (fp32, optional: ma_get_codes) and outa (AT, optional: the project_down GEMM operand).
// Four consecutive d per thread (D % 4 == 0): 16-byte gathers from the codebook rows, one 16- / 8-byte store per output.
template <typename AT>
__global__ void codes_gather2_kernel(const long long* __restrict__ ids, const float* __restrict__ codebook, int D, int nf, float* __restrict__ out32,
                                     AT* __restrict__ outa, unsigned char* __restrict__ mask) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int D4 = D >> 2;
    if (idx >= nf * 3 * D4) return;
    const int f = idx / (3 * D4), rem = idx - f * 3 * D4, v = rem / D4, d = (rem - v * D4) * 4;
    const long long* ip = ids + (size_t)f * 9 + v * 3;
    const long long i0 = ip[0], i1 = ip[1], i2 = ip[2];
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 c0 = i0 < 0 ? z : *reinterpret_cast<const f32x4*>(codebook + (size_t)i0 * D + d);
    const f32x4 c1 = i1 < 0 ? z : *reinterpret_cast<const f32x4*>(codebook + (size_t)i1 * D + d);
    const f32x4 c2 = i2 < 0 ? z : *reinterpret_cast<const f32x4*>(codebook + (size_t)i2 * D + d);
    f32x4 sum;
    sum.x = (c0.x + c1.x) + c2.x; sum.y = (c0.y + c1.y) + c2.y; sum.z = (c0.z + c1.z) + c2.z; sum.w = (c0.w + c1.w) + c2.w;
    const size_t o = (size_t)f * 3 * D + (size_t)v * D + d;
    if (out32) *reinterpret_cast<f32x4*>(out32 + o) = sum;
    if (outa) {
        if constexpr (sizeof(AT) == 4) *reinterpret_cast<f32x4*>(outa + o) = sum;
        else *reinterpret_cast<u32x2*>(outa + o) = pack4<AT>(sum);
    }
    if (rem == 0 && mask) {
        bool ok = true;
        for (int q = 0; q < 9; ++q) ok = ok && ids[(size_t)f * 9 + q] != -1;
        mask[f] = ok ? 1 : 0;
    }
}

template <typename KT> __device__ __forceinline__ void store_kv_elem(KT* p, float v) { *reinterpret_cast<uint16_t*>(p) = H16<KT>::bits(v); }

// fill the KV cache from the prefill's fused q|k|v projection (AT): src (B * rows, ld) with K at column koff + h*64 + d, V at
// voff + ...; grid.y = batch row b: its `rows` source rows start at b * rows, its planes at b * kv_row_stride elements
template <typename AT, typename KT>
__global__ void kv_fill2_kernel(const AT* __restrict__ src, int ld, int koff, int voff, int rows, int H, int max_seq, KT* __restrict__ kc,
                                KT* __restrict__ vc, size_t kv_row_stride) {
    const int b = blockIdx.y;
    if constexpr (sizeof(AT) == 2 && sizeof(KT) == 2) {                    // same 16-bit format: 16-byte copies, 8 elements per thread
        const int idx = blockIdx.x * blockDim.x + threadIdx.x;
        if (idx >= rows * H * 8) return;
        const int d = (idx & 7) * 8, h = (idx >> 3) % H, r = idx / (8 * H);
        const size_t dst = (size_t)b * kv_row_stride + ((size_t)h * max_seq + r) * 64 + d;
        const AT* sp = src + ((size_t)b * rows + r) * ld + h * 64 + d;
        const u32x4 kv = *reinterpret_cast<const u32x4*>(sp + koff), vv = *reinterpret_cast<const u32x4*>(sp + voff);
        *reinterpret_cast<u32x4*>(kc + dst) = kv;
        *reinterpret_cast<u32x4*>(vc + dst) = vv;
    } else {
        const int idx = blockIdx.x * blockDim.x + threadIdx.x;
        const int total = rows * H * 64;
        if (idx >= total) return;
        const int d = idx & 63, h = (idx >> 6) % H, r = idx / (64 * H);
        const size_t dst = (size_t)b * kv_row_stride + ((size_t)h * max_seq + r) * 64 + d;
        const AT* sp = src + ((size_t)b * rows + r) * ld;
        if constexpr (sizeof(AT) == sizeof(KT)) { kc[dst] = sp[koff + h * 64 + d]; vc[dst] = sp[voff + h * 64 + d]; }     // same type: plain copy
        else { store_kv_elem<KT>(kc + dst, (float)sp[koff + h * 64 + d]); store_kv_elem<KT>(vc + dst, (float)sp[voff + h * 64 + d]); }
    }
}

// the same for the rows [r_begin, r_end) of the STACKED tensor (row r = sample r / T, position r % T): the rows whose K / V the q|k|v GEMM did not
// write into the planes itself (gemm256.hpp, GemmTArgs::kv_*: the 64-row tail behind the 256-row tiles; everything when that kernel did not run)
template <typename AT, typename KT>
__global__ void kv_fill_rows_kernel(const AT* __restrict__ src, int ld, int koff, int voff, int r_begin, int r_end, int T, int H, int max_seq, KT* __restrict__ kc,
                                    KT* __restrict__ vc, size_t kv_row_stride) {
    constexpr int V = (sizeof(AT) == 2 && sizeof(KT) == 2) ? 8 : 1;      // elements per thread
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per_row = H * 64 / V;
    if (idx >= (long)(r_end - r_begin) * per_row) return;
    const int r = r_begin + (int)(idx / per_row), e = (int)(idx % per_row) * V, h = e >> 6, d = e & 63;
    const int b = r / T, pos = r - b * T;
    const size_t dst = (size_t)b * kv_row_stride + ((size_t)h * max_seq + pos) * 64 + d;
    const AT* sp = src + (size_t)r * ld + h * 64 + d;
    if constexpr (V == 8) {
        *reinterpret_cast<u32x4*>(kc + dst) = *reinterpret_cast<const u32x4*>(sp + koff);
        *reinterpret_cast<u32x4*>(vc + dst) = *reinterpret_cast<const u32x4*>(sp + voff);
    } else if constexpr (sizeof(AT) == sizeof(KT)) { kc[dst] = sp[koff]; vc[dst] = sp[voff]; }
    else { store_kv_elem<KT>(kc + dst, (float)sp[koff]); store_kv_elem<KT>(vc + dst, (float)sp[voff]); }
}

}  // namespace ma
