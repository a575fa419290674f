// Microbenchmark of the epilogue forms of csrc/gemm256.hpp (round 6): EV bit 2 = straight-line whole-tile stores (bit 1, the hardware 16-bit
// conversion, measured in the first run -- profiles/r06_ubench_gemm256_epi.txt -- is common.hpp's f2bf / pack2 now); 16-bit output and fp32 output + residual.  Block-level stamps as scripts/ubench_gemm256.hip; the outputs of every form are compared
// bit for bit with form 0, and the hardware fp32 -> bf16 conversion is compared with common.hpp's f2bf over ALL 2^32 inputs.
// Build + run: hipcc --offload-arch=gfx950 -O3 -DNDEBUG -std=c++17 scripts/ubench_gemm256_epi.hip -o /tmp/ub_epi && /tmp/ub_epi
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../meshanything_amd/csrc/gemm256.hpp"
using namespace ma;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// the integer round-to-nearest-even form (what common.hpp's f2bf is on the host, and was on the device until round 6)
__device__ inline unsigned short f2bf_int(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__global__ void cvt_sweep_kernel(unsigned long long* bad, unsigned* first_bad) {
    const unsigned long long n = 1ull << 32, stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long cnt = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned u = (unsigned)i;
        if ((u & 0x7fffffffu) > 0x7f800000u) continue;               // NaN payloads: not compared
        const float f = __uint_as_float(u);
        const unsigned short a = f2bf_int(f), b = f2bf(f);
        if (a != b) { if (cnt == 0) atomicMin(first_bad, u); ++cnt; }
    }
    if (cnt) atomicAdd(bad, cnt);
}

template <int EV>
float run(const GemmTArgs& g, int tiles, int ntx, unsigned long long* tr, int reps) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<bf16_t, 0, 0, EV>), hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gemm256_kernel<bf16_t, 0, 0, EV>), dim3(tiles), dim3(512), G256_LDS, 0, g, tiles / ntx, ntx, (unsigned long long*)nullptr);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm256_kernel<bf16_t, 0, 0, EV>), dim3(tiles), dim3(512), G256_LDS, 0, g, tiles / ntx, ntx, (unsigned long long*)nullptr);
    CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (tr) {
        hipLaunchKernelGGL((gemm256_kernel<bf16_t, 0, 0, EV>), dim3(tiles), dim3(512), G256_LDS, 0, g, tiles / ntx, ntx, tr);
        CK(hipDeviceSynchronize());
    }
    return ms / reps * 1e3f;
}

static void stamps(unsigned long long* tr, int tiles, int nk, const char* tag) {
    std::vector<unsigned long long> ht((size_t)tiles * 4);
    CK(hipMemcpy(ht.data(), tr, ht.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> loop, epi;
    for (int b = 0; b < tiles; ++b) { loop.push_back((ht[b * 4 + 1] - ht[b * 4]) / 100.0); epi.push_back((ht[b * 4 + 2] - ht[b * 4 + 1]) / 100.0); }
    std::sort(loop.begin(), loop.end()); std::sort(epi.begin(), epi.end());
    printf("      %s per block (us): K-loop median %.2f = %.3f per K-tile | epilogue (to the last store ISSUED) median %.2f max %.2f\n", tag, loop[tiles / 2], loop[tiles / 2] / nk,
           epi[tiles / 2], epi[tiles - 1]);
}

int main() {
    {
        unsigned long long* bad; unsigned* fb;
        CK(hipMalloc(&bad, 8)); CK(hipMalloc(&fb, 4)); CK(hipMemset(bad, 0, 8)); CK(hipMemset(fb, 0xff, 4));
        hipLaunchKernelGGL(cvt_sweep_kernel, dim3(4096), dim3(256), 0, 0, bad, fb);
        unsigned long long hb; unsigned hf;
        CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hf, fb, 4, hipMemcpyDeviceToHost));
        printf("fp32 -> bf16: hardware conversion vs f2bf over all 2^32 non-NaN inputs: %llu differ (first 0x%08x)\n", hb, hf);
    }
    // {M, N, K, fp32-out + residual?}
    const int shapes[][4] = {{16384, 4096, 1024, 0}, {16384, 3072, 1024, 0}, {16384, 1024, 1024, 1}, {16384, 1024, 4096, 1}, {16384, 2304, 768, 0}, {16384, 768, 768, 1}, {16384, 768, 3072, 1}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2], f32out = sh[3];
        bf16_t *A, *W, *Cb; float *C, *R, *bias; unsigned long long* tr;
        CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&Cb, (size_t)M * N * 2)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&R, (size_t)M * N * 4));
        CK(hipMalloc(&bias, (size_t)N * 4));
        std::vector<bf16_t> h((size_t)std::max(M, N) * K);
        for (size_t i = 0; i < h.size(); ++i) h[i] = f2bf((float)((i * 2654435761u >> 8) & 0xffff) / 32768.f - 1.f);
        CK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
        std::vector<float> hr((size_t)M * N);
        for (size_t i = 0; i < hr.size(); ++i) hr[i] = (float)((i * 40503u >> 4) & 0xfff) / 2048.f - 1.f;
        CK(hipMemcpy(R, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(bias, hr.data(), (size_t)N * 4, hipMemcpyHostToDevice));
        const int ntx = N / 256, tiles = ntx * (M / 256);
        CK(hipMalloc(&tr, (size_t)tiles * 4 * 8)); CK(hipMemset(tr, 0, (size_t)tiles * 4 * 8));
        GemmTArgs g{}; g.A = A; g.lda = K; g.W = W; g.bias = bias; g.M = M; g.N = N; g.K = K; g.cmap = RowMap{0, 0, 0};
        if (f32out) { g.C = C; g.ldc = N; g.R = R; g.ldr = N; } else { g.Cb = Cb; g.ldcb = N; }
        const double fl = 2.0 * M * N * K;
        const size_t obytes = f32out ? (size_t)M * N * 4 : (size_t)M * N * 2;
        void* out = f32out ? (void*)C : (void*)Cb;
        std::vector<char> ref(obytes), got(obytes);
        printf("M %d N %d K %d %s (%d tiles, %d K-tiles)\n", M, N, K, f32out ? "fp32 out + residual" : "16-bit out", tiles, K / 64);
        auto one = [&](int ev, float us) {
            CK(hipMemcpy(got.data(), out, obytes, hipMemcpyDeviceToHost));
            if (ev == 0) ref = got;
            const bool same = memcmp(ref.data(), got.data(), obytes) == 0;
            printf("   EV %d: %.1f us = %.0f TF   output %s form 0\n", ev, us, fl / us * 1e-6, same ? "==" : "DIFFERS FROM");
            char tag[16]; snprintf(tag, sizeof tag, "EV %d", ev);
            stamps(tr, tiles, K / 64, tag);
            CK(hipMemset(out, 0, obytes));
        };
        one(0, run<0>(g, tiles, ntx, tr, 10));
        one(2, run<2>(g, tiles, ntx, tr, 10));
        one(0, run<0>(g, tiles, ntx, tr, 10));
        one(2, run<2>(g, tiles, ntx, tr, 10));
        hipFree(A); hipFree(W); hipFree(Cb); hipFree(C); hipFree(R); hipFree(bias); hipFree(tr);
    }
    return 0;
}
