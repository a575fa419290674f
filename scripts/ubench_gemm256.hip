// Microbenchmark of csrc/gemm256.hpp (the library's own kernel, included as is): what bounds its K-loop?  Ablations (no LDS-DMA / no ds_read / no
// MFMA / no stores in the loop), block-level timestamps (prologue done, loop done, end), per shape.  Build + run: scripts/gpu_ubench_gemm256.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../meshanything_amd/csrc/gemm256.hpp"
using namespace ma;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int ABL>
float run(const GemmTArgs& g, int tiles, int ntx, unsigned long long* tr, int reps) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<bf16_t, 0, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gemm256_kernel<bf16_t, 0, ABL>), dim3(tiles), dim3(512), G256_LDS, 0, g, tiles / ntx, ntx, tr);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm256_kernel<bf16_t, 0, ABL>), dim3(tiles), dim3(512), G256_LDS, 0, g, tiles / ntx, ntx, tr);
    CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps * 1e3f;
}

int main() {
    const int shapes[][3] = {{8192, 8192, 512}, {8192, 8192, 4096}, {16384, 4096, 1024}, {16384, 1024, 1024}, {65536, 768, 768}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        bf16_t *A, *W, *Cb; float* C; unsigned long long* tr;
        CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&Cb, (size_t)M * N * 2)); CK(hipMalloc(&C, 64));
        std::vector<bf16_t> h((size_t)std::max(M, N) * K);
        for (size_t i = 0; i < h.size(); ++i) h[i] = f2bf((float)((i * 2654435761u >> 8) & 0xffff) / 32768.f - 1.f);
        CK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
        const int ntx = N / 256, tiles = ntx * (M / 256);
        CK(hipMalloc(&tr, (size_t)tiles * 4 * 8)); CK(hipMemset(tr, 0, (size_t)tiles * 4 * 8));
        GemmTArgs g{}; g.A = A; g.lda = K; g.W = W; g.Cb = Cb; g.ldcb = N; g.C = nullptr; g.M = M; g.N = N; g.K = K; g.cmap = RowMap{0, 0, 0};
        const double fl = 2.0 * M * N * K;
        const float t0 = run<0>(g, tiles, ntx, nullptr, 10), t1 = run<1>(g, tiles, ntx, nullptr, 10), t2 = run<2>(g, tiles, ntx, nullptr, 10), t4 = run<4>(g, tiles, ntx, nullptr, 10),
                    t8 = run<8>(g, tiles, ntx, nullptr, 10), t3 = run<3>(g, tiles, ntx, nullptr, 10), t7 = run<7>(g, tiles, ntx, nullptr, 10), t15 = run<15>(g, tiles, ntx, nullptr, 10);
        printf("M %d N %d K %d (%d tiles, %d K-tiles): full %.1f us = %.0f TF | no DMA %.1f | no ds_read %.1f | no MFMA %.1f | no stores %.1f | no DMA+read (MFMA only) %.1f = %.0f TF | barriers only %.1f | nothing %.1f\n",
               M, N, K, tiles, K / 64, t0, fl / t0 * 1e-6, t1, t2, t4, t8, t3, fl / t3 * 1e-6, t7, t15);
        run<0>(g, tiles, ntx, tr, 1);
        std::vector<unsigned long long> ht((size_t)tiles * 4);
        CK(hipMemcpy(ht.data(), tr, ht.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> pro, loop, epi;
        unsigned long long tmin = ~0ull;
        for (int b = 0; b < tiles; ++b) tmin = std::min(tmin, ht[b * 4]);
        for (int b = 0; b < tiles; ++b) { loop.push_back((ht[b * 4 + 1] - ht[b * 4]) / 100.0); epi.push_back((ht[b * 4 + 2] - ht[b * 4 + 1]) / 100.0); pro.push_back((ht[b * 4] - tmin) / 100.0); }
        std::sort(loop.begin(), loop.end()); std::sort(epi.begin(), epi.end()); std::sort(pro.begin(), pro.end());
        printf("    per block (us): K-loop median %.2f (min %.2f max %.2f) = %.3f us per K-tile | epilogue median %.2f (max %.2f) | loop start after first block's: median %.2f max %.2f\n",
               loop[tiles / 2], loop[0], loop[tiles - 1], loop[tiles / 2] / (K / 64), epi[tiles / 2], epi[tiles - 1], pro[tiles / 2], pro[tiles - 1]);
        hipFree(A); hipFree(W); hipFree(Cb); hipFree(C); hipFree(tr);
    }
    return 0;
}
