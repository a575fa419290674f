// Microbenchmark of the persistent form of csrc/gemm256.hpp (round 6) against the one-tile kernel on the 16-bit-output dense-phase shapes:
// time, bitwise equality of the outputs, per-tile stamps (K-loop, epilogue = loop end -> last store issued + next bias requested), ablations
// (no stores / no MFMA) of the persistent form.  Build + run: scripts/gpu_ubench.sh ubench_gemm256p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../meshanything_amd/csrc/gemm256.hpp"
using namespace ma;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static int n_cus = 256;

template <int ACT, int ABL>
float run_p(const GemmTArgs& g, int nty, int ntx, unsigned long long* tr, int reps) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<bf16_t, ACT, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, G256P_LDS));
    const int grid = nty * ntx <= n_cus ? nty * ntx : n_cus & ~7;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gemm256p_kernel<bf16_t, ACT, ABL>), dim3(grid), dim3(512), G256P_LDS, 0, g, nty, ntx, (unsigned long long*)nullptr);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm256p_kernel<bf16_t, ACT, ABL>), dim3(grid), dim3(512), G256P_LDS, 0, g, nty, ntx, (unsigned long long*)nullptr);
    CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (tr) { hipLaunchKernelGGL((gemm256p_kernel<bf16_t, ACT, ABL>), dim3(grid), dim3(512), G256P_LDS, 0, g, nty, ntx, tr); CK(hipDeviceSynchronize()); }
    return ms / reps * 1e3f;
}
template <int ACT>
float run_1(const GemmTArgs& g, int nty, int ntx, unsigned long long* tr, int reps) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<bf16_t, ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gemm256_kernel<bf16_t, ACT>), dim3(nty * ntx), dim3(512), G256_LDS, 0, g, nty, ntx, (unsigned long long*)nullptr);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm256_kernel<bf16_t, ACT>), dim3(nty * ntx), dim3(512), G256_LDS, 0, g, nty, ntx, (unsigned long long*)nullptr);
    CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (tr) { hipLaunchKernelGGL((gemm256_kernel<bf16_t, ACT>), dim3(nty * ntx), dim3(512), G256_LDS, 0, g, nty, ntx, tr); CK(hipDeviceSynchronize()); }
    return ms / reps * 1e3f;
}

static void stamps(unsigned long long* tr, int tiles, int nk, const char* tag) {
    std::vector<unsigned long long> ht((size_t)tiles * 4);
    CK(hipMemcpy(ht.data(), tr, ht.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> loop, epi;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < tiles; ++b) {
        loop.push_back((ht[b * 4 + 1] - ht[b * 4]) / 100.0); epi.push_back((ht[b * 4 + 2] - ht[b * 4 + 1]) / 100.0);
        t0 = std::min(t0, ht[b * 4]); t1 = std::max(t1, ht[b * 4 + 2]);
    }
    std::sort(loop.begin(), loop.end()); std::sort(epi.begin(), epi.end());
    printf("      %s per tile (us): K-loop median %.2f = %.3f per K-tile (min %.2f max %.2f) | loop end -> last store issued median %.2f max %.2f | first loop start -> last end %.1f\n", tag,
           loop[tiles / 2], loop[tiles / 2] / nk, loop[0], loop[tiles - 1], epi[tiles / 2], epi[tiles - 1], (t1 - t0) / 100.0);
}

template <int ACT>
void shape(int M, int N, int K) {
    bf16_t *A, *W, *Cb; float* bias; unsigned long long* tr;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&Cb, (size_t)M * N * 2)); CK(hipMalloc(&bias, (size_t)N * 4));
    std::vector<bf16_t> h((size_t)std::max(M, N) * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = f2bf((float)((i * 2654435761u >> 8) & 0xffff) / 32768.f - 1.f);
    CK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
    std::vector<float> hb(N);
    for (int i = 0; i < N; ++i) hb[i] = (float)((i * 40503u >> 4) & 0xfff) / 2048.f - 1.f;
    CK(hipMemcpy(bias, hb.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    const int ntx = N / 256, nty = M / 256, tiles = ntx * nty;
    CK(hipMalloc(&tr, (size_t)tiles * 4 * 8)); CK(hipMemset(tr, 0, (size_t)tiles * 4 * 8));
    GemmTArgs g{}; g.A = A; g.lda = K; g.W = W; g.bias = bias; g.M = M; g.N = N; g.K = K; g.cmap = RowMap{0, 0, 0}; g.Cb = Cb; g.ldcb = N;
    const double fl = 2.0 * M * N * K;
    const size_t obytes = (size_t)M * N * 2;
    std::vector<char> ref(obytes), got(obytes);
    printf("M %d N %d K %d act %d (%d tiles = %.2f rounds, %d K-tiles)\n", M, N, K, ACT, tiles, (double)tiles / n_cus, K / 64);
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(Cb, 0, obytes));
        float us = run_1<ACT>(g, nty, ntx, tr, 10);
        CK(hipMemcpy(ref.data(), Cb, obytes, hipMemcpyDeviceToHost));
        printf("   one tile per workgroup: %.1f us = %.0f TF\n", us, fl / us * 1e-6);
        stamps(tr, tiles, K / 64, "one-tile  ");
        CK(hipMemset(Cb, 0, obytes));
        us = run_p<ACT, 0>(g, nty, ntx, tr, 10);
        CK(hipMemcpy(got.data(), Cb, obytes, hipMemcpyDeviceToHost));
        printf("   persistent:             %.1f us = %.0f TF   output %s the one-tile kernel's\n", us, fl / us * 1e-6, memcmp(ref.data(), got.data(), obytes) == 0 ? "==" : "DIFFERS FROM");
        stamps(tr, tiles, K / 64, "persistent");
    }
    const float t8 = run_p<ACT, 8>(g, nty, ntx, nullptr, 10), t4 = run_p<ACT, 4>(g, nty, ntx, nullptr, 10), t3 = run_p<ACT, 3>(g, nty, ntx, nullptr, 10);
    printf("   persistent ablations: no stores %.1f us | no MFMA %.1f | no DMA + no ds_read (MFMA + epilogue only) %.1f = %.0f TF\n", t8, t4, t3, fl / t3 * 1e-6);
    hipFree(A); hipFree(W); hipFree(Cb); hipFree(bias); hipFree(tr);
}

int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); n_cus = pr.multiProcessorCount;
    printf("%s, %d CUs\n", pr.name, n_cus);
    shape<0>(16384, 4096, 1024);
    shape<1>(16384, 4096, 1024);
    shape<0>(16384, 3072, 1024);
    shape<0>(16384, 2304, 768);
    shape<2>(16384, 3072, 768);
    shape<0>(67584, 2304, 768);
    shape<0>(8192, 8192, 4096);
    shape<0>(4096, 3072, 1024);          // 192 tiles: less than one round (the launcher keeps the one-tile kernel there)
    shape<0>(5120, 4096, 1024);          // 320 tiles: 1.25 rounds
    return 0;
}
