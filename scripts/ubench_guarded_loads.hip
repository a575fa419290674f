// Microbenchmark behind DESIGN.md section 3.5: how long does ONE block take to get N vectors that do not depend on each other, when the source
// asks for them (a) behind a lane-dependent guard `if (i < n) v = p[i]`, (b) behind a block-uniform run-time branch whose other side is a
// different load, (c) with clamped indices and no branch?  One block of 256 threads (nothing else on the chip hides the latency -- the
// situation of a decode-step launch: one wave per SIMD), cold lines for every vector (a new 1 MB-strided region per launch), 100 MHz
// real-time counter around the sequence.  Build + run: scripts/gpu_ubench_guarded_loads.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NV = 8;                                   // vectors per thread
constexpr size_t STRIDE = 1 << 18;                      // floats between two vectors of a thread (1 MB): different DRAM pages

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ p, const float* __restrict__ alt, int n, int use_alt, float* out, unsigned long long* t) {
    const int tid = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    f32x4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = tid + 256 * j;                    // n = 256 NV: every guard is true, every clamp a no-op
        if constexpr (MODE == 0) {                      // (a) lane-dependent guard
            v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < n) v[j] = *reinterpret_cast<const f32x4*>(p + j * STRIDE + 4 * tid);
        } else if constexpr (MODE == 1) {               // (b) uniform branch, two alternative loads merged into one variable
            if (use_alt) v[j] = *reinterpret_cast<const f32x4*>(alt + j * STRIDE + 4 * tid);
            else v[j] = *reinterpret_cast<const f32x4*>(p + j * STRIDE + 4 * tid);
        } else {                                        // (c) clamped index, no branch
            const int ic = min(i, n - 1) - 256 * j;
            v[j] = *reinterpret_cast<const f32x4*>(p + j * STRIDE + 4 * ic);
        }
    }
    f32x4 s = v[0];
#pragma unroll
    for (int j = 1; j < NV; ++j) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
    out[tid] = (s.x + s.y) + (s.z + s.w);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) t[0] = t1 - t0;
}

template <int MODE> double run(const float* base, size_t region, float* out, unsigned long long* t, int reps) {
    std::vector<double> us;
    for (int r = 0; r < reps; ++r) {
        const float* p = base + (size_t)r * region;     // untouched lines every launch
        hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(256), 0, 0, p, p + 64, 256 * NV, 0, out, t);
        unsigned long long h;
        CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
        us.push_back(h / 100.0);
    }
    std::sort(us.begin(), us.end());
    return us[us.size() / 2];
}

int main() {
    const int reps = 41;
    const size_t region = NV * STRIDE + 4096;
    float *base, *out; unsigned long long* t;
    CK(hipMalloc(&base, region * reps * sizeof(float))); CK(hipMemset(base, 0, region * reps * sizeof(float)));
    CK(hipMalloc(&out, 1024)); CK(hipMalloc(&t, 64));
    CK(hipDeviceSynchronize());
    const double a = run<0>(base, region, out, t, reps);
    CK(hipMemset(base, 0, region * reps * sizeof(float))); CK(hipDeviceSynchronize());      // (evicts nothing by itself, but keeps the three runs alike)
    const double b = run<1>(base, region, out, t, reps);
    CK(hipMemset(base, 0, region * reps * sizeof(float))); CK(hipDeviceSynchronize());
    const double c = run<2>(base, region, out, t, reps);
    printf("%d independent 16-byte vectors per thread, one block of 256 threads, cold lines, median of %d launches (us from the first request to the sum stored):\n", NV, reps);
    printf("  (a) lane-dependent guard `if (i < n) v = p[i]`            : %6.2f us\n", a);
    printf("  (b) block-uniform branch between two loads (hipcc folds this simple form into a pointer select: no penalty) : %6.2f us\n", b);
    printf("  (c) clamped index, no branch                              : %6.2f us\n", c);
    printf("  -> (a)/(c) = %.1fx, (b)/(c) = %.1fx\n", a / c, b / c);
    return 0;
}
