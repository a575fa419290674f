// Launch-floor microbenchmark: how long does one node of a linear hipGraph chain cost on this box?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_empty(float* p) { if (p == nullptr) p[0] = 1.f; }
__global__ void k_dep(const float* in, float* out) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < 1024) out[i] = in[i] + 1.f; }
// stream `bytes` with 16-B nontemporal loads, one partial sum per wave
__global__ __launch_bounds__(256) void k_stream(const u32x4* __restrict__ w, size_t n16, const float* __restrict__ in, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float x = in[threadIdx.x & 1023];
    float acc = 0.f;
    u32x4 v[8];
    int cnt = 0;
    for (; i < n16 && cnt < 8; i += stride) v[cnt++] = __builtin_nontemporal_load(w + i);
    for (int c = 0; c < cnt; ++c) acc += __uint_as_float(v[c].x) * x + __uint_as_float(v[c].y) + __uint_as_float(v[c].z) + __uint_as_float(v[c].w);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 4 + (threadIdx.x >> 6)) & 1023] = acc;
}

template <typename F>
int time_graph(const char* name, int nodes, int reps, F enqueue) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nodes; ++i) enqueue(s, i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-44s graph : %7.3f us/node\n", name, ms * 1e3 / (reps * nodes));
    // eager
    CK(hipEventRecord(a, s));
    for (int r = 0; r < reps; ++r) for (int i = 0; i < nodes; ++i) enqueue(s, i);
    CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms, a, b));
    printf("%-44s eager : %7.3f us/node\n", name, ms * 1e3 / (reps * nodes));
    return 0;
}

int main() {
    float *va, *vb; CK(hipMalloc(&va, 4096 * 4)); CK(hipMalloc(&vb, 4096 * 4));
    CK(hipMemset(va, 0, 4096 * 4)); CK(hipMemset(vb, 0, 4096 * 4));
    const size_t total = 640ull << 20;            // > 256 MiB Infinity Cache: every pass streams from HBM
    char* w; CK(hipMalloc(&w, total)); CK(hipMemset(w, 1, total));
    time_graph("empty <<<1,64>>>", 200, 20, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, va); });
    time_graph("empty <<<256,256>>>", 200, 20, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, va); });
    time_graph("empty <<<1024,256>>>", 200, 20, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s, va); });
    time_graph("dependent 4KB vector <<<4,256>>>", 200, 20, [&](hipStream_t s, int i) { hipLaunchKernelGGL(k_dep, dim3(4), dim3(256), 0, s, (i & 1) ? vb : va, (i & 1) ? va : vb); });
    for (size_t mb : {2, 6, 8, 16}) {
        const size_t bytes = mb << 20, n16 = bytes / 16;
        const int nmat = (int)(total / bytes);
        for (int blocks : {256, 512, 1024, 2048}) {
            if ((size_t)blocks * 256 * 8 < n16) continue;     // 8 loads per lane max
            char name[96]; snprintf(name, sizeof name, "stream %zu MB distinct, <<<%d,256>>>", mb, blocks);
            time_graph(name, 160, 10, [&](hipStream_t s, int i) {
                hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, s, (const u32x4*)(w + (size_t)(i % nmat) * bytes), n16, (i & 1) ? vb : va, (i & 1) ? va : vb);
            });
        }
    }
    return 0;
}
