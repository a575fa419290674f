// Does a recently-touched operand stream faster than a cold one?  (Infinity Cache / MALL residency across kernel boundaries.)
//   cold : a 1 GiB trash read, then the timed streaming read of an 8 MB buffer (nt 16-byte loads, 1024 blocks, like gemv_kernel)
//   warm : the same buffer touched (one dword per 128-byte line, default policy) by the PREVIOUS kernel, then the timed read
//   self : the timed read repeated on the buffer it has just streamed with nt loads (does an nt stream leave lines behind?)
// Time = last block end - first block start (s_memrealtime, 100 MHz), median over buffers.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_mall scripts/ubench_mall.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void stream_read(const char* p, size_t bytes, unsigned long long* stamps, unsigned* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const size_t per_block = bytes / gridDim.x;
    const char* b = p + (size_t)blockIdx.x * per_block;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t off = (size_t)threadIdx.x * 16; off < per_block; off += 256 * 16 * 8) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const size_t o = off + (size_t)u * 256 * 16; v[u] = o < per_block ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(b + o)) : u32x4{0, 0, 0, 0}; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u && sink) *sink = acc.x;
    __syncthreads();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
}
__global__ __launch_bounds__(256) void touch_lines(const char* p, size_t bytes, unsigned* sink) {
    unsigned acc = 0;
    const size_t lines = bytes / 128;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < lines; i += (size_t)gridDim.x * 256) acc ^= *reinterpret_cast<const unsigned*>(p + i * 128);
    if (acc == 0x9e3779b9u && sink) *sink = acc;
}
__global__ __launch_bounds__(256) void trash_read(const char* p, size_t bytes, unsigned* sink) {
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; i < bytes; i += (size_t)gridDim.x * 256 * 16) acc ^= *reinterpret_cast<const u32x4*>(p + i);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u && sink) *sink = acc.x;
}
static double span_us(const std::vector<unsigned long long>& s) {
    unsigned long long lo = ~0ull, hi = 0;
    for (size_t i = 0; i < s.size(); i += 2) { lo = std::min(lo, s[i]); hi = std::max(hi, s[i + 1]); }
    return (double)(hi - lo) / 100.0;
}
int main() {
    const size_t trash_bytes = 1ull << 30;
    const int NB = 12, blocks = 1024;
    char* trash; CK(hipMalloc(&trash, trash_bytes)); CK(hipMemset(trash, 1, trash_bytes));
    unsigned long long* d_st; CK(hipMalloc(&d_st, blocks * 2 * sizeof(unsigned long long)));
    std::vector<unsigned long long> st(blocks * 2);
    for (size_t mb : {2, 8, 32}) {
        const size_t bytes = mb << 20;
        std::vector<char*> buf(NB);
        for (auto& b : buf) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 3, bytes)); }
        std::vector<double> cold, warm, self, warm_far;
        for (int i = 0; i < NB; ++i) {
            hipLaunchKernelGGL(trash_read, dim3(2048), dim3(256), 0, 0, trash, trash_bytes, nullptr);
            hipLaunchKernelGGL(stream_read, dim3(blocks), dim3(256), 0, 0, buf[i], bytes, d_st, nullptr);
            CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost)); cold.push_back(span_us(st));
            hipLaunchKernelGGL(stream_read, dim3(blocks), dim3(256), 0, 0, buf[i], bytes, d_st, nullptr);
            CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost)); self.push_back(span_us(st));
            hipLaunchKernelGGL(trash_read, dim3(2048), dim3(256), 0, 0, trash, trash_bytes, nullptr);
            hipLaunchKernelGGL(touch_lines, dim3(256), dim3(256), 0, 0, buf[i], bytes, nullptr);
            hipLaunchKernelGGL(stream_read, dim3(blocks), dim3(256), 0, 0, buf[i], bytes, d_st, nullptr);
            CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost)); warm.push_back(span_us(st));
            // touched, then 64 MB of other traffic (8 launches' worth of weights), then read
            hipLaunchKernelGGL(trash_read, dim3(2048), dim3(256), 0, 0, trash, trash_bytes, nullptr);
            hipLaunchKernelGGL(touch_lines, dim3(256), dim3(256), 0, 0, buf[i], bytes, nullptr);
            hipLaunchKernelGGL(trash_read, dim3(2048), dim3(256), 0, 0, trash, (size_t)64 << 20, nullptr);
            hipLaunchKernelGGL(stream_read, dim3(blocks), dim3(256), 0, 0, buf[i], bytes, d_st, nullptr);
            CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost)); warm_far.push_back(span_us(st));
        }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("%2zu MB stream (1024 blocks, nt 16-B loads): cold %.2f us (%.0f GB/s) | touched by previous kernel %.2f us (%.0f GB/s) | "
               "touched, then 64 MB of other reads %.2f us | re-read after own nt stream %.2f us\n",
               mb, med(cold), bytes / med(cold) / 1e3, med(warm), bytes / med(warm) / 1e3, med(warm_far), med(self));
        for (auto b : buf) CK(hipFree(b));
    }
    return 0;
}
