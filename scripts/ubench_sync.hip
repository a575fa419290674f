// In-launch exchange microbenchmark for a persistent decode kernel on MI355X: what does one "phase edge" cost?
//   every block (one per CU, 256 threads) publishes a slice of a vector, then needs the whole vector of its group.
//   group = all 256 blocks (cross-XCD all-gather)  |  group = the blocks that share an XCC (discovered with HW_REG_XCC_ID).
// Protocol (placement-independent, MI355X guide G16): payload = write-through agent-scope stores, one drained arrival
// ticket per block on a per-group counter, consumers poll the counter relaxed, then read the payload with agent-scope
// (L1-bypassing) loads.  Every word is checked; every spin is bounded (abort flag), so a protocol error cannot hang.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Ctl {
    unsigned counters[64];      // [0]: global arrivals; [8 + x]: arrivals of XCC x
    unsigned xcc_slots[8];      // slot allocator per XCC
    unsigned abort_flag;
    unsigned errors;
    unsigned long long spin_total;
};

__device__ inline unsigned ld_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline float ld_f32(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_f32(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// mode 0: global group (256 blocks), mode 1: per-XCC groups.  per_block = floats published per block and iteration.
__global__ __launch_bounds__(256) void k_exchange(Ctl* ctl, float* buf, int iters, int per_block, int mode, int* xcc_of_block) {
    __shared__ unsigned s_slot, s_xcc, s_ok;
    __shared__ float sink[256];
    const int tid = threadIdx.x;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    if (tid == 0) {
        s_xcc = xcc;
        s_slot = mode == 1 ? atomicAdd(&ctl->xcc_slots[xcc & 7], 1u) : blockIdx.x;
        xcc_of_block[blockIdx.x] = (int)xcc;
    }
    __syncthreads();
    const unsigned slot = s_slot;
    const int gsize = mode == 1 ? 32 : gridDim.x;                 // blocks per group (assumes 32 blocks per XCC when mode 1)
    unsigned* counter = mode == 1 ? &ctl->counters[8 + (s_xcc & 7)] : &ctl->counters[0];
    // double-buffered payload: iteration parity selects the half (a fast block may publish it+1 while a slow one still reads it)
    float* gbase = buf + (size_t)(mode == 1 ? (s_xcc & 7) : 0) * 2 * 256 * per_block;
    const int vec = gsize * per_block;
    float acc = 0.f;
    unsigned long long spins = 0;
    for (int it = 0; it < iters; ++it) {
        float* g = gbase + (size_t)(it & 1) * 256 * per_block;
        // publish my slice
        for (int i = tid; i < per_block; i += 256) st_f32(g + slot * per_block + i, (float)(it * 1000 + (int)slot) + 0.001f * i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)gsize * (unsigned)(it + 1);
            unsigned ok = 1, n = 0;
            while (ld_u32(counter) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++n > 2000000u || ld_u32(&ctl->abort_flag)) { ok = 0; __hip_atomic_store(&ctl->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            spins += n;
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) break;
        // consume the whole group vector
        unsigned bad = 0;
        for (int i = tid; i < vec; i += 256) {
            const float v = ld_f32(g + i);
            const int p = i / per_block, j = i - p * per_block;
            const float want = (float)(it * 1000 + p) + 0.001f * j;
            if (v != want) ++bad;
            acc += v;
        }
        if (bad) atomicAdd(&ctl->errors, bad);
    }
    sink[tid] = acc;
    if (tid == 0) atomicAdd(&ctl->spin_total, spins);
}

int main() {
    Ctl* ctl; float* buf; int* xcc;
    CK(hipMalloc(&ctl, sizeof(Ctl))); CK(hipMalloc(&buf, 8 * 2 * 256 * 64 * sizeof(float))); CK(hipMalloc(&xcc, 256 * sizeof(int)));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 2000;
    for (int mode : {0, 1}) {
        for (int per_block : {4, 16, 64}) {
            if (mode == 0 && per_block > 16) continue;           // global vector: 256 * per_block floats
            CK(hipMemset(ctl, 0, sizeof(Ctl)));
            CK(hipMemset(buf, 0, 8 * 2 * 256 * 64 * sizeof(float)));
            CK(hipEventRecord(a, 0));
            hipLaunchKernelGGL(k_exchange, dim3(256), dim3(256), 0, 0, ctl, buf, iters, per_block, mode, xcc);
            CK(hipEventRecord(b, 0));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            Ctl h; CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
            std::vector<int> hx(256); CK(hipMemcpy(hx.data(), xcc, 256 * sizeof(int), hipMemcpyDeviceToHost));
            int cnt[16] = {0}; for (int v : hx) cnt[v & 15]++;
            printf("%s group, %3d floats/block (vector %5d floats): %7.3f us/edge  errors %u abort %u polls/edge %.1f  blocks per XCC:", mode ? "XCC   " : "global", per_block,
                   (mode ? 32 : 256) * per_block, ms * 1e3 / iters, h.errors, h.abort_flag, (double)h.spin_total / (256.0 * iters));
            for (int i = 0; i < 8; ++i) printf(" %d", cnt[i]);
            printf("\n");
        }
    }
    return 0;
}
