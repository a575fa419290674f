// What does an in-launch all-gather of 16 / 32 / 64 KB into each of 256 resident blocks cost on MI355X?  (the batched
// decode layer's remaining seams -- y1 -> LayerNorm -> fc1 gathers 8 rows x 1024 fp32 = 32 KB into every block, relu(fc1) -> fc2 gathers a
// K quarter of 8 rows x 4096 16-bit values = 16 KB (64 KB without the K split) -- against the launch boundary they would replace: 1.4 us of
// gap + ~2.4 us until the first dependent byte)
//
// 256 blocks x 512 threads (one per CU, all resident), a 64-KB payload published in 256 slices of 256 B, every block gathers a window of G
// bytes (its K quarter / half / everything) per round, `iters` rounds back to back inside ONE launch; every gathered word is checked.
//   V0  8-byte granules {epoch, 4 B}      -- the engine's protocol (common.hpp ps_publish): relaxed agent-scope stores / polled loads
//   V1  16-byte granules {epoch, 12 B}    -- one dwordx4 sc1 store / load per granule (a lane's aligned 16-byte access is one request)
//   V2  flag + bulk                       -- sc1 data stores, s_waitcnt vmcnt(0), one sc1 flag store per slice; consumers poll the flags of
//                                            their window (one 4-byte load per lane), then fetch the window with 8-byte agent-scope loads, all in flight
// Reported: us per round (publish + gather + check), and the same loop with the gather removed (publish only) as the floor.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_allgather scripts/ubench_allgather.hip && scripts/ubench_allgather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
constexpr int NB = 256, NT = 512, PAYLOAD = 64 * 1024, SLICE = PAYLOAD / NB;      // 256 B per block
struct Ctl { unsigned errors, abort_flag; u64 polls; };

__device__ inline u32x4 ld16_sc1(const void* p) { u32x4 v; asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ inline void st16_sc1(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ inline unsigned word(int it, int idx) { return (unsigned)(it * 2654435761u) ^ (unsigned)(idx * 40503u + 17u); }     // payload word `idx` of round `it`

// buf layouts, all double-buffered by round parity: V0: u64[2][PAYLOAD / 4]; V1: u32x4[2][PAYLOAD / 12 + 2]; V2: data u32[2][PAYLOAD / 4] + flags u32[NB]
template <int V, bool GATHER, int G>
__global__ __launch_bounds__(NT) void k_allgather(Ctl* ctl, void* buf, unsigned* flags, int iters) {
    __shared__ unsigned s_bad;
    const int tid = threadIdx.x, b = blockIdx.x;
    // this block's window [w0, w0 + G) of the payload (bytes): the window its OWN slice lies in, so the producers of a window are exactly its
    // consumers (a closed group of G / 256 blocks spread over all XCDs) -- no block runs more than one round ahead of a block that still reads
    // its slice, and two buffers by round parity are enough (a slot holding round `it` is rewritten by round `it + 2` only)
    const int w0 = (b * SLICE / G) * G;
    unsigned bad = 0;
    u64 polls = 0;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    for (int it = 1; it <= iters; ++it) {
        const unsigned epoch = (unsigned)it;
        // ---- publish this block's slice: words b * 64 .. b * 64 + 63 ---------------------------------------------------------------------
        if constexpr (V == 0) {
            if (tid < SLICE / 4) {
                const int idx = b * (SLICE / 4) + tid;
                __hip_atomic_store((u64*)buf + (size_t)(it & 1) * (PAYLOAD / 4) + idx, ((u64)epoch << 32) | word(it, idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if constexpr (V == 1) {                           // granule g carries words 3 g .. 3 g + 2; a block owns the granules whose first word is its own
            const int g0 = (b * (SLICE / 4) + 2) / 3, g1 = ((b + 1) * (SLICE / 4) + 2) / 3;
            if (g0 + tid < g1) {
                const int g = g0 + tid;
                st16_sc1((u32x4*)buf + (size_t)(it & 1) * (PAYLOAD / 12 + 2) + g, u32x4{epoch, word(it, 3 * g), word(it, 3 * g + 1), word(it, 3 * g + 2)});
            }
        } else {
            unsigned* data = (unsigned*)buf + (size_t)(it & 1) * (PAYLOAD / 4);
            if (tid < SLICE / 16) {
                const int idx = b * (SLICE / 4) + tid * 4;
                st16_sc1(data + idx, u32x4{word(it, idx), word(it, idx + 1), word(it, idx + 2), word(it, idx + 3)});
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flags + b, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if constexpr (!GATHER) { __syncthreads(); continue; }
        // ---- gather the window, check every word: every request of a thread goes out before the first is looked at ------------------------------
        if constexpr (V == 0) {
            constexpr int NG = G / 4 / NT;                       // granules per thread: 8 / 16 / 32
            const u64* p = (const u64*)buf + (size_t)(it & 1) * (PAYLOAD / 4) + w0 / 4 + tid;
            u64 v[NG]; unsigned spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int c = 0; c < NG; ++c) { v[c] = __hip_atomic_load(p + c * NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = ok && (unsigned)(v[c] >> 32) == epoch; }
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000u) { ctl->abort_flag = 1; return; }
            }
            polls += spins;
#pragma unroll
            for (int c = 0; c < NG; ++c) bad += (unsigned)v[c] != word(it, w0 / 4 + tid + c * NT);
        } else if constexpr (V == 1) {
            const int g0 = (w0 / 4 + 2) / 3, g1 = ((w0 + G) / 4 + 2) / 3;          // = the granules the window's producers own
            for (int g = g0 + tid; g < g1; g += NT) {            // 3 / 6 / 11 per thread, one round trip each
                u32x4 v; unsigned spins = 0;
                while ((v = ld16_sc1((const u32x4*)buf + (size_t)(it & 1) * (PAYLOAD / 12 + 2) + g)).x != epoch) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 4000000u) { ctl->abort_flag = 1; return; }
                }
                polls += spins;
                bad += (v.y != word(it, 3 * g)) + (v.z != word(it, 3 * g + 1)) + (v.w != word(it, 3 * g + 2));
            }
        } else {
            constexpr int NF = G / SLICE, NL = G / 8 / NT;        // flags (= producer blocks) of the window; 8-byte loads per thread: 4 / 8 / 16
            const int f0 = w0 / SLICE;
            if (tid < NF) {
                unsigned spins = 0;
                while (__hip_atomic_load(flags + f0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {      // (monotonic: the producer may already be one round on)
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 4000000u) { ctl->abort_flag = 1; break; }
                }
                polls += spins;
            }
            __syncthreads();
            const u64* data = (const u64*)((const unsigned*)buf + (size_t)(it & 1) * (PAYLOAD / 4) + w0 / 4) + tid;
            u64 v[NL];
#pragma unroll
            for (int c = 0; c < NL; ++c) v[c] = __hip_atomic_load(data + c * NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // agent scope: not from a stale L2 line
#pragma unroll
            for (int c = 0; c < NL; ++c) {
                const int idx = w0 / 4 + 2 * (tid + c * NT);
                bad += ((unsigned)v[c] != word(it, idx)) + ((unsigned)(v[c] >> 32) != word(it, idx + 1));
            }
        }
        __syncthreads();                                          // the block moves on together (as a GEMM consuming the window would)
    }
    if (bad) atomicAdd(&s_bad, bad);
    __syncthreads();
    if (tid == 0) { if (s_bad) atomicAdd(&ctl->errors, s_bad); }
    if (polls) atomicAdd(&ctl->polls, polls);
}

template <int V>
int run(Ctl* ctl, void* buf, unsigned* flags, int iters, hipEvent_t a, hipEvent_t b) {
    const char* names[3] = {"V0  8-byte granules {epoch, 4 B}  ", "V1 16-byte granules {epoch, 12 B} ", "V2 flag + bulk (8-byte agent loads) "};
    float floor_ms = 0.f;
    CK(hipMemset(ctl, 0, sizeof(Ctl))); CK(hipMemset(buf, 0, 4 * PAYLOAD)); CK(hipMemset(flags, 0, NB * 4));
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((k_allgather<V, false, PAYLOAD>), dim3(NB), dim3(NT), 0, 0, ctl, buf, flags, iters);
    CK(hipEventRecord(b, 0)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&floor_ms, a, b));
    for (int G : {16 * 1024, 32 * 1024, 64 * 1024}) {
        float best = 1e9f; Ctl h{};
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(ctl, 0, sizeof(Ctl))); CK(hipMemset(buf, 0, 4 * PAYLOAD)); CK(hipMemset(flags, 0, NB * 4));
            CK(hipEventRecord(a, 0));
            if (G == 16 * 1024) hipLaunchKernelGGL((k_allgather<V, true, 16 * 1024>), dim3(NB), dim3(NT), 0, 0, ctl, buf, flags, iters);
            else if (G == 32 * 1024) hipLaunchKernelGGL((k_allgather<V, true, 32 * 1024>), dim3(NB), dim3(NT), 0, 0, ctl, buf, flags, iters);
            else hipLaunchKernelGGL((k_allgather<V, true, 64 * 1024>), dim3(NB), dim3(NT), 0, 0, ctl, buf, flags, iters);
            CK(hipEventRecord(b, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
            if (h.errors || h.abort_flag) { printf("%s window %2d KB: errors %u abort %u\n", names[V], G / 1024, h.errors, h.abort_flag); return 1; }
            if (ms < best) best = ms;
        }
        printf("%s window %2d KB per block: %6.2f us per round (publish only %5.2f), %4.1f polls per waiting thread and round, 0 wrong words in %d rounds x 256 blocks\n",
               names[V], G / 1024, best * 1e3 / iters, floor_ms * 1e3 / iters, (double)h.polls / ((double)iters * NB * 64), iters);
    }
    return 0;
}

int main() {
    Ctl* ctl; void* buf; unsigned* flags;
    CK(hipMalloc(&ctl, sizeof(Ctl))); CK(hipMalloc(&buf, 4 * PAYLOAD)); CK(hipMalloc(&flags, NB * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_allgather<0, true, 64 * 1024>, NT, 0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("%s: %d CUs, %d blocks of %d threads per CU -> %d resident (256 needed)\n", prop.name, prop.multiProcessorCount, occ, NT, occ * prop.multiProcessorCount);
    if (occ * prop.multiProcessorCount < NB) { printf("not enough resident blocks\n"); return 1; }
    const int iters = 2000;
    if (run<0>(ctl, buf, flags, iters, a, b)) return 1;
    if (run<1>(ctl, buf, flags, iters, a, b)) return 1;
    if (run<2>(ctl, buf, flags, iters, a, b)) return 1;
    printf("reference: a launch boundary of the decode graph costs 1.4 us of gap + ~2.4 us to the first dependent byte (DESIGN.md 3.5)\n");
    return 0;
}
