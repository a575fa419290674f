// Does a block's in-launch exchange have to queue behind its own bulk stream?  (rows_attn.hpp: with the first cache rounds requested before
// the q/k/v exchange, the exchange ends 1.6 us later per 64 KB requested -- profiles/r05_decode_step_timeline_b8_two_launches_early{2,3,4}.txt:
// the polling wave's vector loads and the publishing stores share the CU's vector-memory path with the other waves' stream and return in order.)
// The scalar unit has its own path to L2 (scalar data cache, bypassed with glc).  This microbenchmark models one exchange stage:
//   256 blocks x 512 threads, one per CU.  Per round: [waves 1..7 request STREAM KB of a large buffer (non-temporal 16-byte loads)], wave 0
//   publishes 64 granules {epoch, value} and polls 128 granules published by 16 other blocks (spread over all XCDs), hands them to the block
//   through LDS, barrier; the stream's data is consumed after the exchange.
// Variants: poll = vector (8-byte agent-scope atomic loads, the engine's) | scalar (s_load_dwordx16 glc); granule buffer = hipMalloc | uncached
// (hipDeviceMallocUncached); stream = none | requested before the exchange | requested after it.
// Reported per variant: us per round, us from the round's start to the end of wave 0's sweep (mean over blocks and rounds), polls per sweep.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_poll_under_stream scripts/ubench_poll_under_stream.hip && scripts/ubench_poll_under_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;
constexpr int NB = 256, NT = 512, GPB = 64;                       // granules published per block
struct Ctl { unsigned errors, abort_flag; u64 polls, sweep_ticks, sweeps; };

__device__ inline u32x4 ld_stream16(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
__device__ inline u32x16 sload16_glc(const void* p) {
    u32x16 v;
    asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

__device__ inline void sload16x2_glc(const void* p, const void* q, u32x16& a, u32x16& b) {
    asm volatile("s_load_dwordx16 %0, %2, 0x0 glc\n\ts_load_dwordx16 %1, %3, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(p), "s"(q) : "memory");
}

// POLL: 0 vector, 1 scalar (wave 0, one 64-byte load at a time), 2 scalar spread over the eight waves (two 64-byte loads in flight per wave).  STREAM: 0 none, 1 before the exchange, 2 after it.  KB: stream bytes per block and round / 1024.
template <int POLL, int STREAM, int KB>
__global__ __launch_bounds__(NT) void k_round(Ctl* ctl, u64* gran, const char* big, size_t big_bytes, unsigned* sink, int iters) {
    __shared__ unsigned got[128];
    __shared__ unsigned s_bad;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), b = blockIdx.x;
    constexpr int NL = STREAM ? KB * 1024 / 16 / (7 * 64) : 0;   // 16-byte loads per lane of waves 1..7
    unsigned bad = 0, acc = 0;
    u64 polls = 0, ticks = 0;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    for (int it = 1; it <= iters; ++it) {
        const unsigned epoch = (unsigned)it;
        const u64 t0 = __builtin_amdgcn_s_memrealtime();
        u32x4 sv[NL > 0 ? NL : 1];
        auto request = [&]() {
            if (w == 0) return;
            // a fresh 64-byte-strided region per (round, block): misses everywhere
            const size_t base = ((size_t)it * NB + b) * (size_t)(KB * 1024) % (big_bytes - (size_t)KB * 1024);
            const char* p = big + (base & ~(size_t)15) + ((size_t)(w - 1) * 64 + lane) * 16;
#pragma unroll
            for (int c = 0; c < NL; ++c) sv[c] = ld_stream16(p + (size_t)c * 7 * 64 * 16);
        };
        if constexpr (STREAM == 1) { request(); asm volatile("" ::: "memory"); }
        if (w == 0) {
            u64* mine = gran + (size_t)(it & 1) * NB * GPB + (size_t)b * GPB;
            __hip_atomic_store(mine + lane, ((u64)epoch << 32) | (unsigned)(b * 64 + lane) ^ epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // 16 producers x 8 consecutive granules: producer p_k = (b * 5 + 16 k + 3) % 256 (one per 16-block stripe: all XCDs), its granules 8 (b % 8) ..
            unsigned spins = 0;
            if constexpr (POLL == 2) {
            } else if constexpr (POLL == 0) {
                const int k0 = lane >> 3, e = lane & 7;          // lane polls (k0, e) and (k0 + 8, e)
                const int p1 = (b * 5 + 16 * k0 + 3) % NB, p2 = (b * 5 + 16 * (k0 + 8) + 3) % NB;
                const u64* a1 = gran + (size_t)(it & 1) * NB * GPB + (size_t)p1 * GPB + 8 * (b % 8) + e;
                const u64* a2 = gran + (size_t)(it & 1) * NB * GPB + (size_t)p2 * GPB + 8 * (b % 8) + e;
                u64 v1, v2;
                for (;;) {
                    v1 = __hip_atomic_load(a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v2 = __hip_atomic_load(a2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__all((unsigned)(v1 >> 32) >= epoch && (unsigned)(v2 >> 32) >= epoch)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 2000000u) { ctl->abort_flag = 1; return; }
                }
                if ((unsigned)(v1 >> 32) == epoch) bad += (unsigned)v1 != ((unsigned)(p1 * 64 + 8 * (b % 8) + e) ^ epoch);
                if ((unsigned)(v2 >> 32) == epoch) bad += (unsigned)v2 != ((unsigned)(p2 * 64 + 8 * (b % 8) + e) ^ epoch);
                got[lane] = (unsigned)v1; got[64 + lane] = (unsigned)v2;
            } else {
                for (int k = 0; k < 16; ++k) {
                    const int p = (b * 5 + 16 * k + 3) % NB;
                    const u64* a = gran + (size_t)(it & 1) * NB * GPB + (size_t)p * GPB + 8 * (b % 8);
                    u32x16 v;
                    for (;;) {
                        v = sload16_glc(a);
                        bool ok = true;
#pragma unroll
                        for (int e = 0; e < 8; ++e) ok = ok && v[2 * e + 1] >= epoch;
                        if (ok) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > 2000000u) { ctl->abort_flag = 2; return; }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (v[2 * e + 1] == epoch) bad += (lane == 0) && v[2 * e] != ((unsigned)(p * 64 + 8 * (b % 8) + e) ^ epoch);
                        if (lane == e) got[k * 8 + e] = v[2 * e];
                    }
                }
            }
            polls += spins;
            if constexpr (POLL != 2) ticks += __builtin_amdgcn_s_memrealtime() - t0;
        }
        if constexpr (POLL == 2) {                               // wave w: producers k = 2 w and 2 w + 1
            const int pa = (b * 5 + 16 * (2 * w) + 3) % NB, pb = (b * 5 + 16 * (2 * w + 1) + 3) % NB;
            const u64* aa = gran + (size_t)(it & 1) * NB * GPB + (size_t)pa * GPB + 8 * (b % 8);
            const u64* ab = gran + (size_t)(it & 1) * NB * GPB + (size_t)pb * GPB + 8 * (b % 8);
            u32x16 va, vb;
            unsigned spins = 0;
            for (;;) {
                sload16x2_glc(aa, ab, va, vb);
                bool ok = true;
#pragma unroll
                for (int e = 0; e < 8; ++e) ok = ok && va[2 * e + 1] >= epoch && vb[2 * e + 1] >= epoch;
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 2000000u) { ctl->abort_flag = 3; return; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (va[2 * e + 1] == epoch) bad += (lane == 0) && va[2 * e] != ((unsigned)(pa * 64 + 8 * (b % 8) + e) ^ epoch);
                if (vb[2 * e + 1] == epoch) bad += (lane == 0) && vb[2 * e] != ((unsigned)(pb * 64 + 8 * (b % 8) + e) ^ epoch);
                if (lane == e) { got[(2 * w) * 8 + e] = va[2 * e]; got[(2 * w + 1) * 8 + e] = vb[2 * e]; }
            }
            if (w == 7) { polls += spins; ticks += __builtin_amdgcn_s_memrealtime() - t0; }
        }
        __syncthreads();
        if constexpr (STREAM == 2) request();
        if constexpr (STREAM != 0) {
            if (w != 0) {
#pragma unroll
                for (int c = 0; c < NL; ++c) acc ^= sv[c].x ^ sv[c].w;
            }
        }
        acc ^= got[tid & 127];
        __syncthreads();
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;
    if (bad) atomicAdd(&s_bad, bad);
    __syncthreads();
    if (tid == 0 && s_bad) atomicAdd(&ctl->errors, s_bad);
    if (lane == 0 && (POLL == 2 ? w == 7 : w == 0)) { atomicAdd(&ctl->polls, polls); atomicAdd(&ctl->sweep_ticks, ticks); atomicAdd(&ctl->sweeps, (u64)iters); }
}

template <int POLL, int STREAM, int KB>
int run(const char* what, Ctl* ctl, u64* gran, const char* big, size_t big_bytes, unsigned* sink, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f; Ctl h{};
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctl, 0, sizeof(Ctl))); CK(hipMemset(gran, 0, 2 * NB * GPB * 8));
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((k_round<POLL, STREAM, KB>), dim3(NB), dim3(NT), 0, 0, ctl, gran, big, big_bytes, sink, iters);
        CK(hipEventRecord(b, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
        if (h.errors || h.abort_flag) { printf("%-64s errors %u abort %u (after %.1f ms)\n", what, h.errors, h.abort_flag, ms); return 0; }
        if (ms < best) best = ms;
    }
    printf("%-64s %6.2f us per round, sweep done %5.2f us after the round's start, %5.1f polls per sweep\n", what, best * 1e3 / iters,
           (double)h.sweep_ticks / (double)h.sweeps * 0.01, (double)h.polls / (double)h.sweeps);
    return 0;
}

int main() {
    Ctl* ctl; unsigned* sink; char* big; const size_t big_bytes = (size_t)4 << 30;
    CK(hipMalloc(&ctl, sizeof(Ctl))); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&big, big_bytes)); CK(hipMemset(big, 1, big_bytes));
    u64 *g_coarse, *g_unc;
    CK(hipMalloc(&g_coarse, 2 * NB * GPB * 8));
    CK(hipExtMallocWithFlags((void**)&g_unc, 2 * NB * GPB * 8, hipDeviceMallocUncached));
    const int iters = 2000;
    for (int pass = 0; pass < 2; ++pass) {
        u64* g = pass == 0 ? g_coarse : g_unc;
        printf("== granules in %s memory\n", pass == 0 ? "hipMalloc" : "hipDeviceMallocUncached");
        run<0, 0, 64>("vector polls, no stream", ctl, g, big, big_bytes, sink, iters);
        run<1, 0, 64>("scalar polls (s_load_dwordx16 glc), no stream", ctl, g, big, big_bytes, sink, iters);
        run<0, 2, 64>("vector polls, 64 KB stream requested AFTER the exchange", ctl, g, big, big_bytes, sink, iters);
        run<0, 1, 64>("vector polls, 64 KB stream requested BEFORE the exchange", ctl, g, big, big_bytes, sink, iters);
        run<1, 1, 64>("scalar polls, 64 KB stream requested BEFORE the exchange", ctl, g, big, big_bytes, sink, iters);
        run<0, 2, 112>("vector polls, 112 KB stream requested AFTER the exchange", ctl, g, big, big_bytes, sink, iters);
        run<0, 1, 112>("vector polls, 112 KB stream requested BEFORE the exchange", ctl, g, big, big_bytes, sink, iters);
        run<1, 1, 112>("scalar polls, 112 KB stream requested BEFORE the exchange", ctl, g, big, big_bytes, sink, iters);
        run<2, 0, 64>("scalar polls over 8 waves, no stream", ctl, g, big, big_bytes, sink, iters);
        run<2, 1, 64>("scalar polls over 8 waves, 64 KB stream BEFORE the exchange", ctl, g, big, big_bytes, sink, iters);
        run<2, 1, 112>("scalar polls over 8 waves, 112 KB stream BEFORE the exchange", ctl, g, big, big_bytes, sink, iters);
        run<2, 2, 112>("scalar polls over 8 waves, 112 KB stream AFTER the exchange", ctl, g, big, big_bytes, sink, iters);
    }
    return 0;
}
