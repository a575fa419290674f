// Microbenchmark (round 3): what does a dependent launch cost when the NEXT launch is allowed to be resident -- weights already
// requested -- while the current one runs?  Models the batch-1 decode chain: 48 launches per step, 256 blocks x 256 threads,
// alternately 24 KB and 72 KB of weights per block (streamed once, non-temporal), input = 1024 tagged granules written by the
// previous launch (each wave sweeps a quarter), output = 4 granules per block.
//   mode 0  one stream, ordered launches (what the engine does today; the sweep finds everything published)
//   mode 1  two streams, launches alternate: launch i + 1 may start while launch i runs and waits on its granules
//   mode 2  one stream, hipExtAnyOrderLaunch (no barrier between the packets, in-order dispatch)
//   mode 3  three streams
//   mode 4  mode 0 captured into a hipGraph (one graph = one step)
//   mode 5  mode 1 captured into a hipGraph (two branches)
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/ubench_overlap.hip -o /tmp/ubench_overlap && /tmp/ubench_overlap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(r_)); exit(1); } } while (0)

template <int NLD, int SLEEP>
__global__ __launch_bounds__(256) void stage_kernel(const u32x4* __restrict__ W, const u64* gin, u64* gout, unsigned epoch_in, unsigned epoch_out,
                                                    unsigned* err, u64* stamps) {
    __shared__ float x[1024];
    const int tid = threadIdx.x, b = blockIdx.x, wv = tid >> 6, lane = tid & 63;
    const u64 t_start = __builtin_amdgcn_s_memrealtime();
    u32x4 w[NLD];
    const u32x4* p = W + (size_t)b * NLD * 256 + tid;
#pragma unroll
    for (int i = 0; i < NLD; ++i) w[i] = __builtin_nontemporal_load(p + i * 256);
    // sweep: wave wv polls granules [256 wv, 256 wv + 256), four per lane, all four in flight
    {
        unsigned spins = 0;
        bool ok[4] = {false, false, false, false};
        for (;;) {
            u64 g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = __hip_atomic_load((gu64*)gin + wv * 256 + j * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool all = true;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!ok[j] && (unsigned)(g[j] >> 32) == epoch_in) { ok[j] = true; x[wv * 256 + j * 64 + lane] = __uint_as_float((unsigned)g[j]); }
                all = all && ok[j];
            }
            if (__all(all)) break;
            if ((++spins & 63u) == 0 && (__builtin_amdgcn_s_memrealtime() - t_start > 2000000ull || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { atomicOr(err, 1u); break; }
            if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
        }
    }
    const u64 t_in = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int k = (i * 256 + tid) * 4 & 1023;
        acc = fmaf(__uint_as_float(w[i].x << 16), x[k], acc);
        acc = fmaf(__uint_as_float(w[i].y << 16), x[k + 1], acc);
        acc = fmaf(__uint_as_float(w[i].z << 16), x[k + 2], acc);
        acc = fmaf(__uint_as_float(w[i].w << 16), x[k + 3], acc);
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0)
        __hip_atomic_store((gu64*)gout + 4 * b + wv, ((u64)epoch_out << 32) | __float_as_uint(acc * 1e-3f + 0.5f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (stamps && tid == 0) { stamps[3 * b] = t_start; stamps[3 * b + 1] = t_in; stamps[3 * b + 2] = __builtin_amdgcn_s_memrealtime(); }
}

__global__ void init_kernel(u64* g, unsigned epoch) { g[blockIdx.x * 256 + threadIdx.x] = ((u64)epoch << 32) | 0x3f000000u; }

struct Bench {
    static constexpr int L = 48;
    uint4* W = nullptr; size_t w_u4 = 0;
    u64* g[2] = {nullptr, nullptr};
    unsigned* err = nullptr;
    u64* stamps = nullptr;             // [L][256][3] of the LAST step
    std::vector<size_t> off;
    hipStream_t st[3];
    hipEvent_t ev[L + 1];
    unsigned step = 0;

    void init() {
        size_t o = 0;
        for (int i = 0; i < L; ++i) { off.push_back(o); o += (size_t)256 * 256 * (i & 1 ? 18 : 6); }
        w_u4 = o;
        CK(hipMalloc(&W, w_u4 * sizeof(uint4)));
        CK(hipMemset(W, 0x3c, w_u4 * sizeof(uint4)));
        for (int i = 0; i < 2; ++i) { CK(hipMalloc(&g[i], 1024 * sizeof(u64))); CK(hipMemset(g[i], 0, 1024 * sizeof(u64))); }
        CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
        CK(hipMalloc(&stamps, (size_t)L * 256 * 3 * sizeof(u64)));
        for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    static unsigned epoch_of(unsigned step, int i) { return step * 64 + i + 1; }
    template <int SLEEP>
    void launch(int i, hipStream_t s, unsigned stp, int flags, bool stamp) {
        const unsigned ein = i == 0 ? epoch_of(stp - 1, L - 1) : epoch_of(stp, i - 1), eout = epoch_of(stp, i);
        u64* sp = stamp ? stamps + (size_t)i * 256 * 3 : nullptr;
        if (i & 1) hipExtLaunchKernelGGL((stage_kernel<18, SLEEP>), dim3(256), dim3(256), 0, s, nullptr, nullptr, flags, (const u32x4*)(W + off[i]), (const u64*)g[i & 1], g[(i + 1) & 1], ein, eout, err, sp);
        else hipExtLaunchKernelGGL((stage_kernel<6, SLEEP>), dim3(256), dim3(256), 0, s, nullptr, nullptr, flags, (const u32x4*)(W + off[i]), (const u64*)g[i & 1], g[(i + 1) & 1], ein, eout, err, sp);
        CK(hipGetLastError());
    }
    // one step of L launches in the given mode (eager modes 0..3)
    template <int SLEEP>
    void step_eager(int mode, bool stamp) {
        ++step;
        for (int i = 0; i < L; ++i) {
            hipStream_t s = mode == 1 ? st[i & 1] : mode == 3 ? st[i % 3] : st[0];
            launch<SLEEP>(i, s, step, mode == 2 ? hipExtAnyOrderLaunch : 0, stamp);
        }
    }
};

template <int SLEEP>
double run(Bench& B, int mode, int steps, bool report_stamps) {
    CK(hipDeviceSynchronize());
    // (re)publish the epoch the first launch waits for
    hipLaunchKernelGGL(init_kernel, dim3(4), dim3(256), 0, B.st[0], B.g[0], Bench::epoch_of(B.step, Bench::L - 1));
    CK(hipStreamSynchronize(B.st[0]));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0.f;
    if (mode <= 3) {
        for (int i = 0; i < 3; ++i) B.step_eager<SLEEP>(mode, false);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a, B.st[0]));
        if (mode == 1 || mode == 3) { CK(hipStreamWaitEvent(B.st[1], a, 0)); CK(hipStreamWaitEvent(B.st[2], a, 0)); }
        for (int i = 0; i < steps; ++i) B.step_eager<SLEEP>(mode, report_stamps && i == steps - 1);
        if (mode == 1 || mode == 3) {
            CK(hipEventRecord(B.ev[0], B.st[1])); CK(hipEventRecord(B.ev[1], B.st[2]));
            CK(hipStreamWaitEvent(B.st[0], B.ev[0], 0)); CK(hipStreamWaitEvent(B.st[0], B.ev[1], 0));
        }
        CK(hipEventRecord(b, B.st[0]));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, a, b));
    } else {
        // graphs: the epochs are arguments, so one graph per step index would be needed -- instead capture `steps` steps into ONE graph
        hipGraph_t gr; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(B.st[0], hipStreamCaptureModeThreadLocal));
        if (mode == 5) { CK(hipEventRecord(B.ev[0], B.st[0])); CK(hipStreamWaitEvent(B.st[1], B.ev[0], 0)); }
        const unsigned first = B.step + 1;
        for (int i = 0; i < steps; ++i) B.step_eager<SLEEP>(mode == 5 ? 1 : 0, false);
        if (mode == 5) { CK(hipEventRecord(B.ev[1], B.st[1])); CK(hipStreamWaitEvent(B.st[0], B.ev[1], 0)); }
        CK(hipStreamEndCapture(B.st[0], &gr));
        CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        // replaying needs the first launch's input epoch in place again, and the SAME epochs: publish the epoch before `first`
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(init_kernel, dim3(4), dim3(256), 0, B.st[0], B.g[0], Bench::epoch_of(first - 1, Bench::L - 1));
            // the granules of the previous replay carry the final epochs: a replay's kernels would find "their" epoch already there only for
            // the last step's launches -- harmless for timing the chain of the first steps, but reset both buffers to be exact
            CK(hipMemsetAsync(B.g[1], 0, 1024 * sizeof(u64), B.st[0]));
            CK(hipEventRecord(a, B.st[0]));
            CK(hipGraphLaunch(ge, B.st[0]));
            CK(hipEventRecord(b, B.st[0]));
            CK(hipStreamSynchronize(B.st[0]));
            CK(hipEventElapsedTime(&ms, a, b));
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
    }
    unsigned herr = 0;
    CK(hipMemcpy(&herr, B.err, 4, hipMemcpyDeviceToHost));
    if (herr) { printf("  !! a sweep timed out (mode %d)\n", mode); CK(hipMemset(B.err, 0, 4)); }
    if (report_stamps && mode <= 3) {
        std::vector<u64> h((size_t)Bench::L * 256 * 3);
        CK(hipMemcpy(h.data(), B.stamps, h.size() * sizeof(u64), hipMemcpyDeviceToHost));
        // per launch: first block start, median block start->input complete, last block end; then gaps between consecutive launches
        double sum_life = 0, sum_wait = 0, sum_period = 0; u64 prev_first = 0;
        for (int i = 0; i < Bench::L; ++i) {
            u64 first = ~0ull, last = 0; double wait = 0;
            for (int bk = 0; bk < 256; ++bk) {
                const u64* s = &h[((size_t)i * 256 + bk) * 3];
                if (s[0] < first) first = s[0];
                if (s[2] > last) last = s[2];
                wait += (double)(s[1] - s[0]);
            }
            sum_life += (double)(last - first); sum_wait += wait / 256;
            if (i) sum_period += (double)(first - prev_first);
            prev_first = first;
        }
        printf("    last step, in-kernel (10 ns ticks -> us): launch life %.2f us, mean block start->input complete %.2f us, start-to-start period %.2f us\n",
               sum_life / Bench::L / 100, sum_wait / Bench::L / 100, sum_period / (Bench::L - 1) / 100);
    }
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms * 1e3 / ((double)steps * Bench::L);
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 40;
    Bench B; B.init();
    const char* names[] = {"one stream, ordered", "two streams, alternating", "one stream, any-order flag", "three streams", "graph of ordered launches", "graph, two branches"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode : {0, 1, 2, 3, 4, 5}) {
            const double us = run<0>(B, mode, steps, rep == 1);
            printf("mode %d (%-28s): %.2f us per launch  (%.1f us per 48-launch step)\n", mode, names[mode], us, us * 48);
            fflush(stdout);
        }
    printf("-- polling with s_sleep 1 between sweeps\n");
    for (int mode : {0, 1, 2}) {
        const double us = run<1>(B, mode, steps, true);
        printf("mode %d (%-28s): %.2f us per launch  (%.1f us per 48-launch step)\n", mode, names[mode], us, us * 48);
    }
    return 0;
}
