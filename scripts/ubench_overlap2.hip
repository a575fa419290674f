// Microbenchmark 2 (round 3): the batch-1 decode step as TWO interleaved launch sequences -- "first halves" (q/k/v + attention-like:
// 24 KB of weights per block, input = a 1024-granule all-gather, output = a 264-byte partial per block + one flag) on stream A, "second
// halves" (out_proj..fc2-like: 72 KB of weights per block, input = the 256 flags + 73 KB of partials read in bulk, output = 4 granules per
// block) on stream B -- with a join per step (the lm_head / pick of the real step), the step counter in device memory (so that a
// captured graph can be replayed), and the hand-over of the partials done two ways:
//   FENCE = 1  plain stores, flag stored with release semantics at agent scope (compiler: buffer_wbl2 sc1 + s_waitcnt + store sc1);
//              consumer: relaxed polls, then ONE acquire fence at agent scope (buffer_inv sc1), then plain 16-byte loads
//   FENCE = 0  partials stored as agent-scope relaxed atomics (write-through), s_waitcnt, relaxed flag; consumer: relaxed polls, then
//              agent-scope relaxed 8-byte atomic loads of the partials
// modes: 0 one stream, ordered (today's engine) | 1 two streams, eager | 4 one stream, one graph per step | 6 two streams, one graph per
// stream and step
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/ubench_overlap2.hip -o /tmp/ub2 && /tmp/ub2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(r_)); exit(1); } } while (0)

constexpr int L = 24;                 // layers: 2 launches each
constexpr int PART = 72;              // floats per partial record (66 used)
constexpr u64 TIMEOUT = 2000000ull;   // 20 ms

struct Args {
    const u32x4* W; const unsigned* step; int layer;
    u64* gran;            // [1024] y granules (second half -> next first half)
    float* ws;            // [256][PART] partials (first half -> second half)
    unsigned* flags;      // [256]
    unsigned* err; u64* stamps;
};

__device__ __forceinline__ bool expired(unsigned& spins, u64 t0, unsigned* err) {
    if ((++spins & 63u) != 0) return false;
    return __builtin_amdgcn_s_memrealtime() - t0 > TIMEOUT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// first half of layer `layer`: waits for the y granules of (step, layer - 1) [layer 0: of (step - 1, L - 1)]
template <int FENCE>
__global__ __launch_bounds__(256) void first_half(Args a) {
    __shared__ float x[1024];
    const int tid = threadIdx.x, b = blockIdx.x, wv = tid >> 6, lane = tid & 63;
    const u64 t_start = __builtin_amdgcn_s_memrealtime();
    const unsigned stp = *a.step;
    const unsigned e_in = a.layer == 0 ? (stp - 1) * 64 + 2 * (L - 1) + 2 : stp * 64 + 2 * (a.layer - 1) + 2, e_out = stp * 64 + 2 * a.layer + 1;
    u32x4 w[6];
    const u32x4* p = a.W + (size_t)b * 6 * 256 + tid;
#pragma unroll
    for (int i = 0; i < 6; ++i) w[i] = __builtin_nontemporal_load(p + i * 256);
    {
        unsigned spins = 0, pend = 0xfu;
        for (;;) {
            u64 g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { g[j] = (u64)e_in << 32; if ((pend >> j) & 1u) g[j] = __hip_atomic_load((gu64*)a.gran + wv * 256 + j * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((pend >> j) & 1u) {
                    const bool ok = (unsigned)(g[j] >> 32) == e_in;
                    if (ok) x[wv * 256 + j * 64 + lane] = __uint_as_float((unsigned)g[j]);
                    if (__all(ok)) pend &= ~(1u << j);
                }
            if (!pend) break;
            __builtin_amdgcn_s_sleep(1);
            if (expired(spins, t_start, a.err)) { atomicOr(a.err, 1u); break; }
        }
    }
    const u64 t_in = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int k = (i * 256 + tid) * 4 & 1023;
        acc = fmaf(__uint_as_float(w[i].x << 16), x[k], acc); acc = fmaf(__uint_as_float(w[i].y << 16), x[k + 1], acc);
        acc = fmaf(__uint_as_float(w[i].z << 16), x[k + 2], acc); acc = fmaf(__uint_as_float(w[i].w << 16), x[k + 3], acc);
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    x[wv] = acc;
    __syncthreads();
    if (wv == 0) {
        const float v = (x[0] + x[1] + x[2] + x[3]) * 1e-3f + 0.01f * lane;
        float* rec = a.ws + (size_t)b * PART;
        if (FENCE) {
            rec[lane] = v; if (lane < 2) rec[64 + lane] = v;
            if (lane == 0) __hip_atomic_store((gu32*)a.flags + b, e_out, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store((__attribute__((address_space(1))) float*)rec + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane < 2) __hip_atomic_store((__attribute__((address_space(1))) float*)rec + 64 + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);                    // every counter to zero: the write-through stores are acknowledged
            if (lane == 0) __hip_atomic_store((gu32*)a.flags + b, e_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (a.stamps && tid == 0) { a.stamps[3 * b] = t_start; a.stamps[3 * b + 1] = t_in; a.stamps[3 * b + 2] = __builtin_amdgcn_s_memrealtime(); }
}

// second half of layer `layer`: waits for the 256 flags of (step, layer), reads all partials, publishes 4 y granules
template <int FENCE>
__global__ __launch_bounds__(256) void second_half(Args a) {
    __shared__ float red[4];
    const int tid = threadIdx.x, b = blockIdx.x, wv = tid >> 6, lane = tid & 63;
    const u64 t_start = __builtin_amdgcn_s_memrealtime();
    const unsigned stp = *a.step;
    const unsigned e_in = stp * 64 + 2 * a.layer + 1, e_out = stp * 64 + 2 * a.layer + 2;
    u32x4 w[18];
    const u32x4* p = a.W + (size_t)b * 18 * 256 + tid;
#pragma unroll
    for (int i = 0; i < 18; ++i) w[i] = __builtin_nontemporal_load(p + i * 256);
    {
        unsigned spins = 0;
        for (;;) {
            const unsigned f = __hip_atomic_load((gu32*)a.flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (f == e_in) break;
            __builtin_amdgcn_s_sleep(1);
            if (expired(spins, t_start, a.err)) { atomicOr(a.err, 2u); break; }
        }
    }
    if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    const u64 t_in = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    if (FENCE) {
        const f32x4* r = reinterpret_cast<const f32x4*>(a.ws);
#pragma unroll
        for (int i = 0; i < 18; ++i) { const f32x4 v = r[i * 256 + tid]; acc += v.x + v.y + v.z + v.w; }       // 256 x 72 floats = 4608 x 16 B
    } else {
        const gu64* r = (const gu64*)a.ws;
#pragma unroll
        for (int i = 0; i < 36; ++i) { const u64 v = __hip_atomic_load(r + i * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); acc += __uint_as_float((unsigned)v) + __uint_as_float((unsigned)(v >> 32)); }
    }
#pragma unroll
    for (int i = 0; i < 18; ++i) acc = fmaf(__uint_as_float(w[i].x << 16), 1e-3f, acc) + __uint_as_float(w[i].w << 16) * 1e-3f;
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) __hip_atomic_store((gu64*)a.gran + 4 * b + wv, ((u64)e_out << 32) | __float_as_uint(acc * 1e-6f + 0.5f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)red;
    if (a.stamps && tid == 0) { a.stamps[3 * b] = t_start; a.stamps[3 * b + 1] = t_in; a.stamps[3 * b + 2] = __builtin_amdgcn_s_memrealtime(); }
}

__global__ void bump_kernel(unsigned* step) { *step += 1; }
__global__ void init_kernel(u64* g, unsigned epoch) { g[blockIdx.x * 256 + threadIdx.x] = ((u64)epoch << 32) | 0x3f000000u; }

struct Bench {
    u32x4* W = nullptr; std::vector<size_t> off1, off2;
    u64* gran = nullptr; float* ws = nullptr; unsigned* flags = nullptr; unsigned* err = nullptr; unsigned* step = nullptr; u64* stamps = nullptr;
    hipStream_t sa, sb; hipEvent_t ev_fork, ev_join;
    void init() {
        size_t o = 0;
        for (int l = 0; l < L; ++l) { off1.push_back(o); o += (size_t)256 * 256 * 6; off2.push_back(o); o += (size_t)256 * 256 * 18; }
        CK(hipMalloc(&W, o * sizeof(u32x4))); CK(hipMemset(W, 0x3c, o * sizeof(u32x4)));
        CK(hipMalloc(&gran, 1024 * 8)); CK(hipMalloc(&ws, 256 * PART * 4)); CK(hipMalloc(&flags, 256 * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&step, 4));
        CK(hipMalloc(&stamps, (size_t)2 * L * 256 * 3 * 8));
        CK(hipMemset(gran, 0, 1024 * 8)); CK(hipMemset(ws, 0, 256 * PART * 4)); CK(hipMemset(flags, 0, 256 * 4)); CK(hipMemset(err, 0, 4));
        CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
        CK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    }
    Args args(int l, bool second, bool stamp) {
        Args a{};
        a.W = W + (second ? off2[l] : off1[l]); a.step = step; a.layer = l; a.gran = gran; a.ws = ws; a.flags = flags; a.err = err;
        a.stamps = stamp ? stamps + (size_t)(2 * l + (second ? 1 : 0)) * 256 * 3 : nullptr;
        return a;
    }
    template <int FENCE> void first(int l, hipStream_t s, bool stamp) { hipLaunchKernelGGL(first_half<FENCE>, dim3(256), dim3(256), 0, s, args(l, false, stamp)); }
    template <int FENCE> void second(int l, hipStream_t s, bool stamp) { hipLaunchKernelGGL(second_half<FENCE>, dim3(256), dim3(256), 0, s, args(l, true, stamp)); }
    // one step, eager.  two == false: everything on sa in dependency order.  two == true: first halves on sa, second halves on sb
    template <int FENCE>
    void step_eager(bool two, bool stamp) {
        if (!two) {
            for (int l = 0; l < L; ++l) { first<FENCE>(l, sa, stamp); second<FENCE>(l, sa, stamp); }
        } else {
            CK(hipEventRecord(ev_fork, sa)); CK(hipStreamWaitEvent(sb, ev_fork, 0));
            for (int l = 0; l < L; ++l) { first<FENCE>(l, sa, stamp); second<FENCE>(l, sb, stamp); }
            CK(hipEventRecord(ev_join, sb)); CK(hipStreamWaitEvent(sa, ev_join, 0));
        }
        hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, sa, step);
    }
};

template <int FENCE>
double run(Bench& B, int mode, int steps) {
    CK(hipDeviceSynchronize());
    unsigned hstep = 1;
    CK(hipMemcpy(B.step, &hstep, 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(init_kernel, dim3(4), dim3(256), 0, B.sa, B.gran, 0u * 64 + 2 * (L - 1) + 2);     // what layer 0 of step 1 waits for
    CK(hipMemsetAsync(B.flags, 0, 256 * 4, B.sa));
    CK(hipStreamSynchronize(B.sa));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0.f;
    hipGraph_t g1 = nullptr, g2 = nullptr; hipGraphExec_t x1 = nullptr, x2 = nullptr;
    if (mode == 4) {
        CK(hipStreamBeginCapture(B.sa, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < L; ++l) { B.first<FENCE>(l, B.sa, false); B.second<FENCE>(l, B.sa, false); }
        hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, B.sa, B.step);
        CK(hipStreamEndCapture(B.sa, &g1)); CK(hipGraphInstantiate(&x1, g1, nullptr, nullptr, 0));
    } else if (mode == 6) {
        CK(hipStreamBeginCapture(B.sa, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < L; ++l) B.first<FENCE>(l, B.sa, false);
        CK(hipStreamEndCapture(B.sa, &g1)); CK(hipGraphInstantiate(&x1, g1, nullptr, nullptr, 0));
        CK(hipStreamBeginCapture(B.sb, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < L; ++l) B.second<FENCE>(l, B.sb, false);
        CK(hipStreamEndCapture(B.sb, &g2)); CK(hipGraphInstantiate(&x2, g2, nullptr, nullptr, 0));
    }
    auto one = [&](bool stamp) {
        if (mode == 0) B.step_eager<FENCE>(false, stamp);
        else if (mode == 1) B.step_eager<FENCE>(true, stamp);
        else if (mode == 4) CK(hipGraphLaunch(x1, B.sa));
        else {
            CK(hipEventRecord(B.ev_fork, B.sa)); CK(hipStreamWaitEvent(B.sb, B.ev_fork, 0));
            CK(hipGraphLaunch(x1, B.sa)); CK(hipGraphLaunch(x2, B.sb));
            CK(hipEventRecord(B.ev_join, B.sb)); CK(hipStreamWaitEvent(B.sa, B.ev_join, 0));
            hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, B.sa, B.step);
        }
    };
    for (int i = 0; i < 3; ++i) one(false);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, B.sa));
    for (int i = 0; i < steps; ++i) one(mode <= 1 && i == steps - 1);
    CK(hipEventRecord(b, B.sa));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, a, b));
    unsigned herr = 0; CK(hipMemcpy(&herr, B.err, 4, hipMemcpyDeviceToHost));
    if (herr) { printf("  !! a wait timed out (mode %d, code %u)\n", mode, herr); CK(hipMemset(B.err, 0, 4)); }
    if (mode <= 1) {
        std::vector<u64> h((size_t)2 * L * 256 * 3);
        CK(hipMemcpy(h.data(), B.stamps, h.size() * 8, hipMemcpyDeviceToHost));
        double life[2] = {0, 0}, wait[2] = {0, 0};
        for (int i = 0; i < 2 * L; ++i) {
            u64 first = ~0ull, last = 0; double w = 0;
            for (int bk = 0; bk < 256; ++bk) { const u64* s = &h[((size_t)i * 256 + bk) * 3]; if (s[0] < first) first = s[0]; if (s[2] > last) last = s[2]; w += (double)(s[1] - s[0]); }
            life[i & 1] += (double)(last - first); wait[i & 1] += w / 256;
        }
        printf("    in-kernel, last step: first half lives %.2f us (input complete %.2f us after block start), second half %.2f us (%.2f us)\n",
               life[0] / L / 100, wait[0] / L / 100, life[1] / L / 100, wait[1] / L / 100);
    }
    if (x1) { CK(hipGraphExecDestroy(x1)); CK(hipGraphDestroy(g1)); }
    if (x2) { CK(hipGraphExecDestroy(x2)); CK(hipGraphDestroy(g2)); }
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms * 1e3 / steps;
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 40;
    Bench B; B.init();
    const char* names[] = {"one stream, ordered, eager", "two streams, eager", "", "", "one stream, graph per step", "", "two streams, graph per stream"};
    for (int rep = 0; rep < 2; ++rep) {
        for (int mode : {0, 1, 4, 6}) { const double us = run<1>(B, mode, steps); printf("FENCE=1 mode %d (%-30s): %7.1f us per step = %.2f us per layer\n", mode, names[mode], us, us / L); fflush(stdout); }
        for (int mode : {0, 1, 4, 6}) { const double us = run<0>(B, mode, steps); printf("FENCE=0 mode %d (%-30s): %7.1f us per step = %.2f us per layer\n", mode, names[mode], us, us / L); fflush(stdout); }
    }
    return 0;
}
