// Do gfx950's two fp32 -> fp16 conversions agree?  v_cvt_f16_f32 (scalar form: what `(_Float16)f` compiles to on its own) vs v_cvt_pk_f16_f32
// (what a pair of them compiles to).  All 2^32 bit patterns; prints the count and the first differing inputs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__global__ void k(unsigned long long* nbad, unsigned* first, unsigned base) {
    const unsigned u = base + blockIdx.x * blockDim.x + threadIdx.x;
    const float f = __uint_as_float(u);
    unsigned a, b;
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(a) : "v"(f));
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(b) : "v"(f));
    a &= 0xffffu;
    const unsigned lo = b & 0xffffu, hi = b >> 16;
    const bool nan = (u & 0x7fffffffu) > 0x7f800000u;
    if (!nan && (a != lo || a != hi)) {
        const unsigned long long i = atomicAdd(nbad, 1ull);
        if (i < 8) { first[3 * i] = u; first[3 * i + 1] = a; first[3 * i + 2] = b; }
    }
}
int main() {
    unsigned long long* nbad; unsigned* first;
    hipMalloc(&nbad, 8); hipMalloc(&first, 96); hipMemset(nbad, 0, 8); hipMemset(first, 0, 96);
    for (unsigned long long base = 0; base < (1ull << 32); base += (1ull << 28))
        hipLaunchKernelGGL(k, dim3(1u << 20), dim3(256), 0, 0, nbad, first, (unsigned)base);
    hipDeviceSynchronize();
    unsigned long long n; unsigned f[24];
    hipMemcpy(&n, nbad, 8, hipMemcpyDeviceToHost); hipMemcpy(f, first, 96, hipMemcpyDeviceToHost);
    printf("v_cvt_f16_f32 vs v_cvt_pk_f16_f32 over all non-NaN fp32 patterns: %llu differ\n", n);
    for (int i = 0; i < 8 && i < (int)n; ++i) { float x; memcpy(&x, &f[3 * i], 4); printf("  input %08x (%g): scalar %04x, packed %08x\n", f[3 * i], x, f[3 * i + 1], f[3 * i + 2]); }
    return 0;
}
