// One launch per decoder layer boundary, batch-1 launch chain (bf16 policy, 350M layer shape): the second half of layer l
// (merge of the split-KV partials, out_proj, LayerNorm 1, fc1, fc2: oproj_fc1.hpp) and the first half of layer l + 1 (LayerNorm 2
// of layer l, q/k/v projection, split-KV attention: qkv_attn.hpp) in ONE launch ([3p] OPTDecoderLayer, reached from
// shape_opt.py:403-410).  The vector between them, y2 = h1 + W2 f + b2 (1024 values), is all-gathered inside the launch with the
// same tagged granules as y1 and relu(fc1) -- a quarter of the sweep per wave -- instead of a kernel boundary + a dependent fetch.
// The only kernel boundary left per layer is the one in front of the partial merge (16.9 K floats per row: too many to gather).
// Block b of the 256 does the work of block b of oproj_fc1_kernel and of block (c = b % 16, h = b / 16) of qkv_attn_kernel: the
// same device functions, so the same bits (tests/test_gpu_persist.py).  MEASURED: break-even -- 415 vs 415 us per step at kv 300,
// 467 vs 463 at kv 3800 (profiles/r02_ab_exchange_and_load_placement.txt): the y2 gather costs what the boundary cost, once layer
// l + 1's weight rows are requested under the relu(fc1) gather (without that: +4 %; with the cache rows requested before the q/k/v
// publish as well: +2 %).  Opt-in (`fuse_layer`), the default chain keeps two launches per layer.  Hazards inside the launch: the layer-l reads of `res` / the partials finish before a
// block publishes y1, and nothing of layer l + 1 is written before a block has seen all of y1 -- every block has published by then.
#pragma once
#include "../oproj_fc1.hpp"
#include "../qkv_attn.hpp"

namespace ma {

struct LayerFusedArgs {
    OprojFc1Args o;          // layer l (W2 set: fc2 inside)
    QkvAttnArgs q;           // layer l + 1; q.x unused (the input arrives through `gran3`), q.ln_* = layer l's final LayerNorm
    u64* gran3;              // [batch][hidden] granules of y2
};

__global__ __launch_bounds__(256) void layer_fused_kernel(LayerFusedArgs a) {
    __shared__ __attribute__((aligned(16))) float ynext[1024];
    const int c = blockIdx.x, h = blockIdx.y, brow = blockIdx.z;
    QkvOperands op;                                      // layer l + 1's weight rows: requested while relu(fc1) is being gathered
    oproj_fc1_body<4, true, true, bf16_t>(a.o, h * ATTN_NCHUNK + c, brow, ynext, a.gran3, [&] { qkv_load_operands<PRO_LN>(a.q, c, h, op); asm volatile("" ::: "memory"); });
    qkv_attn_body<PRO_LN, true>(a.q, c, h, gridDim.y, brow, ynext, op);
}

inline hipError_t launch_layer_fused(const LayerFusedArgs& a, int hidden, int ffn, int heads, int batch, hipStream_t s) {
    if (hidden != 1024 || ffn != 4096 || heads * 64 != hidden || heads * ATTN_NCHUNK != hidden / 4 || !a.o.W2 || !a.q.ln_g) return hipErrorInvalidValue;
    hipLaunchKernelGGL(layer_fused_kernel, dim3(ATTN_NCHUNK, heads, batch), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace ma
