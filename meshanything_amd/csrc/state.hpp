// Device-resident decode state: everything a captured decode step needs that changes from step to step lives
// here (not in kernel arguments), so one hipGraph of the step can be replayed for the whole generation.
#pragma once
#include <stdint.h>

namespace ma {

struct DecState {
    int t;             // tokens generated so far; the step being run produces token index t
    int pos;           // KV-cache row of the token fed this step (= cond_length + t - 1; rows < cond_length = prefix)
    int cur_tok;       // token fed this step (= token t-1)
    int finished;      // this row has emitted eos (it keeps stepping and emits pad, like generate())
    int suppress_eos;  // never pick eos
    int do_sample;     // 0 greedy, 1 top-k/top-p/multinomial
    int top_k;
    float top_p;
    unsigned long long seed;
    const float* uniforms;   // (max_new_tokens) uniforms of this row, or null -> hashed from seed
    int row;                 // batch row (decorrelates the hashed uniform stream)
    int max_new;
    const long long* forced; // (max_new_tokens) tokens of this row to feed instead of the picked ones (teacher forcing), or null
    float* logits_out;       // (max_new_tokens - logits_first, V) of this row: the logits of every step from logits_first on, or null
    int logits_first;        // first step kept in logits_out
    int pad_;
};

// weights of one OPT decoder layer inside the arena ([3p] OPTDecoderLayer; q/k/v fused into one [3H][H] matrix at load)
struct DecLayerPtrs {
    const void *qkv_w, *o_w, *fc1_w, *fc2_w;
    const float *qkv_b, *o_b, *fc1_b, *fc2_b, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
};

}  // namespace ma
