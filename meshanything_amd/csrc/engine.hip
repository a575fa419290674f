// MeshAnything inference engine for MI355X (gfx950): host side + C ABI (include/meshanything_amd.h).
//
// Phases (reference: MeshAnything.forward, MeshAnything/models/meshanything.py:134-176):
//   encode      point cloud -> 257x768 latents -> 257x1024 prefix        (MFMA GEMMs + LDS-tiled attention)
//   prefill     24 OPT layers over the prefix, fills the KV cache        (same kernels, causal)
//   decode      <= 7201 steps, each = 123 launches (batch 1: embed + 24 x [qkv | attention | out_proj+merge | fc1 | fc2] + lm_head
//               + pick) replayed from ONE hipGraph per batch size; all step-varying scalars live in per-row device DecState
//               records, so the graph never changes.  Batches of >= 4 rows (bf16) run the same chain as skinny MFMA GEMMs.
//   detokenize  codebook gather + 6 BERT layers over 1057 tokens + per-coordinate argmax
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <dlfcn.h>
#include <exception>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/meshanything_amd.h"
#include "attn.hpp"
#include "attn2.hpp"
#include "attn_decode.hpp"
#include "common.hpp"
#include "dense_ops.hpp"
#include "gemm.hpp"
#include "gemm_decode.hpp"
#include "gemm_tile.hpp"
#include "gemm256.hpp"
#include "gemv.hpp"
#include "misc.hpp"
#include "oproj_fc1.hpp"
#include "qkv_attn.hpp"
#include "rows_attn.hpp"
#include "rows_mlp.hpp"
#include "state.hpp"
// MA_EXPERIMENTAL (build.py: MA_EXPERIMENTAL=1): the measured-and-rejected decode-step forms -- the persistent one-launch step
// (persist.hpp), the rows-looped two-launch layer (rows_fused.hpp) and the layer-pair launch (layer_fused.hpp); DESIGN.md records why
// each lost.  They are evidence, not product: the shipped library does not contain them, their tests skip without the flag.
#ifdef MA_EXPERIMENTAL
#include "experimental/persist.hpp"
#include "experimental/rows_fused.hpp"
#include "experimental/layer_fused.hpp"
#endif
#include "weights.hpp"

using namespace ma;

namespace {

thread_local std::string g_create_error;

struct MaError : std::exception {
    int code; std::string msg;
    MaError(int c, std::string m) : code(c), msg(std::move(m)) {}
    const char* what() const noexcept override { return msg.c_str(); }
};

// an in-launch exchange of a fused decode launch gave up (its blocks were not all resident): generate() answers by switching this
// engine to the five-launch chain (no co-residency needed, same bits) and running the generation again
struct ChainTimeout : MaError {
    ChainTimeout(std::string m) : MaError(MA_ERR_HIP, std::move(m)) {}
};

#define HIP_CHECK(expr)                                                                                      \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess)                                                                                \
            throw MaError(MA_ERR_HIP, std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// The engine's 16-bit format (ma_config.dtype: MA_DTYPE_BF16 | MA_DTYPE_F16) as a compile-time type of the kernels (common.hpp H16):
// H16_CALL evaluates an expression, H16_DO runs statements, with HT = f16_t or bf16_t.
#define H16_CALL(hdt, HT, ...) ((hdt) == MA_DTYPE_F16 ? [&] { using HT = f16_t; return __VA_ARGS__; }() : [&] { using HT = bf16_t; return __VA_ARGS__; }())
#define H16_DO(hdt, HT, ...) do { if ((hdt) == MA_DTYPE_F16) { using HT = f16_t; __VA_ARGS__; } else { using HT = bf16_t; __VA_ARGS__; } } while (0)
// 16-bit format of the kernel-level entry points that carry no dtype argument (ma_op_set_half_dtype)
thread_local int g_op_hdt = MA_DTYPE_BF16;

// roctx ranges around the phases of the hot path (tracing): resolved lazily from libroctx64.so, active only when
// MA_ROCTX=1 is set in the environment (rocprofv3 --marker-trace then shows encode / prefill / decode / detokenize as ranges)
struct RoctxRange {
    typedef int (*push_fn)(const char*);
    typedef int (*pop_fn)();
    static push_fn& push() { static push_fn f = nullptr; return f; }
    static pop_fn& pop() { static pop_fn f = nullptr; return f; }
    static bool enabled() {
        static int state = -1;
        if (state < 0) {
            state = 0;
            const char* v = getenv("MA_ROCTX");
            if (v && v[0] == '1') {
                void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
                if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
                if (h) {
                    push() = reinterpret_cast<push_fn>(dlsym(h, "roctxRangePushA"));
                    pop() = reinterpret_cast<pop_fn>(dlsym(h, "roctxRangePop"));
                    state = push() && pop() ? 1 : 0;
                }
            }
        }
        return state == 1;
    }
    bool on;
    explicit RoctxRange(const char* name) : on(enabled()) { if (on) push()(name); }
    ~RoctxRange() { if (on) pop()(); }
};

}  // namespace

struct ma_engine {
    ma_config cfg{};
    int device = 0;
    Layout L;
    PackState ps;
    bool weights_ready = false;
    std::string err;
    char* arena = nullptr;
    void* stage = nullptr; size_t stage_bytes = 0;      // upload staging of ma_engine_load_weights (freed by finalize)

    int T = 0, V = 0, maxnew = 0, maxseq = 0, nf = 0, S = 0;
    bool bf16 = true;                // a 16-bit policy (bf16 OR fp16; the name is historical): 16-bit weights + KV, GEMM / attention inputs rounded
    int hdt = MA_DTYPE_BF16;         // ... and which one: MA_DTYPE_BF16 | MA_DTYPE_F16 (the type tag of the 16-bit kernels, H16_CALL)
    size_t kv_elem = 2;
    char* kv = nullptr;              // [max_batch][layers][2][heads][maxseq][64] of KT
    size_t kv_plane = 0;             // bytes of one K (or V) plane of one layer
    size_t kv_row_bytes = 0;         // bytes of one batch row's planes (2 * layers * kv_plane)

    // decode-step buffers
    float *d_e = nullptr, *d_q = nullptr, *d_ypre1 = nullptr, *d_ypre2 = nullptr, *d_h0 = nullptr, *d_h1 = nullptr,
          *d_ffn = nullptr, *d_logits = nullptr, *d_part = nullptr, *d_pval = nullptr;
    int* d_pidx = nullptr;
    int n_parts = 0;
    DecState* d_st = nullptr;        // one record per batch row
    DecState* h_state = nullptr;     // pinned (max_batch)
    long long* h_tokens = nullptr;   // pinned (max_batch * maxnew)
    std::vector<DecLayerPtrs> dl;
    std::map<int, hipGraph_t> graph;           // one captured decode step per batch size
    std::map<int, hipGraphExec_t> gexec;
    hipStream_t cap_stream = nullptr;   // capture happens on a private stream (the caller's may be the legacy null stream)
    // row groups of a batched step (decode_groups): group g steps its rows on grp_stream[g], forked from / joined to the caller's stream
    int opt_decode_groups = 1;          // 1 = one group (default: measured faster), G = that many (decode_group_count caps it)
    std::vector<hipStream_t> grp_stream; std::vector<hipEvent_t> grp_done; hipEvent_t grp_fork = nullptr;

    // dense-phase workspace: dense_rows samples stacked along the rows.  w_* / p_*: fp32 streams; a_*: activation tensors
    // (dense_ops.hpp: act_elem = 2 bytes under the bf16 policy, 4 under the exact policy)
    std::vector<void*> allocs;
    // precision of the dense phase being enqueued (DenseScope below): MA_DTYPE_F32 or the engine's 16-bit type.  The point encoder
    // (ma_encode: encode_latents + process_point_feature, and the detokenizer's projection of the latents) runs in fp32 under a 16-bit
    // policy when cfg.enc_exact is set -- the north star's 1e-5 on encoder activations in the benchmarked mode; prefill and the
    // detokenizer's BERT stack follow the policy dtype.
    bool dense16 = true;
    bool enc_exact = false;          // encoder weights are fp32 arena entries and the encoder's activations fp32 (always true under the fp32 policy)
    size_t act_elem = 2;
    int dense_rows = 1, prefill_rows = 1;
    float *w_data = nullptr, *w_lat = nullptr, *w_lat2 = nullptr, *w_pf = nullptr, *w_x = nullptr, *w_y = nullptr, *w_fe = nullptr, *w_logit = nullptr;
    float *p_h = nullptr, *p_y = nullptr;
    long p_y_part_stride = 0;        // p_y holds up to 4 partial sums of a GEMM split along K (gemm256.hpp GemmSplitK), this many floats apart, for the small prefills that use it
    int opt_fuse_ln = 0;             // (MA_EXPERIMENTAL builds; measured, not kept) prefill: the two LayerNorms of a layer finished inside the out_proj / fc2 GEMMs where those run on whole 256 x 256 tiles (gemm256.hpp LNF form; needs the grid resident like every in-launch exchange: chain_resident)
    u64* d_ln_gran = nullptr; size_t ln_gran_tiles = 0; unsigned ln_epoch = 0;
    int opt_prefill_tail = 2;        // 16-bit prefill of >= 8 samples: the M % 256 rows behind the 256-row tiles run as a chain of their own on a second stream (prefill());
                                     // 2 (default): that stream has the lowest priority -- HIP keeps a pool of hardware queues per priority, so it can never land on the
                                     // hardware queue of the main stream (or of the application's default-priority streams), where it would run IN LINE with them; 1: default priority
    hipStream_t tail_stream_low = nullptr;      // (prefill_tail = 2)
    hipStream_t tail_stream = nullptr; hipEvent_t tail_fork = nullptr, tail_join = nullptr; std::vector<hipEvent_t> tail_kv;      // its stream; per layer: "the main rows' K / V are in the planes"
    void* a_patt_tail = nullptr; bf16_t* a_vt_tail = nullptr; size_t vt_tail_elems = 0;      // its attention output (64 rows) and V^T workspace (one sample)
    int opt_gemm_splitk = 2;         // prefill fc2 (1) and out_proj (2, default) of small batches as 4 | 2 partial sums along K, added up by the LayerNorm that follows (0: never; A/B)
    void *a_feat = nullptr, *a_dataln = nullptr, *a_kv = nullptr, *a_q = nullptr, *a_ln = nullptr, *a_qkv = nullptr, *a_att = nullptr, *a_mlp = nullptr,
         *a_cat = nullptr, *a_mean = nullptr, *a_fein = nullptr, *a_x = nullptr, *a_ph = nullptr, *a_pqkv = nullptr, *a_patt = nullptr, *a_pffn = nullptr;
    unsigned char* w_mask = nullptr;
    float *w_latents = nullptr, *w_prefix = nullptr, *w_coords = nullptr;   // ma_forward intermediates (max_batch rows)
    long long *w_tokens = nullptr, *w_ids = nullptr;

    // options
    int opt_gemm_impl = 0;           // 0 MFMA, 1 VALU reference kernel
    int opt_prefill_stepwise = 0;    // 1: run the prefix through the decode-step chain row by row (debug cross-check)
    int opt_profile_batch = 1;       // batch size ma_profile_decode times (<= max_batch)
    int opt_mfma_min_batch = 4;      // bf16 policy: batches of at least this many rows take the MFMA skinny-GEMM decode path
    int opt_attn_pair = 1;               // final-form attention below 12 rows: two blocks per (row, head)
    int opt_mfma_fc2_ksplit = 0;         // blocks along K of the batched fc2 GEMM: 0 = gemm_dec_ksplit (4) | 1 | 2 | 4
    int opt_mfma_ln_waves = 0;           // waves per block of the LayerNorm-folded skinny GEMM: 0 = by batch (8 for 5..8 rows, else 4) | 4 | 8
    int opt_mfma_fold_ln = 1;            // MFMA decode path, small batches: LayerNorm prologues inside the consuming GEMMs (up to two launches fewer per layer)
    int opt_mfma_fold_fc1_max = 8, opt_mfma_fold_qkv_max = 8;       // largest batch for which LN1 (in front of fc1) / LN2 (in front of q/k/v) is folded
    int opt_attn_final_min_batch = 8;    // MFMA decode path: from this many rows on, one attention block per (row, head) writes the final output (no merge launch)
    int opt_attn_final_waves = 0;        // waves per block of that form: 0 = 4 from 12 rows on, 8 below; or 4 | 8 | 16
    int opt_fuse_layer = 0;              // second half of layer l + first half of layer l + 1 in one launch (layer_fused.hpp)
    int opt_fuse_fc2 = 1;                // fc2 inside the out_proj + fc1 launch (second in-launch all-gather, 4096 values)
    int opt_oproj_fc1_sweep_waves = 4;   // fused out_proj + fc1 launch: waves per block polling the y1 granules (each its own quarter)
    int opt_gemm_xcd_swizzle = 1;    // dense GEMM: hand the tiles out XCD-aware (gemm_tile.hpp)
    int opt_attn_impl = 2;           // bf16 dense attention: 2 = swapped-operand 32x32x16 kernel on packed V^T (attn2.hpp), 1 = attention_mfma_kernel (attn.hpp)
    bf16_t* a_vt = nullptr; size_t vt_elems = 0;      // its V^T workspace
    int opt_attn_rowwave = 1;        // MFMA decode path below that: one wave per (row, head, chunk) (1) or one block (0)
    // persistent decode step (persist.hpp): batch 1, bf16, greedy, 350M-shaped layers on a 256-CU device
    int opt_rows_fused = 0;          // 2 .. 8 rows: the two-launch layer with the rows looped inside the 256 blocks (rows_fused.hpp).  Opt-in: bit-identical
                                     // to batch-1 runs, 51 launches per step, but 1.4-1.9x SLOWER than the matrix-core chain (profiles/r03_rows_fused_*)
    int opt_rows_fused_min = 4;      // smallest batch that takes it (below: the batch-1 fused launches with the rows in the grid)
    bool rf_ok = false;              // its second launch (130 KB of LDS) can be resident on every CU of this device
    u64* d_part_gran = nullptr;      // [max_batch][heads][16][66] granules: its in-launch split-KV partial exchange
    int opt_qkv_xcd_local = 1;       // fused q/k/v + attention launch: the 16 blocks of a head on one XCD (qkv_attn.hpp qkv_block_role)
    int opt_fuse_qkv_attn = 1;       // launch chain, any policy, hidden 1024: q/k/v projection and decode attention in ONE launch (qkv_attn.hpp)
    u64* d_qkv_gran = nullptr;       // its exchange buffer: [max_batch][3 hidden] granules
    int opt_fuse_oproj_fc1 = 1;      // ... and out_proj (+ partial merge) + LayerNorm + fc1 in ONE launch (oproj_fc1.hpp)
    u64* d_y1_gran = nullptr;        // [max_batch][hidden] granules
    unsigned long long* d_attn_pair_gran = nullptr;      // [max_batch][heads][ATTN_PAIR_GRANULES]: hand-over of the two-block final-form attention
    int opt_fuse_rows_attn = 1;      // matrix-core decode path at 8 rows: LayerNorm + q/k/v + attention + out_proj in ONE launch (rows_attn.hpp)
    bool rows_ok = false;            // the two 8-row launches (256 blocks of 512 threads each) can be resident all at once on this device
    int opt_qkv_to_cache = 1;        // prefill (16-bit policies): the q|k|v GEMM writes K / V into the cache planes itself where it can (gemm256.hpp KV form); 0: always by kv_fill_rows_kernel (A/B)
    int opt_rows_attn_early = 6;     // rows_attn.hpp: when the first cache rounds are requested (A/B, see the kernel): 5 = the q/k/v sweep by scalar loads (waves 0 .. 3), two rounds by the waves 4 .. 7 meanwhile; 6 = 5 + rounds wholly below the newest position run without masks; 3 = one round behind the q/k/v MFMAs, sweep by vector loads
    int opt_rows_mlp_prefetch = 0;   // rows_mlp.hpp step F (measured, not kept: 0 = off): the next layer's first operands pulled into L2 by the blocks that idle during step E -- 1 | 2 rounds, 8 = weights only, 9 = half a round
    unsigned* d_pf_sink = nullptr;
    int opt_rows_mlp_ln2 = 1;        // rows_mlp.hpp step E: LayerNorm 2 finished in the MLP launch (the next q/k/v starts from 16-bit rows)
    u64* d_rm_y2_gran = nullptr;     // its exchange: [max_batch][RM_Y2_GRANULES]
    int opt_fuse_rows_mlp = 1;       // ... and LayerNorm 1 + fc1 + fc2 in ONE launch (rows_mlp.hpp; its relu(fc1) exchange uses d_ffn_gran)
    u64 *d_ra_qkv_gran = nullptr, *d_ra_out_gran = nullptr;      // its exchanges: [max_batch][RA_QKV_GRANULES], [max_batch][RA_OUT_GRANULES]
    u64* d_y2_gran = nullptr;        // [max_batch][hidden] granules (y2 handed to the next layer inside a launch)
    u64* d_ffn_gran = nullptr;       // [max_batch][ffn] granules (fc2 in the out_proj + fc1 launch)
    unsigned* d_chain_err = nullptr; unsigned* h_chain_err = nullptr;
    int opt_decode_impl = 0;         // 0: chain of launches; 1: one persistent launch per step (when eligible)
    int n_cus = 0;
    bool persist_shape = false;      // shape / device eligibility (fixed at creation)
    bool chain_resident = false;     // the fused launches' 256 blocks fit on the device at once, with margin (their in-launch exchange needs that)
    long resident_blocks = 0;        // 256-thread blocks of the fused launches the device holds at once (CUs x (occupancy - 1))
    int xchg_last_code = 0;          // the error word of the last generation that fell back (bits of the sweeps that gave up)
    int chain_fallbacks = 0;         // generations that were re-run on the five-launch chain after an exchange timed out
    int gens_since_fallback = 0;     // clean generations on the five-launch chain since then: after CHAIN_REARM_AFTER of them the fused launches get another try
    bool embtab_ready = false;
    DecLayerPtrs* d_layers = nullptr;
    u64* d_gran = nullptr; unsigned* d_serial = nullptr; unsigned* d_err = nullptr; unsigned* h_err = nullptr;
    float* d_embtab = nullptr;       // [codebook_size][hidden] fp32: input_layer(codebook row) + bias, built by the chain's own GEMV
    u64* d_ptrace = nullptr;
    bf16_t *d_xb = nullptr, *d_ffb = nullptr;      // bf16 activations of the batched path: [max_batch][hidden], [max_batch][ffn]
    float *d_ks_o = nullptr, *d_ks_f = nullptr;    // split-K partials of out_proj / fc2: [4][max_batch][hidden]

    template <typename Tp> Tp* dmalloc(size_t n) {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(Tp)));
        allocs.push_back(p);
        return reinterpret_cast<Tp*>(p);
    }
    const void* P(const std::string& name) const {
        auto it = L.entry_by_name.find(name);
        if (it == L.entry_by_name.end()) throw MaError(MA_ERR_INVALID, "internal: no arena entry " + name);
        return arena + L.entries[it->second].offset;
    }
    const float* PF(const std::string& name) const { return reinterpret_cast<const float*>(P(name)); }
    char* kplane(int row, int layer) const { return kv + (size_t)row * kv_row_bytes + (size_t)(2 * layer) * kv_plane; }
    char* vplane(int row, int layer) const { return kv + (size_t)row * kv_row_bytes + (size_t)(2 * layer + 1) * kv_plane; }
};

namespace {

const std::string SM = "point_encoder.model.shape_model.", DEC = "transformer.model.decoder.", TOK = "tokenizer.";

// ------------------------------------------------------------------------------------------------ dense-phase helpers
// Buffer kinds (dense_ops.hpp): fp32 "stream" tensors (float*) and "activation" tensors (void*, element = e->act_elem bytes:
// bf16 under the bf16 policy, fp32 under the exact policy).  All dense phases run a CHUNK of nb <= e->dense_rows samples at
// once, stacked along the GEMM rows.
inline void* aoff(ma_engine* e, void* p, size_t elems) { return reinterpret_cast<char*>(p) + elems * e->act_elem; }
inline const void* aoff(ma_engine* e, const void* p, size_t elems) { return reinterpret_cast<const char*>(p) + elems * e->act_elem; }

// the precision of the launches enqueued while it lives (see ma_engine::dense16)
struct DenseScope {
    ma_engine* e; bool saved16; size_t saved_elem;
    DenseScope(ma_engine* e_, bool use16) : e(e_), saved16(e_->dense16), saved_elem(e_->act_elem) { e->dense16 = use16; e->act_elem = use16 ? 2 : 4; }
    ~DenseScope() { e->dense16 = saved16; e->act_elem = saved_elem; }
    DenseScope(const DenseScope&) = delete; DenseScope& operator=(const DenseScope&) = delete;
};

struct GemmOut {                       // exactly one of: fp32 stream output | activation output
    float* c32 = nullptr; void* act = nullptr; int ld = 0; RowMap map{0, 0, 0};
};
// C = act(A . W^T + bias) + R, A an activation tensor (M, lda).  r_mod > 0: the residual row is m % r_mod.
// kv (optional, 16-bit phases): the K / V columns of a fused q|k|v projection may go straight to these KV-cache planes (GemmTArgs::kv_*);
// kv->rows_done = the leading rows for which they did
struct KvDst { void* k = nullptr; void* v = nullptr; size_t row_stride = 0; int max_seq = 0, T = 0, col0 = 0; int rows_done = 0; };
void gemm(ma_engine* e, hipStream_t s, const void* A, int lda, const std::string& w, const char* bias_name, const float* R, int ldr, GemmOut out,
          int M, int act, int r_mod = 0, KvDst* kv = nullptr, GemmSplitK* sk = nullptr, GemmLnFuse* lnf = nullptr, int part = 0) {
    const Entry& en = e->L.get(w);
    const float* bias = bias_name ? e->PF(bias_name) : nullptr;
    hipError_t r;
    if ((en.dtype != MA_DTYPE_F32) != e->dense16) throw MaError(MA_ERR_INVALID, "internal: weight " + w + " does not have the precision of the phase that uses it");
    if (e->dense16) {
        GemmTArgs t{};
        t.A = reinterpret_cast<const bf16_t*>(A); t.lda = lda; t.W = reinterpret_cast<const bf16_t*>(e->arena + en.offset); t.bias = bias;
        t.R = R; t.ldr = ldr; t.C = out.c32; t.ldc = out.ld; t.Cb = reinterpret_cast<bf16_t*>(out.act); t.ldcb = out.ld;
        t.M = M; t.N = en.rows; t.K = en.cols; t.act = act; t.r_mod = r_mod; t.cmap = out.map; t.xcd_swizzle = e->opt_gemm_xcd_swizzle; t.part = part;
        if (kv) { t.kv_k = reinterpret_cast<bf16_t*>(kv->k); t.kv_v = reinterpret_cast<bf16_t*>(kv->v); t.kv_row_stride = kv->row_stride; t.kv_max_seq = kv->max_seq; t.kv_T = kv->T; t.kv_col0 = kv->col0; }
        r = H16_CALL(e->hdt, HT, launch_gemm_dense<HT>(t, e->n_cus, s, kv ? &kv->rows_done : nullptr, sk, lnf));
    } else {
        if (part != 0) throw MaError(MA_ERR_INVALID, "internal: a GEMM by row parts needs a 16-bit phase");
        GemmArgs g{};
        g.A = reinterpret_cast<const float*>(A); g.lda = lda; g.W = e->arena + en.offset; g.bias = bias; g.R = R; g.ldr = ldr;
        g.C = out.c32 ? out.c32 : reinterpret_cast<float*>(out.act); g.ldc = out.ld; g.M = M; g.N = en.rows; g.K = en.cols; g.act = act;
        g.r_mod = r_mod; g.cmap = out.map;
        r = launch_gemm<float>(g, e->opt_gemm_impl, s);
    }
    if (r != hipSuccess) throw MaError(MA_ERR_HIP, "gemm launch failed for " + w + ": " + hipGetErrorString(r));
}
void gemm(ma_engine* e, hipStream_t s, const void* A, int lda, const std::string& w, const std::string& b, const float* R, int ldr, GemmOut out, int M,
          int act, int r_mod = 0, KvDst* kv = nullptr, GemmSplitK* sk = nullptr, GemmLnFuse* lnf = nullptr, int part = 0) {
    gemm(e, s, A, lda, w, b.c_str(), R, ldr, out, M, act, r_mod, kv, sk, lnf, part);
}

GemmOut to32(float* c, int ld, RowMap m = RowMap{0, 0, 0}) { GemmOut o; o.c32 = c; o.ld = ld; o.map = m; return o; }
GemmOut toact(void* a, int ld, RowMap m = RowMap{0, 0, 0}) { GemmOut o; o.act = a; o.ld = ld; o.map = m; return o; }

// LayerNorm rows: x fp32 (row map xin) -> y32 (fp32, optional) and ya (activation, optional), both at row map yout
// (parts > 1: the input of the first split_rows rows is the sum of `parts` buffers part_stride floats apart -- a GEMM split along K)
void lnrows(ma_engine* e, hipStream_t s, const float* x, int ldx, const std::string& prefix, float eps, float* y32, int ld32, void* ya, int lda, int rows,
            int D, RowMap xin = RowMap{0, 0, 0}, RowMap yout = RowMap{0, 0, 0}, int parts = 1, long part_stride = 0, int split_rows = 0) {
    const float* g = e->PF(prefix + "weight"); const float* b = e->PF(prefix + "bias");
    if (parts > 1 && !(e->dense16 && D == 1024 && (parts == 2 || parts == 4))) throw MaError(MA_ERR_INVALID, "internal: LayerNorm over a split input needs a 16-bit phase and 1024 columns");
    if (e->dense16) H16_DO(e->hdt, HT, launch_ln_rows2<HT>(x, ldx, xin, g, b, eps, y32, ld32, reinterpret_cast<HT*>(ya), lda, yout, rows, D, s, parts, part_stride, split_rows));
    else launch_ln_rows2<float>(x, ldx, xin, g, b, eps, y32, ld32, reinterpret_cast<float*>(ya), lda, yout, rows, D, s);
    HIP_CHECK(hipGetLastError());
}
// h = LN(h + A W^T + b) for M stacked rows, fp32 in place + 16-bit copy hb (the two post-LN sub-layers of an OPT layer, [3p] OPTDecoderLayer): the GEMM
// finishes the LayerNorm itself where it can (gemm256.hpp LNF form: whole 256-row tiles of a launch that fills the chip); the row kernel does the rest from
// the plain sums in `y`.  split: fc2 of small batches may come as partial sums along K instead (GemmSplitK), which the row kernel adds up.
// part (GemmTArgs::part): 0 = all M rows; 1 = rows [0, M - M % 256); 2 = the rows behind them (A, h, hb, y stay the addresses of row 0)
void gemm_res_ln(ma_engine* e, hipStream_t s, const void* A, int lda, const std::string& w, const std::string& b, const std::string& ln_prefix, float eps, float* h, void* hb,
                 float* y, int M, int H, bool allow_split, int part = 0) {
    const bool fuse = part == 0 && e->opt_fuse_ln && e->dense16 && e->chain_resident && e->d_ln_gran && (size_t)(M / 256) * (size_t)(H / 256) <= e->ln_gran_tiles;
    GemmSplitK sk;
    // (from 8 samples on, like the tail chain: below that a sample's prefill keeps the bits of its batch-1 run -- the GEMMs run on row-independent tiles only)
    sk.max_parts = (allow_split && e->opt_gemm_splitk && e->dense16 && H == 1024 && M >= 2048 && (long)M * H <= e->p_y_part_stride) ? 4 : 1;
    sk.part_stride = e->p_y_part_stride;
    if (fuse) {
        GemmLnFuse lf;
        lf.ln.gamma = e->PF(ln_prefix + "weight"); lf.ln.beta = e->PF(ln_prefix + "bias"); lf.ln.eps = eps; lf.ln.gran = e->d_ln_gran; lf.ln.err = e->d_chain_err;
        if (++e->ln_epoch == 0) e->ln_epoch = 1;
        lf.ln.epoch = e->ln_epoch;
        lf.tail_c = y;
        GemmOut o; o.c32 = h; o.act = hb; o.ld = H;
        gemm(e, s, A, lda, w, b, h, H, o, M, ACT_NONE, 0, nullptr, &sk, &lf);
        if (lf.rows < M) {
            const size_t r0 = (size_t)lf.rows;
            // (a split GEMM never takes the LNF form: then lf.rows == 0 and the parts cover sk.rows rows from row 0)
            lnrows(e, s, y + r0 * H, H, ln_prefix, eps, h + r0 * H, H, reinterpret_cast<char*>(hb) + r0 * H * e->act_elem, H, M - lf.rows, H, RowMap{0, 0, 0}, RowMap{0, 0, 0},
                   sk.parts, sk.part_stride, sk.rows);
        }
        return;
    }
    gemm(e, s, A, lda, w, b, h, H, to32(y, H), M, ACT_NONE, 0, nullptr, &sk, nullptr, part);
    const int Mm = M - M % 256;
    if (part == 2) {        // (the rows behind the split ones are complete in the first buffer)
        const size_t r0 = (size_t)Mm;
        if (M > Mm) lnrows(e, s, y + r0 * H, H, ln_prefix, eps, h + r0 * H, H, reinterpret_cast<char*>(hb) + r0 * H * e->act_elem, H, M - Mm, H);
        return;
    }
    const int rows = part == 1 ? Mm : M;
    if (rows > 0) lnrows(e, s, y, H, ln_prefix, eps, h, H, hb, H, rows, H, RowMap{0, 0, 0}, RowMap{0, 0, 0}, sk.parts, sk.part_stride, std::min(sk.rows, rows));
}
// attention over activation tensors; strides in elements; batch = samples (grid.z)
void attention(ma_engine* e, hipStream_t s, const void* Q, int q_rs, int q_hs, const void* K, int k_rs, int k_hs, const void* Vp, int v_rs, int v_hs, void* O,
               int o_rs, int Sq, int Sk, int H, int causal_offset, int batch = 1, size_t q_bs = 0, size_t k_bs = 0, size_t v_bs = 0, size_t o_bs = 0, bf16_t* vt = nullptr,
               size_t vt_elems = 0) {
    AttnArgs a{Q, q_rs, q_hs, K, k_rs, k_hs, Vp, v_rs, v_hs, O, o_rs, Sq, Sk, H, 0.125f, causal_offset, e->dense16 ? 3 : 0};
    a.batch = batch; a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs;
    if (e->dense16 && (e->opt_attn_impl == 2 || e->hdt == MA_DTYPE_F16)) {     // (the first-generation kernel, attn_impl 1, is bf16 only)
        if (attn2_vt_elems(Sk, H, batch) > (vt ? vt_elems : e->vt_elems)) throw MaError(MA_ERR_INVALID, "internal: V^T workspace too small");
        HIP_CHECK(H16_CALL(e->hdt, HT, launch_attention2<HT>(a, vt ? vt : e->a_vt, s)));
    } else HIP_CHECK(launch_attention(a, s));
}
// fp32 stream rows (row map in, optional row mask) -> activation tensor
void cvt_rows(ma_engine* e, hipStream_t s, const float* src, int lds, RowMap in, const unsigned char* mask, void* dst, int ldd, int rows, int cols) {
    if (e->dense16) H16_DO(e->hdt, HT, hipLaunchKernelGGL((cvt_rows_kernel<HT>), dim3(ceil_div(rows * cols, 256)), dim3(256), 0, s, src, lds, in, mask, reinterpret_cast<HT*>(dst), ldd, rows, cols));
    else hipLaunchKernelGGL((cvt_rows_kernel<float>), dim3(ceil_div(rows * cols, 256)), dim3(256), 0, s, src, lds, in, mask, reinterpret_cast<float*>(dst), ldd, rows, cols);
    HIP_CHECK(hipGetLastError());
}
void add_rows(ma_engine* e, hipStream_t s, const float* in, int ld_in, const unsigned char* mask, const float* t0, const float* tab, int ld_tab, int row0,
              float* out32, int ld_out, void* outa, int ld_outa, int rows, int cols, int tab_mod = 0) {
    if (e->dense16) H16_DO(e->hdt, HT, hipLaunchKernelGGL((add_rows2_kernel<HT>), dim3(ceil_div(rows * cols, 256)), dim3(256), 0, s, in, ld_in, mask, t0, tab, ld_tab, row0, out32, ld_out,
                                    reinterpret_cast<HT*>(outa), ld_outa, rows, cols, tab_mod));
    else hipLaunchKernelGGL((add_rows2_kernel<float>), dim3(ceil_div(rows * cols, 256)), dim3(256), 0, s, in, ld_in, mask, t0, tab, ld_tab, row0, out32, ld_out,
                            reinterpret_cast<float*>(outa), ld_outa, rows, cols, tab_mod);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ point encoder
// ResidualAttentionBlock (transformer_blocks.py:109-112): x += proj(attn(c_qkv(ln_1 x))); x += mlp(ln_2 x), for nb samples of S
// rows each stacked in x (nb * S, W) fp32, in place.
void miche_block(ma_engine* e, hipStream_t s, float* x, int S, int nb, const std::string& p) {
    const int W = e->cfg.enc_width, Hh = e->cfg.enc_heads, rows = nb * S;
    lnrows(e, s, x, W, p + "ln_1.", 1e-5f, nullptr, 0, e->a_ln, W, rows, W);
    gemm(e, s, e->a_ln, W, p + "attn.c_qkv.weight", nullptr, nullptr, 0, toact(e->a_qkv, 3 * W), rows, ACT_NONE);
    // per-head interleaved [q|k|v] (transformer_blocks.py:61-62): head stride 192, k at +64, v at +128
    attention(e, s, e->a_qkv, 3 * W, 192, aoff(e, e->a_qkv, 64), 3 * W, 192, aoff(e, e->a_qkv, 128), 3 * W, 192, e->a_att, W, S, S, Hh, -1, nb, (size_t)S * 3 * W,
              (size_t)S * 3 * W, (size_t)S * 3 * W, (size_t)S * W);
    gemm(e, s, e->a_att, W, p + "attn.c_proj.weight", p + "attn.c_proj.bias", x, W, to32(x, W), rows, ACT_NONE);
    lnrows(e, s, x, W, p + "ln_2.", 1e-5f, nullptr, 0, e->a_ln, W, rows, W);
    gemm(e, s, e->a_ln, W, p + "mlp.c_fc.weight", p + "mlp.c_fc.bias", nullptr, 0, toact(e->a_mlp, 4 * W), rows, ACT_GELU);
    gemm(e, s, e->a_mlp, 4 * W, p + "mlp.c_proj.weight", p + "mlp.c_proj.bias", x, W, to32(x, W), rows, ACT_NONE);
}

// encode_latents (asl_pl_module.py:145-157 -> sal_perceiver.py:372-381 -> 74-99) for nb samples -> latents (nb, T, W) fp32
void encode_chunk(ma_engine* e, hipStream_t s, const void* pc, int pc_dtype, int nb, float* latents) {
    const ma_config& c = e->cfg;
    const int N = c.n_points, W = c.enc_width, T = e->T, Hh = c.enc_heads, rowsN = nb * N, rowsT = nb * T;
    {
        const int total = rowsN * 64;
        if (pc_dtype == MA_DTYPE_F16) {
            if (e->dense16) H16_DO(e->hdt, HT, hipLaunchKernelGGL((fourier2_kernel<_Float16, HT>), dim3(ceil_div(total, 256)), dim3(256), 0, s, reinterpret_cast<const _Float16*>(pc), rowsN, c.num_freqs, reinterpret_cast<HT*>(e->a_feat), 64));
            else hipLaunchKernelGGL((fourier2_kernel<_Float16, float>), dim3(ceil_div(total, 256)), dim3(256), 0, s, reinterpret_cast<const _Float16*>(pc), rowsN, c.num_freqs, reinterpret_cast<float*>(e->a_feat), 64);
        } else {
            if (e->dense16) H16_DO(e->hdt, HT, hipLaunchKernelGGL((fourier2_kernel<float, HT>), dim3(ceil_div(total, 256)), dim3(256), 0, s, reinterpret_cast<const float*>(pc), rowsN, c.num_freqs, reinterpret_cast<HT*>(e->a_feat), 64));
            else hipLaunchKernelGGL((fourier2_kernel<float, float>), dim3(ceil_div(total, 256)), dim3(256), 0, s, reinterpret_cast<const float*>(pc), rowsN, c.num_freqs, reinterpret_cast<float*>(e->a_feat), 64);
        }
        HIP_CHECK(hipGetLastError());
    }
    gemm(e, s, e->a_feat, 64, SM + "encoder.input_proj.weight", SM + "encoder.input_proj.bias", nullptr, 0, to32(e->w_data, W), rowsN, ACT_NONE);
    const std::string p = SM + "encoder.cross_attn.";
    const float* query = e->PF(SM + "encoder.query");
    // x = query + attn(ln_1 query, ln_2 data); x += mlp(ln_3 x)     (transformer_blocks.py:223-226).  The query side is the same
    // for every sample: computed once, attended by every sample's keys (q batch stride 0)
    lnrows(e, s, query, W, p + "ln_1.", 1e-5f, nullptr, 0, e->a_ln, W, T, W);
    gemm(e, s, e->a_ln, W, p + "attn.c_q.weight", nullptr, nullptr, 0, toact(e->a_q, W), T, ACT_NONE);
    lnrows(e, s, e->w_data, W, p + "ln_2.", 1e-5f, nullptr, 0, e->a_dataln, W, rowsN, W);
    gemm(e, s, e->a_dataln, W, p + "attn.c_kv.weight", nullptr, nullptr, 0, toact(e->a_kv, 2 * W), rowsN, ACT_NONE);
    // kv viewed (N, heads, 128) split [k|v] (transformer_blocks.py:172-174)
    attention(e, s, e->a_q, W, 64, e->a_kv, 2 * W, 128, aoff(e, e->a_kv, 64), 2 * W, 128, e->a_att, W, T, N, Hh, -1, nb, 0, (size_t)N * 2 * W, (size_t)N * 2 * W,
              (size_t)T * W);
    gemm(e, s, e->a_att, W, p + "attn.c_proj.weight", p + "attn.c_proj.bias", query, W, to32(e->w_lat, W), rowsT, ACT_NONE, /*r_mod=*/T);
    lnrows(e, s, e->w_lat, W, p + "ln_3.", 1e-5f, nullptr, 0, e->a_ln, W, rowsT, W);
    gemm(e, s, e->a_ln, W, p + "mlp.c_fc.weight", p + "mlp.c_fc.bias", nullptr, 0, toact(e->a_mlp, 4 * W), rowsT, ACT_GELU);
    gemm(e, s, e->a_mlp, 4 * W, p + "mlp.c_proj.weight", p + "mlp.c_proj.bias", e->w_lat, W, to32(e->w_lat, W), rowsT, ACT_NONE);
    for (int n = 0; n < c.enc_layers; ++n) miche_block(e, s, e->w_lat, T, nb, SM + "encoder.self_attn.resblocks." + std::to_string(n) + ".");
    lnrows(e, s, e->w_lat, W, SM + "encoder.ln_post.", 1e-5f, latents, W, nullptr, 0, rowsT, W);
}

// to_shape_latents (asl_pl_module.py:182-185 -> sal_perceiver.py:383-396 pre_kl / mode() / post_kl, 273-275 transformer) for nb
// samples: lat rows `in` of a (.., ld) fp32 tensor -> e->w_lat2 (nb * NL, W) fp32
void shape_latents_chunk(ma_engine* e, hipStream_t s, const float* lat, int ld, RowMap in, int nb) {
    const ma_config& c = e->cfg;
    const int W = c.enc_width, E = c.embed_dim, NL = c.num_latents, rows = nb * NL;
    cvt_rows(e, s, lat, ld, in, nullptr, e->a_ln, W, rows, W);
    gemm(e, s, e->a_ln, W, SM + "pre_kl.weight", SM + "pre_kl.bias", nullptr, 0, toact(e->a_mean, E), rows, ACT_NONE);      // posterior.mode(): the mean half
    gemm(e, s, e->a_mean, E, SM + "post_kl.weight", SM + "post_kl.bias", nullptr, 0, to32(e->w_lat2, W), rows, ACT_NONE);
    for (int n = 0; n < c.shape_layers; ++n) miche_block(e, s, e->w_lat2, NL, nb, SM + "transformer.resblocks." + std::to_string(n) + ".");
}

// process_point_feature (meshanything.py:125-132) incl. to_shape_latents for nb samples: latents (nb, T, W) -> prefix (nb, T, H)
void prefix_chunk(ma_engine* e, hipStream_t s, const float* latents, float* prefix, int nb) {
    const ma_config& c = e->cfg;
    const int W = c.enc_width, H = c.hidden, NL = c.num_latents, T = e->T, rows = nb * NL;
    const RowMap tail{NL, T, 1}, head{1, T, 0};                            // point_feature[:, 1:] and [:, 0] inside the T-row blocks
    shape_latents_chunk(e, s, latents, W, tail, nb);
    cvt_rows(e, s, latents, W, tail, nullptr, e->a_cat, 2 * W, rows, W);                            // cat([latents, shape_latents], -1)
    cvt_rows(e, s, e->w_lat2, W, RowMap{0, 0, 0}, nullptr, aoff(e, e->a_cat, W), 2 * W, rows, W);
    cvt_rows(e, s, latents, W, head, nullptr, e->a_ln, W, nb, W);
    gemm(e, s, e->a_ln, W, "cond_head_proj.weight", "cond_head_proj.bias", nullptr, 0, to32(prefix, H, head), nb, ACT_NONE);
    gemm(e, s, e->a_cat, 2 * W, "cond_proj.weight", "cond_proj.bias", nullptr, 0, to32(prefix, H, tail), rows, ACT_NONE);
}

// ------------------------------------------------------------------------------------------------ decoder
template <typename WT>
void gemv_launch(const GemvArgs& a, hipStream_t s) {
    hipError_t r = launch_gemv<WT>(a, s);
    if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("gemv launch failed: ") + hipGetErrorString(r));
}
int gemv_blocks(ma_engine* e, int N, int K) { return e->bf16 ? gemv_num_blocks<bf16_t>(N, K) : gemv_num_blocks<float>(N, K); }     // (the two 16-bit formats share their shapes)

struct StepTimer {                    // launch filter (ma_profile_decode) / in-kernel timestamps (ma_trace_decode)
    int only_cls = -1;                // >= 0: enqueue only the launches of this class (0 gemv, 1 attention, 3 pick)
    int launched[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool on(int c) { if (only_cls >= 0 && c != only_cls) return false; launched[c]++; return true; }
    unsigned long long* tr = nullptr; int tr_max_launches = 0, tr_max_blocks = 0;
    std::vector<int>* tr_kind = nullptr; std::vector<int>* tr_blocks = nullptr;
    // slot for the next launch's timestamps (kind: 0 embed, 1 qkv, 2 attention, 3 out_proj, 4 fc1, 5 fc2, 6 lm_head)
    unsigned long long* trace_slot(int kind, int blocks) {
        if (!tr || (int)tr_kind->size() >= tr_max_launches || blocks > tr_max_blocks) return nullptr;
        unsigned long long* p = tr + (size_t)tr_kind->size() * tr_max_blocks * 4;
        tr_kind->push_back(kind); tr_blocks->push_back(blocks);
        return p;
    }
};

// ---- batched decode: rows r0 .. r0+B-1 of the engine's per-row buffers.  Row b owns the b-th slice of every activation
// buffer, its own KV planes, its own DecState record and its own output token row; the weights are shared.
struct Rows { int r0 = 0, B = 1; };

GemvArgs gemv_base(ma_engine* e, Rows rw) {
    GemvArgs a{};
    a.round_x = e->bf16 ? 1 : 0;
    a.st = e->d_st + rw.r0;
    a.act = ACT_NONE;
    a.epi = EPI_PLAIN;
    return a;
}
void gemv(ma_engine* e, const GemvArgs& a, hipStream_t s, int B) {
    hipError_t r = e->bf16 ? H16_CALL(e->hdt, HT, launch_gemv<HT>(a, s, B)) : launch_gemv<float>(a, s, B);
    if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("gemv launch failed: ") + hipGetErrorString(r));
}

bool use_mfma_decode(ma_engine* e, int B) { return e->bf16 && B >= e->opt_mfma_min_batch && B <= 64 && e->cfg.hidden % 128 == 0 && e->cfg.ffn % 128 == 0 && e->cfg.hidden <= 1024; }

// input of a batched prologue: either a plain fp32 buffer, or the raw partials of a split-K GEMM plus its deferred epilogue
struct ProIn { const float* x = nullptr; int nparts = 1; const float* bias = nullptr; const float* res = nullptr; };

void rows_prologue(ma_engine* e, hipStream_t s, int pro, Rows rw, ProIn in, const float* g, const float* b, float* xn_out, StepTimer& tm) {
    if (!tm.on(0)) return;
    const ma_config& c = e->cfg;
    RowsProArgs a{};
    a.x = in.x; a.x_stride = c.hidden; a.nparts = in.nparts; a.B = rw.B; a.bias = in.bias; a.res = in.res; a.res_stride = c.hidden;
    a.ln_g = g; a.ln_b = b; a.ln_eps = 1e-5f;
    a.attn_ws = e->d_part + (size_t)rw.r0 * attn_workspace_floats(c.heads); a.attn_ws_stride = attn_workspace_floats(c.heads); a.attn_heads = c.heads;
    a.xn_out = xn_out; a.xn_stride = c.hidden; a.xb = e->d_xb + (size_t)rw.r0 * c.hidden; a.xb_stride = c.hidden; a.K = c.hidden;
    hipError_t r = H16_CALL(e->hdt, HT, launch_rows_prologue<HT>(a, pro, rw.B, s));
    if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("rows_prologue launch failed: ") + hipGetErrorString(r));
}
// kind: the timeline's launch kind (StepTimer::trace_slot)
void gemm_dec_ln(ma_engine* e, hipStream_t s, GemmDecArgs a, StepTimer& tm, int kind) {
    if (!tm.on(0)) return;
    a.trace = tm.trace_slot(kind, (a.N + 15) / 16);
    hipError_t r = H16_CALL(e->hdt, HT, launch_gemm_dec_ln<HT>(a, s, e->opt_mfma_ln_waves));
    if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("gemm_dec_ln launch failed: ") + hipGetErrorString(r));
}
void gemm_dec(ma_engine* e, hipStream_t s, GemmDecArgs a, StepTimer& tm, int kind) {
    if (!tm.on(0)) return;
    a.trace = tm.trace_slot(kind, (a.N + 15) / 16 * std::max(1, a.ksplit));
    hipError_t r = H16_CALL(e->hdt, HT, launch_gemm_dec<HT>(a, s));
    if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("gemm_dec launch failed: ") + hipGetErrorString(r));
}

// The gates of the two 8-row launches (rows_attn.hpp, rows_mlp.hpp) that do not depend on the layer -- ONE definition for the step builder
// below and for ma_engine_get_option("fuse_rows_attn" / "fuse_rows_mlp"), which bench.py uses to label the roofline kernel.
struct RowsGates {
    bool fold1, fold;                 // LayerNorm 1 inside fc1 / LayerNorm 2 inside q/k/v (gemm_dec_ln_kernel)
    int ks_f;                         // split of fc2 along K
    bool pair_ok;                     // the two-block final-form attention: both blocks of every (row, head) resident together
    bool attn, mlp;                   // the fused first / second half of a layer, as far as the layer index does not matter
};
RowsGates rows_gates(ma_engine* e, int B, int len_override) {
    const ma_config& c = e->cfg;
    const int H = c.hidden;
    RowsGates g{};
    g.fold1 = e->opt_mfma_fold_ln && B <= e->opt_mfma_fold_fc1_max && B <= 16 && H == 1024;
    g.fold = e->opt_mfma_fold_ln && B <= e->opt_mfma_fold_qkv_max && B <= 16 && H == 1024;
    g.ks_f = e->opt_mfma_fc2_ksplit ? e->opt_mfma_fc2_ksplit : gemm_dec_ksplit(H, c.ffn);
    // (the hand-over epoch is position * 32 + layer + 1: more than 31 layers would alias the next position's layer 0)
    g.pair_ok = e->opt_attn_pair && e->chain_resident && c.layers <= 31 && 2 * B * c.heads <= e->n_cus && (e->opt_attn_final_waves == 0 || e->opt_attn_final_waves == 8);
    // exchange epochs come from DecState.pos (no caller-supplied length); 256 blocks of 8 waves need every CU (pair_ok's gate); row groups
    // stepping on their own streams would put two such launches on the device at once: not with these
    const bool gate = e->rows_ok && e->opt_decode_groups <= 1 && g.pair_ok && B == RA_ROWS && len_override < 0 && H == 1024;
    g.attn = e->opt_fuse_rows_attn && gate && 2 * B * c.heads == 256 && B >= e->opt_attn_final_min_batch && c.heads == 16;
    g.mlp = e->opt_fuse_rows_mlp && gate && c.ffn == 4096 && g.fold1 && g.ks_f == 4 && (size_t)c.ffn >= (size_t)RM_FFN_GRANULES;
    return g;
}

// The 24 OPT layers + lm_head of one decode step for a batch on the matrix cores (gemm_decode.hpp).  Same data flow as the
// GEMV path; the prologues are one-block-per-row launches, and the two N = hidden GEMMs (out_proj, fc2) are split along K
// with their bias / residual folded into the LayerNorm prologue that follows them.
//   layer input:  l == 0: the embedding (its 16-bit copy xb comes from the embedding launch itself);  l > 0: LN2_{l-1}(h1 + fc2 partials + b2)
//                 -> h0 (fp32 residual), xb (16-bit) -- inside the q/k/v GEMM up to 8 rows (gemm_dec_ln_kernel), else by a rows_prologue launch
void enqueue_layers_mfma(ma_engine* e, hipStream_t s, const float* x_embed, int len_override, StepTimer& tm, Rows rw) {
    const ma_config& c = e->cfg;
    const int H = c.hidden, B = rw.B, L = c.layers;
    const size_t r0 = rw.r0, MB = c.max_batch;
    float* h0 = e->d_h0 + r0 * H; float* q = e->d_q + r0 * H; float* h1 = e->d_h1 + r0 * H; float* y1 = e->d_ypre1 + r0 * H; float* y2 = e->d_ypre2 + r0 * H;
    float* part = e->d_part + r0 * attn_workspace_floats(c.heads);
    bf16_t* xb = e->d_xb + r0 * H; bf16_t* ffb = e->d_ffb + r0 * c.ffn;
    // split-K partial buffers [ks][B][H] of THIS row range: the ranges of concurrently stepping row groups do not overlap
    // (4 r0 H floats in front of it belong to the rows before r0, whatever their grouping)
    float* partO = e->d_ks_o + 4 * r0 * H; float* partF = e->d_ks_f + 4 * r0 * H;
    (void)MB;
    const size_t kv_row_elems = e->kv_row_bytes / e->kv_elem;
    const RowsGates rg = rows_gates(e, B, len_override);
    const int ks_o = gemm_dec_ksplit(H, H), ks_f = rg.ks_f;
    // 4..16 rows: the LayerNorm prologues run inside the consuming GEMMs (gemm_dec_ln_kernel) and out_proj is not split along K, so
    // that its epilogue finishes y1: two launches fewer per layer
    // (the folded prologue reads its inputs once per block: 1 buffer in front of fc1, 4 split-K partials + residual in front of q/k/v, so
    //  the second stops paying earlier: profiles/r02_ab_batched_ln_fold.txt)
    const bool fold1 = rg.fold1;                                // LN1 inside fc1
    const bool fold = rg.fold;                                  // LN2 inside q/k/v
    const int ks_o_eff = fold1 ? 1 : ks_o;
    bool ln2_prev = false;                                      // the previous layer's MLP launch finished its LayerNorm 2 (rows_mlp.hpp step E): xb and h0 are ready
    for (int l = 0; l < L; ++l) {
        const DecLayerPtrs& w = e->dl[l];
        const float* resid;
        ProIn qin;                                              // input of this layer's q/k/v GEMM when its LayerNorm is folded
        if (l == 0) {
            resid = x_embed;                                    // (its 16-bit copy xb was written by the embedding launch)
        } else if (ln2_prev) {
            resid = h0;
        } else {
            ProIn in;
            if (ks_f > 1) { in.x = partF; in.nparts = ks_f; in.bias = e->dl[l - 1].fc2_b; in.res = h1; } else in.x = y2;
            if (fold) qin = in;
            else rows_prologue(e, s, PRO_LN, rw, in, e->dl[l - 1].ln2_g, e->dl[l - 1].ln2_b, h0, tm);
            resid = h0;
        }
        // the two-block final-form attention needs both blocks of every (row, head) resident together: 2 B heads <= CUs (8 rows on an MI355X)
        const bool pair_ok = rg.pair_ok;
        // 8 rows: LayerNorm 2 + q/k/v + attention + out_proj in ONE launch (rows_attn.hpp) -- three launches per layer instead of five (gates: rows_gates)
        const bool fused_attn = rg.attn && (fold || ln2_prev);
        if (fused_attn) {
            if (tm.on(1)) {
                RowsAttnArgs a{};
                a.Wqkv = reinterpret_cast<const bf16_t*>(w.qkv_w); a.bqkv = w.qkv_b;
                if (l == 0 || ln2_prev) { a.xb = xb; a.xb_stride = H; a.res = resid; a.res_stride = H; }
                else {
                    a.pin = qin.x; a.pin_stride = H; a.pin_parts = qin.nparts; a.pbias = qin.bias; a.pres = qin.res; a.pres_stride = H;
                    a.ln_g = e->dl[l - 1].ln2_g; a.ln_b = e->dl[l - 1].ln2_b; a.ln_eps = 1e-5f; a.xn_out = h0; a.xn_stride = H;
                }
                a.kcache = reinterpret_cast<bf16_t*>(e->kplane(rw.r0, l)); a.vcache = reinterpret_cast<bf16_t*>(e->vplane(rw.r0, l)); a.kv_row_stride = kv_row_elems; a.max_seq = e->maxseq;
                a.st = e->d_st + r0; a.len_override = len_override; a.layer = l;
                a.qkv_gran = e->d_ra_qkv_gran + r0 * RA_QKV_GRANULES; a.pair_gran = e->d_attn_pair_gran + r0 * c.heads * ATTN_PAIR_GRANULES; a.out_gran = e->d_ra_out_gran + r0 * RA_OUT_GRANULES;
                a.err = e->d_chain_err; a.Wo = reinterpret_cast<const bf16_t*>(w.o_w); a.bo = w.o_b; a.y1 = y1; a.y1_stride = H;
                a.trace = tm.trace_slot(2, 256);
                hipError_t r = H16_CALL(e->hdt, HT, launch_rows_attn<HT>(a, c.heads, B, s, e->opt_rows_attn_early, l == 0 ? 4 : 8));
                if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("rows_attn launch failed: ") + hipGetErrorString(r));
            }
        } else {
        {
            GemmDecArgs a{};
            a.W = reinterpret_cast<const bf16_t*>(w.qkv_w); a.bias = w.qkv_b; a.xb = xb; a.xb_stride = H; a.y = q; a.y_stride = H; a.N = 3 * H; a.K = H; a.B = B; a.ksplit = 1;
            a.epi = EPI_QKV; a.kcache = e->kplane(rw.r0, l); a.vcache = e->vplane(rw.r0, l); a.kv_row_stride = kv_row_elems; a.H = H; a.max_seq = e->maxseq; a.st = e->d_st + r0;
            if (fold && l > 0 && !ln2_prev) {
                a.pin = qin.x; a.pin_stride = H; a.pin_parts = qin.nparts; a.pbias = qin.bias; a.pres = qin.res; a.pres_stride = H;
                a.ln_g = e->dl[l - 1].ln2_g; a.ln_b = e->dl[l - 1].ln2_b; a.ln_eps = 1e-5f; a.xn_out = h0; a.xn_stride = H;
                gemm_dec_ln(e, s, a, tm, 1);
            } else gemm_dec(e, s, a, tm, 1);
        }
        // 8..11 rows give only 128-176 (row, head) blocks: enough up to ~8 K cached positions, beyond that (1600-face configuration) the
        // split form streams better (profiles/r02_ab_batched_attention_forms.txt, r02_bench_config5_*)
        if (B >= e->opt_attn_final_min_batch && (B >= 12 || pair_ok || e->maxseq <= 8192)) {
            // enough (row, head) pairs to fill the chip: the attention launch finishes the softmax itself and writes xb
            if (tm.on(1)) {
                // 8..11 rows: two blocks per (row, head) with an in-launch hand-over, so that every CU streams (attn_decode.hpp)
                const bool pair = pair_ok && B < 12;
                hipError_t r = H16_CALL(e->hdt, HT, launch_attn_decode_final<HT>(q, e->kplane(rw.r0, l), e->vplane(rw.r0, l), c.heads, e->maxseq, e->d_st + r0, len_override, 1, xb, H, s, B, H, kv_row_elems,
                                                                pair ? 8 : e->opt_attn_final_waves, pair ? e->d_attn_pair_gran + r0 * c.heads * ATTN_PAIR_GRANULES : nullptr, e->d_chain_err, l));
                if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("attn_decode_final launch failed: ") + hipGetErrorString(r));
            }
        } else {
            if (tm.on(1)) {
                hipError_t r = H16_CALL(e->hdt, HT, launch_attn_decode<HT>(q, e->kplane(rw.r0, l), e->vplane(rw.r0, l), c.heads, e->maxseq, e->d_st + r0, len_override, 1, part, s, nullptr, B, H, kv_row_elems, e->opt_attn_rowwave != 0));
                if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("attn_decode launch failed: ") + hipGetErrorString(r));
            }
            rows_prologue(e, s, PRO_ATTN, rw, ProIn{}, nullptr, nullptr, nullptr, tm);
        }
        {   // y1 = resid + Wo a + bo
            GemmDecArgs a{};
            a.W = reinterpret_cast<const bf16_t*>(w.o_w); a.xb = xb; a.xb_stride = H; a.N = H; a.K = H; a.B = B; a.ksplit = ks_o_eff; a.y_stride = H;
            if (ks_o_eff > 1) a.y = partO; else { a.y = y1; a.bias = w.o_b; a.res = resid; a.res_stride = H; }
            gemm_dec(e, s, a, tm, 3);
        }
        }
        ProIn in1;
        if (ks_o_eff > 1 && !fused_attn) { in1.x = partO; in1.nparts = ks_o_eff; in1.bias = w.o_b; in1.res = resid; } else in1.x = y1;
        // 8 rows: LayerNorm 1 + fc1 + fc2 in ONE launch (rows_mlp.hpp): the same gates as the fused first half, and y1 complete in one buffer
        const bool fused_mlp = rg.mlp && in1.nparts == 1;
        if (fused_mlp) {
            if (tm.on(0)) {
                RowsMlpArgs a{};
                a.y1 = in1.x; a.y1_stride = H; a.ln_g = w.ln1_g; a.ln_b = w.ln1_b; a.ln_eps = 1e-5f; a.h1_out = h1; a.h1_stride = H;
                a.W1 = reinterpret_cast<const bf16_t*>(w.fc1_w); a.b1 = w.fc1_b; a.W2 = reinterpret_cast<const bf16_t*>(w.fc2_w); a.part = partF; a.part_stride = H;
                a.st = e->d_st + r0; a.layer = l; a.ffn_gran = e->d_ffn_gran + r0 * c.ffn; a.err = e->d_chain_err;
                // LayerNorm 2 in this launch's tail: the next layer starts from 16-bit rows.  Not for the last layer: the rows that feed lm_head are
                // normalised by rows_prologue_kernel (block-level sums: another order than the one-wave LayerNorm of the folded GEMMs)
                if (e->opt_rows_mlp_ln2 && l + 1 < L) {
                    a.b2 = w.fc2_b; a.ln2_g = w.ln2_g; a.ln2_b = w.ln2_b; a.y2_gran = e->d_rm_y2_gran + r0 * RM_Y2_GRANULES;
                    a.x2_out = h0; a.x2_stride = H; a.xb_out = xb; a.xb_stride = H; a.part = nullptr;
                    if (e->opt_rows_mlp_prefetch > 0 && e->opt_fuse_rows_attn) {
                        a.pf_k = reinterpret_cast<const bf16_t*>(e->kplane(rw.r0, l + 1)); a.pf_v = reinterpret_cast<const bf16_t*>(e->vplane(rw.r0, l + 1));
                        a.pf_row_stride = kv_row_elems; a.pf_max_seq = e->maxseq; a.pf_rounds = e->opt_rows_mlp_prefetch;
                        a.pf_wqkv = reinterpret_cast<const bf16_t*>(e->dl[l + 1].qkv_w); a.pf_sink = e->d_pf_sink;
                    }
                }
                a.trace = tm.trace_slot(4, 256);
                hipError_t r = H16_CALL(e->hdt, HT, launch_rows_mlp<HT>(a, B, H, c.ffn, s));
                if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("rows_mlp launch failed: ") + hipGetErrorString(r));
            }
            ln2_prev = e->opt_rows_mlp_ln2 != 0 && l + 1 < L;
            continue;
        }
        ln2_prev = false;
        if (!fold1) rows_prologue(e, s, PRO_LN, rw, in1, w.ln1_g, w.ln1_b, h1, tm);
        {
            GemmDecArgs a{};
            a.W = reinterpret_cast<const bf16_t*>(w.fc1_w); a.bias = w.fc1_b; a.xb = xb; a.xb_stride = H; a.yb = ffb; a.yb_stride = c.ffn; a.N = c.ffn; a.K = H; a.B = B; a.ksplit = 1;
            a.act = ACT_RELU;
            if (fold1) {
                a.pin = in1.x; a.pin_stride = H; a.pin_parts = in1.nparts; a.pbias = in1.bias; a.pres = in1.res; a.pres_stride = H;
                a.ln_g = w.ln1_g; a.ln_b = w.ln1_b; a.ln_eps = 1e-5f; a.xn_out = h1; a.xn_stride = H;
                gemm_dec_ln(e, s, a, tm, 4);
            } else gemm_dec(e, s, a, tm, 4);
        }
        {   // y2 = h1 + W2 f + b2
            GemmDecArgs a{};
            a.W = reinterpret_cast<const bf16_t*>(w.fc2_w); a.xb = ffb; a.xb_stride = c.ffn; a.N = H; a.K = c.ffn; a.B = B; a.ksplit = ks_f; a.y_stride = H;
            if (ks_f > 1) a.y = partF; else { a.y = y2; a.bias = w.fc2_b; a.res = h1; a.res_stride = H; }
            gemm_dec(e, s, a, tm, 5);
        }
    }
    // lm_head on LN2_{L-1}(y2)
    ProIn in;
    if (ks_f > 1) { in.x = partF; in.nparts = ks_f; in.bias = e->dl[L - 1].fc2_b; in.res = h1; } else in.x = y2;
    if (!ln2_prev) rows_prologue(e, s, PRO_LN, rw, in, e->dl[L - 1].ln2_g, e->dl[L - 1].ln2_b, nullptr, tm);      // (else: the last MLP launch left xb)
    GemmDecArgs g{};
    g.W = reinterpret_cast<const bf16_t*>(e->P("transformer.lm_head.weight")); g.xb = xb; g.xb_stride = H;
    g.y = e->d_logits + r0 * e->V; g.y_stride = e->V; g.N = e->V; g.K = H; g.B = B; g.ksplit = 1;
    gemm_dec(e, s, g, tm, 6);
}

// The fused launches spin on granules written by other blocks of the same grid, so EVERY block of the grid (256 per batch row) must
// be resident at once: B rows are fused only while 256 B blocks fit with a quarter of margin (HIP does not promise in-order
// dispatch, so a later row's blocks may not be assumed to wait politely); larger B, a device that turned out to be shared
// (chain_resident cleared after a timeout), or a caller-supplied length (the exchange epochs come from DecState.pos, which
// such a caller does not advance) take the five-launch chain.
bool chain_fits(ma_engine* e, int B, int len_override) { return e->chain_resident && len_override < 0 && e->resident_blocks * 4 >= 256L * B * 5; }
bool fuse_oproj_fc1(ma_engine* e, int B = 1, int len_override = -1) {
    return e->opt_fuse_oproj_fc1 && chain_fits(e, B, len_override) && e->cfg.hidden == 1024 && e->cfg.ffn == 4096 && e->cfg.heads * 64 == e->cfg.hidden && e->cfg.layers <= 30;
}
bool fuse_qkv_attn(ma_engine* e, int B = 1, int len_override = -1) {
    return e->opt_fuse_qkv_attn && chain_fits(e, B, len_override) && e->cfg.hidden == 1024 && e->cfg.heads * 64 == e->cfg.hidden && e->cfg.layers <= 30;
}

// one OPT layer of one decode step.  `x_in` = this layer's input (row stride H) before its (optional) LayerNorm prologue.
QkvAttnArgs make_qkv_attn_args(ma_engine* e, int l, const float* x_in, const float* ln_g, const float* ln_b, int len_override, Rows rw) {
    const ma_config& c = e->cfg;
    const int H = c.hidden; const size_t r0 = rw.r0;
    const DecLayerPtrs& w = e->dl[l];
    QkvAttnArgs a{};
    a.W = reinterpret_cast<const bf16_t*>(w.qkv_w); a.bias = w.qkv_b; a.x = x_in; a.ln_g = ln_g; a.ln_b = ln_b; a.ln_eps = 1e-5f; a.xn_out = ln_g ? e->d_h0 + r0 * H : nullptr;
    a.kcache = reinterpret_cast<bf16_t*>(e->kplane(rw.r0, l)); a.vcache = reinterpret_cast<bf16_t*>(e->vplane(rw.r0, l)); a.max_seq = e->maxseq; a.hidden = H;
    a.st = e->d_st + r0; a.len_override = len_override; a.layer = l; a.ws = e->d_part + r0 * attn_workspace_floats(c.heads); a.gran = e->d_qkv_gran + r0 * 3 * H; a.err = e->d_chain_err;
    a.x_stride = H; a.xn_stride = H; a.kv_row_stride = e->kv_row_bytes / e->kv_elem;
    a.xcd_local = e->opt_qkv_xcd_local;
    return a;
}
OprojFc1Args make_oproj_fc1_args(ma_engine* e, int l, const float* resid, Rows rw, bool with_fc2) {
    const ma_config& c = e->cfg;
    const int H = c.hidden; const size_t r0 = rw.r0;
    const DecLayerPtrs& w = e->dl[l];
    OprojFc1Args a{};
    a.Wo = reinterpret_cast<const bf16_t*>(w.o_w); a.bo = w.o_b; a.W1 = reinterpret_cast<const bf16_t*>(w.fc1_w); a.b1 = w.fc1_b;
    a.ln_g = w.ln1_g; a.ln_b = w.ln1_b; a.ln_eps = 1e-5f; a.attn_ws = e->d_part + r0 * attn_workspace_floats(c.heads); a.heads = c.heads; a.res = resid;
    a.h1_out = e->d_h1 + r0 * H; a.ffn_out = e->d_ffn + r0 * c.ffn;
    a.st = e->d_st + r0; a.layer = l; a.gran = e->d_y1_gran + r0 * H; a.err = e->d_chain_err;
    a.res_stride = H; a.h1_stride = H; a.ffn_stride = c.ffn; a.sweep_waves = e->opt_oproj_fc1_sweep_waves;
    if (with_fc2) { a.W2 = reinterpret_cast<const bf16_t*>(w.fc2_w); a.b2 = w.fc2_b; a.y2_out = e->d_ypre2 + r0 * H; a.y2_stride = H; a.gran2 = e->d_ffn_gran + r0 * c.ffn; }
    return a;
}

// parts: 1 = the layer's first half (LayerNorm + q/k/v + attention), 2 = its second half (out_proj .. fc2), 3 = both
void enqueue_layer(ma_engine* e, hipStream_t s, int l, const float* x_in, const float* ln_g, const float* ln_b, int len_override, StepTimer& tm, Rows rw, int parts = 3) {
    const ma_config& c = e->cfg;
    const int H = c.hidden, B = rw.B;
    const size_t r0 = rw.r0;
    const DecLayerPtrs& w = e->dl[l];
    float* h0 = e->d_h0 + r0 * H; float* q = e->d_q + r0 * H; float* y1 = e->d_ypre1 + r0 * H; float* y2 = e->d_ypre2 + r0 * H;
    float* h1 = e->d_h1 + r0 * H; float* ffn = e->d_ffn + r0 * c.ffn; float* part = e->d_part + r0 * attn_workspace_floats(c.heads);
    const float* resid = ln_g ? h0 : x_in;
    const size_t kv_row_elems = e->kv_row_bytes / e->kv_elem;
    if (!(parts & 1)) {
    } else if (fuse_qkv_attn(e, B, len_override)) {
        // q, k, v projection + split-KV attention in one launch (qkv_attn.hpp): the exchange between them stays inside a head
        QkvAttnArgs a = make_qkv_attn_args(e, l, x_in, ln_g, ln_b, len_override, rw);
        a.trace = tm.trace_slot(2, ATTN_NCHUNK * c.heads);
        if (tm.on(1)) {
            hipError_t r = e->bf16 ? H16_CALL(e->hdt, HT, launch_qkv_attn<HT>(a, c.heads, B, s)) : launch_qkv_attn<float>(a, c.heads, B, s);
            if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("qkv_attn launch failed: ") + hipGetErrorString(r));
        }
    } else {
    {   // q,k,v = W h + b ; k,v appended to the cache in place ([3p] OPTAttention; 4.39.3 grows it with torch.cat)
        GemvArgs a = gemv_base(e, rw);
        a.W = w.qkv_w; a.bias = w.qkv_b; a.x = x_in; a.x_stride = H; a.ln_g = ln_g; a.ln_b = ln_b; a.ln_eps = 1e-5f; a.xn_out = ln_g ? h0 : nullptr; a.xn_stride = H;
        a.y = q; a.y_stride = H; a.N = 3 * H; a.K = H; a.epi = EPI_QKV; a.kcache = e->kplane(rw.r0, l); a.vcache = e->vplane(rw.r0, l); a.kv_row_stride = kv_row_elems;
        a.H = H; a.max_seq = e->maxseq;
        a.trace = tm.trace_slot(1, gemv_blocks(e, a.N, a.K));
        if (tm.on(0)) gemv(e, a, s, B);
    }
    if (tm.on(1)) {
        unsigned long long* tr = tm.trace_slot(2, ATTN_NCHUNK * c.heads);
        hipError_t r = e->bf16 ? H16_CALL(e->hdt, HT, launch_attn_decode<HT>(q, e->kplane(rw.r0, l), e->vplane(rw.r0, l), c.heads, e->maxseq, e->d_st + r0, len_override, 1, part, s, tr, B, H, kv_row_elems))
                               : launch_attn_decode<float>(q, e->kplane(rw.r0, l), e->vplane(rw.r0, l), c.heads, e->maxseq, e->d_st + r0, len_override, 0, part, s, tr, B, H, kv_row_elems);
        if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("attn_decode launch failed: ") + hipGetErrorString(r));
    }
    }
    bool fc2_done = false;
    if (!(parts & 2)) return;
    if (fuse_oproj_fc1(e, B, len_override)) {
        // y1 = h + Wo a + bo; h1 = LN1(y1); f = relu(W1 h1 + b1) [; y2 = h1 + W2 f + b2] in one launch: y1 (and f) all-gathered inside it (oproj_fc1.hpp)
        const bool with_fc2 = e->opt_fuse_fc2 != 0;
        fc2_done = with_fc2;
        OprojFc1Args a = make_oproj_fc1_args(e, l, resid, rw, with_fc2);
        a.trace = tm.trace_slot(3, H / 4);
        if (tm.on(0)) {
            hipError_t r = e->bf16 ? H16_CALL(e->hdt, HT, launch_oproj_fc1<HT>(a, H, c.ffn, B, s)) : launch_oproj_fc1<float>(a, H, c.ffn, B, s);
            if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("oproj_fc1 launch failed: ") + hipGetErrorString(r));
        }
    } else {
    {   // y1 = h + Wo a + bo  (LayerNorm deferred to the consumer's prologue)
        GemvArgs a = gemv_base(e, rw);
        // the attention output is never materialised: this GEMV's prologue merges the split-KV partials
        a.W = w.o_w; a.bias = w.o_b; a.x = nullptr; a.attn_ws = part; a.attn_ws_stride = attn_workspace_floats(c.heads); a.attn_heads = c.heads;
        a.res = resid; a.res_stride = H; a.y = y1; a.y_stride = H; a.N = H; a.K = H;
        a.trace = tm.trace_slot(3, gemv_blocks(e, a.N, a.K));
        if (tm.on(0)) gemv(e, a, s, B);
    }
    {   // f = relu(W1 LN1(y1) + b1); h1 = LN1(y1) kept for the residual
        GemvArgs a = gemv_base(e, rw);
        a.W = w.fc1_w; a.bias = w.fc1_b; a.x = y1; a.x_stride = H; a.ln_g = w.ln1_g; a.ln_b = w.ln1_b; a.ln_eps = 1e-5f; a.xn_out = h1; a.xn_stride = H;
        a.y = ffn; a.y_stride = c.ffn; a.N = c.ffn; a.K = H; a.act = ACT_RELU;
        a.trace = tm.trace_slot(4, gemv_blocks(e, a.N, a.K));
        if (tm.on(0)) gemv(e, a, s, B);
    }
    }
    if (!fc2_done) {   // y2 = h1 + W2 f + b2
        GemvArgs a = gemv_base(e, rw);
        a.W = w.fc2_w; a.bias = w.fc2_b; a.x = ffn; a.x_stride = c.ffn; a.res = h1; a.res_stride = H; a.y = y2; a.y_stride = H; a.N = H; a.K = c.ffn;
        a.trace = tm.trace_slot(5, gemv_blocks(e, a.N, a.K));
        if (tm.on(0)) gemv(e, a, s, B);
    }
}

#ifdef MA_EXPERIMENTAL
// 2 .. 8 rows on the rows-looped two-launch layer: 256 blocks whatever the batch, so the residency condition is the batch-1 one
bool use_rows_fused(ma_engine* e, int B, int len_override) {
    const ma_config& c = e->cfg;
    return e->opt_rows_fused && e->rf_ok && e->chain_resident && len_override < 0 && e->bf16 && B >= std::max(2, e->opt_rows_fused_min) && B <= RF_MAX_ROWS &&
           c.hidden == 1024 && c.ffn == 4096 && c.heads * 64 == c.hidden && c.heads * ATTN_NCHUNK == 256 && c.layers <= 30;
}

void enqueue_layer_rows_fused(ma_engine* e, hipStream_t s, int l, const float* x_in, const float* ln_g, const float* ln_b, StepTimer& tm, Rows rw) {
    const ma_config& c = e->cfg;
    const int H = c.hidden;
    const size_t r0 = rw.r0;
    RowsFusedArgs A{};
    A.q = make_qkv_attn_args(e, l, x_in, ln_g, ln_b, -1, rw);
    A.q.trace = tm.trace_slot(2, ATTN_NCHUNK * c.heads);
    const float* resid = ln_g ? e->d_h0 + r0 * H : x_in;
    A.o = make_oproj_fc1_args(e, l, resid, rw, true);
    A.o.trace = tm.trace_slot(3, H / 4);
    A.part_gran = e->d_part_gran + r0 * c.heads * ATTN_NCHUNK * RF_PART;
    A.attn_out = e->d_xb + r0 * H; A.attn_out_stride = H;
    A.B = rw.B;
    if (tm.on(1)) {
        hipError_t r = launch_qkv_attn_rows(A, c.heads, s);
        if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("qkv_attn_rows launch failed: ") + hipGetErrorString(r));
    }
    if (tm.on(0)) {
        hipError_t r = launch_oproj_fc1_rows(A, H, c.ffn, s);
        if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("oproj_fc1_rows launch failed: ") + hipGetErrorString(r));
    }
}

bool fuse_layer(ma_engine* e, int B = 1, int len_override = -1) { return e->opt_fuse_layer && e->bf16 && e->hdt == MA_DTYPE_BF16 && fuse_qkv_attn(e, B, len_override) && fuse_oproj_fc1(e, B, len_override) && e->opt_fuse_fc2; }

// second half of layer l + first half of layer l + 1 in one launch (layer_fused.hpp); belongs to the "cache" class of the profiler
void enqueue_layer_pair(ma_engine* e, hipStream_t s, int l, const float* resid, int len_override, StepTimer& tm, Rows rw) {
    const ma_config& c = e->cfg;
    LayerFusedArgs a{};
    a.o = make_oproj_fc1_args(e, l, resid, rw, true);
    a.q = make_qkv_attn_args(e, l + 1, nullptr, e->dl[l].ln2_g, e->dl[l].ln2_b, len_override, rw);
    a.gran3 = e->d_y2_gran + (size_t)rw.r0 * c.hidden;
    if (tm.on(1)) {
        hipError_t r = launch_layer_fused(a, c.hidden, c.ffn, c.heads, rw.B, s);
        if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("layer_fused launch failed: ") + hipGetErrorString(r));
    }
}
#else
bool use_rows_fused(ma_engine*, int, int) { return false; }
bool fuse_layer(ma_engine*, int = 1, int = -1) { return false; }
#endif

void enqueue_lm_head(ma_engine* e, hipStream_t s, const float* x, int x_stride, const float* ln_g, const float* ln_b, StepTimer& tm, Rows rw) {
    GemvArgs a = gemv_base(e, rw);
    a.W = e->P("transformer.lm_head.weight"); a.x = x; a.x_stride = x_stride; a.ln_g = ln_g; a.ln_b = ln_b; a.ln_eps = 1e-5f;
    a.y = e->d_logits + (size_t)rw.r0 * e->V; a.y_stride = e->V; a.N = e->V; a.K = e->cfg.hidden; a.epi = EPI_LMHEAD;
    a.part_val = e->d_pval + (size_t)rw.r0 * e->V; a.part_idx = e->d_pidx + (size_t)rw.r0 * e->V; a.part_stride = e->V;
    a.trace = tm.trace_slot(6, gemv_blocks(e, a.N, a.K));
    if (tm.on(0)) gemv(e, a, s, rw.B);
}

void enqueue_pick(ma_engine* e, hipStream_t s, StepTimer& tm, Rows rw) {
    if (!tm.on(3)) return;
    hipLaunchKernelGGL(pick_kernel, dim3(rw.B), dim3(256), (size_t)e->V * sizeof(float), s, e->d_logits + (size_t)rw.r0 * e->V, e->V,
                       e->d_pval + (size_t)rw.r0 * e->V, e->d_pidx + (size_t)rw.r0 * e->V, (use_mfma_decode(e, rw.B) && !use_rows_fused(e, rw.B, -1)) ? 0 : e->n_parts, e->V, e->d_st + rw.r0,
                       e->w_tokens + (size_t)rw.r0 * e->maxnew, e->maxnew, e->T);
    HIP_CHECK(hipGetLastError());
}

#ifdef MA_EXPERIMENTAL
// ---- persistent decode step (persist.hpp) ----------------------------------------------------------------------------------
// eligible: bf16 policy, one row, greedy, the 350M layer shape, a device with exactly the 256 CUs the kernel assigns roles to
bool persist_eligible(ma_engine* e, int B, int do_sample) { return e->persist_shape && B == 1 && !do_sample; }
bool persist_selected(ma_engine* e, int B, int do_sample) { return e->opt_decode_impl == 1 && persist_eligible(e, B, do_sample); }

// embedding table of the persistent step: row v = input_layer(codebook[v]) + bias, computed by the launch chain's own GEMV
// (same kernel, same rounding points: the table holds exactly the bits the chain's embedding launch produces for token v + 3)
void ensure_embtab(ma_engine* e, hipStream_t s) {
    if (e->embtab_ready) return;
    const ma_config& c = e->cfg;
    const float* cb = e->PF(DEC + "quantize_codebooks");
    for (int v = 0; v < c.codebook_size; ++v) {
        GemvArgs a{};
        a.round_x = 1; a.act = ACT_NONE; a.epi = EPI_PLAIN;
        a.W = e->P(DEC + "input_layer.weight"); a.bias = e->PF(DEC + "input_layer.bias"); a.x = cb + (size_t)v * c.codebook_dim;
        a.y = e->d_embtab + (size_t)v * c.hidden; a.N = c.hidden; a.K = c.codebook_dim;
        hipError_t r = launch_gemv<bf16_t>(a, s, 1);
        if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("embedding table gemv failed: ") + hipGetErrorString(r));
    }
    HIP_CHECK(hipStreamSynchronize(s));
    e->embtab_ready = true;
}

void enqueue_persist_step(ma_engine* e, hipStream_t s, StepTimer& tm, u64* trace = nullptr) {
    if (!tm.on(2)) return;
    const ma_config& c = e->cfg;
    PersistArgs a{};
    a.layers = e->d_layers; a.L = c.layers;
    a.lm_head = reinterpret_cast<const bf16_t*>(e->P("transformer.lm_head.weight")); a.V = e->V;
    a.embtab = e->d_embtab; a.extra = e->PF(DEC + "extra_embeds.weight"); a.tokpos = e->PF(DEC + "token_embed_positions.weight");
    a.cond = e->PF(DEC + "cond_embed.weight"); a.postab = e->PF(DEC + "embed_positions.weight"); a.T = e->T;
    a.kv = reinterpret_cast<bf16_t*>(e->kv); a.kv_plane = e->kv_plane / e->kv_elem; a.max_seq = e->maxseq;
    a.st = e->d_st; a.tokens_out = e->w_tokens; a.logits = e->d_logits;
    a.gran = e->d_gran; a.serial = e->d_serial; a.err = e->d_err; a.trace = trace;
    hipError_t r = launch_persist_decode(a, s);
    if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("persistent decode launch failed: ") + hipGetErrorString(r));
}

// the persistent step reports a bounded wait that expired through a device word: turn it into an error (and clear it)
void check_persist_error(ma_engine* e, hipStream_t s) {
    if (!e->persist_shape) return;
    HIP_CHECK(hipMemcpyAsync(e->h_err, e->d_err, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (*e->h_err) {
        const unsigned code = *e->h_err;
        HIP_CHECK(hipMemsetAsync(e->d_err, 0, sizeof(unsigned), s));
        throw MaError(MA_ERR_HIP, "persistent decode step: a bounded wait expired (code " + std::to_string(code) +
                                  ": 1 loader, 2 comm, 4 compute, 8 gather) -- the 256 workgroups were not all resident, or a hand-off was lost");
    }
}
#else
bool persist_selected(ma_engine*, int, int) { return false; }
void ensure_embtab(ma_engine*, hipStream_t) {}
void check_persist_error(ma_engine*, hipStream_t) {}
#endif

// the fused q/k/v + attention launch reports an expired (bounded) granule sweep through a device word
void check_chain_error(ma_engine* e, hipStream_t s) {
    HIP_CHECK(hipMemcpyAsync(e->h_chain_err, e->d_chain_err, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (*e->h_chain_err) {
        e->xchg_last_code = (int)*e->h_chain_err;            // which sweep(s) gave up: QA_ERR_GATHER 16 | OF_ERR_GATHER 32 | attention pair 64 | RA_ERR_QKV 256 | RA_ERR_OUT 512 | RM_ERR_FFN 1024 ...
        HIP_CHECK(hipMemsetAsync(e->d_chain_err, 0, sizeof(unsigned), s));
        throw ChainTimeout("a fused decode launch's in-launch exchange timed out (not all blocks of the grid resident?)");
    }
}

// One full decode step (shape_opt.py:318-328 embedding branch -> 24 layers -> lm_head -> pick) for rows r0..r0+B-1.
// Replayable: no host-side step-dependent argument.
void enqueue_decode_step(ma_engine* e, hipStream_t s, int len_override, StepTimer& tm, Rows rw = Rows{}, int impl = 0) {
#ifdef MA_EXPERIMENTAL
    if (impl == 1) { enqueue_persist_step(e, s, tm); return; }
#else
    if (impl != 0) throw MaError(MA_ERR_STATE, "the persistent decode step is not part of this build (MA_EXPERIMENTAL)");
#endif
    const ma_config& c = e->cfg;
    const int H = c.hidden;
    float* de = e->d_e + (size_t)rw.r0 * H;
    {
        GemvArgs a = gemv_base(e, rw);
        a.W = e->P(DEC + "input_layer.weight"); a.bias = e->PF(DEC + "input_layer.bias"); a.y = de; a.y_stride = H; a.N = H; a.K = c.codebook_dim;
        a.epi = EPI_EMBED; a.codebook = e->PF(DEC + "quantize_codebooks"); a.extra = e->PF(DEC + "extra_embeds.weight");
        a.tokpos = e->PF(DEC + "token_embed_positions.weight"); a.cond = e->PF(DEC + "cond_embed.weight");
        a.postab = e->PF(DEC + "embed_positions.weight"); a.T = e->T;
        // batched matrix-core step: the embedding launch also leaves the 16-bit operand of layer 0's q/k/v GEMM (no prologue launch for it)
        if (use_mfma_decode(e, rw.B)) { a.yb = e->d_xb + (size_t)rw.r0 * H; a.yb_stride = H; }
        a.trace = tm.trace_slot(0, gemv_blocks(e, a.N, a.K));
        if (tm.on(0)) gemv(e, a, s, rw.B);
    }
#ifdef MA_EXPERIMENTAL
    if (use_rows_fused(e, rw.B, len_override)) {
        const float* y2 = e->d_ypre2 + (size_t)rw.r0 * H;
        for (int l = 0; l < c.layers; ++l) {
            if (l == 0) enqueue_layer_rows_fused(e, s, 0, de, nullptr, nullptr, tm, rw);
            else enqueue_layer_rows_fused(e, s, l, y2, e->dl[l - 1].ln2_g, e->dl[l - 1].ln2_b, tm, rw);
        }
        enqueue_lm_head(e, s, y2, H, e->dl[c.layers - 1].ln2_g, e->dl[c.layers - 1].ln2_b, tm, rw);
    } else
#endif
    if (use_mfma_decode(e, rw.B)) {
        enqueue_layers_mfma(e, s, de, len_override, tm, rw);
    } else {
        const float* y2 = e->d_ypre2 + (size_t)rw.r0 * H;
#ifdef MA_EXPERIMENTAL
        if (fuse_layer(e, rw.B, len_override) && c.layers >= 2) {
            // first half of layer 0 | (second half of l + first half of l + 1) x (L - 1) | second half of layer L - 1
            const float* h0 = e->d_h0 + (size_t)rw.r0 * H;
            enqueue_layer(e, s, 0, de, nullptr, nullptr, len_override, tm, rw, 1);
            for (int l = 0; l + 1 < c.layers; ++l) enqueue_layer_pair(e, s, l, l == 0 ? de : h0, len_override, tm, rw);
            enqueue_layer(e, s, c.layers - 1, y2, e->dl[c.layers - 2].ln2_g, e->dl[c.layers - 2].ln2_b, len_override, tm, rw, 2);
        } else
#endif
        for (int l = 0; l < c.layers; ++l) {
            if (l == 0) enqueue_layer(e, s, 0, de, nullptr, nullptr, len_override, tm, rw);
            else enqueue_layer(e, s, l, y2, e->dl[l - 1].ln2_g, e->dl[l - 1].ln2_b, len_override, tm, rw);
        }
        enqueue_lm_head(e, s, y2, H, e->dl[c.layers - 1].ln2_g, e->dl[c.layers - 1].ln2_b, tm, rw);
    }
    enqueue_pick(e, s, tm, rw);
}

void drop_graphs(ma_engine* e) {
    for (auto& kv : e->gexec) if (kv.second) (void)hipGraphExecDestroy(kv.second);
    for (auto& kv : e->graph) if (kv.second) (void)hipGraphDestroy(kv.second);
    e->gexec.clear(); e->graph.clear();
}

// ---- row groups ------------------------------------------------------------------------------------------------------------
// Between 8 and 32 rows the matrix-core chain is bound by the LATENCY of its ~100-150 dependent launches per step (6-7 us each for
// 1-2 us worth of weight bytes), not by HBM.  The rows of a batch never interact, so the batch can be cut into G row groups that
// step independently, each on its own HIP stream with its own captured graph, meeting at the end of a burst of steps where the host
// reads the finished flags.  Group g reproduces an ungrouped run of its rows bit for bit (tests/test_gpu_pipeline.py).
// MEASURED SLOWER, so opt-in (option decode_groups, default one group): two dependent chains on two hardware queues overlap only about
// half-way -- 8 rows as 2 x 4: 1522 vs 1290 us per step at kv 3858, 16 rows as 2 x 8: 1921 vs 1803; 3-4 groups: 1.4-2x slower;
// +2..5 % only at 24-64 rows and long caches (profiles/r03_ab_row_groups.txt).
int decode_group_count(ma_engine* e, int B, int impl) {
    if (impl != 0 || !use_mfma_decode(e, B) || use_rows_fused(e, B, -1)) return 1;
    const int min_rows = std::max(4, e->opt_mfma_min_batch);             // every group stays on the matrix-core path
    const int G = std::max(1, e->opt_decode_groups);
    return std::max(1, std::min(G, B / min_rows));
}
std::vector<Rows> decode_groups(ma_engine* e, int B, int impl) {
    const int G = decode_group_count(e, B, impl);
    std::vector<Rows> g;
    for (int i = 0, r0 = 0; i < G; ++i) { const int n = B / G + (i < B % G ? 1 : 0); g.push_back(Rows{r0, n}); r0 += n; }
    return g;
}
void ensure_group_streams(ma_engine* e, size_t G) {
    if (!e->grp_fork) HIP_CHECK(hipEventCreateWithFlags(&e->grp_fork, hipEventDisableTiming));
    while (e->grp_stream.size() < G) {
        hipStream_t st = nullptr; hipEvent_t ev = nullptr;
        HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        e->grp_stream.push_back(st);
        HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        e->grp_done.push_back(ev);
    }
}

// one captured step per row range (the grids depend on B, the pointers on r0) and step implementation
int graph_key(Rows rw, int impl) { return rw.B + 1000 * impl + 10000 * rw.r0; }
void ensure_graph(ma_engine* e, Rows rw, int impl = 0) {
    const int key = graph_key(rw, impl);
    if (!e->cfg.use_graph || e->gexec.count(key)) return;
    StepTimer none;
    if (!e->cap_stream) HIP_CHECK(hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking));
    hipStream_t s = e->cap_stream;
    hipGraph_t g = nullptr;
    HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    try {
        enqueue_decode_step(e, s, -1, none, rw, impl);
    } catch (...) {
        (void)hipStreamEndCapture(s, &g);
        if (g) (void)hipGraphDestroy(g);
        throw;
    }
    HIP_CHECK(hipStreamEndCapture(s, &g));
    hipGraphExec_t ge = nullptr;
    hipError_t r = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    if (r != hipSuccess) { (void)hipGraphDestroy(g); HIP_CHECK(r); }
    e->graph[key] = g; e->gexec[key] = ge;
}
void ensure_graphs(ma_engine* e, int B, int impl = 0) {
    for (const Rows& rw : decode_groups(e, B, impl)) ensure_graph(e, rw, impl);
}

void launch_step(ma_engine* e, hipStream_t s, Rows rw, int impl = 0) {
    if (e->cfg.use_graph) HIP_CHECK(hipGraphLaunch(e->gexec.at(graph_key(rw, impl)), s));
    else { StepTimer none; enqueue_decode_step(e, s, -1, none, rw, impl); }
}

// n decode steps of rows 0..B-1, ordered after what is on `s` and before what comes next on it
void launch_steps(ma_engine* e, hipStream_t s, int B, int impl, int n) {
    const std::vector<Rows> groups = decode_groups(e, B, impl);
    if (groups.size() == 1) { for (int i = 0; i < n; ++i) launch_step(e, s, groups[0], impl); return; }
    ensure_group_streams(e, groups.size());
    HIP_CHECK(hipEventRecord(e->grp_fork, s));
    for (size_t g = 0; g < groups.size(); ++g) HIP_CHECK(hipStreamWaitEvent(e->grp_stream[g], e->grp_fork, 0));
    for (int i = 0; i < n; ++i)
        for (size_t g = 0; g < groups.size(); ++g) launch_step(e, e->grp_stream[g], groups[g], impl);
    for (size_t g = 0; g < groups.size(); ++g) {
        HIP_CHECK(hipEventRecord(e->grp_done[g], e->grp_stream[g]));
        HIP_CHECK(hipStreamWaitEvent(s, e->grp_done[g], 0));
    }
}

// prefill of rows row0 .. row0+B-1 in ONE pass (the samples are stacked along the GEMM rows: M = B * T): ShapeOPTDecoder.forward
// inputs_embeds branch (shape_opt.py:331-364) + 24 post-LN layers, causal per sample, on the T prefix rows of every sample;
// fills the rows' KV planes and leaves each row's first logits in d_logits[row]
void prefill(ma_engine* e, hipStream_t s, const float* prefix, int row0, int B) {
    const ma_config& c = e->cfg;
    const int T = e->T, H = c.hidden, M = B * T;
    StepTimer none;
    float* h = e->p_h;                       // (B*T, H)
    add_rows(e, s, prefix, H, nullptr, e->PF(DEC + "cond_embed.weight"), e->PF(DEC + "embed_positions.weight"), H, 2, h, H, e->a_ph, H, M, H, T);
    if (e->opt_prefill_stepwise) {
        // debug path: feed the prefix rows through the decode-step kernels one position at a time, one sample at a time
        for (int b = 0; b < B; ++b) {
            const Rows rw{row0 + b, 1};
            for (int j = 0; j < T; ++j) {
                hipLaunchKernelGGL(set_pos_kernel, dim3(1), dim3(1), 0, s, e->d_st + row0 + b, 0, j, 0, 1);
                HIP_CHECK(hipGetLastError());
                for (int l = 0; l < c.layers; ++l) {
                    if (l == 0) enqueue_layer(e, s, 0, h + ((size_t)b * T + j) * H, nullptr, nullptr, -1, none, rw);
                    else enqueue_layer(e, s, l, e->d_ypre2 + (size_t)(row0 + b) * H, e->dl[l - 1].ln2_g, e->dl[l - 1].ln2_b, -1, none, rw);
                }
            }
            enqueue_lm_head(e, s, e->d_ypre2 + (size_t)(row0 + b) * H, H, e->dl[c.layers - 1].ln2_g, e->dl[c.layers - 1].ln2_b, none, rw);
        }
        return;
    }
    void* qkv = e->a_pqkv;                   // (B*T, 3H) activation
    void* att = e->a_patt;                   // (B*T, H) activation
    void* hb = e->a_ph;                      // (B*T, H) activation copy of h
    float* y = e->p_y;                       // (B*T, H)
    void* ffn = e->a_pffn;                   // (B*T, ffn) activation
    const size_t kv_row_elems = e->kv_row_bytes / e->kv_elem;
    // The last rows as a chain of their own (round 6).  M = B x 257 leaves M % 256 = B rows behind the 256-row tiles, and every GEMM of a layer ran them as a
    // launch of its own behind its tiles (the skinny GEMM: 6.5 us at 16 rows, 11 us at 64; with the K / V copy of those rows 25 - 47 us per layer = 9 - 12 % of the
    // prefill).  Those rows are the LAST B positions of the LAST sample: causal attention means no other row ever reads anything of theirs, so the rows in
    // front of them (`part 1`: exact tile rows, no tail launches) run all 24 layers without them, and they (`part 2`) follow on a second stream with the same
    // kernels the one-stream form gives them -- each of their layers needs from the main chain only that layer's K / V of the earlier positions (one event
    // per layer).  The main chain's attention still launches over all B x T query rows: the last B of them are stale rows of the q|k|v buffer, their output
    // goes to rows of `att` nobody reads (the tail chain has its own), and what they do to keys they can see but that the tail chain is still writing is
    // masked in every valid row (the planes are zeroed at creation, so a masked V is a finite number).  Same bits as the one-stream form.
    const int Mm = M - M % 256;
    const bool tail = e->opt_prefill_tail && e->bf16 && B >= 8 && M > Mm && M - Mm <= 64 && M - Mm <= T && e->a_patt_tail && attn2_vt_elems(T, c.heads, 1) <= e->vt_tail_elems;
    hipStream_t s2 = nullptr;
    if (tail) {
        if (!e->tail_stream) {
            HIP_CHECK(hipStreamCreateWithFlags(&e->tail_stream, hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&e->tail_fork, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&e->tail_join, hipEventDisableTiming));
            e->tail_kv.resize(c.layers);
            for (hipEvent_t& ev : e->tail_kv) HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        }
        s2 = e->tail_stream;
        if (e->opt_prefill_tail == 2) {
            if (!e->tail_stream_low) {
                int lo = 0, hi = 0;
                HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));            // (numerically: lo >= hi; lo = the least urgent)
                HIP_CHECK(hipStreamCreateWithPriority(&e->tail_stream_low, hipStreamNonBlocking, lo));
            }
            s2 = e->tail_stream_low;
        }
        HIP_CHECK(hipEventRecord(e->tail_fork, s));              // the embedded rows (h, hb) of every row are there
        HIP_CHECK(hipStreamWaitEvent(s2, e->tail_fork, 0));
    }
    const int mp = tail ? 1 : 0;                                     // the main chain's row part
    const int Mk = tail ? Mm : M;                                    // rows whose K / V the main chain puts into the planes
    for (int l = 0; l < c.layers; ++l) {
        const std::string p = DEC + "layers." + std::to_string(l) + ".";
        if (e->bf16) {
            // 16-bit policies: the K / V columns of the rows on the persistent 256 x 256 tiles go straight into the cache planes (gemm256.hpp, KV form);
            // the rows behind them (the 64-row tail of M = B x 257; every row when another kernel took the GEMM) are copied from the q|k|v tensor.
            // Attention then reads K, and the V^T packing V, from the planes: the cache IS the prefill's K / V operand.
            KvDst kv; kv.k = e->kplane(row0, l); kv.v = e->vplane(row0, l); kv.row_stride = kv_row_elems; kv.max_seq = e->maxseq; kv.T = T; kv.col0 = H;
            gemm(e, s, hb, H, p + "qkv.weight", p + "qkv.bias", nullptr, 0, toact(qkv, 3 * H), M, ACT_NONE, 0, e->opt_qkv_to_cache ? &kv : nullptr, nullptr, nullptr, mp);
            auto kv_fill = [&](hipStream_t st, int r_begin, int r_end) {
                const long n = (long)(r_end - r_begin) * c.heads * 8;
                hipLaunchKernelGGL((kv_fill_rows_kernel<bf16_t, bf16_t>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const bf16_t*>(qkv), 3 * H, H, 2 * H, r_begin, r_end, T,
                                   c.heads, e->maxseq, reinterpret_cast<bf16_t*>(kv.k), reinterpret_cast<bf16_t*>(kv.v), kv_row_elems);      // a copy of 16-bit words: either format
                HIP_CHECK(hipGetLastError());
            };
            if (kv.rows_done < Mk) kv_fill(s, kv.rows_done, Mk);
            if (tail) {
                // ---- the tail chain's layer l (stream s2): enqueued here, between the main chain's q|k|v and its attention ----
                HIP_CHECK(hipEventRecord(e->tail_kv[l], s));
                KvDst kv2 = kv;
                gemm(e, s2, hb, H, p + "qkv.weight", p + "qkv.bias", nullptr, 0, toact(qkv, 3 * H), M, ACT_NONE, 0, e->opt_qkv_to_cache ? &kv2 : nullptr, nullptr, nullptr, 2);
                kv_fill(s2, Mm, M);
                HIP_CHECK(hipStreamWaitEvent(s2, e->tail_kv[l], 0));
                const int nt = M - Mm;                               // its rows: positions T - nt .. T - 1 of sample B - 1
                void* att_t = e->a_patt_tail;
                attention(e, s2, aoff(e, qkv, (size_t)Mm * 3 * H), 3 * H, 64, e->kplane(row0 + B - 1, l), 64, e->maxseq * 64, e->vplane(row0 + B - 1, l), 64, e->maxseq * 64, att_t, H, nt, T,
                          c.heads, T - nt, 1, 0, 0, 0, 0, e->a_vt_tail, e->vt_tail_elems);
                // (row m of the A operand is read at A + m * lda: the address row 0 WOULD have)
                const void* att_t0 = reinterpret_cast<const char*>(att_t) - (size_t)Mm * H * e->act_elem;
                gemm_res_ln(e, s2, att_t0, H, p + "self_attn.out_proj.weight", p + "self_attn.out_proj.bias", p + "self_attn_layer_norm.", 1e-5f, h, hb, y, M, H, e->opt_gemm_splitk >= 2, 2);
                gemm(e, s2, hb, H, p + "fc1.weight", p + "fc1.bias", nullptr, 0, toact(ffn, c.ffn), M, ACT_RELU, 0, nullptr, nullptr, nullptr, 2);
                gemm_res_ln(e, s2, ffn, c.ffn, p + "fc2.weight", p + "fc2.bias", p + "final_layer_norm.", 1e-5f, h, hb, y, M, H, true, 2);
            }
            attention(e, s, qkv, 3 * H, 64, kv.k, 64, e->maxseq * 64, kv.v, 64, e->maxseq * 64, att, H, T, T, c.heads, 0, B, (size_t)T * 3 * H, kv_row_elems, kv_row_elems, (size_t)T * H);
        } else {
            gemm(e, s, hb, H, p + "qkv.weight", p + "qkv.bias", nullptr, 0, toact(qkv, 3 * H), M, ACT_NONE);
            const int n = T * c.heads * 64;
            hipLaunchKernelGGL((kv_fill2_kernel<float, float>), dim3(ceil_div(n, 256), B), dim3(256), 0, s, reinterpret_cast<const float*>(qkv), 3 * H, H, 2 * H, T, c.heads, e->maxseq,
                               reinterpret_cast<float*>(e->kplane(row0, l)), reinterpret_cast<float*>(e->vplane(row0, l)), kv_row_elems);
            HIP_CHECK(hipGetLastError());
            attention(e, s, qkv, 3 * H, 64, aoff(e, qkv, H), 3 * H, 64, aoff(e, qkv, 2 * H), 3 * H, 64, att, H, T, T, c.heads, 0, B, (size_t)T * 3 * H, (size_t)T * 3 * H,
                      (size_t)T * 3 * H, (size_t)T * H);
        }
        gemm_res_ln(e, s, att, H, p + "self_attn.out_proj.weight", p + "self_attn.out_proj.bias", p + "self_attn_layer_norm.", 1e-5f, h, hb, y, M, H, e->opt_gemm_splitk >= 2, mp);
        gemm(e, s, hb, H, p + "fc1.weight", p + "fc1.bias", nullptr, 0, toact(ffn, c.ffn), M, ACT_RELU, 0, nullptr, nullptr, nullptr, mp);
        // small batches: fc2's 256 x 256 tiles (N = hidden: four per tile row) fill a fraction of the chip while each runs 64 K-tiles -- split along K
        // into partial sums that the LayerNorm adds up (gemm256.hpp GemmSplitK; 16 samples: 64 tiles x 4 parts = one round of 16 K-tiles)
        gemm_res_ln(e, s, ffn, c.ffn, p + "fc2.weight", p + "fc2.bias", p + "final_layer_norm.", 1e-5f, h, hb, y, M, H, true, mp);
    }
    if (tail) {
        HIP_CHECK(hipEventRecord(e->tail_join, s2));
        HIP_CHECK(hipStreamWaitEvent(s, e->tail_join, 0));
    }
    // only the last prefix row of every sample feeds lm_head (the reference computes all 257 rows and discards 256, shape_opt.py:155)
    enqueue_lm_head(e, s, h + (size_t)(T - 1) * H, T * H, nullptr, nullptr, none, Rows{row0, B});
}

// state records of rows 0..B-1: identical except for the row id and the row's slice of the injected uniforms
void init_state(ma_engine* e, hipStream_t s, const ma_sample_cfg& sc, int B, int maxn) {
    DecState st{};
    st.t = 0; st.pos = e->T - 1; st.cur_tok = 0; st.finished = 0;
    st.suppress_eos = sc.suppress_eos; st.do_sample = sc.do_sample; st.top_k = sc.top_k; st.top_p = sc.top_p;
    st.seed = sc.seed; st.uniforms = sc.uniforms; st.row = 0; st.max_new = maxn;
    st.forced = reinterpret_cast<const long long*>(sc.forced_tokens); st.logits_out = sc.logits_out; st.logits_first = sc.logits_out ? sc.logits_first_step : 0;
    hipLaunchKernelGGL(init_state_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, s, e->d_st, st, B, e->V);
    HIP_CHECK(hipGetLastError());
    // a word left raised by a call that threw before reading it must not make this generation's sweeps give up early
    HIP_CHECK(hipMemsetAsync(e->d_chain_err, 0, sizeof(unsigned), s));
    // the fused q/k/v + attention launch tags its exchange with the cache position, which restarts here
    HIP_CHECK(hipMemsetAsync(e->d_qkv_gran, 0, (size_t)e->cfg.max_batch * 3 * e->cfg.hidden * sizeof(u64), s));
    HIP_CHECK(hipMemsetAsync(e->d_y1_gran, 0, (size_t)e->cfg.max_batch * e->cfg.hidden * sizeof(u64), s));
    HIP_CHECK(hipMemsetAsync(e->d_ffn_gran, 0, (size_t)e->cfg.max_batch * e->cfg.ffn * sizeof(u64), s));
    HIP_CHECK(hipMemsetAsync(e->d_attn_pair_gran, 0, (size_t)e->cfg.max_batch * e->cfg.heads * ATTN_PAIR_GRANULES * sizeof(unsigned long long), s));
    HIP_CHECK(hipMemsetAsync(e->d_y2_gran, 0, (size_t)e->cfg.max_batch * e->cfg.hidden * sizeof(u64), s));
    HIP_CHECK(hipMemsetAsync(e->d_ra_qkv_gran, 0, (size_t)e->cfg.max_batch * RA_QKV_GRANULES * sizeof(u64), s));
    HIP_CHECK(hipMemsetAsync(e->d_ra_out_gran, 0, (size_t)e->cfg.max_batch * RA_OUT_GRANULES * sizeof(u64), s));
    HIP_CHECK(hipMemsetAsync(e->d_rm_y2_gran, 0, (size_t)e->cfg.max_batch * RM_Y2_GRANULES * sizeof(u64), s));
#ifdef MA_EXPERIMENTAL
    HIP_CHECK(hipMemsetAsync(e->d_part_gran, 0, (size_t)e->cfg.max_batch * e->cfg.heads * ATTN_NCHUNK * RF_PART * sizeof(u64), s));
#endif
}

ma_sample_cfg resolve_sample_cfg(ma_engine* e, const ma_sample_cfg* sc) {
    ma_sample_cfg r{};
    r.struct_size = sizeof(ma_sample_cfg);
    r.top_k = 50; r.top_p = 0.95f;
    if (sc) {
        if (sc->struct_size != (int32_t)sizeof(ma_sample_cfg)) throw MaError(MA_ERR_INVALID, "ma_sample_cfg.struct_size mismatch");
        r = *sc;
    }
    if (r.max_new_tokens <= 0 || r.max_new_tokens > e->maxnew) {
        if (r.max_new_tokens > e->maxnew) throw MaError(MA_ERR_INVALID, "max_new_tokens exceeds 9*n_max_faces+2");
        r.max_new_tokens = e->maxnew;
    }
    if (r.check_every <= 0) r.check_every = 64;
    if (!r.logits_out) r.logits_first_step = 0;                          // (no effect without logits_out: whatever the caller left there is ignored)
    else if (r.logits_first_step < 0 || r.logits_first_step >= r.max_new_tokens) throw MaError(MA_ERR_INVALID, "logits_first_step must be in [0, max_new_tokens) when logits_out is set");
    if (r.do_sample && (r.top_k < 1 || r.top_k > PICK_KMAX)) throw MaError(MA_ERR_INVALID, "top_k must be in [1,64]");
    if (r.do_sample && !(r.top_p > 0.f && r.top_p <= 1.f)) throw MaError(MA_ERR_INVALID, "top_p must be in (0,1]");
    return r;
}

// generate() for a batch of B rows: every row is prefilled, then all rows step together (they share the weight stream and
// the cache position) until every row has emitted eos or max_new_tokens ([3p] GenerationMixin: a finished row keeps
// stepping and emits pad).  tokens_out (B, maxnew) device; lengths host (B).  Returns the number of valid columns.
int generate_batch_once(ma_engine* e, hipStream_t s, const float* prefix, int B, const ma_sample_cfg& sc, long long* tokens_out, int32_t* lengths) {
    const int maxn = sc.max_new_tokens;
    const int total = B * e->maxnew;
    hipLaunchKernelGGL(fill_tokens_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, e->w_tokens, (long long)TOK_PAD, total);
    HIP_CHECK(hipGetLastError());
    // (the persistent step carries its own pick: teacher forcing / logits capture take the launch chain)
    const bool parity_aid = sc.forced_tokens || sc.logits_out;
    const int impl = (persist_selected(e, B, sc.do_sample) && !parity_aid) ? 1 : 0;
    if (impl == 1) ensure_embtab(e, s);
    ensure_graphs(e, B, impl);
    init_state(e, s, sc, B, maxn);
    // prefill in groups of rows (bounded workspace: prefill_rows samples at a time)
    {
        RoctxRange range("ma_generate: prefill");
        for (int b0 = 0; b0 < B; b0 += e->prefill_rows) {
            const int nb = std::min(e->prefill_rows, B - b0);
            prefill(e, s, prefix + (size_t)b0 * e->T * e->cfg.hidden, b0, nb);
        }
    }
    RoctxRange range_decode("ma_generate: decode steps");
    if (e->opt_prefill_stepwise) init_state(e, s, sc, B, maxn);         // the stepwise prefill used the state's pos field
    StepTimer none;
    enqueue_pick(e, s, none, Rows{0, B});                                // token 0 (expected bos; dropped later, meshanything.py:166)
    int produced = 1;
    bool finished = false;
    while (produced < maxn && !finished) {
        // the first burst is ONE step: a grid whose blocks are not all resident (device shared with another stream / process) shows
        // in the error word after the first fused launch, not after 64 steps of zero-filled exchanges
        const int burst = std::min(produced == 1 ? 1 : sc.check_every, maxn - produced);
        launch_steps(e, s, B, impl, burst);
        produced += burst;
        HIP_CHECK(hipMemcpyAsync(e->h_state, e->d_st, (size_t)B * sizeof(DecState), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        if (impl == 1) check_persist_error(e, s);
        else check_chain_error(e, s);
        finished = true;
        for (int b = 0; b < B; ++b) finished = finished && e->h_state[b].finished != 0;
    }
    if (produced == 1 && impl == 0) check_chain_error(e, s);            // (no decode step ran: the prefill's own in-launch exchange -- gemm256.hpp LNF form -- is checked here)
    HIP_CHECK(hipMemcpyAsync(e->h_tokens, e->w_tokens, (size_t)total * sizeof(long long), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    int nmax = 0;
    for (int b = 0; b < B; ++b) {
        const long long* row = e->h_tokens + (size_t)b * e->maxnew;
        int len = produced;
        // (teacher forcing: the row reports the engine's picks along the GIVEN stream, every produced column counts)
        if (!sc.forced_tokens) for (int i = 0; i < produced; ++i) if (row[i] == TOK_EOS) { len = i + 1; break; }
        if (lengths) lengths[b] = len;
        nmax = std::max(nmax, len);
    }
    if (tokens_out != e->w_tokens) HIP_CHECK(hipMemcpyAsync(tokens_out, e->w_tokens, (size_t)total * sizeof(long long), hipMemcpyDeviceToDevice, s));
    return nmax;
}

// A timed-out in-launch exchange is not an error of the request: the fused launches need the whole grid resident, which another
// stream or process on the device can take away at any time.  The engine then stops using them (chain_resident = false: five
// launches per layer, no spinning, bit-identical results -- tests/test_gpu_persist.py) and runs the generation again from the prefill.
// One co-tenant blip must not cost the fused launches for the rest of the engine's life: after CHAIN_REARM_AFTER clean
// generations on the five-launch chain the engine re-arms them; if the device is still shared, the next generation pays one more
// 20 ms deadline and falls back again.  chain_fallbacks / xchg_timeouts (get_option) and the bench line show what happened.
constexpr int CHAIN_REARM_AFTER = 16;
int generate_batch(ma_engine* e, hipStream_t s, const float* prefix, int B, const ma_sample_cfg& sc, long long* tokens_out, int32_t* lengths) {
    if (!e->chain_resident && e->chain_fallbacks > 0 && e->resident_blocks * 4 >= 256L * 5 && ++e->gens_since_fallback > CHAIN_REARM_AFTER) {
        e->chain_resident = true; e->gens_since_fallback = 0;
        drop_graphs(e);
    }
    try {
        return generate_batch_once(e, s, prefix, B, sc, tokens_out, lengths);
    } catch (const ChainTimeout&) {
        if (!e->chain_resident) throw;
        e->chain_resident = false;
        e->chain_fallbacks++;
        e->gens_since_fallback = 0;
        drop_graphs(e);
    }
    return generate_batch_once(e, s, prefix, B, sc, tokens_out, lengths);
}

// ------------------------------------------------------------------------------------------------ detokenizer
// NoiseResistantDecoder.forward (meshanything.py:50-80) for nb samples stacked along the rows: X (nb, S = T + nf, Wt).
// codes != null: the caller's `input_embeds` (B, 3 nf, D) fp32 are used as the face codes (what the reference's signature
// takes); null: they are gathered from the codebook (get_codes, meshanything.py:178-212) inside the chain.
void detok_chunk(ma_engine* e, hipStream_t s, const long long* ids, const float* codes, const float* latents, float* coords, int nb) {
    const ma_config& c = e->cfg;
    const int W = c.enc_width, T = e->T, Wt = c.tok_width, nf = e->nf, S = e->S, D = c.codebook_dim, Hh = c.tok_heads;
    const int rowsS = nb * S, rowsF = nb * nf;
    float* X = e->w_x;                                               // (nb * S, Wt) fp32
    void* Xb = e->a_x;                                               // activation copy
    const RowMap head_in{1, T, 0}, tail_in{T - 1, T, 1};             // latents[:, 0] / [:, 1:] inside the T-row blocks
    const RowMap cond_out{T, S, 0}, face_out{nf, S, T};              // cond rows / face rows inside the S-row blocks of X
    // process_point_feature (meshanything.py:42-48): -> w_pf (nb * T, Wt); the projection of the encoder's latents keeps the encoder's precision
    {
        DenseScope enc(e, e->dense16 && !e->enc_exact);
        cvt_rows(e, s, latents, W, head_in, nullptr, e->a_ln, W, nb, W);
        gemm(e, s, e->a_ln, W, TOK + "cond_head_proj.weight", TOK + "cond_head_proj.bias", nullptr, 0, to32(e->w_pf, Wt, RowMap{1, T, 0}), nb, ACT_NONE);
        cvt_rows(e, s, latents, W, tail_in, nullptr, e->a_ln, W, nb * (T - 1), W);
        gemm(e, s, e->a_ln, W, TOK + "cond_proj.weight", TOK + "cond_proj.bias", nullptr, 0, to32(e->w_pf, Wt, RowMap{T - 1, T, 1}), nb * (T - 1), ACT_NONE);
    }
    add_rows(e, s, e->w_pf, Wt, nullptr, nullptr, e->PF(TOK + "point_pe.weight"), Wt, 0, e->w_pf, Wt, nullptr, 0, nb * T, Wt, T);
    lnrows(e, s, e->w_pf, Wt, TOK + "point_layernorm.", 1e-5f, X, Wt, Xb, Wt, nb * T, Wt, RowMap{0, 0, 0}, cond_out);
    // faces (meshanything.py:53-60): codes -> project_down -> zero masked -> + pos -> LN
    {
        const int total = rowsF * 3 * (D / 4);               // four consecutive d per thread
        if (e->dense16) H16_DO(e->hdt, HT, hipLaunchKernelGGL((codes_gather2_kernel<HT>), dim3(ceil_div(total, 256)), dim3(256), 0, s, ids, e->PF(DEC + "quantize_codebooks"), D, rowsF, (float*)nullptr,
                                        codes ? nullptr : reinterpret_cast<HT*>(e->a_fein), e->w_mask));
        else hipLaunchKernelGGL((codes_gather2_kernel<float>), dim3(ceil_div(total, 256)), dim3(256), 0, s, ids, e->PF(DEC + "quantize_codebooks"), D, rowsF, (float*)nullptr,
                                codes ? nullptr : reinterpret_cast<float*>(e->a_fein), e->w_mask);
        HIP_CHECK(hipGetLastError());
        if (codes) cvt_rows(e, s, codes, 3 * D, RowMap{0, 0, 0}, nullptr, e->a_fein, 3 * D, rowsF, 3 * D);     // 'b (nf nv) d -> b nf (nv d)' is a view
    }
    gemm(e, s, e->a_fein, 3 * D, TOK + "project_down_codebook.weight", TOK + "project_down_codebook.bias", nullptr, 0, to32(e->w_fe, Wt), rowsF, ACT_NONE);
    add_rows(e, s, e->w_fe, Wt, e->w_mask, nullptr, e->PF(TOK + "pos_embedding.weight"), Wt, 0, e->w_fe, Wt, nullptr, 0, rowsF, Wt, nf);
    lnrows(e, s, e->w_fe, Wt, TOK + "layernorm.", 1e-5f, X, Wt, Xb, Wt, rowsF, Wt, RowMap{0, 0, 0}, face_out);
    // 6 BERT post-LN layers, bidirectional, NO mask: padding faces take part as LN(pos_embedding[i]) tokens
    void* qkv = e->a_qkv; void* att = e->a_att; float* y = e->w_y; void* ffn = e->a_mlp;
    for (int n = 0; n < c.tok_layers; ++n) {
        const std::string p = TOK + "decoder.layer." + std::to_string(n) + ".";
        gemm(e, s, Xb, Wt, p + "qkv.weight", p + "qkv.bias", nullptr, 0, toact(qkv, 3 * Wt), rowsS, ACT_NONE);
        attention(e, s, qkv, 3 * Wt, 64, aoff(e, qkv, Wt), 3 * Wt, 64, aoff(e, qkv, 2 * Wt), 3 * Wt, 64, att, Wt, S, S, Hh, -1, nb, (size_t)S * 3 * Wt, (size_t)S * 3 * Wt,
                  (size_t)S * 3 * Wt, (size_t)S * Wt);
        gemm(e, s, att, Wt, p + "attention.output.dense.weight", p + "attention.output.dense.bias", X, Wt, to32(y, Wt), rowsS, ACT_NONE);
        lnrows(e, s, y, Wt, p + "attention.output.LayerNorm.", 1e-12f, X, Wt, Xb, Wt, rowsS, Wt);
        gemm(e, s, Xb, Wt, p + "intermediate.dense.weight", p + "intermediate.dense.bias", nullptr, 0, toact(ffn, c.tok_ffn), rowsS, ACT_GELU);
        gemm(e, s, ffn, c.tok_ffn, p + "output.dense.weight", p + "output.dense.bias", X, Wt, to32(y, Wt), rowsS, ACT_NONE);
        lnrows(e, s, y, Wt, p + "output.LayerNorm.", 1e-12f, X, Wt, Xb, Wt, rowsS, Wt);
    }
    // last_hidden_state[:, cond_length:], masked faces zeroed (meshanything.py:65-68) -> to_coor_logits
    cvt_rows(e, s, X, Wt, face_out, e->w_mask, e->a_ln, Wt, rowsF, Wt);
    gemm(e, s, e->a_ln, Wt, TOK + "to_coor_logits.0.weight", TOK + "to_coor_logits.0.bias", nullptr, 0, to32(e->w_logit, 9 * c.discrete_num), rowsF, ACT_NONE);
    hipLaunchKernelGGL(coords_argmax_kernel, dim3(ceil_div(rowsF * 9, 4)), dim3(256), 0, s, e->w_logit, rowsF, c.discrete_num, e->w_mask, coords);
    HIP_CHECK(hipGetLastError());
}

void require_ready(ma_engine* e) {
    if (!e->weights_ready) throw MaError(MA_ERR_STATE, "weights are not loaded (ma_engine_load_weights + ma_engine_finalize_weights, or ma_engine_mark_weights_loaded)");
}
void check_batch(ma_engine* e, int B) {
    if (B < 1 || B > e->cfg.max_batch) throw MaError(MA_ERR_INVALID, "batch size " + std::to_string(B) + " outside [1, max_batch=" + std::to_string(e->cfg.max_batch) + "]");
}

void validate_config(const ma_config& c) {
    auto bad = [](const std::string& m) { throw MaError(MA_ERR_INVALID, "ma_config: " + m); };
    if (c.struct_size != (int32_t)sizeof(ma_config)) bad("struct_size mismatch (header/library version skew)");
    if (c.enc_width != c.enc_heads * 64 || c.hidden != c.heads * 64 || c.tok_width != c.tok_heads * 64) bad("head_dim must be 64 (width = heads*64)");
    if (c.codebook_dim != c.hidden) bad("codebook_dim must equal hidden (word_embed_proj_dim is forced to hidden_size, meshanything.py:112-113)");
    if (c.dtype != MA_DTYPE_F32 && c.dtype != MA_DTYPE_BF16 && c.dtype != MA_DTYPE_F16) bad("dtype must be MA_DTYPE_F32, MA_DTYPE_BF16 or MA_DTYPE_F16");
    const int dims[] = {c.enc_width, c.hidden, c.ffn, c.tok_width, c.tok_ffn, c.embed_dim, c.codebook_dim};
    for (int d : dims) if (d <= 0 || d % 32) bad("GEMM dimensions must be positive multiples of 32");
    if (3 * (2 * c.num_freqs + 1) + 3 > 64 || c.num_freqs < 1 || c.num_freqs > 20) bad("num_freqs out of range");
    if (c.n_points < 1 || c.num_latents < 1 || c.layers < 1 || c.enc_layers < 0 || c.shape_layers < 0 || c.tok_layers < 0) bad("non-positive size");
    if (c.n_max_faces < 1 || c.n_max_faces > c.tok_max_pos) bad("n_max_faces out of range");
    if (c.num_latents + 1 + c.n_max_faces * 9 + 2 > c.max_positions) bad("max_positions too small for cond_length + 9*n_max_faces + 2");
    if (c.max_batch < 1 || c.kv_splits < 0 || c.discrete_num < 1 || c.codebook_size < 1) bad("policy field out of range");
    // pick_kernel parks the V = codebook_size + 3 logits in dynamic LDS next to ~19 KB of static LDS (64 KB per workgroup without opt-in)
    if ((size_t)(c.codebook_size + 3) * 4 + 20 * 1024 > 64 * 1024) bad("codebook_size too large for the sampler's LDS stage (max 11261)");
}

void build_engine(ma_engine* e) {
    const ma_config& c = e->cfg;
    e->L = build_layout(c);
    pack_state_init(e->L, e->ps);
    e->T = c.num_latents + 1; e->V = c.codebook_size + 3; e->maxnew = c.n_max_faces * 9 + 2; e->maxseq = e->T + e->maxnew;
    e->nf = c.n_max_faces; e->S = e->T + e->nf;
    e->bf16 = c.dtype != MA_DTYPE_F32; e->hdt = c.dtype == MA_DTYPE_F16 ? MA_DTYPE_F16 : MA_DTYPE_BF16; e->kv_elem = e->bf16 ? 2 : 4;
    HIP_CHECK(hipMalloc(&e->arena, e->L.bytes));
    HIP_CHECK(hipMemset(e->arena, 0, e->L.bytes));
    const size_t MB = c.max_batch;
    e->kv_plane = (size_t)c.heads * e->maxseq * 64 * e->kv_elem;
    e->kv_row_bytes = e->kv_plane * 2 * c.layers;
    HIP_CHECK(hipMalloc(&e->kv, e->kv_row_bytes * MB));
    HIP_CHECK(hipMemset(e->kv, 0, e->kv_row_bytes * MB));
    const int H = c.hidden;
    // decode-step buffers: one slice per batch row
    e->d_e = e->dmalloc<float>(MB * H); e->d_q = e->dmalloc<float>(MB * H);
    e->d_ypre1 = e->dmalloc<float>(MB * H); e->d_ypre2 = e->dmalloc<float>(MB * H); e->d_h0 = e->dmalloc<float>(MB * H); e->d_h1 = e->dmalloc<float>(MB * H);
    e->d_ffn = e->dmalloc<float>(MB * c.ffn); e->d_logits = e->dmalloc<float>(MB * e->V);
    e->d_part = e->dmalloc<float>(MB * attn_workspace_floats(c.heads));
    e->n_parts = e->bf16 ? gemv_num_blocks<bf16_t>(e->V, c.hidden) : gemv_num_blocks<float>(e->V, c.hidden);
    e->d_pval = e->dmalloc<float>(MB * e->V); e->d_pidx = e->dmalloc<int>(MB * e->V);        // row stride V >= blocks for any rows-per-block
    e->d_st = e->dmalloc<DecState>(MB);
    e->d_qkv_gran = e->dmalloc<u64>(MB * 3 * H); e->d_chain_err = e->dmalloc<unsigned>(12);      // [0] error bits (cleared when read), [1] expiries ever, [2] longest slow block (ticks), [3] slow blocks ever, [4] scalar sweeps rescued by a vector look (rows_attn.hpp)
    e->d_y1_gran = e->dmalloc<u64>(MB * H);
    HIP_CHECK(hipMemset(e->d_y1_gran, 0, MB * H * sizeof(u64)));
    e->d_attn_pair_gran = e->dmalloc<unsigned long long>(MB * c.heads * ATTN_PAIR_GRANULES);
    HIP_CHECK(hipMemset(e->d_attn_pair_gran, 0, MB * c.heads * ATTN_PAIR_GRANULES * sizeof(unsigned long long)));
    e->d_y2_gran = e->dmalloc<u64>(MB * H);
    HIP_CHECK(hipMemset(e->d_y2_gran, 0, MB * H * sizeof(u64)));
    e->d_ra_qkv_gran = e->dmalloc<u64>(MB * RA_QKV_GRANULES); e->d_ra_out_gran = e->dmalloc<u64>(MB * RA_OUT_GRANULES);
    HIP_CHECK(hipMemset(e->d_ra_qkv_gran, 0, MB * RA_QKV_GRANULES * sizeof(u64)));
    HIP_CHECK(hipMemset(e->d_ra_out_gran, 0, MB * RA_OUT_GRANULES * sizeof(u64)));
    e->d_pf_sink = e->dmalloc<unsigned>(4);
    e->d_rm_y2_gran = e->dmalloc<u64>(MB * RM_Y2_GRANULES);
    HIP_CHECK(hipMemset(e->d_rm_y2_gran, 0, MB * RM_Y2_GRANULES * sizeof(u64)));
    e->d_ffn_gran = e->dmalloc<u64>(MB * (size_t)c.ffn);
    HIP_CHECK(hipMemset(e->d_ffn_gran, 0, MB * (size_t)c.ffn * sizeof(u64)));
#ifdef MA_EXPERIMENTAL
    e->d_part_gran = e->dmalloc<u64>(MB * (size_t)c.heads * ATTN_NCHUNK * RF_PART);
    HIP_CHECK(hipMemset(e->d_part_gran, 0, MB * (size_t)c.heads * ATTN_NCHUNK * RF_PART * sizeof(u64)));
#endif
    HIP_CHECK(hipMemset(e->d_qkv_gran, 0, MB * 3 * H * sizeof(u64)));
    HIP_CHECK(hipMemset(e->d_chain_err, 0, 12 * sizeof(unsigned)));
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&e->h_chain_err), sizeof(unsigned)));
    e->d_xb = e->dmalloc<bf16_t>(MB * H); e->d_ffb = e->dmalloc<bf16_t>(MB * c.ffn);
    e->d_ks_o = e->dmalloc<float>(4 * MB * H); e->d_ks_f = e->dmalloc<float>(4 * MB * H);
    HIP_CHECK(hipMemset(e->d_st, 0, MB * sizeof(DecState)));
    {   // persistent decode step: shape / device eligibility and its buffers
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, e->device));
        e->n_cus = prop.multiProcessorCount;
        {   // the fused launches spin on each other's granules: all 256 blocks of a batch row must be resident together.  The occupancy
            // API can report one block per CU too many (MI355X guide, correctness boundaries), so one block per CU is taken off and a
            // quarter is kept as margin; a partitioned device (CPX: 32 CUs) falls back to the five-launch chain.
            int occ_a = 0, occ_b = 0, occ_c = 1 << 20;
            // (the fp32 policy's instantiations hold twice the weight registers: asked about separately)
            const hipError_t qa = e->bf16 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_a, qkv_attn_kernel<PRO_LN>, 256, 0)
                                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_a, qkv_attn_kernel<PRO_LN, float>, 256, 0);
            const hipError_t qb = e->bf16 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_b, oproj_fc1_kernel<4, true>, 256, 0)
                                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_b, oproj_fc1_kernel<4, true, float>, 256, 0);
            if (qa != hipSuccess || qb != hipSuccess) { (void)hipGetLastError(); occ_a = occ_b = 0; }
#ifdef MA_EXPERIMENTAL
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_c, layer_fused_kernel, 256, 0) != hipSuccess) { (void)hipGetLastError(); occ_c = 0; }
#endif
            const int occ_min = std::min(std::min(occ_a, occ_b), occ_c);
            // (a register-bound occupancy of two -- the fp32 instantiations at 171-210 VGPRs -- is exact: the over-report concerns the SGPR-limited
            // high-occupancy cases; 512 slots for the 256 blocks of batch 1 is the quarter of margin and more)
            const int usable = occ_min > 2 ? occ_min - 1 : occ_min;              // blocks per CU counted on
            e->resident_blocks = (long)e->n_cus * usable;
            e->chain_resident = e->resident_blocks * 4 >= 256L * 5;
            // the two-launch 8-row layer: 256 blocks of 8 waves at ~190-236 registers = one block per CU -- a register / wave-slot bound, where the
            // occupancy query is exact: every block must find a CU
            int occ_ra = 0, occ_rm = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_ra, rows_attn_kernel<true, 4, true, 3, 8, bf16_t>, 512, 0) != hipSuccess ||
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_rm, rows_mlp_kernel<bf16_t>, 512, 0) != hipSuccess) { (void)hipGetLastError(); occ_ra = occ_rm = 0; }
            e->rows_ok = (long)e->n_cus * std::min(occ_ra, occ_rm) >= 256;
#ifdef MA_EXPERIMENTAL
            // the rows-looped launches: the second one holds 66-130 KB of LDS, i.e. ONE block per CU -- an LDS bound, where the occupancy
            // query is exact (its off-by-one concerns the SGPR-limited high-occupancy cases): 256 blocks need 256 CUs
            int occ_r = 0;
            e->rf_ok = e->bf16 && e->hdt == MA_DTYPE_BF16 && rf_prepare() == hipSuccess &&
                       hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_r, oproj_fc1_rows_kernel<8>, 256, rf_oproj_lds(8)) == hipSuccess && (long)e->n_cus * occ_r >= 256;
            if (!e->rf_ok) (void)hipGetLastError();
#endif
        }
#ifdef MA_EXPERIMENTAL
        e->persist_shape = e->bf16 && e->hdt == MA_DTYPE_BF16 && c.hidden == PS_H && c.ffn == PS_F && c.heads == PS_HEADS && c.codebook_dim == PS_H && c.heads * ATTN_NCHUNK == PS_CUS &&
                           e->V >= PS_CUS * 32 && e->V <= PS_CUS * 33 && e->n_cus == PS_CUS && (size_t)prop.sharedMemPerBlockOptin >= PL_TOTAL;
        if (e->persist_shape && persist_prepare() != hipSuccess) { (void)hipGetLastError(); e->persist_shape = false; }
        if (e->persist_shape) {
            e->d_layers = e->dmalloc<DecLayerPtrs>(c.layers);
            e->d_gran = e->dmalloc<u64>(PG_TOTAL); e->d_serial = e->dmalloc<unsigned>(1); e->d_err = e->dmalloc<unsigned>(1);
            e->d_embtab = e->dmalloc<float>((size_t)c.codebook_size * H);
            e->d_ptrace = e->dmalloc<u64>((size_t)PS_CUS * (PS_TRACE_EVENTS + PS_TRACE2_EVENTS));
            HIP_CHECK(hipMemset(e->d_gran, 0, PG_TOTAL * sizeof(u64)));
            const unsigned one = 1u;
            HIP_CHECK(hipMemcpy(e->d_serial, &one, sizeof(unsigned), hipMemcpyHostToDevice));
            HIP_CHECK(hipMemset(e->d_err, 0, sizeof(unsigned)));
            HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&e->h_err), sizeof(unsigned)));
        }
#endif
    }
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&e->h_state), MB * sizeof(DecState)));
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&e->h_tokens), MB * e->maxnew * sizeof(long long)));
    // dense workspace: R = dense_rows samples stacked along the rows (R x 4096 point rows / R x 257 latent rows / R x 1057
    // detokenizer rows per pass)
    const int N = c.n_points, W = c.enc_width, T = e->T, Wt = c.tok_width, S = e->S, NL = c.num_latents;
    e->act_elem = e->bf16 ? 2 : 4;
    e->dense16 = e->bf16;
    e->enc_exact = !e->bf16 || c.enc_exact != 0;
    const size_t enc_elem = e->enc_exact ? 4 : 2;                    // element size of the buffers the encoder's activations live in
    e->dense_rows = std::min(c.max_batch, 64);                        // 64 x 4096 point rows per pass: 5 GB of workspace at the 350M shape (bf16 policy)
    e->prefill_rows = e->dense_rows;
    const size_t R = e->dense_rows;
    const size_t rows_seq = R * std::max(T, S);                      // rows of the latent / token streams
    const size_t wmax = std::max(W, Wt);
    const size_t fmax = std::max(4 * W, c.tok_ffn);
    auto amalloc = [&](size_t elems) -> void* { return e->dmalloc<char>(elems * std::max(e->act_elem, enc_elem)); };      // (buffers shared by the phases take the wider element)
    e->w_data = e->dmalloc<float>(R * N * W);
    e->w_lat = e->dmalloc<float>(R * T * W); e->w_lat2 = e->dmalloc<float>(R * NL * W);
    e->w_pf = e->dmalloc<float>(R * T * Wt); e->w_x = e->dmalloc<float>(R * S * Wt); e->w_y = e->dmalloc<float>(R * S * Wt);
    e->w_fe = e->dmalloc<float>(R * e->nf * Wt); e->w_logit = e->dmalloc<float>(R * e->nf * 9 * c.discrete_num);
    e->w_mask = e->dmalloc<unsigned char>(R * e->nf);
    e->a_feat = amalloc(R * N * 64); e->a_dataln = amalloc(R * N * W); e->a_kv = amalloc(R * N * 2 * W); e->a_q = amalloc((size_t)T * W);
    if (e->bf16) {
        size_t v = attn2_vt_elems(N, c.enc_heads, (int)R);
        v = std::max(v, attn2_vt_elems(T, c.enc_heads, (int)R)); v = std::max(v, attn2_vt_elems(T, c.heads, (int)R)); v = std::max(v, attn2_vt_elems(S, c.tok_heads, (int)R));
        e->vt_elems = v; e->a_vt = e->dmalloc<bf16_t>(v);
    }
    e->a_ln = amalloc(rows_seq * wmax); e->a_qkv = amalloc(rows_seq * 3 * wmax); e->a_att = amalloc(rows_seq * wmax); e->a_mlp = amalloc(rows_seq * fmax);
    e->a_cat = amalloc(R * NL * 2 * W); e->a_mean = amalloc(R * NL * c.embed_dim); e->a_fein = amalloc(R * e->nf * 3 * c.codebook_dim);
    e->a_x = amalloc(R * S * Wt);
    {
        const size_t PR = R * T;
        const size_t PS = std::min<size_t>(PR, 10240);           // (a split GEMM has fewer than 0.6 x CUs tiles of 256 x 256: at most ~40 tile rows)
        e->p_y_part_stride = (long)(PS * H);
        e->p_h = e->dmalloc<float>(PR * H); e->p_y = e->dmalloc<float>(std::max(PR, 4 * PS) * H);
        e->ln_gran_tiles = (PR / 256 + 1) * (size_t)((H + 255) / 256);
        e->d_ln_gran = e->dmalloc<u64>(2 * e->ln_gran_tiles * 256);
        HIP_CHECK(hipMemset(e->d_ln_gran, 0, 2 * e->ln_gran_tiles * 256 * sizeof(u64)));
        e->a_ph = amalloc(PR * H); e->a_pqkv = amalloc(PR * 3 * H); e->a_patt = amalloc(PR * H); e->a_pffn = amalloc(PR * c.ffn);
        if (e->bf16) {
            e->a_patt_tail = amalloc((size_t)64 * H);
            e->vt_tail_elems = attn2_vt_elems(T, c.heads, 1); e->a_vt_tail = e->dmalloc<bf16_t>(e->vt_tail_elems);
        }
    }
    const size_t B = c.max_batch;
    e->w_latents = e->dmalloc<float>(B * T * W); e->w_prefix = e->dmalloc<float>(B * T * H);
    e->w_tokens = e->dmalloc<long long>(B * e->maxnew); e->w_ids = e->dmalloc<long long>(B * (size_t)e->nf * 9);
    // per-layer decode pointers
    e->dl.resize(c.layers);
    for (int l = 0; l < c.layers; ++l) {
        const std::string p = DEC + "layers." + std::to_string(l) + ".";
        DecLayerPtrs& w = e->dl[l];
        w.qkv_w = e->P(p + "qkv.weight"); w.qkv_b = e->PF(p + "qkv.bias");
        w.o_w = e->P(p + "self_attn.out_proj.weight"); w.o_b = e->PF(p + "self_attn.out_proj.bias");
        w.fc1_w = e->P(p + "fc1.weight"); w.fc1_b = e->PF(p + "fc1.bias");
        w.fc2_w = e->P(p + "fc2.weight"); w.fc2_b = e->PF(p + "fc2.bias");
        w.ln1_g = e->PF(p + "self_attn_layer_norm.weight"); w.ln1_b = e->PF(p + "self_attn_layer_norm.bias");
        w.ln2_g = e->PF(p + "final_layer_norm.weight"); w.ln2_b = e->PF(p + "final_layer_norm.bias");
    }
#ifdef MA_EXPERIMENTAL
    if (e->persist_shape) HIP_CHECK(hipMemcpy(e->d_layers, e->dl.data(), c.layers * sizeof(DecLayerPtrs), hipMemcpyHostToDevice));
#endif
}

template <typename F>
int guarded(ma_engine* e, F f) {
    try {
        if (e) { hipError_t r = hipSetDevice(e->device); if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(r)); }
        f();
        return MA_OK;
    } catch (const MaError& x) {
        if (e) e->err = x.msg; else g_create_error = x.msg;
        return x.code;
    } catch (const std::exception& x) {
        if (e) e->err = x.what(); else g_create_error = x.what();
        return MA_ERR_INVALID;
    } catch (...) {
        if (e) e->err = "unknown exception"; else g_create_error = "unknown exception";
        return MA_ERR_INVALID;
    }
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

// MA_SRC_HASH: SHA-256 over the sources and flags this library was compiled from (meshanything_amd/build.py passes it); the loader
// compares it with the tree next to the library, so a stale .so cannot travel to a GPU box unnoticed -- and needs no side file
#ifndef MA_SRC_HASH
#define MA_SRC_HASH "unknown"
#endif
const char* ma_version(void) { return "meshanything_amd 0.1 (gfx950) src=" MA_SRC_HASH; }

const char* ma_last_error(const ma_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int ma_engine_create(ma_engine** out, const ma_config* cfg, int device) {
    if (!out || !cfg) { g_create_error = "null argument"; return MA_ERR_INVALID; }
    *out = nullptr;
    ma_engine* e = nullptr;
    int rc = guarded(nullptr, [&] {
        validate_config(*cfg);
        int ndev = 0;
        HIP_CHECK(hipGetDeviceCount(&ndev));
        if (device < 0 || device >= ndev) throw MaError(MA_ERR_INVALID, "device index out of range (" + std::to_string(ndev) + " visible)");
        HIP_CHECK(hipSetDevice(device));
        e = new ma_engine();
        e->cfg = *cfg; e->device = device;
        build_engine(e);
    });
    if (rc != MA_OK) { if (e) ma_engine_destroy(e); return rc; }
    *out = e;
    return MA_OK;
}

void ma_engine_destroy(ma_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    drop_graphs(e);
    if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
    if (e->tail_stream) (void)hipStreamDestroy(e->tail_stream);
    if (e->tail_stream_low) (void)hipStreamDestroy(e->tail_stream_low);
    if (e->tail_fork) (void)hipEventDestroy(e->tail_fork);
    if (e->tail_join) (void)hipEventDestroy(e->tail_join);
    for (hipEvent_t ev : e->tail_kv) (void)hipEventDestroy(ev);
    for (hipStream_t st : e->grp_stream) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : e->grp_done) (void)hipEventDestroy(ev);
    if (e->grp_fork) (void)hipEventDestroy(e->grp_fork);
    for (void* p : e->allocs) (void)hipFree(p);
    if (e->arena) (void)hipFree(e->arena);
    if (e->stage) (void)hipFree(e->stage);
    if (e->kv) (void)hipFree(e->kv);
    if (e->h_state) (void)hipHostFree(e->h_state);
    if (e->h_err) (void)hipHostFree(e->h_err);
    if (e->h_chain_err) (void)hipHostFree(e->h_chain_err);
    if (e->h_tokens) (void)hipHostFree(e->h_tokens);
    delete e;
}

int ma_engine_set_option(ma_engine* e, const char* name, int64_t value) {
    if (!e || !name) return MA_ERR_INVALID;
    return guarded(e, [&] {
        const std::string n = name;
        if (n == "gemm_impl") e->opt_gemm_impl = (int)value;
        else if (n == "prefill_stepwise") e->opt_prefill_stepwise = (int)value;
        else if (n == "use_graph") e->cfg.use_graph = (int)value;
        else if (n == "profile_batch") e->opt_profile_batch = (int)value;
        else if (n == "mfma_min_batch") { e->opt_mfma_min_batch = (int)value; drop_graphs(e); }
        else if (n == "attn_final_min_batch") { e->opt_attn_final_min_batch = (int)value; drop_graphs(e); }
        else if (n == "attn_rowwave") { e->opt_attn_rowwave = (int)value; drop_graphs(e); }
        else if (n == "mfma_fold_ln") { e->opt_mfma_fold_ln = (int)value; drop_graphs(e); }
        else if (n == "mfma_fc2_ksplit") {
            if (value != 0 && value != 1 && value != 2 && value != 4) throw MaError(MA_ERR_INVALID, "mfma_fc2_ksplit must be 0 (default), 1, 2 or 4");
            if (value > 1 && e->cfg.ffn % (4 * (int)value * 32) != 0) throw MaError(MA_ERR_INVALID, "mfma_fc2_ksplit does not divide the ffn width");
            e->opt_mfma_fc2_ksplit = (int)value; drop_graphs(e);
        }
        else if (n == "mfma_chunks") {
            if (value != 4 && value != 8) throw MaError(MA_ERR_INVALID, "mfma_chunks must be 4 or 8");
            gemm_dec_chunks() = (int)value; drop_graphs(e);
        }
        else if (n == "mfma_ln_waves") {
            if (value != 0 && value != 4 && value != 8) throw MaError(MA_ERR_INVALID, "mfma_ln_waves must be 0 (by batch), 4 or 8");
            e->opt_mfma_ln_waves = (int)value; drop_graphs(e);
        }
        else if (n == "attn_pair") { e->opt_attn_pair = (int)value; drop_graphs(e); }
        else if (n == "fuse_rows_attn") { e->opt_fuse_rows_attn = value ? 1 : 0; drop_graphs(e); }
        else if (n == "fuse_rows_mlp") { e->opt_fuse_rows_mlp = value ? 1 : 0; drop_graphs(e); }
        else if (n == "rows_attn_early") {
            if (value < 0 || value > 6) throw MaError(MA_ERR_INVALID, "rows_attn_early: 0 .. 6");
#ifndef MA_EXPERIMENTAL
            if (value != 3 && value != 5 && value != 6) throw MaError(MA_ERR_STATE, "rows_attn_early: placements 0, 1, 2 and 4 need a library built with MA_EXPERIMENTAL=1 (measured, not kept)");
#endif
            e->opt_rows_attn_early = (int)value; drop_graphs(e);
        }
        else if (n == "rows_mlp_ln2") { e->opt_rows_mlp_ln2 = value ? 1 : 0; drop_graphs(e); }
        else if (n == "rows_mlp_prefetch") { if (value < 0 || value > 9) throw MaError(MA_ERR_INVALID, "rows_mlp_prefetch: 0 off, 1 / 2 rounds, 8 weights only, 9 half a round"); e->opt_rows_mlp_prefetch = (int)value; drop_graphs(e); }
        else if (n == "decode_groups") { if (value < 1 || value > 16) throw MaError(MA_ERR_INVALID, "decode_groups: 1 .. 16"); e->opt_decode_groups = (int)value; drop_graphs(e); }      // (the captured steps embed the fused 8-row launches, which rows_gate refuses beside a second row group)
        else if (n == "mfma_fold_fc1_max") { e->opt_mfma_fold_fc1_max = (int)value; drop_graphs(e); }
        else if (n == "mfma_fold_qkv_max") { e->opt_mfma_fold_qkv_max = (int)value; drop_graphs(e); }
        else if (n == "oproj_fc1_sweep_waves") { e->opt_oproj_fc1_sweep_waves = (int)value; drop_graphs(e); }
        else if (n == "fuse_fc2") { e->opt_fuse_fc2 = (int)value; drop_graphs(e); }
        else if (n == "qkv_xcd_local") { e->opt_qkv_xcd_local = value ? 1 : 0; drop_graphs(e); }
#ifndef MA_EXPERIMENTAL
        else if ((n == "rows_fused" || n == "fuse_layer" || n == "decode_impl") && value != 0)
            throw MaError(MA_ERR_STATE, n + " needs a library built with MA_EXPERIMENTAL=1 (the rejected decode-step forms are not part of the product build)");
#endif
        else if (n == "rows_fused") { e->opt_rows_fused = value ? 1 : 0; drop_graphs(e); }
        else if (n == "rows_fused_min") { e->opt_rows_fused_min = (int)value; drop_graphs(e); }
        else if (n == "fuse_layer") { e->opt_fuse_layer = (int)value; drop_graphs(e); }
        else if (n == "attn_final_waves") { if (value != 0 && value != 4 && value != 8 && value != 16) throw MaError(MA_ERR_INVALID, "attn_final_waves: 0, 4, 8 or 16"); e->opt_attn_final_waves = (int)value; drop_graphs(e); }
        else if (n == "gemm_xcd_swizzle") e->opt_gemm_xcd_swizzle = (int)value;
        else if (n == "qkv_to_cache") e->opt_qkv_to_cache = value ? 1 : 0;
        else if (n == "prefill_tail") { if (value < 0 || value > 2) throw MaError(MA_ERR_INVALID, "prefill_tail: 0 (one stream), 1 (the last rows as a chain on a second stream), 2 (... of the lowest priority)"); e->opt_prefill_tail = (int)value; }
        else if (n == "gemm_splitk") { if (value < 0 || value > 2) throw MaError(MA_ERR_INVALID, "gemm_splitk: 0 (never), 1 (fc2 of small prefills), 2 (+ out_proj)"); e->opt_gemm_splitk = (int)value; }
        else if (n == "fuse_ln") {
#ifndef MA_EXPERIMENTAL
            if (value) throw MaError(MA_ERR_STATE, "fuse_ln needs a library built with MA_EXPERIMENTAL=1 (LayerNorm inside the GEMM epilogue: measured, not kept)");
#endif
            e->opt_fuse_ln = value ? 1 : 0;
        }
        else if (n == "gemm256") { if (value < 0 || value > 2) throw MaError(MA_ERR_INVALID, "gemm256: 0 (128-row tiles), 1 (one tile per workgroup) or 2 (1 + the persistent form)"); gemm256_enabled() = (int)value; }
        else if (n == "attn_impl") { if (value != 1 && value != 2) throw MaError(MA_ERR_INVALID, "attn_impl: 1 (attn.hpp) or 2 (attn2.hpp)"); e->opt_attn_impl = (int)value; }
        else if (n == "gemm_variant") {
#ifndef MA_EXPERIMENTAL
            if (value != 6) throw MaError(MA_ERR_STATE, "gemm_variant: the A/B tile variants need a library built with MA_EXPERIMENTAL=1");
#endif
            gemm_tile_variant() = (int)value;
        }
        else if (n == "gemv_small_rows") {
            if (value != 0 && value != 1 && value != 2 && value != 4) throw MaError(MA_ERR_INVALID, "gemv_small_rows must be 0, 1, 2 or 4");
            gemv_small_rows() = (int)value; e->embtab_ready = false; drop_graphs(e);
        } else if (n == "gemv_k8_ksplit") {
            if (value != 1 && value != 2 && value != 4) throw MaError(MA_ERR_INVALID, "gemv_k8_ksplit must be 1, 2 or 4");
            gemv_k8_ksplit() = (int)value; drop_graphs(e);
        } else if (n == "fuse_qkv_attn") { e->opt_fuse_qkv_attn = value ? 1 : 0; drop_graphs(e); }
        else if (n == "fuse_oproj_fc1") { e->opt_fuse_oproj_fc1 = value ? 1 : 0; drop_graphs(e); }
        else if (n == "chain_resident") {            // 0: never spin on other blocks (five-launch chain); 1: re-arm after a fallback, if the device allows
            e->chain_resident = value != 0 && e->resident_blocks * 4 >= 256L * 5;
            drop_graphs(e);
        }
        else if (n == "decode_impl") {
            if (value != 0 && value != 1) throw MaError(MA_ERR_INVALID, "decode_impl must be 0 (launch chain) or 1 (persistent step)");
            e->opt_decode_impl = (int)value;
        }
        else if (n == "gemv_rpw") {
            if (value != 1 && value != 2 && value != 4) throw MaError(MA_ERR_INVALID, "gemv_rpw must be 1, 2 or 4");
            gemv_rpw_big() = (int)value;
            e->n_parts = e->bf16 ? gemv_num_blocks<bf16_t>(e->V, e->cfg.hidden) : gemv_num_blocks<float>(e->V, e->cfg.hidden);
            drop_graphs(e);                              // the captured steps embed grids and arguments: the next generate() re-captures
        } else throw MaError(MA_ERR_INVALID, "unknown option " + n);
    });
}

int ma_engine_get_option(ma_engine* e, const char* name, int64_t* value) {
    if (!e || !name || !value) return MA_ERR_INVALID;
    return guarded(e, [&] {
        const std::string n = name;
        if (n == "fuse_qkv_attn") *value = fuse_qkv_attn(e) ? 1 : 0;                 // effective values: option AND eligibility
        else if (n == "fuse_oproj_fc1") *value = fuse_oproj_fc1(e) ? 1 : 0;
        else if (n == "experimental") {
#ifdef MA_EXPERIMENTAL
            *value = 1;
#else
            *value = 0;
#endif
        }
        else if (n == "decode_impl") *value = e->opt_decode_impl;
        else if (n == "persist_available") *value = e->persist_shape ? 1 : 0;
        else if (n == "chain_resident") *value = e->chain_resident ? 1 : 0;
        else if (n == "chain_fallbacks") *value = e->chain_fallbacks;
        else if (n == "xchg_last_code") *value = e->xchg_last_code;
        else if (n == "xchg_timeouts" || n == "slow_blocks" || n == "slow_block_max_us" || n == "scalar_sweep_rescues" || n == "xchg_first_giveup_code" ||
                 n == "xchg_first_giveup_block" || n == "xchg_first_giveup_polls" || n == "xchg_descheduled") {
            // device counters of the fused launches, never cleared: sweeps that ever gave up | blocks that lived > 1 ms | the longest of them |
            // scalar sweeps that a vector look had to finish | the first sweep that ever gave up: its error bit, blockIdx.x | y << 8 | z << 16 | wave << 24, its polls |
            // sweeps that found 20 ms gone on the clock after a handful of polls (the wave was off the device: common.hpp xchg_expired) and went on
            unsigned v[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            HIP_CHECK(hipDeviceSynchronize());
            HIP_CHECK(hipMemcpy(v, e->d_chain_err, sizeof(v), hipMemcpyDeviceToHost));
            *value = n == "xchg_timeouts" ? v[1] : n == "slow_blocks" ? v[3] : n == "scalar_sweep_rescues" ? v[4] : n == "xchg_first_giveup_code" ? (v[5] & 0x7fffffffu)
                   : n == "xchg_first_giveup_block" ? v[6] : n == "xchg_first_giveup_polls" ? v[7] : n == "xchg_descheduled" ? v[8] : v[2] / 100;
        }
        else if (n == "resident_blocks") *value = e->resident_blocks;
        else if (n == "use_graph") *value = e->cfg.use_graph;
        else if (n == "dense_rows") *value = e->dense_rows;
        else if (n == "mfma_min_batch") *value = e->opt_mfma_min_batch;
        else if (n == "attn_final_min_batch") *value = e->opt_attn_final_min_batch;
        else if (n == "attn_rowwave") *value = e->opt_attn_rowwave;
        else if (n == "mfma_fold_ln") *value = e->opt_mfma_fold_ln;
        else if (n == "mfma_ln_waves") *value = e->opt_mfma_ln_waves;
        else if (n == "mfma_chunks") *value = gemm_dec_chunks();
        else if (n == "mfma_fc2_ksplit") *value = e->opt_mfma_fc2_ksplit;
        else if (n == "attn_pair") *value = e->opt_attn_pair;
        else if (n == "fuse_rows_attn" || n == "fuse_rows_mlp") {      // as the engine will apply it at 8 rows: the very gates of enqueue_layers_mfma
            const RowsGates rg = rows_gates(e, RA_ROWS, -1);
            // (the first half also needs its LayerNorm 2 folded in, or left by the previous layer's second half)
            *value = n == "fuse_rows_mlp" ? rg.mlp : (rg.attn && (rg.fold || (rg.mlp && e->opt_rows_mlp_ln2)));
        }
        else if (n == "rows_attn_early") *value = e->opt_rows_attn_early;
        else if (n == "rows_mlp_ln2") *value = e->opt_rows_mlp_ln2;
        else if (n == "rows_mlp_prefetch") *value = e->opt_rows_mlp_prefetch;
        else if (n == "decode_groups") *value = decode_group_count(e, std::max(1, std::min(e->opt_profile_batch, e->cfg.max_batch)), 0);   // effective, for profile_batch rows
        else if (n == "mfma_fold_fc1_max") *value = e->opt_mfma_fold_fc1_max;
        else if (n == "mfma_fold_qkv_max") *value = e->opt_mfma_fold_qkv_max;
        else if (n == "oproj_fc1_sweep_waves") *value = e->opt_oproj_fc1_sweep_waves;
        else if (n == "fuse_fc2") *value = e->opt_fuse_fc2;
        else if (n == "qkv_xcd_local") *value = e->opt_qkv_xcd_local;
        else if (n == "rows_fused") *value = e->opt_rows_fused && e->rf_ok && e->chain_resident ? 1 : 0;
        else if (n == "rows_fused_min") *value = e->opt_rows_fused_min;
        else if (n == "fuse_layer") *value = fuse_layer(e) ? 1 : 0;
        else if (n == "attn_final_waves") *value = e->opt_attn_final_waves;
        else if (n == "gemm_xcd_swizzle") *value = e->opt_gemm_xcd_swizzle;
        else if (n == "qkv_to_cache") *value = e->opt_qkv_to_cache;
        else if (n == "gemm_splitk") *value = e->opt_gemm_splitk;
        else if (n == "prefill_tail") *value = e->opt_prefill_tail;
        else if (n == "fuse_ln") *value = e->opt_fuse_ln && e->chain_resident;      // (effective)
        else if (n == "attn_impl") *value = e->opt_attn_impl;
        else if (n == "gemm_variant") *value = gemm_tile_variant();
        else if (n == "gemm256") *value = gemm256_enabled();
        else throw MaError(MA_ERR_INVALID, "unknown option " + n);
    });
}

// Large tensors are uploaded in their STORED dtype and converted into their arena slot on the device (the host-side loop of
// pack_tensor converts ~150 M elements per second on one core: 4.4 s for the 350M checkpoint, against 0.4 s this way; same f2bf
// rounding, same bytes -- tests/test_gpu_model_api.py compares the two arenas).  Returns false when the tensor is not eligible:
// pack_tensor then handles it, including every error message.
static bool load_tensor_on_device(ma_engine* e, const ma_tensor_desc& t) {
    if (!t.name || !t.data || t.ndim < 1 || t.ndim > 3 || t.dtype < MA_DTYPE_F32 || t.dtype > MA_DTYPE_F16) return false;
    bool dropped;
    const Source* s = find_source(e->L, t.name, &dropped);
    if (!s || dropped) return false;
    long long lead = 1;
    for (int i = 0; i + 1 < t.ndim; ++i) lead *= t.shape[i];
    if (lead != s->src_rows || t.shape[t.ndim - 1] != s->src_cols) return false;
    const size_t elems = (size_t)s->take_rows * s->take_cols;
    if (elems < (1u << 16)) return false;
    const Entry& en = e->L.entries[s->entry];
    const size_t src_esz = t.dtype == MA_DTYPE_F32 ? 4 : 2, src_bytes = (size_t)s->take_rows * s->src_cols * src_esz;
    if (e->stage_bytes < src_bytes) {
        if (e->stage) (void)hipFree(e->stage);
        e->stage = nullptr; e->stage_bytes = 0;
        HIP_CHECK(hipMalloc(&e->stage, src_bytes));
        e->stage_bytes = src_bytes;
    }
    // the previous tensor's conversion kernel reads this staging buffer: wait for it explicitly (the implicit ordering of a pageable
    // copy behind kernels holds on the legacy null stream only, not under -fgpu-default-stream=per-thread)
    HIP_CHECK(hipStreamSynchronize(nullptr));
    HIP_CHECK(hipMemcpy(e->stage, t.data, src_bytes, hipMemcpyHostToDevice));
    const int esz = en.dtype == MA_DTYPE_F32 ? 4 : 2;
    void* dst = e->arena + en.offset + s->dst_elem * esz;
    const int blocks = (int)std::min<size_t>((elems + 255) / 256, 65535u * 16u);
    hipLaunchKernelGGL(cvt_weight_kernel, dim3(blocks), dim3(256), 0, nullptr, e->stage, t.dtype, s->src_cols, dst, en.dtype, en.cols, s->take_rows, s->take_cols);
    HIP_CHECK(hipGetLastError());
    e->ps.filled[s->entry] += elems;
    return true;
}

int ma_engine_load_weights(ma_engine* e, const ma_tensor_desc* tensors, int n) {
    if (!e || (!tensors && n > 0)) return MA_ERR_INVALID;
    return guarded(e, [&] {
        for (int i = 0; i < n; ++i) {
            if (load_tensor_on_device(e, tensors[i])) continue;
            std::string err;
            int rc = pack_tensor(e->L, e->ps, tensors[i], err, [&](size_t off, const void* p, size_t nb) {
                HIP_CHECK(hipMemcpy(e->arena + off, p, nb, hipMemcpyHostToDevice));
            });
            if (rc != MA_OK) throw MaError(rc, err);
        }
    });
}

int ma_engine_finalize_weights(ma_engine* e) {
    if (!e) return MA_ERR_INVALID;
    return guarded(e, [&] {
        std::string missing;
        if (!pack_complete(e->L, e->ps, missing)) throw MaError(MA_ERR_MISSING, "checkpoint incomplete, missing: " + missing);
        HIP_CHECK(hipDeviceSynchronize());                   // the device-side conversions of the last tensors
        if (e->stage) { (void)hipFree(e->stage); e->stage = nullptr; e->stage_bytes = 0; }
        e->weights_ready = true; e->embtab_ready = false;
    });
}

int ma_engine_arena(ma_engine* e, void** dev_ptr, size_t* bytes) {
    if (!e || !dev_ptr || !bytes) return MA_ERR_INVALID;
    *dev_ptr = e->arena; *bytes = e->L.bytes;
    return MA_OK;
}

int ma_engine_mark_weights_loaded(ma_engine* e) {
    if (!e) return MA_ERR_INVALID;
    e->weights_ready = true; e->embtab_ready = false;
    return MA_OK;
}

int ma_engine_broadcast_weights(ma_engine* e, void* nccl_comm, int root, void* stream) {
    if (!e || !nccl_comm) return MA_ERR_INVALID;
    return guarded(e, [&] {
        // RCCL is resolved lazily so that the library loads on hosts without librccl (CPU-only checks)
        typedef int (*bcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
        static bcast_fn fn = nullptr;
        if (!fn) {
            void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) throw MaError(MA_ERR_NCCL, std::string("cannot load librccl.so: ") + dlerror());
            fn = reinterpret_cast<bcast_fn>(dlsym(h, "ncclBroadcast"));
            if (!fn) throw MaError(MA_ERR_NCCL, "ncclBroadcast not found in librccl.so");
        }
        const int rc = fn(e->arena, e->arena, e->L.bytes, /*ncclInt8*/ 0, root, nccl_comm, reinterpret_cast<hipStream_t>(stream));
        if (rc != 0) throw MaError(MA_ERR_NCCL, "ncclBroadcast failed with code " + std::to_string(rc));
        e->weights_ready = true; e->embtab_ready = false;
    });
}

// ---- host-only arena description / packing ------------------------------------------------------------------------
static int layout_for(const ma_config* cfg, Layout& L, std::string& err) {
    try { validate_config(*cfg); L = build_layout(*cfg); return MA_OK; }
    catch (const MaError& x) { err = x.msg; return x.code; }
    catch (const std::exception& x) { err = x.what(); return MA_ERR_INVALID; }
}

int64_t ma_arena_bytes(const ma_config* cfg) {
    if (!cfg) return MA_ERR_INVALID;
    Layout L; std::string err;
    int rc = layout_for(cfg, L, err);
    if (rc != MA_OK) { g_create_error = err; return rc; }
    return (int64_t)L.bytes;
}

int ma_arena_num_entries(const ma_config* cfg) {
    if (!cfg) return MA_ERR_INVALID;
    Layout L; std::string err;
    int rc = layout_for(cfg, L, err);
    if (rc != MA_OK) { g_create_error = err; return rc; }
    return (int)L.entries.size();
}

int ma_arena_entry(const ma_config* cfg, int i, char* name, int name_cap, int64_t* offset, int64_t* bytes, int32_t* dtype, int32_t* rows, int32_t* cols) {
    if (!cfg) return MA_ERR_INVALID;
    Layout L; std::string err;
    int rc = layout_for(cfg, L, err);
    if (rc != MA_OK) { g_create_error = err; return rc; }
    if (i < 0 || i >= (int)L.entries.size()) return MA_ERR_INVALID;
    const Entry& en = L.entries[i];
    if (name && name_cap > 0) { snprintf(name, name_cap, "%s", en.name.c_str()); }
    if (offset) *offset = (int64_t)en.offset;
    if (bytes) *bytes = (int64_t)en.bytes;
    if (dtype) *dtype = en.dtype;
    if (rows) *rows = en.rows;
    if (cols) *cols = en.cols;
    return MA_OK;
}

int ma_pack_weights_host(const ma_config* cfg, const ma_tensor_desc* tensors, int n, void* host_arena, char* err, int err_cap) {
    auto fail = [&](int code, const std::string& m) { if (err && err_cap > 0) snprintf(err, err_cap, "%s", m.c_str()); return code; };
    if (!cfg || !host_arena || (!tensors && n > 0)) return fail(MA_ERR_INVALID, "null argument");
    Layout L; std::string e;
    int rc = layout_for(cfg, L, e);
    if (rc != MA_OK) return fail(rc, e);
    PackState ps; pack_state_init(L, ps);
    std::memset(host_arena, 0, L.bytes);
    for (int i = 0; i < n; ++i) {
        rc = pack_tensor(L, ps, tensors[i], e, [&](size_t off, const void* p, size_t nb) { std::memcpy(reinterpret_cast<char*>(host_arena) + off, p, nb); });
        if (rc != MA_OK) return fail(rc, e);
    }
    std::string missing;
    if (!pack_complete(L, ps, missing)) return fail(MA_ERR_MISSING, "checkpoint incomplete, missing: " + missing);
    return MA_OK;
}

int ma_engine_upload_arena(ma_engine* e, const void* host_arena, size_t bytes) {
    if (!e || !host_arena) return MA_ERR_INVALID;
    return guarded(e, [&] {
        if (bytes != e->L.bytes) throw MaError(MA_ERR_SHAPE, "arena size mismatch");
        HIP_CHECK(hipMemcpy(e->arena, host_arena, bytes, hipMemcpyHostToDevice));
        e->weights_ready = true; e->embtab_ready = false;
    });
}

// ---- hot path ------------------------------------------------------------------------------------------------------
int ma_encode(ma_engine* e, const void* pc, int pc_dtype, int B, float* latents, float* prefix, void* stream) {
    if (!e || !pc || !latents) return MA_ERR_INVALID;
    return guarded(e, [&] {
        require_ready(e); check_batch(e, B);
        if (pc_dtype != MA_DTYPE_F32 && pc_dtype != MA_DTYPE_F16) throw MaError(MA_ERR_INVALID, "pc_dtype must be F32 or F16");
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        RoctxRange range("ma_encode");
        DenseScope enc(e, e->bf16 && !e->enc_exact);
        const size_t pstride = (size_t)e->cfg.n_points * 6 * (pc_dtype == MA_DTYPE_F16 ? 2 : 4);
        for (int b0 = 0; b0 < B; b0 += e->dense_rows) {          // the whole chunk goes through every GEMM at once (M = nb x rows)
            const int nb = std::min(e->dense_rows, B - b0);
            float* lat = latents + (size_t)b0 * e->T * e->cfg.enc_width;
            encode_chunk(e, s, reinterpret_cast<const char*>(pc) + b0 * pstride, pc_dtype, nb, lat);
            if (prefix) prefix_chunk(e, s, lat, prefix + (size_t)b0 * e->T * e->cfg.hidden, nb);
        }
    });
}

int ma_to_shape_latents(ma_engine* e, const float* latents, int B, float* out, void* stream) {
    if (!e || !latents || !out) return MA_ERR_INVALID;
    return guarded(e, [&] {
        require_ready(e); check_batch(e, B);
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        DenseScope enc(e, e->bf16 && !e->enc_exact);
        const size_t n = (size_t)e->cfg.num_latents * e->cfg.enc_width;
        for (int b0 = 0; b0 < B; b0 += e->dense_rows) {
            const int nb = std::min(e->dense_rows, B - b0);
            shape_latents_chunk(e, s, latents + b0 * n, e->cfg.enc_width, RowMap{0, 0, 0}, nb);
            HIP_CHECK(hipMemcpyAsync(out + b0 * n, e->w_lat2, (size_t)nb * n * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
    });
}

int ma_process_point_feature(ma_engine* e, const float* point_feature, int B, float* prefix, void* stream) {
    if (!e || !point_feature || !prefix) return MA_ERR_INVALID;
    return guarded(e, [&] {
        require_ready(e); check_batch(e, B);
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        DenseScope enc(e, e->bf16 && !e->enc_exact);
        for (int b0 = 0; b0 < B; b0 += e->dense_rows) {
            const int nb = std::min(e->dense_rows, B - b0);
            prefix_chunk(e, s, point_feature + (size_t)b0 * e->T * e->cfg.enc_width, prefix + (size_t)b0 * e->T * e->cfg.hidden, nb);
        }
    });
}

int ma_get_codes(ma_engine* e, const int64_t* ids, int B, float* codes, void* stream) {
    if (!e || !ids || !codes) return MA_ERR_INVALID;
    return guarded(e, [&] {
        require_ready(e); check_batch(e, B);
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        const int D = e->cfg.codebook_dim, nf = e->nf;
        for (int b0 = 0; b0 < B; b0 += e->dense_rows) {
            const int nb = std::min(e->dense_rows, B - b0);
            hipLaunchKernelGGL((codes_gather2_kernel<float>), dim3(ceil_div(nb * nf * 3 * (D / 4), 256)), dim3(256), 0, s, reinterpret_cast<const long long*>(ids) + (size_t)b0 * nf * 9,
                               e->PF(DEC + "quantize_codebooks"), D, nb * nf, codes + (size_t)b0 * nf * 3 * D, (float*)nullptr, e->w_mask);
            HIP_CHECK(hipGetLastError());
        }
    });
}

int ma_generate(ma_engine* e, const float* prefix, int B, const ma_sample_cfg* sc_in, int64_t* tokens, int32_t* lengths, int32_t* n_generated, void* stream) {
    if (!e || !prefix || !tokens) return MA_ERR_INVALID;
    return guarded(e, [&] {
        require_ready(e); check_batch(e, B);
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        const ma_sample_cfg sc = resolve_sample_cfg(e, sc_in);
        // rows are independent (no cross-batch op anywhere in meshanything.py:134-176) but step together: one weight stream per step
        const int nmax = generate_batch(e, s, prefix, B, sc, reinterpret_cast<long long*>(tokens), lengths);
        HIP_CHECK(hipStreamSynchronize(s));
        if (n_generated) *n_generated = nmax;
    });
}

int ma_postprocess_tokens(ma_engine* e, const int64_t* tokens, int ld_tokens, int B, int n_generated, int64_t* ids, void* stream) {
    if (!e || !tokens || !ids) return MA_ERR_INVALID;
    return guarded(e, [&] {
        check_batch(e, B);
        if (n_generated < 0 || n_generated > e->maxnew) throw MaError(MA_ERR_INVALID, "n_generated out of range [0, 9*n_max_faces+2]");
        if (ld_tokens < n_generated) throw MaError(MA_ERR_INVALID, "ld_tokens smaller than n_generated");
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        const int total = B * (e->maxnew - 2);
        hipLaunchKernelGGL(postprocess_tokens_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, reinterpret_cast<const long long*>(tokens), ld_tokens,
                           n_generated, e->maxnew, reinterpret_cast<long long*>(ids), B);
        HIP_CHECK(hipGetLastError());
    });
}

int ma_detokenize_embeds(ma_engine* e, const int64_t* ids, const float* codes, const float* latents, int B, float* coords, void* stream) {
    if (!e || !ids || !latents || !coords) return MA_ERR_INVALID;
    return guarded(e, [&] {
        require_ready(e); check_batch(e, B);
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        RoctxRange range("ma_detokenize");
        const size_t nf = e->nf;
        for (int b0 = 0; b0 < B; b0 += e->dense_rows) {
            const int nb = std::min(e->dense_rows, B - b0);
            detok_chunk(e, s, reinterpret_cast<const long long*>(ids) + (size_t)b0 * nf * 9, codes ? codes + (size_t)b0 * nf * 3 * e->cfg.codebook_dim : nullptr,
                        latents + (size_t)b0 * e->T * e->cfg.enc_width, coords + (size_t)b0 * nf * 9, nb);
        }
    });
}

int ma_detokenize(ma_engine* e, const int64_t* ids, const float* latents, int B, float* coords, void* stream) {
    return ma_detokenize_embeds(e, ids, nullptr, latents, B, coords, stream);
}

int ma_forward(ma_engine* e, const void* pc, int pc_dtype, int B, const ma_sample_cfg* sc, float* coords, int64_t* tokens, int32_t* lengths,
               int32_t* n_generated, int64_t* ids, float* latents, void* stream) {
    if (!e || !pc || !coords) return MA_ERR_INVALID;
    int rc = guarded(e, [&] { require_ready(e); check_batch(e, B); });
    if (rc != MA_OK) return rc;
    float* lat = latents ? latents : e->w_latents;
    int64_t* tok = tokens ? tokens : reinterpret_cast<int64_t*>(e->w_tokens);
    int64_t* idp = ids ? ids : reinterpret_cast<int64_t*>(e->w_ids);
    int32_t ngen = 0;
    if ((rc = ma_encode(e, pc, pc_dtype, B, lat, e->w_prefix, stream)) != MA_OK) return rc;
    if ((rc = ma_generate(e, e->w_prefix, B, sc, tok, lengths, &ngen, stream)) != MA_OK) return rc;
    if (n_generated) *n_generated = ngen;
    if ((rc = ma_postprocess_tokens(e, tok, e->maxnew, B, ngen, idp, stream)) != MA_OK) return rc;
    if ((rc = ma_detokenize(e, idp, lat, B, coords, stream)) != MA_OK) return rc;
    return guarded(e, [&] { HIP_CHECK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream))); });
}

// ---- kernel-level entry points -------------------------------------------------------------------------------------
int ma_op_gemv(int wdtype, const void* W, const float* bias, const float* x, const float* ln_g, const float* ln_b, float ln_eps, const float* res,
               float* y, float* xn_out, int N, int K, int act, void* stream) {
    return guarded(nullptr, [&] {
        if (!W || !x || !y || N < 1 || K < 8 || K % 8) throw MaError(MA_ERR_INVALID, "ma_op_gemv: bad arguments");
        GemvArgs a{};
        a.W = W; a.bias = bias; a.x = x; a.ln_g = ln_g; a.ln_b = ln_b; a.ln_eps = ln_eps; a.xn_out = xn_out; a.res = res; a.y = y; a.N = N; a.K = K;
        a.act = act; a.round_x = wdtype == MA_DTYPE_BF16; a.epi = EPI_PLAIN;
        a.round_x = wdtype != MA_DTYPE_F32;
        if (wdtype == MA_DTYPE_BF16) gemv_launch<bf16_t>(a, reinterpret_cast<hipStream_t>(stream));
        else if (wdtype == MA_DTYPE_F16) gemv_launch<f16_t>(a, reinterpret_cast<hipStream_t>(stream));
        else if (wdtype == MA_DTYPE_F32) gemv_launch<float>(a, reinterpret_cast<hipStream_t>(stream));
        else throw MaError(MA_ERR_INVALID, "ma_op_gemv: wdtype");
    });
}

int ma_op_gemm(int wdtype, int impl, const float* A, int lda, const void* W, const float* bias, const float* R, int ldr, float* C, int ldc, int M, int N,
               int K, int act, void* stream) {
    return guarded(nullptr, [&] {
        if (!A || !W || !C) throw MaError(MA_ERR_INVALID, "ma_op_gemm: null pointer");
        GemmArgs g{A, lda, W, bias, R, ldr, C, ldc, M, N, K, act};
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        hipError_t r;
        if (wdtype == MA_DTYPE_BF16 && impl == 0) {
            // the engine's bf16 GEMM takes bf16 activations (gemm_tile.hpp): round A first, as the producing kernel would have
            bf16_t* Ab = nullptr;
            HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&Ab), (size_t)M * K * sizeof(bf16_t)));
            hipLaunchKernelGGL(f32_to_bf16_rows_kernel, dim3(ceil_div(M * K, 256)), dim3(256), 0, s, A, lda, Ab, K, M, K);
            GemmTArgs t{Ab, K, reinterpret_cast<const bf16_t*>(W), bias, R, ldr, C, ldc, nullptr, 0, M, N, K, act};
            t.xcd_swizzle = 1;
            r = launch_gemm_tile<bf16_t>(t, s);
            (void)hipStreamSynchronize(s);
            (void)hipFree(Ab);
        } else if (wdtype == MA_DTYPE_BF16) r = launch_gemm<bf16_t>(g, 1, s);                 // the scalar cross-check kernel
        else if (wdtype == MA_DTYPE_F32) r = launch_gemm<float>(g, impl, s);
        else throw MaError(MA_ERR_INVALID, "ma_op_gemm: wdtype");
        if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("ma_op_gemm: ") + hipGetErrorString(r));
    });
}

// the bf16 policy's dense GEMM on its native operands (gemm_tile.hpp): A (M, lda) bf16, W (N, K) bf16; fp32 output C and / or bf16 output Cb
int ma_op_gemm_bf16(const void* A, int lda, const void* W, const float* bias, const float* R, int ldr, float* C, int ldc, void* Cb, int ldcb, int M, int N,
                    int K, int act, void* stream) {
    return guarded(nullptr, [&] {
        if (!A || !W || (!C && !Cb)) throw MaError(MA_ERR_INVALID, "ma_op_gemm_bf16: null pointer");
        GemmTArgs t{reinterpret_cast<const bf16_t*>(A), lda, reinterpret_cast<const bf16_t*>(W), bias, R, ldr, C, ldc, reinterpret_cast<bf16_t*>(Cb), ldcb, M, N, K, act};
        t.xcd_swizzle = 1;
        static int n_cus = -1;
        if (n_cus < 0) { int dev = 0; hipDeviceProp_t prop; n_cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 0; }
        hipError_t r = H16_CALL(g_op_hdt, HT, launch_gemm_dense<HT>(t, n_cus, reinterpret_cast<hipStream_t>(stream)));
        if (r != hipSuccess) throw MaError(r == hipErrorInvalidValue ? MA_ERR_INVALID : MA_ERR_HIP, std::string("ma_op_gemm_bf16: ") + hipGetErrorString(r));
    });
}

int ma_op_layernorm(const float* x, int ldx, const float* g, const float* b, float eps, float* y, int ldy, int rows, int D, void* stream) {
    return guarded(nullptr, [&] {
        if (!x || !g || !b || !y) throw MaError(MA_ERR_INVALID, "ma_op_layernorm: null pointer");
        launch_ln_rows2<float>(x, ldx, RowMap{0, 0, 0}, g, b, eps, y, ldy, (float*)nullptr, 0, RowMap{0, 0, 0}, rows, D, reinterpret_cast<hipStream_t>(stream));
        HIP_CHECK(hipGetLastError());
    });
}

int ma_op_attention(const float* Q, int q_rs, int q_hs, const float* K, int k_rs, int k_hs, const float* V, int v_rs, int v_hs, float* O, int o_rs, int Sq,
                    int Sk, int H, float scale, int causal_offset, int round_bf16, void* stream) {
    return guarded(nullptr, [&] {
        if (!Q || !K || !V || !O) throw MaError(MA_ERR_INVALID, "ma_op_attention: null pointer");
        AttnArgs a{Q, q_rs, q_hs, K, k_rs, k_hs, V, v_rs, v_hs, O, o_rs, Sq, Sk, H, scale, causal_offset, round_bf16};
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        if (round_bf16 == 4) {                               // bf16 tensors, the engine's default kernel (attn2.hpp): V^T packing + swapped-operand attention
            bf16_t* vt = nullptr;
            HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&vt), attn2_vt_elems(Sk, H, 1) * sizeof(bf16_t)));
            hipError_t r = H16_CALL(g_op_hdt, HT, launch_attention2<HT>(a, vt, s));
            (void)hipStreamSynchronize(s);
            (void)hipFree(vt);
            if (r != hipSuccess) throw MaError(r == hipErrorInvalidValue ? MA_ERR_INVALID : MA_ERR_HIP, std::string("ma_op_attention: ") + hipGetErrorString(r));
            return;
        }
        HIP_CHECK(launch_attention(a, s));
    });
}

int ma_op_decode_attention(int kvdtype, const float* q, const void* kcache, const void* vcache, int H, int max_seq, int len, float* out,
                           void* workspace, void* stream) {
    return guarded(nullptr, [&] {
        if (!q || !kcache || !vcache || !out || !workspace || H < 1 || len < 1 || len > max_seq) throw MaError(MA_ERR_INVALID, "ma_op_decode_attention: bad arguments");
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        float* ws = reinterpret_cast<float*>(workspace);
        hipError_t r;
        if (kvdtype == MA_DTYPE_BF16) r = launch_attn_decode<bf16_t>(q, kcache, vcache, H, max_seq, nullptr, len, 1, ws, s);
        else if (kvdtype == MA_DTYPE_F16) r = launch_attn_decode<f16_t>(q, kcache, vcache, H, max_seq, nullptr, len, 1, ws, s);
        else if (kvdtype == MA_DTYPE_F32) r = launch_attn_decode<float>(q, kcache, vcache, H, max_seq, nullptr, len, 0, ws, s);
        else throw MaError(MA_ERR_INVALID, "ma_op_decode_attention: kvdtype");
        HIP_CHECK(r);
        // in the engine the merge of the split partials is the prologue of the out_proj GEMV; here it runs on its own
        hipLaunchKernelGGL(attn_merge_kernel, dim3(ceil_div(H * 16, 256)), dim3(256), 0, s, ws, H, out);
        HIP_CHECK(hipGetLastError());
    });
}

// batched single-query attention, final form (attn_decode_final_kernel): B rows, each its own cache plane, all of length `len`;
// out = bf16 [B][H * 64]
int ma_op_decode_attention_rows(const float* q, const void* kcache, const void* vcache, int H, int max_seq, int len, int B, size_t kv_row_stride,
                                int waves, int halves, void* out, void* stream) {
    return guarded(nullptr, [&] {
        if (!q || !kcache || !vcache || !out || H < 1 || B < 1 || len < 1 || len > max_seq || kv_row_stride < (size_t)H * max_seq * 64)
            throw MaError(MA_ERR_INVALID, "ma_op_decode_attention_rows: bad arguments");
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        if (halves == 2) {                                   // two blocks per (row, head): scratch granules + error word for this call
            unsigned long long* g = nullptr; unsigned* er = nullptr;
            HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&g), (size_t)B * H * ATTN_PAIR_GRANULES * sizeof(unsigned long long) + 64));
            er = reinterpret_cast<unsigned*>(g + (size_t)B * H * ATTN_PAIR_GRANULES);
            (void)hipMemsetAsync(g, 0, (size_t)B * H * ATTN_PAIR_GRANULES * sizeof(unsigned long long) + 64, s);
            hipError_t r = H16_CALL(g_op_hdt, HT, launch_attn_decode_final<HT>(q, kcache, vcache, H, max_seq, nullptr, len, 1, reinterpret_cast<bf16_t*>(out), H * 64, s, B, H * 64, kv_row_stride, 8, g, er, 3));
            unsigned herr = 0;
            (void)hipMemcpyAsync(&herr, er, sizeof(unsigned), hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            (void)hipFree(g);
            HIP_CHECK(r);
            if (herr) throw MaError(MA_ERR_HIP, "ma_op_decode_attention_rows: the hand-over between the two blocks of a pair timed out");
        } else if (halves == 1 || halves == 0) {
            HIP_CHECK(H16_CALL(g_op_hdt, HT, launch_attn_decode_final<HT>(q, kcache, vcache, H, max_seq, nullptr, len, 1, reinterpret_cast<bf16_t*>(out), H * 64, s, B, H * 64, kv_row_stride, waves)));
        } else throw MaError(MA_ERR_INVALID, "ma_op_decode_attention_rows: halves must be 1 or 2");
    });
}

// ---- batched decode step kernels (gemm_decode.hpp) --------------------------------------------------------------------
int ma_op_gemm_dec(const void* W, const float* bias, const void* xb, const float* res, float* y, void* yb, int N, int K, int B, int act, int ksplit,
                   void* stream) {
    return guarded(nullptr, [&] {
        if (!W || !xb || (!y && !yb)) throw MaError(MA_ERR_INVALID, "ma_op_gemm_dec: null pointer");
        GemmDecArgs a{};
        a.W = reinterpret_cast<const bf16_t*>(W); a.bias = bias; a.xb = reinterpret_cast<const bf16_t*>(xb); a.xb_stride = K;
        a.res = res; a.res_stride = N; a.y = y; a.y_stride = N; a.yb = reinterpret_cast<bf16_t*>(yb); a.yb_stride = N;
        a.N = N; a.K = K; a.B = B; a.act = act; a.epi = EPI_PLAIN; a.ksplit = ksplit;
        hipError_t r = H16_CALL(g_op_hdt, HT, launch_gemm_dec<HT>(a, reinterpret_cast<hipStream_t>(stream)));
        if (r != hipSuccess) throw MaError(r == hipErrorInvalidValue ? MA_ERR_INVALID : MA_ERR_HIP, std::string("ma_op_gemm_dec: ") + hipGetErrorString(r));
    });
}

int ma_op_gemm_dec_ln(const void* W, const float* bias, const float* pin, int parts, const float* pbias, const float* pres, const float* ln_g,
                      const float* ln_b, float eps, float* xn_out, float* y, void* yb, int N, int B, int act, void* stream) {
    return guarded(nullptr, [&] {
        if (!W || !pin || !ln_g || !ln_b || (!y && !yb)) throw MaError(MA_ERR_INVALID, "ma_op_gemm_dec_ln: null pointer");
        GemmDecArgs a{};
        a.W = reinterpret_cast<const bf16_t*>(W); a.bias = bias; a.y = y; a.y_stride = N; a.yb = reinterpret_cast<bf16_t*>(yb); a.yb_stride = N;
        a.N = N; a.K = 1024; a.B = B; a.act = act; a.epi = EPI_PLAIN; a.ksplit = 1;
        a.pin = pin; a.pin_stride = 1024; a.pin_parts = parts; a.pbias = pbias; a.pres = pres; a.pres_stride = 1024;
        a.ln_g = ln_g; a.ln_b = ln_b; a.ln_eps = eps; a.xn_out = xn_out; a.xn_stride = 1024;
        hipError_t r = H16_CALL(g_op_hdt, HT, launch_gemm_dec_ln<HT>(a, reinterpret_cast<hipStream_t>(stream)));
        if (r != hipSuccess) throw MaError(r == hipErrorInvalidValue ? MA_ERR_INVALID : MA_ERR_HIP, std::string("ma_op_gemm_dec_ln: ") + hipGetErrorString(r));
    });
}

int ma_op_gemm_dec_qkv(const void* W, const float* bias, const void* xb, float* q, void* kcache, void* vcache, int H, int max_seq, int pos, int B,
                       size_t kv_row_stride, void* stream) {
    return guarded(nullptr, [&] {
        if (!W || !xb || !q || !kcache || !vcache || pos < 0 || pos >= max_seq) throw MaError(MA_ERR_INVALID, "ma_op_gemm_dec_qkv: bad arguments");
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        DecState* st = nullptr;
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&st), (size_t)B * sizeof(DecState)));
        try {
            HIP_CHECK(hipMemsetAsync(st, 0, (size_t)B * sizeof(DecState), s));
            hipLaunchKernelGGL(set_pos_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, s, st, 1, pos, 5, B);
            HIP_CHECK(hipGetLastError());
            GemmDecArgs a{};
            a.W = reinterpret_cast<const bf16_t*>(W); a.bias = bias; a.xb = reinterpret_cast<const bf16_t*>(xb); a.xb_stride = H;
            a.y = q; a.y_stride = H; a.N = 3 * H; a.K = H; a.B = B; a.ksplit = 1; a.epi = EPI_QKV;
            a.kcache = kcache; a.vcache = vcache; a.kv_row_stride = kv_row_stride; a.H = H; a.max_seq = max_seq; a.st = st;
            hipError_t r = H16_CALL(g_op_hdt, HT, launch_gemm_dec<HT>(a, s));
            if (r != hipSuccess) throw MaError(MA_ERR_HIP, std::string("ma_op_gemm_dec_qkv: ") + hipGetErrorString(r));
            HIP_CHECK(hipStreamSynchronize(s));
        } catch (...) { (void)hipFree(st); throw; }
        HIP_CHECK(hipFree(st));
    });
}

int ma_op_rows_prologue(int pro, const float* x, int nparts, int B, const float* bias, const float* res, const float* ln_g, const float* ln_b, float ln_eps,
                        const float* attn_ws, int attn_heads, float* xn_out, void* xb_out, int K, void* stream) {
    return guarded(nullptr, [&] {
        if (!xb_out || (pro != PRO_ATTN && !x) || (pro == PRO_ATTN && !attn_ws) || (pro == PRO_LN && (!ln_g || !ln_b)))
            throw MaError(MA_ERR_INVALID, "ma_op_rows_prologue: null pointer");
        RowsProArgs a{};
        a.x = x; a.x_stride = K; a.nparts = nparts < 1 ? 1 : nparts; a.B = B; a.bias = bias; a.res = res; a.res_stride = K;
        a.ln_g = ln_g; a.ln_b = ln_b; a.ln_eps = ln_eps; a.attn_ws = attn_ws; a.attn_ws_stride = attn_workspace_floats(attn_heads); a.attn_heads = attn_heads;
        a.xn_out = xn_out; a.xn_stride = K; a.xb = reinterpret_cast<bf16_t*>(xb_out); a.xb_stride = K; a.K = K;
        hipError_t r = H16_CALL(g_op_hdt, HT, launch_rows_prologue<HT>(a, pro, B, reinterpret_cast<hipStream_t>(stream)));
        if (r != hipSuccess) throw MaError(r == hipErrorInvalidValue ? MA_ERR_INVALID : MA_ERR_HIP, std::string("ma_op_rows_prologue: ") + hipGetErrorString(r));
    });
}

int ma_op_occupy_cus(int n_blocks, int lds_bytes, int64_t microseconds, const int32_t* release, void* stream) {
    return guarded(nullptr, [&] {
        if (n_blocks < 1 || n_blocks > 4096 || lds_bytes < 64 || lds_bytes > 160 * 1024 || microseconds < 1 || microseconds > 2000000)
            throw MaError(MA_ERR_INVALID, "ma_op_occupy_cus: bad arguments");
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        hipLaunchKernelGGL(occupy_kernel, dim3(n_blocks), dim3(64), (size_t)lds_bytes, reinterpret_cast<hipStream_t>(stream), (unsigned long long)microseconds * 100ull,
                           reinterpret_cast<const int*>(release), (unsigned*)nullptr);
        HIP_CHECK(hipGetLastError());
    });
}

int ma_op_set_half_dtype(int dtype) {
    if (dtype != MA_DTYPE_BF16 && dtype != MA_DTYPE_F16) return MA_ERR_INVALID;
    g_op_hdt = dtype;
    return MA_OK;
}

int ma_op_stream_copy(void* dst, const void* src, size_t bytes, int mode, void* stream) {
    return guarded(nullptr, [&] {
        if (!dst || !src || bytes % 16 || mode < 0 || mode > 2) throw MaError(MA_ERR_INVALID, "ma_op_stream_copy: null pointer, size not a multiple of 16 or mode outside 0 .. 2");
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        const size_t n16 = bytes / 16;
        const u32x4* sp = reinterpret_cast<const u32x4*>(src); u32x4* dp = reinterpret_cast<u32x4*>(dst);
        if (mode == 0) hipLaunchKernelGGL(stream_copy_kernel<0>, dim3(2048), dim3(256), 0, s, sp, dp, n16);
        else if (mode == 1) hipLaunchKernelGGL(stream_copy_kernel<1>, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, sp, dp, n16);
        else hipLaunchKernelGGL(stream_copy_kernel<2>, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, sp, dp, n16);
        HIP_CHECK(hipGetLastError());
    });
}

size_t ma_decode_attention_workspace_bytes(int H) {
    if (H < 1) return 0;
    return attn_workspace_floats(H) * sizeof(float);
}

// ---- measurement ---------------------------------------------------------------------------------------------------
int ma_profile_decode(ma_engine* e, int kv_len, int steps, ma_kernel_timing* out, void* stream) {
    if (!e || !out) return MA_ERR_INVALID;
    return guarded(e, [&] {
        require_ready(e);
        if (steps < 1 || steps > 64) throw MaError(MA_ERR_INVALID, "steps must be in [1,64]");
        if (kv_len < e->T + 1 || kv_len + 3 * steps + 8 > e->maxseq) throw MaError(MA_ERR_INVALID, "kv_len out of range");
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        std::memset(out, 0, sizeof(*out));
        ma_sample_cfg sc = resolve_sample_cfg(e, nullptr);
        sc.suppress_eos = 1;
        const int B = std::max(1, std::min(e->opt_profile_batch, e->cfg.max_batch));
        const int impl = persist_selected(e, B, 0) ? 1 : 0;
        if (impl == 1) ensure_embtab(e, s);
        ensure_graphs(e, B, impl);
        auto reset = [&] {
            init_state(e, s, sc, B, e->maxnew);
            // state as if t tokens had been generated and the cache held kv_len-1 rows
            hipLaunchKernelGGL(set_pos_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, s, e->d_st, kv_len - e->T, kv_len - 1, 5, B);
            HIP_CHECK(hipGetLastError());
        };
        hipEvent_t a, b;
        HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
        // One event pair around `steps` back-to-back steps (no per-launch events: an event record between two launches
        // costs more than the launch boundary it would measure).  only_cls >= 0 enqueues just that class of launches, so
        // class time / launches is the average launch duration, boundary to the next launch included -- the same view as
        // a rocprofv3 kernel trace of the replayed graph.  only_cls == -2: graph replays (with row groups on their streams when
        // decode_groups > 1; the per-class and eager figures are always one ungrouped chain on `s`).
        auto timed = [&](int only_cls, int* launches) -> float {
            StepTimer tm; tm.only_cls = only_cls;
            reset();
            if (only_cls == -2) launch_steps(e, s, B, impl, 1);                                   // warm
            else { StepTimer w; w.only_cls = only_cls; enqueue_decode_step(e, s, -1, w, Rows{0, B}, impl); }
            reset();
            HIP_CHECK(hipEventRecord(a, s));
            if (only_cls == -2) launch_steps(e, s, B, impl, steps);          // what generate() runs: row groups on their streams, joined on s
            else for (int i = 0; i < steps; ++i) {
                enqueue_decode_step(e, s, -1, tm, Rows{0, B}, impl);
                if (only_cls >= 0 && only_cls != 3) {
                    // without the pick launch the state would not advance, and the fused launches tag their in-launch exchanges
                    // with the cache position: move it by hand (one tiny launch per step, charged to the class being timed)
                    hipLaunchKernelGGL(set_pos_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, s, e->d_st, kv_len - e->T + i + 1, kv_len + i, 5, B);
                    HIP_CHECK(hipGetLastError());
                }
            }
            HIP_CHECK(hipEventRecord(b, s));
            HIP_CHECK(hipStreamSynchronize(s));
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, a, b));
            if (launches) *launches = only_cls >= 0 ? tm.launched[only_cls] : 0;
            return ms;
        };
        out->step_ms_eager = timed(-1, nullptr) / steps;
        if (e->cfg.use_graph) out->step_ms_graph = timed(-2, nullptr) / steps;
        for (int cls : {0, 1, 2, 3}) {
            if ((impl == 1) != (cls == 2)) continue;          // the persistent step is one launch of its own class
            int n = 0;
            out->ms[cls] = timed(cls, &n);
            out->launches[cls] = n;
        }
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        if (impl == 1) check_persist_error(e, s);
        else check_chain_error(e, s);                       // timings of zero-filled exchanges are not timings
    });
}

int ma_trace_decode(ma_engine* e, int kv_len, uint64_t* host_out, int max_launches, int max_blocks, int32_t* kinds, int32_t* blocks, int32_t* n_launches,
                    void* stream) {
    if (!e || !host_out || !kinds || !blocks || !n_launches || max_launches < 1 || max_blocks < 1) return MA_ERR_INVALID;
    return guarded(e, [&] {
        require_ready(e);
        if (kv_len < e->T + 1 || kv_len + 16 > e->maxseq) throw MaError(MA_ERR_INVALID, "kv_len out of range");
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        ma_sample_cfg sc = resolve_sample_cfg(e, nullptr);
        sc.suppress_eos = 1;
        const int TB = std::max(1, std::min(e->opt_profile_batch, e->cfg.max_batch));      // rows of the traced step (option profile_batch)
        init_state(e, s, sc, TB, e->maxnew);
        hipLaunchKernelGGL(set_pos_kernel, dim3(ceil_div(TB, 64)), dim3(64), 0, s, e->d_st, kv_len - e->T, kv_len - 1, 5, TB);
        HIP_CHECK(hipGetLastError());
        const size_t n64 = (size_t)max_launches * max_blocks * 4;
        unsigned long long* d_tr = nullptr;
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_tr), n64 * sizeof(unsigned long long)));
        try {
            HIP_CHECK(hipMemsetAsync(d_tr, 0, n64 * sizeof(unsigned long long), s));
            StepTimer none;
            for (int i = 0; i < 3; ++i) enqueue_decode_step(e, s, -1, none, Rows{0, TB});            // warm: clocks, caches
            std::vector<int> k, b;
            StepTimer tm; tm.tr = d_tr; tm.tr_max_launches = max_launches; tm.tr_max_blocks = max_blocks; tm.tr_kind = &k; tm.tr_blocks = &b;
            enqueue_decode_step(e, s, -1, tm, Rows{0, TB});
            HIP_CHECK(hipMemcpyAsync(host_out, d_tr, n64 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            *n_launches = (int)k.size();
            for (size_t i = 0; i < k.size(); ++i) { kinds[i] = k[i]; blocks[i] = b[i]; }
            check_chain_error(e, s);
        } catch (...) { (void)hipFree(d_tr); throw; }
        HIP_CHECK(hipFree(d_tr));
    });
}

// In-kernel timeline of ONE persistent decode step (diagnostics): for every workgroup, the 100 MHz real-time counter at kernel
// start, then two stamps per edge (local share published = start of the sweep | gather complete), then the end.
int ma_persist_trace(ma_engine* e, int kv_len, uint64_t* host_out, int32_t* n_events, void* stream) {
    if (!e || !host_out || !n_events) return MA_ERR_INVALID;
#ifndef MA_EXPERIMENTAL
    return guarded(e, [&] { (void)kv_len; (void)stream; throw MaError(MA_ERR_STATE, "ma_persist_trace needs a library built with MA_EXPERIMENTAL=1"); });
#else
    return guarded(e, [&] {
        require_ready(e);
        if (!e->persist_shape) throw MaError(MA_ERR_STATE, "the persistent decode step is not available for this configuration / device");
        if (kv_len < e->T + 1 || kv_len + 16 > e->maxseq) throw MaError(MA_ERR_INVALID, "kv_len out of range");
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        ma_sample_cfg sc = resolve_sample_cfg(e, nullptr);
        sc.suppress_eos = 1;
        ensure_embtab(e, s);
        init_state(e, s, sc, 1, e->maxnew);
        hipLaunchKernelGGL(set_pos_kernel, dim3(1), dim3(1), 0, s, e->d_st, kv_len - e->T, kv_len - 1, 5, 1);
        HIP_CHECK(hipGetLastError());
        StepTimer none;
        for (int i = 0; i < 3; ++i) enqueue_persist_step(e, s, none);
        const size_t tr_words = (size_t)PS_CUS * (PS_TRACE_EVENTS + PS_TRACE2_EVENTS);
        HIP_CHECK(hipMemsetAsync(e->d_ptrace, 0, tr_words * sizeof(u64), s));
        enqueue_persist_step(e, s, none, e->d_ptrace);
        HIP_CHECK(hipMemcpyAsync(host_out, e->d_ptrace, tr_words * sizeof(u64), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        check_persist_error(e, s);
        *n_events = 2 * (6 * e->cfg.layers + 1) + 2;
    });
#endif
}

// the last decode step's logits of batch row `row` (V floats, device -> caller's device buffer): parity tests compare the two
// step implementations bit for bit
int ma_engine_read_logits(ma_engine* e, int row, float* out, void* stream) {
    if (!e || !out) return MA_ERR_INVALID;
    return guarded(e, [&] {
        if (row < 0 || row >= e->cfg.max_batch) throw MaError(MA_ERR_INVALID, "row out of range");
        HIP_CHECK(hipMemcpyAsync(out, e->d_logits + (size_t)row * e->V, (size_t)e->V * sizeof(float), hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream)));
    });
}

// 1 when the persistent decode step can run on this engine (bf16, 350M layer shape, 256-CU device), else 0
int ma_engine_persist_available(ma_engine* e) { return e && e->persist_shape ? 1 : 0; }

}  // extern "C"
