/*
 * meshanything_amd.h -- C ABI of the MI355X-native MeshAnything inference engine.
 *
 * The reference (buaacyw/MeshAnything @ 2024_08_07) has no native layer and no FFI: its hot path sits behind
 * four Python call boundaries.  Each entry point below names the reference call it
 * replaces; `meshanything_amd/model.py` re-exposes them under the reference's own Python names
 * (MeshAnything.forward / point_encoder.encode_latents / transformer.generate / tokenizer(...)).
 *
 * Conventions
 *   - return 0 (MA_OK) on success, a negative MA_ERR_* code on failure; text via ma_last_error().
 *     No C++ exception crosses this boundary.
 *   - all tensor arguments are caller-owned DEVICE pointers (row-major, dense) unless marked "host";
 *     the engine owns weights, KV cache, workspace and the captured hipGraph.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls are asynchronous on that
 *     stream except where noted (ma_generate / ma_forward read back lengths and therefore synchronise).
 *   - one engine per device; an engine is not thread-safe.
 */
#ifndef MESHANYTHING_AMD_H
#define MESHANYTHING_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MA_API __attribute__((visibility("default")))

enum {
    MA_OK = 0,
    MA_ERR_INVALID = -1,        /* bad argument / config */
    MA_ERR_HIP = -2,            /* a HIP runtime call failed */
    MA_ERR_STATE = -3,          /* call made in the wrong state (e.g. weights not loaded) */
    MA_ERR_UNKNOWN_TENSOR = -4, /* ma_engine_load_weights: key not part of the checkpoint layout */
    MA_ERR_SHAPE = -5,          /* tensor shape/dtype does not match the layout */
    MA_ERR_MISSING = -6,        /* ma_engine_finalize_weights: required tensors were never loaded */
    MA_ERR_NCCL = -7            /* RCCL call failed / librccl not loadable */
};

/* element types: engine policy (ma_config.dtype) and source-tensor dtypes of ma_tensor_desc */
enum { MA_DTYPE_F32 = 0, MA_DTYPE_BF16 = 1, MA_DTYPE_F16 = 2 };

/* Shape + policy.  Field order mirrors meshanything_amd/config.py::MAConfig (all int32).
 * Defaults of the 350M checkpoint in comments; reference sources: MeshAnything/miche/shapevae-256.yaml:7-19,
 * MeshAnything/models/meshanything.py:18,27,88-113, main.py:77-80. */
typedef struct ma_config {
    int32_t struct_size;   /* = sizeof(ma_config) */
    /* point encoder (Michelangelo perceiver) */
    int32_t n_points;      /* 4096 */
    int32_t num_freqs;     /* 8    */
    int32_t enc_width;     /* 768  */
    int32_t enc_heads;     /* 12   */
    int32_t num_latents;   /* 256 (+1 shape token = cond_length 257) */
    int32_t enc_layers;    /* 8    */
    int32_t shape_layers;  /* 16   */
    int32_t embed_dim;     /* 64   */
    /* autoregressive decoder (OPT-350m shape, post-LN, ReLU) */
    int32_t hidden;        /* 1024 */
    int32_t heads;         /* 16   */
    int32_t layers;        /* 24   */
    int32_t ffn;           /* 4096 */
    int32_t codebook_size; /* 8192 (vocab = +3: bos 0, eos 1, pad 2) */
    int32_t codebook_dim;  /* 1024 */
    int32_t n_max_faces;   /* 800  (max_new_tokens = 9*faces + 2) */
    int32_t max_positions; /* 18259 (embed_positions has +2 offset rows) */
    /* detokenizer (BERT-base shape, 6 layers) */
    int32_t tok_width;     /* 768  */
    int32_t tok_heads;     /* 12   */
    int32_t tok_layers;    /* 6    */
    int32_t tok_ffn;       /* 3072 */
    int32_t tok_max_pos;   /* 18000 */
    int32_t discrete_num;  /* 128  */
    /* engine policy */
    int32_t max_batch;     /* largest B accepted by encode/generate/detokenize/forward */
    int32_t dtype;         /* MA_DTYPE_BF16: bf16 weights + KV, GEMM/attention inputs rounded to bf16, fp32 accumulate (BASELINE.json's policy);
                              MA_DTYPE_F16: the same with IEEE half -- the reference's own arithmetic (fp16 autocast, main.py:114-118,149);
                              MA_DTYPE_F32: everything fp32 ("exact" mode for the parity gates) */
    int32_t kv_splits;     /* reserved (the decode attention always splits a head's cache into 16 equal chunks) */
    int32_t use_graph;     /* 1: replay one captured decode step (hipGraph); 0: eager launches */
    int32_t enc_exact;     /* 16-bit policies: 1 = the point encoder (ma_encode: encode_latents + process_point_feature, and the detokenizer's
                              projection of the latents) keeps fp32 weights and computes in fp32 on the fp32 matrix path, so the encoder
                              activations meet the fp32 policy's 1e-5 while prefill / decode / detokenizer stay 16-bit; 0 = everything in
                              the policy dtype.  Ignored under MA_DTYPE_F32. */
} ma_config;

typedef struct ma_engine ma_engine;

/* One checkpoint tensor, by its key in the reference state dict (main.py:99-104; layout SURVEY.md Appendix A).
 * `data` is a HOST pointer to a dense row-major array of `dtype`. */
typedef struct ma_tensor_desc {
    const char *name;
    int32_t dtype;          /* MA_DTYPE_F32 | MA_DTYPE_BF16 | MA_DTYPE_F16 */
    int32_t ndim;           /* 1..3 */
    int64_t shape[4];
    const void *data;
} ma_tensor_desc;

/* Decoding options of transformer.generate(...) as called at meshanything.py:143-162. */
typedef struct ma_sample_cfg {
    int32_t struct_size;    /* = sizeof(ma_sample_cfg) */
    int32_t do_sample;      /* 0: greedy (num_beams=1); 1: top_k -> top_p -> multinomial */
    int32_t top_k;          /* 50 */
    float   top_p;          /* 0.95 */
    int32_t max_new_tokens; /* <= 9*n_max_faces + 2; 0 = that maximum */
    int32_t suppress_eos;   /* 1: never emit eos (full-length throughput runs on random weights) */
    int32_t check_every;    /* poll the all-rows-finished flag every this many steps (0 = 64) */
    int32_t logits_first_step; /* with logits_out: the first step whose logits are kept, in [0, max_new_tokens) (0 = all; outside that range:
                                  MA_ERR_INVALID); ignored, whatever its value, when logits_out is NULL; see logits_out */
    uint64_t seed;          /* in-kernel uniform stream when `uniforms` is NULL */
    const float *uniforms;  /* DEVICE (B, max_new_tokens) uniforms in [0,1), or NULL.  Injected uniforms define
                               sampling parity with the oracle (the reference's Philox stream is not reproducible). */
    /* teacher forcing (parity along the REFERENCE's own token path, tests/test_gpu_reference_anchor.py): when non-NULL, step t still
     * picks its token from its logits and reports it in `tokens`, but the token FED to step t + 1 is forced_tokens[b][t] -- the engine
     * walks the given stream and `tokens` shows what it would have chosen at every step of it.  A forced eos finishes the row. */
    const int64_t *forced_tokens;   /* DEVICE (B, max_new_tokens) int64, or NULL */
    /* when non-NULL, the logits every generated token was picked from: DEVICE (B, max_new_tokens, codebook_size + 3) fp32, row [b][t] =
     * the distribution of token t (as returned by ma_engine_read_logits for the last step; eos is NOT masked in the copy).  With
     * logits_first_step = f > 0 only steps t >= f are kept: DEVICE (B, max_new_tokens - f, vocab), row [b][t - f] (deep-cache parity
     * checks of large batches: 64 rows x 7202 steps of logits would be 15 GB) */
    float *logits_out;
} ma_sample_cfg;

/* ---- lifecycle ---------------------------------------------------------------------------------------- */
MA_API const char *ma_version(void);
MA_API const char *ma_last_error(const ma_engine *e);      /* e may be NULL: last error of a failed create */
/* replaces: MeshAnything(args) construction, meshanything.py:83-123 */
MA_API int  ma_engine_create(ma_engine **out, const ma_config *cfg, int device);
MA_API void ma_engine_destroy(ma_engine *e);
/* integer options (debug / A-B switches); see DESIGN.md.  Unknown names -> MA_ERR_INVALID. */
MA_API int  ma_engine_set_option(ma_engine *e, const char *name, int64_t value);
/* reads an option back as the engine will apply it (e.g. "fuse_qkv_attn" is 1 only if the option is on AND the configuration is
 * eligible); names: fuse_qkv_attn, fuse_oproj_fc1, decode_impl, persist_available, use_graph, dense_rows, mfma_min_batch, ...
 * (INTEGRATION.md section 3 lists them all, with the read-only health counters of the fused launches).
 * Streams: every entry point enqueues on the caller's stream and is ordered with it.  ma_generate's prefill of >= 8 samples (16-bit
 * policies) additionally runs the last rows of its GEMMs on a second, engine-owned stream of the lowest priority, forked from and
 * joined to the caller's stream inside the call (option "prefill_tail" = 0 keeps everything on the caller's stream). */
MA_API int  ma_engine_get_option(ma_engine *e, const char *name, int64_t *value);

/* ---- weights ------------------------------------------------------------------------------------------ */
/* replaces: safe_open(...) + load_state_dict(strict=True), main.py:99-104.  May be called repeatedly with
 * subsets of the checkpoint; q/k/v projections are fused, matrices converted to the policy dtype.
 * Keys the hot path never reads (embed_tokens, shape_projection, geo_decoder.*) are accepted and dropped. */
MA_API int  ma_engine_load_weights(ma_engine *e, const ma_tensor_desc *tensors, int n);
/* strict=True check: every tensor the hot path needs has been loaded */
MA_API int  ma_engine_finalize_weights(ma_engine *e);
/* the packed device weight arena (for a collective broadcast driven from the host framework) */
MA_API int  ma_engine_arena(ma_engine *e, void **dev_ptr, size_t *bytes);
/* declare the arena valid after it was filled by a broadcast (ranks != root) */
MA_API int  ma_engine_mark_weights_loaded(ma_engine *e);
/* replaces: accelerate.prepare(model) -> DDP initial parameter broadcast, main.py:113-118,146.
 * `nccl_comm` is an ncclComm_t (RCCL); one ncclBroadcast of the arena over xGMI; no per-step collectives. */
MA_API int  ma_engine_broadcast_weights(ma_engine *e, void *nccl_comm, int root, void *stream);

/* host-only arena description (no GPU needed): layout is a pure function of the config */
MA_API int64_t ma_arena_bytes(const ma_config *cfg);
MA_API int     ma_arena_num_entries(const ma_config *cfg);
MA_API int     ma_arena_entry(const ma_config *cfg, int i, char *name, int name_cap, int64_t *offset,
                              int64_t *bytes, int32_t *dtype, int32_t *rows, int32_t *cols);
/* host-only packing of checkpoint tensors into a caller-provided host arena of ma_arena_bytes() bytes
 * (same conversion/fusion as ma_engine_load_weights); ma_engine_upload_arena copies it to the device. */
MA_API int  ma_pack_weights_host(const ma_config *cfg, const ma_tensor_desc *tensors, int n, void *host_arena,
                                 char *err, int err_cap);
MA_API int  ma_engine_upload_arena(ma_engine *e, const void *host_arena, size_t bytes);

/* ---- the hot path ------------------------------------------------------------------------------------- */
/* replaces: point_encoder.encode_latents(pc_normal) (asl_pl_module.py:145-157) and
 * MeshAnything.process_point_feature (meshanything.py:125-132, which calls to_shape_latents).
 *   pc       (B, n_points, 6) xyz+normal, `pc_dtype` F32 or F16 (Dataset yields fp16, main.py:56)
 *   latents  (B, cond_length, enc_width) fp32  -- raw encoder latents (the detokenizer's point_feature)
 *   prefix   (B, cond_length, hidden) fp32     -- decoder prefix (may be NULL to skip)              */
MA_API int  ma_encode(ma_engine *e, const void *pc, int pc_dtype, int B, float *latents, float *prefix, void *stream);

/* the two halves of ma_encode's prefix stage on their own, under the reference's names:
 * replaces: point_encoder.to_shape_latents(latents) (asl_pl_module.py:182-185): (B, num_latents, enc_width) -> same shape */
MA_API int  ma_to_shape_latents(ma_engine *e, const float *latents, int B, float *out, void *stream);
/* replaces: MeshAnything.process_point_feature(point_feature) (meshanything.py:125-132): (B, cond_length, enc_width) -> (B, cond_length, hidden) */
MA_API int  ma_process_point_feature(ma_engine *e, const float *point_feature, int B, float *prefix, void *stream);

/* replaces: transformer.generate(inputs_embeds=prefix, max_new_tokens=..., ...) (meshanything.py:143-162).
 *   tokens      (B, max_new_tokens) int64 device: new tokens only; finished rows padded with pad=2
 *   lengths     host (B): tokens generated per row including its eos
 *   n_generated host: number of valid columns (= max over rows; what generate() would return as shape[1])
 * Synchronises `stream` before returning. */
MA_API int  ma_generate(ma_engine *e, const float *prefix, int B, const ma_sample_cfg *sc, int64_t *tokens,
                        int32_t *lengths, int32_t *n_generated, void *stream);

/* replaces: meshanything.py:163-172 (eos-pad to 9F+2, drop first/last, specials -> -1, others -= 3).
 *   tokens (B rows of n_generated valid columns, ld_tokens elements apart: what generate() returned, in place)
 *   ->  ids (B, 9*n_max_faces) int64 in [-1, codebook_size) */
MA_API int  ma_postprocess_tokens(ma_engine *e, const int64_t *tokens, int ld_tokens, int B, int n_generated, int64_t *ids, void *stream);

/* replaces: MeshAnything.get_codes(indices) (meshanything.py:178-212): ids (B, 9F) in [-1, codebook) -> codes (B, 3F, codebook_dim)
 * fp32, the sum of the three residual-VQ rows of every vertex (pad contributes 0) */
MA_API int  ma_get_codes(ma_engine *e, const int64_t *ids, int B, float *codes, void *stream);

/* replaces: get_codes (meshanything.py:178-212) + tokenizer(ids, codes, point_feature=latents) (50-80).
 *   coords (B, n_max_faces, 3, 3) fp32, NaN rows = invalid faces */
MA_API int  ma_detokenize(ma_engine *e, const int64_t *ids, const float *latents, int B, float *coords, void *stream);
/* the same with the caller's `input_embeds` (B, 3*n_max_faces, codebook_dim) fp32 as the face codes -- the reference's
 * tokenizer(input_ids, input_embeds, point_feature=...) signature (meshanything.py:50-55); codes == NULL: ma_detokenize */
MA_API int  ma_detokenize_embeds(ma_engine *e, const int64_t *ids, const float *codes, const float *latents, int B, float *coords, void *stream);

/* replaces: MeshAnything.forward(pc_normal, sampling) (meshanything.py:134-176): encode -> generate ->
 * postprocess -> detokenize.  `tokens` / `ids` / `latents` may be NULL.  Synchronises. */
MA_API int  ma_forward(ma_engine *e, const void *pc, int pc_dtype, int B, const ma_sample_cfg *sc, float *coords,
                       int64_t *tokens, int32_t *lengths, int32_t *n_generated, int64_t *ids, float *latents, void *stream);

/* ---- kernel-level entry points (parity tests and microbenchmarks call the same kernels the engine runs) */
enum { MA_ACT_NONE = 0, MA_ACT_RELU = 1, MA_ACT_GELU = 2 };
/* y[N] = act(W[N,K] . norm(x)[K] + bias) + res ; optional LayerNorm prologue on x (ln_g != NULL); wdtype F32|BF16.
 * With wdtype BF16, x is rounded to bf16 after the prologue (the engine's bf16 policy). */
MA_API int  ma_op_gemv(int wdtype, const void *W, const float *bias, const float *x, const float *ln_g, const float *ln_b,
                       float ln_eps, const float *res, float *y, float *xn_out, int N, int K, int act, void *stream);
/* C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) + R[M,N]; K % 32 == 0; impl 0 = MFMA, 1 = plain VALU reference kernel */
MA_API int  ma_op_gemm(int wdtype, int impl, const float *A, int lda, const void *W, const float *bias, const float *R, int ldr,
                       float *C, int ldc, int M, int N, int K, int act, void *stream);
/* the same GEMM on the bf16 policy's native operands (csrc/gemm_tile.hpp: 128x128x64 LDS-DMA-staged MFMA tile): A (M, lda) bf16,
 * W (N, K) bf16; fp32 output C and / or bf16 output Cb (either may be NULL); K % 32 == 0, lda % 8 == 0, ld* % 4 == 0 */
MA_API int  ma_op_gemm_bf16(const void *A, int lda, const void *W, const float *bias, const float *R, int ldr, float *C, int ldc,
                            void *Cb, int ldcb, int M, int N, int K, int act, void *stream);
MA_API int  ma_op_layernorm(const float *x, int ldx, const float *g, const float *b, float eps, float *y, int ldy,
                            int rows, int D, void *stream);
/* O[b,q,h*64+d] = softmax(Q K^T * scale) V, head_dim 64; strides in elements.  round_bf16: 0 fp32 tensors, exact fp32 kernel | 1 fp32 tensors
 * rounded to bf16, first-generation matrix-core kernel | 2 fp32 kernel on bf16-rounded q,k,v | 3 bf16 tensors, first-generation kernel |
 * 4 bf16 tensors, the engine's kernel (csrc/attn2.hpp: packed V^T + swapped-operand 32x32x16 MFMA; all strides multiples of 8) */
MA_API int  ma_op_attention(const float *Q, int q_rs, int q_hs, const float *K, int k_rs, int k_hs, const float *V, int v_rs,
                            int v_hs, float *O, int o_rs, int Sq, int Sk, int H, float scale, int causal_offset /* <0: none */,
                            int round_bf16, void *stream);
/* single-query attention over a KV cache laid out (H, max_seq, 64) of kvdtype; len = number of cached positions.
 * `workspace`: device buffer of ma_decode_attention_workspace_bytes(H) bytes (the split-KV partials). */
MA_API int  ma_op_decode_attention(int kvdtype, const float *q, const void *kcache, const void *vcache, int H, int max_seq,
                                   int len, float *out, void *workspace, void *stream);
MA_API size_t ma_decode_attention_workspace_bytes(int H);
/* the batched decode path's "final" form: one block of `waves` waves (4 | 8 | 16; 0 = the engine's choice for B) per (row, head)
 * over all `len` cached positions, output already normalised and rounded: out = bf16 [B][H*64].  q fp32 [B][H*64]; row b's cache planes start at b * kv_row_stride elements
 * (bf16, each (H, max_seq, 64)).  Replaces [3p] flash_attn_func(q_len = 1) for a batch (meshanything.py:143-162 batch semantics). */
MA_API int  ma_op_decode_attention_rows(const float *q, const void *kcache, const void *vcache, int H, int max_seq, int len, int B,
                                        size_t kv_row_stride, int waves, int halves /* 1 | 2: blocks per (row, head); 2 = in-launch hand-over, 8 waves */,
                                        void *out, void *stream);

/* ---- measurement --------------------------------------------------------------------------------------- */
/* Time the decode step with HIP events on `stream` at KV length `kv_len` (cache contents arbitrary): `steps` back-to-back
 * steps between ONE event pair, (a) eager, (b) as graph replays, (c) once per kernel class with only that class's launches
 * enqueued, so ms[c] / launches[c] is that class's average launch duration including the boundary to the next launch
 * (the view a rocprofv3 kernel trace gives).  Classes: 0 gemv (all weight-streaming launches; out_proj includes the
 * split-KV merge), 1 decode attention, 3 pick/sample. */
typedef struct ma_kernel_timing { int32_t launches[8]; float ms[8]; float step_ms_graph; float step_ms_eager; } ma_kernel_timing;
MA_API int  ma_profile_decode(ma_engine *e, int kv_len, int steps, ma_kernel_timing *out, void *stream);

/* In-kernel timeline of ONE eager decode step at KV length `kv_len`: every weight-streaming / attention launch of the
 * step records, per block, the 100 MHz real-time counter at (0) block start, (1) input vector staged, (2) weights / KV
 * consumed, (3) block end.  host_out[(launch * max_blocks + block) * 4 + point]; kinds[launch]: 0 embed, 1 qkv,
 * 2 attention, 3 out_proj(+merge), 4 fc1, 5 fc2, 6 lm_head; blocks[launch] = grid size.  Diagnostics only. */
MA_API int  ma_trace_decode(ma_engine *e, int kv_len, uint64_t *host_out, int max_launches, int max_blocks, int32_t *kinds,
                            int32_t *blocks, int32_t *n_launches, void *stream);

/* ma_op_gemm_dec_ln (B <= 16, K = 1024): the same GEMM with the LayerNorm prologue of ma_op_rows_prologue inside it -- activation row b =
 * LN(sum of `parts` partial buffers pin[parts][B][1024] + pbias + pres[b]) rounded to bf16; xn_out (B, 1024) fp32 = the LayerNorm
 * output (may be NULL).  The batched decode path's form for small batches ([3p] OPTDecoderLayer post-LN + Linear). */
MA_API int  ma_op_gemm_dec_ln(const void *W, const float *bias, const float *pin, int parts, const float *pbias, const float *pres,
                              const float *ln_g, const float *ln_b, float eps, float *xn_out, float *y, void *yb, int N, int B, int act,
                              void *stream);
/* ---- batched decode step kernels (csrc/gemm_decode.hpp; replace the same nn.Linear calls as ma_op_gemv when B rows step
 * together, meshanything.py:143-162 with batch > 1).  All pointers device.
 * ma_op_gemm_dec: Y[B,N] = act(Xb[B,K] . W[N,K]^T + bias) + res on the bf16 matrix cores; W, Xb bf16; y fp32 and / or yb bf16
 *   output; ksplit > 1: y receives the raw partial sums [ksplit][B][N] (bias / res / act must be null / none).
 * ma_op_gemm_dec_qkv: the fused q/k/v projection epilogue: rows [0,H) -> q (B,H) fp32, [H,2H) / [2H,3H) -> K / V cache
 *   ([B] planes kv_row_stride elements apart, each (H/64, max_seq, 64) bf16) at position `pos`.
 * ma_op_rows_prologue: per-row prologue (pro 0 plain | 1 LayerNorm | 2 merge of the split-KV attention partials): sums
 *   `nparts` partial buffers [nparts][B][K] + bias + res, normalises, writes fp32 (xn_out, may be NULL) and bf16 (xb_out). */
MA_API int  ma_op_gemm_dec(const void *W, const float *bias, const void *xb, const float *res, float *y, void *yb, int N, int K,
                           int B, int act, int ksplit, void *stream);
MA_API int  ma_op_gemm_dec_qkv(const void *W, const float *bias, const void *xb, float *q, void *kcache, void *vcache, int H,
                               int max_seq, int pos, int B, size_t kv_row_stride, void *stream);
MA_API int  ma_op_rows_prologue(int pro, const float *x, int nparts, int B, const float *bias, const float *res, const float *ln_g,
                                const float *ln_b, float ln_eps, const float *attn_ws, int attn_heads, float *xn_out,
                                void *xb_out, int K, void *stream);

/* ---- test aid: holds `n_blocks` workgroups of `lds_bytes` of LDS each (163840 = a whole CU) on the device for `microseconds`
 * (bounded: <= 2 s) on `stream`, doing nothing; ends early once `*release` (device-visible host memory, may be NULL) is non-zero.  Lets a test take CUs away from the engine's stream and check that the fused decode
 * launches -- which need their whole grid resident -- fall back to the five-launch chain instead of failing the request.  Has no
 * reference counterpart (the reference never shares a device between streams). */
MA_API int  ma_op_occupy_cus(int n_blocks, int lds_bytes, int64_t microseconds, const int32_t *release, void *stream);

/* ---- the 16-bit format (MA_DTYPE_BF16, the default, or MA_DTYPE_F16) of the kernel-level entry points above that carry no dtype argument
 * (ma_op_gemm_bf16, ma_op_attention mode 4, ma_op_decode_attention_rows, ma_op_gemm_dec*, ma_op_rows_prologue): their "bf16" operands are then
 * IEEE half.  Per calling thread; parity tests run every 16-bit kernel in both formats.  No reference counterpart. */
MA_API int  ma_op_set_half_dtype(int dtype);

/* ---- measurement aid: dst[0, bytes) = src[0, bytes) as a 16-byte-per-lane streaming copy; bytes % 16 == 0; mode 0 = 2048 blocks grid-stride with
 * non-temporal accesses, 1 = one element per thread with plain accesses, 2 = one element per thread non-temporal.  bench.py times all three and
 * reports the box's achievable HBM rate next to the 8 TB/s vendor number.  No reference counterpart. */
MA_API int  ma_op_stream_copy(void *dst, const void *src, size_t bytes, int mode, void *stream);

/* ---- persistent decode step (csrc/experimental/persist.hpp; only in libraries built with MA_EXPERIMENTAL=1 -- measured 1.3-1.6x slower than the launch
 * chain; the product build answers MA_ERR_STATE / 0): the whole batch-1 greedy step as ONE resident launch instead of the
 * 123-launch chain.  Select with ma_engine_set_option(e, "decode_impl", 1); it is used when ma_engine_persist_available()
 * and the call is batch 1 / greedy, otherwise the chain runs.  replaces: the same reference calls as ma_generate's steps
 * (shape_opt.py:318-364,403-410,155; meshanything.py:143-151). */
MA_API int  ma_engine_persist_available(ma_engine *e);
/* host_out: 256 * 320 uint64 (comm wave of every workgroup: start, then per edge {sweep start, gather done}, end; 100 MHz
 * ticks), followed by 256 * 512 uint64 (compute wave 0: per weight op {input ready, weights landed, dots done, published},
 * per attention {partial published, partials gathered, merged output published}) */
MA_API int  ma_persist_trace(ma_engine *e, int kv_len, uint64_t *host_out, int32_t *n_events, void *stream);
/* copies the logits of the most recent decode step of batch row `row` (codebook_size + 3 floats) into a device buffer */
MA_API int  ma_engine_read_logits(ma_engine *e, int row, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MESHANYTHING_AMD_H */
