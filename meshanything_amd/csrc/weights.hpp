// Packed weight arena: layout (a pure function of ma_config) and host-side packing of checkpoint tensors.
//
// The reference loads a safetensors state dict key by key (main.py:99-104); here every tensor the hot path reads is
// copied once into ONE device allocation ("arena"): matrices in the policy dtype (bf16 or fp32) with q/k/v
// projections fused row-wise, everything else (biases, LayerNorm affine, embedding tables, codebook) in fp32.
// One arena = one RCCL broadcast at load time and one contiguous region for the HBM streamer.
#pragma once
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/meshanything_amd.h"
#include "common.hpp"

namespace ma {

struct Entry {
    std::string name;
    int rows, cols;      // cols = padded leading dimension
    int dtype;           // MA_DTYPE_F32, MA_DTYPE_BF16 or MA_DTYPE_F16
    size_t offset, bytes;
    size_t want;         // elements the checkpoint must write into this entry (padding excluded)
};

struct Source {          // one reference state-dict key -> where it lands
    int entry;           // -1: accepted and dropped (unused on the hot path)
    size_t dst_elem;     // element offset inside the entry of source element (0,0)
    int src_rows, src_cols;    // expected source shape (product of leading dims, last dim)
    int take_rows, take_cols;  // the part that is copied (pre_kl keeps only the mean half)
};

struct Layout {
    std::vector<Entry> entries;
    std::unordered_map<std::string, int> entry_by_name;
    std::unordered_map<std::string, Source> sources;
    size_t bytes = 0;

    int add_entry(const std::string& name, int rows, int cols, int dtype) {
        Entry e{name, rows, cols, dtype, bytes, (size_t)rows * cols * (dtype == MA_DTYPE_F32 ? 4 : 2), 0};
        bytes += (e.bytes + 255) & ~(size_t)255;
        entry_by_name[name] = (int)entries.size();
        entries.push_back(e);
        return (int)entries.size() - 1;
    }
    // primary source: counts towards the entry's completeness; alias: same destination under another key
    void add_source(const std::string& key, int entry, size_t dst_elem, int src_rows, int src_cols, int take_rows = -1,
                    int take_cols = -1, bool alias = false) {
        Source s{entry, dst_elem, src_rows, src_cols, take_rows < 0 ? src_rows : take_rows, take_cols < 0 ? src_cols : take_cols};
        if (!alias) entries[entry].want += (size_t)s.take_rows * s.take_cols;
        sources[key] = s;
    }
    void mat(const std::string& key, int rows, int cols, int mdt, int pad_cols = 0) {
        int e = add_entry(key, rows, pad_cols ? pad_cols : cols, mdt);
        add_source(key, e, 0, rows, cols);
    }
    void vec(const std::string& key, int n) { int e = add_entry(key, 1, n, MA_DTYPE_F32); add_source(key, e, 0, 1, n); }
    void tab(const std::string& key, int rows, int cols) { int e = add_entry(key, rows, cols, MA_DTYPE_F32); add_source(key, e, 0, rows, cols); }
    void ignore(const std::string& key) { sources[key] = Source{-1, 0, 0, 0, 0, 0}; }
    const Entry& get(const std::string& name) const { return entries[entry_by_name.at(name)]; }
};

inline void layout_miche_block(Layout& L, const std::string& p, int W, int mdt) {
    L.mat(p + "attn.c_qkv.weight", 3 * W, W, mdt);
    L.mat(p + "attn.c_proj.weight", W, W, mdt);  L.vec(p + "attn.c_proj.bias", W);
    L.vec(p + "ln_1.weight", W); L.vec(p + "ln_1.bias", W);
    L.mat(p + "mlp.c_fc.weight", 4 * W, W, mdt); L.vec(p + "mlp.c_fc.bias", 4 * W);
    L.mat(p + "mlp.c_proj.weight", W, 4 * W, mdt); L.vec(p + "mlp.c_proj.bias", W);
    L.vec(p + "ln_2.weight", W); L.vec(p + "ln_2.bias", W);
}

inline Layout build_layout(const ma_config& c) {
    Layout L;
    const int mdt = c.dtype == MA_DTYPE_F32 ? MA_DTYPE_F32 : (c.dtype == MA_DTYPE_F16 ? MA_DTYPE_F16 : MA_DTYPE_BF16);
    // the point encoder's matrices (and the two projections of its latents in front of the decoder and of the detokenizer): fp32 when
    // cfg.enc_exact asks for the exact encoder under a 16-bit policy (engine.hip, DenseScope)
    const int edt = (c.dtype == MA_DTYPE_F32 || c.enc_exact) ? MA_DTYPE_F32 : mdt;
    const int W = c.enc_width, T = c.num_latents + 1, H = c.hidden, E = c.embed_dim;
    const int fourier = 3 * (2 * c.num_freqs + 1), pin = fourier + 3;
    const std::string PE = "point_encoder.model.", SM = PE + "shape_model.", DEC = "transformer.model.decoder.", TOK = "tokenizer.";
    // ---- point encoder ----
    L.ignore(PE + "shape_projection");
    L.tab(SM + "encoder.query", T, W);
    L.mat(SM + "encoder.input_proj.weight", W, pin, edt, 64);      // K padded 54 -> 64 (zero columns)
    L.vec(SM + "encoder.input_proj.bias", W);
    {
        const std::string p = SM + "encoder.cross_attn.";
        L.mat(p + "attn.c_q.weight", W, W, edt);
        L.mat(p + "attn.c_kv.weight", 2 * W, W, edt);
        L.mat(p + "attn.c_proj.weight", W, W, edt); L.vec(p + "attn.c_proj.bias", W);
        for (int i = 1; i <= 3; ++i) { L.vec(p + "ln_" + std::to_string(i) + ".weight", W); L.vec(p + "ln_" + std::to_string(i) + ".bias", W); }
        L.mat(p + "mlp.c_fc.weight", 4 * W, W, edt); L.vec(p + "mlp.c_fc.bias", 4 * W);
        L.mat(p + "mlp.c_proj.weight", W, 4 * W, edt); L.vec(p + "mlp.c_proj.bias", W);
    }
    for (int n = 0; n < c.enc_layers; ++n) layout_miche_block(L, SM + "encoder.self_attn.resblocks." + std::to_string(n) + ".", W, edt);
    L.vec(SM + "encoder.ln_post.weight", W); L.vec(SM + "encoder.ln_post.bias", W);
    {   // pre_kl: only the mean half (rows [0,E)) is ever used (DiagonalGaussianDistribution.mode, distributions.py:34,69-70)
        int e = L.add_entry(SM + "pre_kl.weight", E, W, edt); L.add_source(SM + "pre_kl.weight", e, 0, 2 * E, W, E, W);
        int b = L.add_entry(SM + "pre_kl.bias", 1, E, MA_DTYPE_F32); L.add_source(SM + "pre_kl.bias", b, 0, 1, 2 * E, 1, E);
    }
    L.mat(SM + "post_kl.weight", W, E, edt); L.vec(SM + "post_kl.bias", W);
    for (int n = 0; n < c.shape_layers; ++n) layout_miche_block(L, SM + "transformer.resblocks." + std::to_string(n) + ".", W, edt);
    // geo_decoder.*: SDF reconstruction head, never run by MeshAnything.forward -> matched by prefix in find_source()
    // ---- top-level prefix projections (A.4) ----
    L.mat("cond_head_proj.weight", H, W, edt); L.vec("cond_head_proj.bias", H);
    L.mat("cond_proj.weight", H, 2 * W, edt);  L.vec("cond_proj.bias", H);
    // ---- ShapeOPT decoder (A.2) ----
    L.ignore(DEC + "embed_tokens.weight");
    L.tab(DEC + "extra_embeds.weight", 3, H);
    L.mat(DEC + "input_layer.weight", H, c.codebook_dim, mdt); L.vec(DEC + "input_layer.bias", H);
    L.tab(DEC + "embed_positions.weight", c.max_positions + 2, H);
    L.tab(DEC + "token_embed_positions.weight", 12, H);
    L.tab(DEC + "cond_embed.weight", 2, H);
    for (int n = 0; n < c.layers; ++n) {
        const std::string p = DEC + "layers." + std::to_string(n) + ".";
        int e = L.add_entry(p + "qkv.weight", 3 * H, H, mdt), b = L.add_entry(p + "qkv.bias", 1, 3 * H, MA_DTYPE_F32);
        const char* nm[3] = {"q_proj", "k_proj", "v_proj"};
        for (int i = 0; i < 3; ++i) {
            L.add_source(p + "self_attn." + nm[i] + ".weight", e, (size_t)i * H * H, H, H);
            L.add_source(p + "self_attn." + nm[i] + ".bias", b, (size_t)i * H, 1, H);
        }
        L.mat(p + "self_attn.out_proj.weight", H, H, mdt); L.vec(p + "self_attn.out_proj.bias", H);
        L.vec(p + "self_attn_layer_norm.weight", H); L.vec(p + "self_attn_layer_norm.bias", H);
        L.mat(p + "fc1.weight", c.ffn, H, mdt); L.vec(p + "fc1.bias", c.ffn);
        L.mat(p + "fc2.weight", H, c.ffn, mdt); L.vec(p + "fc2.bias", H);
        L.vec(p + "final_layer_norm.weight", H); L.vec(p + "final_layer_norm.bias", H);
    }
    {   // quantize_codebooks (1, C, D) -> fp32 table (C, D)
        int e = L.add_entry(DEC + "quantize_codebooks", c.codebook_size, c.codebook_dim, MA_DTYPE_F32);
        L.add_source(DEC + "quantize_codebooks", e, 0, c.codebook_size, c.codebook_dim);
    }
    L.mat("transformer.lm_head.weight", c.codebook_size + 3, H, mdt);
    // ---- detokenizer (A.3) ----
    const int Wt = c.tok_width;
    L.tab(TOK + "pos_embedding.weight", c.tok_max_pos, Wt);
    L.tab(TOK + "point_pe.weight", T, Wt);
    L.vec(TOK + "layernorm.weight", Wt); L.vec(TOK + "layernorm.bias", Wt);
    L.vec(TOK + "point_layernorm.weight", Wt); L.vec(TOK + "point_layernorm.bias", Wt);
    L.mat(TOK + "cond_proj.weight", Wt, W, edt); L.vec(TOK + "cond_proj.bias", Wt);
    L.mat(TOK + "cond_head_proj.weight", Wt, W, edt); L.vec(TOK + "cond_head_proj.bias", Wt);
    L.mat(TOK + "project_down_codebook.weight", Wt, 3 * c.codebook_dim, mdt); L.vec(TOK + "project_down_codebook.bias", Wt);
    L.mat(TOK + "to_coor_logits.0.weight", 9 * c.discrete_num, Wt, mdt); L.vec(TOK + "to_coor_logits.0.bias", 9 * c.discrete_num);
    for (int n = 0; n < c.tok_layers; ++n) {
        const std::string p = TOK + "decoder.layer." + std::to_string(n) + ".";
        // arena entries carry the vanilla HF names; optimum BetterTransformer names are aliases
        int e = L.add_entry(p + "qkv.weight", 3 * Wt, Wt, mdt), b = L.add_entry(p + "qkv.bias", 1, 3 * Wt, MA_DTYPE_F32);
        const char* nm[3] = {"query", "key", "value"};
        for (int i = 0; i < 3; ++i) {
            L.add_source(p + "attention.self." + nm[i] + ".weight", e, (size_t)i * Wt * Wt, Wt, Wt);
            L.add_source(p + "attention.self." + nm[i] + ".bias", b, (size_t)i * Wt, 1, Wt);
        }
        L.add_source(p + "in_proj_weight", e, 0, 3 * Wt, Wt, -1, -1, true);
        L.add_source(p + "in_proj_bias", b, 0, 1, 3 * Wt, -1, -1, true);
        struct Alias { const char* van; const char* fused; int rows, cols; bool mat; };
        const Alias al[] = {
            {"attention.output.dense.weight", "out_proj_weight", Wt, Wt, true}, {"attention.output.dense.bias", "out_proj_bias", 1, Wt, false},
            {"attention.output.LayerNorm.weight", "norm1_weight", 1, Wt, false}, {"attention.output.LayerNorm.bias", "norm1_bias", 1, Wt, false},
            {"intermediate.dense.weight", "linear1_weight", c.tok_ffn, Wt, true}, {"intermediate.dense.bias", "linear1_bias", 1, c.tok_ffn, false},
            {"output.dense.weight", "linear2_weight", Wt, c.tok_ffn, true}, {"output.dense.bias", "linear2_bias", 1, Wt, false},
            {"output.LayerNorm.weight", "norm2_weight", 1, Wt, false}, {"output.LayerNorm.bias", "norm2_bias", 1, Wt, false}};
        for (const Alias& a : al) {
            int en = L.add_entry(p + a.van, a.rows, a.cols, a.mat ? mdt : MA_DTYPE_F32);
            L.add_source(p + a.van, en, 0, a.rows, a.cols);
            L.add_source(p + a.fused, en, 0, a.rows, a.cols, -1, -1, true);
        }
    }
    return L;
}

// coverage bookkeeping: elements written per entry must reach entry.want (fused / aliased destinations uniformly)
struct PackState {
    std::vector<size_t> filled;
};

inline const Source* find_source(const Layout& L, const std::string& key, bool* dropped) {
    *dropped = false;
    auto it = L.sources.find(key);
    if (it != L.sources.end()) { if (it->second.entry < 0) *dropped = true; return &it->second; }
    if (key.find(".geo_decoder.") != std::string::npos) { *dropped = true; return nullptr; }   // sal_perceiver.py:113-158, unused
    return nullptr;
}

inline float src_elem(const void* data, int dtype, size_t i) {
    if (dtype == MA_DTYPE_F32) return reinterpret_cast<const float*>(data)[i];
    if (dtype == MA_DTYPE_BF16) return bf2f(reinterpret_cast<const uint16_t*>(data)[i]);
    return half2float_host(reinterpret_cast<const uint16_t*>(data)[i]);
}

// Convert one checkpoint tensor into its arena bytes.  `emit(offset_in_arena, ptr, nbytes)` receives contiguous pieces.
// The arena is zero-initialised by its owner, so padding columns are never emitted.
template <typename Emit>
inline int pack_tensor(const Layout& L, PackState& ps, const ma_tensor_desc& t, std::string& err, Emit emit) {
    if (!t.name || !t.data || t.ndim < 1 || t.ndim > 3 || t.dtype < MA_DTYPE_F32 || t.dtype > MA_DTYPE_F16) {
        err = "bad tensor descriptor"; return MA_ERR_INVALID;
    }
    bool dropped;
    const Source* s = find_source(L, t.name, &dropped);
    if (dropped) return MA_OK;
    if (!s) { err = std::string("unknown checkpoint key: ") + t.name; return MA_ERR_UNKNOWN_TENSOR; }
    long long lead = 1;
    for (int i = 0; i + 1 < t.ndim; ++i) lead *= t.shape[i];
    const long long last = t.shape[t.ndim - 1];
    if (lead != s->src_rows || last != s->src_cols) {
        char buf[320];
        snprintf(buf, sizeof buf, "shape mismatch for %s: got (%lld x %lld), layout expects (%d x %d)", t.name, lead, last, s->src_rows, s->src_cols);
        err = buf; return MA_ERR_SHAPE;
    }
    const Entry& e = L.entries[s->entry];
    const int esz = e.dtype == MA_DTYPE_F32 ? 4 : 2;
    // contiguous destination (no padding between rows): convert the whole tensor and emit it in one piece
    const bool contiguous = (s->take_cols == e.cols) || s->take_rows == 1;
    const size_t rows_per_emit = contiguous ? (size_t)s->take_rows : 1;
    std::vector<uint8_t> buf(rows_per_emit * s->take_cols * esz);
    for (int r0 = 0; r0 < s->take_rows; r0 += (int)rows_per_emit) {
        for (size_t rr = 0; rr < rows_per_emit; ++rr) {
            const size_t r = r0 + rr;
            uint8_t* dst = buf.data() + rr * s->take_cols * esz;
            if (t.dtype == MA_DTYPE_F32 && esz == 4) {
                std::memcpy(dst, reinterpret_cast<const float*>(t.data) + r * s->src_cols, (size_t)s->take_cols * 4);
            } else {
                for (int k = 0; k < s->take_cols; ++k) {
                    const float v = src_elem(t.data, t.dtype, r * s->src_cols + k);
                    if (esz == 4) reinterpret_cast<float*>(dst)[k] = v;
                    else reinterpret_cast<uint16_t*>(dst)[k] = e.dtype == MA_DTYPE_F16 ? float2half_host(v) : f2bf(v);
                }
            }
        }
        // a matrix source advances by the entry's leading dimension; a vector segment has a single row
        emit(e.offset + (s->dst_elem + (size_t)r0 * e.cols) * esz, buf.data(), buf.size());
    }
    ps.filled[s->entry] += (size_t)s->take_rows * s->take_cols;
    return MA_OK;
}

inline void pack_state_init(const Layout& L, PackState& ps) { ps.filled.assign(L.entries.size(), 0); }

inline bool pack_complete(const Layout& L, const PackState& ps, std::string& missing) {
    missing.clear();
    int n = 0;
    for (size_t i = 0; i < L.entries.size(); ++i)
        if (ps.filled[i] < L.entries[i].want) { if (n++ < 8) { missing += L.entries[i].name; missing += "; "; } }
    if (n > 8) missing += "... (" + std::to_string(n) + " arena entries incomplete)";
    return n == 0;
}

}  // namespace ma
