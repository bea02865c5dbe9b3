// Text-encoder executor: the HF `CLIPTextModel` that `FrozenCLIPEmbedder.forward` runs
// (ldm/modules/encoders/modules.py:137-162; model code: transformers==4.19.2 (environment.yaml:25)
// models/clip/modeling_clip.py CLIPTextTransformer / CLIPEncoderLayer / CLIPAttention / CLIPMLP) -- SURVEY.md 8 f-2.
//   embeddings = token_embedding(ids) + position_embedding(0..L-1)
//   per layer:  x += out_proj(causal_softmax(q k^T d^-1/2) v),  q|k|v = Linear(LayerNorm1(x))     (biases everywhere)
//               x += fc2(quick_gelu(fc1(LayerNorm2(x))))
//   last_hidden_state = final_layer_norm(x)
// Same kernels as the UNet's transformer blocks: igemm (per-head scatter epilogue with bias, plain epilogue with the
// LayerNorm post-op), the flash-attention kernel with a causal mask, layernorm; the token stream is fp32.
#include "clip.h"

#include <math.h>

namespace sdmi {

void ClipText::expect(const std::string& key, std::vector<int64_t> shape, CWKind kind, void** dst, int row0, int total_rows) {
  CWeightSlot s;
  s.key = key; s.shape = std::move(shape); s.kind = kind; s.dst = dst; s.row0 = row0; s.total_rows = total_rows;
  slot_index_[key] = (int)slots_.size();
  slots_.push_back(std::move(s));
}

int ClipText::build(const sdmi_clip_cfg& c) {
  cfg_ = c;
  SDMI_CHECK(c.hidden_size % 64 == 0 && c.intermediate_size % 64 == 0, "hidden / intermediate size must be multiples of 64");
  SDMI_CHECK(c.num_heads >= 1 && c.hidden_size % c.num_heads == 0, "hidden_size % num_heads");
  const int dh = c.hidden_size / c.num_heads;
  SDMI_CHECK(dh == 32 || dh == 40 || dh == 64 || dh == 80 || dh == 128 || dh == 160, "head dim must be one the attention kernel has");
  SDMI_CHECK(c.num_layers >= 1 && c.vocab_size >= 1 && c.max_positions >= 1 && c.hidden_size <= 2560, "bad text-model config");
  const int64_t C = c.hidden_size, I = c.intermediate_size;
  layers_.resize(c.num_layers);
  const std::string tm = "text_model.";
  expect(tm + "embeddings.token_embedding.weight", {c.vocab_size, C}, CW_F32, (void**)&tok_);
  expect(tm + "embeddings.position_embedding.weight", {c.max_positions, C}, CW_F32, (void**)&pos_);
  for (int i = 0; i < c.num_layers; ++i) {   // NOTE: slots point into layers_, which must not reallocate from here on
    CLayer& L = layers_[i];
    const std::string p = tm + "encoder.layers." + std::to_string(i) + ".";
    const char* names[3] = {"q_proj", "k_proj", "v_proj"};
    for (int j = 0; j < 3; ++j) {
      expect(p + "self_attn." + names[j] + ".weight", {C, C}, CW_ROWS16, (void**)&L.wqkv, j * (int)C, 3 * (int)C);
      expect(p + "self_attn." + names[j] + ".bias", {C}, CW_BIAS_ROWS, (void**)&L.bqkv, j * (int)C, 3 * (int)C);
    }
    expect(p + "self_attn.out_proj.weight", {C, C}, CW_ROWS16, (void**)&L.wo, 0, (int)C);
    expect(p + "self_attn.out_proj.bias", {C}, CW_F32, (void**)&L.bo);
    expect(p + "layer_norm1.weight", {C}, CW_F32, (void**)&L.ln[0]);
    expect(p + "layer_norm1.bias", {C}, CW_F32, (void**)&L.ln[1]);
    expect(p + "mlp.fc1.weight", {I, C}, CW_ROWS16, (void**)&L.w1, 0, (int)I);
    expect(p + "mlp.fc1.bias", {I}, CW_F32, (void**)&L.b1);
    expect(p + "mlp.fc2.weight", {C, I}, CW_ROWS16, (void**)&L.w2, 0, (int)C);
    expect(p + "mlp.fc2.bias", {C}, CW_F32, (void**)&L.b2);
    expect(p + "layer_norm2.weight", {C}, CW_F32, (void**)&L.ln[2]);
    expect(p + "layer_norm2.bias", {C}, CW_F32, (void**)&L.ln[3]);
  }
  expect(tm + "final_layer_norm.weight", {C}, CW_F32, (void**)&fln_g_);
  expect(tm + "final_layer_norm.bias", {C}, CW_F32, (void**)&fln_b_);
  return 0;
}

ClipText::~ClipText() {
  for (void* p : owned_) (void)hipFree(p);
}

int ClipText::dev_alloc(void** dst, size_t bytes) {
  if (*dst) return 0;
  SDMI_HIP_OK(hipMalloc(dst, bytes));
  owned_.push_back(*dst);
  return 0;
}

int ClipText::set_weight(const char* key, const float* ptr, const int64_t* shape, int ndim, hipStream_t stream) {
  auto it = slot_index_.find(key);
  if (it == slot_index_.end()) return fail(std::string("unexpected weight key: ") + key);
  CWeightSlot& s = slots_[it->second];
  SDMI_CHECK((int)s.shape.size() == ndim, std::string("rank mismatch for ") + key);
  int64_t numel = 1;
  for (int i = 0; i < ndim; ++i) {
    SDMI_CHECK(shape[i] == s.shape[i], std::string("shape mismatch for ") + key);
    numel *= shape[i];
  }
  DevStage st;
  if (st.acquire(ptr, numel, stream)) return -1;
  int rc = 0;
  switch (s.kind) {
    case CW_F32:
      rc = dev_alloc(s.dst, numel * sizeof(float));
      if (!rc) SDMI_HIP_OK(hipMemcpyAsync(*s.dst, st.dptr, numel * sizeof(float), hipMemcpyDeviceToDevice, stream));
      break;
    case CW_ROWS16:      // rows [row0, row0 + rows) of an fp16 [total_rows][cols] matrix (q | k | v concatenation)
      rc = dev_alloc(s.dst, (size_t)s.total_rows * shape[1] * sizeof(f16));
      if (!rc) rc = launch_pack_rows(st.dptr, (f16*)*s.dst, (int)shape[0], (int)shape[1], s.row0, (int)shape[1], stream);
      break;
    case CW_BIAS_ROWS:   // slice [row0, row0 + n) of a concatenated fp32 bias
      rc = dev_alloc(s.dst, (size_t)s.total_rows * sizeof(float));
      if (!rc)
        SDMI_HIP_OK(hipMemcpyAsync((float*)*s.dst + s.row0, st.dptr, numel * sizeof(float), hipMemcpyDeviceToDevice, stream));
      break;
  }
  if (st.release(stream)) return -1;
  if (rc) return rc;
  s.set = true;
  finalized_ = false;
  return 0;
}

int ClipText::finalize() {
  for (auto& s : slots_)
    if (!s.set) return fail("weight not set: " + s.key);
  if (!zero_) {
    SDMI_HIP_OK(hipMalloc((void**)&zero_, 4096));
    owned_.push_back(zero_);
    SDMI_HIP_OK(hipMemset(zero_, 0, 4096));
  }
  finalized_ = true;
  return 0;
}

int ClipText::forward(const int64_t* ids, float* out, int B, int L, void* workspace, int64_t ws_bytes, hipStream_t stream,
                      bool dry, int64_t* bytes_needed) {
  SDMI_CHECK(dry || finalized_, "sdmi_clip_finalize() has not succeeded yet");
  SDMI_CHECK(B >= 1 && B <= 64 && L >= 1 && L <= cfg_.max_positions, "batch 1..64, 1 <= L <= max_positions");
  SDMI_CHECK(dry || (ids != nullptr && out != nullptr), "ids / out is NULL");
  const int C = cfg_.hidden_size, I = cfg_.intermediate_size, H = cfg_.num_heads, dh = C / H;
  const int M = B * L, Lp = (int)round_up(L, 8);
  const float scale = 1.0f / sqrtf((float)dh);
  FwdBase f;
  f.s = stream; f.B = B; f.zero = zero_; f.precise_1x1 = false;
  int64_t persist_bytes = 0;
  for (int pass = 0; pass < 2; ++pass) {
    const bool d = pass == 0;
    if (pass == 1 && dry) break;
    f.dry = d; f.rc = 0;
    f.persist = Arena(); f.scratch = Arena();
    f.persist.dry = f.scratch.dry = d;
    if (!d) {
      SDMI_CHECK(persist_bytes <= ws_bytes, "workspace too small: need " + std::to_string(persist_bytes) + " bytes, got " +
                                                std::to_string(ws_bytes));
      SDMI_CHECK(workspace != nullptr, "workspace is NULL");
      f.persist.base = (char*)workspace; f.persist.cap = (size_t)persist_bytes;
    }
    if (f.begin_pass((int64_t)4 << 20, /*with_gn=*/false)) return -1;     // split-K slabs only: no GroupNorm in this model
    float* x = f.P<float>((size_t)M * C);
    f16* ln = f.P<f16>((size_t)M * C);
    f16* q = f.P<f16>((size_t)M * C);
    f16* k = f.P<f16>((size_t)M * C);
    f16* vt = f.P<f16>((size_t)B * C * Lp);
    f16* ao = f.P<f16>((size_t)M * C);
    float* h1 = f.P<float>((size_t)M * I);
    f16* g = f.P<f16>((size_t)M * I);
    if (!d) {
      if (launch_embed_tokens(ids, tok_, pos_, x, M, L, C, cfg_.vocab_size, stream)) return -1;
      if (launch_layernorm(x, layers_[0].ln[0], layers_[0].ln[1], ln, M, C, 1e-5f, stream)) return -1;
      if (Lp != L) SDMI_HIP_OK(hipMemsetAsync(vt, 0, (size_t)B * C * Lp * sizeof(f16), stream));   // pad keys of V^T stay zero
    }
    for (int i = 0; i < cfg_.num_layers; ++i) {
      CLayer& Ly = layers_[i];
      {   // q | k | v = ln Wqkv^T + b, scattered per head (v transposed)
        IGemmParams p = f.dense(ln, M, C, Ly.wqkv, 3 * C, L);
        p.mode = EPI_HEADS; p.bias = Ly.bqkv; p.seg_dst[0] = q; p.seg_dst[1] = k; p.seg_dst[2] = vt;
        p.seg_kind[0] = 0; p.seg_kind[1] = 0; p.seg_kind[2] = 1;
        p.heads = H; p.dh = dh; p.ntok = L; p.ntok_pad = Lp; p.segC = C; p.splitk = 1;
        f.gemm(p);
      }
      if (!d && !f.rc) {
        AttnParams a;
        a.q = q; a.k = k; a.vt = vt; a.out = ao; a.BH = B * H; a.heads = H; a.nq = L; a.nkv = L; a.nkv_pad = Lp; a.d = dh;
        a.scale = scale; a.causal = 1;
        f.ok(launch_attention(a, stream));
      }
      {   // x += ao Wo^T + bo ; ln = LayerNorm2(x)
        IGemmParams p = f.dense(ao, M, C, Ly.wo, C, L);
        p.bias = Ly.bo; p.residual = x; p.ldr = C; p.out_f32 = x; p.ldo = C;
        p.ln_gamma = Ly.ln[2]; p.ln_beta = Ly.ln[3]; p.ln_out = ln; p.ln_eps = 1e-5f;
        f.gemm(p);
      }
      {   // h1 = ln W1^T + b1 ; g = quick_gelu(h1)
        IGemmParams p = f.dense(ln, M, C, Ly.w1, I, L);
        p.bias = Ly.b1; p.out_f32 = h1; p.ldo = I;
        f.gemm(p);
        if (!d && !f.rc) f.ok(launch_quick_gelu(h1, g, (int64_t)M * I, stream));
      }
      {   // x += g W2^T + b2 ; ln = LayerNorm1 of the next layer
        IGemmParams p = f.dense(g, M, I, Ly.w2, C, L);
        p.bias = Ly.b2; p.residual = x; p.ldr = C; p.out_f32 = x; p.ldo = C;
        if (i + 1 < cfg_.num_layers) {
          p.ln_gamma = layers_[i + 1].ln[0]; p.ln_beta = layers_[i + 1].ln[1]; p.ln_out = ln; p.ln_eps = 1e-5f;
        }
        f.gemm(p);
      }
    }
    if (!d && !f.rc) f.ok(launch_layernorm(x, fln_g_, fln_b_, nullptr, M, C, 1e-5f, stream, out));
    if (f.rc) return f.rc;
    if (d) {
      persist_bytes = (int64_t)round_up((int64_t)f.persist.peak, 4096) + 4096;
      if (bytes_needed) *bytes_needed = persist_bytes;
    } else {
      SDMI_CHECK(!f.persist.overflow, "internal: arena overflow");
    }
  }
  return 0;
}

}  // namespace sdmi
