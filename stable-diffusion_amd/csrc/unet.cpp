// UNet executor: the static launch graph of UNetModel.forward
// (ldm/modules/diffusionmodules/openaimodel.py:710-742) over the gfx950 kernels of this library.
//
// Data layout in HBM (per forward call, B = CFG batch):
//   * residual stream and skip stack: fp32 NHWC [B*H*W][C] (rounded nowhere; see DESIGN.md "precision")
//   * MFMA A operands (normalised activations, q/k/v^T, GEGLU output): fp16, written once by the producing
//     kernel and read once by the consuming GEMM
//   * weights: fp16 [N][K] (K = ky,kx,cin), packed once at load; norm/bias/time-embedding parameters fp32
//   * workspace = [persist | scratch]: every layer output lives in `persist` until the call ends (skip stack),
//     `scratch` is rewound after each layer so temporaries stay in the 256 MB Infinity Cache.
#include "unet.h"

extern "C" char** environ;       // (the launch tapes hash the SDMI_* knobs)
#include "prof.h"

#include <math.h>
#include <string.h>

#include <algorithm>

namespace sdmi {

// ------------------------------------------------------------------------------------------------------
// construction: mirror of UNetModel.__init__ (openaimodel.py:443-692) for the SD-v1 family
// ------------------------------------------------------------------------------------------------------
static bool in_list(const int* v, int n, int x) {
  for (int i = 0; i < n; ++i)
    if (v[i] == x) return true;
  return false;
}

void UNet::expect(const std::string& key, std::vector<int64_t> shape, WKind kind, void** dst, int row0, int ld,
                  void** dst2) {
  WeightSlot s;
  s.key = key; s.shape = std::move(shape); s.kind = kind; s.dst = dst; s.row0 = row0; s.ld = ld; s.dst2 = dst2;
  slot_index_[key] = (int)slots_.size();
  slots_.push_back(std::move(s));
}

thread_local Tape* g_tape_rec = nullptr;

namespace {
// FNV-1a over every SDMI_* entry of the environment: the knobs the library reads per call / per launch are part of a tape's identity
// (the tests flip them between two forwards of one shape)
uint64_t sdmi_env_hash() {
  uint64_t h = 1469598103934665603ull;
  for (char** e = environ; e && *e; ++e) {
    if (strncmp(*e, "SDMI_", 5) != 0) continue;
    for (const char* c = *e; *c; ++c) { h ^= (unsigned char)*c; h *= 1099511628211ull; }
    h ^= 0xff; h *= 1099511628211ull;
  }
  return h;
}
struct TapeRecGuard {         // recording ends with the scope, whatever path leaves it
  explicit TapeRecGuard(Tape* t) { g_tape_rec = t; }
  ~TapeRecGuard() { g_tape_rec = nullptr; }
};
}  // namespace

int UNet::build(const sdmi_unet_cfg& c) {
  cfg_ = c;
  SDMI_CHECK(c.n_levels >= 1 && c.n_levels <= 8 && c.n_attention_resolutions >= 0 && c.n_attention_resolutions <= 8,
             "bad level / attention_resolutions count");
  SDMI_CHECK(c.model_channels % 64 == 0, "model_channels must be a multiple of 64 on this path");
  SDMI_CHECK(c.context_dim % 64 == 0, "context_dim must be a multiple of 64 on this path");
  SDMI_CHECK(c.num_heads >= 1 && c.transformer_depth >= 1, "num_heads / transformer_depth");
  const int mc = c.model_channels;
  te_ = 4 * mc;
#ifdef SDMI_EXPERIMENTS      // knobs that LOWER the arithmetic or lost their A/B: not read by the product library
  if (const char* e = getenv("SDMI_PRECISE_1X1")) precise_1x1_ = atoi(e) != 0;
  if (const char* e = getenv("SDMI_PRECISE_KV")) precise_kv_ = atoi(e) != 0;
  if (const char* e = getenv("SDMI_PRECISE_LAST_RES")) precise_last_res_ = atoi(e) != 0;
  if (const char* e = getenv("SDMI_PRECISE_1X1_MAX_DS")) precise_1x1_max_ds_ = atoi(e);
  if (const char* e = getenv("SDMI_SIDE_STREAM")) side_stream_ = atoi(e) != 0;
#endif
  if (const char* e = getenv("SDMI_FUSE_GN_STATS")) fuse_gn_stats_ = atoi(e) != 0;
  if (const char* e = getenv("SDMI_LN_FOLD")) ln_fold_ = atoi(e) != 0;
  if (const char* e = getenv("SDMI_LN_FOLD_MIN_ROWS")) ln_fold_min_rows_ = atoi(e);

  int cur_ds = 1;               // downsample factor of the layer being added (ds below; the middle block sits at the deepest one)
  auto add_res = [&](const std::string& p, int cin, int cout) {
    Layer L; L.kind = L_RES; L.prefix = p; L.cin = cin; L.cout = cout;
    L.emb_off = emb_total_; emb_total_ += cout;
    L.p1x1 = precise_1x1_ && cur_ds < precise_1x1_max_ds_;
    return L;
  };
  auto add_attn = [&](const std::string& p, int ch) {
    Layer L; L.kind = L_ATTN; L.prefix = p; L.cin = ch; L.cout = ch; L.heads = c.num_heads; L.dh = ch / c.num_heads;
    L.attn_index = n_attn_++;
    L.p1x1 = precise_1x1_ && cur_ds < precise_1x1_max_ds_;
    return L;
  };

  {
    Layer L; L.kind = L_CONV_IN; L.prefix = "input_blocks.0.0"; L.cin = c.in_channels; L.cout = mc;
    input_blocks_.push_back({L});
  }
  std::vector<int> chans{mc};
  int ch = mc, ds = 1;
  for (int level = 0; level < c.n_levels; ++level) {
    const int mult = c.channel_mult[level];
    for (int r = 0; r < c.num_res_blocks; ++r) {
      const int n = (int)input_blocks_.size();
      std::vector<Layer> blk;
      cur_ds = ds;
      blk.push_back(add_res("input_blocks." + std::to_string(n) + ".0", ch, mult * mc));
      ch = mult * mc;
      if (in_list(c.attention_resolutions, c.n_attention_resolutions, ds))
        blk.push_back(add_attn("input_blocks." + std::to_string(n) + ".1", ch));
      input_blocks_.push_back(blk);
      chans.push_back(ch);
    }
    if (level != c.n_levels - 1) {
      const int n = (int)input_blocks_.size();
      Layer L; L.kind = L_DOWN; L.prefix = "input_blocks." + std::to_string(n) + ".0"; L.cin = ch; L.cout = ch;
      input_blocks_.push_back({L});
      chans.push_back(ch);
      ds *= 2;
    }
  }
  cur_ds = ds;
  middle_.push_back(add_res("middle_block.0", ch, ch));
  middle_.push_back(add_attn("middle_block.1", ch));
  middle_.push_back(add_res("middle_block.2", ch, ch));
  for (int level = c.n_levels - 1; level >= 0; --level) {
    const int mult = c.channel_mult[level];
    for (int i = 0; i <= c.num_res_blocks; ++i) {
      const int ich = chans.back(); chans.pop_back();
      const int n = (int)output_blocks_.size();
      std::vector<Layer> blk;
      cur_ds = ds;
      blk.push_back(add_res("output_blocks." + std::to_string(n) + ".0", ch + ich, mc * mult));
      ch = mc * mult;
      if (in_list(c.attention_resolutions, c.n_attention_resolutions, ds))
        blk.push_back(add_attn("output_blocks." + std::to_string(n) + "." + std::to_string(blk.size()), ch));
      if (level && i == c.num_res_blocks) {
        Layer L; L.kind = L_UP; L.prefix = "output_blocks." + std::to_string(n) + "." + std::to_string(blk.size());
        L.cin = ch; L.cout = ch;
        blk.push_back(L);
        ds /= 2;
      }
      output_blocks_.push_back(blk);
    }
  }

  // the last ResBlock (output_blocks.<last>.0: its two 3x3 convs are the largest single contributors to the eps error)
  if (precise_last_res_ && !output_blocks_.empty() && output_blocks_.back()[0].kind == L_RES && output_blocks_.back()[0].cin % 64 == 0 &&
      output_blocks_.back()[0].cout % 64 == 0)
    output_blocks_.back()[0].precise3 = true;

  // ---- expected state_dict entries and where each one is packed to -------------------
  const int64_t TE = te_;
  expect("time_embed.0.weight", {TE, mc}, W_F32, (void**)&te_w0_);
  expect("time_embed.0.bias", {TE}, W_F32, (void**)&te_b0_);
  expect("time_embed.2.weight", {TE, TE}, W_F32, (void**)&te_w2_);
  expect("time_embed.2.bias", {TE}, W_F32, (void**)&te_b2_);
  auto visit = [&](Layer& L) {
    const std::string& p = L.prefix;
    const int64_t ci = L.cin, co = L.cout;
    switch (L.kind) {
      case L_CONV_IN:
        expect(p + ".weight", {co, ci, 3, 3}, W_F32, (void**)&L.w32[0]);
        expect(p + ".bias", {co}, W_F32, (void**)&L.f32[0]);
        break;
      case L_RES:
        expect(p + ".in_layers.0.weight", {ci}, W_F32, (void**)&L.f32[0]);
        expect(p + ".in_layers.0.bias", {ci}, W_F32, (void**)&L.f32[1]);
        expect(p + ".in_layers.2.weight", {co, ci, 3, 3}, L.precise3 ? W_CONV_SPLIT3 : W_CONV, (void**)&L.w16[0]);
        expect(p + ".in_layers.2.bias", {co}, W_F32, (void**)&L.f32[2]);
        expect(p + ".emb_layers.1.weight", {co, TE}, W_F32_ROWS, (void**)&emb_w_, L.emb_off, te_);
        expect(p + ".emb_layers.1.bias", {co}, W_F32_ROWS, (void**)&emb_b_, L.emb_off, 1);
        expect(p + ".out_layers.0.weight", {co}, W_F32, (void**)&L.f32[3]);
        expect(p + ".out_layers.0.bias", {co}, W_F32, (void**)&L.f32[4]);
        expect(p + ".out_layers.3.weight", {co, co, 3, 3}, L.precise3 ? W_CONV_SPLIT3 : W_CONV, (void**)&L.w16[1]);
        expect(p + ".out_layers.3.bias", {co}, W_F32, (void**)&L.f32[5]);
        if (ci != co) {
          expect(p + ".skip_connection.weight", {co, ci, 1, 1}, L.p1x1 ? W_SPLIT3 : W_CONV, (void**)&L.w16[2]);
          expect(p + ".skip_connection.bias", {co}, W_F32, (void**)&L.f32[6]);
        }
        break;
      case L_ATTN: {
        const int64_t C = ci, CD = cfg_.context_dim;
        expect(p + ".norm.weight", {C}, W_F32, (void**)&L.f32[0]);
        expect(p + ".norm.bias", {C}, W_F32, (void**)&L.f32[1]);
        expect(p + ".proj_in.weight", {C, C, 1, 1}, L.p1x1 ? W_SPLIT3 : W_CONV, (void**)&L.w16[0]);
        expect(p + ".proj_in.bias", {C}, W_F32, (void**)&L.f32[2]);
        expect(p + ".proj_out.weight", {C, C, 1, 1}, L.p1x1 ? W_SPLIT3 : W_CONV, (void**)&L.w16[1]);
        expect(p + ".proj_out.bias", {C}, W_F32, (void**)&L.f32[3]);
        L.tb.resize(cfg_.transformer_depth);
        for (int d = 0; d < cfg_.transformer_depth; ++d) {
          TBlock& T = L.tb[d];
          const std::string t = p + ".transformer_blocks." + std::to_string(d);
          expect(t + ".attn1.to_q.weight", {C, C}, W_ROWS16, (void**)&T.wqkv, 0, (int)C);
          expect(t + ".attn1.to_k.weight", {C, C}, W_ROWS16, (void**)&T.wqkv, (int)C, (int)C);
          expect(t + ".attn1.to_v.weight", {C, C}, W_ROWS16, (void**)&T.wqkv, 2 * (int)C, (int)C);
          expect(t + ".attn1.to_out.0.weight", {C, C}, W_ROWS16, (void**)&T.wo1, 0, (int)C);
          expect(t + ".attn1.to_out.0.bias", {C}, W_F32, (void**)&T.bo1);
          expect(t + ".attn2.to_q.weight", {C, C}, W_ROWS16, (void**)&T.wq2, 0, (int)C);
          // round 6: the context K / V projections as 3-pass split-fp16 (k_hi w_hi + k_lo w_hi + k_hi w_lo, one K-concatenated GEMM): the
          // context is the one operand of the call with channel outliers by construction (CLIP's last_hidden_state has channels at
          // |x| ~ 30) and its fp16 rounding was the error class that grew most (11x) on the outlier goldens (tools/precision_emul.py);
          // computed once per prompt and cached for all 51 calls, so the extra passes cost nothing per UNet call
          expect(t + ".attn2.to_k.weight", {C, CD}, precise_kv_ ? W_SPLIT3_ROWS : W_ROWS16, (void**)&T.wkv2, 0, (int)CD);
          expect(t + ".attn2.to_v.weight", {C, CD}, precise_kv_ ? W_SPLIT3_ROWS : W_ROWS16, (void**)&T.wkv2, (int)C, (int)CD);
          expect(t + ".attn2.to_out.0.weight", {C, C}, W_ROWS16, (void**)&T.wo2, 0, (int)C);
          expect(t + ".attn2.to_out.0.bias", {C}, W_F32, (void**)&T.bo2);
          expect(t + ".ff.net.0.proj.weight", {8 * C, C}, W_GEGLU_W, (void**)&T.wgg);
          expect(t + ".ff.net.0.proj.bias", {8 * C}, W_GEGLU_B, (void**)&T.bgg);
          expect(t + ".ff.net.2.weight", {C, 4 * C}, W_ROWS16, (void**)&T.wff2, 0, 4 * (int)C);
          expect(t + ".ff.net.2.bias", {C}, W_F32, (void**)&T.bff2);
          expect(t + ".norm1.weight", {C}, W_F32, (void**)&T.ln[0]);
          expect(t + ".norm1.bias", {C}, W_F32, (void**)&T.ln[1]);
          expect(t + ".norm2.weight", {C}, W_F32, (void**)&T.ln[2]);
          expect(t + ".norm2.bias", {C}, W_F32, (void**)&T.ln[3]);
          expect(t + ".norm3.weight", {C}, W_F32, (void**)&T.ln[4]);
          expect(t + ".norm3.bias", {C}, W_F32, (void**)&T.ln[5]);
        }
        break;
      }
      case L_DOWN:
        expect(p + ".op.weight", {co, ci, 3, 3}, W_CONV, (void**)&L.w16[0]);
        expect(p + ".op.bias", {co}, W_F32, (void**)&L.f32[0]);
        break;
      case L_UP:
        expect(p + ".conv.weight", {co, ci, 3, 3}, W_CONV, (void**)&L.w16[0]);
        expect(p + ".conv.bias", {co}, W_F32, (void**)&L.f32[0]);
        break;
    }
  };
  // NOTE: slots hold pointers into the Layer objects, so the containers must not reallocate after this point.
  for (auto& blk : input_blocks_) for (auto& L : blk) visit(L);
  for (auto& L : middle_) visit(L);
  for (auto& blk : output_blocks_) for (auto& L : blk) visit(L);
  expect("out.0.weight", {mc}, W_F32, (void**)&out_gamma_);
  expect("out.0.bias", {mc}, W_F32, (void**)&out_beta_);
  expect("out.2.weight", {c.out_channels, mc, 3, 3}, W_CONV_OUT, (void**)&out_w_);
  expect("out.2.bias", {c.out_channels}, W_F32, (void**)&out_b_);
  return 0;
}

int DevStage::acquire(const float* ptr, int64_t numel, hipStream_t stream) {
  dptr = ptr;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, ptr);
  const bool on_device = (e == hipSuccess) && (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged);
  if (e != hipSuccess) (void)hipGetLastError();
  if (!on_device) {
    SDMI_HIP_OK(hipMalloc((void**)&staged, numel * sizeof(float)));
    SDMI_HIP_OK(hipMemcpyAsync(staged, ptr, numel * sizeof(float), hipMemcpyHostToDevice, stream));
    dptr = staged;
  }
  return 0;
}
int DevStage::release(hipStream_t stream) {
  if (staged) {
    SDMI_HIP_OK(hipStreamSynchronize(stream));
    (void)hipFree(staged);
    staged = nullptr;
  }
  return 0;
}

UNet::~UNet() {
  if (side_) {
    (void)hipStreamSynchronize(side_);
    for (auto& e : side_ev_) if (e) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(side_);
  }
  for (void* p : owned_) (void)hipFree(p);
  if (emb_tab_) (void)hipFree(emb_tab_);
  if (emb_tab_tdev_) (void)hipFree(emb_tab_tdev_);
  auto drop_ctx = [](Layer& L) {            // cross-attention K / V^T caches (ensure_ctx_cache)
    for (auto& T : L.tb) {
      if (T.ck) (void)hipFree(T.ck);
      if (T.cvt) (void)hipFree(T.cvt);
    }
  };
  for (auto& blk : input_blocks_) for (auto& L : blk) drop_ctx(L);
  for (auto& L : middle_) drop_ctx(L);
  for (auto& blk : output_blocks_) for (auto& L : blk) drop_ctx(L);
}

int UNet::dev_alloc(void** dst, size_t bytes) {
  if (*dst) return 0;
  SDMI_HIP_OK(hipMalloc(dst, bytes));
  owned_.push_back(*dst);
  return 0;
}

// bytes of the packed device buffer a slot writes into (several slots may share one buffer: q|k|v rows, emb_layers rows)
size_t UNet::slot_bytes(const WeightSlot& s) const {
  size_t numel = 1;
  for (int64_t d : s.shape) numel *= (size_t)d;
  switch (s.kind) {
    case W_F32: case W_CONV_OUT: case W_GEGLU_B: return numel * sizeof(float);
    case W_F32_ROWS: return (size_t)emb_total_ * s.ld * sizeof(float);
    case W_CONV: case W_GEGLU_W: return numel * sizeof(f16);
    case W_SPLIT3: case W_CONV_SPLIT3: return 3 * numel * sizeof(f16);
    case W_SPLIT3_ROWS: return 2 * 3 * numel * sizeof(f16);      // (to_k | to_v share one [2C][3 K] buffer)
    case W_ROWS16: {
      size_t total_rows = (size_t)s.shape[0];
      if (s.key.find(".attn1.to_") != std::string::npos && s.key.find("to_out") == std::string::npos) total_rows *= 3;
      if (s.key.find(".attn2.to_k") != std::string::npos || s.key.find(".attn2.to_v") != std::string::npos) total_rows *= 2;
      return total_rows * s.ld * sizeof(f16);
    }
  }
  return 0;
}

int UNet::set_weight(const char* key, const float* ptr, const int64_t* shape, int ndim, hipStream_t stream) {
  auto it = slot_index_.find(key);
  if (it == slot_index_.end()) return fail(std::string("unexpected weight key: ") + key);
  WeightSlot& s = slots_[it->second];
  SDMI_CHECK((int)s.shape.size() == ndim, std::string("rank mismatch for ") + key);
  int64_t numel = 1;
  for (int i = 0; i < ndim; ++i) {
    SDMI_CHECK(shape[i] == s.shape[i], std::string("shape mismatch for ") + key);
    numel *= shape[i];
  }
  DevStage st;
  if (st.acquire(ptr, numel, stream)) return -1;
  const float* dptr = st.dptr;
  int rc = 0;
  switch (s.kind) {
    case W_F32:
      rc = dev_alloc(s.dst, slot_bytes(s));
      if (!rc) SDMI_HIP_OK(hipMemcpyAsync(*s.dst, dptr, numel * sizeof(float), hipMemcpyDeviceToDevice, stream));
      break;
    case W_F32_ROWS: {   // rows of a concatenated fp32 matrix (the 22 emb_layers)
      rc = dev_alloc(s.dst, slot_bytes(s));
      if (!rc)
        SDMI_HIP_OK(hipMemcpyAsync((float*)*s.dst + (size_t)s.row0 * s.ld, dptr, numel * sizeof(float),
                                   hipMemcpyDeviceToDevice, stream));
      break;
    }
    case W_CONV:
      rc = dev_alloc(s.dst, slot_bytes(s));
      if (!rc) rc = launch_pack_conv_weight(dptr, (f16*)*s.dst, (int)shape[0], (int)shape[1], (int)shape[2], (int)shape[3], stream);
      break;
    case W_SPLIT3:
      rc = dev_alloc(s.dst, slot_bytes(s));
      if (!rc) rc = launch_pack_split3(dptr, (f16*)*s.dst, (int)shape[0], (int)shape[1], stream);
      break;
    case W_CONV_SPLIT3:
      rc = dev_alloc(s.dst, slot_bytes(s));
      if (!rc) rc = launch_pack_conv_split3(dptr, (f16*)*s.dst, (int)shape[0], (int)shape[1], (int)shape[2], (int)shape[3], stream);
      break;
    case W_SPLIT3_ROWS:  // rows [row0, row0 + rows) of a split-fp16 [*][3 ld] matrix
      rc = dev_alloc(s.dst, slot_bytes(s));
      if (!rc) rc = launch_pack_split3(dptr, (f16*)*s.dst + (size_t)s.row0 * 3 * s.ld, (int)shape[0], (int)shape[1], stream);
      break;
    case W_CONV_OUT:
      rc = dev_alloc(s.dst, slot_bytes(s));
      if (!rc) rc = launch_pack_conv_out(dptr, (float*)*s.dst, (int)shape[0], (int)shape[1], stream);
      break;
    case W_ROWS16: {     // rows [row0, row0+rows) of an fp16 [*, ld] matrix (q|k|v and k|v concatenations)
      const int rows = (int)shape[0];
      rc = dev_alloc(s.dst, slot_bytes(s));
      if (!rc) rc = launch_pack_rows(dptr, (f16*)*s.dst, rows, (int)shape[1], s.row0, s.ld, stream);
      break;
    }
    case W_GEGLU_W: {
      // weight and bias arrive separately; the weight packer does not need the bias (and vice versa)
      rc = dev_alloc(s.dst, slot_bytes(s));
      if (!rc) rc = launch_pack_geglu(dptr, nullptr, (f16*)*s.dst, nullptr, (int)shape[0], (int)shape[1], stream);
      break;
    }
    case W_GEGLU_B: {
      rc = dev_alloc(s.dst, slot_bytes(s));
      // permute the bias with the same 32-row interleave: reuse the packer with K = 1 on a [N][1] "matrix"
      if (!rc) {
        f16* tmp = nullptr;
        SDMI_HIP_OK(hipMalloc((void**)&tmp, numel * sizeof(f16)));
        rc = launch_pack_geglu(dptr, dptr, tmp, (float*)*s.dst, (int)shape[0], 1, stream);
        SDMI_HIP_OK(hipStreamSynchronize(stream));
        (void)hipFree(tmp);
      }
      break;
    }
  }
  if (st.release(stream)) return -1;
  if (rc) return rc;
  s.set = true;
  ++weights_gen_;            // (recorded launch tapes point into the packed buffers / were planned for them)
  finalized_ = false;
  ctx_valid_ = false;      // cached cross-attention K/V were computed with the previous to_k / to_v weights
  drop_timestep_table();   // ... and the timestep table with the previous time_embed / emb_layers weights
  return 0;
}

int UNet::finalize() {
  for (auto& s : slots_)
    if (!s.set) return fail("weight not set: " + s.key);
  if (!zero_) {
    SDMI_HIP_OK(hipMalloc((void**)&zero_, 4096));
    owned_.push_back(zero_);
    SDMI_HIP_OK(hipMemset(zero_, 0, 4096));
  }
  if (reserve_ctx_cache(8, 77)) return -1;       // default K/V capacity (a no-op once reserved)
  // column terms of the GEMMs that fold a LayerNorm of their input rows (derived from the packed weights: not part of the blob)
  {
    SDMI_HIP_OK(hipDeviceSynchronize());          // (the packing kernels ran on the caller's streams; finalize is off the hot path)
    auto each = [&](Layer& L) -> int {
      if (L.kind != L_ATTN) return 0;
      const int C = L.cin;
      for (auto& T : L.tb) {
        const int n_[3] = {3 * C, C, 8 * C};
        const f16* w_[3] = {T.wqkv, T.wq2, T.wgg};
        const float* b_[3] = {nullptr, nullptr, T.bgg};
        for (int i = 0; i < 3; ++i) {
          if (dev_alloc((void**)&T.lnf[2 * i], (size_t)n_[i] * sizeof(float)) || dev_alloc((void**)&T.lnf[2 * i + 1], (size_t)n_[i] * sizeof(float)))
            return -1;
          if (launch_ln_fold_prep(w_[i], n_[i], C, C, T.ln[2 * i], T.ln[2 * i + 1], b_[i], T.lnf[2 * i], T.lnf[2 * i + 1], nullptr)) return -1;
        }
        if (ff_tail_supported(C, 32, 32)) {            // (geometry only: rows are checked per call)
          if (dev_alloc((void**)&T.lnf_csd, (size_t)16 * C * sizeof(float))) return -1;
          for (int h = 0; h < 4; ++h) {                 // (default stream, behind the prep kernels above)
            SDMI_HIP_OK(hipMemcpyAsync(T.lnf_csd + (size_t)h * 4 * C, T.lnf[4] + (size_t)h * 2 * C, (size_t)2 * C * sizeof(float), hipMemcpyDeviceToDevice, nullptr));
            SDMI_HIP_OK(hipMemcpyAsync(T.lnf_csd + (size_t)h * 4 * C + 2 * C, T.lnf[5] + (size_t)h * 2 * C, (size_t)2 * C * sizeof(float), hipMemcpyDeviceToDevice, nullptr));
          }
        }
      }
      return 0;
    };
    for (auto& blk : input_blocks_) for (auto& L : blk) if (each(L)) return -1;
    for (auto& L : middle_) if (each(L)) return -1;
    for (auto& blk : output_blocks_) for (auto& L : blk) if (each(L)) return -1;
    SDMI_HIP_OK(hipDeviceSynchronize());
  }
  ++weights_gen_;
  finalized_ = true;
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// packed-weight blob: the device buffers exactly as set_weight() leaves them, one after another
// in slot order, behind a header that pins the configuration -- loading it skips the fp32 checkpoint and the repack
// ------------------------------------------------------------------------------------------------------
namespace {
struct PackedHeader {
  char magic[8];               // "SDMIPK01"
  int32_t abi, precise_1x1, n_buffers, reserved;
  sdmi_unet_cfg cfg;
  int64_t total_bytes;
};
constexpr int64_t PK_ALIGN = 256;
}  // namespace

int UNet::packed_layout(std::vector<std::pair<void**, size_t>>* bufs, int64_t* total) const {
  std::vector<void**> seen;
  int64_t off = (int64_t)round_up((int64_t)sizeof(PackedHeader), PK_ALIGN);
  for (const auto& s : slots_) {
    if (std::find(seen.begin(), seen.end(), s.dst) != seen.end()) continue;
    seen.push_back(s.dst);
    const size_t b = slot_bytes(s);
    if (bufs) bufs->push_back({s.dst, b});
    off += (int64_t)round_up((int64_t)b, PK_ALIGN);
  }
  *total = off;
  return 0;
}

int UNet::export_packed(void* host_buf, int64_t bytes, hipStream_t stream) {
  SDMI_CHECK(finalized_, "export needs a finalized handle");
  std::vector<std::pair<void**, size_t>> bufs;
  int64_t total = 0;
  packed_layout(&bufs, &total);
  SDMI_CHECK(host_buf && bytes >= total, "packed buffer too small: need " + std::to_string(total) + " bytes");
  PackedHeader h{};
  memcpy(h.magic, "SDMIPK01", 8);
  h.abi = SDMI_ABI_VERSION; h.precise_1x1 = precise_1x1_ ? 1 : 0; h.n_buffers = (int32_t)bufs.size(); h.cfg = cfg_;
  h.reserved = (precise_kv_ ? 1 : 0) | (precise_last_res_ ? 2 : 0) | (precise_1x1_max_ds_ << 8);      // (ABI 17: the precision allocation)
  h.total_bytes = total;
  memcpy(host_buf, &h, sizeof(h));
  int64_t off = (int64_t)round_up((int64_t)sizeof(PackedHeader), PK_ALIGN);
  for (auto& b : bufs) {
    SDMI_HIP_OK(hipMemcpyAsync((char*)host_buf + off, *b.first, b.second, hipMemcpyDeviceToHost, stream));
    off += (int64_t)round_up((int64_t)b.second, PK_ALIGN);
  }
  SDMI_HIP_OK(hipStreamSynchronize(stream));
  return 0;
}

int UNet::import_packed(const void* host_buf, int64_t bytes, hipStream_t stream) {
  SDMI_CHECK(host_buf && bytes >= (int64_t)sizeof(PackedHeader), "packed blob truncated");
  PackedHeader h;
  memcpy(&h, host_buf, sizeof(h));
  SDMI_CHECK(memcmp(h.magic, "SDMIPK01", 8) == 0, "not a libsdmi packed-weight blob");
  SDMI_CHECK(h.abi == SDMI_ABI_VERSION, "packed blob was written by a different ABI version: repack it");
  SDMI_CHECK(memcmp(&h.cfg, &cfg_, sizeof(cfg_)) == 0, "packed blob was written for a different UNet configuration");
  SDMI_CHECK((h.precise_1x1 != 0) == precise_1x1_, "packed blob was written with a different SDMI_PRECISE_1X1 setting");
  SDMI_CHECK(h.reserved == ((precise_kv_ ? 1 : 0) | (precise_last_res_ ? 2 : 0) | (precise_1x1_max_ds_ << 8)),
             "packed blob was written with a different precision allocation (SDMI_PRECISE_KV / _LAST_RES / _1X1_MAX_DS)");
  std::vector<std::pair<void**, size_t>> bufs;
  int64_t total = 0;
  packed_layout(&bufs, &total);
  SDMI_CHECK(h.n_buffers == (int32_t)bufs.size() && h.total_bytes == total && bytes >= total, "packed blob truncated or inconsistent");
  int64_t off = (int64_t)round_up((int64_t)sizeof(PackedHeader), PK_ALIGN);
  for (auto& b : bufs) {
    if (dev_alloc(b.first, b.second)) return -1;
    SDMI_HIP_OK(hipMemcpyAsync(*b.first, (const char*)host_buf + off, b.second, hipMemcpyHostToDevice, stream));
    off += (int64_t)round_up((int64_t)b.second, PK_ALIGN);
  }
  SDMI_HIP_OK(hipStreamSynchronize(stream));
  for (auto& s : slots_) s.set = true;
  ctx_valid_ = false;
  drop_timestep_table();
  return finalize();
}

// ------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------
struct Fwd : FwdBase {
  UNet* u; int Lctx;
  bool ln_fold_on = false;      // this call folds LayerNorms into their consuming GEMMs (UNet::ln_fold_; read per call: A/B knobs)
  bool ff_tail_on = false;      // ... and runs SpatialTransformer tails as row-strip chain launches (UNet::ff_tail_; SDMI_FF_TAIL, read per call)
  bool st_head_on = false;      // ... and SpatialTransformer heads (UNet::st_head_; SDMI_ST_HEAD, read per call)
  bool st_mid_on = false;       // ... and the out-projection of attn1 with attn2's to_q (SDMI_ST_MID)
  bool st_tail_on = false;      // ... and the out-projection of attn2 in front of the tail's chain launch (SDMI_ST_TAIL)
  bool st_mid_ctx_on = false;   // ... and the cross-attention inside the st_mid launch (SDMI_ST_MID_CTX)
  bool gn_conv_on = false;      // ResBlock GroupNorm + SiLU + conv3x3 as one launch where a workgroup can own all output columns (gnconv.hip; SDMI_GN_CONV)
  // cross-attention with the to_q projection inside the kernel (attn_ctx.hip), SDMI_ATTN_CTX_FUSED=1.  Default off: same-box A/B,
  // round 3 (profiles/experiments_r03.txt): 5.98 vs 5.88 ms per UNet call -- -3.8 us per launch at d = 40, +1.4 at d = 80, +13 at d = 160
  bool fuse_ctx_q = false;
  int fuse_ctx_maxd = 160;      // ... for head dims up to this (SDMI_ATTN_CTX_MAXD)
  // SpatialTransformer norm applied inside the proj_in GEMM (gemm_split16_gn_kernel), SDMI_GN_PROJ_FOLD=1.  Default off: bit-identical,
  // 15 launches fewer, but 7.05 vs 6.86 ms per UNet call same box (profiles/experiments_r03.txt): -4.7 us per site at 2048 rows, a tie at
  // 8192, +7 us at 512 rows / 1280 channels (20 k-tiles, each one drained register-load round trip)
  bool gn_proj_fold = false;
  float* emb_all = nullptr;     // [B][emb_total] (emb_ld = emb_total), or one row of the timestep table shared by every sample (emb_ld = 0)
  int emb_ld = 0;
  const f16* ctx16 = nullptr;   // [B*L][context_dim], null when the cached K/V are used
  const f16* ctx16_lo = nullptr; // ... its split-fp16 low half fp16(ctx - fp16(ctx)) (precise K / V projections)

  // conv3x3(SiLU(GroupNorm32(cat(x0, x1)))) with the normalisation folded into the convolution's halo staging
  // (openaimodel.py:201-204,225-231): statistics first (from the producers' epilogues when the plan has them), then ONE launch
  IGemmParams conv3_gn(const Act& x0, const Act* x1, const float* gamma, const float* beta, const f16* w, int N, f16* raw_hi,
                       f16* raw_lo) {
    long long* acc = groupnorm(x0, x1, gamma, beta, 1e-5f, 1, nullptr, nullptr, nullptr, nullptr, nullptr, /*stats_only=*/true);
    const int Cin = x0.C + (x1 ? x1->C : 0);
    IGemmParams p = conv3(nullptr, Cin, x0.H, x0.W, x0.H, x0.W, 1, 0, w, N);
    p.xf0 = x0.p; p.c0 = x0.C; p.xf1 = x1 ? x1->p : nullptr; p.c1 = x1 ? x1->C : 0; p.lda0 = Cin;
    p.gn_in_acc = acc; p.gn_in_gamma = gamma; p.gn_in_beta = beta; p.gn_in_eps = 1e-5f; p.gn_in_silu = 1;
    p.raw_hi = raw_hi; p.raw_lo = raw_lo;
    // (scratch for the launcher's two-launch alternative: per shape, the tuning table decides between the folding kernel and
    // GroupNorm-apply + LDS-DMA conv -- same operand bits either way)
    p.gn_scratch = S<f16>((size_t)B * x0.H * x0.W * Cin);
    return p;
  }

  Act res_block(Layer& L, const Act& x0, const Act* x1) {
    const int H = x0.H, W = x0.W, M = B * H * W;
    const int Cin = x0.C + (x1 ? x1->C : 0), Cout = L.cout;
    if (Cin != L.cin) { ok(fail("res block channel mismatch at " + L.prefix)); }
    if (Cin == Cout && x1) ok(fail("identity skip with a concatenated input at " + L.prefix));
    const size_t mark = scratch.off;
    // GroupNorm + SiLU folded into the halo staging of the 3x3 convs where the geometry allows (every level of SD v1)
    // (SDMI_FUSE_GN_WHICH: bit 0 in_layers, bit 1 out_layers; SDMI_FUSE_GN_W: only at this width -- bisecting knobs)
#ifdef SDMI_EXPERIMENTS
    static const int fold_which = getenv("SDMI_FUSE_GN_WHICH") ? atoi(getenv("SDMI_FUSE_GN_WHICH")) : 3;
    static const int fold_w = getenv("SDMI_FUSE_GN_W") ? atoi(getenv("SDMI_FUSE_GN_W")) : 0;
#else
    constexpr int fold_which = 3, fold_w = 0;     // (gn_fold_conv_supported() is false in the product build: no fold site)
#endif
    const bool fold_here = fold_w == 0 || fold_w == W;
    const bool fold1 = !L.precise3 && (fold_which & 1) && fold_here && gn_fold_conv_supported(B, H, W, x0.C, x1 ? x1->C : 0, Cout);
    const bool fold2 = !L.precise3 && (fold_which & 2) && fold_here && gn_fold_conv_supported(B, H, W, Cout, 0, Cout);
    f16* raw = (Cin != Cout) ? S<f16>((size_t)M * Cin) : nullptr;
    const bool p1 = L.p1x1;              // this layer's skip convolution as 3-pass split-fp16
    const bool p3 = L.precise3;          // ... and its two 3x3 convs (the last ResBlock): operands [hi | lo | hi] against packed [w_hi | w_hi | w_lo]
    f16* raw_lo = (Cin != Cout && p1) ? S<f16>((size_t)M * Cin) : nullptr;
    auto split3 = [&](IGemmParams& q, const f16* hi, const f16* lo, int C) {
      q.a0 = hi; q.c0 = C; q.lda0 = C; q.a1 = lo; q.c1 = C; q.lda1 = C; q.a2 = hi; q.c2 = C; q.lda2 = C; q.K = 27 * C; q.k_alg = 9 * C;
    };
    float* h = S<float>((size_t)M * Cout);
    Act out = make_act(P<float>((size_t)M * Cout), Cout, H, W, true);
    Act hact = make_act(h, Cout, H, W, true);
    f16* a2 = nullptr; int gn2_applied = 0;
    // in_layers / out_layers as ONE launch each (gnconv.hip: 32 pixels x all 320 output channels per workgroup, the halo normalised once):
    // the 64 x 64 level of SD v1.  (in_layers only without a skip convolution: that one reads raw fp16 copies the GroupNorm launch writes.)
    const bool gc1 = gn_conv_on && !L.precise3 && !fold1 && Cin == Cout && gn_conv3_supported(B, H, W, x0.C, x1 ? x1->C : 0, Cout);
    const bool gc2 = gn_conv_on && !L.precise3 && !fold2 && gn_conv3_supported(B, H, W, Cout, 0, Cout);
    auto gn_conv = [&](const Act& a0, const Act* a1, const float* gamma, const float* beta, const f16* w, IGemmParams& e) {
      GnConvParams g;
      g.gn_acc = groupnorm(a0, a1, gamma, beta, 1e-5f, 1, nullptr, nullptr, nullptr, nullptr, nullptr, /*stats_only=*/true);
      g.x0 = a0.p; g.c0 = a0.C; g.x1 = a1 ? a1->p : nullptr; g.c1 = a1 ? a1->C : 0;
      g.gn_gamma = gamma; g.gn_beta = beta; g.gn_eps = 1e-5f; g.w = w; g.epi = e;
      if (!dry && !rc) ok(launch_gn_conv3(g, s));
    };
    if (gc1) {
      IGemmParams p = conv3(nullptr, Cin, H, W, H, W, 1, 0, L.w16[0], Cout);
      p.bias = L.f32[2]; p.rowvec = emb_all + L.emb_off; p.ld_rowvec = emb_ld;
      p.out_f32 = h; p.ldo = Cout;
      attach_gn_targets(p, hact);
      gn_conv(x0, x1, L.f32[0], L.f32[1], L.w16[0], p);
    } else {
      IGemmParams p;
      if (fold1) {
        p = conv3_gn(x0, x1, L.f32[0], L.f32[1], L.w16[0], Cout, raw, raw_lo);    // (+ the skip conv's raw hi | lo operand)
      } else {
        f16* a = S<f16>((size_t)M * Cin);
        f16* a_lo = p3 ? S<f16>((size_t)M * Cin) : nullptr;
        groupnorm(x0, x1, L.f32[0], L.f32[1], 1e-5f, 1, a, nullptr, raw, a_lo, raw_lo);
        p = conv3(a, Cin, H, W, H, W, 1, 0, L.w16[0], Cout);
        if (p3) split3(p, a, a_lo, Cin);
      }
      if (Cin != Cout && !fold1) fork_side();
      p.bias = L.f32[2]; p.rowvec = emb_all + L.emb_off; p.ld_rowvec = emb_ld;
      p.out_f32 = h; p.ldo = Cout;
      attach_gn_targets(p, hact);          // statistics of out_layers' GroupNorm come out of this epilogue
      if (!fold2 && !gc2 && !p3) {
        // ... or, where this conv ends up split along K (8x8, 16x16, the concat blocks of 32x32), GroupNorm + SiLU are applied by
        // its split-K reduction: a2 = the conv2 operand comes straight out of it (IGemmParams::pgn_*; gn2_applied says so)
        a2 = S<f16>((size_t)M * Cout);
        p.pgn_gamma = L.f32[3]; p.pgn_beta = L.f32[4]; p.pgn_eps = 1e-5f; p.pgn_silu = 1; p.pgn_out = a2; p.pgn_applied = &gn2_applied;
      }
      gemm(p);
    }
    const float* residual = x0.p;
    if (Cin != Cout) {       // skip_connection: needs only the raw fp16 copies (written by GroupNorm 1 / by conv1's staging)
      IGemmParams p = dense1x1(raw, raw_lo, M, Cin, L.w16[2], Cout, H * W, p1);
      p.bias = L.f32[6]; p.out_f32 = out.p; p.ldo = Cout;
      if (fold1) gemm(p); else gemm_side(p);
      residual = out.p;
    }
    if (gc2) {
      IGemmParams p = conv3(nullptr, Cout, H, W, H, W, 1, 0, L.w16[1], Cout);
      if (Cin != Cout && !fold1) join_side();
      p.bias = L.f32[5]; p.residual = residual; p.ldr = Cout; p.out_f32 = out.p; p.ldo = Cout;
      attach_gn_targets(p, out);
      attach_f16_copy(p, out);
      gn_conv(hact, nullptr, L.f32[3], L.f32[4], L.w16[1], p);
    } else {
      IGemmParams p;
      if (fold2) {
        p = conv3_gn(hact, nullptr, L.f32[3], L.f32[4], L.w16[1], Cout, nullptr, nullptr);
      } else {
        f16* a2_lo = nullptr;
        if (p3) { a2 = S<f16>((size_t)M * Cout); a2_lo = S<f16>((size_t)M * Cout); }
        groupnorm(hact, nullptr, L.f32[3], L.f32[4], 1e-5f, 1, a2, nullptr, nullptr, a2_lo, nullptr, false, gn2_applied != 0);
        p = conv3(a2, Cout, H, W, H, W, 1, 0, L.w16[1], Cout);
        if (p3) split3(p, a2, a2_lo, Cout);
      }
      if (Cin != Cout && !fold1) join_side();
      p.bias = L.f32[5]; p.residual = residual; p.ldr = Cout; p.out_f32 = out.p; p.ldo = Cout;
      attach_gn_targets(p, out);           // ... and those of the GroupNorm(s) that read this block's output
      attach_f16_copy(p, out);             // ... and the fp16 copy a Downsample / Upsample behind this block wants
      gemm(p);
    }
    scratch.off = mark;
    return out;
  }

  // K / V^T of the cross-attention for one transformer block (depends on the context only)
  void context_kv(Layer& L, int d) {
    TBlock& T = L.tb[d];
    const int C = L.cin, Lp = (int)round_up(Lctx, 8);
    const int CD = u->cfg_.context_dim;
    IGemmParams p = dense(ctx16, B * Lctx, CD, T.wkv2, 2 * C, Lctx);
    if (ctx16_lo) {          // split-fp16: A' = [hi | lo | hi] against W' = [hi | hi | lo]
      p.a1 = ctx16_lo; p.c1 = CD; p.lda1 = CD; p.a2 = ctx16; p.c2 = CD; p.lda2 = CD; p.K = 3 * CD; p.k_alg = CD;
    }
    p.mode = EPI_HEADS; p.seg_dst[0] = T.ck; p.seg_dst[1] = T.cvt; p.seg_kind[0] = 0; p.seg_kind[1] = 1;
    p.heads = L.heads; p.dh = L.dh; p.ntok = Lctx; p.ntok_pad = Lp; p.segC = C; p.splitk = 1;
    if (!dry && !rc && Lp != Lctx) {
      hipError_t e = memset_async(T.cvt, 0, (size_t)B * C * Lp * sizeof(f16), s);
      if (e != hipSuccess) ok(fail(std::string("hipMemsetAsync: ") + hipGetErrorString(e)));
    }
    gemm(p);
  }

  Act attn_block(Layer& L, const Act& x) {
    const int H = x.H, W = x.W, N = H * W, M = B * N, C = L.cin;
    const int Np = (int)round_up(N, 8), Lp = (int)round_up(Lctx, 8);
    const float scale = 1.0f / sqrtf((float)L.dh);
    const size_t mark = scratch.off;
    f16* xn = S<f16>((size_t)M * C);
    const bool p1 = L.p1x1;              // proj_in / proj_out of this SpatialTransformer as 3-pass split-fp16
    f16* xn_lo = p1 ? S<f16>((size_t)M * C) : nullptr;
    // proj_in(norm(x)), attention.py:254-255: the GroupNorm either as its own launch (fp32 stream -> split-fp16 hi | lo operands) or
    // applied inside the GEMM while it stages its A operand (gemm_split16_gn_kernel: same operand bits, same products, one launch less)
    IGemmParams pin;
    pin.M = M; pin.N = C; pin.K = C; pin.ksize = 1; pin.Hout = N; pin.Wout = 1; pin.B = B;
    const bool fold_ln = ln_fold_on && C % 64 == 0 && C <= 1280 && N % 64 == 0 && M % 64 == 0 && M >= u->ln_fold_min_rows_;
    const bool gn_in_gemm = gn_proj_fold && fold_ln && p1 && M >= 512 && split16_gn_supported(pin);
    // the head of the SpatialTransformer (GroupNorm-apply -> proj_in -> q | k | v of the first transformer block) as one row-strip chain
    // launch (rowchain.hip st_head_kernel; UNet::st_head_): LayerNorm fold on, split-fp16 proj_in, C = 320
    const bool chain_head = st_head_on && fold_ln && p1 && !gn_in_gemm && L.tb[0].lnf[0] != nullptr &&
                            st_head_supported(C, M, N, Np, L.heads, L.dh) && dense1x1(nullptr, nullptr, M, C, L.w16[0], C, N, p1).split16;
    long long* gn_stats = nullptr;
    if (gn_in_gemm || chain_head) gn_stats = groupnorm(x, nullptr, L.f32[0], L.f32[1], 1e-6f, 0, nullptr, nullptr, nullptr, nullptr, nullptr, /*stats_only=*/true);
    else groupnorm(x, nullptr, L.f32[0], L.f32[1], 1e-6f, 0, xn, nullptr, nullptr, xn_lo, nullptr);
    float* t = S<float>((size_t)M * C);
    f16* ln = S<f16>((size_t)M * C);
    f16* q = S<f16>((size_t)M * C);
    f16* k = S<f16>((size_t)M * C);
    f16* vt = S<f16>((size_t)B * C * Np);
    f16* ao = S<f16>((size_t)M * C);
    f16* gg = S<f16>((size_t)M * 4 * C);
    // Every LayerNorm of the block reads the token stream `t` right after the GEMM that produced it.  Two forms:
    //  * folded into the GEMM that READS it (IGemmParams::lnp_out / lnf_*): the producer stores ln = fp16(gamma * t) and the row
    //    statistics, the consumer corrects its accumulators -- no launch; taken where the producer is not split (many rows);
    //  * a post-op launch behind the producer (launch_igemm issues it after the GEMM / its split-K reduce): ln = LN(t).
    float* lnp = fold_ln ? S<float>((size_t)(C / 32) * M * 2) : nullptr;
    auto with_ln = [&](IGemmParams& p, const float* gamma, const float* beta) {
      if (fold_ln) { p.out_f16 = ln; p.f16_scale = gamma; p.lnp_out = lnp; p.splitk = 1; }
      else { p.ln_gamma = gamma; p.ln_beta = beta; p.ln_out = ln; p.ln_eps = 1e-5f; }
    };
    auto fold_in = [&](IGemmParams& p, const float* cs, const float* dn) {      // the GEMM that reads `ln`
      if (!fold_ln) return;
      p.lnf_part = lnp; p.lnf_npart = C / 32; p.lnf_eps = 1e-5f; p.lnf_cs = cs; p.lnf_d = dn; p.bias = nullptr;
    };
    if (chain_head) {
      TBlock& T = L.tb[0];
      StHeadParams h;
      h.x = x.p; h.gn_acc = gn_stats; h.gn_gamma = L.f32[0]; h.gn_beta = L.f32[1]; h.gn_eps = 1e-6f;
      h.w_in = L.w16[0]; h.b_in = L.f32[2]; h.t = t; h.ln_gamma = T.ln[0]; h.ln_eps = 1e-5f;
      h.wqkv = T.wqkv; h.lnf_cs = T.lnf[0]; h.lnf_d = T.lnf[1]; h.q = q; h.k = k; h.vt = vt;
      h.M = M; h.B = B; h.ntok = N; h.ntok_pad = Np; h.heads = L.heads; h.dh = L.dh; h.C = C;
      if (!dry && !rc && Np != N) {
        hipError_t e = memset_async(vt, 0, (size_t)B * C * Np * sizeof(f16), s);
        if (e != hipSuccess) ok(fail(std::string("hipMemsetAsync: ") + hipGetErrorString(e)));
      }
      if (!dry && !rc) ok(launch_st_head(h, s));
    } else if (gn_in_gemm) {
      IGemmParams p = dense(nullptr, M, C, L.w16[0], C, N);
      p.xf0 = x.p; p.gn_in_acc = gn_stats; p.gn_in_gamma = L.f32[0]; p.gn_in_beta = L.f32[1]; p.gn_in_eps = 1e-6f; p.gn_in_silu = 0;
      p.ldw = 3 * C; p.splitk = 1;
      p.bias = L.f32[2]; p.out_f32 = t; p.ldo = C;
      with_ln(p, L.tb[0].ln[0], L.tb[0].ln[1]);
      if (p.ln_out) ok(fail("internal: GroupNorm-folding proj_in needs the LayerNorm fold"));
      if (!dry && !rc) ok(launch_split16_gn(p, s));
    } else {
      IGemmParams p = dense1x1(xn, xn_lo, M, C, L.w16[0], C, N, p1);
      p.bias = L.f32[2]; p.out_f32 = t; p.ldo = C;
      with_ln(p, L.tb[0].ln[0], L.tb[0].ln[1]);                      // norm1 of the first block
      gemm(p);
    }
    const int depth = (int)L.tb.size();
    // the tail of the SpatialTransformer (GEGLU -> FF-out -> proj_out) as one row-strip chain launch: last (= only) transformer block,
    // LayerNorm fold on, split-fp16 proj_out, C = 320 (UNet::ff_tail_)
    const bool chain_ff = ff_tail_on && fold_ln && p1 && depth == 1 && L.tb[0].lnf_csd != nullptr && ff_tail_supported(C, M, N) &&
                          dense1x1(nullptr, nullptr, M, C, L.w16[1], C, N, p1).split16;
    const bool chain_tail = chain_ff && st_tail_on;      // ... with attn2's out-projection in front (ff_tail_kernel HEAD)
    for (int d = 0; d < depth; ++d) {
      TBlock& T = L.tb[d];
      // x = attn1(norm1(x)) + x                                   attention.py:212
      if (!(chain_head && d == 0)) {       // (the chain launch above has written q, k, v^T of the first block)
        IGemmParams p = dense(ln, M, C, T.wqkv, 3 * C, N);
        p.mode = EPI_HEADS; p.seg_dst[0] = q; p.seg_dst[1] = k; p.seg_dst[2] = vt;
        p.seg_kind[0] = 0; p.seg_kind[1] = 0; p.seg_kind[2] = 1;
        p.heads = L.heads; p.dh = L.dh; p.ntok = N; p.ntok_pad = Np; p.segC = C; p.splitk = 1;
        fold_in(p, T.lnf[0], T.lnf[1]);
        if (!dry && !rc && Np != N) {
          hipError_t e = memset_async(vt, 0, (size_t)B * C * Np * sizeof(f16), s);
          if (e != hipSuccess) ok(fail(std::string("hipMemsetAsync: ") + hipGetErrorString(e)));
        }
        gemm(p);
      }
      attention(q, k, vt, ao, L, N, N, Np, scale);
      // attn1's out-projection (+ x) and attn2's to_q over norm2 as one row-strip chain launch (rowchain.hip, st_head_kernel KIND 1)
      const bool ctx_fused_here = fuse_ctx_q && L.dh <= fuse_ctx_maxd && attention_ctx_supported(L.dh, C, Lctx) && !(fold_ln && C > 640);
      bool chain_ctx = false;
      const bool chain_mid = st_mid_on && fold_ln && !ctx_fused_here && T.lnf[2] != nullptr && st_head_supported(C, M, N, Np, L.heads, L.dh);
      if (chain_mid) {
        StHeadParams h;
        h.a16 = ao; h.w_in = T.wo1; h.b_in = T.bo1; h.t = t; h.ln_gamma = T.ln[2]; h.ln_eps = 1e-5f;
        h.wqkv = T.wq2; h.lnf_cs = T.lnf[2]; h.lnf_d = T.lnf[3]; h.q = q;
        h.M = M; h.B = B; h.ntok = N; h.ntok_pad = Np; h.heads = L.heads; h.dh = L.dh; h.C = C;
        if (ctx16) context_kv(L, d);
        // ... and the cross-attention itself inside that launch (st_head_kernel CTX: q never leaves the CU)
        chain_ctx = st_mid_ctx_on && L.dh == 40 && Lctx <= 128;
        if (chain_ctx) { h.ctx_k = T.ck; h.ctx_vt = T.cvt; h.ctx_nkv = Lctx; h.ctx_nkv_pad = Lp; h.ctx_scale = scale; h.ao_out = ao; }
        if (!dry && !rc) ok(launch_st_mid(h, s));
      } else {
        IGemmParams p = dense(ao, M, C, T.wo1, C, N);
        p.bias = T.bo1; p.residual = t; p.ldr = C; p.out_f32 = t; p.ldo = C;
        with_ln(p, T.ln[2], T.ln[3]);                                // norm2
        gemm(p);
      }
      // x = attn2(norm2(x), context) + x                           attention.py:213
      if (ctx16 && !chain_mid) context_kv(L, d);
      if (chain_mid) {
        if (!chain_ctx) attention(q, T.ck, T.cvt, ao, L, N, Lctx, Lp, scale);       // (q = to_q(norm2(t)) came out of the chain launch)
      } else if (fuse_ctx_q && L.dh <= fuse_ctx_maxd && attention_ctx_supported(L.dh, C, Lctx) && !(fold_ln && C > 640)) {
        // to_q inside the attention kernel (attn_ctx.hip): one launch for q = norm2(x) Wq^T and softmax(q K^T) V
        AttnCtxParams a;
        a.x = ln; a.wq = T.wq2; a.k = T.ck; a.vt = T.cvt; a.out = ao;
        a.BH = B * L.heads; a.heads = L.heads; a.nq = N; a.nkv = Lctx; a.nkv_pad = Lp; a.d = L.dh; a.C = C; a.scale = scale;
        if (fold_ln) { a.lnf_part = lnp; a.lnf_npart = C / 32; a.lnf_eps = 1e-5f; a.M = M; a.lnf_cs = T.lnf[2]; a.lnf_d = T.lnf[3]; }
        if (!dry && !rc) ok(launch_attention_ctx(a, s));
      } else {
        IGemmParams p = dense(ln, M, C, T.wq2, C, N);
        p.mode = EPI_HEADS; p.seg_dst[0] = q; p.seg_kind[0] = 0;
        p.heads = L.heads; p.dh = L.dh; p.ntok = N; p.ntok_pad = Np; p.segC = C; p.splitk = 1;
        fold_in(p, T.lnf[2], T.lnf[3]);
        gemm(p);
        attention(q, T.ck, T.cvt, ao, L, N, Lctx, Lp, scale);
      }
      if (!chain_tail) {                     // (else: inside the chain launch below)
        IGemmParams p = dense(ao, M, C, T.wo2, C, N);
        p.bias = T.bo2; p.residual = t; p.ldr = C; p.out_f32 = t; p.ldo = C;
        with_ln(p, T.ln[4], T.ln[5]);                                // norm3
        gemm(p);
      }
      // x = ff(norm3(x)) + x                                       attention.py:214
      if (chain_ff) break;                   // (depth 1: GEGLU, FF-out and proj_out are the chain launch below)
      {
        IGemmParams p = dense(ln, M, C, T.wgg, 8 * C, N);
        p.mode = EPI_GEGLU; p.bias = T.bgg; p.out_f16 = gg; p.ldo = 4 * C; p.splitk = 1;
        fold_in(p, T.lnf[4], T.lnf[5]);
        gemm(p);
      }
      {
        IGemmParams p = dense(gg, M, 4 * C, T.wff2, C, N);
        p.bias = T.bff2; p.residual = t; p.ldr = C; p.ldo = C;
        if (d + 1 < depth) {
          p.out_f32 = t;
          with_ln(p, L.tb[d + 1].ln[0], L.tb[d + 1].ln[1]);          // norm1 of the next block
        } else {
          // last block: only proj_out reads the result -- emit its split-fp16 operand (hi | lo) directly
          p.out_f16 = ln; p.out_lo = xn_lo;
        }
        gemm(p);
      }
    }
    Act out = make_act(P<float>((size_t)M * C), C, H, W, true);
    {
      IGemmParams p = dense1x1(ln, xn_lo, M, C, L.w16[1], C, N, p1);
      p.Hout = H * W;                       // (dense: rows per sample)
      p.bias = L.f32[3]; p.residual = x.p; p.ldr = C; p.out_f32 = out.p; p.ldo = C;
      attach_gn_targets(p, out);
      attach_f16_copy(p, out);
      if (chain_ff) {
        // out = x + proj_out(t + FF(norm3(t))): one row-strip chain launch (rowchain.hip); `ln` / `lnp` are what attn2's out-projection stored
        TBlock& T = L.tb[0];
        FfTailParams q;
        q.ln = ln; q.lnp = lnp; q.ln_eps = 1e-5f; q.csd = T.lnf_csd; q.wgg = T.wgg; q.wff2 = T.wff2; q.bff2 = T.bff2; q.t = t; q.wpo = L.w16[1];
        if (chain_tail) { q.a16 = ao; q.wo = T.wo2; q.bo = T.bo2; q.ln_gamma = T.ln[4]; }     // t += ao Wo2^T + bo2 first (attention.py:213)
        q.epi = p;
        if (!dry && !rc) ok(launch_ff_tail(q, s));
      } else {
        gemm(p);
      }
    }
    scratch.off = mark;
    return out;
  }

  void attention(const f16* q, const f16* k, const f16* vt, f16* out, Layer& L, int nq, int nkv, int nkv_pad, float scale) {
    AttnParams a;
    a.q = q; a.k = k; a.vt = vt; a.out = out; a.BH = B * L.heads; a.heads = L.heads; a.nq = nq; a.nkv = nkv;
    a.nkv_pad = nkv_pad; a.d = L.dh; a.scale = scale;
    if (!dry && !rc) ok(launch_attention(a, s));
  }

  Act resample(Layer& L, const Act& x, bool up) {
    const int Hin = x.H, Win = x.W, C = x.C;
    const int Hout = up ? 2 * Hin : (Hin - 1) / 2 + 1, Wout = up ? 2 * Win : (Win - 1) / 2 + 1;
    const size_t mark = scratch.off;
    // the fp16 operand: stored by the producing GEMM's epilogue when there is one (see FwdBase::attach_f16_copy), else cast here
    const f16* x16 = nullptr;
    if (dry) { if (!want_f16_copy(x)) (void)S<f16>((size_t)B * Hin * Win * C); }
    else {
      x16 = f16_copy(x);
      if (!x16) {
        f16* c = S<f16>((size_t)B * Hin * Win * C);
        if (!rc) ok(launch_cast_f16(x.p, c, nullptr, (int64_t)B * Hin * Win * C, s));
        x16 = c;
      }
    }
    Act out = make_act(P<float>((size_t)B * Hout * Wout * C), L.cout, Hout, Wout, true);
    IGemmParams p = conv3(x16, C, Hin, Win, Hout, Wout, up ? 1 : 2, up ? 1 : 0, L.w16[0], L.cout);
    p.bias = L.f32[0]; p.out_f32 = out.p; p.ldo = L.cout;
    attach_gn_targets(p, out);
    gemm(p);
    scratch.off = mark;
    return out;
  }

  Act run_layer(Layer& L, const Act& x, const Act* skip) {
    switch (L.kind) {
      case L_RES: return res_block(L, x, skip);
      case L_ATTN: return attn_block(L, x);
      case L_DOWN: return resample(L, x, false);
      case L_UP: return resample(L, x, true);
      default: ok(fail("unexpected layer kind")); return x;
    }
  }
};

// Cross-attention K / V^T caches of every transformer block: capacity-sized device buffers owned by the handle.  They are
// allocated by finalize() for the default capacity (8 rows x 80 padded context tokens: SD v1's 77-token prompts at the largest
// batch one call takes) and grown only by reserve_ctx_cache() -- sdmi_unet_reserve_context / sdmi_unet_cache_context, both off
// the hot path.  sdmi_unet_forward never allocates: a context beyond the capacity is an error that names the remedy.
int UNet::reserve_ctx_cache(int B, int Lctx) {
  const int64_t need = (int64_t)B * round_up(Lctx, 8);          // (B * Lctx * C <= B * Lp * C: one capacity covers K and V^T)
  if (need <= ctx_cap_) return 0;
  ++ctx_gen_;                // (the K / V^T buffers move: recorded launch tapes are stale)
  auto each = [&](Layer& L) -> int {
    if (L.kind != L_ATTN) return 0;
    for (auto& T : L.tb) {
      if (T.ck) { (void)hipFree(T.ck); T.ck = nullptr; }
      if (T.cvt) { (void)hipFree(T.cvt); T.cvt = nullptr; }
      SDMI_HIP_OK(hipMalloc((void**)&T.ck, (size_t)need * L.cin * sizeof(f16)));
      SDMI_HIP_OK(hipMalloc((void**)&T.cvt, (size_t)need * L.cin * sizeof(f16)));
    }
    return 0;
  };
  for (auto& blk : input_blocks_) for (auto& L : blk) if (each(L)) return -1;
  for (auto& L : middle_) if (each(L)) return -1;
  for (auto& blk : output_blocks_) for (auto& L : blk) if (each(L)) return -1;
  ctx_cap_ = need; ctx_B_ = 0; ctx_L_ = 0; ctx_valid_ = false;
  return 0;
}

// bind the cache to the shape of this call: no allocation; a different shape only invalidates the cached contents
int UNet::ensure_ctx_cache(int B, int Lctx, bool may_grow) {
  if (ctx_B_ == B && ctx_L_ == Lctx) return 0;
  const int64_t need = (int64_t)B * round_up(Lctx, 8);
  if (need > ctx_cap_) {
    if (may_grow) { if (reserve_ctx_cache(B, Lctx)) return -1; }
    else return fail("context of " + std::to_string(B) + " x " + std::to_string(Lctx) + " tokens exceeds the reserved K/V capacity (" +
                     std::to_string(ctx_cap_) + " padded rows): call sdmi_unet_reserve_context(h, B, Lctx) once, outside the sampling loop");
  }
  ctx_B_ = B; ctx_L_ = Lctx; ctx_valid_ = false;
  return 0;
}

int UNet::cache_timesteps(const int64_t* t_host, int n, hipStream_t stream) {
  SDMI_CHECK(finalized_, "sdmi_unet_finalize() has not succeeded yet");
  SDMI_CHECK(n >= 0 && n <= 4096 && (n == 0 || t_host != nullptr), "bad timestep list");
  drop_timestep_table();
  if (n == 0) return 0;
  const int mc = cfg_.model_channels;
  const size_t row = (size_t)emb_total_, tmp = (size_t)8 * (mc + 2 * (size_t)te_);
  const size_t need = (size_t)n * row + tmp;
  if (need > emb_tab_floats_) {              // (grow-only; called once per sampling run, not per UNet call)
    if (emb_tab_) (void)hipFree(emb_tab_);
    emb_tab_ = nullptr; emb_tab_floats_ = 0;
    SDMI_HIP_OK(hipMalloc((void**)&emb_tab_, need * sizeof(float)));
    emb_tab_floats_ = need;
  }
  if ((size_t)n > emb_tab_tcap_) {
    if (emb_tab_tdev_) (void)hipFree(emb_tab_tdev_);
    emb_tab_tdev_ = nullptr; emb_tab_tcap_ = 0;
    SDMI_HIP_OK(hipMalloc((void**)&emb_tab_tdev_, (size_t)n * sizeof(int64_t)));
    emb_tab_tcap_ = (size_t)n;
  }
  std::vector<int64_t> host(t_host, t_host + n);      // (the copy below reads a buffer this object owns, not the caller's)
  emb_tab_src_.swap(host);
  SDMI_HIP_OK(hipMemcpyAsync(emb_tab_tdev_, emb_tab_src_.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, stream));
  float* temb = emb_tab_ + (size_t)n * row;
  float* e1 = temb + (size_t)8 * mc;
  float* emb = e1 + (size_t)8 * te_;
  for (int i0 = 0; i0 < n; i0 += 8) {        // the same launches as a forward's timestep path, 8 timesteps as the batch rows
    const int nb = std::min(8, n - i0);
    int r = launch_timestep_embedding(emb_tab_tdev_ + i0, nullptr, temb, nb, mc, stream);
    if (!r) r = launch_small_linear(temb, mc, te_w0_, te_b0_, e1, te_, nb, te_, mc, 0, stream);
    if (!r) r = launch_small_linear(e1, te_, te_w2_, te_b2_, emb, te_, nb, te_, te_, 1, stream);
    if (!r) r = launch_small_linear(emb, te_, emb_w_, emb_b_, emb_tab_ + (size_t)i0 * row, emb_total_, nb, emb_total_, te_, 1, stream);
    if (r) return r;
  }
  emb_tab_t_ = emb_tab_src_;
  return 0;
}

int UNet::hint_timestep(int64_t t) {
  emb_hint_row_ = -1;
  for (size_t i = 0; i < emb_tab_t_.size(); ++i)
    if (emb_tab_t_[i] == t) { emb_hint_row_ = (int)i; break; }
  return 0;
}

int UNet::run(const float* x, const int64_t* t_i64, const float* t_f32, const float* ctx, float* eps_out, int B, int H,
              int W, int Lctx, void* workspace, int64_t ws_bytes, hipStream_t stream, bool dry, bool ctx_only,
              int64_t* bytes_needed) {
  SDMI_CHECK(dry || finalized_, "sdmi_unet_finalize() has not succeeded yet");
  SDMI_CHECK(B >= 1 && B <= 8, "batch (CFG rows) must be 1..8 per call");
  SDMI_CHECK(H >= 1 && W >= 1 && Lctx >= 1, "bad shape");
  const int down = 1 << (cfg_.n_levels - 1);
  SDMI_CHECK(H % down == 0 && W % down == 0, "H and W must be divisible by 2^(levels-1) (the UNet's skip concat requires it)");

  // The timestep hint is an announcement about THIS call only: consume it up front, so that a call that fails on any of the
  // early returns below (workspace too small, missing context, a launch error) cannot leave it behind for an unrelated
  // later forward, which would then silently take the table row of the old timestep.  (Sizing and context-only calls
  // do not touch the timestep path and leave the hint alone.)
  const int hint_row = emb_hint_row_;
  if (!dry && !ctx_only) emb_hint_row_ = -1;

  // ---- launch tape (tape.h): replay the recorded launch list of this (shape, workspace, mode, knobs), or record it below -------------
  // (both read per call: the tests flip them between two forwards)
  const char* e_rp = getenv("SDMI_REPLAY"); const char* e_rv = getenv("SDMI_REPLAY_VERIFY");
  const bool replay_on = !(e_rp && atoi(e_rp) == 0);
  const bool replay_verify = e_rv && atoi(e_rv) != 0;
  const bool hinted = t_i64 != nullptr && hint_row >= 0 && hint_row < (int)emb_tab_t_.size();
  const bool tape_ok = replay_on && !dry && !ctx_only && !side_stream_ && !prof_enabled() && !tune_collecting() && !range_check_enabled() &&
                       workspace != nullptr && (t_i64 || t_f32);
  TapeKey tkey;
  uintptr_t caller[Tape::R_COUNT] = {0, 0, 0, 0, 0};
  Tape* rec = nullptr;
  std::unique_ptr<Tape> verify_against;
  if (tape_ok) {
    tkey.B = B; tkey.H = H; tkey.W = W; tkey.Lctx = Lctx; tkey.mode = hinted ? 0 : (t_i64 ? 1 : 2);
    tkey.ws = workspace; tkey.ws_bytes = ws_bytes; tkey.have_ctx = ctx != nullptr;
    tkey.env = sdmi_env_hash() ^ (tune_generation() * 0x9e3779b97f4a7c15ull); tkey.weights_gen = weights_gen_; tkey.ctx_gen = ctx_gen_;
    caller[Tape::R_X] = (uintptr_t)x; caller[Tape::R_OUT] = (uintptr_t)eps_out;
    caller[Tape::R_T] = hinted ? 0 : (t_i64 ? (uintptr_t)t_i64 : (uintptr_t)t_f32);
    caller[Tape::R_CTX] = (uintptr_t)ctx;
    caller[Tape::R_EMB] = hinted ? (uintptr_t)(emb_tab_ + (size_t)hint_row * emb_total_) : 0;
    for (size_t i = 0; i < tapes_.size(); ++i) {
      if (!(tapes_[i].first == tkey)) continue;
      std::unique_ptr<Tape> hit = std::move(tapes_[i].second);
      tapes_.erase(tapes_.begin() + (long)i);
      SDMI_CHECK(finalized_, "sdmi_unet_finalize() has not succeeded yet");
      if (ensure_ctx_cache(B, Lctx, false)) return -1;
      if (!ctx) SDMI_CHECK(ctx_valid_, "ctx == NULL but no cached context for this (B, Lctx); call sdmi_unet_cache_context first");
      hit->retarget(caller);
      if (replay_verify) { verify_against = std::move(hit); break; }      // run the executor and compare what it launches
      const int e = hit->replay(stream);
      const bool sets = hit->sets_ctx_valid;
      const int64_t need = hit->bytes_needed;
      tapes_.emplace_back(tkey, std::move(hit));
      if (e) return fail(std::string("launch tape replay: ") + hipGetErrorString((hipError_t)e));
      if (sets) ctx_valid_ = true;
      if (bytes_needed) *bytes_needed = need;
      ++tape_hits_;
      return 0;
    }
  }
  std::unique_ptr<Tape> fresh;
  if (tape_ok) { fresh.reset(new Tape()); rec = fresh.get(); }

  Fwd f;
  f.u = this; f.s = stream; f.dry = dry; f.B = B; f.Lctx = Lctx; f.zero = zero_; f.precise_1x1 = precise_1x1_;
  {
    // (both knobs are read per call -- the tests flip them between two forwards; the row statistics ride on the 16-byte epilogue)
    const char* e_fold = getenv("SDMI_LN_FOLD"); const char* e_vec = getenv("SDMI_EPI_VEC");
    f.ln_fold_on = (e_fold ? atoi(e_fold) != 0 : ln_fold_) && !(e_vec && atoi(e_vec) == 0);
    const char* e_ff = getenv("SDMI_FF_TAIL");
    f.ff_tail_on = e_ff ? atoi(e_ff) != 0 : ff_tail_;
    const char* e_sh = getenv("SDMI_ST_HEAD");
    f.st_head_on = e_sh ? atoi(e_sh) != 0 : st_head_;
    const char* e_sm = getenv("SDMI_ST_MID");
    f.st_mid_on = e_sm ? atoi(e_sm) != 0 : st_head_;
    const char* e_st = getenv("SDMI_ST_TAIL");
    f.st_tail_on = e_st ? atoi(e_st) != 0 : ff_tail_;
    const char* e_mc = getenv("SDMI_ST_MID_CTX");
    f.st_mid_ctx_on = e_mc ? atoi(e_mc) != 0 : st_head_;
    const char* e_gc = getenv("SDMI_GN_CONV");
    f.gn_conv_on = e_gc ? atoi(e_gc) != 0 : false;       // (opt-in: bit-identical, 36 us against 41 us with hot operands, +4 us per launch inside a UNet call -- profiles/gn_conv3_r05.txt)
#ifdef SDMI_EXPERIMENTS      // (kernels of the experiments build: attn_ctx.hip, gemm_split16_gn_kernel)
    const char* e_ctx = getenv("SDMI_ATTN_CTX_FUSED");
    f.fuse_ctx_q = e_ctx && atoi(e_ctx) != 0;
    if (const char* e_md = getenv("SDMI_ATTN_CTX_MAXD")) f.fuse_ctx_maxd = atoi(e_md);
    const char* e_gp = getenv("SDMI_GN_PROJ_FOLD");
    f.gn_proj_fold = (e_gp && atoi(e_gp) != 0) && f.ln_fold_on;       // (the kernel has no LayerNorm post-op launch: it rides on the fold)
#endif
  }
  if (side_stream_ && !dry && !prof_enabled()) {      // (the per-launch profiler times launches on one stream)
    if (!side_) {
      SDMI_HIP_OK(hipStreamCreateWithFlags(&side_, hipStreamNonBlocking));
      for (auto& e : side_ev_) SDMI_HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    f.side = side_; f.side_ev = side_ev_; f.side_nev = 32;
  }
  // first pass (always dry) sizes the two arenas; the persist arena sits in front of the scratch arena
  int64_t persist_bytes = 0, scratch_bytes = 0;
  for (int pass = (dry ? 0 : 0); pass < 2; ++pass) {
    const bool d = (pass == 0) ? true : false;
    if (pass == 1 && dry) break;
    f.dry = d; f.rc = 0;
    f.n_acts = 0;
    TapeRecGuard tape_guard(d ? nullptr : rec);         // (the dry pass launches nothing)
    f.plan = fuse_gn_stats_ ? &gn_plan_ : nullptr;
    if (d) gn_plan_.clear();
    f.persist = Arena(); f.scratch = Arena();
    f.persist.dry = f.scratch.dry = d;
    if (!d) {
      SDMI_CHECK(persist_bytes + scratch_bytes <= ws_bytes, "workspace too small: need " +
                 std::to_string(persist_bytes + scratch_bytes) + " bytes, got " + std::to_string(ws_bytes));
      SDMI_CHECK(workspace != nullptr, "workspace is NULL");
      f.persist.base = (char*)workspace; f.persist.cap = (size_t)persist_bytes;
      f.scratch.base = (char*)workspace + persist_bytes; f.scratch.cap = (size_t)scratch_bytes;
      if (ensure_ctx_cache(B, Lctx, ctx_only)) return -1;
    }
    const int mc = cfg_.model_channels;
    if (f.begin_pass((int64_t)12 << 20)) return -1;    // 48 MB of fp32 split-K slabs (largest user: 8 x 512 x 1280)
    f16* ctx16 = f.P<f16>((size_t)B * Lctx * cfg_.context_dim);
    f16* ctx16_lo = precise_kv_ ? f.P<f16>((size_t)B * Lctx * cfg_.context_dim) : nullptr;
    const bool have_ctx = (ctx != nullptr) || d;
    if (have_ctx) {
      if (!d) { int r = launch_cast_f16(ctx, ctx16, ctx16_lo, (int64_t)B * Lctx * cfg_.context_dim, stream); if (r) return r; }
      f.ctx16 = ctx16; f.ctx16_lo = ctx16_lo;
    } else {
      SDMI_CHECK(ctx_valid_, "ctx == NULL but no cached context for this (B, Lctx); call sdmi_unet_cache_context first");
      f.ctx16 = nullptr; f.ctx16_lo = nullptr;
    }
    if (ctx_only) {
      auto each = [&](Layer& L) { if (L.kind == L_ATTN) for (int dd = 0; dd < (int)L.tb.size(); ++dd) f.context_kv(L, dd); };
      for (auto& blk : input_blocks_) for (auto& L : blk) each(L);
      for (auto& L : middle_) each(L);
      for (auto& blk : output_blocks_) for (auto& L : blk) each(L);
    } else {
      // ---- time embedding (util.py:151-171, openaimodel.py:506-511,723-724) and all emb_layers at once ----
      float* temb = f.P<float>((size_t)B * mc);
      float* e1 = f.P<float>((size_t)B * te_);
      float* emb = f.P<float>((size_t)B * te_);
      f.emb_all = f.P<float>((size_t)B * emb_total_);
      f.emb_ld = emb_total_;
      if (!d && t_i64 != nullptr && hint_row >= 0 && hint_row < (int)emb_tab_t_.size()) {
        // every row has the hinted timestep and its emb_layers outputs are in the table: one shared row, nothing to launch
        f.emb_all = emb_tab_ + (size_t)hint_row * emb_total_;
        f.emb_ld = 0;
      } else if (!d) {
        int r = launch_timestep_embedding(t_i64, t_f32, temb, B, mc, stream);
        if (!r) r = launch_small_linear(temb, mc, te_w0_, te_b0_, e1, te_, B, te_, mc, 0, stream);
        if (!r) r = launch_small_linear(e1, te_, te_w2_, te_b2_, emb, te_, B, te_, te_, 1, stream);
        if (!r) r = launch_small_linear(emb, te_, emb_w_, emb_b_, f.emb_all, emb_total_, B, emb_total_, te_, 1, stream);
        if (r) return r;
      }
      // ---- input blocks ----
      std::vector<Act> hs;
      Act h;
      {
        Layer& L = input_blocks_[0][0];
        // round 6: conv_in emits the GroupNorm statistics of its output itself (the two GroupNorms that read it -- input_blocks.1.0 and, through
        // the skip concat, the last output block -- ran the statistics kernel; rounds 1-5: "conv_in is not an igemm: no fused statistics")
        const char* e_cis = getenv("SDMI_CONV_IN_STATS");                 // (A/B knob, read per call; 0 = the statistics kernel as in rounds 1-5)
        h = f.make_act(f.P<float>((size_t)B * H * W * mc), mc, H, W, (H * W) % 16 == 0 && !(e_cis && atoi(e_cis) == 0));
        IGemmParams st;                                   // (carrier of the statistics targets only)
        f.attach_gn_targets(st, h);
        if (!d) {
          int r = launch_conv_in(x, L.w32[0], L.f32[0], h.p, B, cfg_.in_channels, H, W, mc, stream, st.gn_n, st.gn_acc, st.gn_cpg, st.gn_cbase);
          if (r) return r;
        }
        hs.push_back(h);
      }
      for (size_t bi = 1; bi < input_blocks_.size(); ++bi) {
        for (auto& L : input_blocks_[bi]) h = f.run_layer(L, h, nullptr);
        hs.push_back(h);
      }
      for (auto& L : middle_) h = f.run_layer(L, h, nullptr);
      for (auto& blk : output_blocks_) {
        Act skip = hs.back(); hs.pop_back();
        SDMI_CHECK(skip.H == h.H && skip.W == h.W, "skip/h spatial mismatch");
        bool first = true;
        for (auto& L : blk) { h = f.run_layer(L, h, first ? &skip : nullptr); first = false; }
      }
      // ---- output head: GN -> SiLU -> conv3x3 (fp32) ----
      float* hn = f.S<float>((size_t)B * H * W * mc);
      f.groupnorm(h, nullptr, out_gamma_, out_beta_, 1e-5f, 1, nullptr, hn, nullptr);
      if (!d && !f.rc) { int r = launch_conv_out(hn, out_w_, out_b_, eps_out, B, H, W, mc, cfg_.out_channels, stream); if (r) return r; }
    }
    if (f.rc) return f.rc;
    if (d) { persist_bytes = (int64_t)f.persist.peak + 256; scratch_bytes = (int64_t)f.scratch.peak + 256; }
    else {
      SDMI_CHECK(!f.persist.overflow && !f.scratch.overflow, "internal: arena overflow");
      if (have_ctx) ctx_valid_ = true;
    }
  }
  if (bytes_needed) *bytes_needed = persist_bytes + scratch_bytes;
  if (fresh) {
    // the caller ranges the recorded parameter bytes may point into
    const int64_t cd = cfg_.context_dim;
    fresh->base[Tape::R_X] = caller[Tape::R_X]; fresh->span[Tape::R_X] = (size_t)B * cfg_.in_channels * H * W * sizeof(float);
    fresh->base[Tape::R_OUT] = caller[Tape::R_OUT]; fresh->span[Tape::R_OUT] = (size_t)B * cfg_.out_channels * H * W * sizeof(float);
    fresh->base[Tape::R_T] = caller[Tape::R_T]; fresh->span[Tape::R_T] = caller[Tape::R_T] ? (size_t)B * (t_i64 ? 8 : 4) : 0;
    fresh->base[Tape::R_CTX] = caller[Tape::R_CTX]; fresh->span[Tape::R_CTX] = ctx ? (size_t)B * Lctx * cd * sizeof(float) : 0;
    fresh->base[Tape::R_EMB] = caller[Tape::R_EMB]; fresh->span[Tape::R_EMB] = caller[Tape::R_EMB] ? (size_t)emb_total_ * sizeof(float) : 0;
    fresh->bytes_needed = persist_bytes + scratch_bytes;
    fresh->sets_ctx_valid = ctx != nullptr;
    fresh->find_relocs();
    if (verify_against) {
      const Tape& a = *verify_against; const Tape& b = *fresh;
      // (the parameter structs carry padding bytes of unspecified content: compared are the launch list, the argument layout, the set of
      // relocated words and their values -- i.e. every caller pointer after the retarget)
      bool same = a.ops.size() == b.ops.size() && a.blob.size() == b.blob.size() && a.arg_off == b.arg_off && a.relocs.size() == b.relocs.size();
      for (size_t i = 0; same && i < a.relocs.size(); ++i)
        same = a.relocs[i].off == b.relocs[i].off && a.relocs[i].which == b.relocs[i].which &&
               memcmp(a.blob.data() + a.relocs[i].off, b.blob.data() + b.relocs[i].off, 8) == 0;
      for (size_t i = 0; same && i < a.ops.size(); ++i) {
        const Tape::Op &p = a.ops[i], &q = b.ops[i];
        same = p.kind == q.kind && p.fn == q.fn && p.grid.x == q.grid.x && p.grid.y == q.grid.y && p.grid.z == q.grid.z && p.block.x == q.block.x &&
               p.shmem == q.shmem && p.ptr == q.ptr && p.value == q.value && p.bytes == q.bytes;
      }
      SDMI_CHECK(same, "SDMI_REPLAY_VERIFY: the retargeted launch tape differs from what the executor launches for this call");
      ++tape_hits_;
    } else {
      ++tape_records_;
    }
    if (tapes_.size() >= kMaxTapes) tapes_.erase(tapes_.begin());
    tapes_.emplace_back(tkey, std::move(fresh));
  }
  return 0;
}

}  // namespace sdmi
