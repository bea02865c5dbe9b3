// UNet executor state (see unet.cpp).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/sdmi.h"
#include "common.h"

namespace sdmi {

enum LayerKind { L_CONV_IN, L_RES, L_ATTN, L_DOWN, L_UP };
enum WKind { W_F32, W_F32_ROWS, W_CONV, W_CONV_OUT, W_ROWS16, W_GEGLU_W, W_GEGLU_B, W_SPLIT3 };

struct TBlock {   // BasicTransformerBlock (ldm/modules/attention.py:196-215)
  f16* wqkv = nullptr;   // [3C][C]   attn1 to_q | to_k | to_v
  f16* wo1 = nullptr; float* bo1 = nullptr;
  f16* wq2 = nullptr;    // [C][C]
  f16* wkv2 = nullptr;   // [2C][context_dim]  attn2 to_k | to_v
  f16* wo2 = nullptr; float* bo2 = nullptr;
  f16* wgg = nullptr; float* bgg = nullptr;     // GEGLU proj, rows interleaved (value32 | gate32)
  f16* wff2 = nullptr; float* bff2 = nullptr;
  float* ln[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  f16* ck = nullptr; f16* cvt = nullptr;        // cached cross-attention K [B*h][L][d] and V^T [B*h][d][Lpad]
};

struct Layer {
  LayerKind kind = L_RES;
  std::string prefix;
  int cin = 0, cout = 0, heads = 0, dh = 0, emb_off = 0, attn_index = -1;
  f16* w16[3] = {nullptr, nullptr, nullptr};
  float* w32[1] = {nullptr};
  float* f32[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  std::vector<TBlock> tb;
};

struct WeightSlot {
  std::string key;
  std::vector<int64_t> shape;
  WKind kind = W_F32;
  void** dst = nullptr; int row0 = 0, ld = 0; void** dst2 = nullptr;
  bool set = false;
};

class UNet {
 public:
  UNet() = default;
  ~UNet();
  UNet(const UNet&) = delete;
  UNet& operator=(const UNet&) = delete;

  int build(const sdmi_unet_cfg& cfg);
  int set_weight(const char* key, const float* ptr, const int64_t* shape, int ndim, hipStream_t stream);
  int finalize();
  // dry = size only; ctx_only = just the cross-attention K/V of every SpatialTransformer
  int run(const float* x, const int64_t* t_i64, const float* t_f32, const float* ctx, float* eps_out, int B, int H, int W,
          int Lctx, void* workspace, int64_t ws_bytes, hipStream_t stream, bool dry, bool ctx_only, int64_t* bytes_needed);

  const std::vector<WeightSlot>& slots() const { return slots_; }

  sdmi_unet_cfg cfg_{};
  int te_ = 0, emb_total_ = 0, n_attn_ = 0;
  f16* zero_ = nullptr;
  // 1x1 convs on the residual stream (skip_connection, proj_in, proj_out) run as 3-pass split-fp16 GEMMs
  // (a_hi*w_hi + a_lo*w_hi + a_hi*w_lo): ~22-bit operands for 5 % of the FLOPs (DESIGN.md "precision")
  bool precise_1x1_ = true;
  // ResBlock convs at >= 32x32 run as fused GroupNorm+SiLU+conv3x3 with halo-staged input tiles (conv3gn.hip)
  bool fuse_gn_conv_ = true;

 private:
  friend struct Fwd;
  void expect(const std::string& key, std::vector<int64_t> shape, WKind kind, void** dst, int row0 = 0, int ld = 0,
              void** dst2 = nullptr);
  int dev_alloc(void** dst, size_t bytes);
  int ensure_ctx_cache(int B, int Lctx);

  std::vector<std::vector<Layer>> input_blocks_, output_blocks_;
  std::vector<Layer> middle_;
  std::vector<WeightSlot> slots_;
  std::map<std::string, int> slot_index_;
  std::vector<void*> owned_;
  float *te_w0_ = nullptr, *te_b0_ = nullptr, *te_w2_ = nullptr, *te_b2_ = nullptr;
  float *emb_w_ = nullptr, *emb_b_ = nullptr;       // concatenated emb_layers [emb_total][te], [emb_total]
  float *out_gamma_ = nullptr, *out_beta_ = nullptr, *out_w_ = nullptr, *out_b_ = nullptr;
  bool finalized_ = false;
  int ctx_B_ = 0, ctx_L_ = 0; bool ctx_valid_ = false;
};

}  // namespace sdmi
