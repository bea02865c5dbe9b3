// First-stage (AutoencoderKL) executor state (see vae.cpp).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/sdmi.h"
#include "common.h"
#include "unet.h"

namespace sdmi {

enum VKind { V_RES, V_ATTN, V_UP, V_DOWN };
enum VWKind { VW_F32, VW_CONV, VW_SPLIT3, VW_PLAIN16, VW_CONV_OUT };

struct VLayer {
  VKind kind = V_RES;
  std::string prefix;
  int cin = 0, cout = 0;
  // V_RES : w16 = {conv1, conv2, nin_shortcut(split3)}, f32 = {norm1.w, norm1.b, conv1.b, norm2.w, norm2.b, conv2.b, nin.b}
  // V_ATTN: w16 = {q, k, v, proj_out},                   f32 = {norm.w, norm.b, q.b, k.b, v.b, proj_out.b}
  // V_UP / V_DOWN: w16 = {conv},                         f32 = {conv.b}
  f16* w16[4] = {nullptr, nullptr, nullptr, nullptr};
  float* f32[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

struct VWeightSlot {
  std::string key;
  std::vector<int64_t> shape;
  VWKind kind = VW_F32;
  void** dst = nullptr;
  bool set = false;
};

class Vae {
 public:
  Vae() = default;
  ~Vae();
  Vae(const Vae&) = delete;
  Vae& operator=(const Vae&) = delete;

  int build(const sdmi_vae_cfg& cfg, int parts);
  int set_weight(const char* key, const float* ptr, const int64_t* shape, int ndim, hipStream_t stream);
  int finalize();
  // z [B, embed_dim, H, W] fp32 NCHW -> img [B, out_ch, H*f, W*f] fp32 NCHW; z is multiplied by z_scale first
  int decode(const float* z, float z_scale, float* img, int B, int H, int W, void* workspace, int64_t ws_bytes,
             hipStream_t stream, bool dry, int64_t* bytes_needed);
  // img [B, in_channels, H, W] fp32 NCHW -> moments [B, 2*embed_dim, H/f, W/f] fp32 NCHW
  int encode(const float* img, float* moments, int B, int H, int W, void* workspace, int64_t ws_bytes, hipStream_t stream,
             bool dry, int64_t* bytes_needed);

  const std::vector<VWeightSlot>& slots() const { return slots_; }
  int factor() const { return 1 << (cfg_.n_levels - 1); }

  sdmi_vae_cfg cfg_{};
  int parts_ = 0;
  f16* zero_ = nullptr;
  bool precise_1x1_ = true;

 private:
  friend struct VFwd;
  void expect(const std::string& key, std::vector<int64_t> shape, VWKind kind, void** dst);
  int dev_alloc(void** dst, size_t bytes);

  std::vector<VLayer> dec_, enc_;
  std::vector<VWeightSlot> slots_;
  std::map<std::string, int> slot_index_;
  std::vector<void*> owned_;
  // decoder ends
  float *pq_w_ = nullptr, *pq_b_ = nullptr;                 // post_quant_conv
  float *dci_w_ = nullptr, *dci_b_ = nullptr;               // decoder.conv_in (raw OIHW fp32)
  float *dno_g_ = nullptr, *dno_b_ = nullptr, *dco_w_ = nullptr, *dco_b_ = nullptr;   // decoder.norm_out, conv_out (OHWI fp32)
  // encoder ends
  float *eci_w_ = nullptr, *eci_b_ = nullptr;
  float *eno_g_ = nullptr, *eno_b_ = nullptr, *eco_w_ = nullptr, *eco_b_ = nullptr;
  float *q_w_ = nullptr, *q_b_ = nullptr;                   // quant_conv
  int dec_c_end_ = 0, enc_c_end_ = 0;                       // channels entering norm_out
  bool finalized_ = false;
};

}  // namespace sdmi
