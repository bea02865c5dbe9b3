// Text-encoder executor state (see clip.cpp): HF CLIPTextModel as wrapped by FrozenCLIPEmbedder.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/sdmi.h"
#include "common.h"
#include "unet.h"

namespace sdmi {

enum CWKind { CW_F32, CW_ROWS16, CW_BIAS_ROWS };

struct CLayer {   // CLIPEncoderLayer
  f16* wqkv = nullptr; float* bqkv = nullptr;     // [3C][C], [3C]   q_proj | k_proj | v_proj
  f16* wo = nullptr; float* bo = nullptr;
  f16* w1 = nullptr; float* b1 = nullptr;         // fc1 [I][C]
  f16* w2 = nullptr; float* b2 = nullptr;         // fc2 [C][I]
  float* ln[4] = {nullptr, nullptr, nullptr, nullptr};   // layer_norm1.{weight,bias}, layer_norm2.{weight,bias}
};

struct CWeightSlot {
  std::string key;
  std::vector<int64_t> shape;
  CWKind kind = CW_F32;
  void** dst = nullptr; int row0 = 0, total_rows = 0;
  bool set = false;
};

class ClipText {
 public:
  ClipText() = default;
  ~ClipText();
  ClipText(const ClipText&) = delete;
  ClipText& operator=(const ClipText&) = delete;

  int build(const sdmi_clip_cfg& cfg);
  int set_weight(const char* key, const float* ptr, const int64_t* shape, int ndim, hipStream_t stream);
  int finalize();
  // ids int64 [B][L] (device) -> last_hidden_state fp32 [B][L][hidden] (after final_layer_norm)
  int forward(const int64_t* ids, float* out, int B, int L, void* workspace, int64_t ws_bytes, hipStream_t stream, bool dry,
              int64_t* bytes_needed);
  const std::vector<CWeightSlot>& slots() const { return slots_; }
  sdmi_clip_cfg cfg_{};

 private:
  void expect(const std::string& key, std::vector<int64_t> shape, CWKind kind, void** dst, int row0 = 0, int total_rows = 0);
  int dev_alloc(void** dst, size_t bytes);
  std::vector<CLayer> layers_;
  std::vector<CWeightSlot> slots_;
  std::map<std::string, int> slot_index_;
  std::vector<void*> owned_;
  float *tok_ = nullptr, *pos_ = nullptr, *fln_g_ = nullptr, *fln_b_ = nullptr;
  f16* zero_ = nullptr;
  bool finalized_ = false;
};

}  // namespace sdmi
