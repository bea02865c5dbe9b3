// First-stage executor: AutoencoderKL.decode / .encode (ldm/models/autoencoder.py:324-333) as a static launch
// sequence over the same gfx950 kernels the UNet uses.
//   Decoder.forward   ldm/modules/diffusionmodules/model.py:528-568   (ctor :462-526)
//   Encoder.forward   model.py:427-460                                 (ctor :368-425)
//   ResnetBlock       model.py:119-141 (temb is None for the first stage)
//   AttnBlock         model.py:172-202 (one head, d = C)
//   Upsample / Downsample  model.py:41-79 (nearest x2 + conv3x3 / zero pad (0,1,0,1) + conv3x3 stride 2)
//
// Data layout: the activation stream is fp32 NHWC [B*H*W][C]; GroupNorm(32, eps 1e-6)+SiLU writes the fp16 A operand of
// the following implicit-GEMM conv; nin_shortcut (1x1 on the raw stream) runs as a 3-pass split-fp16 GEMM; the first
// (z -> 512) and last (128 -> 3) convs and the two 1x1 "quant" convs are fp32.  Attention over N = H*W tokens with
// d = C = 512 is three GEMMs (q k^T, softmax rows, P v) on the igemm kernel, in query chunks so S stays cache-sized.
#include "vae.h"

#include <math.h>

#include <algorithm>

namespace sdmi {

void Vae::expect(const std::string& key, std::vector<int64_t> shape, VWKind kind, void** dst) {
  VWeightSlot s;
  s.key = key; s.shape = std::move(shape); s.kind = kind; s.dst = dst;
  slot_index_[key] = (int)slots_.size();
  slots_.push_back(std::move(s));
}

int Vae::build(const sdmi_vae_cfg& c, int parts) {
  cfg_ = c; parts_ = parts;
  SDMI_CHECK(parts >= 1 && parts <= 3, "parts: 1 decoder, 2 encoder, 3 both");
  SDMI_CHECK(c.n_levels >= 1 && c.n_levels <= 8 && c.num_res_blocks >= 1, "bad level / res block count");
  SDMI_CHECK(c.ch % 64 == 0, "ch must be a multiple of 64 on this path");
  SDMI_CHECK(c.in_channels >= 1 && c.in_channels <= 16 && c.z_channels >= 1 && c.z_channels <= 4 && c.embed_dim >= 1 &&
                 c.embed_dim <= 8 && c.out_ch >= 1 && c.out_ch <= 8,
             "in_channels <= 16, z_channels <= 4, embed_dim <= 8, out_ch <= 8 on this path");
#ifdef SDMI_EXPERIMENTS
  if (const char* e = getenv("SDMI_PRECISE_1X1")) precise_1x1_ = atoi(e) != 0;
#endif
  const int n = c.n_levels;
  auto res = [&](const std::string& p, int ci, int co) { VLayer L; L.kind = V_RES; L.prefix = p; L.cin = ci; L.cout = co; return L; };
  auto one = [&](VKind k, const std::string& p, int ch) { VLayer L; L.kind = k; L.prefix = p; L.cin = ch; L.cout = ch; return L; };

  if (parts & 1) {
    int bi = c.ch * c.ch_mult[n - 1];
    dec_.push_back(res("decoder.mid.block_1", bi, bi));
    dec_.push_back(one(V_ATTN, "decoder.mid.attn_1", bi));
    dec_.push_back(res("decoder.mid.block_2", bi, bi));
    for (int lvl = n - 1; lvl >= 0; --lvl) {
      const int bo = c.ch * c.ch_mult[lvl];
      for (int i = 0; i <= c.num_res_blocks; ++i) {
        dec_.push_back(res("decoder.up." + std::to_string(lvl) + ".block." + std::to_string(i), bi, bo));
        bi = bo;
      }
      if (lvl != 0) dec_.push_back(one(V_UP, "decoder.up." + std::to_string(lvl) + ".upsample", bi));
    }
    dec_c_end_ = bi;
  }
  if (parts & 2) {
    int bi = c.ch;
    for (int lvl = 0; lvl < n; ++lvl) {
      const int bo = c.ch * c.ch_mult[lvl];
      for (int i = 0; i < c.num_res_blocks; ++i) {
        enc_.push_back(res("encoder.down." + std::to_string(lvl) + ".block." + std::to_string(i), bi, bo));
        bi = bo;
      }
      if (lvl != n - 1) enc_.push_back(one(V_DOWN, "encoder.down." + std::to_string(lvl) + ".downsample", bi));
    }
    enc_.push_back(res("encoder.mid.block_1", bi, bi));
    enc_.push_back(one(V_ATTN, "encoder.mid.attn_1", bi));
    enc_.push_back(res("encoder.mid.block_2", bi, bi));
    enc_c_end_ = bi;
  }

  // ---- expected state_dict entries (AutoencoderKL.state_dict() minus loss.*) -------------------------------------------
  auto visit = [&](VLayer& L) {
    const std::string& p = L.prefix;
    const int64_t ci = L.cin, co = L.cout;
    switch (L.kind) {
      case V_RES:
        expect(p + ".norm1.weight", {ci}, VW_F32, (void**)&L.f32[0]);
        expect(p + ".norm1.bias", {ci}, VW_F32, (void**)&L.f32[1]);
        expect(p + ".conv1.weight", {co, ci, 3, 3}, VW_CONV, (void**)&L.w16[0]);
        expect(p + ".conv1.bias", {co}, VW_F32, (void**)&L.f32[2]);
        expect(p + ".norm2.weight", {co}, VW_F32, (void**)&L.f32[3]);
        expect(p + ".norm2.bias", {co}, VW_F32, (void**)&L.f32[4]);
        expect(p + ".conv2.weight", {co, co, 3, 3}, VW_CONV, (void**)&L.w16[1]);
        expect(p + ".conv2.bias", {co}, VW_F32, (void**)&L.f32[5]);
        if (ci != co) {
          expect(p + ".nin_shortcut.weight", {co, ci, 1, 1}, precise_1x1_ ? VW_SPLIT3 : VW_PLAIN16, (void**)&L.w16[2]);
          expect(p + ".nin_shortcut.bias", {co}, VW_F32, (void**)&L.f32[6]);
        }
        break;
      case V_ATTN: {
        expect(p + ".norm.weight", {ci}, VW_F32, (void**)&L.f32[0]);
        expect(p + ".norm.bias", {ci}, VW_F32, (void**)&L.f32[1]);
        const char* names[4] = {"q", "k", "v", "proj_out"};
        for (int i = 0; i < 4; ++i) {
          expect(p + "." + names[i] + ".weight", {ci, ci, 1, 1}, VW_PLAIN16, (void**)&L.w16[i]);
          expect(p + "." + names[i] + ".bias", {ci}, VW_F32, (void**)&L.f32[2 + i]);
        }
        break;
      }
      case V_UP:
      case V_DOWN:
        expect(p + ".conv.weight", {co, ci, 3, 3}, VW_CONV, (void**)&L.w16[0]);
        expect(p + ".conv.bias", {co}, VW_F32, (void**)&L.f32[0]);
        break;
    }
  };
  // NOTE: slots hold pointers into the VLayer objects: dec_ / enc_ must not reallocate after this point.
  if (parts & 2) {
    expect("encoder.conv_in.weight", {c.ch, c.in_channels, 3, 3}, VW_F32, (void**)&eci_w_);
    expect("encoder.conv_in.bias", {c.ch}, VW_F32, (void**)&eci_b_);
    for (auto& L : enc_) visit(L);
    expect("encoder.norm_out.weight", {enc_c_end_}, VW_F32, (void**)&eno_g_);
    expect("encoder.norm_out.bias", {enc_c_end_}, VW_F32, (void**)&eno_b_);
    expect("encoder.conv_out.weight", {2 * c.z_channels, enc_c_end_, 3, 3}, VW_CONV_OUT, (void**)&eco_w_);
    expect("encoder.conv_out.bias", {2 * c.z_channels}, VW_F32, (void**)&eco_b_);
    expect("quant_conv.weight", {2 * c.embed_dim, 2 * c.z_channels, 1, 1}, VW_F32, (void**)&q_w_);
    expect("quant_conv.bias", {2 * c.embed_dim}, VW_F32, (void**)&q_b_);
  }
  if (parts & 1) {
    expect("post_quant_conv.weight", {c.z_channels, c.embed_dim, 1, 1}, VW_F32, (void**)&pq_w_);
    expect("post_quant_conv.bias", {c.z_channels}, VW_F32, (void**)&pq_b_);
    expect("decoder.conv_in.weight", {c.ch * c.ch_mult[n - 1], c.z_channels, 3, 3}, VW_F32, (void**)&dci_w_);
    expect("decoder.conv_in.bias", {c.ch * c.ch_mult[n - 1]}, VW_F32, (void**)&dci_b_);
    for (auto& L : dec_) visit(L);
    expect("decoder.norm_out.weight", {dec_c_end_}, VW_F32, (void**)&dno_g_);
    expect("decoder.norm_out.bias", {dec_c_end_}, VW_F32, (void**)&dno_b_);
    expect("decoder.conv_out.weight", {c.out_ch, dec_c_end_, 3, 3}, VW_CONV_OUT, (void**)&dco_w_);
    expect("decoder.conv_out.bias", {c.out_ch}, VW_F32, (void**)&dco_b_);
  }
  return 0;
}

Vae::~Vae() {
  for (void* p : owned_) (void)hipFree(p);
}

int Vae::dev_alloc(void** dst, size_t bytes) {
  if (*dst) return 0;
  SDMI_HIP_OK(hipMalloc(dst, bytes));
  owned_.push_back(*dst);
  return 0;
}

int Vae::set_weight(const char* key, const float* ptr, const int64_t* shape, int ndim, hipStream_t stream) {
  auto it = slot_index_.find(key);
  if (it == slot_index_.end()) return fail(std::string("unexpected weight key: ") + key);
  VWeightSlot& s = slots_[it->second];
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
    case VW_F32:
      rc = dev_alloc(s.dst, numel * sizeof(float));
      if (!rc) SDMI_HIP_OK(hipMemcpyAsync(*s.dst, dptr, numel * sizeof(float), hipMemcpyDeviceToDevice, stream));
      break;
    case VW_CONV:
      rc = dev_alloc(s.dst, numel * sizeof(f16));
      if (!rc) rc = launch_pack_conv_weight(dptr, (f16*)*s.dst, (int)shape[0], (int)shape[1], (int)shape[2], (int)shape[3], stream);
      break;
    case VW_SPLIT3:
      rc = dev_alloc(s.dst, 3 * numel * sizeof(f16));
      if (!rc) rc = launch_pack_split3(dptr, (f16*)*s.dst, (int)shape[0], (int)shape[1], stream);
      break;
    case VW_PLAIN16:
      rc = dev_alloc(s.dst, numel * sizeof(f16));
      if (!rc) rc = launch_pack_rows(dptr, (f16*)*s.dst, (int)shape[0], (int)shape[1], 0, (int)shape[1], stream);
      break;
    case VW_CONV_OUT:
      rc = dev_alloc(s.dst, numel * sizeof(float));
      if (!rc) rc = launch_pack_conv_out(dptr, (float*)*s.dst, (int)shape[0], (int)shape[1], stream);
      break;
  }
  if (st.release(stream)) return -1;
  if (rc) return rc;
  s.set = true;
  finalized_ = false;
  return 0;
}

int Vae::finalize() {
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

// ------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------
struct VFwd : FwdBase {
  static constexpr float EPS = 1e-6f;        // Normalize(): GroupNorm(32, eps=1e-6)   model.py:37-38

  Act res_block(VLayer& L, const Act& x) {   // model.py:119-141
    const int H = x.H, W = x.W, M = B * H * W, Cin = L.cin, Cout = L.cout;
    if (x.C != Cin) ok(fail("res block channel mismatch at " + L.prefix));
    const size_t mark = scratch.off;
    const bool nin = Cin != Cout;
    f16* a = S<f16>((size_t)M * Cin);
    f16* raw = nin ? S<f16>((size_t)M * Cin) : nullptr;
    f16* raw_lo = (nin && precise_1x1) ? S<f16>((size_t)M * Cin) : nullptr;
    float* h = S<float>((size_t)M * Cout);
    Act out; out.p = P<float>((size_t)M * Cout); out.C = Cout; out.H = H; out.W = W;
    groupnorm(x, nullptr, L.f32[0], L.f32[1], EPS, 1, a, nullptr, raw, nullptr, raw_lo);
    {
      IGemmParams p = conv3(a, Cin, H, W, H, W, 1, 0, L.w16[0], Cout);
      p.bias = L.f32[2]; p.out_f32 = h; p.ldo = Cout;
      gemm(p);
    }
    const float* residual = x.p;
    if (nin) {
      IGemmParams p = dense1x1(raw, raw_lo, M, Cin, L.w16[2], Cout, H * W, precise_1x1);
      p.bias = L.f32[6]; p.out_f32 = out.p; p.ldo = Cout;
      gemm(p);
      residual = out.p;
    }
    Act hact; hact.p = h; hact.C = Cout; hact.H = H; hact.W = W;
    f16* a2 = S<f16>((size_t)M * Cout);
    groupnorm(hact, nullptr, L.f32[3], L.f32[4], EPS, 1, a2, nullptr, nullptr);
    {
      IGemmParams p = conv3(a2, Cout, H, W, H, W, 1, 0, L.w16[1], Cout);
      p.bias = L.f32[5]; p.residual = residual; p.ldr = Cout; p.out_f32 = out.p; p.ldo = Cout;
      gemm(p);
    }
    scratch.off = mark;
    return out;
  }

  Act attn_block(VLayer& L, const Act& x) {  // model.py:172-202
    const int H = x.H, W = x.W, N = H * W, M = B * N, C = L.cin;
    if (N % 64) ok(fail("first-stage attention needs H*W % 64 == 0 (latent sides multiples of 8)"));
    const float scale = 1.0f / sqrtf((float)C);
    const size_t mark = scratch.off;
    f16* xn = S<f16>((size_t)M * C);
    groupnorm(x, nullptr, L.f32[0], L.f32[1], EPS, 0, xn, nullptr, nullptr);
    f16* q = S<f16>((size_t)M * C);
    f16* k = S<f16>((size_t)M * C);
    f16* vt = S<f16>((size_t)M * C);      // per image [C][N]
    f16* ao = S<f16>((size_t)M * C);
    for (int i = 0; i < 2; ++i) {         // q = xn Wq^T + bq,  k = xn Wk^T + bk
      IGemmParams p = dense(xn, M, C, L.w16[i], C, N);
      p.bias = L.f32[2 + i]; p.out_f16 = i ? k : q; p.ldo = C; p.splitk = 1;
      gemm(p);
    }
    const int QC = std::min(N, 2048);     // query rows per chunk: S (fp32) + P (fp16) <= 48 MB at N = 4096
    float* Sm = S<float>((size_t)QC * N);
    f16* Pm = S<f16>((size_t)QC * N);
    for (int b = 0; b < B; ++b) {
      {   // v^T [C][N] = Wv [C][C] x xn_b[N][C]^T  (the bias is added after P v: softmax rows sum to 1)
        IGemmParams p = dense(L.w16[2], C, C, xn + (size_t)b * N * C, N, C);
        p.out_f16 = vt + (size_t)b * N * C; p.ldo = N; p.splitk = 1;
        gemm(p);
      }
      for (int r0 = 0; r0 < N; r0 += QC) {
        const int rows = std::min(QC, N - r0);
        {
          IGemmParams p = dense(q + ((size_t)b * N + r0) * C, rows, C, k + (size_t)b * N * C, N, rows);
          p.out_f32 = Sm; p.ldo = N; p.splitk = 1;
          gemm(p);
        }
        if (!dry && !rc) ok(launch_softmax_rows(Sm, Pm, rows, N, N, N, scale, s));
        {
          IGemmParams p = dense(Pm, rows, N, vt + (size_t)b * N * C, C, rows);
          p.bias = L.f32[4]; p.out_f16 = ao + ((size_t)b * N + r0) * C; p.ldo = C; p.splitk = 1;
          gemm(p);
        }
      }
    }
    Act out; out.p = P<float>((size_t)M * C); out.C = C; out.H = H; out.W = W;
    {
      IGemmParams p = dense(ao, M, C, L.w16[3], C, N);
      p.bias = L.f32[5]; p.residual = x.p; p.ldr = C; p.out_f32 = out.p; p.ldo = C;
      gemm(p);
    }
    scratch.off = mark;
    return out;
  }

  Act resample(VLayer& L, const Act& x, bool up) {   // model.py:41-79
    const int Hin = x.H, Win = x.W, C = x.C;
    if (!up && ((Hin | Win) & 1)) ok(fail("first-stage Downsample needs even H and W"));
    const int Hout = up ? 2 * Hin : Hin / 2, Wout = up ? 2 * Win : Win / 2;
    const size_t mark = scratch.off;
    f16* x16 = S<f16>((size_t)B * Hin * Win * C);
    if (!dry && !rc) ok(launch_cast_f16(x.p, x16, nullptr, (int64_t)B * Hin * Win * C, s));
    Act out; out.p = P<float>((size_t)B * Hout * Wout * C); out.C = C; out.H = Hout; out.W = Wout;
    IGemmParams p = conv3(x16, C, Hin, Win, Hout, Wout, up ? 1 : 2, up ? 1 : 0, L.w16[0], C);
    if (!up) p.pad = 0;                   // F.pad(x, (0,1,0,1)) + conv(stride 2, padding 0)
    p.bias = L.f32[0]; p.out_f32 = out.p; p.ldo = C;
    gemm(p);
    scratch.off = mark;
    return out;
  }

  Act run_layer(VLayer& L, const Act& x) {
    switch (L.kind) {
      case V_RES: return res_block(L, x);
      case V_ATTN: return attn_block(L, x);
      case V_UP: return resample(L, x, true);
      case V_DOWN: return resample(L, x, false);
    }
    return x;
  }
};

// two passes over one executor: the first (dry) sizes the persist / scratch arenas, the second launches
template <class Body>
static int run_two_pass(VFwd& f, bool dry, void* workspace, int64_t ws_bytes, int64_t* bytes_needed, Body body) {
  int64_t persist_bytes = 0, scratch_bytes = 0;
  for (int pass = 0; pass < 2; ++pass) {
    const bool d = pass == 0;
    if (pass == 1 && dry) break;
    f.dry = d; f.rc = 0;
    f.persist = Arena(); f.scratch = Arena();
    f.persist.dry = f.scratch.dry = d;
    if (!d) {
      SDMI_CHECK(persist_bytes + scratch_bytes <= ws_bytes, "workspace too small: need " +
                 std::to_string(persist_bytes + scratch_bytes) + " bytes, got " + std::to_string(ws_bytes));
      SDMI_CHECK(workspace != nullptr, "workspace is NULL");
      f.persist.base = (char*)workspace; f.persist.cap = (size_t)persist_bytes;
      f.scratch.base = (char*)workspace + persist_bytes; f.scratch.cap = (size_t)scratch_bytes;
    }
    if (f.begin_pass((int64_t)8 << 20)) return -1;
    if (body(f)) return -1;
    if (f.rc) return f.rc;
    if (d) {
      persist_bytes = (int64_t)round_up((int64_t)f.persist.peak, 4096) + 4096;
      scratch_bytes = (int64_t)round_up((int64_t)f.scratch.peak, 4096) + 4096;
      if (bytes_needed) *bytes_needed = persist_bytes + scratch_bytes;
    } else {
      SDMI_CHECK(!f.persist.overflow && !f.scratch.overflow, "internal: arena overflow");
    }
  }
  return 0;
}

int Vae::decode(const float* z, float z_scale, float* img, int B, int H, int W, void* workspace, int64_t ws_bytes,
                hipStream_t stream, bool dry, int64_t* bytes_needed) {
  SDMI_CHECK(parts_ & 1, "this handle was created without the decoder");
  SDMI_CHECK(dry || finalized_, "sdmi_vae_finalize() has not succeeded yet");
  SDMI_CHECK(B >= 1 && B <= 8, "batch must be 1..8 per call");
  SDMI_CHECK(H >= 1 && W >= 1, "bad shape");
  SDMI_CHECK(dry || (z != nullptr && img != nullptr), "z / img is NULL");
  VFwd f;
  f.s = stream; f.B = B; f.zero = zero_; f.precise_1x1 = precise_1x1_;
  const int n = cfg_.n_levels, c_in = cfg_.ch * cfg_.ch_mult[n - 1];
  return run_two_pass(f, dry, workspace, ws_bytes, bytes_needed, [&](VFwd& f) -> int {
    const bool d = f.dry;
    float* zq = f.P<float>((size_t)B * cfg_.z_channels * H * W);
    if (!d && launch_pointwise_nchw(z, pq_w_, pq_b_, zq, B, cfg_.embed_dim, cfg_.z_channels, H * W, z_scale, stream)) return -1;
    Act x; x.p = f.P<float>((size_t)B * H * W * c_in); x.C = c_in; x.H = H; x.W = W;
    if (!d && launch_conv_in(zq, dci_w_, dci_b_, x.p, B, cfg_.z_channels, H, W, c_in, stream)) return -1;
    for (auto& L : dec_) x = f.run_layer(L, x);
    if (f.rc) return f.rc;
    float* hn = f.S<float>((size_t)B * x.H * x.W * x.C);
    f.groupnorm(x, nullptr, dno_g_, dno_b_, VFwd::EPS, 1, nullptr, hn, nullptr);
    if (!d && !f.rc && launch_conv_out(hn, dco_w_, dco_b_, img, B, x.H, x.W, x.C, cfg_.out_ch, stream)) return -1;
    return f.rc;
  });
}

int Vae::encode(const float* img, float* moments, int B, int H, int W, void* workspace, int64_t ws_bytes,
                hipStream_t stream, bool dry, int64_t* bytes_needed) {
  SDMI_CHECK(parts_ & 2, "this handle was created without the encoder");
  SDMI_CHECK(dry || finalized_, "sdmi_vae_finalize() has not succeeded yet");
  SDMI_CHECK(B >= 1 && B <= 8, "batch must be 1..8 per call");
  const int fct = factor();
  SDMI_CHECK(H >= fct && W >= fct && H % fct == 0 && W % fct == 0, "H and W must be multiples of 2^(levels-1)");
  SDMI_CHECK(dry || (img != nullptr && moments != nullptr), "img / moments is NULL");
  VFwd f;
  f.s = stream; f.B = B; f.zero = zero_; f.precise_1x1 = precise_1x1_;
  return run_two_pass(f, dry, workspace, ws_bytes, bytes_needed, [&](VFwd& f) -> int {
    const bool d = f.dry;
    Act x; x.p = f.P<float>((size_t)B * H * W * cfg_.ch); x.C = cfg_.ch; x.H = H; x.W = W;
    if (!d && launch_conv_in(img, eci_w_, eci_b_, x.p, B, cfg_.in_channels, H, W, cfg_.ch, stream)) return -1;
    for (auto& L : enc_) x = f.run_layer(L, x);
    if (f.rc) return f.rc;
    float* hn = f.S<float>((size_t)B * x.H * x.W * x.C);
    f.groupnorm(x, nullptr, eno_g_, eno_b_, VFwd::EPS, 1, nullptr, hn, nullptr);
    const int zc2 = 2 * cfg_.z_channels;
    float* mo = f.S<float>((size_t)B * zc2 * x.H * x.W);
    if (!d && !f.rc && launch_conv_out(hn, eco_w_, eco_b_, mo, B, x.H, x.W, x.C, zc2, stream)) return -1;
    if (!d && !f.rc && launch_pointwise_nchw(mo, q_w_, q_b_, moments, B, zc2, 2 * cfg_.embed_dim, x.H * x.W, 1.0f, stream))
      return -1;
    return f.rc;
  });
}

}  // namespace sdmi
