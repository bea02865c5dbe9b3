// extern "C" surface of libsdmi (include/sdmi.h).  Thin: argument checks, struct translation, error capture.
#include <string.h>

#include <new>

#include "prof.h"
#include "unet.h"
#include "vae.h"
#include "clip.h"

namespace sdmi {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(const std::string& msg) { g_err = msg; return -1; }
}  // namespace sdmi

struct sdmi_unet { sdmi::UNet impl; };
struct sdmi_vae { sdmi::Vae impl; };
struct sdmi_clip { sdmi::ClipText impl; };

using namespace sdmi;

static void* g_zero_page[16] = {nullptr};
static int zero_page(const f16** out) {
  int dev = 0;
  SDMI_HIP_OK(hipGetDevice(&dev));
  SDMI_CHECK(dev >= 0 && dev < 16, "device index");
  if (!g_zero_page[dev]) {
    SDMI_HIP_OK(hipMalloc(&g_zero_page[dev], 4096));
    SDMI_HIP_OK(hipMemset(g_zero_page[dev], 0, 4096));
  }
  *out = (const f16*)g_zero_page[dev];
  return 0;
}

extern "C" {

const char* sdmi_last_error(void) { return g_err.c_str(); }
int sdmi_abi_version(void) { return SDMI_ABI_VERSION; }
int sdmi_has_experiments(void) {
#ifdef SDMI_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}

int sdmi_unet_create(const sdmi_unet_cfg* cfg, sdmi_unet** out) {
  SDMI_CHECK(cfg && out, "null argument");
  sdmi_unet* h = new (std::nothrow) sdmi_unet();
  SDMI_CHECK(h != nullptr, "out of host memory");
  if (h->impl.build(*cfg)) { delete h; return -1; }
  *out = h;
  return 0;
}
int sdmi_unet_destroy(sdmi_unet* h) { delete h; return 0; }
int sdmi_unet_num_weights(const sdmi_unet* h) { return h ? (int)h->impl.slots().size() : fail("null handle"); }
int sdmi_unet_weight_info(const sdmi_unet* h, int idx, char* key_buf, int key_buf_len, int64_t* shape4, int* ndim) {
  SDMI_CHECK(h && key_buf && shape4 && ndim, "null argument");
  SDMI_CHECK(idx >= 0 && idx < (int)h->impl.slots().size(), "weight index out of range");
  const WeightSlot& s = h->impl.slots()[idx];
  SDMI_CHECK((int)s.key.size() + 1 <= key_buf_len, "key buffer too small");
  memcpy(key_buf, s.key.c_str(), s.key.size() + 1);
  *ndim = (int)s.shape.size();
  for (int i = 0; i < 4; ++i) shape4[i] = i < *ndim ? s.shape[i] : 1;
  return 0;
}
int sdmi_unet_set_weight(sdmi_unet* h, const char* key, const float* ptr, const int64_t* shape, int ndim, void* stream) {
  SDMI_CHECK(h && key && ptr && shape, "null argument");
  return h->impl.set_weight(key, ptr, shape, ndim, (hipStream_t)stream);
}
int sdmi_unet_finalize(sdmi_unet* h) { SDMI_CHECK(h, "null handle"); return h->impl.finalize(); }

int64_t sdmi_unet_packed_bytes(sdmi_unet* h) {
  if (!h) { fail("null handle"); return 0; }
  int64_t total = 0;
  h->impl.packed_layout(nullptr, &total);
  return total;
}
int sdmi_unet_export_packed(sdmi_unet* h, void* host_buf, int64_t bytes, void* stream) {
  SDMI_CHECK(h && host_buf, "null argument");
  return h->impl.export_packed(host_buf, bytes, (hipStream_t)stream);
}
int sdmi_unet_import_packed(sdmi_unet* h, const void* host_buf, int64_t bytes, void* stream) {
  SDMI_CHECK(h && host_buf, "null argument");
  return h->impl.import_packed(host_buf, bytes, (hipStream_t)stream);
}

int64_t sdmi_unet_workspace_bytes(sdmi_unet* h, int B, int H, int W, int Lctx) {
  if (!h) { fail("null handle"); return 0; }
  int64_t need = 0;
  if (h->impl.run(nullptr, nullptr, nullptr, nullptr, nullptr, B, H, W, Lctx, nullptr, 0, nullptr, true, false, &need)) return 0;
  return need;
}
int sdmi_unet_cache_context(sdmi_unet* h, const float* ctx, int B, int Lctx, void* workspace, int64_t workspace_bytes,
                            void* stream) {
  SDMI_CHECK(h && ctx, "null argument");
  const int down = 1 << (h->impl.cfg_.n_levels - 1);
  return h->impl.run(nullptr, nullptr, nullptr, ctx, nullptr, B, down, down, Lctx, workspace, workspace_bytes,
                     (hipStream_t)stream, false, true, nullptr);
}
int sdmi_unet_reserve_context(sdmi_unet* h, int B, int Lctx) {
  SDMI_CHECK(h, "null argument");
  SDMI_CHECK(B >= 1 && B <= 8 && Lctx >= 1, "bad context shape");
  return h->impl.reserve_ctx_cache(B, Lctx);
}
int sdmi_unet_cache_timesteps(sdmi_unet* h, const int64_t* t_host, int n, void* stream) {
  SDMI_CHECK(h, "null argument");
  return h->impl.cache_timesteps(t_host, n, (hipStream_t)stream);
}
int sdmi_unet_hint_timestep(sdmi_unet* h, int64_t t) {
  SDMI_CHECK(h, "null argument");
  return h->impl.hint_timestep(t);
}
int sdmi_unet_tape_stats(sdmi_unet* h, int64_t* replayed, int64_t* recorded) {
  SDMI_CHECK(h, "null argument");
  if (replayed) *replayed = (int64_t)h->impl.tape_hits_;
  if (recorded) *recorded = (int64_t)h->impl.tape_records_;
  return 0;
}
int sdmi_unet_forward(sdmi_unet* h, const float* x, const int64_t* t_i64, const float* t_f32, const float* ctx,
                      float* eps_out, int B, int H, int W, int Lctx, void* workspace, int64_t workspace_bytes, void* stream) {
  SDMI_CHECK(h && x && eps_out, "null argument");
  SDMI_CHECK((t_i64 != nullptr) != (t_f32 != nullptr), "pass exactly one of t_i64 / t_f32");
  return h->impl.run(x, t_i64, t_f32, ctx, eps_out, B, H, W, Lctx, workspace, workspace_bytes, (hipStream_t)stream, false,
                     false, nullptr);
}

int sdmi_sampler_step(const float* eps_model, int cfg, float scale, const float* x, int mode, const float* old0,
                      const float* old1, const float* old2, float a_t, float a_prev, float sigma, float sqrt_1m_at,
                      const float* noise, float* e_t_out, float* x_prev, float* pred_x0, int64_t n, void* stream) {
  SamplerStepParams p;
  p.eps_model = eps_model; p.cfg = cfg; p.scale = scale; p.x = x; p.mode = mode;
  p.old0 = old0; p.old1 = old1; p.old2 = old2;
  p.a_t = a_t; p.a_prev = a_prev; p.sigma = sigma; p.sqrt_1m_at = sqrt_1m_at;
  p.noise = noise; p.e_t_out = e_t_out; p.x_prev = x_prev; p.pred_x0 = pred_x0; p.n = n;
  return launch_sampler_step(p, (hipStream_t)stream);
}


int sdmi_dpm_solver_step(const float* eps_model, int cfg, float scale, const float* x, const float* m_prev, float alpha_s,
                         float sigma_s, float cx, float a, float inv_r0, int order, float* m_out, float* x_next, int64_t n,
                         void* stream) {
  DpmStepParams p;
  p.eps_model = eps_model; p.cfg = cfg; p.scale = scale; p.x = x; p.m_prev = m_prev; p.alpha_s = alpha_s; p.sigma_s = sigma_s;
  p.cx = cx; p.a = a; p.inv_r0 = inv_r0; p.order = order; p.m_out = m_out; p.x_next = x_next; p.n = n;
  return launch_dpm_step(p, (hipStream_t)stream);
}

// ---- first stage --------------------------------------------------------------------------------------
int sdmi_vae_create(const sdmi_vae_cfg* cfg, int parts, sdmi_vae** out) {
  SDMI_CHECK(cfg && out, "null argument");
  sdmi_vae* h = new (std::nothrow) sdmi_vae();
  SDMI_CHECK(h != nullptr, "out of host memory");
  if (h->impl.build(*cfg, parts)) { delete h; return -1; }
  *out = h;
  return 0;
}
int sdmi_vae_destroy(sdmi_vae* h) { delete h; return 0; }
int sdmi_vae_num_weights(const sdmi_vae* h) { return h ? (int)h->impl.slots().size() : fail("null handle"); }
int sdmi_vae_weight_info(const sdmi_vae* h, int idx, char* key_buf, int key_buf_len, int64_t* shape4, int* ndim) {
  SDMI_CHECK(h && key_buf && shape4 && ndim, "null argument");
  SDMI_CHECK(idx >= 0 && idx < (int)h->impl.slots().size(), "weight index out of range");
  const VWeightSlot& s = h->impl.slots()[idx];
  SDMI_CHECK((int)s.key.size() + 1 <= key_buf_len, "key buffer too small");
  memcpy(key_buf, s.key.c_str(), s.key.size() + 1);
  *ndim = (int)s.shape.size();
  for (int i = 0; i < 4; ++i) shape4[i] = i < *ndim ? s.shape[i] : 1;
  return 0;
}
int sdmi_vae_set_weight(sdmi_vae* h, const char* key, const float* ptr, const int64_t* shape, int ndim, void* stream) {
  SDMI_CHECK(h && key && ptr && shape, "null argument");
  return h->impl.set_weight(key, ptr, shape, ndim, (hipStream_t)stream);
}
int sdmi_vae_finalize(sdmi_vae* h) { SDMI_CHECK(h, "null handle"); return h->impl.finalize(); }
int64_t sdmi_vae_decode_workspace_bytes(sdmi_vae* h, int B, int H, int W) {
  if (!h) { fail("null handle"); return 0; }
  int64_t need = 0;
  if (h->impl.decode(nullptr, 1.f, nullptr, B, H, W, nullptr, 0, nullptr, true, &need)) return 0;
  return need;
}
int sdmi_vae_decode(sdmi_vae* h, const float* z, float z_scale, float* img, int B, int H, int W, void* workspace,
                    int64_t workspace_bytes, void* stream) {
  SDMI_CHECK(h && z && img, "null argument");
  return h->impl.decode(z, z_scale, img, B, H, W, workspace, workspace_bytes, (hipStream_t)stream, false, nullptr);
}
int64_t sdmi_vae_encode_workspace_bytes(sdmi_vae* h, int B, int H, int W) {
  if (!h) { fail("null handle"); return 0; }
  int64_t need = 0;
  if (h->impl.encode(nullptr, nullptr, B, H, W, nullptr, 0, nullptr, true, &need)) return 0;
  return need;
}
int sdmi_vae_encode(sdmi_vae* h, const float* img, float* moments, int B, int H, int W, void* workspace,
                    int64_t workspace_bytes, void* stream) {
  SDMI_CHECK(h && img && moments, "null argument");
  return h->impl.encode(img, moments, B, H, W, workspace, workspace_bytes, (hipStream_t)stream, false, nullptr);
}
int sdmi_k_pointwise_nchw(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int HW,
                          float in_scale, void* stream) {
  return launch_pointwise_nchw(x, w, bias, out, B, Cin, Cout, HW, in_scale, (hipStream_t)stream);
}
int sdmi_k_softmax_rows(const float* S, void* P_f16, int rows, int cols, float scale, void* stream) {
  return launch_softmax_rows(S, (f16*)P_f16, rows, cols, cols, cols, scale, (hipStream_t)stream);
}


// ---- text encoder -------------------------------------------------------------------------------------
int sdmi_clip_create(const sdmi_clip_cfg* cfg, sdmi_clip** out) {
  SDMI_CHECK(cfg && out, "null argument");
  sdmi_clip* h = new (std::nothrow) sdmi_clip();
  SDMI_CHECK(h != nullptr, "out of host memory");
  if (h->impl.build(*cfg)) { delete h; return -1; }
  *out = h;
  return 0;
}
int sdmi_clip_destroy(sdmi_clip* h) { delete h; return 0; }
int sdmi_clip_num_weights(const sdmi_clip* h) { return h ? (int)h->impl.slots().size() : fail("null handle"); }
int sdmi_clip_weight_info(const sdmi_clip* h, int idx, char* key_buf, int key_buf_len, int64_t* shape4, int* ndim) {
  SDMI_CHECK(h && key_buf && shape4 && ndim, "null argument");
  SDMI_CHECK(idx >= 0 && idx < (int)h->impl.slots().size(), "weight index out of range");
  const CWeightSlot& s = h->impl.slots()[idx];
  SDMI_CHECK((int)s.key.size() + 1 <= key_buf_len, "key buffer too small");
  memcpy(key_buf, s.key.c_str(), s.key.size() + 1);
  *ndim = (int)s.shape.size();
  for (int i = 0; i < 4; ++i) shape4[i] = i < *ndim ? s.shape[i] : 1;
  return 0;
}
int sdmi_clip_set_weight(sdmi_clip* h, const char* key, const float* ptr, const int64_t* shape, int ndim, void* stream) {
  SDMI_CHECK(h && key && ptr && shape, "null argument");
  return h->impl.set_weight(key, ptr, shape, ndim, (hipStream_t)stream);
}
int sdmi_clip_finalize(sdmi_clip* h) { SDMI_CHECK(h, "null handle"); return h->impl.finalize(); }
int64_t sdmi_clip_workspace_bytes(sdmi_clip* h, int B, int L) {
  if (!h) { fail("null handle"); return 0; }
  int64_t need = 0;
  if (h->impl.forward(nullptr, nullptr, B, L, nullptr, 0, nullptr, true, &need)) return 0;
  return need;
}
int sdmi_clip_forward(sdmi_clip* h, const int64_t* ids, float* out, int B, int L, void* workspace, int64_t workspace_bytes,
                      void* stream) {
  SDMI_CHECK(h && ids && out, "null argument");
  return h->impl.forward(ids, out, B, L, workspace, workspace_bytes, (hipStream_t)stream, false, nullptr);
}
int sdmi_k_attention_causal(const void* q, const void* k, const void* vt, void* out, int BH, int heads, int n, int n_pad,
                            int d, float scale, void* stream) {
  AttnParams a;
  a.q = (const f16*)q; a.k = (const f16*)k; a.vt = (const f16*)vt; a.out = (f16*)out;
  a.BH = BH; a.heads = heads; a.nq = n; a.nkv = n; a.nkv_pad = n_pad; a.d = d; a.scale = scale; a.causal = 1;
  return launch_attention(a, (hipStream_t)stream);
}

// ---- kernel-level entry points ------------------------------------------------------------------------
static int igemm_params_of(const sdmi_igemm_desc* d, IGemmParams& p) {
  p.a0 = (const f16*)d->a0; p.a1 = (const f16*)d->a1; p.a2 = (const f16*)d->a2;
  p.c0 = d->c0; p.c1 = d->c1; p.c2 = d->c2; p.lda0 = d->lda0; p.lda1 = d->lda1; p.lda2 = d->lda2;
  p.B = d->B; p.Hin = d->Hin; p.Win = d->Win; p.Hout = d->Hout; p.Wout = d->Wout;
  p.ksize = d->ksize; p.stride = d->stride; p.up = d->up; p.pad = d->asym_pad ? 0 : 1;
  p.w = (const f16*)d->w; p.M = d->B * d->Hout * d->Wout; p.N = d->N; p.K = d->ksize * d->ksize * (d->c0 + d->c1 + d->c2);
  p.mode = d->mode; p.bias = d->bias; p.rowvec = d->rowvec; p.ld_rowvec = d->ld_rowvec;
  p.residual = d->residual; p.ldr = d->ldr; p.out_f32 = d->out_f32; p.out_f16 = (f16*)d->out_f16; p.ldo = d->ldo;
  for (int i = 0; i < 3; ++i) { p.seg_dst[i] = (f16*)d->seg_dst[i]; p.seg_kind[i] = d->seg_kind[i]; }
  p.heads = d->heads; p.dh = d->dh; p.ntok = d->ntok; p.ntok_pad = d->ntok_pad; p.segC = d->segC;
  p.splitk = d->splitk; p.splitk_ws = d->splitk_ws; p.splitk_ws_floats = d->splitk_ws_floats;
  p.splitk_cnt = d->splitk_cnt; p.splitk_cnt_ints = d->splitk_cnt_ints;
  p.gn_n = d->gn_n;
  for (int i = 0; i < 2; ++i) { p.gn_acc[i] = (long long*)d->gn_acc[i]; p.gn_cpg[i] = d->gn_cpg[i]; p.gn_cbase[i] = d->gn_cbase[i]; }
  if (d->split16) {                      // a0 = hi, a1 = lo (not a channel concat), w = packed [N][3K]
    p.split16 = 1; p.c1 = 0; p.K = d->c0; p.ldw = 3 * d->c0;
  }
  p.f16_scale = d->f16_scale; p.lnp_out = d->lnp_out;
  p.lnf_part = d->lnf_part; p.lnf_npart = d->lnf_npart; p.lnf_eps = d->lnf_eps; p.lnf_cs = d->lnf_cs; p.lnf_d = d->lnf_d;
  p.pgn_gamma = d->pgn_gamma; p.pgn_beta = d->pgn_beta; p.pgn_eps = d->pgn_eps; p.pgn_silu = d->pgn_silu;
  p.pgn_out = (f16*)d->pgn_out; p.pgn_keep_f32 = d->pgn_keep_f32; p.pgn_applied = d->pgn_applied;
  p.out_lo = (f16*)d->out_lo;
  return zero_page(&p.zero_page);
}
int sdmi_k_igemm(const sdmi_igemm_desc* d, void* stream) {
  SDMI_CHECK(d, "null descriptor");
  IGemmParams p;
  if (igemm_params_of(d, p)) return -1;
  IGemmTune t; t.tile = d->tile; t.dma = d->dma;
  return launch_igemm(p, t, (hipStream_t)stream);
}
int sdmi_k_ff_tail(const sdmi_igemm_desc* proj_out, const void* ln_f16, const float* lnp, float ln_eps, const float* csd,
                   const void* wgg_f16, const void* wff2_f16, const float* bff2, const float* t, void* stream) {
  SDMI_CHECK(proj_out, "null descriptor");
  FfTailParams q;
  if (igemm_params_of(proj_out, q.epi)) return -1;
  SDMI_CHECK(proj_out->split16 && proj_out->ksize == 1, "ff_tail: the proj_out descriptor is a split-fp16 1x1 (w = sdmi_k_pack_split3)");
  q.ln = (const f16*)ln_f16; q.lnp = lnp; q.ln_eps = ln_eps; q.csd = csd; q.wgg = (const f16*)wgg_f16; q.wff2 = (const f16*)wff2_f16;
  q.bff2 = bff2; q.t = t; q.wpo = (const f16*)proj_out->w;
  return launch_ff_tail(q, (hipStream_t)stream);
}
int sdmi_k_st_tail(const sdmi_igemm_desc* proj_out, const void* a_f16, const void* wo_f16, const float* bo, float* t, const float* ln_gamma,
                   float ln_eps, const float* csd, const void* wgg_f16, const void* wff2_f16, const float* bff2, void* stream) {
  SDMI_CHECK(proj_out && a_f16, "null argument");
  FfTailParams q;
  if (igemm_params_of(proj_out, q.epi)) return -1;
  SDMI_CHECK(proj_out->split16 && proj_out->ksize == 1, "st_tail: the proj_out descriptor is a split-fp16 1x1 (w = sdmi_k_pack_split3)");
  q.a16 = (const f16*)a_f16; q.wo = (const f16*)wo_f16; q.bo = bo; q.ln_gamma = ln_gamma; q.ln_eps = ln_eps; q.csd = csd;
  q.wgg = (const f16*)wgg_f16; q.wff2 = (const f16*)wff2_f16; q.bff2 = bff2; q.t = t; q.wpo = (const f16*)proj_out->w;
  return launch_ff_tail(q, (hipStream_t)stream);
}
int sdmi_k_gn_conv3(const sdmi_igemm_desc* conv, const float* x0, const float* x1, int c0, int c1, float* gn_ws, int64_t gn_ws_floats,
                    const float* gn_gamma, const float* gn_beta, float gn_eps, void* stream) {
  SDMI_CHECK(conv && x0 && gn_ws && gn_gamma && gn_beta, "gn_conv3: null argument");
  GnConvParams q;
  if (igemm_params_of(conv, q.epi)) return -1;
  const int B = q.epi.B;
  SDMI_CHECK(gn_ws_floats >= gn_acc_words(B) * 2, "groupnorm workspace too small");
  SDMI_HIP_OK(hipMemsetAsync(gn_ws, 0, gn_acc_words(B) * sizeof(long long), (hipStream_t)stream));
  GroupNormParams g;
  g.x0 = x0; g.c0 = c0; g.x1 = x1; g.c1 = c1; g.B = B; g.HW = q.epi.Hout * q.epi.Wout; g.gamma = gn_gamma; g.beta = gn_beta; g.eps = gn_eps;
  g.stats_only = 1; g.acc = (long long*)gn_ws;
  if (launch_groupnorm(g, (hipStream_t)stream)) return -1;
  q.x0 = x0; q.x1 = x1; q.c0 = c0; q.c1 = c1; q.gn_acc = (const long long*)gn_ws; q.gn_gamma = gn_gamma; q.gn_beta = gn_beta; q.gn_eps = gn_eps;
  q.w = (const f16*)conv->w;
  return launch_gn_conv3(q, (hipStream_t)stream);
}
int sdmi_k_st_head(const float* x, float* gn_ws, int64_t gn_ws_floats, const float* gn_gamma, const float* gn_beta, float gn_eps,
                   const void* w_in3, const float* b_in, float* t, const float* ln_gamma, float ln_eps, const void* wqkv_f16,
                   const float* lnf_cs, const float* lnf_d, void* q, void* k, void* vt, int B, int ntok, int ntok_pad, int heads, int dh,
                   int C, void* stream) {
  SDMI_CHECK(x && gn_ws && gn_gamma && gn_beta, "st_head: null argument");
  SDMI_CHECK(gn_ws_floats >= gn_acc_words(B) * 2, "groupnorm workspace too small");
  SDMI_HIP_OK(hipMemsetAsync(gn_ws, 0, gn_acc_words(B) * sizeof(long long), (hipStream_t)stream));
  GroupNormParams g;
  g.x0 = x; g.c0 = C; g.B = B; g.HW = ntok; g.gamma = gn_gamma; g.beta = gn_beta; g.eps = gn_eps; g.stats_only = 1; g.acc = (long long*)gn_ws;
  if (launch_groupnorm(g, (hipStream_t)stream)) return -1;
  StHeadParams p;
  p.x = x; p.gn_acc = (const long long*)gn_ws; p.gn_gamma = gn_gamma; p.gn_beta = gn_beta; p.gn_eps = gn_eps;
  p.w_in = (const f16*)w_in3; p.b_in = b_in; p.t = t; p.ln_gamma = ln_gamma; p.ln_eps = ln_eps; p.wqkv = (const f16*)wqkv_f16;
  p.lnf_cs = lnf_cs; p.lnf_d = lnf_d; p.q = (f16*)q; p.k = (f16*)k; p.vt = (f16*)vt;
  p.M = B * ntok; p.B = B; p.ntok = ntok; p.ntok_pad = ntok_pad; p.heads = heads; p.dh = dh; p.C = C;
  return launch_st_head(p, (hipStream_t)stream);
}
int sdmi_k_st_mid(const void* a_f16, const void* wo_f16, const float* bo, float* t, const float* ln_gamma, float ln_eps, const void* wq_f16,
                  const float* lnf_cs, const float* lnf_d, void* q, int B, int ntok, int heads, int dh, int C, void* stream) {
  StHeadParams p;
  p.a16 = (const f16*)a_f16; p.w_in = (const f16*)wo_f16; p.b_in = bo; p.t = t; p.ln_gamma = ln_gamma; p.ln_eps = ln_eps;
  p.wqkv = (const f16*)wq_f16; p.lnf_cs = lnf_cs; p.lnf_d = lnf_d; p.q = (f16*)q;
  p.M = B * ntok; p.B = B; p.ntok = ntok; p.ntok_pad = ntok; p.heads = heads; p.dh = dh; p.C = C;
  return launch_st_mid(p, (hipStream_t)stream);
}
int sdmi_k_st_mid_ctx(const void* a_f16, const void* wo_f16, const float* bo, float* t, const float* ln_gamma, float ln_eps, const void* wq_f16,
                      const float* lnf_cs, const float* lnf_d, const void* ctx_k, const void* ctx_vt, int nkv, int nkv_pad, float scale, void* ao_out,
                      int B, int ntok, int heads, int dh, int C, void* stream) {
  StHeadParams p;
  p.a16 = (const f16*)a_f16; p.w_in = (const f16*)wo_f16; p.b_in = bo; p.t = t; p.ln_gamma = ln_gamma; p.ln_eps = ln_eps;
  p.wqkv = (const f16*)wq_f16; p.lnf_cs = lnf_cs; p.lnf_d = lnf_d;
  p.ctx_k = (const f16*)ctx_k; p.ctx_vt = (const f16*)ctx_vt; p.ctx_nkv = nkv; p.ctx_nkv_pad = nkv_pad; p.ctx_scale = scale; p.ao_out = (f16*)ao_out;
  SDMI_CHECK(ctx_k && ctx_vt && ao_out, "st_mid_ctx: null argument");
  p.M = B * ntok; p.B = B; p.ntok = ntok; p.ntok_pad = ntok; p.heads = heads; p.dh = dh; p.C = C;
  return launch_st_mid(p, (hipStream_t)stream);
}
int sdmi_k_ln_fold_prep(const void* w_f16, int N, int K, int ldw, const float* gamma, const float* beta, const float* bias,
                        float* cs, float* d, void* stream) {
  return launch_ln_fold_prep((const f16*)w_f16, N, K, ldw, gamma, beta, bias, cs, d, (hipStream_t)stream);
}
int sdmi_k_attention(const void* q, const void* k, const void* vt, void* out, int BH, int heads, int nq, int nkv,
                     int nkv_pad, int d, float scale, void* stream) {
  AttnParams a;
  a.q = (const f16*)q; a.k = (const f16*)k; a.vt = (const f16*)vt; a.out = (f16*)out;
  a.BH = BH; a.heads = heads; a.nq = nq; a.nkv = nkv; a.nkv_pad = nkv_pad; a.d = d; a.scale = scale;
  if (const char* e = getenv("SDMI_ATTN_NW")) a.nw = atoi(e);     // test / tuning knob
  return launch_attention(a, (hipStream_t)stream);
}
int sdmi_k_attention_ctx(const void* x, const void* wq, const void* k, const void* vt, void* out, int BH, int heads, int nq,
                         int nkv, int nkv_pad, int d, float scale, const float* lnf_part, float lnf_eps, const float* lnf_cs,
                         const float* lnf_d, void* stream) {
  AttnCtxParams a;
  a.x = (const f16*)x; a.wq = (const f16*)wq; a.k = (const f16*)k; a.vt = (const f16*)vt; a.out = (f16*)out;
  a.BH = BH; a.heads = heads; a.nq = nq; a.nkv = nkv; a.nkv_pad = nkv_pad; a.d = d; a.C = heads * d; a.scale = scale;
  if (lnf_part) { a.lnf_part = lnf_part; a.lnf_npart = a.C / 32; a.lnf_eps = lnf_eps; a.M = (BH / heads) * nq; a.lnf_cs = lnf_cs; a.lnf_d = lnf_d; }
  return launch_attention_ctx(a, (hipStream_t)stream);
}
int64_t sdmi_k_groupnorm_ws_floats(int B, int HW) { (void)HW; return gn_acc_words(B) * 2; }   // int64 words, counted in floats
int sdmi_k_groupnorm(const float* x0, const float* x1, int c0, int c1, int B, int HW, const float* gamma,
                     const float* beta, float eps, int silu, void* out_f16, float* out_f32, void* raw_f16, void* out_lo,
                     void* raw_lo, float* partial_ws, int64_t partial_floats, void* stream) {
  SDMI_CHECK(partial_floats >= gn_acc_words(B) * 2, "groupnorm workspace too small");
  // the workspace holds the fixed-point statistics accumulators; they must start at zero
  SDMI_HIP_OK(hipMemsetAsync(partial_ws, 0, gn_acc_words(B) * sizeof(long long), (hipStream_t)stream));
  GroupNormParams g;
  g.x0 = x0; g.x1 = x1; g.c0 = c0; g.c1 = c1; g.B = B; g.HW = HW; g.gamma = gamma; g.beta = beta; g.eps = eps;
  g.silu = silu; g.out_f16 = (f16*)out_f16; g.out_f32 = out_f32; g.raw_f16 = (f16*)raw_f16; g.out_lo = (f16*)out_lo; g.raw_lo = (f16*)raw_lo;
  g.acc = (long long*)partial_ws;
  return launch_groupnorm(g, (hipStream_t)stream);
}
int sdmi_k_conv3gn(const float* x0, const float* x1, int c0, int c1, int B, int H, int W, const float* gamma,
                   const float* beta, float eps, const void* w_packed, int N, const float* bias, const float* rowvec,
                   int ld_rowvec, const float* residual, int ldr, float* out, int ldo, int splitk, float* splitk_ws,
                   int64_t splitk_ws_floats, float* gn_ws, int64_t gn_ws_floats, int tile, void* raw_hi, void* raw_lo,
                   void* stream) {
  SDMI_CHECK(gn_ws_floats >= gn_acc_words(B) * 2, "groupnorm workspace too small");
  SDMI_HIP_OK(hipMemsetAsync(gn_ws, 0, gn_acc_words(B) * sizeof(long long), (hipStream_t)stream));
  GroupNormParams g;
  g.x0 = x0; g.x1 = x1; g.c0 = c0; g.c1 = c1; g.B = B; g.HW = H * W; g.gamma = gamma; g.beta = beta; g.eps = eps;
  g.stats_only = 1; g.acc = (long long*)gn_ws;
  if (launch_groupnorm(g, (hipStream_t)stream)) return -1;
  IGemmParams p;                       // the 3x3 convolution that normalises its own input (conv3halo.hip)
  p.xf0 = x0; p.xf1 = x1; p.c0 = c0; p.c1 = c1; p.lda0 = c0 + c1;
  p.gn_in_acc = (const long long*)gn_ws; p.gn_in_gamma = gamma; p.gn_in_beta = beta; p.gn_in_eps = eps; p.gn_in_silu = 1;
  p.raw_hi = (f16*)raw_hi; p.raw_lo = (f16*)raw_lo;
  p.B = B; p.Hin = p.Hout = H; p.Win = p.Wout = W; p.ksize = 3; p.stride = 1; p.up = 0; p.pad = 1;
  p.w = (const f16*)w_packed; p.M = B * H * W; p.N = N; p.K = 9 * (c0 + c1);
  p.bias = bias; p.rowvec = rowvec; p.ld_rowvec = ld_rowvec; p.residual = residual; p.ldr = ldr; p.out_f32 = out; p.ldo = ldo;
  p.splitk = splitk; p.splitk_ws = splitk_ws; p.splitk_ws_floats = splitk_ws_floats;
  if (zero_page(&p.zero_page)) return -1;
  IGemmTune t; t.tile = tile;
  return launch_igemm(p, t, (hipStream_t)stream);
}
int sdmi_k_layernorm(const float* x, const float* gamma, const float* beta, void* out_f16, int M, int C, float eps,
                     void* stream) {
  return launch_layernorm(x, gamma, beta, (f16*)out_f16, M, C, eps, (hipStream_t)stream);
}
int sdmi_k_cast_f16(const float* x, void* out_f16, void* out_lo, int64_t n, void* stream) {
  return launch_cast_f16(x, (f16*)out_f16, (f16*)out_lo, n, (hipStream_t)stream);
}
int sdmi_k_timestep_embedding(const int64_t* t_i64, const float* t_f32, float* out, int B, int dim, void* stream) {
  return launch_timestep_embedding(t_i64, t_f32, out, B, dim, (hipStream_t)stream);
}
int sdmi_k_small_linear(const float* in, int ld_in, const float* w, const float* bias, float* out, int ld_out, int B,
                        int N, int K, int silu_in, void* stream) {
  return launch_small_linear(in, ld_in, w, bias, out, ld_out, B, N, K, silu_in, (hipStream_t)stream);
}
int sdmi_k_conv_in(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int H, int W, int Cout,
                   void* stream) {
  return launch_conv_in(x, w, bias, out, B, Cin, H, W, Cout, (hipStream_t)stream);
}
int sdmi_k_conv_out(const float* h, const float* w, const float* bias, float* out, int B, int H, int W, int Cin, int Cout,
                    void* stream) {
  return launch_conv_out(h, w, bias, out, B, H, W, Cin, Cout, (hipStream_t)stream);
}
int sdmi_k_pack_conv_weight(const float* w, void* dst, int O, int I, int KH, int KW, void* stream) {
  return launch_pack_conv_weight(w, (f16*)dst, O, I, KH, KW, (hipStream_t)stream);
}
int sdmi_k_pack_conv_out(const float* w, float* dst, int O, int I, void* stream) {
  return launch_pack_conv_out(w, dst, O, I, (hipStream_t)stream);
}
int sdmi_k_pack_split3(const float* w, void* dst, int N, int K, void* stream) {
  return launch_pack_split3(w, (f16*)dst, N, K, (hipStream_t)stream);
}
int sdmi_k_pack_geglu(const float* w, const float* bias, void* wdst, float* bdst, int N, int K, void* stream) {
  return launch_pack_geglu(w, bias, (f16*)wdst, bdst, N, K, (hipStream_t)stream);
}
int sdmi_image_to_uint8(const float* img_nchw, void* out_nhwc_u8, int B, int C, int H, int W, void* stream) {
  return launch_image_u8(img_nchw, (unsigned char*)out_nhwc_u8, B, C, H, W, (hipStream_t)stream);
}
int sdmi_range_check(int enable) { return range_check_set(enable); }
int sdmi_range_report(char* buf, int buflen) {
  SDMI_CHECK(buf && buflen > 1, "null buffer");
  std::string js;
  if (range_report(&js)) return -1;
  SDMI_CHECK((int)js.size() + 1 <= buflen, "range report buffer too small");
  memcpy(buf, js.c_str(), js.size() + 1);
  return 0;
}
int sdmi_tune_begin(void) { return tune_begin(); }
int sdmi_tune_round(int r) { return tune_round(r); }
int sdmi_tune_end(const char* path, int* n_keys) { return tune_end(path, n_keys); }
int sdmi_tune_dump(char* buf, int buflen) {
  SDMI_CHECK(buf && buflen > 1, "null buffer");
  std::string txt;
  if (tune_dump(&txt)) return -1;
  SDMI_CHECK((int)txt.size() + 1 <= buflen, "tune dump buffer too small");
  memcpy(buf, txt.c_str(), txt.size() + 1);
  return 0;
}
int sdmi_profile_begin(void) { return prof_begin(); }
int sdmi_profile_end(char* buf, int buflen) {
  SDMI_CHECK(buf && buflen > 2, "null buffer");
  std::string js;
  if (prof_end(&js)) return -1;
  SDMI_CHECK((int)js.size() + 1 <= buflen, "profile buffer too small");
  memcpy(buf, js.c_str(), js.size() + 1);
  return 0;
}
int sdmi_k_prefetch_lines(const void* ptr, int64_t bytes, void* stream) {
  SDMI_CHECK(ptr != nullptr && bytes >= 0, "bad range");
  return launch_prefetch_lines(ptr, bytes, (hipStream_t)stream);
}
const void* sdmi_zero_page(void) {
  const f16* z = nullptr;
  if (zero_page(&z)) return nullptr;
  return z;
}

}  // extern "C"
