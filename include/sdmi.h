/* libsdmi -- MI355X (gfx950) native UNet eps-prediction + PLMS/DDIM step for Stable Diffusion v1.
 *
 * C ABI of the drop-in boundary.  The reference (CompVis/stable-diffusion) is pure Python and has no FFI
 * of its own; each entry point below names the reference call it stands in for (paths relative to the
 * reference root).  INTEGRATION.md shows the ctypes binding and the yaml `target:` switch that plugs it in.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; sdmi_last_error() gives the thread-local message.
 *   - all pointers are DEVICE pointers on the current HIP device unless stated otherwise;
 *     activations cross the boundary as contiguous fp32 in the reference's own layouts (NCHW latents,
 *     [B,L,D] context, int64 timesteps).
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); the library only enqueues
 *     work on it and never synchronises the device.
 *   - the library owns packed weights; the caller owns inputs, outputs and the workspace.
 */
#ifndef SDMI_H_
#define SDMI_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDMI_ABI_VERSION 17

typedef struct sdmi_unet sdmi_unet;

/* Constructor arguments of UNetModel used by configs/stable-diffusion/v1-inference.yaml:29-44
 * (ldm/modules/diffusionmodules/openaimodel.py:443-470).  use_spatial_transformer=True, legacy=False. */
typedef struct sdmi_unet_cfg {
  int32_t in_channels;
  int32_t out_channels;
  int32_t model_channels;
  int32_t num_res_blocks;
  int32_t n_levels;
  int32_t channel_mult[8];
  int32_t n_attention_resolutions;
  int32_t attention_resolutions[8];
  int32_t num_heads;
  int32_t transformer_depth;
  int32_t context_dim;
} sdmi_unet_cfg;

const char* sdmi_last_error(void);
int sdmi_abi_version(void);
/* 1 if the library was built with -DSDMI_EXPERIMENTS: the kernels that lost their same-box A/Bs (the GroupNorm-folding halo conv / split-fp16
 * GEMM / split-K reduction, the five-wave tile 22, attention with the to_q projection inside, the ping-pong attention schedule) and their
 * environment knobs are compiled in.  The product library (0) has neither; their entry points fail with a message that says so. */
int sdmi_has_experiments(void);

/* ---- UNet handle: replaces instantiate_from_config(unet_config) + load_state_dict ----------------------- */
/* UNetModel.__init__, openaimodel.py:443-692 */
int sdmi_unet_create(const sdmi_unet_cfg* cfg, sdmi_unet** out);
int sdmi_unet_destroy(sdmi_unet* h);
/* enumerate the state_dict keys the handle expects (= UNetModel.state_dict().keys()) */
int sdmi_unet_num_weights(const sdmi_unet* h);
int sdmi_unet_weight_info(const sdmi_unet* h, int idx, char* key_buf, int key_buf_len, int64_t* shape4, int* ndim);
/* model.load_state_dict (scripts/txt2img.py:56): fp32 tensor in the reference layout (conv OIHW, linear [out,in]);
 * `ptr` may be a device or a host pointer.  The library repacks (fp16 [N][K], K ordered (64-channel chunk, ky, kx, channel)) and keeps its own copy. */
int sdmi_unet_set_weight(sdmi_unet* h, const char* key, const float* ptr, const int64_t* shape, int ndim, void* stream);
/* fails (listing the first missing key) unless every expected tensor was set */
int sdmi_unet_finalize(sdmi_unet* h);

/* Packed-weight blob: the packed device buffers of a finalized handle, behind a header that pins the
 * configuration and ABI version.  `import` replaces every set_weight call + finalize (no fp32 checkpoint, no repack);
 * the buffers are host memory (e.g. an mmap of the file tools/pack_checkpoint.py writes). */
int64_t sdmi_unet_packed_bytes(sdmi_unet* h);
int sdmi_unet_export_packed(sdmi_unet* h, void* host_buf, int64_t bytes, void* stream);
int sdmi_unet_import_packed(sdmi_unet* h, const void* host_buf, int64_t bytes, void* stream);

/* bytes of scratch `sdmi_unet_forward` needs for this shape (0 on error) */
int64_t sdmi_unet_workspace_bytes(sdmi_unet* h, int B, int H, int W, int Lctx);

/* Cross-attention K/V of all SpatialTransformers depend only on the context (attention.py:174-176):
 * compute them once per prompt.  ctx: fp32 [B, Lctx, context_dim]. */
int sdmi_unet_cache_context(sdmi_unet* h, const float* ctx, int B, int Lctx, void* workspace, int64_t workspace_bytes,
                            void* stream);
/* The K / V^T caches are handle-owned device buffers of a fixed capacity: sdmi_unet_finalize reserves room for 8 rows x 77
 * context tokens (the SD-v1 prompt length at the largest batch one call takes).  sdmi_unet_forward NEVER allocates: a
 * context beyond the capacity fails with a message naming this call.  `reserve_context` grows the capacity (grow-only,
 * allocates: call it outside the sampling loop); sdmi_unet_cache_context grows it itself when needed. */
int sdmi_unet_reserve_context(sdmi_unet* h, int B, int Lctx);

/* The timestep path of UNetModel.forward -- timestep_embedding -> time_embed (openaimodel.py:723-724) -> every ResBlock's
 * emb_layers (openaimodel.py:218-224, 22 Linear layers = a 20160 x 1280 fp32 matrix for SD v1) -- depends on the timestep
 * only.  A sampler knows its timesteps in advance (plms.py:121-128, ddim.py:129-134): `cache_timesteps` computes the rows for
 * a list of INTEGER timesteps in one batch (the weights are read once per 8 timesteps instead of once per UNet call; same
 * kernels, rows are independent: bit-identical values), on `stream`, replacing any earlier table.  `hint_timestep(t)`
 * tells the NEXT sdmi_unet_forward that every row of its timestep tensor equals t: if t is in the table the call takes
 * the cached rows and launches nothing for the timestep path; otherwise (or without a hint) it computes them from its
 * timestep tensor as before.  The hint is the caller's assertion -- it is not checked against the device tensor -- and is
 * consumed by that one call.  t_host: host memory.  Setting weights drops the table. */
int sdmi_unet_cache_timesteps(sdmi_unet* h, const int64_t* t_host, int n, void* stream);
int sdmi_unet_hint_timestep(sdmi_unet* h, int64_t t);
/* Launch tapes (ABI 17; csrc/tape.h): sdmi_unet_forward records the launch list of a (B, H, W, Lctx, workspace, timestep mode, knobs) once and
 * replays it -- no dry pass, no table lookups, no descriptor fills -- patching only the caller's pointers (x, eps_out, timesteps, context, the
 * timestep-table row).  There is nothing to call: this reports how many forwards were replayed / recorded (tests, bench.py).  SDMI_REPLAY=0
 * turns the tapes off; SDMI_REPLAY_VERIFY=1 runs the executor on every would-be replay and fails if the patched tape differs from it.
 * No counterpart in the reference (pure Python, scripts/txt2img.py drives torch ops one by one). */
int sdmi_unet_tape_stats(sdmi_unet* h, int64_t* replayed, int64_t* recorded);

/* UNetModel.forward(x, timesteps, context) openaimodel.py:710-742, reached through
 * LatentDiffusion.apply_model ddpm.py:891-900,986-992 and DiffusionWrapper.forward ddpm.py:1402-1410.
 *   x        fp32 [B, in_channels, H, W] (NCHW, contiguous)
 *   t_i64 / t_f32   exactly one non-NULL: [B] timesteps (int64 as the samplers pass; fp32 for DPM-Solver)
 *   ctx      fp32 [B, Lctx, context_dim], or NULL to reuse sdmi_unet_cache_context()'s result
 *   eps_out  fp32 [B, out_channels, H, W] -- written fresh on every call */
int sdmi_unet_forward(sdmi_unet* h, const float* x, const int64_t* t_i64, const float* t_f32, const float* ctx,
                      float* eps_out, int B, int H, int W, int Lctx, void* workspace, int64_t workspace_bytes,
                      void* stream);

/* ---- sampler step: classifier-free guidance combine + PLMS / DDIM update in one launch -------------------
 * PLMSSampler.p_sample_plms plms.py:172-236, DDIMSampler.p_sample_ddim ddim.py:165-204.
 *   eps_model: model output; cfg != 0 -> rows [0,n) are the unconditional half, [n,2n) the conditional half
 *   mode: 0 e'=e_t | 1 (3e-o0)/2 | 2 (23e-16o0+5o1)/12 | 3 (55e-59o0+37o1-9o2)/24 | 4 (o0+e_t)/2
 *   a_t, a_prev, sigma, sqrt_1m_at: the four table entries the reference indexes per step
 *   noise (optional, sigma>0), e_t_out (optional: post-CFG eps for the history), pred_x0 (optional) */
int sdmi_sampler_step(const float* eps_model, int cfg, float scale, const float* x, int mode, const float* old0,
                      const float* old1, const float* old2, float a_t, float a_prev, float sigma, float sqrt_1m_at,
                      const float* noise, float* e_t_out, float* x_prev, float* pred_x0, int64_t n, void* stream);


/* DPM-Solver++ (2M) step as `scripts/txt2img.py --dpm_solver` runs it: classifier-free combine
 * (dpm_solver.py:340-346), data prediction m0 = (x - sigma_s e) / alpha_s (:386-399) and the multistep update
 * order 1: x_next = cx x - a m0 (:519-530);  order 2: x_next = cx x - a m0 - 0.5 a inv_r0 (m0 - m_prev) (:776-790).
 * m_out (optional) receives m0; x_next may be NULL (model value only). */
int sdmi_dpm_solver_step(const float* eps_model, int cfg, float scale, const float* x, const float* m_prev, float alpha_s,
                         float sigma_s, float cx, float a, float inv_r0, int order, float* m_out, float* x_next, int64_t n,
                         void* stream);

/* ---- first stage (AutoencoderKL): SURVEY.md 8 f-1 ---------------------------------------------------------------
 * Replaces instantiate_from_config(first_stage_config) (ddpm.py:462-467) for inference:
 * `decode_first_stage` (ddpm.py:705-763) -> AutoencoderKL.decode (autoencoder.py:330-333) -> Decoder.forward
 * (ldm/modules/diffusionmodules/model.py:528-568), and `encode_first_stage` (ddpm.py:825-863) ->
 * AutoencoderKL.encode (autoencoder.py:324-328) -> Encoder.forward (model.py:427-460). */
typedef struct sdmi_vae sdmi_vae;
/* ddconfig of configs/stable-diffusion/v1-inference.yaml:51-65 (attn_resolutions = [], dropout 0, double_z) + embed_dim */
typedef struct sdmi_vae_cfg {
  int32_t ch;
  int32_t out_ch;
  int32_t n_levels;
  int32_t ch_mult[8];
  int32_t num_res_blocks;
  int32_t in_channels;
  int32_t z_channels;
  int32_t embed_dim;
} sdmi_vae_cfg;
/* parts: 1 = decoder (+post_quant_conv), 2 = encoder (+quant_conv), 3 = both */
int sdmi_vae_create(const sdmi_vae_cfg* cfg, int parts, sdmi_vae** out);
int sdmi_vae_destroy(sdmi_vae* h);
/* the state_dict keys the handle expects (= AutoencoderKL.state_dict() of the chosen parts, without `loss.*`) */
int sdmi_vae_num_weights(const sdmi_vae* h);
int sdmi_vae_weight_info(const sdmi_vae* h, int idx, char* key_buf, int key_buf_len, int64_t* shape4, int* ndim);
int sdmi_vae_set_weight(sdmi_vae* h, const char* key, const float* ptr, const int64_t* shape, int ndim, void* stream);
int sdmi_vae_finalize(sdmi_vae* h);
/* z fp32 [B, embed_dim, H, W] -> img fp32 [B, out_ch, H*f, W*f], f = 2^(n_levels-1); z is multiplied by z_scale first
 * (decode_first_stage passes 1/scale_factor, ddpm.py:713; AutoencoderKL.decode passes 1) */
int64_t sdmi_vae_decode_workspace_bytes(sdmi_vae* h, int B, int H, int W);
int sdmi_vae_decode(sdmi_vae* h, const float* z, float z_scale, float* img, int B, int H, int W, void* workspace,
                    int64_t workspace_bytes, void* stream);
/* img fp32 [B, in_channels, H, W] (H, W multiples of f) -> moments fp32 [B, 2*embed_dim, H/f, W/f]: the parameters of
 * DiagonalGaussianDistribution (mean | logvar), ldm/modules/distributions/distributions.py:24-33 */
int64_t sdmi_vae_encode_workspace_bytes(sdmi_vae* h, int B, int H, int W);
int sdmi_vae_encode(sdmi_vae* h, const float* img, float* moments, int B, int H, int W, void* workspace,
                    int64_t workspace_bytes, void* stream);


/* ---- text encoder (FrozenCLIPEmbedder): SURVEY.md 8 f-2 ----------------------------------------------------------
 * Replaces `self.transformer(input_ids=tokens).last_hidden_state` of FrozenCLIPEmbedder.forward
 * (ldm/modules/encoders/modules.py:155-160); `self.transformer` is transformers' CLIPTextModel (transformers==4.19.2,
 * environment.yaml:25; models/clip/modeling_clip.py).  The tokenizer stays on the host. */
typedef struct sdmi_clip sdmi_clip;
/* CLIPTextConfig fields (openai/clip-vit-large-patch14: 49408, 768, 3072, 12, 12, 77); hidden_act = quick_gelu */
typedef struct sdmi_clip_cfg {
  int32_t vocab_size;
  int32_t hidden_size;
  int32_t intermediate_size;
  int32_t num_layers;
  int32_t num_heads;
  int32_t max_positions;
} sdmi_clip_cfg;
int sdmi_clip_create(const sdmi_clip_cfg* cfg, sdmi_clip** out);
int sdmi_clip_destroy(sdmi_clip* h);
/* state_dict keys of CLIPTextModel as transformers 4.19.2 names them (`text_model.embeddings...`, `text_model.encoder
 * .layers.N...`, `text_model.final_layer_norm...`), i.e. the checkpoint's `cond_stage_model.transformer.` sub-tree */
int sdmi_clip_num_weights(const sdmi_clip* h);
int sdmi_clip_weight_info(const sdmi_clip* h, int idx, char* key_buf, int key_buf_len, int64_t* shape4, int* ndim);
int sdmi_clip_set_weight(sdmi_clip* h, const char* key, const float* ptr, const int64_t* shape, int ndim, void* stream);
int sdmi_clip_finalize(sdmi_clip* h);
int64_t sdmi_clip_workspace_bytes(sdmi_clip* h, int B, int L);
/* ids: int64 [B, L] token ids (device); out: fp32 [B, L, hidden_size] = last_hidden_state */
int sdmi_clip_forward(sdmi_clip* h, const int64_t* ids, float* out, int B, int L, void* workspace, int64_t workspace_bytes,
                      void* stream);

/* ---- kernel-level entry points (parity tests and micro-benchmarks; same kernels the UNet uses) ---------- */
typedef struct sdmi_igemm_desc {
  const void* a0; const void* a1; const void* a2;   /* fp16 NHWC sources, channel concat [a0|a1|a2] (a1, a2 optional) */
  int32_t c0, c1, c2, lda0, lda1, lda2;
  int32_t B, Hin, Win, Hout, Wout, ksize, stride, up;
  const void* w;                        /* fp16 [N][K] from sdmi_k_pack_conv_weight: K = ksize*ksize*(c0+c1+c2), ordered (64-ch chunk, ky, kx, ch) */
  int32_t N;
  int32_t mode;                         /* 0 plain, 1 GEGLU (w/bias packed by sdmi_k_pack_geglu), 2 per-head scatter */
  const float* bias; const float* rowvec; int32_t ld_rowvec;
  const float* residual; int32_t ldr;
  float* out_f32; void* out_f16; int32_t ldo;
  void* seg_dst[3]; int32_t seg_kind[3];
  int32_t heads, dh, ntok, ntok_pad, segC;
  int32_t splitk;                       /* 1 none, 0 auto, >1 forced (plain mode; needs splitk_ws) */
  float* splitk_ws; int64_t splitk_ws_floats;   /* fp32 slabs: scratch, splitk * round_up(M, BM) * round_up(N, BN) floats cover every
                                                   layout (whole tiles in the MFMA register order -- the default since round 4,
                                                   SDMI_SLAB_TILED -- or [splitk][M][N]; see also splitk_cnt) */
  int32_t tile;                         /* -1 auto (tuning table); BMxBN/waves/LDS-DMA stages: 0 128x128/4/2, 1 128x64/4/2,
                                           2 64x64/4/2, 3 256x128/8/2, 4 128x64/4/3, 5 64x64/4/3, 6 256x128/8/3, 7 128x128/4/3,
                                           8 64x128/4/3, 9 128x128/8/3, 10 64x64/4/4, 11 128x256/8/2, 12 64x256/4/3, 13 256x64/4/3;
                                           halo-staged 3x3 conv (stride 1, pad 1, width 16/32/64, whole-row tiles):
                                           14 256x64/8/5, 15 256x128/8/3, 16 128x64/4/8, 17 128x128/4/5;
                                           deep rings (generic): 18 64x64/4/8, 19 64x128/4/6, 20 128x64/4/6, 21 128x128/8/4;
                                           22 64x160/5/5: five waves side by side (csrc/igemm5.hip; 1x1 / 3x3 stride 1-2, N % 160 == 0
                                           for the auto choice): M = 8192, N = 320 -> 256 workgroups = one per CU */
  int32_t dma;                          /* -1 default, 0 register staging, 1 LDS-DMA */
  int32_t asym_pad;                     /* 3x3 only: 0 = zero pad 1 on every side; 1 = pad right/bottom only, i.e.
                                           F.pad(x,(0,1,0,1)) + conv(padding=0) of the VAE Downsample (model.py:72-76) */
  /* optional (mode 0, Hout*Wout % 32 == 0): GroupNorm(32) statistics of the output for up to two consuming GroupNorms
   * (util.py:199-216), accumulated by the epilogue / the split-K reduce into int64 fixed-point words
   * gn_acc[t][B][32 groups][8 slots][16] (one 128-byte line per slot; words 0..3 = {sum int, sum frac * 2^40, sumsq int,
   * sumsq frac * 2^40}; zero them first);
   * the output is channels [gn_cbase, gn_cbase + N) of that GroupNorm's input, gn_cpg channels per group */
  int32_t gn_n; void* gn_acc[2]; int32_t gn_cpg[2]; int32_t gn_cbase[2];
  /* optional: one int per output tile, zero before the first launch (the kernel leaves them zero).  With it the split-K
   * reduction happens inside the GEMM (the last block of a tile to arrive sums the splits in index order and runs the
   * epilogue; no reduce launch) and splitk_ws holds splitk * round_up(M, BM) * round_up(N, BN) floats in register order;
   * without it: slabs [splitk][M][N] and a separate reduce kernel */
  int32_t* splitk_cnt; int32_t splitk_cnt_ints;
  /* split-fp16 dense GEMM (csrc/gemm_split16.hip; ksize 1, mode 0): a0 = high halves and a1 = low halves of the fp32
   * activation ([M][c0] each, sdmi_k_cast_f16), w = sdmi_k_pack_split3 ([N][3 c0] = [hi | hi | lo]);
   * out = a_hi w_hi^T + a_lo w_hi^T + a_hi w_lo^T from four operand tiles per 64-channel chunk.  tile: -1, 0, 1, 2, 4, 5, 8, 10 */
  int32_t split16;
  /* LayerNorm folded into the consuming GEMM (attention.py:211-215; csrc/common.h IGemmParams::lnp_out / lnf_*).
   * Producer (mode 0, out_f32 and out_f16, no split-K): out_f16 = fp16(f16_scale[n] * v) and lnp_out[(n / 32) * M + m] =
   * float2{sum, sum of squares} of row m over each 32-column block.  Consumer (dense, one source, no bias, no split-K): a0 =
   * that operand, lnf_part = its partials (lnf_npart = K / 32); every accumulator becomes
   * rstd_m * (acc - mean_m * lnf_cs[n]) + lnf_d[n] (vectors from sdmi_k_ln_fold_prep) before the mode's epilogue. */
  const float* f16_scale; float* lnp_out;
  const float* lnf_part; int32_t lnf_npart; float lnf_eps; const float* lnf_cs; const float* lnf_d;
  /* GroupNorm(32, pgn_eps) (+ SiLU) of the finished OUTPUT inside the split-K reduction (ABI 11; mode 0, N % 128 == 0, Hout*Wout *
   * N / 128 <= 5120: ResBlock conv1 -> out_layers.0-1, openaimodel.py:225-227): when the GEMM is split, one workgroup per (sample,
   * group) sums the slabs, takes the group's statistics and stores pgn_out [M][N] fp16 = SiLU(GN(v)); out_f32 is then written
   * only with pgn_keep_f32.  *pgn_applied (host int, optional) is set to 1 when that happened; when it is left 0 the launch was an
   * ordinary one and pgn_out is untouched. */
  const float* pgn_gamma; const float* pgn_beta; float pgn_eps; int32_t pgn_silu;
  void* pgn_out; int32_t pgn_keep_f32; int32_t* pgn_applied;
  /* optional (ABI 12; mode 0, ldo % 4 == 0): fp16(v - float(fp16(v))) beside out_f16 -- the low half of the split-fp16 operand a
   * following split16 GEMM reads (the last FF-out of a SpatialTransformer feeds proj_out this way) */
  void* out_lo;
} sdmi_igemm_desc;
int sdmi_k_igemm(const sdmi_igemm_desc* d, void* stream);
/* GEGLU -> FF-out -> proj_out of a SpatialTransformer as ONE launch (ABI 12; csrc/rowchain.hip; ldm/modules/attention.py:58-64,214,
 * 258-261): out = residual + proj_out(t + FF(norm3(t))) for C = 320 channels, a workgroup per 32 token rows.
 * proj_out: the split-fp16 1x1 descriptor of the last GEMM (split16 = 1, c0 = N = C, w = sdmi_k_pack_split3, bias, residual, out_f32,
 * optional out_f16 copy / gn_* statistics targets; B * Hout * Wout = M rows, Hout * Wout % 32 == 0) -- a0 / a1 are not read.
 * ln_f16 [M][C] = fp16(gamma3 * t) and lnp [C / 32][M][2]: what the producer of t stored (f16_scale / lnp_out above);
 * csd: the GEGLU projection's LayerNorm-fold column terms (sdmi_k_ln_fold_prep over the packed weights, bias inside d) regrouped per
 * hidden chunk of C: [4][cs of 2C packed columns | d of the same 2C]; wgg [8C][C], wff2 [C][4C] fp16, bff2 [C], t [M][C] fp32. */
int sdmi_k_ff_tail(const sdmi_igemm_desc* proj_out, const void* ln_f16, const float* lnp, float ln_eps, const float* csd,
                   const void* wgg_f16, const void* wff2_f16, const float* bff2, const float* t, void* stream);
/* ... with the out-projection of the cross-attention in front (ABI 14; the same kernel; attention.py:213, 191-192): t += a Wo^T + bo in place
 * (a [M][C] fp16 = the attn2 output rows, wo [C][C] fp16, t [M][C] fp32 in / out), then the chain above over norm3(t) with ln_gamma = the norm3
 * weight (its bias lives in csd).  The same bits as sdmi_k_igemm (bias, residual = t, out_f32 = t, f16_scale, lnp_out) -> sdmi_k_ff_tail. */
int sdmi_k_st_tail(const sdmi_igemm_desc* proj_out, const void* a_f16, const void* wo_f16, const float* bo, float* t, const float* ln_gamma,
                   float ln_eps, const float* csd, const void* wgg_f16, const void* wff2_f16, const float* bff2, void* stream);
/* ResBlock in_layers / out_layers as ONE launch (ABI 15; csrc/gnconv.hip; ldm/modules/diffusionmodules/openaimodel.py:201-204, 225-231,
 * util.py:199-216): out = conv3x3(SiLU(GroupNorm32(cat(x0, x1)))) + bias (+ rowvec, + residual) for N = 320 output channels, a workgroup per
 * 32 pixels of an image row x all output channels (the input halo is normalised once per workgroup).
 * conv: the descriptor of the stride-1 3x3 convolution (ksize 3, c0 + c1 + c2 = Cin input channels, B, Hin = Hout, Win = Wout % 32 == 0,
 * w = sdmi_k_pack_conv, N = 320, mode 0, bias / rowvec / residual / out_f32 / out_f16 / gn_* as for sdmi_k_igemm) -- a0 .. a2 are not read;
 * x0 [B * H * W][c0], x1 [..][c1] (or NULL, c1 = 0) fp32 NHWC, 256 <= c0 + c1 <= 960, c0 % 64 == 0; gn_ws: sdmi_k_groupnorm_ws_floats(B, H * W)
 * floats of scratch (the statistics are computed by a statistics launch in front).  The same bits as sdmi_k_groupnorm (silu, fp16 out) ->
 * sdmi_k_igemm with splitk = 1. */
int sdmi_k_gn_conv3(const sdmi_igemm_desc* conv, const float* x0, const float* x1, int c0, int c1, float* gn_ws, int64_t gn_ws_floats,
                    const float* gn_gamma, const float* gn_beta, float gn_eps, void* stream);
/* GroupNorm-apply -> proj_in -> q | k | v of a SpatialTransformer as ONE launch (ABI 13; csrc/rowchain.hip; ldm/modules/attention.py:254-256,
 * 212, 170-176): t = proj_in(GroupNorm(x)) + b_in (fp32 [M][C], M = B * ntok), q | k | v = norm1(t) Wqkv^T scattered per head
 * (q, k: [B * heads][ntok][dh] fp16, vt: [B * heads][dh][ntok_pad] fp16) for C = 320 channels, a workgroup per 32 token rows.
 * x [M][C] fp32; gn_ws: sdmi_k_groupnorm_ws_floats(B, ntok) floats of scratch (the statistics are computed here, as sdmi_k_groupnorm does);
 * w_in3 = sdmi_k_pack_split3(proj_in weight) [C][3C]; ln_gamma = norm1 weight; wqkv [3C][C] fp16 (to_q | to_k | to_v rows);
 * lnf_cs / lnf_d [3C] = sdmi_k_ln_fold_prep(wqkv, norm1 weight, norm1 bias).  Same arithmetic, in the same order, as sdmi_k_groupnorm
 * (f16 + lo outputs) -> sdmi_k_igemm (split16, f16_scale, lnp_out) -> sdmi_k_igemm (mode 2, lnf_*): the outputs are the same bits. */
int sdmi_k_st_head(const float* x, float* gn_ws, int64_t gn_ws_floats, const float* gn_gamma, const float* gn_beta, float gn_eps,
                   const void* w_in3, const float* b_in, float* t, const float* ln_gamma, float ln_eps, const void* wqkv_f16,
                   const float* lnf_cs, const float* lnf_d, void* q, void* k, void* vt, int B, int ntok, int ntok_pad, int heads, int dh,
                   int C, void* stream);
/* The middle of a BasicTransformerBlock as ONE launch (ABI 13; same kernel; attention.py:212-213, 170, 191-192): t += a Wo^T + bo in place
 * (a [M][C] fp16 = the self-attention output, wo [C][C] fp16, t [M][C] fp32), q = norm2(t) Wq^T scattered per head ([B * heads][ntok][dh] fp16);
 * ln_gamma = norm2 weight, lnf_cs / lnf_d [C] = sdmi_k_ln_fold_prep(wq, norm2 weight, norm2 bias).  The same bits as sdmi_k_igemm (bias,
 * residual = t, out_f32 = t, f16_scale, lnp_out) -> sdmi_k_igemm (mode 2, lnf_*). */
int sdmi_k_st_mid(const void* a_f16, const void* wo_f16, const float* bo, float* t, const float* ln_gamma, float ln_eps, const void* wq_f16,
                  const float* lnf_cs, const float* lnf_d, void* q, int B, int ntok, int heads, int dh, int C, void* stream);
/* ... with the cross-attention behind to_q inside the same launch (ABI 16; attention.py:213, 170-193): ctx_k [B * heads][nkv][dh], ctx_vt
 * [B * heads][dh][nkv_pad] fp16 (the cached context projections, sdmi_k_igemm mode 2), scale = dh^-1/2, ao_out [B * ntok][C] fp16 = the attention
 * output rows (q is not written).  dh = 40, nkv <= 128.  The same bits as sdmi_k_st_mid -> sdmi_k_attention. */
int sdmi_k_st_mid_ctx(const void* a_f16, const void* wo_f16, const float* bo, float* t, const float* ln_gamma, float ln_eps, const void* wq_f16,
                      const float* lnf_cs, const float* lnf_d, const void* ctx_k, const void* ctx_vt, int nkv, int nkv_pad, float scale, void* ao_out,
                      int B, int ntok, int heads, int dh, int C, void* stream);
/* cs[n] = sum_k gamma[k] * w[n][k], d[n] = sum_k beta[k] * w[n][k] (+ bias[n]) over the PACKED fp16 weights w [N][ldw]
 * (first K columns of a row): the column terms of a GEMM that folds LayerNorm(gamma, beta) of its input rows */
int sdmi_k_ln_fold_prep(const void* w_f16, int N, int K, int ldw, const float* gamma, const float* beta, const float* bias,
                        float* cs, float* d, void* stream);
/* q [BH,nq,d], k [BH,nkv,d], vt [BH,d,nkv_pad] fp16 -> out fp16 [BH/heads, nq, heads*d]; attention.py:178-192 */
int sdmi_k_attention(const void* q, const void* k, const void* vt, void* out, int BH, int heads, int nq, int nkv,
                     int nkv_pad, int d, float scale, void* stream);
/* cross-attention with the to_q projection inside the kernel (csrc/attn_ctx.hip; attention.py:161,170-193): x fp16 [B*nq][C] token
 * rows, wq fp16 [C][C] (rows = head * d + dd), k / vt as above with nkv <= 128 -> out fp16 [B, nq, C]; d in {40, 80, 160}.
 * Optional LayerNorm fold (lnf_part != NULL): x = fp16(gamma * t), partials [C / 32][B * nq][2], cs / d from sdmi_k_ln_fold_prep. */
int sdmi_k_attention_ctx(const void* x, const void* wq, const void* k, const void* vt, void* out, int BH, int heads, int nq,
                         int nkv, int nkv_pad, int d, float scale, const float* lnf_part, float lnf_eps, const float* lnf_cs,
                         const float* lnf_d, void* stream);
/* same with a causal mask (query i attends to keys <= i; nq == nkv): CLIPTextModel's self-attention */
int sdmi_k_attention_causal(const void* q, const void* k, const void* vt, void* out, int BH, int heads, int n, int n_pad,
                            int d, float scale, void* stream);
/* GroupNorm(32) over cat(x0,x1) fp32 NHWC; any of the outputs may be NULL.  out_lo / raw_lo = fp16(v - fp16(v)):
 * the low halves of split-fp16 operands (3-pass 1x1 convs) */
int sdmi_k_groupnorm(const float* x0, const float* x1, int c0, int c1, int B, int HW, const float* gamma,
                     const float* beta, float eps, int silu, void* out_f16, float* out_f32, void* raw_f16, void* out_lo,
                     void* raw_lo, float* partial_ws, int64_t partial_floats, void* stream);
int64_t sdmi_k_groupnorm_ws_floats(int B, int HW);
/* GroupNorm(32, eps) + SiLU folded into the staging of a 3x3 convolution (stride 1, pad 1) over cat(x0, x1) fp32 NHWC ->
 * out fp32 [B*H*W][N] (+bias[n] +rowvec[b][n] +residual): ResBlock in_layers / out_layers, openaimodel.py:201-204,225-231.
 * Runs the statistics kernel, then the halo-staged convolution that normalises its own input (csrc/conv3halo.hip);
 * w_packed from sdmi_k_pack_conv_weight; power-of-two W in 8..64, (c0 + c1) / 32 >= 8; tile = -1 (auto) or a halo tile id
 * 14..17; raw_hi / raw_lo (optional): split-fp16 copy of the raw input [B*H*W][c0 + c1]. */
int sdmi_k_conv3gn(const float* x0, const float* x1, int c0, int c1, int B, int H, int W, const float* gamma,
                   const float* beta, float eps, const void* w_packed, int N, const float* bias, const float* rowvec,
                   int ld_rowvec, const float* residual, int ldr, float* out, int ldo, int splitk, float* splitk_ws,
                   int64_t splitk_ws_floats, float* gn_ws, int64_t gn_ws_floats, int tile, void* raw_hi, void* raw_lo,
                   void* stream);
int sdmi_k_layernorm(const float* x, const float* gamma, const float* beta, void* out_f16, int M, int C, float eps,
                     void* stream);
int sdmi_k_cast_f16(const float* x, void* out_f16, void* out_lo, int64_t n, void* stream);
int sdmi_k_timestep_embedding(const int64_t* t_i64, const float* t_f32, float* out, int B, int dim, void* stream);
int sdmi_k_small_linear(const float* in, int ld_in, const float* w, const float* bias, float* out, int ld_out, int B,
                        int N, int K, int silu_in, void* stream);
int sdmi_k_conv_in(const float* x_nchw, const float* w_oihw, const float* bias, float* out_nhwc, int B, int Cin, int H,
                   int W, int Cout, void* stream);
int sdmi_k_conv_out(const float* h_nhwc, const float* w_ohwi, const float* bias, float* out_nchw, int B, int H, int W,
                    int Cin, int Cout, void* stream);
int sdmi_k_pack_conv_weight(const float* w_oihw, void* dst_f16, int O, int I, int KH, int KW, void* stream);
int sdmi_k_pack_conv_out(const float* w_oihw, float* dst_ohwi, int O, int I, void* stream);
/* [N][K] fp32 -> fp16 [N][3K] = [hi | hi | lo] for the 3-pass split-fp16 1x1 convs */
int sdmi_k_pack_split3(const float* w, void* dst_f16, int N, int K, void* stream);
/* first-stage helpers: 1x1 conv NCHW->NCHW on <= 16 channels (input pre-scaled); row softmax fp32 -> fp16 */
int sdmi_k_pointwise_nchw(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int HW,
                          float in_scale, void* stream);
int sdmi_k_softmax_rows(const float* S, void* P_f16, int rows, int cols, float scale, void* stream);
int sdmi_k_pack_geglu(const float* w, const float* bias, void* wdst_f16, float* bdst, int N, int K, void* stream);
/* Host post-processing of scripts/txt2img.py:313-324, on the device: img fp32 [B, C, H, W] (the
 * decode_first_stage output) -> uint8 [B, H, W, C] = astype(uint8)(255 * clamp((img + 1) / 2, 0, 1)), bit-identical to
 * the reference's torch / numpy ops. */
int sdmi_image_to_uint8(const float* img_nchw, void* out_nhwc_u8, int B, int C, int H, int W, void* stream);
/* fp16 range guard (debug; also SDMI_CHECK_RANGE=1 in the environment): after every launch that writes fp16 activations
 * (MFMA operands: GroupNorm / LayerNorm outputs, q / k / v^T, GEGLU, attention output, fp16 copies of the residual
 * stream) the buffer is scanned.  The reference has the same exposure under torch.autocast (scripts/txt2img.py:283);
 * this makes an overflow visible per launch.  Synchronises the stream after each launch: never on in a timed run.
 * sdmi_range_check(enable) also clears the counters; sdmi_range_report writes
 * {"over_6e4": n, "nonfinite": m, "max_abs": x, "first": "<first offending kernel class>"}. */
int sdmi_range_check(int enable);
int sdmi_range_report(char* json_buf, int json_buf_len);
/* In-situ tuning of the implicit-GEMM tile / split-K choice (no reference counterpart: the reference delegates to
 * MIOpen / hipBLASLt heuristics).  begin -> for r in rounds: sdmi_tune_round(r), run the workloads -> end: every
 * auto-configured GEMM launch between begin and end runs candidate (r mod #candidates) of its shape, timed with HIP
 * events on its stream; end folds the timings into the table and writes it (path NULL = next to libsdmi.so, where it
 * is loaded from at start-up).  sdmi_tune_dump: per-candidate timings of the last collection as text. */
int sdmi_tune_begin(void);
int sdmi_tune_round(int r);
int sdmi_tune_end(const char* path, int* n_keys);
int sdmi_tune_dump(char* buf, int buflen);
/* per-launch timing of the library's kernels (HIP events on the launch stream): begin, run forwards, then end
 * writes a JSON array [{"name","launches","ms","flops","bytes"}] (algorithmic flops / bytes per kernel class) */
int sdmi_profile_begin(void);
int sdmi_profile_end(char* json_buf, int json_buf_len);
/* cache hint used by experiments (tools/bench_prefetch.py): touch every 128-byte line of a device range on `stream` */
int sdmi_k_prefetch_lines(const void* ptr, int64_t bytes, void* stream);
/* a device buffer of >= 256 zero bytes owned by the library (out-of-image conv taps read it) */
const void* sdmi_zero_page(void);

#ifdef __cplusplus
}
#endif
#endif /* SDMI_H_ */
