// Shared device/host helpers for the MI355X (gfx950) kernels of libsdmi.
// wave = 64 lanes, MFMA 32x32x16 f16 -> f32, LDS tiles read as ds_read_b128.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <string>

#include "tape.h"

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef __HIPCC__
// The GroupNorm(32) (+ SiLU) arithmetic of one element, shared by the stand-alone apply kernel (norm.hip) and the convolution that
// applies it while staging its input (conv3halo.hip): one expression, written operation by operation, so both produce the same bits.
__device__ __forceinline__ float gn_apply_elem(float v, float mean, float rstd, float gamma, float beta, int silu) {
  float t = __builtin_fmaf((v - mean) * rstd, gamma, beta);
  if (silu) t = t * __builtin_amdgcn_rcpf(1.0f + __expf(-t));
  return t;
}
// GroupNorm statistics words (see GroupNormParams::acc): add one fp32 partial / read a folded total
__device__ __forceinline__ void gn_acc_add(unsigned long long* dst, float v) {
  const double d = (double)v;
  const double hi = rint(d);
  atomicAdd(dst, (unsigned long long)(long long)hi);
  atomicAdd(dst + 1, (unsigned long long)__double2ll_rn((d - hi) * 1099511627776.0));
}
__device__ __forceinline__ double gn_acc_value(long long hi, long long lo) {
  return (double)hi + (double)lo * (1.0 / 1099511627776.0);
}
// the two words gn_acc_add adds for one fp32 partial (same expressions: a kernel that sums its partials in registers instead of
// through the accumulator words arrives at the same integers)
__device__ __forceinline__ void gn_fixed_split(float v, long long* hi, long long* lo) {
  const double d = (double)v;
  const double h = rint(d);
  *hi = (long long)h;
  *lo = __double2ll_rn((d - h) * 1099511627776.0);
}
// {mean, rstd} of a group from its folded totals {sum, sum of squares} (integer + 2^-40 fraction words), n elements: ONE expression
// for every kernel that normalises (norm.hip's apply kernel, the split-K reduction that applies a GroupNorm) -- same bits
__device__ __forceinline__ void gn_mean_rstd(long long s, long long sl, long long ss, long long ssl, double n, float eps,
                                             float* mean, float* rstd) {
  const double m = gn_acc_value(s, sl) / n;
  double var = gn_acc_value(ss, ssl) / n - m * m;
  if (var < 0.0) var = 0.0;
  *mean = (float)m;
  *rstd = (float)(1.0 / sqrt(var + (double)eps));
}
#endif

namespace sdmi {

// thread-local error text surfaced through sdmi_last_error()
void set_error(const std::string& msg);
int fail(const std::string& msg);   // sets the error, returns -1

#define SDMI_HIP_OK(expr)                                                                    \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return ::sdmi::fail(std::string(#expr) + ": " + hipGetErrorString(_e));                \
  } while (0)

#define SDMI_CHECK(cond, msg)                                                                \
  do {                                                                                       \
    if (!(cond)) return ::sdmi::fail(std::string("check failed: ") + #cond + " -- " + (msg)); \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// ----------------------------------------------------------------------------------------------
// Implicit-GEMM descriptor:  out[M,N] (+epilogue) = gatherA[M,K] * W[N,K]^T
//   A is fp16 NHWC activations (one or two channel-concatenated sources), gathered as a
//   1x1 or 3x3 (stride 1/2, optional nearest-x2 upsampled input) convolution; a Linear is ksize=1.
//   W is fp16 [N][K], K ordered (64-channel chunk, ky, kx, channel within chunk); for 1x1 / Linear simply [N][Cin]
//   -- packed once by small.hip.
// ----------------------------------------------------------------------------------------------
enum EpiMode { EPI_PLAIN = 0, EPI_GEGLU = 1, EPI_HEADS = 2 };

// Output stores of the big producers (GEMM epilogues, split-K reduce, GroupNorm-apply, LayerNorm, attention): nothing inside the
// producing kernel reads them back, the next kernel does.  -DSDMI_NT_STORES marks them non-temporal (experiment: does streaming
// them out shorten the write-back at the kernel boundary?  SDMI_CXXFLAGS=-DSDMI_NT_STORES SDMI_LIB_OUT=libsdmi_nt.so build.py).
#ifdef SDMI_NT_STORES
#define SDMI_ST(T, ptr, val) __builtin_nontemporal_store((T)(val), (T*)(ptr))
#else
#define SDMI_ST(T, ptr, val) (*(T*)(ptr) = (T)(val))
#endif
// Cache policy of the WEIGHT tiles' LDS-DMA loads in the GEMM kernels (igemm / conv3halo / gemm_split16): aux bits of
// raw_ptr_buffer_load_lds, 0 = default, 2 = nt (MI355X_MICROARCH.md "nt-weights": issued -> landed -18 % on a stream that one CU
// reads once, -6 % end to end where every CU re-reads the slices).  -DSDMI_W_AUX=2 is a build experiment (round 4).
#ifndef SDMI_W_AUX
#define SDMI_W_AUX 0
#endif
// Write-through output stores (round 4, default): the 16-byte fp32 output stores of the GEMM epilogues, the split-K reduce and
// GroupNorm-apply carry the sc1 bit, i.e. they are written THROUGH the XCD's L2 while the kernel runs instead of sitting dirty in
// it until the end-of-kernel write-back (MI355X_MICROARCH.md: a boundary costs + dirty bytes / 6 TB/s; `nt` is not write-through,
// sc1 is; a 16-byte sc1 store costs what a plain one does, narrower ones are one fabric write each).  Same values, same addresses:
// bit-identical.  Same-box A/B (profiles/wt_stores_r04.txt): -0.03 ms per UNet call (-0.5 %), first-stage decode unchanged;
// -DSDMI_WT_STORES=2 (the 8-byte fp16 stores as well) measured the same as 1, -DSDMI_WT_STORES=0 = plain stores.
// base = wave-uniform tensor base, off = element offset.  The buffer instruction takes a 32-bit byte offset checked against 2^31
// records: a lane whose byte offset does not fit (outputs beyond 2 GB: first-stage decodes at 768 x 768 with 4+ images, 1024 x 1024
// with 3+) takes the plain 64-bit store instead -- same value, same address; a store past 2^31 would otherwise be dropped by the range
// check and one past 2^32 would wrap onto an earlier sample.
#ifndef SDMI_WT_STORES
#define SDMI_WT_STORES 1
#endif
#if defined(__HIPCC__) && SDMI_WT_STORES >= 1
typedef unsigned sdmi_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned sdmi_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sdmi_st_wt16(const void* base, size_t byte_off, f32x4 v) {
  if (__builtin_expect(byte_off < (size_t)0x80000000u - 16, 1)) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)0x80000000, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sdmi_u32x4, v), r, (int)byte_off, 0, 16);
  } else {
    *(f32x4*)((char*)base + byte_off) = v;
  }
}
__device__ __forceinline__ void sdmi_st_wt8(const void* base, size_t byte_off, f16x4 v) {
  if (__builtin_expect(byte_off < (size_t)0x80000000u - 16, 1)) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)0x80000000, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(sdmi_u32x2, v), r, (int)byte_off, 0, 16);
  } else {
    *(f16x4*)((char*)base + byte_off) = v;
  }
}
#define SDMI_ST_F32X4(base, off, val) sdmi_st_wt16((base), (size_t)(off) * 4, (val))
#if SDMI_WT_STORES >= 2
#define SDMI_ST_F16X4(base, off, val) sdmi_st_wt8((base), (size_t)(off) * 2, (val))
#else
#define SDMI_ST_F16X4(base, off, val) SDMI_ST(f16x4, (base) + (off), (val))
#endif
#else
#define SDMI_ST_F32X4(base, off, val) SDMI_ST(f32x4, (base) + (off), (val))
#define SDMI_ST_F16X4(base, off, val) SDMI_ST(f16x4, (base) + (off), (val))
#endif

struct IGemmParams {
  const f16* a0 = nullptr; const f16* a1 = nullptr; const f16* a2 = nullptr;   // A sources, channel concat [a0 | a1 | a2]
  int c0 = 0, c1 = 0, c2 = 0;                          // channels taken from each source (Cin = c0 + c1 + c2)
  int lda0 = 0, lda1 = 0, lda2 = 0;                    // row pitch (elements) of each source
  int B = 1, Hin = 1, Win = 1;                         // source spatial dims (M = B*Hout*Wout)
  int Hout = 1, Wout = 1;
  int ksize = 1, stride = 1, up = 0;
  int pad = 1;                                         // 3x3 only: 1 = symmetric zero pad; 0 = pad right/bottom only (taps at +0..+2)
  const f16* w = nullptr;                              // [N][K]
  int M = 0, N = 0, K = 0;
  // K of the reference op when the descriptor executes more (the K-concatenated 3-pass split-fp16 1x1 convs: K = 3 * k_alg);
  // 0 = K.  Only the profiler's algorithmic FLOP count reads it.
  int k_alg = 0;
  // split-fp16 dense GEMM (gemm_split16.hip): a0 = high halves, a1 = low halves of the activation ([M][K], pitch lda0), w =
  // packed [N][ldw >= 3 K] = [hi | hi | lo]; out = a_hi w_hi^T + a_lo w_hi^T + a_hi w_lo^T.  K = the reference op's K.
  int split16 = 0;
  int ldw = 0;                                         // weight row pitch (elements) of the split-fp16 GEMM
  // epilogue
  int mode = EPI_PLAIN;
  const float* bias = nullptr;                         // [N]
  const float* rowvec = nullptr; int ld_rowvec = 0;    // per-batch vector [B][ld_rowvec] added to every row of batch b
  const float* residual = nullptr; int ldr = 0;        // fp32 [M][ldr]
  float* out_f32 = nullptr; f16* out_f16 = nullptr; int ldo = 0;   // either / both
  f16* out_lo = nullptr;                               // optional (plain mode): fp16(v - float(fp16(v))), the low half of a split-fp16 operand
  // optional (plain mode): LayerNorm of the finished output rows -> ln_out fp16 [M][N] (needs out_f32 with ldo == N);
  // launch_igemm issues the layernorm kernel after the GEMM (and its split-K reduce).  A reduce kernel with the
  // LayerNorm folded in (one wave per row) was measured and was no faster than the two launches.
  const float* ln_gamma = nullptr; const float* ln_beta = nullptr; f16* ln_out = nullptr; float ln_eps = 1e-5f;
  // ---- LayerNorm folded into the CONSUMING GEMM (attention.py:211-215: x + attn1(norm1(x)), ... -- no LayerNorm launch) --------
  //   sum_k LN(x)_k W_nk = rstd_m * ( sum_k (gamma_k x_k) W_nk  -  mean_m * cs_n ) + d_n,   cs_n = sum_k gamma_k W_nk,
  //                                                                                          d_n  = sum_k beta_k W_nk (+ bias_n)
  // Producer side (plain mode, the GEMM whose output rows x a LayerNorm reads): beside the fp32 stream it stores the operand
  // fp16(gamma_n * x) through out_f16 (f16_scale = gamma: one rounding, as the stand-alone kernel rounds LN(x) once) and the
  // {sum, sum of squares} of every row over each 32-column block of its tile: lnp_out[(n / 32) * M + m] (float2).  Needs the
  // 16-byte epilogue (full tiles inside one sample, N % BN == 0: the launcher picks such a tile) and no split-K.
  const float* f16_scale = nullptr;                    // [N] fp32, multiplies the out_f16 copy only
  float* lnp_out = nullptr;                            // [N / 32][M][2] fp32 row-statistics partials
  // Consumer side (any mode): a0 = that fp16(gamma * x) operand [M][K]; lnf_part = its row partials ([lnf_npart = K / 32][M][2]),
  // folded per row in the prologue (fp64) into {mean, rstd}; the epilogue turns every accumulator into
  // rstd_m * (acc - mean_m * lnf_cs[n]) + lnf_d[n] before its mode-specific part (bias must be NULL: it is inside lnf_d).
  const float* lnf_part = nullptr; int lnf_npart = 0; float lnf_eps = 1e-5f;
  const float* lnf_cs = nullptr; const float* lnf_d = nullptr;
  // EPI_HEADS: N = nseg * C, column n -> segment n / C, head (n % C) / dh, dd = n % dh
  //   seg_kind 0: row layout   dst[((b*heads + head) * ntok + tok) * dh + dd]
  //   seg_kind 1: transposed   dst[((b*heads + head) * dh + dd) * ntok_pad + tok]
  f16* seg_dst[3] = {nullptr, nullptr, nullptr};
  int seg_kind[3] = {0, 0, 0};
  int heads = 0, dh = 0, ntok = 0, ntok_pad = 0, segC = 0;
  // split-K (plain mode only): every split stores its partial tile into a private fp32 slab of splitk_ws
  // ([split][M][N]); a second kernel sums the slabs in a fixed order and applies the epilogue (deterministic).
  int splitk = 1;                                      // 1 none, 0 auto, >1 forced
  float* splitk_ws = nullptr; int64_t splitk_ws_floats = 0;
  // fused reduction (default when splitk_cnt is given): every split stores its accumulators in register order, takes a
  // ticket from the tile's counter, and the LAST block to arrive sums the splits in index order and runs the normal
  // epilogue -- no reduce kernel.  Counters: one int per output tile, zero before the first launch (the last block resets
  // its counter).  Slabs then need splitk * round_up(M, BM) * round_up(N, BN) floats.
  int* splitk_cnt = nullptr; int splitk_cnt_ints = 0;
  int gn_safe = 0;                                     // debugging (SDMI_GN_SAFE=1): the GroupNorm-folding conv drains the queue at every counted wait
#ifdef SDMI_IGEMM_TIMING
  long long* dbg_times = nullptr;                      // timing build only: 6 s_memtime slots per workgroup (5 used)
  int dbg_abl = 0;                                     // timing build only (SDMI_EPI_ABL): 1 no residual loads, 2 no GroupNorm statistics, 4 no output stores
#endif
  int splitk_fused = 0;                                // set by the launcher
  // set by the launcher (round 4): the unfused split-K slabs hold whole TILES in the MFMA register order -- slab float index
  // ((split * ntiles + tile) * BM * BN) + ((i * TN + j) * 4 + r4) * (NT * 4) + thread * 4 + e -- instead of [split][M][N]: a lane
  // stores its 4 consecutive accumulator rows of a column as ONE 16-byte write-through store (a wave writes 1 KB runs; the
  // row-major layout takes 4-byte stores, 128-byte runs), and splitk_reduce_tiled_kernel turns 4 x 4 blocks between 4 adjacent
  // lanes back into row-major quads.  slab_bm / bn / wm / wn: the tile geometry the reduction needs to decode it.
  int slab_tiled = 0, slab_bm = 0, slab_bn = 0, slab_wm = 0, slab_wn = 0;
  int slab_sh_qpt = 0, slab_sh_nt = 0, slab_sh_tn = 0, slab_sh_wn = 0;      // log2 of BM * BN / 4, threads, TN, WARPS_N
  int epi_vec = 0;                                     // set by the launcher: 16-byte epilogue (pointer / pitch alignment checked there)
  int tile_n_fastest = 0;                              // set by the launcher: tile numbering inside an XCD's range
  const f16* zero_page = nullptr;                      // >= 16 bytes of zeros (for out-of-image taps)
  // optional (plain mode): GroupNorm(32) statistics of the finished output for up to two consuming GroupNorms -- the
  // output's channels are channels [gn_cbase, gn_cbase + N) of that GroupNorm's (possibly concatenated) input with
  // gn_cpg channels per group; gn_acc = its accumulator region (GroupNormParams::acc).  Needs Hout*Wout % 32 == 0.
  int gn_n = 0;
  long long* gn_acc[2] = {nullptr, nullptr};
  int gn_cpg[2] = {0, 0}, gn_cbase[2] = {0, 0};
  unsigned long long gn_magic[2] = {0, 0};             // ceil(2^40 / gn_cpg), filled by the launcher
  // optional (plain mode, round 4): GroupNorm(32) (+ SiLU) of the finished OUTPUT applied by the split-K reduction itself --
  // ResBlock conv1 -> out_layers' GroupNorm + SiLU -> conv2 (openaimodel.py:225-231).  When this GEMM ends up split and the
  // geometry fits (splitk_reduce_gn_kernel: one workgroup per (sample, group) sums the slabs, has the whole group in registers,
  // takes its statistics there and stores pgn_out = fp16(SiLU(GN(v))) -- the conv2 operand; no statistics atomics, no
  // GroupNorm-apply launch, and the fp32 value v is not stored unless pgn_keep_f32), *pgn_applied is set to 1; otherwise it is
  // left alone and the caller runs its GroupNorm-apply launch as before.
  const float* pgn_gamma = nullptr; const float* pgn_beta = nullptr; float pgn_eps = 1e-5f; int pgn_silu = 1;
  f16* pgn_out = nullptr;                              // [M][N] fp16
  int pgn_keep_f32 = 0;                                // also store out_f32 (someone besides that GroupNorm reads it)
  int* pgn_applied = nullptr;                          // host flag, written by the launcher
  // filled by the launcher: ceil(2^40 / (Hout*Wout)) and ceil(2^40 / Wout) for the kernel's division-free row split
  unsigned long long magic_hw = 0, magic_w = 0, magic_w2 = 0;   // (magic_w2: Wout + 2, halo-staged conv)
  int log2w = 0;
  // ---- GroupNorm(32) (+ SiLU) of the INPUT folded into the halo staging (conv3halo.hip, conv3halo_gn_kernel): the A operand is
  // then the fp32 residual stream itself (channel concat [xf0 | xf1], row pitches c0 / c1) and a0 / a1 / a2 are unused.  The
  // statistics accumulators must be complete when the kernel starts (GroupNormParams::acc of that GroupNorm).  Optional raw_hi /
  // raw_lo: the split-fp16 copy of the raw input ([M][c0 + c1], the operand of a ResBlock's 1x1 skip convolution), written once
  // per pixel by the tile_n == 0 workgroups.
  const float* xf0 = nullptr; const float* xf1 = nullptr;
  const long long* gn_in_acc = nullptr;
  const float* gn_in_gamma = nullptr; const float* gn_in_beta = nullptr; float gn_in_eps = 1e-5f; int gn_in_silu = 1;
  f16* raw_hi = nullptr; f16* raw_lo = nullptr;
  // optional fp16 [M][c0 + c1] scratch: with it the launcher may run this convolution as TWO launches instead (GroupNorm-apply
  // kernel into the scratch, then the LDS-DMA convolution) where the tuning table measured that faster (tile id SDMI_TILE_TWO_LAUNCH)
  f16* gn_scratch = nullptr;
  unsigned long long magic_cpg_in = 0;                 // launcher: ceil(2^40 / ((c0 + c1) / 32))
  // halo-staged conv geometry (launcher): output rows per image in a tile, images per tile, log2(pixels per image part)
  int halo_thi = 0, halo_ipt = 1, log2_tpi = 30;
  unsigned long long magic_hpi = 0;
};

constexpr int SDMI_TILE_TWO_LAUNCH = 99;   // (tuning table, GroupNorm-folding conv keys only) GroupNorm-apply launch + LDS-DMA conv
constexpr int SDMI_NUM_TILES = 23;   // tile ids 0 .. 22 (14..17: halo-staged 3x3 conv, 22: five-wave 64 x 160), see kTiles in igemm.hip and include/sdmi.h
struct IGemmTune {        // runtime knobs (tests sweep them; the executor takes the tuning table's choice)
  int tile = -1;          // -1 auto (tuning table, then heuristic); else a tile id
  int dma = -1;           // -1 default, 0 register-staged loads, 1 global_load_lds
};

int launch_igemm(const IGemmParams& p, const IGemmTune& tune, hipStream_t stream);
// split-fp16 1x1 conv with GroupNorm(32) of its fp32 input rows applied while the A operand is staged (gemm_split16.hip,
// gemm_split16_gn_kernel): x = xf0 [M][K], statistics gn_in_acc (complete), gn_in_gamma / beta / eps, weights packed [N][3K]
bool split16_gn_supported(const IGemmParams& p);
int launch_split16_gn(const IGemmParams& p, hipStream_t stream);
// may a stride-1 3x3 convolution over cat(c0, c1) fp32 channels at B x H x W fold the GroupNorm(32) of its input into its
// staging (IGemmParams::xf0 / gn_in_*; conv3halo.hip)?
bool gn_fold_conv_supported(int B, int H, int W, int c0, int c1, int N);
// fp16 range guard (range.hip, debug): scan an fp16 activation buffer a launch just wrote; see SDMI_CHECK_RANGE
bool range_check_enabled();
uint64_t tune_generation();
bool tune_collecting();                 // a tuning collection is running (igemm.hip): launches must go through the executor
int range_check_set(int enable);
int range_scan(const char* what, const f16* p, int64_t n, hipStream_t stream);
int range_report(std::string* json);

// in-situ tuning (igemm.hip): begin a collection run, select the candidate index for the following launches, end it
// (folds the timings into the table and writes it to `path`, or next to libsdmi.so when NULL)
int tune_begin();
int tune_round(int r);
int tune_end(const char* path, int* n_keys);
int tune_dump(std::string* out);
// out = sum_s slab[s] + bias + rowvec[batch] + residual (fixed order); uses M, N, Hout*Wout, splitk_ws, out_f32/out_f16
int launch_splitk_reduce(const IGemmParams& p, int nsplit, hipStream_t stream);

// Row-strip chain (rowchain.hip): GEGLU -> FF-out -> proj_out of a SpatialTransformer as ONE launch (attention.py:58-64,214,258-261).
// A workgroup owns 32 token rows for the whole chain; the weights stream through one LDS-DMA ring.  `epi` is the descriptor of the
// proj_out GEMM (M, N = K = C, B, Hout * Wout = rows per sample, bias, residual = the SpatialTransformer's input, out_f32 / ldo,
// optional fp16 copy and GroupNorm-statistics targets); the operand pointers of that descriptor are not read.
struct FfTailParams {
  const f16* ln = nullptr;        // [M][C] fp16(gamma3 * t): the LayerNorm-folded GEGLU operand its producer stored (IGemmParams::f16_scale)
  const float* lnp = nullptr;     // [C / 32][M][2] row partials of t (IGemmParams::lnp_out)
  float ln_eps = 1e-5f;
  const float* csd = nullptr;     // LayerNorm-fold column terms of the packed GEGLU columns, per hidden chunk of C: [4][cs 2C | d 2C]
  const f16* wgg = nullptr;       // [8C][C] packed GEGLU weights (value32 | gate32 interleave)
  const f16* wff2 = nullptr;      // [C][4C]
  const float* bff2 = nullptr;    // [C]
  const float* t = nullptr;       // [M][C] fp32 token stream (residual of FF-out)
  const f16* wpo = nullptr;       // [C][3C] split-fp16 proj_out weights [hi | hi | lo]
  // the out-projection of attn2 in front (a16 != nullptr; `ln` / `lnp` are then unused, t is updated in place): t += a16 Wo^T + bo,
  // the GEGLU operand fp16(ln_gamma * t) and its row statistics stay on the CU                 attention.py:213, 191-192
  const f16* a16 = nullptr;       // [M][C] fp16: the cross-attention output rows
  const f16* wo = nullptr;        // [C][C] attn2.to_out weights
  const float* bo = nullptr;      // [C]
  const float* ln_gamma = nullptr;  // [C] norm3 weight
#ifdef SDMI_RC_TIMING
  long long* dbg = nullptr;       // timing build only: [workgroups][128] s_memtime stamps
#endif
  IGemmParams epi;
};
bool ff_tail_supported(int C, int M, int rows_per_sample);
int launch_ff_tail(const FfTailParams& p, hipStream_t stream);

// Row-strip chain kernel, the HEAD of a SpatialTransformer (rowchain.hip): GroupNorm-apply -> proj_in -> (norm1 folded) q | k | v
//   t = proj_in(norm(x)) + b;  q | k | v = norm1(t) Wqkv^T        ldm/modules/attention.py:254-256 (norm, proj_in), :212 + :170-176 (attn1's
// projections) as ONE launch instead of GroupNorm-apply, the split-fp16 proj_in GEMM and the q|k|v GEMM.  Same arithmetic in the same order as
// those launches (the outputs are the same bits).
struct StHeadParams {
  const f16* a16 = nullptr;        // (launch_st_mid only) [M][C] fp16: the self-attention output rows, operand of the out-projection
  const float* x = nullptr;        // [M][C] fp32: the SpatialTransformer's input (NHWC rows)
  const long long* gn_acc = nullptr;   // GroupNorm statistics accumulators of x (GroupNormParams::acc, complete before the launch)
  const float* gn_gamma = nullptr; const float* gn_beta = nullptr; float gn_eps = 1e-6f;
  const f16* w_in = nullptr;       // [C][3C] split-fp16 proj_in weights [hi | hi | lo]
  const float* b_in = nullptr;     // [C]
  float* t = nullptr;              // [M][C] fp32 out: the token stream
  const float* ln_gamma = nullptr; // norm1 weight: the q|k|v operand is fp16(gamma * t) (IGemmParams::f16_scale), beta lives in lnf_d
  float ln_eps = 1e-5f;
  const f16* wqkv = nullptr;       // [3C][C] attn1 to_q | to_k | to_v
  const float* lnf_cs = nullptr; const float* lnf_d = nullptr;   // [3C] LayerNorm-fold column terms (IGemmParams::lnf_cs / lnf_d)
  f16* q = nullptr; f16* k = nullptr;  // [B * heads][ntok][dh]
  f16* vt = nullptr;               // [B * heads][dh][ntok_pad]
  int M = 0, B = 0, ntok = 0, ntok_pad = 0, heads = 0, dh = 0, C = 0;
  // (launch_st_mid, optional) the cross-attention behind to_q inside the launch: cached context K [B * heads][nkv][dh] and V^T
  // [B * heads][dh][nkv_pad] (TBlock::ck / cvt), softmax scale, the attention output rows ao_out [M][C] fp16; `q` is then not written
  const f16* ctx_k = nullptr; const f16* ctx_vt = nullptr; int ctx_nkv = 0, ctx_nkv_pad = 0; float ctx_scale = 0.f; f16* ao_out = nullptr;
#ifdef SDMI_RC_TIMING
  long long* dbg = nullptr;
#endif
};
bool st_head_supported(int C, int M, int ntok, int ntok_pad, int heads, int dh);
int launch_st_head(const StHeadParams& p, hipStream_t stream);
// ... and the MIDDLE of a BasicTransformerBlock on the same kernel: t += a16 Wo^T + b_in in place (w_in = Wo [C][C] plain fp16), q = norm2(t) Wq^T
// (ln_gamma = norm2 weight, wqkv = Wq [C][C], lnf_cs / lnf_d [C]; k / vt / x / gn_* unused)     attention.py:212-213, 170, 191-192
int launch_st_mid(const StHeadParams& p, hipStream_t stream);

// ResBlock in_layers / out_layers as one launch (gnconv.hip): out = conv3x3(SiLU(GroupNorm32(cat(x0, x1)))) + bias (+ row vector, + residual)
//   openaimodel.py:201-204, 225-231; util.py:199-216.  A workgroup owns 32 pixels x ALL 320 output channels: the halo is normalised once.
struct GnConvParams {
  const float* x0 = nullptr; const float* x1 = nullptr; int c0 = 0, c1 = 0;      // fp32 NHWC sources (channel concat; c0 % 64 == 0)
  const long long* gn_acc = nullptr;        // GroupNorm statistics accumulators of cat(x0, x1) (GroupNormParams::acc, complete before the launch)
  const float* gn_gamma = nullptr; const float* gn_beta = nullptr; float gn_eps = 1e-5f;
  const f16* w = nullptr;                   // packed conv weights [320][9 Cin] (launch_pack_conv_weight)
#ifdef SDMI_RC_TIMING
  long long* dbg = nullptr;                 // timing build only: [workgroups][16] cycle stamps
#endif
  IGemmParams epi;                          // the convolution's descriptor (conv3(): B, H, W, N = 320, K = 9 Cin) with its epilogue fields
};
bool gn_conv3_supported(int B, int H, int W, int c0, int c1, int Cout);
int launch_gn_conv3(const GnConvParams& p, hipStream_t stream);

// Flash attention over per-head layouts produced by EPI_HEADS
struct AttnParams {
  const f16* q = nullptr;    // [BH][nq][d]
  const f16* k = nullptr;    // [BH][nkv][d]
  const f16* vt = nullptr;   // [BH][d][nkv_pad]   (nkv_pad = round_up(nkv, 8), pad tokens zero)
  f16* out = nullptr;        // [B][nq][heads*d]  (heads merged, '(b h) n d -> b n (h d)')
  int BH = 0, heads = 0, nq = 0, nkv = 0, nkv_pad = 0, d = 0;
  float scale = 1.f;
  int nw = 0;                // waves per workgroup (0 = auto): each wave owns 32 queries
  int causal = 0;            // 1: key j attends only to queries i >= j (CLIP text model); needs nq == nkv
  int prio = 0;              // launcher (SDMI_ATTN_PRIO=1): s_setprio 1 for the second-dispatched half of an 8-wave workgroup
  int pingpong = 0;          // launcher (SDMI_ATTN_PP): 8-wave launches on attn_pp_kernel (the halves of a workgroup alternate matrix / VALU blocks)
};
int launch_attention(const AttnParams& p, hipStream_t stream);

// Cross-attention with the query projection inside the kernel (attn_ctx.hip): out = softmax((x Wq^T) K^T scale) V over <= 128 cached keys
struct AttnCtxParams {
  const f16* x = nullptr;    // [B * nq][C] token rows: LayerNorm output, or fp16(gamma * t) with the LayerNorm fold below
  const f16* wq = nullptr;   // [C][C] to_q weight, rows = output channels (head * d + dd)
  const f16* k = nullptr;    // [BH][nkv][d]
  const f16* vt = nullptr;   // [BH][d][nkv_pad]
  f16* out = nullptr;        // [B][nq][heads * d]
  int BH = 0, heads = 0, nq = 0, nkv = 0, nkv_pad = 0, d = 0, C = 0;
  float scale = 1.f;
  // optional LayerNorm fold (as IGemmParams::lnf_*): partials [lnf_npart = C / 32][M][2], M = B * nq rows
  const float* lnf_part = nullptr; int lnf_npart = 0; float lnf_eps = 1e-5f; int M = 0;
  const float* lnf_cs = nullptr; const float* lnf_d = nullptr;
};
bool attention_ctx_supported(int d, int C, int nkv);
int launch_attention_ctx(const AttnCtxParams& p, hipStream_t stream);

// GroupNorm(32) (+SiLU) over fp32 NHWC, channel-concat of two sources, fp16 or fp32 output
struct GroupNormParams {
  const float* x0 = nullptr; const float* x1 = nullptr; int c0 = 0, c1 = 0;
  int B = 0, HW = 0;
  const float* gamma = nullptr; const float* beta = nullptr; float eps = 1e-5f;
  int silu = 0;
  int stats_only = 0;          // 1: only fill the statistics accumulators (consumed by a GroupNorm-folding convolution, IGemmParams::gn_in_acc)
  int skip_stats = 0;          // 1: the accumulators were already filled by the producing GEMM epilogues (IGemmParams::gn_*)
  f16* out_f16 = nullptr;      // [B*HW][C] normalised (+SiLU)
  float* out_f32 = nullptr;    // same in fp32 (used by the output head)
  f16* raw_f16 = nullptr;      // optional: un-normalised fp16 copy of cat(x0,x1) (A operand of the 1x1 skip conv)
  f16* out_lo = nullptr;       // optional: fp16(y - float(fp16(y)))   -- low half of a split-fp16 operand
  f16* raw_lo = nullptr;       // optional: same for the raw copy
  // fixed-point statistics accumulators of THIS GroupNorm call: [B][32 groups][GN_SLOTS][GN_STRIDE] int64 (GN_WORDS used), zero before
  // the launch.  Each of {sum, sumsq} is kept as an integer part and a 2^-40 fraction (two words), so the range is that
  // of an int64 and nothing can wrap.  Integer atomics are associative: the statistics are bit-reproducible without a
  // finalize pass or inter-block ordering; consumers fold the slots themselves (gn_fold in norm.hip).
  long long* acc = nullptr;
};
constexpr int GN_SLOTS = 8;
constexpr int GN_MAX_CALLS = 96;   // accumulator regions per UNet forward (SD v1 has 61 GroupNorms)
constexpr int GN_WORDS = 4;         // {sum int, sum frac * 2^40, sumsq int, sumsq frac * 2^40}
// words between two slot entries: every (sample, group, slot) entry owns a 128-byte line, so the atomic adds of one
// GroupNorm spread over 512+ lines / all memory channels instead of queueing on 128 (measured: the adds, not the
// arithmetic, were what the statistics cost)
constexpr int GN_STRIDE = 16;
static inline int64_t gn_acc_words(int B) { return (int64_t)B * 32 * GN_SLOTS * GN_STRIDE; }   // int64 words per GroupNorm call
int launch_groupnorm(const GroupNormParams& p, hipStream_t stream);

int launch_layernorm(const float* x, const float* gamma, const float* beta, f16* out, int M, int C, float eps,
                     hipStream_t stream, float* out_f32 = nullptr);     // out / out_f32: either or both
int launch_cast_f16(const float* x, f16* out, f16* out_lo, int64_t n, hipStream_t stream);

// fp32 "small" path
int launch_timestep_embedding(const int64_t* t_i64, const float* t_f32, float* out, int B, int dim, hipStream_t s);
int launch_small_linear(const float* in, int ld_in, const float* w, const float* bias, float* out, int ld_out,
                        int B, int N, int K, int silu_in, hipStream_t s);
// gn_*: GroupNorm statistics of the output for up to two consuming GroupNorms (accumulator regions, channels per group, channel offset of
// this tensor inside the GroupNorm's input: IGemmParams::gn_acc / gn_cpg / gn_cbase); needs H * W % 16 == 0
int launch_conv_in(const float* x_nchw, const float* w, const float* bias, float* out_nhwc, int B, int Cin, int H,
                   int W, int Cout, hipStream_t s, int gn_n = 0, long long* const* gn_acc = nullptr, const int* gn_cpg = nullptr,
                   const int* gn_cbase = nullptr);
int launch_conv_out(const float* h_nhwc, const float* w_khwc, const float* bias, float* out_nchw, int B, int H, int W,
                    int Cin, int Cout, hipStream_t s);

// DPM-Solver++ multistep step (see sampler.hip)
struct DpmStepParams {
  const float* eps_model = nullptr;  // model output: rows [0,n) uncond, [n,2n) cond when cfg, else [0,n)
  int cfg = 0; float scale = 1.f;
  const float* x = nullptr;          // latent the model was evaluated at
  const float* m_prev = nullptr;     // previous data prediction (order 2)
  float alpha_s = 1.f, sigma_s = 0.f;      // marginal alpha / std at the evaluation time
  float cx = 1.f, a = 0.f, inv_r0 = 0.f;   // update coefficients
  int order = 1;
  float* m_out = nullptr;            // data prediction at this step (kept as history)
  float* x_next = nullptr;           // optional: the updated latent
  int64_t n = 0;
};
int launch_dpm_step(const DpmStepParams& p, hipStream_t s);

// text-encoder helpers: out[m][:] = tok_emb[ids[m]][:] + pos_emb[m % L][:] (fp32); out16 = fp16(x * sigmoid(1.702 x))
int launch_embed_tokens(const int64_t* ids, const float* tok_emb, const float* pos_emb, float* out, int M, int L, int C,
                        int vocab, hipStream_t s);
int launch_quick_gelu(const float* x, f16* out, int64_t n, hipStream_t s);

// first-stage (VAE) helpers
int launch_pointwise_nchw(const float* x_nchw, const float* w, const float* bias, float* out_nchw, int B, int Cin, int Cout,
                          int HW, float in_scale, hipStream_t s);
int launch_softmax_rows(const float* S, f16* P, int rows, int cols, int lds, int ldp, float scale, hipStream_t s);

// cache hint (small.hip): touch every 128-byte line of a device range
int launch_prefetch_lines(const void* ptr, int64_t bytes, hipStream_t s);

// weight packing (device pointers, fp32 reference layouts -> packed)
int launch_pack_conv_weight(const float* w_oihw, f16* dst, int O, int I, int KH, int KW, hipStream_t s);  // -> [O][KH][KW][I]
int launch_pack_rows(const float* w, f16* dst, int rows, int cols, int dst_row0, int dst_ld, hipStream_t s);
// column terms of a GEMM that folds LayerNorm(gamma, beta) of its input rows (IGemmParams::lnf_cs / lnf_d), from the PACKED fp16
// weights: cs[n] = sum_k gamma[k] w[n][k], d[n] = sum_k beta[k] w[n][k] (+ bias[n])
int launch_ln_fold_prep(const f16* w, int N, int K, int ldw, const float* gamma, const float* beta, const float* bias, float* cs,
                        float* d, hipStream_t s);
// split-fp16 weights for the 3-pass 1x1 convs: dst [N][3K] = [hi | hi | lo], lo = fp16(w - float(hi))
int launch_pack_split3(const float* w, f16* dst, int N, int K, hipStream_t s);
int launch_pack_conv_split3(const float* w, f16* dst, int O, int I, int KH, int KW, hipStream_t s);
int launch_pack_geglu(const float* w, const float* bias, f16* wdst, float* bdst, int N, int K, hipStream_t s);
int launch_pack_conv_out(const float* w_oihw, float* dst, int O, int I, hipStream_t s);   // -> [O][3][3][I] fp32

// sampler update (fp32): see sampler.hip
struct SamplerStepParams {
  const float* eps_model = nullptr;  // model output: rows [0,n) uncond, [n,2n) cond when cfg, else [0,n)
  int cfg = 0; float scale = 1.f;    // e_t = e_u + scale * (e_c - e_u)                      plms.py:182-186
  const float* x = nullptr;          // [B,4,H,W] current latent
  // multistep combination (plms.py:218-232), evaluated in the reference's fp32 operation order:
  //   0: e' = e_t                      (DDIM, ddim.py:189-204)
  //   1: e' = (3 e_t - o0) / 2         2: e' = (23 e_t - 16 o0 + 5 o1) / 12
  //   3: e' = (55 e_t - 59 o0 + 37 o1 - 9 o2) / 24
  //   4: e' = (o0 + e_t) / 2           (first PLMS step: o0 = e_t, e_t = e_t_next)
  int mode = 0;
  const float* old0 = nullptr; const float* old1 = nullptr; const float* old2 = nullptr;  // newest first
  float a_t = 1.f, a_prev = 1.f, sigma = 0.f, sqrt_1m_at = 0.f;                           // table entries
  const float* noise = nullptr;      // optional [B,4,H,W] (sigma > 0), already scaled by temperature
  float* e_t_out = nullptr;          // optional: combined (post-CFG) eps, kept for the history
  float* x_prev = nullptr; float* pred_x0 = nullptr;   // pred_x0 optional
  int64_t n = 0;                     // elements per output (B*4*H*W)
};
int launch_sampler_step(const SamplerStepParams& p, hipStream_t s);
// decode output fp32 NCHW -> uint8 NHWC exactly as scripts/txt2img.py:313-324 does on the host (sampler.hip)
int launch_image_u8(const float* img_nchw, unsigned char* out_nhwc, int B, int C, int H, int W, hipStream_t s);

}  // namespace sdmi
