// GroupNorm(32)(+SiLU), LayerNorm and fp32->fp16 cast over NHWC activations (HBM-bound kernels).
//
// Reference sites: GroupNorm32 (ldm/modules/diffusionmodules/util.py:199-216, eps 1e-5, used in
// openaimodel.py:201-203,225-227,683-684), SpatialTransformer.norm (ldm/modules/attention.py:76-77, eps 1e-6),
// LayerNorm x3 per BasicTransformerBlock (attention.py:203-205).  Statistics are fp32 (combined in fp64);
// the normalised value is rounded to fp16 exactly once -- it is the MFMA A operand of the following conv / GEMM.
// The UNet skip concat (openaimodel.py:736) is folded in: the kernels read two channel-concatenated sources.
#include "common.h"
#include "prof.h"

namespace sdmi {
namespace {

// pixels per statistics block: small at the low-resolution levels, where the parallelism has to come from
// many tiny blocks (the kernels are latency-, not bandwidth-bound there)
// pixels per statistics block: small at UNet sizes (many blocks for a short kernel); at VAE sizes (HW >= 16K) ~16K elements
// per block, which also bounds the number of atomics that land on one accumulator word
static inline int gn_chunk(int HW, int C) {
  if (HW >= 16384) { int px = 16; while (px * C < 16384 && px < 256) px *= 2; return px; }
  return HW >= 2048 ? 16 : (HW >= 512 ? 8 : 4);
}
constexpr int GN_MAXC = 2560 * 2;

__device__ __forceinline__ f32x4 load_cat4(const float* x0, const float* x1, int c0, int c1, size_t pix, int c) {
  return (c < c0) ? *(const f32x4*)(x0 + pix * c0 + c) : *(const f32x4*)(x1 + pix * c1 + (c - c0));
}
__device__ __forceinline__ f16x4 lo_half(const f32x4 v) {   // fp16(v - float(fp16(v)))
  f16x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = (f16)(v[j] - (float)(f16)v[j]);
  return r;
}

// Statistics: every block reduces its pixel chunk to one {sum, sumsq} per group and adds them, as fixed-point
// int64 words (integer part + 2^-40 fraction), into one of GN_SLOTS accumulator slots of (b, g).  Integer addition is associative:
// the totals are bit-identical whatever the block order, with no finalize kernel and no inter-block ordering.
// Threads are laid out as [pixel lane][channel quad] so (almost) all 256 threads have loads in flight even at C = 320.
__global__ void __launch_bounds__(256) gn_stats_kernel(GroupNormParams p, int nchunk, int chunk_px) {
  __shared__ float csum[GN_MAXC], csq[GN_MAXC];
  __shared__ float lsum[1024], lsq[1024];             // extra pixel lanes: (PL - 1) * C < 1024 floats
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int C = p.c0 + p.c1;
  const int nq = C / 4;
  const int pix0 = chunk * chunk_px;
  const int npix = min(chunk_px, p.HW - pix0);
  // pixel lanes: 8 on first-stage-sized maps so all 256 threads load even at C = 128 (UNet-sized maps keep <= 4)
  const int PL = (nq <= 256) ? min(p.HW >= 16384 ? 8 : 4, 256 / nq) : 1;
  const int pl = (nq <= 256) ? tid / nq : 0;
  const int q0 = (nq <= 256) ? tid - pl * nq : tid;
  if (pl < PL) {
    for (int q = q0; q < nq; q += 256) {
      const int c = q * 4;
      f32x4 s = {0, 0, 0, 0}, ss = {0, 0, 0, 0};
      for (int i0 = pl; i0 < npix; i0 += 8 * PL) {      // 8 independent loads in flight per thread
        f32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int pi = i0 + i * PL;
          v[i] = (pi < npix) ? load_cat4(p.x0, p.x1, p.c0, p.c1, (size_t)b * p.HW + pix0 + pi, c) : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { s += v[i]; ss += v[i] * v[i]; }
      }
      if (pl == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { csum[c + j] = s[j]; csq[c + j] = ss[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { lsum[(pl - 1) * C + c + j] = s[j]; lsq[(pl - 1) * C + c + j] = ss[j]; }
      }
    }
  }
  __syncthreads();
  if (tid < 32) {
    const int cpg = C / 32;
    float s = 0.f, ss = 0.f;
    for (int j = 0; j < cpg; ++j) {
      const int c = tid * cpg + j;
      float a = csum[c], q = csq[c];
      for (int l = 1; l < PL; ++l) { a += lsum[(l - 1) * C + c]; q += lsq[(l - 1) * C + c]; }
      s += a; ss += q;
    }
    unsigned long long* dst = (unsigned long long*)p.acc + ((size_t)(b * 32 + tid) * GN_SLOTS + (chunk & (GN_SLOTS - 1))) * GN_STRIDE;
    gn_acc_add(dst, s);
    gn_acc_add(dst + 2, ss);
  }
}

// fold the GN_SLOTS accumulators of (b, g): called by 8 consecutive lanes (sub = lane & 7), result valid on sub == 0
__device__ __forceinline__ void gn_fold(const long long* acc, int b, int g, int sub, double n, float eps, float* mean,
                                        float* rstd) {
  const long long* src = acc + ((size_t)(b * 32 + g) * GN_SLOTS + sub) * GN_STRIDE;
  long long s = src[0], sl = src[1], ss = src[2], ssl = src[3];
#pragma unroll
  for (int o = GN_SLOTS / 2; o >= 1; o >>= 1) {
    s += __shfl_xor(s, o); sl += __shfl_xor(sl, o); ss += __shfl_xor(ss, o); ssl += __shfl_xor(ssl, o);
  }
  gn_mean_rstd(s, sl, ss, ssl, n, eps, mean, rstd);
}

// floor(m / d) for 0 <= m, m * d < 2^40, magic = ceil(2^40 / d) (host computed; same scheme as igemm.hip's fast_div)
__device__ __forceinline__ int gn_fast_div(int m, unsigned long long magic) {
  return (int)(((unsigned long long)(unsigned)m * magic) >> 40);
}
static unsigned long long gn_div_magic(int d) { return ((1ull << 40) + (unsigned long long)d - 1) / (unsigned long long)d; }

// normalise (+SiLU); U channel quads per thread (U = 4 on first-stage-sized maps so the per-block statistics fold is
// amortised); grid (blocks per batch row, B)
// xcd_affine: the dispatcher places block L (x fastest) on XCD L % 8; renumbered so that XCD x handles the x-th eighth of the
// (sample, pixel) range -- the rows a following row-affine GEMM (igemm: "an XCD owns rows of A" when M > N) reads on that same XCD,
// whose L2 then already holds the fp16 operand this kernel wrote (plain stores stay in the L2).  Speed only; any placement is correct.
template <int U>
__global__ void __launch_bounds__(256) gn_apply_kernel(GroupNormParams p, const unsigned long long magic_nq,
                                                       const unsigned long long magic_cpg, const int xcd_affine) {
  __shared__ float s_mean[32], s_rstd[32];
  const int C = p.c0 + p.c1;
  const int cpg = C / 32;
  const int nq = C / 4;
  int bx = blockIdx.x, by = blockIdx.y;
  if (xcd_affine) {
    const int nb = gridDim.x, lin = by * nb + bx, per = (nb * gridDim.y) >> 3;
    const int wg = (lin & 7) * per + (lin >> 3);
    by = wg / nb; bx = wg - by * nb;
  }
  const int b = by, tid = threadIdx.x;
  // The activation quads (and their gamma / beta) are requested FIRST: they do not depend on the statistics, so their
  // latency runs under the accumulator loads and the fp64 fold below instead of behind them (a block handles 256 * U quads:
  // with U = 1 the fold's round trip was as long as the block's useful work).
  const int64_t total = (int64_t)p.HW * nq;
  const int64_t base = (int64_t)bx * (256 * U) + tid;           // quad index inside this batch row
  f32x4 v[U], ga[U], be[U]; size_t pix[U]; int ch[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t idx = base + u * 256;
    if (idx < total) {
      // (magic != 0: the launcher found the quad index small enough for the multiply-shift division -- the int64 division
      // sequence was ~200 instructions in front of the first load)
      const int64_t px = magic_nq ? (int64_t)gn_fast_div((int)idx, magic_nq) : idx / nq;
      pix[u] = (size_t)b * p.HW + (size_t)px;
      ch[u] = (int)(idx - px * nq) * 4;
      v[u] = load_cat4(p.x0, p.x1, p.c0, p.c1, pix[u], ch[u]);
      ga[u] = *(const f32x4*)(p.gamma + ch[u]);
      be[u] = *(const f32x4*)(p.beta + ch[u]);
    }
  }
  {
    float m, r;
    gn_fold(p.acc, b, tid >> 3, tid & 7, (double)cpg * (double)p.HW, p.eps, &m, &r);
    if ((tid & 7) == 0) { s_mean[tid >> 3] = m; s_rstd[tid >> 3] = r; }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (base + u * 256 >= total) continue;
    const int c = ch[u];
    const int g0 = gn_fast_div(c, magic_cpg), g1 = gn_fast_div(c + 3, magic_cpg);   // a quad touches at most two groups (cpg >= 2)
    const float m0 = s_mean[g0], r0 = s_rstd[g0], m1 = s_mean[g1], r1 = s_rstd[g1];
    const int split = (g0 + 1) * cpg - c;                // first channel offset that belongs to g1
    f32x4 y;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool second = j >= split;
      y[j] = gn_apply_elem(v[u][j], second ? m1 : m0, second ? r1 : r0, ga[u][j], be[u][j], p.silu);
    }
    const size_t o = pix[u] * C + c;
    if (p.out_f16) SDMI_ST_F16X4(p.out_f16, o, (f16x4{(f16)y[0], (f16)y[1], (f16)y[2], (f16)y[3]}));
    if (p.out_lo) SDMI_ST_F16X4(p.out_lo, o, lo_half(y));
    if (p.out_f32) SDMI_ST_F32X4(p.out_f32, o, y);
    if (p.raw_f16) *(f16x4*)(p.raw_f16 + o) = f16x4{(f16)v[u][0], (f16)v[u][1], (f16)v[u][2], (f16)v[u][3]};
    if (p.raw_lo) *(f16x4*)(p.raw_lo + o) = lo_half(v[u]);
  }
}

// one wave per row
template <int MAXQ>
__global__ void __launch_bounds__(256) layernorm_kernel(const float* x, const float* gamma, const float* beta, f16* out,
                                                        int M, int C, float eps, float* out32) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * C;
  f32x4 v[MAXQ];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) { v[i] = *(const f32x4*)(xr + c); s += v[i][0] + v[i][1] + v[i][2] + v[i][3]; }
    else v[i] = f32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; ss += d * d; }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
  const float rstd = 1.0f / sqrtf(ss / (float)C + eps);
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
      const f32x4 ga = *(const f32x4*)(gamma + c);
      const f32x4 be = *(const f32x4*)(beta + c);
      f32x4 y32;
#pragma unroll
      for (int j = 0; j < 4; ++j) y32[j] = (v[i][j] - mean) * rstd * ga[j] + be[j];
      if (out) SDMI_ST(f16x4, out + (size_t)row * C + c, (f16x4{(f16)y32[0], (f16)y32[1], (f16)y32[2], (f16)y32[3]}));
      if (out32) *(f32x4*)(out32 + (size_t)row * C + c) = y32;
    }
  }
}

__global__ void cast_f16_kernel(const float* x, f16* out, f16* out_lo, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = *(const f32x4*)(x + i * 4);
  *(f16x4*)(out + i * 4) = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
  if (out_lo) *(f16x4*)(out_lo + i * 4) = lo_half(v);
}

}  // namespace

int launch_groupnorm(const GroupNormParams& p, hipStream_t stream) {
  const int C = p.c0 + p.c1;
  SDMI_CHECK(C % 32 == 0 && C <= GN_MAXC && p.c0 % 4 == 0 && p.c1 % 4 == 0, "GroupNorm(32) channel constraint");
  SDMI_CHECK(p.acc != nullptr && p.gamma && p.beta && p.x0, "GroupNorm: missing pointer");
  SDMI_CHECK(p.c1 == 0 || p.x1 != nullptr, "GroupNorm: second source missing");
  SDMI_CHECK(C / 32 >= 2, "GroupNorm: at least 2 channels per group");
  const int chunk_px = gn_chunk(p.HW, C);
  const int nchunk = cdiv(p.HW, chunk_px);
  const double nel = (double)p.B * p.HW * C;
  ProfScope ps("groupnorm", 0.0, nel * 4.0 + nel * ((p.out_f16 ? 2.0 : 0.0) + (p.out_f32 ? 4.0 : 0.0) + (p.raw_f16 ? 2.0 : 0.0)), stream);
  if (!p.skip_stats) SDMI_LAUNCH(gn_stats_kernel, dim3(nchunk, p.B), dim3(256), 0, stream, p, nchunk, chunk_px);
  if (!p.stats_only && (p.out_f16 || p.out_f32 || p.raw_f16 || p.out_lo || p.raw_lo)) {
    const int64_t quads = (int64_t)p.HW * (C / 4);
    static const int64_t u4_from = getenv("SDMI_GN_APPLY_U4_QUADS") ? atoll(getenv("SDMI_GN_APPLY_U4_QUADS")) : 300000;   // A/B knob; default: the 64x64 level of the 512 x 512 workload (327 680 quads per sample at 320 channels) and everything larger -- round 4, same-box A/B -0.03 ms per UNet call, 32x32 maps and below measured no gain (profiles/wt_stores_r04.txt)
    const int nq = C / 4;
    // multiply-shift division needs (quad index + a block's overshoot) * divisor < 2^40
    const unsigned long long magic_nq = ((quads + 1024) * (int64_t)nq < ((int64_t)1 << 40) && quads + 1024 < ((int64_t)1 << 31)) ? gn_div_magic(nq) : 0ull;
    const unsigned long long magic_cpg = gn_div_magic(C / 32);           // (c + 3) * cpg < 2^40 always
    // XCD-affine block numbering (see the kernel): SDMI_GN_XCD=1, round-4 experiment; needs a block count that is a multiple of 8
    const int want_xcd = getenv("SDMI_GN_XCD") ? atoi(getenv("SDMI_GN_XCD")) : 0;
    if (quads >= u4_from) {
      const unsigned nb = (unsigned)((quads + 1023) / 1024);
      SDMI_LAUNCH(gn_apply_kernel<4>, dim3(nb, p.B), dim3(256), 0, stream, p, magic_nq, magic_cpg, (want_xcd && (nb * p.B) % 8 == 0) ? 1 : 0);
    } else {
      const unsigned nb = (unsigned)((quads + 255) / 256);
      SDMI_LAUNCH(gn_apply_kernel<1>, dim3(nb, p.B), dim3(256), 0, stream, p, magic_nq, magic_cpg, (want_xcd && (nb * p.B) % 8 == 0) ? 1 : 0);
    }
  }
  SDMI_HIP_OK(hipGetLastError());
  if (range_check_enabled()) {
    const int64_t nel = (int64_t)p.B * p.HW * C;
    if (range_scan("GroupNorm fp16 output", p.out_f16, nel, stream)) return -1;
    if (range_scan("GroupNorm raw fp16 copy of the residual stream (1x1 skip conv operand)", p.raw_f16, nel, stream)) return -1;
  }
  return 0;
}

int launch_layernorm(const float* x, const float* gamma, const float* beta, f16* out, int M, int C, float eps,
                     hipStream_t stream, float* out_f32) {
  SDMI_CHECK(C % 4 == 0 && C <= 2560, "LayerNorm: C must be a multiple of 4 and <= 2560");
  dim3 grid(cdiv(M, 4)), block(256);
  ProfScope ps("layernorm", 0.0, (double)M * C * 6.0, stream);
  // MAXQ = quads per lane, instantiated per width class: the 5-slot kernel needs 108 VGPRs (4 waves per SIMD) whatever C is,
  // the 2- and 3-slot ones of C <= 512 / 768 (the 64x64 and 32x32 levels: 320 / 640 channels) keep twice the rows in flight.
  // Same per-lane order of the same terms: bit-identical.  SDMI_LN_SLOTS=0 = always the 5-slot kernel (A/B).
  const char* e_slots = getenv("SDMI_LN_SLOTS");
  const bool narrow = !(e_slots && atoi(e_slots) == 0);
  if (narrow && C <= 512) SDMI_LAUNCH(layernorm_kernel<2>, grid, block, 0, stream, x, gamma, beta, out, M, C, eps, out_f32);
  else if (narrow && C <= 768) SDMI_LAUNCH(layernorm_kernel<3>, grid, block, 0, stream, x, gamma, beta, out, M, C, eps, out_f32);
  else if (C <= 1280) SDMI_LAUNCH(layernorm_kernel<5>, grid, block, 0, stream, x, gamma, beta, out, M, C, eps, out_f32);
  else SDMI_LAUNCH(layernorm_kernel<10>, grid, block, 0, stream, x, gamma, beta, out, M, C, eps, out_f32);
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}

int launch_cast_f16(const float* x, f16* out, f16* out_lo, int64_t n, hipStream_t stream) {
  SDMI_CHECK(n % 4 == 0, "cast: n % 4 != 0");
  const int64_t n4 = n / 4;
  ProfScope ps("cast_f16", 0.0, (double)n * 6.0, stream);
  SDMI_LAUNCH(cast_f16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, x, out, out_lo, n4);
  SDMI_HIP_OK(hipGetLastError());
  if (range_check_enabled() && range_scan("fp16 cast of the residual stream / context", out, n, stream)) return -1;
  return 0;
}

}  // namespace sdmi
