// The fp32 "small" path of the UNet: everything whose cost is negligible but whose rounding is not.
//   * sinusoidal timestep embedding + time_embed MLP + the 22 ResBlock emb_layers
//     (ldm/modules/diffusionmodules/util.py:151-171, openaimodel.py:506-511,218-224) -- M = batch rows, GEMV-like,
//     weight-bandwidth bound, kept in fp32.
//   * the 4->C input conv and the C->4 output conv (openaimodel.py:519,685): K = 36 resp. N = 4, useless for MFMA;
//     they also fold the NCHW<->NHWC layout change at the library boundary.
//   * weight packing kernels (reference layouts -> fp16 [N][K] with K = (ky,kx,cin)).
#include "common.h"
#include "prof.h"
#include <math.h>
#include <vector>

namespace sdmi {
namespace {

__global__ void temb_kernel(const int64_t* t_i64, const float* t_f32, const float* freqs, float* out, int B, int dim) {
  const int half = dim / 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * half) return;
  const int b = idx / half, i = idx - b * half;
  const float t = t_i64 ? (float)t_i64[b] : t_f32[b];
  const float arg = t * freqs[i];
  out[(size_t)b * dim + i] = cosf(arg);
  out[(size_t)b * dim + half + i] = sinf(arg);
  if ((dim & 1) && i == 0) out[(size_t)b * dim + dim - 1] = 0.f;
}

constexpr int SL_MAXB = 8;
// one wave per output feature n: out[b][n] = bias[n] + sum_k W[n][k] * f(in[b][k]),  f = SiLU or identity
__global__ void __launch_bounds__(256) small_linear_kernel(const float* in, int ld_in, const float* w, const float* bias,
                                                           float* out, int ld_out, int B, int N, int K, int silu_in) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc[SL_MAXB];
#pragma unroll
  for (int b = 0; b < SL_MAXB; ++b) acc[b] = 0.f;
  const float* wr = w + (size_t)n * K;
  for (int k = lane * 4; k < K; k += 256) {
    const f32x4 wv = *(const f32x4*)(wr + k);
#pragma unroll
    for (int b = 0; b < SL_MAXB; ++b) {
      if (b < B) {
        f32x4 xv = *(const f32x4*)(in + (size_t)b * ld_in + k);
        if (silu_in) {
#pragma unroll
          for (int j = 0; j < 4; ++j) xv[j] = xv[j] / (1.0f + expf(-xv[j]));
        }
        acc[b] += wv[0] * xv[0] + wv[1] * xv[1] + wv[2] * xv[2] + wv[3] * xv[3];
      }
    }
  }
#pragma unroll
  for (int b = 0; b < SL_MAXB; ++b) {
    if (b < B) {
      float v = acc[b];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) out[(size_t)b * ld_out + n] = v + (bias ? bias[n] : 0.f);
    }
  }
}

// The same product with the (activated) input rows staged ONCE per block in LDS and all of a lane's weight quads in flight
// before the first FMA.  In small_linear_kernel every wave re-evaluates SiLU of all B x K inputs (an expf and an IEEE
// division per element) and waits for one 16-byte load at a time: on the 20160 x 1280 emb_layers matrix that is VALU- and
// latency-bound (~45 us for 103 MB).  Same expressions in the same per-lane order: the results are bit-identical
// (tests/test_kernels_gpu.py); the three launches of a UNet call 78.7 -> 52.7 us (profiles/small_kernels_r02.txt).
// KI = ceil(K / 256) weight quads per lane; dynamic LDS = B * K floats.
template <int KI>
__global__ void __launch_bounds__(512) small_linear_lds_kernel(const float* in, int ld_in, const float* w, const float* bias,
                                                               float* out, int ld_out, int B, int N, int K, int silu_in) {
  extern __shared__ __attribute__((aligned(16))) float sl_xs[];       // [B][K]
  const int tid = threadIdx.x, lane = tid & 63;
  for (int idx = tid * 4; idx < B * K; idx += 512 * 4) {              // K % 4 == 0: a quad stays inside one row
    const int b = idx / K, k = idx - b * K;
    f32x4 xv = *(const f32x4*)(in + (size_t)b * ld_in + k);
    if (silu_in) {
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[j] = xv[j] / (1.0f + expf(-xv[j]));
    }
    *(f32x4*)(sl_xs + idx) = xv;
  }
  __syncthreads();
  const int n = blockIdx.x * 8 + (tid >> 6);
  if (n >= N) return;
  const float* wr = w + (size_t)n * K;
  f32x4 wv[KI];
#pragma unroll
  for (int i = 0; i < KI; ++i) {
    const int k = lane * 4 + i * 256;
    wv[i] = k < K ? __builtin_nontemporal_load((const f32x4*)(wr + k)) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float acc[SL_MAXB];
#pragma unroll
  for (int b = 0; b < SL_MAXB; ++b) acc[b] = 0.f;
#pragma unroll
  for (int i = 0; i < KI; ++i) {
    const int k = lane * 4 + i * 256;
    if (k < K) {
#pragma unroll
      for (int b = 0; b < SL_MAXB; ++b) {
        if (b < B) {
          const f32x4 xv = *(const f32x4*)(sl_xs + b * K + k);
          acc[b] += wv[i][0] * xv[0] + wv[i][1] * xv[1] + wv[i][2] * xv[2] + wv[i][3] * xv[3];
        }
      }
    }
  }
#pragma unroll
  for (int b = 0; b < SL_MAXB; ++b) {
    if (b < B) {
      float v = acc[b];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) out[(size_t)b * ld_out + n] = v + (bias ? bias[n] : 0.f);
    }
  }
}

constexpr int CI_PIX = 16;
constexpr int CI_MAXK = 9 * 16;   // Cin <= 16
// x NCHW fp32 -> out NHWC fp32, 3x3 pad 1
// Round 6: the GroupNorm statistics of the output for up to two consuming GroupNorms (ConvInStats; the same fixed-point accumulators the igemm
// epilogues feed: per thread the {sum, sum of squares} of its channel over the block's 16 pixels, combined per group in LDS as integers, one
// atomic set per (group, word) and block) -- the two GroupNorms that read conv_in's output (input_blocks.1.0, and output_blocks.<last>.0 through the
// skip concat) no longer need the statistics kernel.  A block's 16 pixels lie inside one sample (H * W % 16 == 0: the launcher checks).
struct ConvInStats {
  int n = 0;
  long long* acc[2] = {nullptr, nullptr};
  int cpg[2] = {1, 1}, cbase[2] = {0, 0};
};
__global__ void __launch_bounds__(256) conv_in_kernel(const float* x, const float* w, const float* bias, float* out, int B,
                                                      int Cin, int H, int W, int Cout, const ConvInStats st) {
  __shared__ float patch[CI_PIX][CI_MAXK];
  __shared__ unsigned long long s_gn[2][32][GN_WORDS];
  if (st.n > 0) {
    for (int i = threadIdx.x; i < 2 * 32 * GN_WORDS; i += 256) (&s_gn[0][0][0])[i] = 0ull;
  }
  const int tid = threadIdx.x;
  const int K = Cin * 9;
  const int M = B * H * W;
  const int m0 = blockIdx.x * CI_PIX;
  for (int idx = tid; idx < CI_PIX * K; idx += 256) {
    const int pi = idx / K, k = idx - pi * K;      // k = ci*9 + ky*3 + kx  (OIHW order of the reference weight)
    const int m = m0 + pi;
    float v = 0.f;
    if (m < M) {
      const int b = m / (H * W), rem = m - b * H * W, y = rem / W, xx = rem - y * W;
      const int ci = k / 9, t = k - ci * 9, ky = t / 3, kx = t - ky * 3;
      const int iy = y + ky - 1, ix = xx + kx - 1;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((size_t)b * Cin + ci) * H + iy) * W + ix];
    }
    patch[pi][k] = v;
  }
  __syncthreads();
  for (int co = tid; co < Cout; co += 256) {
    float acc[CI_PIX];
    const float bv = bias ? bias[co] : 0.f;
#pragma unroll
    for (int pi = 0; pi < CI_PIX; ++pi) acc[pi] = bv;
    const float* wr = w + (size_t)co * K;
    for (int k = 0; k < K; ++k) {
      const float wv = wr[k];
#pragma unroll
      for (int pi = 0; pi < CI_PIX; ++pi) acc[pi] += wv * patch[pi][k];
    }
#pragma unroll
    for (int pi = 0; pi < CI_PIX; ++pi)
      if (m0 + pi < M) out[(size_t)(m0 + pi) * Cout + co] = acc[pi];
    if (st.n > 0) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int pi = 0; pi < CI_PIX; ++pi)
        if (m0 + pi < M) { s1 += acc[pi]; s2 += acc[pi] * acc[pi]; }
      for (int tg = 0; tg < st.n; ++tg) {
        const int g = (st.cbase[tg] + co) / st.cpg[tg];
        gn_acc_add(&s_gn[tg][g][0], s1);
        gn_acc_add(&s_gn[tg][g][2], s2);
      }
    }
  }
  if (st.n > 0) {
    __syncthreads();
    const int b = m0 / (H * W), slot = blockIdx.x & (GN_SLOTS - 1);
    for (int e = tid; e < st.n * 32 * GN_WORDS; e += 256) {
      const unsigned long long wv = (&s_gn[0][0][0])[e];
      if (wv == 0ull) continue;
      const int word = e % GN_WORDS, g = (e / GN_WORDS) % 32, tg = e / (GN_WORDS * 32);
      atomicAdd((unsigned long long*)st.acc[tg] + ((size_t)(b * 32 + g) * GN_SLOTS + slot) * GN_STRIDE + word, wv);
    }
  }
}

constexpr int CO_MAXN = 8;
// h NHWC fp32 (already GroupNorm+SiLU'd), w [Cout][3][3][Cin] fp32 -> out NCHW fp32; one wave per output pixel
__global__ void __launch_bounds__(256) conv_out_kernel(const float* h, const float* w, const float* bias, float* out, int B,
                                                       int H, int W, int Cin, int Cout) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int M = B * H * W;
  if (m >= M) return;
  const int b = m / (H * W), rem = m - b * H * W, y = rem / W, x = rem - y * W;
  float acc[CO_MAXN];
#pragma unroll
  for (int n = 0; n < CO_MAXN; ++n) acc[n] = 0.f;
  const int K = 9 * Cin;
  for (int tap = 0; tap < 9; ++tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const int iy = y + ky - 1, ix = x + kx - 1;
    if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
    const float* src = h + ((size_t)(b * H + iy) * W + ix) * Cin;
    for (int c = lane * 4; c < Cin; c += 256) {
      const f32x4 xv = *(const f32x4*)(src + c);
#pragma unroll
      for (int n = 0; n < CO_MAXN; ++n) {
        if (n < Cout) {
          const f32x4 wv = *(const f32x4*)(w + (size_t)n * K + tap * Cin + c);
          acc[n] += wv[0] * xv[0] + wv[1] * xv[1] + wv[2] * xv[2] + wv[3] * xv[3];
        }
      }
    }
  }
#pragma unroll
  for (int n = 0; n < CO_MAXN; ++n) {
    if (n < Cout) {
      float v = acc[n];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) out[(((size_t)b * Cout + n) * H + y) * W + x] = v + (bias ? bias[n] : 0.f);
    }
  }
}

// The same convolution with one wave per COP = 4 consecutive pixels of an image row: a weight quad is loaded once per wave
// and tap instead of once per pixel and tap, and an input pixel once per wave and kernel row instead of once per output
// pixel and tap (27 vector loads per pixel instead of 90: the one-wave-per-pixel kernel moves 4.6 KB of weights through
// the vector-memory path per pixel and tap row).  Every accumulator sees its terms in conv_out_kernel's order -- taps
// ascending, channel quads ascending inside a tap, out-of-image taps skipped; the compiler contracts the FMAs differently,
// so the two kernels agree to a few fp32 ulp, not bit for bit (this is the last operation of the UNet / first stage:
// nothing downstream rounds the difference up).  UNet `out` head: 41.8 -> 22.9 us (profiles/small_kernels_r02.txt).
// NOUT = 4 or 8 (>= Cout), Cin <= 512 (two channel passes of 64 lanes x 4), W % COP == 0.
constexpr int COP = 4;
template <int NOUT>
__global__ void __launch_bounds__(256) conv_out4_kernel(const float* h, const float* w, const float* bias, float* out, int B,
                                                        int H, int W, int Cin, int Cout) {
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);       // pixel group
  const int wq = W / COP;
  if (g >= B * H * wq) return;
  const int b = g / (H * wq), rem = g - b * H * wq, y = rem / wq, x0 = (rem - y * wq) * COP;
  const int K = 9 * Cin;
  const int c0 = lane * 4, c1 = lane * 4 + 256;
  const bool v0 = c0 < Cin, v1 = c1 < Cin;
  float acc[COP][NOUT];
#pragma unroll
  for (int p = 0; p < COP; ++p)
#pragma unroll
    for (int n = 0; n < NOUT; ++n) acc[p][n] = 0.f;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = y + ky - 1;
    if (iy < 0 || iy >= H) continue;                       // (wave-uniform)
    f32x4 xa[COP + 2], xb[COP + 2];                        // input columns x0 - 1 .. x0 + COP of this row, both channel passes
#pragma unroll
    for (int j = 0; j < COP + 2; ++j) {
      const int ix = x0 - 1 + j;
      const bool in = ix >= 0 && ix < W;
      const float* src = h + ((size_t)(b * H + iy) * W + (in ? ix : 0)) * Cin;
      xa[j] = (in && v0) ? *(const f32x4*)(src + c0) : zero;
      xb[j] = (in && v1) ? *(const f32x4*)(src + c1) : zero;
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int tap = ky * 3 + kx;
      f32x4 wa[NOUT], wb[NOUT];
#pragma unroll
      for (int n = 0; n < NOUT; ++n) {
        const float* wp = w + (size_t)(n < Cout ? n : 0) * K + tap * Cin;
        wa[n] = v0 ? *(const f32x4*)(wp + c0) : zero;
        wb[n] = v1 ? *(const f32x4*)(wp + c1) : zero;
      }
#pragma unroll
      for (int p = 0; p < COP; ++p) {
        const int ix = x0 + p + kx - 1;
        if (ix < 0 || ix >= W) continue;                   // (wave-uniform) out-of-image tap: skipped, as in conv_out_kernel
        const f32x4 xv = xa[p + kx], xw = xb[p + kx];
        if (v0) {
#pragma unroll
          for (int n = 0; n < NOUT; ++n) acc[p][n] += wa[n][0] * xv[0] + wa[n][1] * xv[1] + wa[n][2] * xv[2] + wa[n][3] * xv[3];
        }
        if (v1) {
#pragma unroll
          for (int n = 0; n < NOUT; ++n) acc[p][n] += wb[n][0] * xw[0] + wb[n][1] * xw[1] + wb[n][2] * xw[2] + wb[n][3] * xw[3];
        }
      }
    }
  }
#pragma unroll
  for (int n = 0; n < NOUT; ++n) {
    f32x4 r;
#pragma unroll
    for (int p = 0; p < COP; ++p) {
      float v = acc[p][n];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
      r[p] = v + ((bias && n < Cout) ? bias[n] : 0.f);
    }
    if (lane == 0 && n < Cout) *(f32x4*)(out + (((size_t)b * Cout + n) * H + y) * W + x0) = r;
  }
}

// ---- packing -------------------------------------------------------------------------------------------
__global__ void pack_conv_kernel(const float* w, f16* dst, int O, int I, int KH, int KW) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)O * I * KH * KW;
  if (idx >= total) return;
  // K order (chunk-major): k = ((i / 64) * KH*KW + ky*KW + kx) * 64 + i % 64 -- all 9 taps of a 64-channel chunk are
  // adjacent, so a kernel that stages one input-channel chunk (with its halo) consumes 9 consecutive k-tiles.
  // For 1x1 convs this is the identity order.  3x3 convs need I % 64 == 0 (checked by the GEMM launcher).
  const int64_t K = (int64_t)I * KH * KW;
  const int o = (int)(idx / K);
  const int64_t k = idx - (int64_t)o * K;
  int i, ky, kx;
  if (KH * KW == 1) { i = (int)k; ky = kx = 0; }
  else {
    const int chunk = (int)(k / (64 * KH * KW));
    const int rem = (int)(k - (int64_t)chunk * 64 * KH * KW);
    const int tap = rem / 64;
    i = chunk * 64 + rem % 64; ky = tap / KW; kx = tap % KW;
  }
  dst[idx] = (f16)w[(((int64_t)o * I + i) * KH + ky) * KW + kx];
}
// 3x3 conv weights of a 3-pass split-fp16 convolution (UNet: the last ResBlock, round 6): the K-concatenated operand A' = [a_hi | a_lo | a_hi]
// (three channel blocks of I) meets W' = [w_hi | w_hi | w_lo] -- a virtual [O][3 I][KH][KW] tensor in pack_conv_kernel's chunk-major K order
__global__ void pack_conv_split3_kernel(const float* w, f16* dst, int O, int I, int KH, int KW) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int I3 = 3 * I;
  const int64_t K = (int64_t)I3 * KH * KW;
  if (idx >= (int64_t)O * K) return;
  const int o = (int)(idx / K);
  const int64_t k = idx - (int64_t)o * K;
  const int chunk = (int)(k / (64 * KH * KW));
  const int rem = (int)(k - (int64_t)chunk * 64 * KH * KW);
  const int tap = rem / 64;
  const int i3 = chunk * 64 + rem % 64, ky = tap / KW, kx = tap % KW;
  const int part = i3 / I, i = i3 - part * I;
  const float v = w[(((int64_t)o * I + i) * KH + ky) * KW + kx];
  const f16 hi = (f16)v;
  dst[idx] = part < 2 ? hi : (f16)(v - (float)hi);
}
__global__ void pack_conv_f32_kernel(const float* w, float* dst, int O, int I, int KH, int KW) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)O * I * KH * KW;
  if (idx >= total) return;
  const int i = (int)(idx % I);
  int64_t r = idx / I;
  const int kx = (int)(r % KW); r /= KW;
  const int ky = (int)(r % KH);
  const int o = (int)(r / KH);
  dst[idx] = w[(((int64_t)o * I + i) * KH + ky) * KW + kx];
}
__global__ void pack_rows_kernel(const float* w, f16* dst, int rows, int cols, int dst_row0, int dst_ld) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)rows * cols) return;
  const int r = (int)(idx / cols), c = (int)(idx - (int64_t)r * cols);
  dst[(int64_t)(dst_row0 + r) * dst_ld + c] = (f16)w[idx];
}
// split-fp16 weight for the 3-pass 1x1 convs: dst[n] = [hi(K) | hi(K) | lo(K)]
__global__ void pack_split3_kernel(const float* w, f16* dst, int N, int K) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * K) return;
  const int n = (int)(idx / K), k = (int)(idx - (int64_t)n * K);
  const float v = w[idx];
  const f16 hi = (f16)v;
  const f16 lo = (f16)(v - (float)hi);
  f16* d = dst + (int64_t)n * 3 * K;
  d[k] = hi; d[K + k] = hi; d[2 * K + k] = lo;
}
// GEGLU: dst row 64q + j  <- value row 32q + j (j < 32) | gate row N/2 + 32q + (j - 32)
__global__ void pack_geglu_kernel(const float* w, const float* bias, f16* wdst, float* bdst, int N, int K) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * K) return;
  const int r = (int)(idx / K), c = (int)(idx - (int64_t)r * K);
  const int q = r >> 6, j = r & 63;
  const int src = (j < 32) ? (32 * q + j) : (N / 2 + 32 * q + (j - 32));
  wdst[idx] = (f16)w[(int64_t)src * K + c];
  if (c == 0 && bias) bdst[r] = bias[src];
}

}  // namespace

// frequency table of the sinusoidal embedding, computed on the host exactly like util.py:160-162
//   freqs = exp(-log(max_period) * arange(half, float32) / half)   (fp32 ops)
static float* g_freqs[16] = {nullptr};
static int g_freqs_half[16] = {0};
static int get_freqs(int half, const float** out) {
  int dev = 0;
  SDMI_HIP_OK(hipGetDevice(&dev));
  SDMI_CHECK(dev < 16, "device index");
  if (g_freqs[dev] == nullptr || g_freqs_half[dev] != half) {
    if (g_freqs[dev]) (void)hipFree(g_freqs[dev]);
    std::vector<float> f(half);
    const float nl = (float)(-9.210340371976184);   // -math.log(10000), cast to the tensor dtype (fp32) as torch does
    for (int i = 0; i < half; ++i) f[i] = expf(nl * (float)i / (float)half);
    SDMI_HIP_OK(hipMalloc((void**)&g_freqs[dev], half * sizeof(float)));
    SDMI_HIP_OK(hipMemcpy(g_freqs[dev], f.data(), half * sizeof(float), hipMemcpyHostToDevice));
    g_freqs_half[dev] = half;
  }
  *out = g_freqs[dev];
  return 0;
}


// ---- text-encoder helpers ---------------------------------------------------------------------------------------------
// CLIPTextEmbeddings (transformers models/clip/modeling_clip.py): token_embedding(ids) + position_embedding(arange(L))
__global__ void __launch_bounds__(256) embed_tokens_kernel(const int64_t* ids, const float* tok, const float* pos, float* out,
                                                           int M, int L, int C, int vocab) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int nq = C / 4;
  if (idx >= (int64_t)M * nq) return;
  const int m = (int)(idx / nq), c = (int)(idx - (int64_t)m * nq) * 4;
  int64_t id = ids[m];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);              // the host shim validates the range; never read out of bounds
  *(f32x4*)(out + (size_t)m * C + c) = *(const f32x4*)(tok + (size_t)id * C + c) + *(const f32x4*)(pos + (size_t)(m % L) * C + c);
}

// quick_gelu (CLIP's hidden_act): x * sigmoid(1.702 x), fp32 in -> fp16 out
__global__ void __launch_bounds__(256) quick_gelu_kernel(const float* x, f16* out, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = *(const f32x4*)(x + i * 4);
  f16x4 y;
#pragma unroll
  for (int j = 0; j < 4; ++j) y[j] = (f16)(v[j] / (1.0f + __expf(-1.702f * v[j])));
  *(f16x4*)(out + i * 4) = y;
}

// ---- first-stage (VAE) helpers ---------------------------------------------------------------------------------------
// 1x1 conv on a few channels, NCHW fp32 -> NCHW fp32, input optionally pre-scaled:
// post_quant_conv / quant_conv (ldm/models/autoencoder.py:302-303) and the 1/scale_factor of decode_first_stage
constexpr int PW_MAXC = 16;
__global__ void __launch_bounds__(256) pointwise_nchw_kernel(const float* x, const float* w, const float* bias, float* out,
                                                             int B, int Cin, int Cout, int HW, float in_scale) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * HW) return;
  const int b = (int)(idx / HW), pix = (int)(idx - (int64_t)b * HW);
  float v[PW_MAXC];
#pragma unroll
  for (int c = 0; c < PW_MAXC; ++c) v[c] = (c < Cin) ? x[((size_t)b * Cin + c) * HW + pix] * in_scale : 0.f;
  for (int o = 0; o < Cout; ++o) {
    float acc = bias ? bias[o] : 0.f;
#pragma unroll
    for (int c = 0; c < PW_MAXC; ++c)
      if (c < Cin) acc += w[o * Cin + c] * v[c];
    out[((size_t)b * Cout + o) * HW + pix] = acc;
  }
}

// P[r][c] = softmax_c(S[r][c] * scale), fp32 in -> fp16 out; one block per row (AttnBlock, model.py:186-187)
__global__ void __launch_bounds__(256) softmax_rows_kernel(const float* S, f16* P, int cols, int lds, int ldp, float scale) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* src = S + (size_t)blockIdx.x * lds;
  f16* dst = P + (size_t)blockIdx.x * ldp;
  float mx = -INFINITY;
  for (int c = tid * 4; c < cols; c += 1024) {
    const f32x4 v = *(const f32x4*)(src + c);
    mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale;
  float sum = 0.f;
  for (int c = tid * 4; c < cols; c += 1024) {
    const f32x4 v = *(const f32x4*)(src + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += __expf(v[j] * scale - mx);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int c = tid * 4; c < cols; c += 1024) {
    const f32x4 v = *(const f32x4*)(src + c);
    f16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (f16)(__expf(v[j] * scale - mx) * inv);
    *(f16x4*)(dst + c) = o;
  }
}

int launch_timestep_embedding(const int64_t* t_i64, const float* t_f32, float* out, int B, int dim, hipStream_t s) {
  SDMI_CHECK((t_i64 != nullptr) != (t_f32 != nullptr), "exactly one of int64 / fp32 timesteps");
  const float* freqs = nullptr;
  if (get_freqs(dim / 2, &freqs)) return -1;
  const int total = B * (dim / 2);
  SDMI_LAUNCH(temb_kernel, dim3(cdiv(total, 128)), dim3(128), 0, s, t_i64, t_f32, freqs, out, B, dim);
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}

int launch_small_linear(const float* in, int ld_in, const float* w, const float* bias, float* out, int ld_out, int B,
                        int N, int K, int silu_in, hipStream_t s) {
  SDMI_CHECK(B >= 1 && B <= SL_MAXB, "small_linear: batch must be 1..8");
  SDMI_CHECK(K % 4 == 0 && ld_in % 4 == 0, "small_linear: K % 4");
  ProfScope ps("small_linear_f32", 2.0 * B * (double)N * K, (double)N * K * 4.0, s);
  const char* e_lds = getenv("SDMI_SMALL_LDS");                  // A/B knob, read per launch (3 per UNet call); bit-identical results
  const int env_lds = e_lds ? atoi(e_lds) : 1;
  const size_t lds = (size_t)B * K * sizeof(float);
  const bool al16 = ((((uintptr_t)in | (uintptr_t)w) & 15) == 0);
  if (env_lds && K <= 1280 && lds <= 64 * 1024 && al16) {           // every input row of the block in LDS, <= 5 weight quads per lane
    if (K <= 512) SDMI_LAUNCH(small_linear_lds_kernel<2>, dim3(cdiv(N, 8)), dim3(512), lds, s, in, ld_in, w, bias, out, ld_out, B, N, K, silu_in);
    else SDMI_LAUNCH(small_linear_lds_kernel<5>, dim3(cdiv(N, 8)), dim3(512), lds, s, in, ld_in, w, bias, out, ld_out, B, N, K, silu_in);
  } else {
    SDMI_LAUNCH(small_linear_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, in, ld_in, w, bias, out, ld_out, B, N, K,
                       silu_in);
  }
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}

int launch_conv_in(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int H, int W, int Cout,
                   hipStream_t s, int gn_n, long long* const* gn_acc, const int* gn_cpg, const int* gn_cbase) {
  SDMI_CHECK(Cin * 9 <= CI_MAXK, "conv_in: in_channels <= 16");
  ConvInStats st;
  if (gn_n > 0) {
    SDMI_CHECK(gn_n <= 2 && (H * W) % CI_PIX == 0, "conv_in statistics: at most two GroupNorm targets, H * W % 16 == 0");
    st.n = gn_n;
    for (int i = 0; i < gn_n; ++i) { st.acc[i] = gn_acc[i]; st.cpg[i] = gn_cpg[i]; st.cbase[i] = gn_cbase[i]; }
  }
  ProfScope ps("conv_in_f32", 2.0 * B * H * W * (double)Cout * Cin * 9, (double)B * H * W * (Cin + Cout) * 4.0, s);
  SDMI_LAUNCH(conv_in_kernel, dim3(cdiv(B * H * W, CI_PIX)), dim3(256), 0, s, x, w, bias, out, B, Cin, H, W, Cout, st);
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}

int launch_conv_out(const float* h, const float* w, const float* bias, float* out, int B, int H, int W, int Cin, int Cout,
                    hipStream_t s) {
  SDMI_CHECK(Cout <= CO_MAXN && Cin % 4 == 0, "conv_out: out_channels <= 8, Cin % 4 == 0");
  ProfScope ps("conv_out_f32", 2.0 * B * H * W * (double)Cout * Cin * 9, (double)B * H * W * (Cin + Cout) * 4.0, s);
  // (an LDS-resident-weights variant, 32 pixels per block, was measured at 169 us vs 64 us: one wave per pixel keeps 1024
  // independent waves in flight, which this latency-bound fp32 kernel needs more than it needs fewer weight re-reads)
  const char* e_co4 = getenv("SDMI_CONV_OUT4");                  // A/B knob, read per launch (1 per UNet call)
  const int env_co4 = e_co4 ? atoi(e_co4) : 1;
  const bool al16 = ((((uintptr_t)h | (uintptr_t)w | (uintptr_t)out) & 15) == 0);
  if (env_co4 && W % COP == 0 && Cin <= 512 && al16) {       // (out rows of W floats at x0 % 4 == 0: 16-byte aligned stores)
    const int groups = B * H * (W / COP);
    if (Cout <= 4) SDMI_LAUNCH(conv_out4_kernel<4>, dim3(cdiv(groups, 4)), dim3(256), 0, s, h, w, bias, out, B, H, W, Cin, Cout);
    else SDMI_LAUNCH(conv_out4_kernel<8>, dim3(cdiv(groups, 4)), dim3(256), 0, s, h, w, bias, out, B, H, W, Cin, Cout);
  } else {
    SDMI_LAUNCH(conv_out_kernel, dim3(cdiv(B * H * W, 4)), dim3(256), 0, s, h, w, bias, out, B, H, W, Cin, Cout);
  }
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}


int launch_embed_tokens(const int64_t* ids, const float* tok_emb, const float* pos_emb, float* out, int M, int L, int C,
                        int vocab, hipStream_t s) {
  SDMI_CHECK(C % 4 == 0 && M >= 1 && L >= 1 && vocab >= 1, "embed_tokens: C % 4");
  const int64_t total = (int64_t)M * (C / 4);
  SDMI_LAUNCH(embed_tokens_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ids, tok_emb, pos_emb, out, M,
                     L, C, vocab);
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}

int launch_quick_gelu(const float* x, f16* out, int64_t n, hipStream_t s) {
  SDMI_CHECK(n % 4 == 0, "quick_gelu: n % 4");
  SDMI_LAUNCH(quick_gelu_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, x, out, n / 4);
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}

int launch_pointwise_nchw(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int HW,
                          float in_scale, hipStream_t s) {
  SDMI_CHECK(Cin >= 1 && Cin <= PW_MAXC && Cout >= 1, "pointwise conv: 1 <= Cin <= 16");
  const int64_t total = (int64_t)B * HW;
  SDMI_LAUNCH(pointwise_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, w, bias, out, B, Cin,
                     Cout, HW, in_scale);
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}

int launch_softmax_rows(const float* S, f16* P, int rows, int cols, int lds, int ldp, float scale, hipStream_t s) {
  SDMI_CHECK(cols % 4 == 0 && lds % 4 == 0 && ldp % 4 == 0 && rows >= 1, "softmax rows: cols % 4");
  ProfScope ps("softmax_rows", 0.0, (double)rows * cols * 6.0, s);
  SDMI_LAUNCH(softmax_rows_kernel, dim3(rows), dim3(256), 0, s, S, P, cols, lds, ldp, scale);
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}

int launch_pack_conv_weight(const float* w, f16* dst, int O, int I, int KH, int KW, hipStream_t s) {
  const int64_t total = (int64_t)O * I * KH * KW;
  SDMI_LAUNCH(pack_conv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, dst, O, I, KH, KW);
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}
// Touch one dword of every 128-byte line of [ptr, ptr + bytes): pulls the range through this XCD's L2 into the Infinity Cache.
// The loaded values are only kept alive (never stored).  A cache hint: no correctness role.
__global__ void __launch_bounds__(256) prefetch_lines_kernel(const unsigned* p, long long nlines) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nlines) return;
  const unsigned v = __builtin_nontemporal_load(p + i * 32);
  asm volatile("" ::"v"(v));
}
int launch_prefetch_lines(const void* ptr, int64_t bytes, hipStream_t s) {
  const long long nlines = bytes / 128;
  if (nlines <= 0) return 0;
  SDMI_LAUNCH(prefetch_lines_kernel, dim3((unsigned)((nlines + 255) / 256)), dim3(256), 0, s, (const unsigned*)ptr, nlines);
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}

int launch_pack_conv_split3(const float* w, f16* dst, int O, int I, int KH, int KW, hipStream_t s) {
  SDMI_CHECK(I % 64 == 0, "split-fp16 conv pack: input channels % 64");
  const int64_t total = (int64_t)O * 3 * I * KH * KW;
  SDMI_LAUNCH(pack_conv_split3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, dst, O, I, KH, KW);
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}
int launch_pack_conv_out(const float* w, float* dst, int O, int I, hipStream_t s) {
  const int64_t total = (int64_t)O * I * 9;
  SDMI_LAUNCH(pack_conv_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, dst, O, I, 3, 3);
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}
int launch_pack_rows(const float* w, f16* dst, int rows, int cols, int dst_row0, int dst_ld, hipStream_t s) {
  const int64_t total = (int64_t)rows * cols;
  SDMI_LAUNCH(pack_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, dst, rows, cols,
                     dst_row0, dst_ld);
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}

// LayerNorm folded into the consuming GEMM (attention.py:211-215; IGemmParams::lnf_*): the two column vectors, computed from the
// fp16 weights the MFMAs will multiply (so that mean * cs cancels the mean's share of the accumulator exactly as far as the
// weights go).  One wave per weight row, lanes stride over K, fixed butterfly: deterministic.
__global__ void __launch_bounds__(256) ln_fold_prep_kernel(const f16* w, int N, int K, int ldw, const float* gamma, const float* beta,
                                                           const float* bias, float* cs, float* d) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const f16* row = w + (size_t)n * ldw;
  float a = 0.f, b = 0.f;
  for (int k = lane; k < K; k += 64) { const float wv = (float)row[k]; a = fmaf(gamma[k], wv, a); b = fmaf(beta[k], wv, b); }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
  if (lane == 0) { cs[n] = a; d[n] = b + (bias ? bias[n] : 0.f); }
}
int launch_ln_fold_prep(const f16* w, int N, int K, int ldw, const float* gamma, const float* beta, const float* bias, float* cs,
                        float* d, hipStream_t s) {
  SDMI_CHECK(w && gamma && beta && cs && d && N > 0 && K > 0 && ldw >= K, "ln_fold_prep: bad arguments");
  SDMI_LAUNCH(ln_fold_prep_kernel, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, s, w, N, K, ldw, gamma, beta, bias, cs, d);
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}
int launch_pack_split3(const float* w, f16* dst, int N, int K, hipStream_t s) {
  const int64_t total = (int64_t)N * K;
  SDMI_LAUNCH(pack_split3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, dst, N, K);
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}
int launch_pack_geglu(const float* w, const float* bias, f16* wdst, float* bdst, int N, int K, hipStream_t s) {
  SDMI_CHECK(N % 64 == 0, "GEGLU pack: N % 64");
  const int64_t total = (int64_t)N * K;
  SDMI_LAUNCH(pack_geglu_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, bias, wdst, bdst, N, K);
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}

}  // namespace sdmi
