// Implicit-GEMM convolution / linear kernel for gfx950 (MI355X).
//
//   out[M,N] = epilogue( gatherA[M,K] * W[N,K]^T )
//
// Replaces the reference's F.conv2d / F.linear call sites on the UNet hot path
// (ldm/modules/diffusionmodules/openaimodel.py:204,230,241,150-153,116-118 and
//  ldm/modules/attention.py:161-168,40,58,233-248) -- SURVEY.md K4-K9, K11, K14.
//
// Design (wave64, MFMA v_mfma_f32_32x32x16_f16, fp32 accumulate):
//   * A (activations, fp16 NHWC) is gathered on the fly: 1x1 / 3x3, stride 1/2, nearest-x2 upsample
//     folded into the index math, two channel-concatenated sources (UNet skip concat) -- nothing is
//     materialised.  Out-of-image taps read a zero page.
//   * k-tile = 64 halves; both operands are staged as [rows][128 B] LDS tiles, double buffered, either by
//     LDS-DMA (global_load_lds_dwordx4: wave-uniform LDS base + lane*16, per-lane global source) or through
//     registers.  16-byte chunks are XOR-swizzled with ((row>>1)&7) so the ds_read_b128 fragment reads of
//     any 16 rows that differ mod 16 are bank-conflict free; with DMA the swizzle is applied to the
//     per-lane *source* chunk and to the read address (LDS destination stays linear).
//   * one barrier per k-tile: loads for tile t+1 are issued right after the barrier and land while the
//     MFMAs of tile t run.
//   * epilogues: +bias[n] +rowvec[batch][n] (time-embedding add) +fp32 residual, fp32 and/or fp16 store;
//     GEGLU (value * gelu_erf(gate), weights pre-interleaved in 32-row groups); per-head scatter of
//     q / k / v^T for the attention kernel; split-K into per-split fp32 slabs summed in a fixed order (reduce kernel).
//   * blockIdx is remapped so that each XCD (private L2) owns a contiguous range of tiles.
#include <dlfcn.h>

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "igemm_dev.h"

namespace sdmi {
namespace {

// out = sum_s slab[s] + bias + rowvec[batch] + residual   (fixed summation order -> deterministic)
// A block owns a strip of 32 columns x `rows_per_block` rows (thread = one 16-byte quad of a row; 8 threads cover a
// 128-byte line), walking the rows 32 at a time.  With GroupNorm statistics (see the GEMM epilogue): the strip lies inside
// one sample (rows_per_block divides Hout*Wout) and touches at most 32 / cpg + 2 groups; the per-thread sums are combined
// as fixed-point int64 in LDS (integer adds: order independent) and leave the block as one global atomic set per group.
__global__ void __launch_bounds__(256) splitk_reduce_kernel(IGemmParams p, int nsplit, int rows_per_block) {
  __shared__ unsigned long long s_gn[2][32][GN_WORDS];        // [target][group]
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int n = (blockIdx.x * 8 + tx) * 4;
  const int row0 = blockIdx.y * rows_per_block;
  const int HW = p.Hout * p.Wout;
  const bool gn = p.gn_n > 0;
  const bool ncol = n < p.N;
  if (gn) {
    for (int i = threadIdx.x; i < 2 * 32 * GN_WORDS; i += 256) (&s_gn[0][0][0])[i] = 0ull;
    __syncthreads();
  }
  const size_t slab_sz = (size_t)p.M * p.N;
  f32x4 biasv = {0.f, 0.f, 0.f, 0.f};
  if (ncol && p.bias) biasv = *(const f32x4*)(p.bias + n);
  float a1[2] = {0.f, 0.f}, a2[2] = {0.f, 0.f}, c1[2] = {0.f, 0.f}, c2[2] = {0.f, 0.f};
  int g0[2] = {0, 0}, g1[2] = {0, 0}, gsplit[2] = {4, 4};
  for (int t = 0; t < p.gn_n; ++t) {
    const int c = p.gn_cbase[t] + n;
    g0[t] = fast_div(c, p.gn_magic[t]); g1[t] = fast_div(c + 3, p.gn_magic[t]);
    gsplit[t] = (g0[t] + 1) * p.gn_cpg[t] - c;                  // first of the 4 channels that belongs to g1
  }
  for (int r = ty; r < rows_per_block; r += 32) {
    const int m = row0 + r;
    if (!ncol || m >= p.M) continue;
    const float* src = p.splitk_ws + (size_t)m * p.N + n;
    // every load of the row is requested before the first wait: the row vector and the residual used to be fetched
    // behind the partial sums, one dependent round trip each (three per row instead of one)
    f32x4 rvv = {0.f, 0.f, 0.f, 0.f}, resv = {0.f, 0.f, 0.f, 0.f};
    if (p.rowvec) rvv = *(const f32x4*)(p.rowvec + (size_t)(m / HW) * p.ld_rowvec + n);
    if (p.residual) resv = *(const f32x4*)(p.residual + (size_t)m * p.ldr + n);
    f32x4 part[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) part[s] = (s < nsplit) ? *(const f32x4*)(src + s * slab_sz) : f32x4{0, 0, 0, 0};
    f32x4 v = part[0];
#pragma unroll
    for (int s = 1; s < 16; ++s) v += part[s];       // fixed order; absent splits add +0
    v += biasv;
    if (p.rowvec) v += rvv;
    if (p.residual) v += resv;
    if (p.out_f32) SDMI_ST_F32X4(p.out_f32, (size_t)m * p.ldo + n, v);
    if (p.out_f16) SDMI_ST_F16X4(p.out_f16, (size_t)m * p.ldo + n, (f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]}));
    if (p.out_lo) {
      f16x4 lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) lo[j] = (f16)(v[j] - (float)(f16)v[j]);
      *(f16x4*)(p.out_lo + (size_t)m * p.ldo + n) = lo;
    }
    if (gn) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j < gsplit[t]) { a1[t] += v[j]; a2[t] += v[j] * v[j]; } else { c1[t] += v[j]; c2[t] += v[j] * v[j]; }
        }
      }
    }
  }
  if (gn) {
    if (ncol) {
      for (int t = 0; t < p.gn_n; ++t) {
        gn_acc_add(&s_gn[t][g0[t]][0], a1[t]);
        gn_acc_add(&s_gn[t][g0[t]][2], a2[t]);
        if (g1[t] != g0[t]) { gn_acc_add(&s_gn[t][g1[t]][0], c1[t]); gn_acc_add(&s_gn[t][g1[t]][2], c2[t]); }
      }
    }
    __syncthreads();
    const int b = row0 / HW;                              // the strip lies inside one sample
    const int slot = (blockIdx.x + blockIdx.y) & (GN_SLOTS - 1);
    for (int e = threadIdx.x; e < p.gn_n * 32 * GN_WORDS; e += 256) {
      const unsigned long long w = (&s_gn[0][0][0])[e];
      if (w == 0ull) continue;
      const int word = e % GN_WORDS, g = (e / GN_WORDS) % 32, t = e / (GN_WORDS * 32);
      atomicAdd((unsigned long long*)p.gn_acc[t] + ((size_t)(b * 32 + g) * GN_SLOTS + slot) * GN_STRIDE + word, w);
    }
  }
}

// the same reduction for the per-head scatter epilogue: quad (m, n..n+3) lies inside one head (dh % 4 == 0)
__global__ void __launch_bounds__(256) splitk_reduce_heads_kernel(IGemmParams p, int nsplit) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int nq = p.N / 4;
  if (idx >= (int64_t)p.M * nq) return;
  const int m = (int)(idx / nq);
  const int n = (int)(idx - (int64_t)m * nq) * 4;
  const size_t slab_sz = (size_t)p.M * p.N;
  const float* src = p.splitk_ws + (size_t)m * p.N + n;
  f32x4 part[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) part[s] = (s < nsplit) ? *(const f32x4*)(src + s * slab_sz) : f32x4{0, 0, 0, 0};
  f32x4 v = part[0];
#pragma unroll
  for (int s = 1; s < 16; ++s) v += part[s];
  if (p.bias) v += *(const f32x4*)(p.bias + n);
  const int seg = n / p.segC;
  const int c = n - seg * p.segC;
  const int head = c / p.dh;
  const int dd = c - head * p.dh;
  const int b = m / p.ntok;
  const int tok = m - b * p.ntok;
  const size_t bh = (size_t)b * p.heads + head;
  f16* dst = p.seg_dst[seg];
  if (p.seg_kind[seg] == 0) {
    *(f16x4*)(dst + (bh * p.ntok + tok) * p.dh + dd) = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[(bh * p.dh + dd + j) * p.ntok_pad + tok] = (f16)v[j];
  }
}

// 4 x 4 transpose between the four lanes of a quad: in, lane (sub = lane & 3) holds v[e] = element (row e, column sub) of a 4 x 4
// block; out, it holds w[c] = element (row sub, column c).  Round k: every lane offers v[(sub + k) & 3], lane d takes the offer of
// lane (d - k) & 3 -- DPP quad_perm, a VALU modifier: no LDS traffic.
__device__ __forceinline__ float pick4(const f32x4 v, int idx) {
  return idx == 0 ? v[0] : (idx == 1 ? v[1] : (idx == 2 ? v[2] : v[3]));
}
template <int CTRL>
__device__ __forceinline__ float quad_dpp(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ f32x4 quad_transpose(const f32x4 v, int sub) {
  const float r0 = pick4(v, sub);
  const float r1 = quad_dpp<0x93>(pick4(v, (sub + 1) & 3));      // quad_perm [3, 0, 1, 2]: lane d reads lane d - 1
  const float r2 = quad_dpp<0x4E>(pick4(v, (sub + 2) & 3));      // [2, 3, 0, 1]
  const float r3 = quad_dpp<0x39>(pick4(v, (sub + 3) & 3));      // [1, 2, 3, 0]
  const f32x4 r = {r0, r1, r2, r3};                               // r[k] = element (row sub, column (sub - k) & 3)
  return f32x4{pick4(r, sub & 3), pick4(r, (sub - 1) & 3), pick4(r, (sub - 2) & 3), pick4(r, (sub - 3) & 3)};
}
#ifdef SDMI_EXPERIMENTS      // (bit-identical, 15 launches fewer, measured slower in round 4: profiles/splitk_slabs_r04.txt)
// Split-K reduction + GroupNorm(32) (+ SiLU) of the result in ONE launch (IGemmParams::pgn_*; ResBlock conv1 -> out_layers'
// GroupNorm -> SiLU, openaimodel.py:225-231, at the levels where conv1 is split: 8x8, 16x16, the concat blocks of 32x32).
// Workgroup (g, b) owns group g of sample b: HW rows x cpg = N / 32 channels.  Thread t handles the 16-byte quads t, t + 1024, ...
// of that block (quad = 4 channels of a row): it sums their nsplit partials in slab order, + bias + rowvec + residual -- the
// operations of splitk_reduce_kernel in the same order, i.e. the same fp32 value v -- and keeps v in registers.
// Statistics, BIT-IDENTICAL to the path it replaces (splitk_reduce_kernel's statistics + norm.hip's fold): there, with 32 rows per
// block (every shape this kernel accepts: the launcher checks), a thread's partial is ONE quad's {((v0 + v1) + v2) + v3, the same
// over the rounded squares}, added as fixed-point int64 words -- exact, order-free -- and folded by gn_mean_rstd.  Here the same
// per-quad fp32 partials are split into the same words (gn_fixed_split), summed as integers in registers / LDS, and folded by
// the same function; every thread then normalises its quads with gn_apply_elem (the apply kernel's arithmetic) into pgn_out, the
// fp16 operand of conv2.  No statistics atomics, no GroupNorm-apply launch, and v is not written at all unless pgn_keep_f32.
__device__ __forceinline__ void quad_partials(const f32x4 a, float* q1, float* q2) {
#pragma clang fp contract(off)
  // (no fused multiply-add: splitk_reduce_kernel adds the rounded squares -- its product feeds two exec-masked branches)
  *q1 = ((a[0] + a[1]) + a[2]) + a[3];
  const float s0 = a[0] * a[0], s1 = a[1] * a[1], s2 = a[2] * a[2], s3 = a[3] * a[3];
  *q2 = ((s0 + s1) + s2) + s3;
}
// TILED: the slabs hold whole tiles in the MFMA register order (IGemmParams::slab_tiled).  Thread idx then loads the slab quad
// (4 rows x 1 column) of row group idx / cpg, column idx % cpg -- four consecutive threads = four consecutive columns of the same
// four rows (cpg % 4 == 0), a run of <= 32 columns is <= 512 contiguous bytes -- and the 4 x 4 lane transpose hands every lane the
// row-major quad (row 4 * (idx / cpg) + (idx & 3), columns 4 * ((idx % cpg) / 4) ..).  Which thread holds which quad does not matter
// to anything below (integer statistics, stores).
template <int MAXQ, bool TILED>
__global__ void __launch_bounds__(1024) splitk_reduce_gn_kernel(IGemmParams p, int nsplit, unsigned long long magic_qpr) {
  __shared__ long long s_red[16][4];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int HW = p.Hout * p.Wout;
  const int cpg = p.N >> 5, qpr = cpg >> 2;
  const int total = HW * qpr;
  const size_t slab_sz = TILED ? (size_t)(((p.M + p.slab_bm - 1) / p.slab_bm) * ((p.N + p.slab_bn - 1) / p.slab_bn)) * (size_t)(p.slab_bm * p.slab_bn)
                               : (size_t)p.M * p.N;
  f32x4 v[MAXQ];
  int rown[MAXQ][2];                                    // the quad this thread holds: row inside the sample, first column
  long long w[4] = {0, 0, 0, 0};                       // {sum int, sum frac, sumsq int, sumsq frac} of this thread's quads
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    const int idx = tid + i * 1024;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    rown[i][0] = 0; rown[i][1] = g * cpg;
    if (idx < total) {                                   // (total % 4 == 0 and idx % 4 == lane % 4: a lane quad is inside or outside as a whole)
      int row, n;
      const float* src;
      if (TILED) {
        const int row4 = fast_div(idx, p.gn_magic[0]);   // idx / cpg (the launcher checked gn_cpg[0] == cpg)
        const int c = idx - row4 * cpg;
        const int m4 = b * HW + 4 * row4, nc = g * cpg + c;
        const int BM = p.slab_bm, BN = p.slab_bn, WTM = BM / p.slab_wm, WTN = BN >> p.slab_sh_wn;
        const int tiles_n = (p.N + BN - 1) / BN;
        const int tile_m = m4 / BM, mm = m4 - tile_m * BM, wm = mm / WTM, mw = mm - wm * WTM, ii = mw >> 5, r32 = mw & 31;
        const int tile_n = nc / BN, nn = nc - tile_n * BN, wn = nn / WTN, nw = nn - wn * WTN, jj = nw >> 5, l31 = nw & 31;
        const int blk = (((ii << p.slab_sh_tn) + jj) << 2) + (r32 >> 3);
        const int t_in = (((wm << p.slab_sh_wn) + wn) << 6) + (((r32 >> 2) & 1) << 5) + l31;
        src = p.splitk_ws + (size_t)(tile_m * tiles_n + tile_n) * (size_t)(BM * BN) + (((size_t)blk << p.slab_sh_nt) + t_in) * 4;
        row = 4 * row4 + (tid & 3);
        n = g * cpg + (c & ~3);
      } else {
        row = fast_div(idx, magic_qpr);
        n = g * cpg + (idx - row * qpr) * 4;
        src = p.splitk_ws + ((size_t)b * HW + row) * p.N + n;
      }
      rown[i][0] = row; rown[i][1] = n;
      const size_t m = (size_t)b * HW + row;
      f32x4 biasv = {0.f, 0.f, 0.f, 0.f}, rvv = {0.f, 0.f, 0.f, 0.f}, resv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) biasv = *(const f32x4*)(p.bias + n);
      if (p.rowvec) rvv = *(const f32x4*)(p.rowvec + (size_t)b * p.ld_rowvec + n);
      if (p.residual) resv = *(const f32x4*)(p.residual + m * p.ldr + n);
      f32x4 part[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) part[s] = (s < nsplit) ? *(const f32x4*)(src + s * slab_sz) : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 a = part[0];
#pragma unroll
      for (int s = 1; s < 16; ++s) a += part[s];       // fixed order; absent splits add +0
      if (TILED) a = quad_transpose(a, tid & 3);       // (all four lanes of a quad are in this branch together)
      a += biasv;
      if (p.rowvec) a += rvv;
      if (p.residual) a += resv;
      v[i] = a;
      if (p.out_f32 && p.pgn_keep_f32) SDMI_ST_F32X4(p.out_f32, m * p.ldo + n, a);
      // the quad's partials as splitk_reduce_kernel forms them: plain adds over the values, plain adds over the ROUNDED squares
      float q1, q2;
      quad_partials(a, &q1, &q2);
      long long hi, lo;
      gn_fixed_split(q1, &hi, &lo); w[0] += hi; w[1] += lo;
      gn_fixed_split(q2, &hi, &lo); w[2] += hi; w[3] += lo;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) w[k] += __shfl_xor(w[k], o);      // integer adds: exact in any order
  }
  if ((tid & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) s_red[tid >> 6][k] = w[k];
  }
  __syncthreads();
  long long t[4] = {0, 0, 0, 0};
#pragma unroll
  for (int wv = 0; wv < 16; ++wv)
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] += s_red[wv][k];
  float mean, rstd;
  gn_mean_rstd(t[0], t[1], t[2], t[3], (double)cpg * (double)HW, p.pgn_eps, &mean, &rstd);
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    const int idx = tid + i * 1024;
    if (idx < total) {
      const int n = rown[i][1];
      const size_t m = (size_t)b * HW + rown[i][0];
      const f32x4 ga = *(const f32x4*)(p.pgn_gamma + n), be = *(const f32x4*)(p.pgn_beta + n);
      f16x4 y;
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = (f16)gn_apply_elem(v[i][j], mean, rstd, ga[j], be[j], p.pgn_silu);
      SDMI_ST_F16X4(p.pgn_out, m * p.N + n, y);
    }
  }
}

#endif  // SDMI_EXPERIMENTS

// ---- reduction over register-order slabs (IGemmParams::slab_tiled) -------------------------------------------------------
// a quad's statistics partials as splitk_reduce_kernel forms them with one row per thread: channels j < gsplit belong to the first
// group, the rest to the next; plain adds over the values and over the ROUNDED squares (no fused multiply-add)
__device__ __forceinline__ void quad_partials_split(const f32x4 v, int gsplit, float* a1, float* a2, float* c1, float* c2) {
#pragma clang fp contract(off)
  float s1 = 0.f, s2 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float sq = v[j] * v[j];
    if (j < gsplit) { s1 += v[j]; s2 += sq; } else { t1 += v[j]; t2 += sq; }
  }
  *a1 = s1; *a2 = s2; *c1 = t1; *c2 = t2;
}
// where thread `rem` (0 .. BM * BN / 4) of tile `tile` finds its quad, and which output quad it owns after the transpose
struct TiledQuad { size_t off; int m, n; bool valid; };
__device__ __forceinline__ TiledQuad tiled_quad(const IGemmParams& p, int q, int lane) {
  const int BM = p.slab_bm, BN = p.slab_bn;
  const int WTM = BM / p.slab_wm, WTN = BN >> p.slab_sh_wn;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tile = q >> p.slab_sh_qpt, rem = q - (tile << p.slab_sh_qpt);       // (every divisor but tiles_n is a power of two)
  const int blk = rem >> p.slab_sh_nt, t_in = rem - (blk << p.slab_sh_nt);
  const int wave = t_in >> 6;
  const int ij = blk >> 2, r4 = blk & 3, i = ij >> p.slab_sh_tn, j = ij - (i << p.slab_sh_tn);
  const int wm = wave >> p.slab_sh_wn, wn = wave - (wm << p.slab_sh_wn);
  const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
  const int l31 = lane & 31, lg = lane >> 5;
  TiledQuad t;
  t.off = (size_t)tile * (size_t)(BM * BN) + (size_t)rem * 4;
  t.m = tile_m * BM + wm * WTM + i * 32 + 8 * r4 + 4 * lg + (l31 & 3);
  t.n = tile_n * BN + wn * WTN + j * 32 + (l31 & ~3);
  t.valid = t.m < p.M && t.n < p.N;
  return t;
}
// out = sum_s slab[s] + bias + rowvec[batch] + residual: the arithmetic (and with one row per thread the statistics partials) of
// splitk_reduce_kernel, value by value.  A block = 256 consecutive slab quads = four GEMM waves' share of one 32-row slab each, so a
// wave's rows lie inside one sample: the statistics are combined per wave in LDS and leave as one atomic set per (wave, group).
//
// COOP: the reduction also APPLIES the GroupNorm (+ SiLU) whose statistics it has just produced (IGemmParams::pgn_*: ResBlock conv1 ->
// out_layers' GroupNorm -> SiLU -> conv2 at the levels where conv1 is split, openaimodel.py:225-231) -- the GroupNorm-apply launch behind it
// disappears.  The statistics are global, so the workgroups meet at a grid barrier between producing and using them: every thread waits for
// its statistics atomics to be acknowledged, one ticket per workgroup on a counter, a bounded spin (coherent loads) until all tickets are in;
// the launcher only takes this path when every workgroup of the grid is resident at once (<= 256 workgroups of 1024 threads).  Then the (sample, group) totals
// are folded exactly as norm.hip's apply kernel folds them (gn_mean_rstd over the eight slots) and the quad still sitting in registers is
// normalised with gn_apply_elem: the same fp16 bits as the two launches.  A spin that runs out (cannot happen with a resident grid) stores
// NaNs: loud, not a hang.
// (COOP workgroups are 1024 threads: round 5's barrier put the tickets of a grid on ONE address, where device-scope atomics retire
// at ~25 ns each -- 640 workgroups of 256 threads spent 16 us at the barrier, profiles/reduce_gn_coop_r05.txt; round 6: hierarchical, see below)
#ifndef SDMI_REDUCE_GN_XCD_DEFAULT
#define SDMI_REDUCE_GN_XCD_DEFAULT 0
#endif
constexpr int COOP_BAR_INTS = 18 * 32;            // the grid barrier's words (see below)
template <bool COOP>
__global__ void __launch_bounds__(COOP ? 1024 : 256) splitk_reduce_tiled_kernel(IGemmParams p, int nsplit) {
  constexpr int NWV = COOP ? 16 : 4;                              // waves per workgroup (at most)
  __shared__ unsigned long long s_gn[2][NWV][32][GN_WORDS];      // [target][wave][group]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, NTB = blockDim.x;      // (256 threads, or 128: SDMI_REDUCE_BLOCK)
  const bool gn = p.gn_n > 0;
  if (gn) {
    for (int i = tid; i < 2 * NWV * 32 * GN_WORDS; i += NTB) (&s_gn[0][0][0][0])[i] = 0ull;
    __syncthreads();
  }
  const TiledQuad t = tiled_quad(p, blockIdx.x * NTB + tid, lane);
  const int ntiles = ((p.M + p.slab_bm - 1) / p.slab_bm) * ((p.N + p.slab_bn - 1) / p.slab_bn);
  const size_t split_stride = (size_t)ntiles * (size_t)(p.slab_bm * p.slab_bn);
  const float* src = p.splitk_ws + t.off;
  const int HW = p.Hout * p.Wout;
  const int m = min(t.m, p.M - 1), n = min(t.n, p.N - 4);        // (clamped: every load below is in bounds; stores are masked)
  f32x4 biasv = {0.f, 0.f, 0.f, 0.f}, rvv = {0.f, 0.f, 0.f, 0.f}, resv = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) biasv = *(const f32x4*)(p.bias + n);
  if (p.rowvec) rvv = *(const f32x4*)(p.rowvec + (size_t)(m / HW) * p.ld_rowvec + n);
  if (p.residual) resv = *(const f32x4*)(p.residual + (size_t)m * p.ldr + n);
  f32x4 part[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) part[s] = (s < nsplit) ? *(const f32x4*)(src + s * split_stride) : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 a = part[0];
#pragma unroll
  for (int s = 1; s < 16; ++s) a += part[s];       // fixed order; absent splits add +0 (element-wise: the transpose commutes with it)
  f32x4 v = quad_transpose(a, lane & 3);
  v += biasv;
  if (p.rowvec) v += rvv;
  if (p.residual) v += resv;
  if (t.valid) {
    if (p.out_f32 && (!COOP || p.pgn_keep_f32)) SDMI_ST_F32X4(p.out_f32, (size_t)m * p.ldo + n, v);
    if (p.out_f16) SDMI_ST_F16X4(p.out_f16, (size_t)m * p.ldo + n, (f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]}));
    if (p.out_lo) {
      f16x4 lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) lo[j] = (f16)(v[j] - (float)(f16)v[j]);
      *(f16x4*)(p.out_lo + (size_t)m * p.ldo + n) = lo;
    }
  }
  if (gn) {
    if (t.valid) {
      for (int tg = 0; tg < p.gn_n; ++tg) {
        const int c = p.gn_cbase[tg] + n;
        const int g0 = fast_div(c, p.gn_magic[tg]), g1 = fast_div(c + 3, p.gn_magic[tg]);
        const int gsplit = (g0 + 1) * p.gn_cpg[tg] - c;              // first of the 4 channels that belongs to g1
        float a1, a2, c1, c2;
        quad_partials_split(v, gsplit, &a1, &a2, &c1, &c2);
        gn_acc_add(&s_gn[tg][wv][g0][0], a1);
        gn_acc_add(&s_gn[tg][wv][g0][2], a2);
        if (g1 != g0) { gn_acc_add(&s_gn[tg][wv][g1][0], c1); gn_acc_add(&s_gn[tg][wv][g1][2], c2); }
      }
    }
    __syncthreads();
    const int slot = blockIdx.x & (GN_SLOTS - 1);
    for (int e = tid; e < p.gn_n * NWV * 32 * GN_WORDS; e += NTB) {
      const unsigned long long w = (&s_gn[0][0][0][0])[e];
      if (w == 0ull) continue;
      const int word = e % GN_WORDS, g = (e / GN_WORDS) % 32, w4 = (e / (GN_WORDS * 32)) % NWV, tg = e / (GN_WORDS * 32 * NWV);
      // the sample of that wave's 32-row slab: its first row (lane 0 of the wave would own it)
      if (w4 * 64 >= NTB) continue;
      const TiledQuad t0 = tiled_quad(p, blockIdx.x * NTB + w4 * 64, 0);
      if (t0.m >= p.M) continue;
      const int b = t0.m / HW;
      atomicAdd((unsigned long long*)p.gn_acc[tg] + ((size_t)(b * 32 + g) * GN_SLOTS + slot) * GN_STRIDE + word, w);
    }
  }
  if constexpr (COOP) {
    // ---- grid barrier: this workgroup's statistics are in memory, then its ticket; wait for everybody's ----
    __shared__ int s_ok;
    __shared__ float2 s_tab[NWV][4];                      // [wave][group - first group of the wave's 32 columns] {mean, rstd}
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // Round 6: the HIERARCHICAL barrier (MI355X_MICROARCH.md price list, "barrier-xcd" instead of "barrier-counter": round 5 put every ticket of the
    // grid on ONE address and let every workgroup poll it: 8 - 16 us).  Workgroups form eight groups by blockIdx % 8 (the dispatcher's XCD
    // round-robin -- used for speed only; the populations are static, so any placement is correct).  A workgroup arrives on its GROUP's counter
    // (<= 32 arrivals per address, eight addresses in parallel); the last arriver of a group arrives on the TOP counter; the last of those resets
    // the counters and stores the call-local epoch into the eight RELEASE words.  Nobody polls a counter: every workgroup polls its group's
    // release word (<= 32 relaxed pollers per line, no atomics on it).  The payload (the statistics) is written by device-scope atomics and read
    // back by agent-scope loads, so the barrier needs no release / acquire fence -- arrival counting is all of it.
    // Words (128-byte lines, the tail of the tile-counter region, zeroed by the per-call memset): line g < 8: group counter, line 8: top counter,
    // line 9: epoch of the previous barrier in this call, line 10 + g: release word of group g.
    int* const blk = p.splitk_cnt + p.splitk_cnt_ints - COOP_BAR_INTS;
    const int nblk = (int)gridDim.x, ngrp = min(nblk, 8);
    if (tid == 0) {
      const int g = (int)blockIdx.x & 7, pop = (nblk - g + 7) >> 3;
      const int ep = __hip_atomic_load(blk + 9 * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;      // (written by the previous launch's last leader)
      if (__hip_atomic_fetch_add(blk + g * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == pop - 1) {
        if (__hip_atomic_fetch_add(blk + 8 * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngrp - 1) {
          for (int j = 0; j <= 8; ++j) __hip_atomic_store(blk + j * 32, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(blk + 9 * 32, ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int j = 0; j < ngrp; ++j) __hip_atomic_store(blk + (10 + j) * 32, ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      int budget = 1 << 16, seen = 0;
      while ((seen = __hip_atomic_load(blk + (10 + g) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != ep && --budget > 0) __builtin_amdgcn_s_sleep(2);
      s_ok = seen == ep;
    }
    __syncthreads();
    // ---- {mean, rstd} of the (at most four) groups this wave's 32 columns touch: lanes 8 k .. 8 k + 7 fold the eight slots of group gfirst + k
    // (norm.hip gn_fold), coherent loads: the totals were written by every XCD ----
    const int cpg = p.N >> 5;
    const TiledQuad tw = tiled_quad(p, blockIdx.x * NTB + wv * 64, 0);   // the wave's first quad: first row, first column of its 32 x 32 block
    const int bw = min(tw.m, p.M - 1) / HW;
    const int gfirst = fast_div(min(tw.n, p.N - 1), p.gn_magic[0]);
    if (lane < 32) {
      const int g = min(gfirst + (lane >> 3), 31), sub = lane & 7;
      const long long* srcw = (const long long*)p.gn_acc[0] + ((size_t)(bw * 32 + g) * GN_SLOTS + sub) * GN_STRIDE;
      long long a0 = __hip_atomic_load(srcw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a1 = __hip_atomic_load(srcw + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long long q0 = __hip_atomic_load(srcw + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), q1 = __hip_atomic_load(srcw + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int o = GN_SLOTS / 2; o >= 1; o >>= 1) {
        a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); q0 += __shfl_xor(q0, o); q1 += __shfl_xor(q1, o);
      }
      if (sub == 0) {
        float mu, rs;
        gn_mean_rstd(a0, a1, q0, q1, (double)cpg * (double)HW, p.pgn_eps, &mu, &rs);
        s_tab[wv][lane >> 3] = float2{mu, rs};
      }
    }
    __syncthreads();
    if (t.valid) {
      // (cpg % 4 == 0: the launcher checked -- a quad lies inside one group)
      const float2 mr = s_tab[wv][min(fast_div(n, p.gn_magic[0]) - gfirst, 3)];
      const f32x4 ga = *(const f32x4*)(p.pgn_gamma + n), be = *(const f32x4*)(p.pgn_beta + n);
      f16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = s_ok ? (f16)gn_apply_elem(v[j], mr.x, mr.y, ga[j], be[j], p.pgn_silu) : (f16)__builtin_nanf("");
      SDMI_ST_F16X4(p.pgn_out, (size_t)m * p.N + n, o);
    }
  }
}

// the same reduction for the per-head scatter epilogue: quad (m, n..n+3) lies inside one head (dh % 4 == 0)
__global__ void __launch_bounds__(256) splitk_reduce_tiled_heads_kernel(IGemmParams p, int nsplit) {
  const int tid = threadIdx.x, lane = tid & 63;
  const TiledQuad t = tiled_quad(p, blockIdx.x * 256 + tid, lane);
  const int ntiles = ((p.M + p.slab_bm - 1) / p.slab_bm) * ((p.N + p.slab_bn - 1) / p.slab_bn);
  const size_t split_stride = (size_t)ntiles * (size_t)(p.slab_bm * p.slab_bn);
  const float* src = p.splitk_ws + t.off;
  f32x4 part[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) part[s] = (s < nsplit) ? *(const f32x4*)(src + s * split_stride) : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 a = part[0];
#pragma unroll
  for (int s = 1; s < 16; ++s) a += part[s];
  f32x4 v = quad_transpose(a, lane & 3);
  if (!t.valid) return;
  const int m = t.m, n = t.n;
  if (p.bias) v += *(const f32x4*)(p.bias + n);
  const int seg = n / p.segC;
  const int c = n - seg * p.segC;
  const int head = c / p.dh;
  const int dd = c - head * p.dh;
  const int b = m / p.ntok;
  const int tok = m - b * p.ntok;
  const size_t bh = (size_t)b * p.heads + head;
  f16* dst = p.seg_dst[seg];
  if (p.seg_kind[seg] == 0) {
    *(f16x4*)(dst + (bh * p.ntok + tok) * p.dh + dd) = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[(bh * p.dh + dd + j) * p.ntok_pad + tok] = (f16)v[j];
  }
}


}  // namespace

int launch_splitk_reduce(const IGemmParams& p, int nsplit, hipStream_t stream) {
  SDMI_CHECK(nsplit >= 1 && nsplit <= 16 && p.N % 4 == 0 && p.splitk_ws, "splitk_reduce: bad arguments");
#ifdef SDMI_EXPERIMENTS
  if (const int maxq = reduce_gn_maxq(p, nsplit)) {       // the consuming GroupNorm (+ SiLU) inside the reduction: see the kernel
    const unsigned long long magic_qpr = div_magic(p.N / 128);
    const double mn = (double)p.M * p.N;
    ProfScope psg("splitk_reduce_gn", 0.0, mn * 4.0 * nsplit + mn * 2.0 + (p.residual ? mn * 4.0 : 0.0) + (p.pgn_keep_f32 ? mn * 4.0 : 0.0), stream);
    const dim3 grid(32, (unsigned)p.B), block(1024);
    if (p.slab_tiled) {
      if (maxq == 1) SDMI_LAUNCH((splitk_reduce_gn_kernel<1, true>), grid, block, 0, stream, p, nsplit, magic_qpr);
      else if (maxq == 3) SDMI_LAUNCH((splitk_reduce_gn_kernel<3, true>), grid, block, 0, stream, p, nsplit, magic_qpr);
      else SDMI_LAUNCH((splitk_reduce_gn_kernel<5, true>), grid, block, 0, stream, p, nsplit, magic_qpr);
    } else {
      if (maxq == 1) SDMI_LAUNCH((splitk_reduce_gn_kernel<1, false>), grid, block, 0, stream, p, nsplit, magic_qpr);
      else if (maxq == 3) SDMI_LAUNCH((splitk_reduce_gn_kernel<3, false>), grid, block, 0, stream, p, nsplit, magic_qpr);
      else SDMI_LAUNCH((splitk_reduce_gn_kernel<5, false>), grid, block, 0, stream, p, nsplit, magic_qpr);
    }
    SDMI_HIP_OK(hipGetLastError());
    psg.end();
    if (p.pgn_applied) *p.pgn_applied = 1;
    if (range_check_enabled() && range_scan("GroupNorm fp16 output (split-K reduction)", p.pgn_out, (int64_t)p.M * p.N, stream)) return -1;
    return 0;
  }
#endif
  if (p.slab_tiled) {                  // register-order slabs: one thread per slab quad, whole tiles (padding rows / columns masked)
    const int64_t quads = (int64_t)cdiv(p.M, p.slab_bm) * cdiv(p.N, p.slab_bn) * (p.slab_bm * p.slab_bn / 4);
    SDMI_CHECK(quads % 256 == 0 && quads / 256 < (1ll << 31) && p.slab_wm > 0 && p.slab_wn > 0, "tiled split-K slabs: bad geometry");
    const double mn = (double)p.M * p.N;
    if (p.mode == EPI_HEADS) {
      ProfScope psh("splitk_reduce", 0.0, mn * (4.0 * nsplit + 2.0), stream);
      SDMI_LAUNCH(splitk_reduce_tiled_heads_kernel, dim3((unsigned)(quads / 256)), dim3(256), 0, stream, p, nsplit);
      SDMI_HIP_OK(hipGetLastError());
      return 0;
    }
    if (p.gn_n > 0) SDMI_CHECK((p.Hout * p.Wout) % 32 == 0, "GroupNorm statistics need Hout*Wout % 32 == 0");
    // the consuming GroupNorm (+ SiLU) applied by the reduction itself, behind a grid barrier (splitk_reduce_tiled_kernel<true>): where
    // that GroupNorm's statistics come from this very reduction, channels-per-group % 4 == 0 and the whole grid is resident at once
    // (one workgroup of 1024 threads per CU at most: <= 256 workgroups).  Round 5: experiments build, one ticket counter (measured slower).
    // Round 6: product build, run-time switch SDMI_REDUCE_GN_XCD (read per launch: the tests flip it between two forwards).
    const bool coop = env_int("SDMI_REDUCE_GN_XCD", SDMI_REDUCE_GN_XCD_DEFAULT) && p.pgn_out && p.pgn_gamma && p.pgn_beta && p.mode == EPI_PLAIN && p.gn_n == 1 &&
                      p.gn_cbase[0] == 0 && p.gn_cpg[0] == p.N / 32 && p.N % 128 == 0 && p.N / 32 >= 16 && p.M == p.B * p.Hout * p.Wout &&
                      (p.Hout * p.Wout) % 32 == 0 && !p.out_f16 && !p.out_lo && !p.ln_out && p.splitk_cnt && p.splitk_cnt_ints >= COOP_BAR_INTS &&
                      quads % 1024 == 0 && quads / 1024 <= 256 && (!p.pgn_keep_f32 || (p.out_f32 && p.ldo % 4 == 0));
    ProfScope pst(coop ? "splitk_reduce_gn" : "splitk_reduce", 0.0, coop ? mn * (4.0 * nsplit + 2.0 + (p.pgn_keep_f32 ? 4.0 : 0.0)) : mn * 4.0 * (nsplit + 1), stream);
    // threads per block: 256, or 128 where that still leaves fewer than two blocks per CU (SDMI_REDUCE_BLOCK: 0 auto, 128 / 256 forced; A/B)
    const int rb_env = env_int("SDMI_REDUCE_BLOCK", 256);
    const int rb = (rb_env == 128 || (rb_env == 0 && quads / 256 < 512)) ? 128 : 256;
    if (coop) {
      SDMI_LAUNCH(splitk_reduce_tiled_kernel<true>, dim3((unsigned)(quads / 1024)), dim3(1024), 0, stream, p, nsplit);
      SDMI_HIP_OK(hipGetLastError());
      pst.end();
      if (p.pgn_applied) *p.pgn_applied = 1;
      if (range_check_enabled() && range_scan("GroupNorm fp16 output (split-K reduction)", p.pgn_out, (int64_t)p.M * p.N, stream)) return -1;
      return 0;
    }
    SDMI_LAUNCH(splitk_reduce_tiled_kernel<false>, dim3((unsigned)(quads / rb)), dim3(rb), 0, stream, p, nsplit);
    SDMI_HIP_OK(hipGetLastError());
    pst.end();
    if (p.ln_out) return launch_layernorm(p.out_f32, p.ln_gamma, p.ln_beta, p.ln_out, p.M, p.N, p.ln_eps, stream);
    return 0;
  }
  if (p.mode == EPI_HEADS) {
    const int64_t total_h = (int64_t)p.M * (p.N / 4);
    ProfScope psh("splitk_reduce", 0.0, (double)p.M * p.N * (4.0 * nsplit + 2.0), stream);
    SDMI_LAUNCH(splitk_reduce_heads_kernel, dim3((unsigned)((total_h + 255) / 256)), dim3(256), 0, stream, p, nsplit);
    SDMI_HIP_OK(hipGetLastError());
    return 0;
  }
  // rows per block: 32 (one row per thread) up to 256, doubling while the grid keeps >= 1024 blocks (the kernel is a
  // latency-bound stream of nsplit 16-byte loads per thread: it wants every CU busy); with GroupNorm statistics it must
  // also divide the sample's row count, so a strip never straddles two samples
  const int hw = p.Hout * p.Wout;
  if (p.gn_n > 0) SDMI_CHECK(hw % 32 == 0, "GroupNorm statistics need Hout*Wout % 32 == 0");
  const int rpb = reduce_rows_per_block(p);
  ProfScope ps2("splitk_reduce", 0.0, (double)p.M * p.N * 4.0 * (nsplit + 1), stream);
  SDMI_LAUNCH(splitk_reduce_kernel, dim3((unsigned)cdiv(p.N / 4, 8), (unsigned)cdiv(p.M, rpb)), dim3(256), 0, stream, p, nsplit,
                     rpb);
  SDMI_HIP_OK(hipGetLastError());
  ps2.end();
  if (p.ln_out) return launch_layernorm(p.out_f32, p.ln_gamma, p.ln_beta, p.ln_out, p.M, p.N, p.ln_eps, stream);
  return 0;
}

// ---- tile table ------------------------------------------------------------------------------------------------------
// id: BM x BN, waves (M x N), per-wave MFMA tiles TM x TN, LDS-DMA stages.  TN even is required by the GEGLU epilogue.
struct TileCfg { int bm, bn, wm, wn, ns; };
static const TileCfg kTiles[SDMI_NUM_TILES] = {
    {128, 128, 2, 2, 2},   //  0  2x2 tiles per wave, 64 KB  (2 blocks / CU)
    {128, 64, 2, 2, 2},    //  1  2x1
    {64, 64, 2, 2, 2},     //  2  1x1
    {256, 128, 4, 2, 2},   //  3  8 waves, 2x2, 96 KB
    {128, 64, 2, 2, 3},    //  4  2x1, 72 KB
    {64, 64, 2, 2, 3},     //  5  1x1, 48 KB (3 blocks / CU)
    {256, 128, 4, 2, 3},   //  6  8 waves, 2x2, 144 KB
    {128, 128, 2, 2, 3},   //  7  2x2, 96 KB
    {64, 128, 2, 2, 3},    //  8  1x2, 72 KB
    {128, 128, 4, 2, 3},   //  9  8 waves, 1x2, 96 KB
    {64, 64, 2, 2, 4},     // 10  1x1, 64 KB
    {128, 256, 2, 4, 2},   // 11  8 waves, 2x2, 96 KB
    {64, 256, 1, 4, 3},    // 12  4 waves, 2x2, 120 KB (small M, wide N: one A tile shared by the 4 waves)
    {256, 64, 4, 1, 3},    // 13  4 waves, 2x2, 120 KB (large M, N = 5 x 64)
    // halo-staged 3x3 convolution (conv3halo_kernel): tiles of whole image rows
    {256, 64, 4, 2, 5},    // 14  8 waves, 2x1 per wave, 152 KB
    {256, 128, 4, 2, 3},   // 15  8 waves, 2x2, 160 KB
    {128, 64, 2, 2, 8},    // 16  4 waves, 2x1, 136 KB
    {128, 128, 2, 2, 5},   // 17  4 waves, 2x2, 152 KB
    // generic again: deep LDS-DMA rings for the weight-streaming shapes (small M, K in the thousands: every weight tile is a
    // first touch of the XCD's L2, so the ring has to cover HBM latency, ~1 us = 5+ k-tiles of MFMA work)
    {64, 64, 2, 2, 8},     // 18  1x1, 128 KB
    {64, 128, 2, 2, 6},    // 19  1x2, 144 KB
    {128, 64, 2, 2, 6},    // 20  2x1, 144 KB
    {128, 128, 4, 2, 4},   // 21  8 waves, 1x2, 128 KB
    // five waves side by side (igemm5.hip): N = 160 k, M = 8192 -> exactly one workgroup per CU
    {64, 160, 1, 5, 5},    // 22  5 waves, 2x1 per wave, 140 KB
};
static inline bool tile_is5(int t) { return t == 22; }
#ifdef SDMI_EXPERIMENTS
constexpr bool kExperiments = true;
#else
constexpr bool kExperiments = false;      // product build: tile 22, the GroupNorm-folding kernels and the GroupNorm-applying split-K reduction are not compiled in
#endif
static inline bool tile_is_halo(int t) { return t >= 14 && t <= 17; }
static inline bool tile_tn_even(int t) { return (kTiles[t].bn / kTiles[t].wn / 32) % 2 == 0; }

static int launch_tile(int tile, const IGemmParams& p, bool dma, int splitk, hipStream_t stream) {
  if (p.split16) return launch_split16_tile(tile, p, splitk, stream);
  switch (tile) {            // tile ids: see include/sdmi.h (sdmi_igemm_desc.tile); the generic tiles are instantiated in igemm_t0 / t1 / t2.hip
    case 0: case 1: case 2: case 3: case 4: case 5: return launch_generic_tile_g0(tile, p, dma, splitk, stream);
    case 6: case 7: case 8: case 9: case 10: return launch_generic_tile_g1(tile, p, dma, splitk, stream);
    case 11: case 12: case 13: case 18: case 19: case 20: case 21: return launch_generic_tile_g2(tile, p, dma, splitk, stream);
    case 14: case 15: case 16: case 17:
      return p.xf0 ? launch_halo_gn_tile(tile, p, splitk, stream) : launch_halo_tile(tile, p, splitk, stream);
    case 22: return launch_igemm5_tile(tile, p, splitk, stream);
    default: return fail("unknown igemm tile id");
  }
}

// ---- per-shape tuning table ---------------------------------------------------------------------------------------
// Tile shape and split-K of every auto-configured GEMM come from a table keyed by the GEMM's shape, measured ON the
// MI355X in situ: during a collection run (sdmi_tune_begin / _round / _end, tools/tune.py) every launch site of a real
// UNet / first-stage / text-encoder call runs candidate (round mod #candidates) of its shape, timed with HIP events on
// the launch stream -- so each candidate sees the cache state of the real call (weights cold in HBM, activations warm
// from the producing kernel), which a stand-alone micro-benchmark of one shape does not.  The table is a text file next
// to libsdmi.so (stable-diffusion_amd/tune_gfx950.txt, committed): the choice is fixed, so results stay bit-reproducible.
struct TuneKey {
  int M, N, K, ksize, stride, up, mode, splitk_req;
  bool operator<(const TuneKey& o) const {
    return std::tie(M, N, K, ksize, stride, up, mode, splitk_req) <
           std::tie(o.M, o.N, o.K, o.ksize, o.stride, o.up, o.mode, o.splitk_req);
  }
};
struct TuneChoice { int tile, splitk; double us; };
struct TuneRec { TuneKey key; int cand; hipEvent_t e0, e1; };

class Tuner {
 public:
  std::mutex mu;
  std::map<TuneKey, TuneChoice> table;
  bool loaded = false, collecting = false;
  int round = 0;
  std::vector<TuneRec> recs;
  std::map<TuneKey, std::vector<std::pair<TuneChoice, std::vector<float>>>> stats;   // per candidate: the samples (us)
  std::vector<hipEvent_t> pool;

  static std::string default_path() {
    if (const char* e = getenv("SDMI_TUNE_FILE")) return e;
    Dl_info info;
    if (dladdr((const void*)&Tuner::default_path, &info) && info.dli_fname) {
      std::string so = info.dli_fname;
      const size_t k = so.find_last_of('/');
      return (k == std::string::npos ? std::string(".") : so.substr(0, k)) + "/tune_gfx950.txt";
    }
    return "tune_gfx950.txt";
  }
  void load_file(const std::string& path) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return;
    char line[256];
    while (fgets(line, sizeof line, f)) {
      if (line[0] == '#') continue;
      TuneKey k; TuneChoice c; c.us = 0;
      if (sscanf(line, "%d %d %d %d %d %d %d %d %d %d %lf", &k.M, &k.N, &k.K, &k.ksize, &k.stride, &k.up, &k.mode,
                 &k.splitk_req, &c.tile, &c.splitk, &c.us) >= 10 && c.tile >= 0 &&
              (c.tile < SDMI_NUM_TILES || (c.tile == SDMI_TILE_TWO_LAUNCH && k.ksize == 13)) && c.splitk >= 1 &&
          c.splitk <= 16)
        table[k] = c;
    }
    fclose(f);
  }
  void ensure_loaded() {
    if (loaded) return;
    loaded = true;
    if (env_int("SDMI_TUNE_DISABLE", 0)) return;
    load_file(default_path());
  }
  hipEvent_t ev() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
};
static Tuner g_tuner;

// candidate (tile, split-K) pairs of a shape, in a fixed order (the collection run indexes them by round)
static std::vector<TuneChoice> tune_candidates(const IGemmParams& p, bool can_split) {
  std::vector<TuneChoice> out;
  const int nkt = p.K / BK;
  static const int splits[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16};     // 5 / 10: halo tiles only (20 / 40 channel chunks)
  for (int t = 0; t < SDMI_NUM_TILES; ++t) {
    const TileCfg& c = kTiles[t];
    if (tile_is_halo(t) && !halo_supported(p, c.bm)) continue;
    if (p.xf0 && (!tile_is_halo(t) || !halo_gn_supported(p, c.bm))) continue;     // GroupNorm-folding conv: halo tiles only
    if (p.split16 && !split16_tile_supported(t)) continue;                         // split-fp16 GEMM: its own instantiations
    if (tile_is5(t) && (!kExperiments || p.up || p.split16 || p.xf0 || p.N % c.bn != 0)) continue;   // five-wave tile: whole 160-column tiles, plain gathers
    // never chosen by any of the round-2 collection runs (profiles/tune_candidates_r02.txt): the 2-stage twins of the
    // 3-stage tiles, 128x128 / 256x128 with 2 stages, and the 64x256 / 256x64 4-wave tiles -- fewer candidates = more
    // samples per candidate
    if (!p.split16 && ((c.ns == 2 && t != 11 && !tile_is_halo(t)) || t == 12 || t == 13)) continue;
    if (p.mode == EPI_GEGLU && !tile_tn_even(t)) continue;
    if (p.lnp_out && (p.M % c.bm || p.N % c.bn || (p.Hout * p.Wout) % c.bm)) continue;   // row statistics: full tiles inside one sample
    const long blocks = (long)cdiv(p.M, c.bm) * cdiv(p.N, c.bn);
    if ((long)c.bm > 2L * p.M && c.bm > 64) continue;                  // tile mostly padding
    if ((long)c.bn > 2L * p.N && c.bn > 64) continue;
    for (int sk : splits) {
      if ((sk == 5 || sk == 10) && !tile_is_halo(t)) continue;
      if (sk > 1) {
        if (p.splitk != 0 || !can_split) break;                        // caller pinned the split
        if (nkt / sk < 4) break;                                       // >= 4 k-tiles per split
        if (tile_is_halo(t) && (nkt / 9) % sk != 0) continue;          // halo tiles split at 64-channel chunk granularity
        if (blocks * (sk / 2 + 1) > 1536) break;                       // already plenty of blocks one step earlier
        if (splitk_ws_need(p, c.bm, c.bn, sk) > p.splitk_ws_floats) break;
      }
      if (blocks * sk < 48 && sk < 16 && nkt / (sk * 2) >= 4 && can_split && p.splitk == 0) continue;   // hopelessly few blocks
      out.push_back({t, p.splitk > 1 ? p.splitk : sk, 0.0});
      if (p.splitk != 0) break;
    }
  }
  if (p.xf0 && p.gn_scratch) out.push_back({SDMI_TILE_TWO_LAUNCH, 1, 0.0});      // GroupNorm-apply launch + LDS-DMA conv (its own table entry)
  if (out.empty()) out.push_back({5, p.splitk > 0 ? p.splitk : 1, 0.0});
  return out;
}

int tune_begin() {
  std::lock_guard<std::mutex> lk(g_tuner.mu);
  g_tuner.ensure_loaded();
  g_tuner.collecting = true; g_tuner.round = 0;
  g_tuner.recs.clear(); g_tuner.stats.clear();
  return 0;
}
int tune_round(int r) {
  std::lock_guard<std::mutex> lk(g_tuner.mu);
  g_tuner.round = r;
  return 0;
}
// fold the finished event pairs into the statistics (call with the device idle, e.g. after a stream synchronize)
static int tune_drain() {
  for (auto& r : g_tuner.recs) {
    float ms = 0.f;
    SDMI_HIP_OK(hipEventSynchronize(r.e1));
    SDMI_HIP_OK(hipEventElapsedTime(&ms, r.e0, r.e1));
    auto& v = g_tuner.stats[r.key];
    if ((int)v.size() > r.cand) v[r.cand].second.push_back(ms * 1e3f);
    g_tuner.pool.push_back(r.e0); g_tuner.pool.push_back(r.e1);
  }
  g_tuner.recs.clear();
  return 0;
}
bool tune_collecting() { return g_tuner.collecting; }
static uint64_t g_tune_generation = 0;       // bumped whenever the table's contents may have changed (launch tapes were planned with it)
uint64_t tune_generation() { return g_tune_generation; }
int tune_end(const char* path, int* n_keys) {
  std::lock_guard<std::mutex> lk(g_tuner.mu);
  if (!g_tuner.collecting) return fail("sdmi_tune_end without sdmi_tune_begin");
  if (tune_drain()) return -1;
  g_tuner.collecting = false;
  // score of a candidate = median of its samples (in-situ samples carry launch-order and cache-state outliers; the mean of
  // 2-3 of them flipped choices from run to run)
  auto score = [](std::vector<float> v) -> double {
    if (v.empty()) return 1e30;
    std::sort(v.begin(), v.end());
    return v.size() % 2 ? v[v.size() / 2] : 0.5 * (v[v.size() / 2 - 1] + v[v.size() / 2]);
  };
  for (auto& kv : g_tuner.stats) {
    double best = 1e30; const TuneChoice* bc = nullptr;
    for (auto& c : kv.second) {
      const double sc = score(c.second);
      if (sc < best) { best = sc; bc = &c.first; }
    }
    if (bc) g_tuner.table[kv.first] = {bc->tile, bc->splitk, best};
  }
  if (n_keys) *n_keys = (int)g_tuner.stats.size();
  ++g_tune_generation;
  const std::string out = (path && *path) ? std::string(path) : Tuner::default_path();
  FILE* f = fopen(out.c_str(), "w");
  if (!f) return fail("cannot write the tuning table to " + out);
  fprintf(f, "# libsdmi igemm tuning table (gfx950), measured in situ by tools/tune.py -- M N K ksize stride up mode splitk_req tile splitk us\n");
  for (auto& kv : g_tuner.table) {
    const TuneKey& k = kv.first;
    fprintf(f, "%d %d %d %d %d %d %d %d %d %d %.2f\n", k.M, k.N, k.K, k.ksize, k.stride, k.up, k.mode, k.splitk_req, kv.second.tile,
            kv.second.splitk, kv.second.us);
  }
  fclose(f);
  return 0;
}
// per-candidate timings of the last collection as text lines (for profiles/): key | tile splitk us n
int tune_dump(std::string* out) {
  std::lock_guard<std::mutex> lk(g_tuner.mu);
  char buf[256];
  for (auto& kv : g_tuner.stats) {
    const TuneKey& k = kv.first;
    for (auto& c : kv.second) {
      if (c.second.empty()) continue;
      std::vector<float> v = c.second;
      std::sort(v.begin(), v.end());
      snprintf(buf, sizeof buf, "%d %d %d %d %d %d %d %d | %d %d %.2f %.2f %d\n", k.M, k.N, k.K, k.ksize, k.stride, k.up, k.mode,
               k.splitk_req, c.first.tile, c.first.splitk, (double)v[v.size() / 2], (double)v[0], (int)v.size());
      *out += buf;
    }
  }
  return 0;
}

int launch_igemm(const IGemmParams& p, const IGemmTune& tune, hipStream_t stream) {
  SDMI_CHECK(p.M > 0 && p.N > 0 && p.K > 0, "empty GEMM");
  SDMI_CHECK(p.ksize == 1 || p.ksize == 3, "ksize must be 1 or 3");
  const int Cin = p.c0 + p.c1 + p.c2;
  SDMI_CHECK(p.K == p.ksize * p.ksize * Cin, "K != ksize^2 * (c0 + c1 + c2)");
  SDMI_CHECK(Cin % BK == 0 && p.c0 % BK == 0 && p.c1 % BK == 0, "channel counts must be multiples of 64");
  const bool gn_fold = p.xf0 != nullptr;       // GroupNorm + SiLU folded into the conv's staging: the A operand is the fp32 stream
  if (gn_fold)
    SDMI_CHECK(p.ksize == 3 && p.stride == 1 && p.pad == 1 && !p.up && p.mode == EPI_PLAIN && p.c2 == 0 && p.gn_in_acc &&
                   p.gn_in_gamma && p.gn_in_beta && (p.c1 == 0 || p.xf1),
               "GroupNorm-folding conv: 3x3 stride 1 pad 1, plain epilogue, statistics + gamma + beta");
  SDMI_CHECK(gn_fold || (p.lda0 % 8 == 0 && (p.a1 == nullptr || p.lda1 % 8 == 0)), "A row pitch must be a multiple of 8 halves");
  SDMI_CHECK(p.zero_page != nullptr, "zero page missing");
  SDMI_CHECK(p.M == p.B * p.Hout * p.Wout, "M != B * Hout * Wout");
  SDMI_CHECK(gn_fold || p.c1 == 0 || p.a1 != nullptr, "second A source missing");
  SDMI_CHECK(p.c2 == 0 || (p.a2 != nullptr && p.lda2 % 8 == 0), "third A source missing");
  SDMI_CHECK(gn_fold || ((p.c1 == 0 || p.lda1 == p.lda0) && (p.c2 == 0 || p.lda2 == p.lda0)), "all A sources must share one row pitch");
  SDMI_CHECK(!p.up || (p.ksize == 3 && p.stride == 1), "upsample folding needs a 3x3 stride-1 conv");
  if (p.mode == EPI_GEGLU) SDMI_CHECK(p.N % 64 == 0 && p.out_f16 != nullptr, "GEGLU needs N % 64 == 0 and an fp16 output");
  if (p.ln_out)
    SDMI_CHECK(p.mode == EPI_PLAIN && p.out_f32 && p.ldo == p.N && p.N % 4 == 0 && p.N <= 2560 && p.ln_gamma && p.ln_beta,
               "LayerNorm post-op needs plain mode, an fp32 output with ldo == N <= 2560, gamma and beta");
  if (p.out_lo) SDMI_CHECK(p.mode == EPI_PLAIN && p.ldo % 4 == 0, "out_lo needs plain mode");
  if (p.lnp_out || p.f16_scale)        // producer of a LayerNorm that its consumer folds (IGemmParams::lnp_out)
    SDMI_CHECK(p.mode == EPI_PLAIN && p.out_f32 && p.out_f16 && !p.out_lo && !p.ln_out && p.N % 64 == 0 && p.M % 64 == 0 &&
                   (p.Hout * p.Wout) % 64 == 0 && p.splitk == 1 && epi_vec_ok(p),
               "LayerNorm row statistics: plain mode, fp32 + fp16 outputs, no split-K, M / N / rows per sample multiples of 64, 16-byte aligned rows");
  if (p.lnf_part)                      // consumer that folds the LayerNorm of its A rows (IGemmParams::lnf_*)
    SDMI_CHECK(p.ksize == 1 && p.c1 == 0 && p.c2 == 0 && !p.split16 && p.lnf_npart * 32 == p.K && p.lnf_cs && p.lnf_d && !p.bias &&
                   p.splitk == 1 && p.lnf_npart <= 40,
               "LayerNorm-folding GEMM: dense, one source, K = 32 * lnf_npart <= 1280, column sums + offsets, no bias, no split-K");
  if (p.gn_n) {
    SDMI_CHECK(p.mode == EPI_PLAIN && p.gn_n <= 2 && (p.Hout * p.Wout) % 32 == 0, "GroupNorm statistics need plain mode and Hout*Wout % 32 == 0");
    SDMI_CHECK(p.N % 4 == 0, "GroupNorm statistics: N % 4 == 0");
    for (int t = 0; t < p.gn_n; ++t)
      SDMI_CHECK(p.gn_acc[t] && p.gn_cpg[t] >= 2 && (p.gn_cbase[t] + p.N + p.gn_cpg[t] - 1) / p.gn_cpg[t] <= 32, "bad GroupNorm statistics target");
  }
  if (p.mode == EPI_HEADS) SDMI_CHECK(p.segC > 0 && p.dh > 0 && p.N % p.segC == 0 && p.N / p.segC <= 3, "bad head scatter");

  SDMI_CHECK((gn_fold || (int64_t)p.B * p.Hin * p.Win * p.lda0 * 2 + (int64_t)(p.Win + 1) * p.lda0 * 2 < ((int64_t)1 << 31) - 65536) &&
                 (int64_t)p.N * p.K * 2 < ((int64_t)1 << 31) - 65536,
             "tensor too large for 31-bit byte offsets (buffer addressing)");
  static const int env_dma = env_int("SDMI_IGEMM_DMA", 1);
  static const int env_tile = env_int("SDMI_IGEMM_TILE", -1);
  const bool dma = (tune.dma >= 0 ? tune.dma : env_dma) != 0;
  int tile = tune.tile >= 0 ? tune.tile : env_tile;
  SDMI_CHECK(tile < SDMI_NUM_TILES, "unknown igemm tile id");
  int splitk = p.splitk;
  const int nkt = p.K / BK;
  static const int env_split = env_int("SDMI_SPLITK", -1);     // 1 disables split-K everywhere
  const bool can_split_plain = p.mode == EPI_PLAIN && p.splitk_ws && p.N % 4 == 0 && p.ldo % 4 == 0 &&
                               (p.residual == nullptr || p.ldr % 4 == 0);
  const bool can_split_heads = p.mode == EPI_HEADS && p.splitk_ws && p.dh % 4 == 0 && p.segC % 4 == 0;
  const bool can_split = can_split_plain || can_split_heads;
  if (env_split >= 0 && splitk == 0) splitk = env_split;

  // ---- (tile, split-K): explicit request > tuning table / collection run > heuristic ------------------------------
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // (the GroupNorm-folding conv is its own family of kernels: keyed apart with ksize 13)
  // (... and so is the split-fp16 dense GEMM: ksize 11)
  TuneKey tkey{p.M, p.N, p.K, gn_fold ? 13 : (p.split16 ? 11 : p.ksize), p.stride, p.up, p.mode, splitk};
  int tcand = -1;
  if (tile < 0) {
    std::lock_guard<std::mutex> lk(g_tuner.mu);
    g_tuner.ensure_loaded();
    // SDMI_TUNE_ONLY_KSIZE=<k> restricts a collection run to one kernel family (the key's ksize: 1, 3, or 13 = the GroupNorm-
    // folding conv); every other launch keeps its table entry, so the run measures the new family inside an unchanged call
    static const int only_ksize = env_int("SDMI_TUNE_ONLY_KSIZE", -1);
    if (g_tuner.collecting && (only_ksize < 0 || tkey.ksize == only_ksize)) {
      IGemmParams q = p; q.splitk = splitk;
      const std::vector<TuneChoice> cands = tune_candidates(q, can_split);
      auto& st = g_tuner.stats[tkey];
      if (st.empty()) for (auto& c : cands) st.push_back({c, {}});
      tcand = g_tuner.round % (int)cands.size();
      tile = cands[tcand].tile; splitk = cands[tcand].splitk;
      ev0 = g_tuner.ev(); ev1 = g_tuner.ev();
    } else {
      // a table entry is taken only if it passes the predicates tune_candidates() generated it under (the key does not carry
      // the whole geometry, and the file may be stale or hand-edited): otherwise the heuristic below decides
      auto it = g_tuner.table.find(tkey);
      if (it != g_tuner.table.end() && it->second.tile == SDMI_TILE_TWO_LAUNCH) {
        if (gn_fold && p.gn_scratch) tile = SDMI_TILE_TWO_LAUNCH;
      } else if (it != g_tuner.table.end()) {
        const int tt = it->second.tile, sk = it->second.splitk;
        const bool halo = tile_is_halo(tt);
        const bool ws_ok = sk == 1 || (can_split && splitk_ws_need(p, kTiles[tt].bm, kTiles[tt].bn, sk) <= p.splitk_ws_floats);
        const bool split_ok = splitk != 0 ? (sk == splitk && ws_ok)                      // the caller pinned the split
                                          : (sk == 1 || (ws_ok && nkt / sk >= 4 && (halo || (sk != 5 && sk != 10)) &&
                                                         (!halo || (nkt / 9) % sk == 0)));
        if (split_ok && (p.mode != EPI_GEGLU || tile_tn_even(tt)) && (!halo || halo_supported(p, kTiles[tt].bm)) &&
            (!tile_is5(tt) || (kExperiments && !p.up && !p.split16 && p.N % kTiles[tt].bn == 0)) &&
            (!gn_fold || (halo && halo_gn_supported(p, kTiles[tt].bm))) && (!p.split16 || split16_tile_supported(tt))) {
          tile = tt; splitk = sk;
        }
      }
    }
  }
  if (p.mode == EPI_GEGLU && tile >= 0 && !tile_tn_even(tile)) tile = 0;   // GEGLU pairs 32-col tiles inside a wave
  static const int force_two = env_int("SDMI_GN_FORCE_TWO", 0);         // A/B: 1 = every folded conv as two launches, -1 = none
  if (gn_fold && p.gn_scratch && force_two > 0 && tcand < 0) tile = SDMI_TILE_TWO_LAUNCH;
  if (gn_fold && force_two < 0 && tile == SDMI_TILE_TWO_LAUNCH && tcand < 0) tile = -1;
  if (gn_fold && tile < 0 && p.gn_scratch && force_two >= 0 && (p.M < 512 || p.c1 > 0)) {
    // no table entry: the round-3 measurements have the two launches ahead at the 8x8 level (the normalisation repeats in every
    // one of 20 N-tiles) and on the skip-concat inputs of the output blocks, the folding kernel elsewhere
    tile = SDMI_TILE_TWO_LAUNCH;
  }
  if (gn_fold && tile == SDMI_TILE_TWO_LAUNCH) {
    SDMI_CHECK(p.gn_scratch != nullptr, "two-launch GroupNorm + conv needs the fp16 scratch");
    if (ev0) SDMI_HIP_OK(hipEventRecord(ev0, stream));
    GroupNormParams g;
    g.x0 = p.xf0; g.x1 = p.xf1; g.c0 = p.c0; g.c1 = p.c1; g.B = p.B; g.HW = p.Hout * p.Wout;
    g.gamma = p.gn_in_gamma; g.beta = p.gn_in_beta; g.eps = p.gn_in_eps; g.silu = p.gn_in_silu;
    g.skip_stats = 1; g.acc = (long long*)p.gn_in_acc;                   // (complete: see IGemmParams::gn_in_acc)
    g.out_f16 = p.gn_scratch; g.raw_f16 = p.raw_hi; g.raw_lo = p.raw_lo;
    if (launch_groupnorm(g, stream)) return -1;
    IGemmParams q = p;
    q.xf0 = q.xf1 = nullptr; q.gn_in_acc = nullptr; q.raw_hi = q.raw_lo = nullptr; q.gn_scratch = nullptr;
    q.a0 = p.gn_scratch; q.c0 = Cin; q.c1 = 0; q.lda0 = Cin;
    const int rc2 = launch_igemm(q, IGemmTune(), stream);
    if (ev0) {
      SDMI_HIP_OK(hipEventRecord(ev1, stream));
      std::lock_guard<std::mutex> lk(g_tuner.mu);
      g_tuner.recs.push_back({tkey, tcand, ev0, ev1});
    }
    return rc2;
  }
  if (gn_fold) {
    // halo tiles only.  Without a table entry: the 256 x 64 tile (whole images of the 16x16 / 8x8 levels fit it too), the
    // 128 x 64 one for 128 rows; split at 64-channel-chunk granularity until ~200 workgroups exist.
    static const int pref[4] = {14, 16, 15, 17};
    if (tile >= 0) SDMI_CHECK(tile_is_halo(tile) && halo_gn_supported(p, kTiles[tile].bm), "GroupNorm-folding conv: unsupported tile for this shape");
    for (int i = 0; i < 4 && tile < 0; ++i)
      if (halo_gn_supported(p, kTiles[pref[i]].bm) && kTiles[pref[i]].bm <= std::max(p.M, 128)) tile = pref[i];
    SDMI_CHECK(tile >= 0, "GroupNorm-folding conv: no halo tile fits this shape (the executor checks halo_gn_supported first)");
    if (splitk <= 0) {
      splitk = 1;
      if (can_split) {
        const long blocks = (long)cdiv(p.M, kTiles[tile].bm) * cdiv(p.N, kTiles[tile].bn);
        const int nch = Cin / BK;
        static const int cand[] = {2, 3, 4, 5, 6, 8, 10};
        for (int sk : cand)
          if (nch % sk == 0 && nch / sk >= 2 && blocks * sk <= 320 && splitk_ws_need(p, kTiles[tile].bm, kTiles[tile].bn, sk) <= p.splitk_ws_floats)
            splitk = sk;
      }
    }
  }
  if (p.split16) {
    if (tile >= 0) SDMI_CHECK(split16_tile_supported(tile), "split-fp16 GEMM: tile id without an instantiation");
    else tile = 5;                       // 64 x 64, 3 stages: the many-workgroup tile (table entries refine it)
  }
  if (tile < 0) {
    // heuristic for shapes the table does not know (round-1 sweep: the many-block 64x64 tile except for >= 25 GFLOP)
    const double gflop = 2.0 * p.M * (double)p.N * p.K * 1e-9;
    if (p.mode == EPI_GEGLU) tile = 0;
    else if (p.mode == EPI_HEADS) tile = 2;
    else tile = gflop >= 25.0 ? 3 : 5;
  }
  if (p.lnp_out) {
    // the row statistics ride on the 16-byte epilogue: every tile full and inside one sample -- a table / heuristic tile that
    // does not divide this shape gives way to the 64 x 64 one (3 stages), which does (checked above)
    const int hw = p.Hout * p.Wout;
    if (p.M % kTiles[tile].bm || p.N % kTiles[tile].bn || hw % kTiles[tile].bm || (p.split16 && !split16_tile_supported(tile))) tile = 5;
  }
  if (splitk <= 0) {  // auto: enough blocks to keep bytes in flight on all 256 CUs, >= 8 k-tiles per split
    splitk = 1;
    if (can_split) {
      const long blocks = (long)cdiv(p.M, kTiles[tile].bm) * cdiv(p.N, kTiles[tile].bn);
      const long want = (kTiles[tile].wm * kTiles[tile].wn == 8) ? 160 : 512;
      while (blocks * splitk < want && nkt / (splitk * 2) >= 8 && splitk < 16 &&
             splitk_ws_need(p, kTiles[tile].bm, kTiles[tile].bn, splitk * 2) <= p.splitk_ws_floats)
        splitk *= 2;
    }
  }
  if (splitk > 1) {
    SDMI_CHECK(can_split, "split-K needs plain or head-scatter mode, a slab workspace and N / ldo / ldr multiples of 4");
    SDMI_CHECK(splitk_ws_need(p, kTiles[tile].bm, kTiles[tile].bn, splitk) <= p.splitk_ws_floats, "split-K workspace too small");
  }
  if (ev0) SDMI_HIP_OK(hipEventRecord(ev0, stream));
#ifdef SDMI_IGEMM_TIMING
  static const char* dbg_path = getenv("SDMI_IGEMM_TIMING");       // debug: per-workgroup phase timing appended to this file
  static const int dbg_abl = env_int("SDMI_EPI_ABL", 0);
  static long long* dbg_buf = nullptr;
  constexpr int DBG_WG = 16384;
  IGemmParams pd = p;
  pd.dbg_abl = dbg_abl;
  if (dbg_path) {
    if (!dbg_buf) SDMI_HIP_OK(hipMalloc((void**)&dbg_buf, 6 * DBG_WG * sizeof(long long)));
    SDMI_HIP_OK(hipMemsetAsync(dbg_buf, 0, 6 * DBG_WG * sizeof(long long), stream));
    pd.dbg_times = dbg_buf;
  }
  const int rc = launch_tile(tile, pd, dma, splitk, stream);
  if (dbg_path && pd.dbg_times && rc == 0) {
    SDMI_HIP_OK(hipStreamSynchronize(stream));
    std::vector<long long> h(6 * DBG_WG);
    SDMI_HIP_OK(hipMemcpy(h.data(), dbg_buf, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    std::vector<long long> pro, loop, epi, sto, start;
    long long tmin = 0, tmax = 0;
    for (int b = 0; b < DBG_WG; ++b) {
      const long long* d = &h[6 * b];
      if (d[4] == 0) continue;
      pro.push_back(d[1] - d[0]); loop.push_back(d[2] - d[1]); epi.push_back(d[4] - d[2]);
      sto.push_back(d[3] ? d[3] - d[2] : d[4] - d[2]);           // epilogue up to the end of the output stores
      if (start.empty() || d[0] < tmin) tmin = d[0];
      if (d[4] > tmax) tmax = d[4];
      start.push_back(d[0]);
    }
    if (!pro.empty()) {
      auto med = [](std::vector<long long>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
      auto mx = [](std::vector<long long>& v) { return *std::max_element(v.begin(), v.end()); };
      long long late = 0;
      for (long long t : start) late = std::max(late, t - tmin);
      if (FILE* f = fopen(dbg_path, "a")) {
        const int nkt_split = (p.K / BK + std::max(splitk, 1) - 1) / std::max(splitk, 1);
        fprintf(f, "M%d N%d K%d k%d mode%d tile%d split%d res%d gn%d blocks%zu k-tiles/block %d | shader cycles: span %lld last-start %lld | median prologue %lld loop %lld epilogue %lld (stores %lld) | max loop %lld max epi %lld | loop cycles per k-tile %.1f\n",
                p.M, p.N, p.K, p.ksize, p.mode, tile, splitk, p.residual ? 1 : 0, p.gn_n, pro.size(), nkt_split, tmax - tmin, late, med(pro), med(loop),
                med(epi), med(sto), mx(loop), mx(epi), (double)med(loop) / nkt_split);
        fclose(f);
      }
    }
  }
#else
  const int rc = launch_tile(tile, p, dma, splitk, stream);
#endif
  if (rc == 0 && range_check_enabled()) {          // debug: fp16 outputs of this GEMM (MFMA operands of the next)
    const char* what = p.mode == EPI_GEGLU ? "igemm GEGLU output" : (p.mode == EPI_HEADS ? "igemm q/k/v^T" : "igemm fp16 output");
    if (p.mode == EPI_HEADS) {
      for (int sg = 0; sg < p.N / p.segC; ++sg)
        if (range_scan(what, p.seg_dst[sg], p.seg_kind[sg] == 0 ? (int64_t)p.M * p.segC : (int64_t)p.B * p.segC * p.ntok_pad, stream)) return -1;
    } else if (p.out_f16) {
      if (range_scan(what, p.out_f16, (int64_t)(p.M - 1) * p.ldo + (p.mode == EPI_GEGLU ? p.N / 2 : p.N), stream)) return -1;
    }
    if (p.ln_out && range_scan("LayerNorm output", p.ln_out, (int64_t)p.M * p.N, stream)) return -1;
  }
  if (ev0) {
    SDMI_HIP_OK(hipEventRecord(ev1, stream));
    std::lock_guard<std::mutex> lk(g_tuner.mu);
    g_tuner.recs.push_back({tkey, tcand, ev0, ev1});
  }
  return rc;
}

}  // namespace sdmi

