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
//     q / k / v^T for the attention kernel; split-K via fp32 atomics onto a pre-initialised output.
//   * blockIdx is remapped so that each XCD (private L2) owns a contiguous range of tiles.
#include "common.h"
#include "prof.h"

namespace sdmi {

int launch_splitk_reduce(const IGemmParams& p, int nsplit, hipStream_t stream);

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

namespace {

constexpr int BK = 64;

// exact-erf GELU (F.gelu default, attention.py:43).  erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e.
// 3 orders of magnitude below the fp16 rounding of the GEGLU output) -- the libm erff costs ~3x more VALU.
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = 1.0f / (1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * __expf(-z * z);
  const float erf_v = x < 0.f ? -erf_abs : erf_abs;
  return 0.5f * x * (1.0f + erf_v);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {   // counted wait: the immediate must be a literal
  static_assert(N == 0 || N == 4 || N == 6 || N == 8 || N == 12 || N == 16 || N == 18 || N == 24, "add the literal");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
}

// NS = LDS pipeline depth.  DMA path: NS-1 k-tiles are in flight across the (raw) barrier, retired by a counted
// s_waitcnt vmcnt(N); the global->LDS latency (~1 us under load) is several k-tiles of MFMA work, so NS = 2 leaves
// every block waiting on its single outstanding tile.
template <int BM, int BN, int WARPS_M, int WARPS_N, bool DMA, int NS, bool UP>
__global__ void __launch_bounds__(WARPS_M* WARPS_N * 64) igemm_kernel(const IGemmParams p, const int tiles_m,
                                                                       const int tiles_n, const int kt_per_split) {
  static_assert(DMA || NS == 2, "the register-staged path is double buffered");
  constexpr int NT = WARPS_M * WARPS_N * 64;
  constexpr int RPP = NT / 8;  // rows per load pass (8 chunks of 16 B per 128-B row)
  constexpr int A_PASSES = BM / RPP;
  constexpr int B_PASSES = BN / RPP;
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  static_assert(A_PASSES >= 1 && B_PASSES >= 1 && TM >= 1 && TN >= 1, "tile/wave shape");
  static_assert(RPP % 16 == 0, "swizzle assumes pass offset keeps row bits 1..3");

  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE_BYTES];

  // ---- XCD-aware tile assignment (dispatcher places block b on XCD b % 8; speed only) ----------------
  const int nblk = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tiles_mn = tiles_m * tiles_n;
  const int split = wgid / tiles_mn;
  const int tmn = wgid - split * tiles_mn;
  const int tile_n = tmn / tiles_m;
  const int tile_m = tmn - tile_n * tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nkt = p.K / BK;
  const int kt_begin = split * kt_per_split;
  const int kt_end = min(nkt, kt_begin + kt_per_split);
  if (kt_begin >= kt_end) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int cpos = tid & 7;                      // chunk position inside the LDS row
  const int lrow = tid >> 3;                     // row inside a load pass
  const int gch = cpos ^ ((lrow >> 1) & 7);      // global chunk that lands at (row, cpos)

  // ---- per-row gather metadata (computed once; the k-loop only adds wave-uniform offsets) -----------------
  // Row m of the implicit A matrix is output pixel (b, oy, ox).  Tap (ky, kx) of a 3x3 conv reads input pixel
  // (oy*stride + ky - 1, ox*stride + kx - 1): an offset that is affine in the tap, so per row we keep the element
  // offset of the centre tap and a 9-bit mask of the taps that fall inside the image.  (UP: nearest-x2 upsampled
  // input -- the source pixel is ((oy+ky-1)>>1, (ox+kx-1)>>1), not affine, so the three row / column offsets
  // are tabulated per row instead.)
  const int HWout = p.Hout * p.Wout;
  const int Cin = p.c0 + p.c1 + p.c2;
  const bool k3 = p.ksize == 3;
  const int pad = k3 ? p.pad : 0;                   // 1, or 0 for the VAE encoder's (0,1,0,1)-padded stride-2 conv
  const int ntap = p.ksize * p.ksize;
  const int ld = p.lda0;                            // all sources share the row pitch (checked by the launcher)
  int a_off[A_PASSES]; unsigned a_mask[A_PASSES];
  int a_ro[UP ? A_PASSES : 1][3], a_co[UP ? A_PASSES : 1][3];
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const int m = m0 + i * RPP + lrow;
    a_off[i] = 0; a_mask[i] = 0;
    if (m < p.M) {
      const int b = m / HWout;
      const int rem = m - b * HWout;
      const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
      const int pb = b * p.Hin * p.Win;
      if constexpr (UP) {
        const int Hv = 2 * p.Hin, Wv = 2 * p.Win;
        unsigned mk = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const int iy = oy + d - 1, ix = ox + d - 1;
          a_ro[i][d] = (pb + (max(iy, 0) >> 1) * p.Win) * ld;
          a_co[i][d] = (max(ix, 0) >> 1) * ld + gch * 8;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
          if (iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) mk |= 1u << t;
        }
        a_mask[i] = mk;
      } else {
        const int cy = oy * p.stride, cx = ox * p.stride;          // centre tap
        a_off[i] = (pb + cy * p.Win + cx) * ld + gch * 8;
        unsigned mk = 0;
        for (int t = 0; t < ntap; ++t) {
          const int iy = cy + (k3 ? t / 3 - pad : 0), ix = cx + (k3 ? t % 3 - pad : 0);
          if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) mk |= 1u << t;
        }
        a_mask[i] = mk;
      }
    }
  }
  int b_off[B_PASSES];
#pragma unroll
  for (int i = 0; i < B_PASSES; ++i) {
    const int n = n0 + i * RPP + lrow;
    b_off[i] = (n < p.N) ? (n * p.K + gch * 8) : -1;
  }

  f16x8 regA[DMA ? 1 : A_PASSES], regB[DMA ? 1 : B_PASSES];
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);     // provably scalar -> LDS-DMA bases stay in SGPRs

  // load cursor (wave-uniform): next k-tile to issue, its tap and first channel
  int ld_kt = kt_begin;
  // K order is chunk-major: k-tile kt = (64-channel chunk, tap), tap fastest (see pack_conv_kernel)
  int ld_tap = kt_begin % ntap;
  int ld_cin0 = (kt_begin / ntap) * BK;
  int ld_ky = k3 ? ld_tap / 3 : 0, ld_kx = k3 ? ld_tap - 3 * (ld_tap / 3) : 0;

  auto issue_loads = [&](int stage) {
    const bool live = ld_kt < kt_end;             // past the end (NS > 2): dummy loads from the zero page
    const f16* src; int coff;
    if (ld_cin0 < p.c0) { src = p.a0; coff = ld_cin0; }
    else if (ld_cin0 < p.c0 + p.c1) { src = p.a1; coff = ld_cin0 - p.c0; }
    else { src = p.a2; coff = ld_cin0 - p.c0 - p.c1; }
    const int tapoff = ((ld_ky - pad) * p.Win + (ld_kx - pad)) * ld + coff;     // scalar
    const unsigned tapbit = live ? (1u << ld_tap) : 0u;
    unsigned char* As = smem + stage * STAGE_BYTES;
    unsigned char* Bs = As + BM * 128;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      int off;
      if constexpr (UP) off = a_ro[i][ld_ky] + a_co[i][ld_kx] + coff;
      else off = a_off[i] + tapoff;
      const f16* g = (a_mask[i] & tapbit) ? (src + off) : p.zero_page;
      if constexpr (DMA) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(As + (i * RPP + wave_u * 8) * 128),
                                         16, 0, 0);
      } else {
        regA[i] = *(const f16x8*)g;
      }
    }
    const f16* wk = p.w + ld_kt * BK;
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
      const f16* g = (live && b_off[i] >= 0) ? (wk + b_off[i]) : p.zero_page;
      if constexpr (DMA) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(Bs + (i * RPP + wave_u * 8) * 128),
                                         16, 0, 0);
      } else {
        regB[i] = *(const f16x8*)g;
      }
    }
    // advance the cursor
    ++ld_kt;
    ++ld_tap;
    if (++ld_kx == 3) { ld_kx = 0; ++ld_ky; }
    if (ld_tap == ntap) { ld_tap = 0; ld_ky = 0; ld_kx = 0; ld_cin0 += BK; }
  };
  auto commit_regs = [&](int stage) {   // register-staged path: write the prefetched tile into LDS
    if constexpr (!DMA) {
      unsigned char* As = smem + stage * STAGE_BYTES;
      unsigned char* Bs = As + BM * 128;
#pragma unroll
      for (int i = 0; i < A_PASSES; ++i) *(f16x8*)(As + (i * RPP + lrow) * 128 + cpos * 16) = regA[i];
#pragma unroll
      for (int i = 0; i < B_PASSES; ++i) *(f16x8*)(Bs + (i * RPP + lrow) * 128 + cpos * 16) = regB[i];
    }
  };

  // ---- main loop -----------------------------------------------------------------------------------
  const int wm = wave / WARPS_N, wn = wave - wm * WARPS_N;
  const int l31 = lane & 31, lg = lane >> 5;
  const int rsw = (l31 >> 1) & 7;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int stage) {
    const unsigned char* As = smem + stage * STAGE_BYTES + (wm * WTM + l31) * 128;
    const unsigned char* Bs = smem + stage * STAGE_BYTES + BM * 128 + (wn * WTN + l31) * 128;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int coff = (((ks * 2 + lg) ^ rsw) << 4);
      f16x8 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *(const f16x8*)(As + i * 32 * 128 + coff);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *(const f16x8*)(Bs + j * 32 * 128 + coff);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };

  if constexpr (DMA) {
    constexpr int LPT = A_PASSES + B_PASSES;       // DMA instructions per thread per k-tile
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue_loads(s);
    int cur = 0, nxt = NS - 1;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      wait_vmcnt<LPT*(NS - 2)>();                  // this wave's share of tile kt has landed
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... and everybody's; stage nxt is free again
      if (!(p.debug & 1)) issue_loads(nxt);
      if (!(p.debug & 2)) compute(cur);
      cur = (cur + 1 == NS) ? 0 : cur + 1;
      nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
    }
    wait_vmcnt<0>();
  } else {
    issue_loads(0);
    commit_regs(0);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const int cur = (kt - kt_begin) & 1;
      __syncthreads();
      const bool more = (kt + 1 < kt_end);
      if (more) issue_loads(cur ^ 1);
      compute(cur);
      if (more) commit_regs(cur ^ 1);
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------------
  // C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int mw = m0 + wm * WTM, nw = n0 + wn * WTN;
  // Full interior tiles take a branch-free path: all residual loads of a 32-row slab are issued back to back
  // (independent), column terms are hoisted, and no per-element bounds checks split the stores into dependent
  // load -> wait -> store chains (those chains were ~70 % of the short-K kernels' time, profiles/ablate2_r01.txt).
  const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.N);
  if (p.mode == EPI_PLAIN) {
    const bool atomic = p.splitk > 1;      // split-K: raw partial sums go to this split's slab
    float* slab = atomic ? (p.splitk_ws + (size_t)split * p.M * p.N) : nullptr;
    const int b_first = m0 / HWout;
    const bool one_batch = ((m0 + BM - 1) / HWout == b_first);
    if (full && one_batch) {
      if (atomic) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float* row = slab + (size_t)(mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg) * p.N + nw + l31;
#pragma unroll
            for (int j = 0; j < TN; ++j) row[j * 32] = acc[i][j][r];
          }
      } else {
        float colv[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = nw + j * 32 + l31;
          colv[j] = p.bias ? p.bias[n] : 0.f;
          if (p.rowvec) colv[j] += p.rowvec[(size_t)b_first * p.ld_rowvec + n];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          float resv[16][TN];
          if (p.residual) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float* row = p.residual + (size_t)(mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg) * p.ldr + nw + l31;
#pragma unroll
              for (int j = 0; j < TN; ++j) resv[r][j] = row[j * 32];
            }
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
              for (int j = 0; j < TN; ++j) resv[r][j] = 0.f;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const size_t ro = (size_t)(mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg) * p.ldo + nw + l31;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const float v = acc[i][j][r] + colv[j] + resv[r][j];
              if (p.out_f32) p.out_f32[ro + j * 32] = v;
              if (p.out_f16) p.out_f16[ro + j * 32] = (f16)v;
              if (p.out_lo) p.out_lo[ro + j * 32] = (f16)(v - (float)(f16)v);
            }
          }
        }
      }
    } else {
      float bias_v[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = nw + j * 32 + l31;
        bias_v[j] = (!atomic && p.bias && n < p.N) ? p.bias[n] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
          if (m >= p.M) continue;
          const float* rv = (!atomic && p.rowvec) ? (p.rowvec + (size_t)(m / HWout) * p.ld_rowvec) : nullptr;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int n = nw + j * 32 + l31;
            if (n >= p.N) continue;
            float v = acc[i][j][r];
            if (atomic) {
              slab[(size_t)m * p.N + n] = v;
            } else {
              v += bias_v[j];
              if (rv) v += rv[n];
              if (p.residual) v += p.residual[(size_t)m * p.ldr + n];
              if (p.out_f32) p.out_f32[(size_t)m * p.ldo + n] = v;
              if (p.out_f16) p.out_f16[(size_t)m * p.ldo + n] = (f16)v;
              if (p.out_lo) p.out_lo[(size_t)m * p.ldo + n] = (f16)(v - (float)(f16)v);
            }
          }
        }
      }
    }
  } else if (p.mode == EPI_GEGLU) {
    if constexpr (TN % 2 == 0) {
#pragma unroll
      for (int j2 = 0; j2 < TN / 2; ++j2) {
        const int nv = nw + (2 * j2) * 32 + l31;      // value column (packed order), gate = nv + 32
        if (nv >= p.N) continue;
        const float bv = p.bias ? p.bias[nv] : 0.f, bg = p.bias ? p.bias[nv + 32] : 0.f;
        const int oc = (nw >> 1) + j2 * 32 + l31;
        if (full) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int m = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
              const float val = acc[i][2 * j2][r] + bv;
              const float gate = acc[i][2 * j2 + 1][r] + bg;
              p.out_f16[(size_t)m * p.ldo + oc] = (f16)(val * gelu_erf(gate));
            }
        } else {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int m = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
              if (m >= p.M) continue;
              const float val = acc[i][2 * j2][r] + bv;
              const float gate = acc[i][2 * j2 + 1][r] + bg;
              p.out_f16[(size_t)m * p.ldo + oc] = (f16)(val * gelu_erf(gate));
            }
        }
      }
    }
  } else {  // EPI_HEADS
    if (p.splitk > 1) {   // split-K: raw partial tile to this split's slab; splitk_reduce_heads_kernel scatters the sum
      float* slab = p.splitk_ws + (size_t)split * p.M * p.N;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
          if (m >= p.M) continue;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int n = nw + j * 32 + l31;
            if (n < p.N) slab[(size_t)m * p.N + n] = acc[i][j][r];
          }
        }
      return;
    }
    // lane = output column (seg, head, dd); registers 4q..4q+3 = 4 consecutive rows (tokens)
    const bool vec4 = (p.ntok % 4 == 0) && (p.ntok_pad % 4 == 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = nw + j * 32 + l31;
      if (n >= p.N) continue;
      const int seg = n / p.segC;
      const int c = n - seg * p.segC;
      const int head = c / p.dh;
      const int dd = c - head * p.dh;
      f16* dst = p.seg_dst[seg];
      const int kind = p.seg_kind[seg];
      if (p.bias) {                        // q/k/v projections with a bias (CLIP text model); the UNet's have none
        const float bv = p.bias[n];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += bv;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int mq = mw + i * 32 + 8 * r4 + 4 * lg;       // first of 4 consecutive rows (multiple of 4)
          if (mq >= p.M) continue;
          const int b = mq / p.ntok;
          const int tok = mq - b * p.ntok;
          const size_t bh = (size_t)b * p.heads + head;
          if (kind == 1 && vec4 && mq + 3 < p.M) {
            *(f16x4*)(dst + (bh * p.dh + dd) * p.ntok_pad + tok) =
                f16x4{(f16)acc[i][j][r4 * 4 + 0], (f16)acc[i][j][r4 * 4 + 1], (f16)acc[i][j][r4 * 4 + 2],
                      (f16)acc[i][j][r4 * 4 + 3]};
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int m = mq + e;
              if (m >= p.M) continue;
              const int b2 = m / p.ntok;
              const int t2 = m - b2 * p.ntok;
              const size_t bh2 = (size_t)b2 * p.heads + head;
              const size_t off = kind == 0 ? ((bh2 * p.ntok + t2) * p.dh + dd) : ((bh2 * p.dh + dd) * p.ntok_pad + t2);
              dst[off] = (f16)acc[i][j][r4 * 4 + e];
            }
          }
        }
    }
  }
}

// out = sum_s slab[s] + bias + rowvec[batch] + residual   (fixed summation order -> deterministic)
__global__ void __launch_bounds__(256) splitk_reduce_kernel(IGemmParams p, int nsplit) {
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
  for (int s = 1; s < 16; ++s) v += part[s];       // fixed order; absent splits add +0
  if (p.bias) v += *(const f32x4*)(p.bias + n);
  if (p.rowvec) v += *(const f32x4*)(p.rowvec + (size_t)(m / (p.Hout * p.Wout)) * p.ld_rowvec + n);
  if (p.residual) v += *(const f32x4*)(p.residual + (size_t)m * p.ldr + n);
  if (p.out_f32) *(f32x4*)(p.out_f32 + (size_t)m * p.ldo + n) = v;
  if (p.out_f16) *(f16x4*)(p.out_f16 + (size_t)m * p.ldo + n) = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
  if (p.out_lo) {
    f16x4 lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) lo[j] = (f16)(v[j] - (float)(f16)v[j]);
    *(f16x4*)(p.out_lo + (size_t)m * p.ldo + n) = lo;
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

template <int BM, int BN, int WARPS_M, int WARPS_N, int NS>
int launch_cfg(const IGemmParams& p, bool dma, int splitk, hipStream_t stream) {
  const int tiles_m = cdiv(p.M, BM), tiles_n = cdiv(p.N, BN);
  const int nkt = p.K / BK;
  const int kt_per_split = cdiv(nkt, splitk);
  const int nsplit = cdiv(nkt, kt_per_split);
  IGemmParams q = p;
  q.splitk = nsplit;
  static const int ablate = env_int("SDMI_IGEMM_ABLATE", 0);
  q.debug = ablate;
  dim3 grid(tiles_m * tiles_n * nsplit), block(WARPS_M * WARPS_N * 64);
  static const int by_shape = env_int("SDMI_PROF_SHAPES", 0);
  std::string pname = std::string("igemm_") + std::to_string(BM) + "x" + std::to_string(BN) + "s" + std::to_string(NS);
  if (by_shape && prof_enabled())
    pname += "_M" + std::to_string(p.M) + "_N" + std::to_string(p.N) + "_K" + std::to_string(p.K) + "_k" +
             std::to_string(p.ksize) + "_m" + std::to_string(p.mode) + "_s" + std::to_string(nsplit);
  const double src_pix = (double)p.B * p.Hin * p.Win;
  const double out_b = (p.out_f32 ? 4.0 : 0.0) + ((p.out_f16 || p.mode != EPI_PLAIN) ? 2.0 : 0.0);
  const double n_out = p.mode == EPI_GEGLU ? p.N / 2.0 : (double)p.N;
  ProfScope ps(pname.c_str(), 2.0 * p.M * (double)p.N * p.K,
               src_pix * (p.c0 + p.c1) * 2.0 + (double)p.N * p.K * 2.0 + (double)p.M * n_out * out_b +
                   (p.residual ? (double)p.M * p.N * 4.0 : 0.0),
               stream);
  if (p.up) {
    if (dma) hipLaunchKernelGGL((igemm_kernel<BM, BN, WARPS_M, WARPS_N, true, NS, true>), grid, block, 0, stream, q, tiles_m, tiles_n, kt_per_split);
    else hipLaunchKernelGGL((igemm_kernel<BM, BN, WARPS_M, WARPS_N, false, 2, true>), grid, block, 0, stream, q, tiles_m, tiles_n, kt_per_split);
  } else {
    if (dma) hipLaunchKernelGGL((igemm_kernel<BM, BN, WARPS_M, WARPS_N, true, NS, false>), grid, block, 0, stream, q, tiles_m, tiles_n, kt_per_split);
    else hipLaunchKernelGGL((igemm_kernel<BM, BN, WARPS_M, WARPS_N, false, 2, false>), grid, block, 0, stream, q, tiles_m, tiles_n, kt_per_split);
  }
  SDMI_HIP_OK(hipGetLastError());
  ps.end();
  if (nsplit > 1) return launch_splitk_reduce(q, nsplit, stream);     // (+ the LayerNorm launch when q.ln_out)
  if (q.ln_out) return launch_layernorm(q.out_f32, q.ln_gamma, q.ln_beta, q.ln_out, q.M, q.N, q.ln_eps, stream);
  return 0;
}

}  // namespace

int launch_splitk_reduce(const IGemmParams& p, int nsplit, hipStream_t stream) {
  SDMI_CHECK(nsplit >= 1 && nsplit <= 16 && p.N % 4 == 0 && p.splitk_ws, "splitk_reduce: bad arguments");
  if (p.mode == EPI_HEADS) {
    const int64_t total_h = (int64_t)p.M * (p.N / 4);
    ProfScope psh("splitk_reduce", 0.0, (double)p.M * p.N * (4.0 * nsplit + 2.0), stream);
    hipLaunchKernelGGL(splitk_reduce_heads_kernel, dim3((unsigned)((total_h + 255) / 256)), dim3(256), 0, stream, p, nsplit);
    SDMI_HIP_OK(hipGetLastError());
    return 0;
  }
  const int64_t total = (int64_t)p.M * (p.N / 4);
  ProfScope ps2("splitk_reduce", 0.0, (double)p.M * p.N * 4.0 * (nsplit + 1), stream);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p, nsplit);
  SDMI_HIP_OK(hipGetLastError());
  ps2.end();
  if (p.ln_out) return launch_layernorm(p.out_f32, p.ln_gamma, p.ln_beta, p.ln_out, p.M, p.N, p.ln_eps, stream);
  return 0;
}

int launch_igemm(const IGemmParams& p, const IGemmTune& tune, hipStream_t stream) {
  SDMI_CHECK(p.M > 0 && p.N > 0 && p.K > 0, "empty GEMM");
  SDMI_CHECK(p.ksize == 1 || p.ksize == 3, "ksize must be 1 or 3");
  const int Cin = p.c0 + p.c1 + p.c2;
  SDMI_CHECK(p.K == p.ksize * p.ksize * Cin, "K != ksize^2 * (c0 + c1 + c2)");
  SDMI_CHECK(Cin % BK == 0 && p.c0 % BK == 0 && p.c1 % BK == 0, "channel counts must be multiples of 64");
  SDMI_CHECK(p.lda0 % 8 == 0 && (p.a1 == nullptr || p.lda1 % 8 == 0), "A row pitch must be a multiple of 8 halves");
  SDMI_CHECK(p.zero_page != nullptr, "zero page missing");
  SDMI_CHECK(p.M == p.B * p.Hout * p.Wout, "M != B * Hout * Wout");
  SDMI_CHECK(p.c1 == 0 || p.a1 != nullptr, "second A source missing");
  SDMI_CHECK(p.c2 == 0 || (p.a2 != nullptr && p.lda2 % 8 == 0), "third A source missing");
  SDMI_CHECK((p.c1 == 0 || p.lda1 == p.lda0) && (p.c2 == 0 || p.lda2 == p.lda0), "all A sources must share one row pitch");
  SDMI_CHECK(!p.up || (p.ksize == 3 && p.stride == 1), "upsample folding needs a 3x3 stride-1 conv");
  SDMI_CHECK((int64_t)p.B * p.Hin * p.Win * p.lda0 < (int64_t)1 << 31 && (int64_t)p.N * p.K < (int64_t)1 << 31,
             "tensor too large for 32-bit element offsets");
  if (p.mode == EPI_GEGLU) SDMI_CHECK(p.N % 64 == 0 && p.out_f16 != nullptr, "GEGLU needs N % 64 == 0 and an fp16 output");
  if (p.ln_out)
    SDMI_CHECK(p.mode == EPI_PLAIN && p.out_f32 && p.ldo == p.N && p.N % 4 == 0 && p.N <= 2560 && p.ln_gamma && p.ln_beta,
               "LayerNorm post-op needs plain mode, an fp32 output with ldo == N <= 2560, gamma and beta");
  if (p.out_lo) SDMI_CHECK(p.mode == EPI_PLAIN && p.ldo % 4 == 0, "out_lo needs plain mode");
  if (p.mode == EPI_HEADS) SDMI_CHECK(p.segC > 0 && p.dh > 0 && p.N % p.segC == 0 && p.N / p.segC <= 3, "bad head scatter");

  static const int env_dma = env_int("SDMI_IGEMM_DMA", 1);
  static const int env_tile = env_int("SDMI_IGEMM_TILE", -1);
  const bool dma = (tune.dma >= 0 ? tune.dma : env_dma) != 0;
  int tile = tune.tile >= 0 ? tune.tile : env_tile;
  static const int env_geglu = env_int("SDMI_TILE_GEGLU", 0);
  if (p.mode == EPI_GEGLU && !(tile == 0 || tile == 3 || tile == 6 || tile == 7)) tile = env_geglu;   // GEGLU pairs 32-col tiles inside a wave
  // Tile / split-K choice, from the per-shape sweep of tests/tools/bench_kernels.py on MI355X (profiles/kbench_r01.txt):
  // every shape of this UNet is bound by L2->LDS bytes in flight, so the many-block 64x64 tile wins except for
  // the few >= 25 GFLOP convs, where the 256x128 tile (fewest bytes per FLOP) is ~10 % faster.
  const double gflop = 2.0 * p.M * (double)p.N * p.K * 1e-9;
  if (tile < 0) {
    static const int env_small = env_int("SDMI_TILE_SMALL", 5);       // tuning knobs for same-box A/B runs (tools/gpu_ab.sh)
    static const int env_t5kt = env_int("SDMI_T5_MIN_KT", 0);         // 3-stage tile only when K has >= this many k-tiles
    static const int env_heads = env_int("SDMI_TILE_HEADS", 2);
    static const int env_t3 = env_int("SDMI_T3_GFLOP", 25);
    static const int env_big = env_int("SDMI_TILE_BIG", 3);
    if (p.mode == EPI_HEADS) tile = env_heads;
    else tile = (gflop >= (double)env_t3) ? env_big : ((env_small == 5 && p.K / BK < env_t5kt) ? 2 : env_small);
  }
  const int BMs[8] = {128, 128, 64, 256, 128, 64, 256, 128}, BNs[8] = {128, 64, 64, 128, 64, 64, 128, 128};
  int splitk = p.splitk;
  const int nkt = p.K / BK;
  static const int env_split = env_int("SDMI_SPLITK", -1);     // 1 disables split-K everywhere
  const bool can_split_plain = p.mode == EPI_PLAIN && p.splitk_ws && p.N % 4 == 0 && p.ldo % 4 == 0 &&
                               (p.residual == nullptr || p.ldr % 4 == 0);
  // (the executors keep the head-scatter GEMMs unsplit: a same-box A/B showed split-K + reduce no faster there)
  const bool can_split_heads = p.mode == EPI_HEADS && p.splitk_ws && p.dh % 4 == 0 && p.segC % 4 == 0;
  const bool can_split = can_split_plain || can_split_heads;
  if (env_split >= 0 && splitk == 0) splitk = env_split;
  if (splitk <= 0) {  // auto: enough blocks to keep bytes in flight on all 256 CUs, >= 8 k-tiles per split
    splitk = 1;
    if (can_split) {
      const long blocks = (long)cdiv(p.M, BMs[tile]) * cdiv(p.N, BNs[tile]);
      static const int env_want = env_int("SDMI_SPLIT_WANT", 512);
      static const int env_want3 = env_int("SDMI_SPLIT_WANT3", 160);
      const long want = (tile == 3 || tile == 6) ? env_want3 : env_want;
      while (blocks * splitk < want && nkt / (splitk * 2) >= 8 && splitk < 16 &&
             (int64_t)(splitk * 2) * p.M * p.N <= p.splitk_ws_floats)
        splitk *= 2;
    }
  }
  if (splitk > 1) {
    SDMI_CHECK(can_split, "split-K needs plain or head-scatter mode, a slab workspace and N / ldo / ldr multiples of 4");
    SDMI_CHECK((int64_t)splitk * p.M * p.N <= p.splitk_ws_floats, "split-K workspace too small");
  }
  switch (tile) {            // tile ids: see include/sdmi.h (sdmi_igemm_desc.tile)
    case 0: return launch_cfg<128, 128, 2, 2, 2>(p, dma, splitk, stream);
    case 1: return launch_cfg<128, 64, 2, 2, 2>(p, dma, splitk, stream);
    case 2: return launch_cfg<64, 64, 2, 2, 2>(p, dma, splitk, stream);
    case 3: return launch_cfg<256, 128, 4, 2, 2>(p, dma, splitk, stream);
    case 4: return launch_cfg<128, 64, 2, 2, 3>(p, dma, splitk, stream);
    case 5: return launch_cfg<64, 64, 2, 2, 3>(p, dma, splitk, stream);
    case 6: return launch_cfg<256, 128, 4, 2, 3>(p, dma, splitk, stream);
    case 7: return launch_cfg<128, 128, 2, 2, 3>(p, dma, splitk, stream);
    default: return fail("unknown igemm tile id");
  }
}

}  // namespace sdmi
