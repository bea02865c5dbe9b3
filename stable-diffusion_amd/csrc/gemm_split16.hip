// Split-fp16 dense GEMM for gfx950 (MI355X): the 1x1 convolutions on the fp32 residual stream -- ResBlock skip_connection
// (ldm/modules/diffusionmodules/openaimodel.py:241), SpatialTransformer proj_in / proj_out (ldm/modules/attention.py:233-248).
//
//   out[M,N] = epilogue( a_hi * w_hi^T + a_lo * w_hi^T + a_hi * w_lo^T ),   x_lo = fp16(x - float(fp16(x)))
//
// ~22-bit operands on the fp16 MFMA (DESIGN.md "precision": these 46 GEMMs are 5 % of the FLOPs and were 45 % of the error
// variance as plain fp16).  Rounds 1-2 ran them through the generic kernel as ONE K-concatenated GEMM (A' = [hi | lo | hi],
// W' = [hi | hi | lo], K' = 3 K): six operand tiles per 64-channel chunk where four distinct ones exist, and -- the k-loop
// being bound by the latency of one k-tile, not by its bytes -- three loop iterations where one does.  This kernel stages the
// four tiles {a_hi, a_lo, w_hi, w_lo} of a 64-channel chunk once and issues the three MFMAs per fragment pair from them:
// a third of the iterations, two thirds of the bytes, the same products (summed per chunk instead of per pass).
//
// Structure = the generic kernel's LDS-DMA path (igemm.hip): [rows][128 B] tiles filled by MUBUF LDS-DMA with the 16-byte
// chunks XOR-swizzled on the source side, NS-deep ring with counted vmcnt waits, one raw s_barrier per k-tile, register
// double-buffered fragments, XCD-aware tile numbering, the shared epilogue (igemm_dev.h).  Weights keep the packed layout of
// rounds 1-2 ([N][3K] = [hi | hi | lo]: the blob format does not change): w_hi = columns [0, K), w_lo = columns [2K, 3K).
#include "igemm_dev.h"

namespace sdmi {
namespace {

template <int BM, int BN, int WARPS_M, int WARPS_N, int NS>
__global__ void __launch_bounds__(WARPS_M* WARPS_N * 64) gemm_split16_kernel(const IGemmParams p, const int tiles_m, const int tiles_n,
                                                                               const int kt_per_split) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NT = WARPS_M * WARPS_N * 64;
  constexpr int RPP = NT / 8;                    // rows per DMA pass (8 chunks of 16 B per 128-B row)
  constexpr int A_PASSES = BM / RPP, B_PASSES = BN / RPP;
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int STAGE_BYTES = 2 * (A_BYTES + B_BYTES);        // [a_hi | a_lo | w_hi | w_lo]
  constexpr int LPT = 2 * (A_PASSES + B_PASSES);              // DMA instructions per thread per k-tile
  constexpr int KS = BK / 16;
  constexpr int PPU = (LPT + KS - 2) / (KS - 1);              // DMA pieces issued in each of the first KS - 1 k-steps
  static_assert(A_PASSES >= 1 && B_PASSES >= 1 && TM >= 1 && TN >= 1 && RPP % 16 == 0 && NS >= 2, "tile/wave shape");
  static_assert(NS * STAGE_BYTES <= 160 * 1024, "LDS budget");

  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE_BYTES];

  // ---- XCD-aware tile assignment (as igemm_kernel) ----
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tiles_mn = tiles_m * tiles_n;
  const int split = wgid / tiles_mn;
  const int tmn = wgid - split * tiles_mn;
  int tile_m, tile_n;
  if (p.tile_n_fastest) { tile_m = tmn / tiles_n; tile_n = tmn - tile_m * tiles_n; }
  else { tile_n = tmn / tiles_m; tile_m = tmn - tile_n * tiles_m; }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nkt = p.K / BK;
  const int kt_begin = split * kt_per_split;
  const int kt_end = min(nkt, kt_begin + kt_per_split);
  if (kt_begin >= kt_end) return;

  const int tid = threadIdx.x;
  SDMI_STAMP(dbg_t0);
  const int lane = tid & 63, wave = tid >> 6;
  const int cpos = tid & 7, lrow = tid >> 3;
  const int gch = cpos ^ ((lrow >> 1) & 7);      // global chunk that lands at (row, cpos)
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  // rows past M / N are clamped to the last valid row (copies the epilogue never stores): every load unconditional, in bounds
  int a_off[A_PASSES], b_off[B_PASSES];
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) a_off[i] = (min(m0 + i * RPP + lrow, p.M - 1) * p.lda0 + gch * 8) * 2;
#pragma unroll
  for (int i = 0; i < B_PASSES; ++i) b_off[i] = (min(n0 + i * RPP + lrow, p.N - 1) * p.ldw + gch * 8) * 2;
  constexpr int OOB = (int)0x80000000;
  const __amdgpu_buffer_rsrc_t rsrc_hi = __builtin_amdgcn_make_buffer_rsrc((void*)p.a0, 0, OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_lo = __builtin_amdgcn_make_buffer_rsrc((void*)p.a1, 0, OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, OOB, 0x00020000);
  const int w_lo_off = 2 * p.K * 2;              // byte offset of the low halves inside a packed weight row ([hi | hi | lo])

  // DMA piece q of k-tile kt into ring stage `stage`: q in [0, A_PASSES) a_hi, then a_lo, then w_hi, then w_lo
  auto issue_piece = [&](int kt, int stage, int q) {
    const int soff = kt * (BK * 2);
    const bool isA = q < 2 * A_PASSES;
    const int qa = isA ? q : q - 2 * A_PASSES;
    const int npass = isA ? A_PASSES : B_PASSES;
    const int half = qa >= npass ? 1 : 0, pass = qa - half * npass;
    const unsigned row0 = (isA ? half * BM : 2 * BM + half * BN) + pass * RPP + wave_u * 8;
    auto dst = (__attribute__((address_space(3))) void*)(smem + stage * STAGE_BYTES + row0 * 128);
    if (isA) __builtin_amdgcn_raw_ptr_buffer_load_lds(half ? rsrc_lo : rsrc_hi, dst, 16, a_off[pass], soff, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, dst, 16, b_off[pass], soff + (half ? w_lo_off : 0), 0, SDMI_W_AUX);
  };

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

  // fragments of k-step ks: a_hi / a_lo rows of the wave's TM tiles, w_hi / w_lo rows of its TN tiles
  struct Frags { f16x8 ah[TM], al[TM], bh[TN], bl[TN]; };
  const int a_lds = (wm * WTM + l31) * 128, b_lds = 2 * A_BYTES + (wn * WTN + l31) * 128;
  auto read_frags = [&](int stage, int ks, Frags& f) {
    const unsigned char* st = smem + stage * STAGE_BYTES + (((ks * 2 + lg) ^ rsw) << 4);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      f.ah[i] = *(const f16x8*)(st + a_lds + i * 32 * 128);
      f.al[i] = *(const f16x8*)(st + a_lds + A_BYTES + i * 32 * 128);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      f.bh[j] = *(const f16x8*)(st + b_lds + j * 32 * 128);
      f.bl[j] = *(const f16x8*)(st + b_lds + B_BYTES + j * 32 * 128);
    }
  };
  auto mfma_step = [&](const Frags& f) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
      }
  };

  SDMI_STAMP(dbg_t1);
  // ---- software pipeline: NS - 1 k-tiles in flight across one raw barrier per k-tile (see igemm_kernel) ----
  // The load cursor stops on the last k-tile of the split: the NS - 1 surplus issues reload it (in bounds, never consumed).
  int ld_kt = kt_begin;
  auto next_kt = [&]() { const int k = ld_kt; if (ld_kt + 1 < kt_end) ++ld_kt; return k; };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) {
    const int k = next_kt();
#pragma unroll
    for (int q = 0; q < LPT; ++q) issue_piece(k, s, q);
  }
  wait_vmcnt<LPT*(NS - 2)>();
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  Frags fr[2];
  read_frags(0, 0, fr[0]);
  int cur = 0, nxt = NS - 1;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int kload = next_kt();
    const int cur1 = (cur + 1 == NS) ? 0 : cur + 1;
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      if (u + 1 < KS) {
        read_frags(cur, u + 1, fr[(u + 1) & 1]);
#pragma unroll
        for (int q = u * PPU; q < (u + 1) * PPU && q < LPT; ++q) issue_piece(kload, nxt, q);
      } else {
        wait_vmcnt<LPT*(NS - 2)>();               // this wave's share of tile kt + 1 has landed
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... and everybody's; tile kt is fully read
        read_frags(cur1, 0, fr[0]);
      }
      mfma_step(fr[u & 1]);
      // issue order of the unit: the next unit's fragment reads first (they land under this unit's MFMAs), then MFMAs with
      // one LDS-DMA issue in each of the first gaps (masks: 0x100 DS read, 0x008 MFMA, 0x010 VMEM)
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);
#pragma unroll
      for (int e = 0; e < 3 * TM * TN; ++e) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (u + 1 < KS && e < PPU && u * PPU + e < LPT) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
    }
    cur = cur1;
    nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
  }
  wait_vmcnt<0>();
  SDMI_STAMP(dbg_t2);
  igemm_epilogue<BM, BN, WARPS_M, WARPS_N, NS * STAGE_BYTES>(p, acc, m0, n0, split, tile_m, tile_n, smem);
#ifdef SDMI_IGEMM_TIMING
  if (p.dbg_times && tid == 0) {
    long long* d = p.dbg_times + 6 * (size_t)blockIdx.x;
    d[0] = dbg_t0; d[1] = dbg_t1; d[2] = dbg_t2; d[4] = (long long)__builtin_readcyclecounter();
  }
#endif
#endif  // __HIP_DEVICE_COMPILE__
}

#ifdef SDMI_EXPERIMENTS      // (bit-identical, 15 launches fewer, +0.19 ms per UNet call in round 3; the row-strip chain kernel st_head does this job now)
// ---- the same GEMM with GroupNorm(32) of its input rows applied while the A operand is staged -------------------------------------
// SpatialTransformer.forward, attention.py:254-255: x = proj_in(norm(x)) -- GroupNorm(32, eps 1e-6, no activation) straight into a 1x1
// conv.  The stand-alone path is a GroupNorm-apply launch (fp32 stream in, split-fp16 hi | lo out) and this GEMM reading those two
// operands.  Here the GEMM reads the fp32 stream itself: a thread owns 2 x 8 channels of a 64 x 64 A tile per k-tile, applies
// gn_apply_elem (the stand-alone kernel's arithmetic: the same operand bits) and writes the hi and lo tiles into LDS; the weight
// tiles (w_hi, w_lo) come by LDS-DMA as before.  Unlike the 3x3 convolution (conv3halo_gn_kernel, measured slower) there is no
// SiLU -- 4 VALU operations per element, no transcendental -- and the 16 launches this removes cost ~9 us each.
// Two kinds of loads are in flight (register loads of x / gamma / beta, LDS-DMA of the weights): every wait is a full drain
// (vmcnt(0)), once per k-tile -- the two kinds do not retire through one in-order queue (profiles/gn_fold_r03.txt).
//   per k-tile:  [drain] [convert the x registers of tile kt -> a_hi | a_lo of stage kt & 1] [barrier]
//                [request: weights of tile kt + 1 (DMA, other stage), x / gamma / beta of tile kt + 1 (registers)] [12 MFMAs of tile kt]
template <int NSTAT>
__global__ void __launch_bounds__(256) gemm_split16_gn_kernel(const IGemmParams p, const int tiles_m, const int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 64, BN = 64, WARPS_M = 2, WARPS_N = 2;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int STAGE_BYTES = 2 * (A_BYTES + B_BYTES);        // [a_hi | a_lo | w_hi | w_lo]
  constexpr int KS = BK / 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_BYTES + 256];
  float* const tab = (float*)(smem + 2 * STAGE_BYTES);        // {mean, rstd} of the 32 groups of this tile's sample

  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  int tile_m, tile_n;
  if (p.tile_n_fastest) { tile_m = wgid / tiles_n; tile_n = wgid - tile_m * tiles_n; }
  else { tile_n = wgid / tiles_m; tile_m = wgid - tile_n * tiles_m; }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nkt = p.K / BK;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int cpos = tid & 7, lrow = tid >> 3;                  // this thread's 8 channels of a k-tile, its rows lrow and lrow + 32
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int C = p.K, cpg = C / 32;
  const int HW = p.Hout * p.Wout;
  const int bsample = m0 / HW;                                // (the launcher checked HW % 64 == 0: one sample per tile)

  // ---- weights by LDS-DMA: 2 x 2 pieces per thread and k-tile ----
  constexpr int OOB = (int)0x80000000;
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, OOB, 0x00020000);
  const int gch = cpos ^ ((lrow >> 1) & 7);
  int b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) b_off[i] = (min(n0 + i * 32 + lrow, p.N - 1) * p.ldw + gch * 8) * 2;
  const int w_lo_off = 2 * p.K * 2;
  auto issue_w = [&](int kt, int stage) {
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        auto dst = (__attribute__((address_space(3))) void*)(smem + stage * STAGE_BYTES + (2 * BM + half * BN + i * 32 + wave_u * 8) * 128);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, dst, 16, b_off[i], kt * (BK * 2) + (half ? w_lo_off : 0), 0, 0);
      }
  };
  // ---- x, gamma, beta into registers: rows m0 + lrow (+ 32), channels 64 kt + 8 cpos .. + 7 ----
  const float* const xr0 = p.xf0 + (size_t)min(m0 + lrow, p.M - 1) * C + cpos * 8;
  const float* const xr1 = p.xf0 + (size_t)min(m0 + lrow + 32, p.M - 1) * C + cpos * 8;
  f32x4 xv[2][2], gv[2], bv[2];
  auto load_x = [&](int kt) {
    const int c = kt * BK;
    xv[0][0] = *(const f32x4*)(xr0 + c); xv[0][1] = *(const f32x4*)(xr0 + c + 4);
    xv[1][0] = *(const f32x4*)(xr1 + c); xv[1][1] = *(const f32x4*)(xr1 + c + 4);
    gv[0] = *(const f32x4*)(p.gn_in_gamma + c + cpos * 8); gv[1] = *(const f32x4*)(p.gn_in_gamma + c + cpos * 8 + 4);
    bv[0] = *(const f32x4*)(p.gn_in_beta + c + cpos * 8); bv[1] = *(const f32x4*)(p.gn_in_beta + c + cpos * 8 + 4);
  };
  issue_w(0, 0);
  load_x(0);
  // ---- {mean, rstd} of the sample's 32 groups (complete before the launch): 8 lanes fold the 8 slots of a group ----
  {
    const int g = tid >> 3, sub = tid & 7;
    const long long* src = p.gn_in_acc + ((size_t)(bsample * 32 + g) * GN_SLOTS + sub) * GN_STRIDE;
    long long a = src[0], al = src[1], q = src[2], ql = src[3];
#pragma unroll
    for (int o = GN_SLOTS / 2; o >= 1; o >>= 1) {
      a += __shfl_xor(a, o); al += __shfl_xor(al, o); q += __shfl_xor(q, o); ql += __shfl_xor(ql, o);
    }
    if (sub == 0) {
      const double nel = (double)cpg * (double)HW;
      const double m = gn_acc_value(a, al) / nel;
      double var = gn_acc_value(q, ql) / nel - m * m;
      if (var < 0.0) var = 0.0;
      tab[g * 2] = (float)m;
      tab[g * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.gn_in_eps));
    }
  }

  const int wm = wave / WARPS_N, wn = wave - wm * WARPS_N;
  const int l31 = lane & 31, lg = lane >> 5;
  const int rsw = (l31 >> 1) & 7;
  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  const int a_lds = (wm * 32 + l31) * 128, b_lds = 2 * A_BYTES + (wn * 32 + l31) * 128;
  // LDS position of this thread's 16 bytes in an A tile: row lrow (+ 32), chunk cpos ^ swizzle(row) -- what the DMA of the
  // stand-alone path writes (RPP = 32: the swizzle of row lrow + 32 is that of lrow)
  const int a_wr = lrow * 128 + ((cpos ^ ((lrow >> 1) & 7)) << 4);
  const unsigned long long magic_cpg = p.magic_cpg_in;

  for (int kt = 0; kt < nkt; ++kt) {
    const int st = kt & 1;
    wait_vmcnt<0>();                                          // the weights and the x / gamma / beta registers of tile kt have landed
    if (kt == 0) __syncthreads();                             // (the statistics table)
    {
      // normalise, split into hi | lo, write the two A tiles of this stage
      const int c0 = kt * BK + cpos * 8;
      const int g0 = fast_div(c0, magic_cpg);
      const int nfirst = (g0 + 1) * cpg - c0;                 // channels of the octet in group g0 (cpg >= 8: at most two groups)
      const float m_a = tab[g0 * 2], r_a = tab[g0 * 2 + 1], m_b = tab[min(g0 + 1, 31) * 2], r_b = tab[min(g0 + 1, 31) * 2 + 1];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool second = j >= nfirst;
          const float y = gn_apply_elem(xv[h][j >> 2][j & 3], second ? m_b : m_a, second ? r_b : r_a, gv[j >> 2][j & 3], bv[j >> 2][j & 3], 0);
          hi[j] = (f16)y; lo[j] = (f16)(y - (float)hi[j]);
        }
        *(f16x8*)(smem + st * STAGE_BYTES + h * 32 * 128 + a_wr) = hi;
        *(f16x8*)(smem + st * STAGE_BYTES + A_BYTES + h * 32 * 128 + a_wr) = lo;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // tile kt is complete for everybody; tile kt - 1 is fully read
    if (kt + 1 < nkt) { issue_w(kt + 1, st ^ 1); load_x(kt + 1); }
    const unsigned char* base = smem + st * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int ch = (((ks * 2 + lg) ^ rsw) << 4);
      const f16x8 ah = *(const f16x8*)(base + a_lds + ch), al = *(const f16x8*)(base + A_BYTES + a_lds + ch);
      const f16x8 bh = *(const f16x8*)(base + b_lds + ch), bl = *(const f16x8*)(base + B_BYTES + b_lds + ch);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[0][0], 0, 0, 0);
    }
  }
  wait_vmcnt<0>();
  igemm_epilogue<BM, BN, WARPS_M, WARPS_N, 2 * STAGE_BYTES>(p, acc, m0, n0, 0, tile_m, tile_n, smem);
#endif  // __HIP_DEVICE_COMPILE__
}

#endif  // SDMI_EXPERIMENTS

template <int BM, int BN, int WARPS_M, int WARPS_N, int NS>
int launch_split16_cfg(const IGemmParams& p, int splitk, hipStream_t stream) {
  const int tiles_m = cdiv(p.M, BM), tiles_n = cdiv(p.N, BN);
  const int nkt = p.K / BK;
  const int kt_per_split = cdiv(nkt, splitk);
  const int nsplit = cdiv(nkt, kt_per_split);
  IGemmParams q = p;
  q.splitk = nsplit;
  q.tile_n_fastest = tile_order_n_fastest(p);
  q.splitk_fused = nsplit > 1 && splitk_fusable(p, BM, BN);
  slab_layout(q, BM, BN, WARPS_M, WARPS_N, nsplit);
  q.epi_vec = epi_vec_ok(p);
  SDMI_CHECK(splitk_ws_need(p, BM, BN, nsplit) <= p.splitk_ws_floats, "split-K workspace too small");
  SDMI_CHECK((int64_t)p.M < (int64_t)65536 * p.Hout * p.Wout, "fast_div_hw: at most 65535 samples per launch");
  q.magic_hw = div_magic_hw(p.Hout * p.Wout);
  q.magic_w = div_magic(p.Wout);
  for (int t = 0; t < p.gn_n; ++t) q.gn_magic[t] = div_magic(p.gn_cpg[t]);
  dim3 grid(tiles_m * tiles_n * nsplit), block(WARPS_M * WARPS_N * 64);
  static const int by_shape = env_int("SDMI_PROF_SHAPES", 0);
  std::string pname = std::string("gemm_split16_") + std::to_string(BM) + "x" + std::to_string(BN) + "w" +
                      std::to_string(WARPS_M * WARPS_N) + "s" + std::to_string(NS);
  if (by_shape && prof_enabled())
    pname += "_M" + std::to_string(p.M) + "_N" + std::to_string(p.N) + "_K" + std::to_string(p.K) + "_s" + std::to_string(nsplit);
  // algorithmic FLOPs / bytes of the reference 1x1 conv (one fp16 read of the activation and of the weights); executed: 3 passes
  ProfScope ps(pname.c_str(), 2.0 * p.M * (double)p.N * p.K,
               (double)p.M * p.K * 2.0 + (double)p.N * p.K * 2.0 + (double)p.M * p.N * ((p.out_f32 ? 4.0 : 0.0) + (p.out_f16 ? 2.0 : 0.0)) +
                   (p.residual ? (double)p.M * p.N * 4.0 : 0.0),
               stream, 6.0 * p.M * (double)p.N * p.K);
  SDMI_LAUNCH((gemm_split16_kernel<BM, BN, WARPS_M, WARPS_N, NS>), grid, block, 0, stream, q, tiles_m, tiles_n, kt_per_split);
  SDMI_HIP_OK(hipGetLastError());
  ps.end();
  if (nsplit > 1 && !q.splitk_fused) return launch_splitk_reduce(q, nsplit, stream);
  if (q.ln_out) return launch_layernorm(q.out_f32, q.ln_gamma, q.ln_beta, q.ln_out, q.M, q.N, q.ln_eps, stream);
  return 0;
}

}  // namespace

// tile ids of the table in igemm.hip (kTiles) this family instantiates: the tile SHAPE of that id with an LDS ring that fits the
// doubled stage (four operand tiles per k-tile)
bool split16_tile_supported(int tile) { return tile == 0 || tile == 1 || tile == 2 || tile == 4 || tile == 5 || tile == 8 || tile == 10; }

// out = epilogue( GroupNorm32(x) W^T ) with the split-fp16 operands produced inside the GEMM (gemm_split16_gn_kernel): x = p.xf0 fp32
// [M][K], statistics p.gn_in_acc (complete), p.gn_in_gamma / beta / eps, weights packed [N][3K]
bool split16_gn_supported(const IGemmParams& p) {
#ifndef SDMI_EXPERIMENTS
  return false;      // (product build: the kernel is not compiled in)
#endif
  const int hw = p.Hout * p.Wout;
  return p.ksize == 1 && p.mode == EPI_PLAIN && p.K % BK == 0 && (p.K / 32) >= 8 && p.M % 64 == 0 && hw % 64 == 0 && p.N % 64 == 0 &&
         (int64_t)p.M * p.K * 4 < ((int64_t)1 << 31);
}
int launch_split16_gn(const IGemmParams& p, hipStream_t stream) {
#ifndef SDMI_EXPERIMENTS
  return fail("the GroupNorm-folding split-fp16 GEMM (gemm_split16_gn_kernel) is an experiment: build with SDMI_CXXFLAGS=-DSDMI_EXPERIMENTS");
#else
  SDMI_CHECK(p.xf0 && p.gn_in_acc && p.gn_in_gamma && p.gn_in_beta && p.w && p.ldw >= 3 * p.K && split16_gn_supported(p),
             "GroupNorm-folding split-fp16 GEMM: fp32 rows, statistics, gamma / beta, packed [N][3K] weights, M / N / rows per sample multiples of 64");
  const int tiles_m = p.M / 64, tiles_n = p.N / 64;
  IGemmParams q = p;
  q.splitk = 1;
  q.tile_n_fastest = tile_order_n_fastest(p);
  q.splitk_fused = 0;
  q.epi_vec = epi_vec_ok(p);
  SDMI_CHECK((int64_t)p.M < (int64_t)65536 * p.Hout * p.Wout, "fast_div_hw: at most 65535 samples per launch");
  q.magic_hw = div_magic_hw(p.Hout * p.Wout);
  q.magic_w = div_magic(p.Wout);
  q.magic_cpg_in = div_magic(p.K / 32);
  for (int t = 0; t < p.gn_n; ++t) q.gn_magic[t] = div_magic(p.gn_cpg[t]);
  static const int by_shape = env_int("SDMI_PROF_SHAPES", 0);
  std::string pname = "gemm_split16_gn_64x64w4s2";
  if (by_shape && prof_enabled()) pname += "_M" + std::to_string(p.M) + "_N" + std::to_string(p.N) + "_K" + std::to_string(p.K);
  ProfScope ps(pname.c_str(), 2.0 * p.M * (double)p.N * p.K,
               (double)p.M * p.K * 4.0 + (double)p.N * p.K * 2.0 + (double)p.M * p.N * ((p.out_f32 ? 4.0 : 0.0) + (p.out_f16 ? 2.0 : 0.0)) +
                   (p.residual ? (double)p.M * p.N * 4.0 : 0.0),
               stream, 6.0 * p.M * (double)p.N * p.K);
  SDMI_LAUNCH((gemm_split16_gn_kernel<0>), dim3(tiles_m * tiles_n), dim3(256), 0, stream, q, tiles_m, tiles_n);
  SDMI_HIP_OK(hipGetLastError());
  ps.end();
  return 0;
#endif
}

int launch_split16_tile(int tile, const IGemmParams& p, int splitk, hipStream_t stream) {
  SDMI_CHECK(p.a0 && p.a1 && p.ksize == 1 && p.mode == EPI_PLAIN && p.c1 == 0 && p.c2 == 0 && p.K == p.c0 && p.K % BK == 0 &&
                 p.lda0 % 8 == 0 && p.ldw % 8 == 0 && p.ldw >= 3 * p.K,
             "split-fp16 GEMM: hi + lo operands, plain epilogue, K % 64 == 0, packed [N][3K] weights");
  switch (tile) {
    case 0: return launch_split16_cfg<128, 128, 2, 2, 2>(p, splitk, stream);      // 128 KB
    case 1: return launch_split16_cfg<128, 64, 2, 2, 2>(p, splitk, stream);       //  96 KB
    case 2: return launch_split16_cfg<64, 64, 2, 2, 2>(p, splitk, stream);        //  64 KB (2 workgroups / CU)
    case 4: return launch_split16_cfg<128, 64, 2, 2, 3>(p, splitk, stream);       // 144 KB
    case 5: return launch_split16_cfg<64, 64, 2, 2, 3>(p, splitk, stream);        //  96 KB
    case 8: return launch_split16_cfg<64, 128, 2, 2, 3>(p, splitk, stream);       // 144 KB
    case 10: return launch_split16_cfg<64, 64, 2, 2, 4>(p, splitk, stream);       // 128 KB
    default: return fail("split-fp16 GEMM: tile id without an instantiation");
  }
}

}  // namespace sdmi
