// The generic implicit-GEMM kernel (igemm_kernel) and its launcher template (launch_cfg): included by the translation units that
// instantiate its tile shapes (igemm_t0.hip / igemm_t1.hip / igemm_t2.hip -- three files so that the 60-odd instantiations compile in
// parallel; igemm.hip keeps the split-K reductions, the tile table, the tuner and launch_igemm).  Design notes: igemm.hip.
#pragma once
#include <string>

#include "igemm_dev.h"

namespace sdmi {
namespace {


// NS = LDS pipeline depth.  DMA path: NS-1 k-tiles are in flight across the (raw) barrier, retired by a counted
// s_waitcnt vmcnt(N); the global->LDS latency (~1 us under load) is several k-tiles of MFMA work, so NS = 2 leaves
// every block waiting on its single outstanding tile.
template <int BM, int BN, int WARPS_M, int WARPS_N, bool DMA, int NS, int KIND>
__global__ void __launch_bounds__(WARPS_M* WARPS_N * 64) igemm_kernel(const IGemmParams p, const int tiles_m,
                                                                       const int tiles_n, const int kt_per_split) {
  static_assert(DMA || NS == 2, "the register-staged path is double buffered");
  // The body uses gfx950-only types / builtins (buffer descriptors, LDS-DMA); hipcc's host pass only needs the launch
  // stub, and silently drops the stub of an instantiation whose body it cannot type-check -- so the body is device-only.
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr bool UP = KIND == KIND_3X3_UP;
  constexpr bool K3 = KIND != KIND_1X1;
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
  // which operand an XCD keeps to itself: an XCD runs a contiguous range of tile numbers, and its L2 is private.  With more
  // A bytes than weight bytes (M > N) the range walks N fastest -- few row panels of A, every weight panel -- so A is
  // fetched from the fabric by ONE XCD instead of all eight; the weight-heavy shapes (M <= N) keep walking M fastest.
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
  // LayerNorm of the A rows folded into this GEMM: the row-statistics partials are requested first of all (see lnf_request)
  float2 lnf_pv[LNF_MAXP];
  float lnf_mean = 0.f, lnf_rstd = 1.f;
  const bool lnf_mine = p.lnf_part != nullptr && tid < BM;
  if (lnf_mine) lnf_request(p, min(m0 + tid, p.M - 1), lnf_pv);
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int cpos = tid & 7;                      // chunk position inside the LDS row
  const int lrow = tid >> 3;                     // row inside a load pass
  const int gch = cpos ^ ((lrow >> 1) & 7);      // global chunk that lands at (row, cpos)

  // ---- per-row gather metadata (computed once; the k-loop only adds wave-uniform offsets) -----------------
  // Row m of the implicit A matrix is output pixel (b, oy, ox).  Tap (ky, kx) of a 3x3 conv reads input pixel
  // (oy*stride + ky - pad, ox*stride + kx - pad): an offset that is affine in the tap, so per row we keep the byte
  // offset of tap (pad, pad) and a 9-bit mask of the taps that fall inside the image.  (UP: nearest-x2 upsampled
  // input -- the source pixel is ((oy+ky-1)>>1, (ox+kx-1)>>1), not affine, so the three row / column offsets
  // are tabulated per row instead.)  Rows past M (and weight rows past N) are CLAMPED to the last valid row: they
  // compute a copy of it that the epilogue never stores, which keeps every load unconditional and in bounds.
  const int HWout = p.Hout * p.Wout;
  const int pad = K3 ? p.pad : 0;                   // 1, or 0 for the VAE encoder's (0,1,0,1)-padded stride-2 conv
  constexpr int ntap = K3 ? 9 : 1;
  const int ld = p.lda0;                            // all sources share the row pitch (checked by the launcher)
  int a_off[A_PASSES];                              // byte offsets (< 2^31, checked by the launcher)
  unsigned a_mask[K3 ? A_PASSES : 1];
  int a_ro[UP ? A_PASSES : 1][3], a_co[UP ? A_PASSES : 1][3];
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const int m = min(m0 + i * RPP + lrow, p.M - 1);
    const int b = fast_div_hw(m, p.magic_hw);
    const int rem = m - b * HWout;
    const int oy = fast_div(rem, p.magic_w), ox = rem - oy * p.Wout;
    const int pb = b * p.Hin * p.Win;
    if constexpr (UP) {
      const int Hv = 2 * p.Hin, Wv = 2 * p.Win;
      unsigned mk = 0;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int iy = oy + d - 1, ix = ox + d - 1;
        a_ro[i][d] = (pb + (max(iy, 0) >> 1) * p.Win) * ld * 2;
        a_co[i][d] = ((max(ix, 0) >> 1) * ld + gch * 8) * 2;
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
        if (iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) mk |= 1u << t;
      }
      a_mask[i] = mk;
      a_off[i] = 0;
    } else {
      const int cy = oy * p.stride, cx = ox * p.stride;          // tap (pad, pad)
      a_off[i] = ((pb + cy * p.Win + cx) * ld + gch * 8) * 2;
      if constexpr (K3) {
        unsigned mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = cy + t / 3 - pad, ix = cx + t % 3 - pad;
          if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) mk |= 1u << t;
        }
        a_mask[i] = mk;
      }
    }
  }
  int b_off[B_PASSES];
#pragma unroll
  for (int i = 0; i < B_PASSES; ++i) {
    const int n = min(n0 + i * RPP + lrow, p.N - 1);
    b_off[i] = (n * p.K + gch * 8) * 2;
  }

  f16x8 regA[DMA ? 1 : A_PASSES], regB[DMA ? 1 : B_PASSES];
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);     // provably scalar -> LDS-DMA bases stay in SGPRs

  // load cursor (wave-uniform): next k-tile to issue, its tap and first channel.  It stops on the last k-tile of this
  // split: the NS - 1 surplus issues at the end of the pipeline reload that tile (in bounds, never consumed).
  int ld_kt = kt_begin;
  // K order is chunk-major: k-tile kt = (64-channel chunk, tap), tap fastest (see pack_conv_kernel)
  int ld_tap = K3 ? kt_begin % ntap : 0;
  int ld_cin0 = (kt_begin / ntap) * BK;
  int ld_ky = K3 ? ld_tap / 3 : 0, ld_kx = K3 ? ld_tap - 3 * (ld_tap / 3) : 0;

  // Operands are addressed through buffer descriptors (MUBUF): address = base + per-lane voffset + scalar soffset, so a
  // pass costs no 64-bit VALU address arithmetic, and an out-of-image tap is a lane whose voffset is beyond num_records:
  // the load returns zeros (also into LDS), no zero page and no pointer select.  MUBUF LDS-DMA also keeps the compiler's
  // LDS wait counts exact: beside a FLAT-encoded global_load_lds every ds_read wait degrades to lgkmcnt(0) (round-1 ISA).
  // The A base is moved back by the offset of tap (0, 0) relative to tap (pad, pad), so every tap's soffset is >= 0.
  // Everything the k-loop touches lives in registers (no IGemmParams re-reads: those are scalar memory loads).
  constexpr int OOB = (int)0x80000000;              // >= num_records of every descriptor below
  const long long a_shift = (K3 && !UP) ? (long long)(pad * p.Win + pad) * ld * 2 : 0;
  const char* const srcA0 = (const char*)p.a0 - a_shift; const char* const srcA1 = (const char*)p.a1 - a_shift;
  const char* const srcA2 = (const char*)p.a2 - a_shift;
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, OOB, 0x00020000);
  const int pc0 = p.c0, pc01 = p.c0 + p.c1, pWin = p.Win;

  // the scalar part of one k-tile's addresses, captured when the tile is scheduled; the per-pass issues may come later
  struct TileCursor { __amdgpu_buffer_rsrc_t rsrc_a; int a_soff, b_soff; unsigned tapbit; int ky, kx; unsigned lds; };
  auto next_tile = [&](int stage) {
    TileCursor c;
    const char* src; int coff;
    if (ld_cin0 < pc0) { src = srcA0; coff = ld_cin0; }
    else if (ld_cin0 < pc01) { src = srcA1; coff = ld_cin0 - pc0; }
    else { src = srcA2; coff = ld_cin0 - pc01; }
    c.rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, OOB, 0x00020000);
    c.a_soff = ((K3 && !UP) ? (ld_ky * pWin + ld_kx) * ld + coff : coff) * 2;
    c.b_soff = ld_kt * (BK * 2);
    c.tapbit = 1u << ld_tap; c.ky = ld_ky; c.kx = ld_kx;
    c.lds = stage * STAGE_BYTES;
    if (ld_kt + 1 < kt_end) {      // advance (wave-uniform)
      ++ld_kt;
      if constexpr (K3) {
        ++ld_tap;
        if (++ld_kx == 3) { ld_kx = 0; ++ld_ky; }
        if (ld_tap == ntap) { ld_tap = 0; ld_ky = 0; ld_kx = 0; ld_cin0 += BK; }
      } else {
        ld_cin0 += BK;
      }
    }
    return c;
  };
  // per-lane byte offset of activation pass i for tile c (OOB = this tap is outside the image: reads as zeros)
  auto a_voff = [&](const TileCursor& c, int i) -> int {
    int v;
    if constexpr (UP) v = a_ro[i][c.ky] + a_co[i][c.kx];
    else v = a_off[i];
    if constexpr (K3) v = (a_mask[i] & c.tapbit) ? v : OOB;
    return v;
  };
  auto issue_piece = [&](const TileCursor& c, int q) {     // LDS-DMA: wave-uniform LDS base + lane * 16
    const unsigned row0 = (q < A_PASSES ? q * RPP : BM + (q - A_PASSES) * RPP) + wave_u * 8;
    auto dst = (__attribute__((address_space(3))) void*)(smem + c.lds + row0 * 128);
    if (q < A_PASSES) __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rsrc_a, dst, 16, a_voff(c, q), c.a_soff, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, dst, 16, b_off[q - A_PASSES], c.b_soff, 0, SDMI_W_AUX);
  };
  auto issue_loads = [&](int stage) {
    const TileCursor c = next_tile(stage);
#pragma unroll
    for (int q = 0; q < A_PASSES + B_PASSES; ++q) {
      if constexpr (DMA) {
        issue_piece(c, q);
      } else {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        i32x4 v;
        if (q < A_PASSES) v = __builtin_amdgcn_raw_buffer_load_b128(c.rsrc_a, a_voff(c, q), c.a_soff, 0);
        else v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, b_off[q - A_PASSES], c.b_soff, 0);
        if (q < A_PASSES) regA[q] = __builtin_bit_cast(f16x8, v);
        else regB[q - A_PASSES] = __builtin_bit_cast(f16x8, v);
      }
    }
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

  // fragment reads of k-step ks (16 halves of K) of one stage: TM + TN ds_read_b128
  const int a_lds = (wm * WTM + l31) * 128, b_lds = BM * 128 + (wn * WTN + l31) * 128;
  auto read_frags = [&](int stage, int ks, f16x8 (&a)[TM], f16x8 (&b)[TN]) {
    const unsigned char* st = smem + stage * STAGE_BYTES + (((ks * 2 + lg) ^ rsw) << 4);
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = *(const f16x8*)(st + a_lds + i * 32 * 128);
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = *(const f16x8*)(st + b_lds + j * 32 * 128);
  };
  auto mfma_step = [&](const f16x8 (&a)[TM], const f16x8 (&b)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
  };

  SDMI_STAMP(dbg_t1);
  if constexpr (!DMA) { if (lnf_mine) lnf_finish(p, lnf_pv, &lnf_mean, &lnf_rstd); }
  if constexpr (DMA) {
    // Software pipeline (one raw s_barrier per k-tile, NS - 1 LDS-DMA tiles in flight across it):
    //   * fragments are double buffered in registers: the ds_reads of k-step s + 1 are issued before the MFMAs of
    //     k-step s, so the LDS latency sits under TM * TN MFMAs instead of in front of them (round 1 read, waited,
    //     multiplied -- the compiler reused one fragment register set);
    //   * the barrier that publishes tile t + 1 is taken BEFORE the last k-step of tile t and the first fragments of
    //     tile t + 1 are read right behind it, under the cover of that last k-step's MFMAs;
    //   * the LDS-DMA issues of tile t + NS - 1 (address VALU + one instruction per pass) are spread over the first
    //     KS - 1 k-steps, in the shadow of the MFMAs, instead of in one block in front of them;
    //   * stage (t - 1) % NS is refilled during iteration t: every wave finished (lgkmcnt(0)) all reads of tile t - 1
    //     before it entered the barrier of iteration t - 1.
    constexpr int LPT = A_PASSES + B_PASSES;       // DMA instructions per thread per k-tile
    constexpr int KS = BK / 16;
    // pipeline unit = G k-steps: at least 4 MFMAs (128 cycles) of cover for the unit's TM + TN fragment reads
    constexpr int G = (TM * TN >= 4) ? 1 : 2;
    constexpr int U = KS / G;                      // units per k-tile (4 or 2)
    constexpr int PPU = (LPT + U - 2) / (U - 1);   // DMA pieces issued in each of the first U - 1 units
    constexpr int MPU = G * TM * TN;               // MFMAs per unit
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue_loads(s);
    // LayerNorm folded into this GEMM: fold the row statistics requested at the top.  The explicit drain: register loads and
    // LDS-DMA do not retire through one in-order queue (profiles/gn_fold_r03.txt), so a counted wait across both is not sound
    if (p.lnf_part) { wait_vmcnt<0>(); if (lnf_mine) lnf_finish(p, lnf_pv, &lnf_mean, &lnf_rstd); }
    wait_vmcnt<LPT*(NS - 2)>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    f16x8 fa[2][G][TM], fb[2][G][TN];
#pragma unroll
    for (int g = 0; g < G; ++g) read_frags(0, g, fa[0][g], fb[0][g]);
    int cur = 0, nxt = NS - 1;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const TileCursor c = next_tile(nxt);
      const int cur1 = (cur + 1 == NS) ? 0 : cur + 1;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (u + 1 < U) {
#pragma unroll
          for (int g = 0; g < G; ++g) read_frags(cur, (u + 1) * G + g, fa[(u + 1) & 1][g], fb[(u + 1) & 1][g]);
#pragma unroll
          for (int q = u * PPU; q < (u + 1) * PPU && q < LPT; ++q) issue_piece(c, q);
        } else {
          wait_vmcnt<LPT*(NS - 2)>();               // this wave's share of tile kt + 1 has landed
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... and everybody's; tile kt is fully read
#pragma unroll
          for (int g = 0; g < G; ++g) read_frags(cur1, g, fa[0][g], fb[0][g]);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) mfma_step(fa[u & 1][g], fb[u & 1][g]);
        // pin the issue order of this unit: all fragment reads of the NEXT unit first (they land under this unit's
        // MFMAs), then MFMAs with one LDS-DMA issue in each gap (masks: 0x100 DS read, 0x008 MFMA, 0x010 VMEM)
        __builtin_amdgcn_sched_group_barrier(0x100, G * (TM + TN), 0);
#pragma unroll
        for (int e = 0; e < MPU; ++e) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (u + 1 < U && e < PPU && u * PPU + e < LPT) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
      }
      cur = cur1;
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
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        f16x8 a[TM], b[TN];
        read_frags(cur, ks, a, b);
        mfma_step(a, b);
      }
      if (more) commit_regs(cur ^ 1);
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------------
  SDMI_STAMP(dbg_t2);
  igemm_epilogue<BM, BN, WARPS_M, WARPS_N, NS * STAGE_BYTES>(p, acc, m0, n0, split, tile_m, tile_n, smem, lnf_mean, lnf_rstd);
#ifdef SDMI_IGEMM_TIMING
  if (p.dbg_times && tid == 0) {        // (where a workgroup's time goes; blocks that return early in the epilogue are not stamped)
    long long* d = p.dbg_times + 6 * (size_t)blockIdx.x;      // d[3] = after the output stores (written inside the epilogue)
    d[0] = dbg_t0; d[1] = dbg_t1; d[2] = dbg_t2; d[4] = (long long)__builtin_readcyclecounter();
  }
#endif
#endif  // __HIP_DEVICE_COMPILE__
}

template <int BM, int BN, int WARPS_M, int WARPS_N, int NS>
int launch_cfg(const IGemmParams& p, bool dma, int splitk, hipStream_t stream) {
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
  std::string pname = std::string("igemm_") + std::to_string(BM) + "x" + std::to_string(BN) + "w" +
                      std::to_string(WARPS_M * WARPS_N) + "s" + std::to_string(NS);
  if (by_shape && prof_enabled())
    pname += "_M" + std::to_string(p.M) + "_N" + std::to_string(p.N) + "_K" + std::to_string(p.K) + "_k" +
             std::to_string(p.ksize) + "_m" + std::to_string(p.mode) + "_s" + std::to_string(nsplit);
  const double src_pix = (double)p.B * p.Hin * p.Win;
  const double out_b = (p.out_f32 ? 4.0 : 0.0) + ((p.out_f16 || p.mode != EPI_PLAIN) ? 2.0 : 0.0);
  const double n_out = p.mode == EPI_GEGLU ? p.N / 2.0 : (double)p.N;
  // FLOPs: algorithmic (2 x MACs of the reference op) and executed (the K-concatenated split-fp16 1x1 convs
  // run three passes); bytes likewise count the reference op's operands once (one fp16 activation read, one weight read)
  const int k_alg = p.k_alg > 0 ? p.k_alg : p.K;
  const double cin_alg = p.k_alg > 0 ? (double)p.k_alg : (double)(p.c0 + p.c1 + p.c2);
  ProfScope ps(pname.c_str(), 2.0 * p.M * (double)p.N * k_alg,
               src_pix * cin_alg * 2.0 + (double)p.N * k_alg * 2.0 + (double)p.M * n_out * out_b +
                   (p.residual ? (double)p.M * p.N * 4.0 : 0.0),
               stream, 2.0 * p.M * (double)p.N * p.K);
  const int kind = p.ksize == 1 ? KIND_1X1 : (p.up ? KIND_3X3_UP : KIND_3X3);
#define SDMI_LAUNCH_KIND(K_)                                                                                        \
  do {                                                                                                              \
    if (dma) SDMI_LAUNCH((igemm_kernel<BM, BN, WARPS_M, WARPS_N, true, NS, K_>), grid, block, 0, stream, q,   \
                                tiles_m, tiles_n, kt_per_split);                                                    \
    else SDMI_LAUNCH((igemm_kernel<BM, BN, WARPS_M, WARPS_N, false, 2, K_>), grid, block, 0, stream, q,       \
                            tiles_m, tiles_n, kt_per_split);                                                        \
  } while (0)
  if (kind == KIND_1X1) SDMI_LAUNCH_KIND(KIND_1X1);
  else if (kind == KIND_3X3) SDMI_LAUNCH_KIND(KIND_3X3);
  else SDMI_LAUNCH_KIND(KIND_3X3_UP);
#undef SDMI_LAUNCH_KIND
  SDMI_HIP_OK(hipGetLastError());
  ps.end();
  if (nsplit > 1 && !q.splitk_fused) return launch_splitk_reduce(q, nsplit, stream);     // (+ the LayerNorm launch when q.ln_out)
  if (q.ln_out) return launch_layernorm(q.out_f32, q.ln_gamma, q.ln_beta, q.ln_out, q.M, q.N, q.ln_eps, stream);
  return 0;
}

}  // namespace
}  // namespace sdmi
