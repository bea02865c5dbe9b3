// Implicit-GEMM convolution / linear kernel with a wave count that is not a power of two (gfx950 / MI355X): the 64 x 160 tile.
//
// Same operation, operand layouts and epilogue as igemm.hip's igemm_kernel (F.conv2d / F.linear call sites of the UNet hot path:
// ldm/modules/diffusionmodules/openaimodel.py:204,230,241 and ldm/modules/attention.py:161-168,58,233-248); what differs is the
// TILE GEOMETRY.  Every channel count of SD v1 is a multiple of 160 (320 / 640 / 960 / 1280 / 1920 / 2560 / 3840 / 5120 / 10240), and the
// 64x64-pixel level has M = 8192 rows: 8192 / 64 x 320 / 160 = 256 tiles = ONE workgroup per CU of the MI355X, where the 64 x 64
// tile gives 640 workgroups that sit three-and-two on the 256 CUs (the launch lasts as long as the three-deep CUs) and moves
// 128 B of operands through the vector-memory path per 32 FLOP/B instead of 46 FLOP/B (profiles/igemm_phase_timing_r02.txt: the
// k-loop of the small tiles runs at the ~40 B/clk/CU LDS-DMA rate, not at an MFMA or LDS limit -- flops per byte is the lever).
// 160 = 5 x 32 columns: five waves side by side, each 64 rows x 32 columns (2 x 1 MFMA tiles, v_mfma_f32_32x32x16_f16).
//
// With 320 threads the generic kernel's "thread t loads row t / 8 of every pass" no longer tiles the operands (40 rows per pass,
// 64 + 160 rows).  Here the unit is the OCTET: 8 consecutive rows x 128 B = one LDS-DMA wave instruction (lane -> row lane / 8, 16-byte
// chunk lane % 8).  The (A | W) rows of a k-tile are 28 octets; pass i hands octet i * 5 + wave to each wave (6 passes; the two
// surplus issues repeat the last octet: same bytes, and every wave issues the same number of DMA instructions per k-tile, so one
// counted vmcnt literal serves all).  An octet lies entirely in A or entirely in W (64 % 8 == 0): the descriptor is a wave-uniform
// choice.  The 16-byte chunks are XOR-swizzled with (row >> 1) & 7 on the source side, as in the generic kernel.
#include "igemm_dev.h"

#ifdef SDMI_EXPERIMENTS      // (lost its in-situ tuning A/B in round 3: never chosen; kept as the record of what was measured)
namespace sdmi {
namespace {

template <int BM, int BN, int NW, int NS, int KIND>
__global__ void __launch_bounds__(NW * 64) igemm5_kernel(const IGemmParams p, const int tiles_m, const int tiles_n,
                                                         const int kt_per_split) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(KIND == KIND_1X1 || KIND == KIND_3X3, "1x1 / 3x3 gathers");
  constexpr bool K3 = KIND == KIND_3X3;
  constexpr int NT = NW * 64;
  constexpr int WTM = BM, WTN = BN / NW;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int OCT_A = BM / 8, OCT = (BM + BN) / 8;              // octets of A rows, of all rows
  constexpr int NPASS = (OCT + NW - 1) / NW;                       // DMA instructions per wave per k-tile
  constexpr int STAGE_BYTES = OCT * 8 * 128;
  static_assert(BM % 32 == 0 && BN % (32 * NW) == 0 && TM >= 1 && TN >= 1 && BM % 16 == 0, "tile / wave shape");
  static_assert(NS >= 3 && NS * STAGE_BYTES <= 160 * 1024, "LDS budget");

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
  float2 lnf_pv[LNF_MAXP];                       // LayerNorm of the A rows folded into this GEMM (igemm_dev.h lnf_request)
  float lnf_mean = 0.f, lnf_rstd = 1.f;
  const bool lnf_mine = p.lnf_part != nullptr && tid < BM;
  if (lnf_mine) lnf_request(p, min(m0 + tid, p.M - 1), lnf_pv);
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int cpos = lane & 7, r8l = lane >> 3;    // 16-byte chunk inside the LDS row, row inside the octet

  // ---- per-pass metadata of this lane: the source byte offset of its 16 bytes (k-tile 0, tap (pad, pad)) and, for 3x3
  // convolutions, the 9-bit mask of the taps inside the image (all ones for weight rows) ----
  const int HWout = p.Hout * p.Wout;
  const int pad = K3 ? p.pad : 0;
  constexpr int ntap = K3 ? 9 : 1;
  const int ld = p.lda0;
  int voff[NPASS];
  unsigned vmask[K3 ? NPASS : 1];
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    const int o = min(i * NW + wave_u, OCT - 1);                   // (wave-uniform)
    const int r = o * 8 + r8l;                                     // row in the combined (A | W) space
    const int gch = cpos ^ ((r >> 1) & 7);                         // (BM % 16 == 0: the swizzle of a W row is that of r - BM)
    if (o < OCT_A) {
      const int m = min(m0 + r, p.M - 1);
      const int b = fast_div_hw(m, p.magic_hw);
      const int rem = m - b * HWout;
      const int oy = fast_div(rem, p.magic_w), ox = rem - oy * p.Wout;
      const int cy = oy * p.stride, cx = ox * p.stride;
      voff[i] = (((b * p.Hin + cy) * p.Win + cx) * ld + gch * 8) * 2;
      if constexpr (K3) {
        unsigned mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = cy + t / 3 - pad, ix = cx + t % 3 - pad;
          if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) mk |= 1u << t;
        }
        vmask[i] = mk;
      }
    } else {
      const int n = min(n0 + r - BM, p.N - 1);
      voff[i] = (n * p.K + gch * 8) * 2;
      if constexpr (K3) vmask[i] = 0x1ffu;
    }
  }

  int ld_kt = kt_begin;
  int ld_tap = K3 ? kt_begin % ntap : 0;
  int ld_cin0 = (kt_begin / ntap) * BK;
  int ld_ky = K3 ? ld_tap / 3 : 0, ld_kx = K3 ? ld_tap - 3 * (ld_tap / 3) : 0;
  constexpr int OOB = (int)0x80000000;
  const long long a_shift = K3 ? (long long)(pad * p.Win + pad) * ld * 2 : 0;
  const char* const srcA0 = (const char*)p.a0 - a_shift; const char* const srcA1 = (const char*)p.a1 - a_shift;
  const char* const srcA2 = (const char*)p.a2 - a_shift;
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, OOB, 0x00020000);
  const int pc0 = p.c0, pc01 = p.c0 + p.c1, pWin = p.Win;

  struct TileCursor { __amdgpu_buffer_rsrc_t rsrc_a; int a_soff, b_soff; unsigned tapbit; unsigned lds; };
  auto next_tile = [&](int stage) {
    TileCursor c;
    const char* src; int coff;
    if (ld_cin0 < pc0) { src = srcA0; coff = ld_cin0; }
    else if (ld_cin0 < pc01) { src = srcA1; coff = ld_cin0 - pc0; }
    else { src = srcA2; coff = ld_cin0 - pc01; }
    c.rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, OOB, 0x00020000);
    c.a_soff = (K3 ? (ld_ky * pWin + ld_kx) * ld + coff : coff) * 2;
    c.b_soff = ld_kt * (BK * 2);
    c.tapbit = 1u << ld_tap;
    c.lds = stage * STAGE_BYTES;
    if (ld_kt + 1 < kt_end) {
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
  auto issue_piece = [&](const TileCursor& c, int i) {
    const int o = min(i * NW + wave_u, OCT - 1);
    auto dst = (__attribute__((address_space(3))) void*)(smem + c.lds + o * (8 * 128));
    if (o < OCT_A) {                                               // (wave-uniform)
      int v = voff[i];
      if constexpr (K3) v = (vmask[i] & c.tapbit) ? v : OOB;       // a tap outside the image reads zeros
      __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rsrc_a, dst, 16, v, c.a_soff, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, dst, 16, voff[i], c.b_soff, 0, 0);
    }
  };
  auto issue_loads = [&](int stage) {
    const TileCursor c = next_tile(stage);
#pragma unroll
    for (int i = 0; i < NPASS; ++i) issue_piece(c, i);
  };

  const int l31 = lane & 31, lg = lane >> 5;
  const int rsw = (l31 >> 1) & 7;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int a_lds = l31 * 128, b_lds = BM * 128 + (wave * WTN + l31) * 128;
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
  // ---- software pipeline: igemm_kernel's (NS - 1 tiles in flight across one raw barrier per k-tile, register double-buffered
  // fragments, DMA issues spread over the first units of a k-tile) ----
  constexpr int LPT = NPASS;
  constexpr int KS = BK / 16;
  constexpr int G = (TM * TN >= 4) ? 1 : 2;
  constexpr int U = KS / G;
  constexpr int PPU = (LPT + U - 2) / (U - 1);
  constexpr int MPU = G * TM * TN;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue_loads(s);
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
        wait_vmcnt<LPT*(NS - 2)>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int g = 0; g < G; ++g) read_frags(cur1, g, fa[0][g], fb[0][g]);
      }
#pragma unroll
      for (int g = 0; g < G; ++g) mfma_step(fa[u & 1][g], fb[u & 1][g]);
    }
    cur = cur1;
    nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
  }
  wait_vmcnt<0>();

  SDMI_STAMP(dbg_t2);
  igemm_epilogue<BM, BN, 1, NW, NS * STAGE_BYTES>(p, acc, m0, n0, split, tile_m, tile_n, smem, lnf_mean, lnf_rstd);
#ifdef SDMI_IGEMM_TIMING
  if (p.dbg_times && tid == 0) {
    long long* d = p.dbg_times + 6 * (size_t)blockIdx.x;
    d[0] = dbg_t0; d[1] = dbg_t1; d[2] = dbg_t2; d[4] = (long long)__builtin_readcyclecounter();
  }
#endif
#endif  // __HIP_DEVICE_COMPILE__
}

template <int BM, int BN, int NW, int NS>
int launch_cfg5(const IGemmParams& p, int splitk, hipStream_t stream) {
  SDMI_CHECK(!p.up && !p.split16 && !p.xf0, "igemm5: plain 1x1 / 3x3 gathers only");
  const int tiles_m = cdiv(p.M, BM), tiles_n = cdiv(p.N, BN);
  const int nkt = p.K / BK;
  const int kt_per_split = cdiv(nkt, splitk);
  const int nsplit = cdiv(nkt, kt_per_split);
  IGemmParams q = p;
  q.splitk = nsplit;
  q.tile_n_fastest = tile_order_n_fastest(p);
  q.splitk_fused = nsplit > 1 && splitk_fusable(p, BM, BN);
  q.epi_vec = epi_vec_ok(p);
  SDMI_CHECK(splitk_ws_need(p, BM, BN, nsplit) <= p.splitk_ws_floats, "split-K workspace too small");
  SDMI_CHECK((int64_t)p.M < (int64_t)65536 * p.Hout * p.Wout, "fast_div_hw: at most 65535 samples per launch");
  q.magic_hw = div_magic_hw(p.Hout * p.Wout);
  q.magic_w = div_magic(p.Wout);
  for (int t = 0; t < p.gn_n; ++t) q.gn_magic[t] = div_magic(p.gn_cpg[t]);
  dim3 grid(tiles_m * tiles_n * nsplit), block(NW * 64);
  static const int by_shape = env_int("SDMI_PROF_SHAPES", 0);
  std::string pname = std::string("igemm_") + std::to_string(BM) + "x" + std::to_string(BN) + "w" + std::to_string(NW) + "s" + std::to_string(NS);
  if (by_shape && prof_enabled())
    pname += "_M" + std::to_string(p.M) + "_N" + std::to_string(p.N) + "_K" + std::to_string(p.K) + "_k" + std::to_string(p.ksize) + "_m" +
             std::to_string(p.mode) + "_s" + std::to_string(nsplit);
  const double src_pix = (double)p.B * p.Hin * p.Win;
  const double out_b = (p.out_f32 ? 4.0 : 0.0) + ((p.out_f16 || p.mode != EPI_PLAIN) ? 2.0 : 0.0);
  const double n_out = p.mode == EPI_GEGLU ? p.N / 2.0 : (double)p.N;
  const int k_alg = p.k_alg > 0 ? p.k_alg : p.K;
  const double cin_alg = p.k_alg > 0 ? (double)p.k_alg : (double)(p.c0 + p.c1 + p.c2);
  ProfScope ps(pname.c_str(), 2.0 * p.M * (double)p.N * k_alg,
               src_pix * cin_alg * 2.0 + (double)p.N * k_alg * 2.0 + (double)p.M * n_out * out_b + (p.residual ? (double)p.M * p.N * 4.0 : 0.0),
               stream, 2.0 * p.M * (double)p.N * p.K);
  if (p.ksize == 1) SDMI_LAUNCH((igemm5_kernel<BM, BN, NW, NS, KIND_1X1>), grid, block, 0, stream, q, tiles_m, tiles_n, kt_per_split);
  else SDMI_LAUNCH((igemm5_kernel<BM, BN, NW, NS, KIND_3X3>), grid, block, 0, stream, q, tiles_m, tiles_n, kt_per_split);
  SDMI_HIP_OK(hipGetLastError());
  ps.end();
  if (nsplit > 1 && !q.splitk_fused) return launch_splitk_reduce(q, nsplit, stream);
  if (q.ln_out) return launch_layernorm(q.out_f32, q.ln_gamma, q.ln_beta, q.ln_out, q.M, q.N, q.ln_eps, stream);
  return 0;
}

}  // namespace

// tile id 22 of the table in igemm.hip (kTiles): 64 x 160, five waves, 5 LDS-DMA stages (140 KB)
int launch_igemm5_tile(int tile, const IGemmParams& p, int splitk, hipStream_t stream) {
  switch (tile) {
    case 22: return launch_cfg5<64, 160, 5, 5>(p, splitk, stream);
    default: return fail("not a five-wave tile id");
  }
}

}  // namespace sdmi
#else
namespace sdmi {
int launch_igemm5_tile(int, const IGemmParams&, int, hipStream_t) {
  return fail("tile 22 (five-wave 64 x 160 tile, igemm5.hip) is an experiment: build with SDMI_CXXFLAGS=-DSDMI_EXPERIMENTS");
}
}  // namespace sdmi
#endif
