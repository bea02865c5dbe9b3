// Halo-staged 3x3 convolutions for gfx950 (MI355X): ResBlock in_layers / out_layers convs
// (ldm/modules/diffusionmodules/openaimodel.py:201-204,225-231) -- see the kernel comments.  Split from igemm.hip (round 3) so that
// the two kernel families compile in parallel; the shared device code (epilogue, counted waits) is igemm_dev.h.
#include <type_traits>
#include <utility>

#include "igemm_dev.h"

namespace sdmi {
namespace {

// ---- halo-staged 3x3 convolution (stride 1, pad 1) --------------------------------------------------------------------
// ResBlock in_layers / out_layers convs (openaimodel.py:204,230) at the 64x64 .. 16x16 levels.  The generic kernel above
// streams the A operand once per TAP: the nine shifted copies of the same pixels are nine separate k-tiles, so a 3x3 conv
// moves 9x its activation bytes through the CU's vector-memory path -- and that path (64 B/clk/CU), not MFMA issue, is
// what bounds these 15 GFLOP launches.  Here a block owns TH = BM / W whole image rows; for every 64-channel chunk it
// stages the (TH + 2) x (W + 2) input HALO once (LDS-DMA; out-of-image pixels are out-of-range buffer offsets and read as
// zeros) and all nine taps read their A fragments from it at a row offset -- only the weights stream per tap.
// Bytes through the vector-memory path per chunk, 256 x 64 tile: 51 KB halo + 72 KB weights vs 9 x 40 KB = 360 KB.
//   LDS: [halo buffer 0 | halo buffer 1 | NS weight stages]; halo rows are pixels (128 B = 64 channels), XOR-swizzled by
//   the absolute LDS row exactly like the generic tiles, so fragment reads at any row offset stay conflict free.
//   The nine taps are unrolled: every DMA issue and every counted vmcnt wait is static.
template <int BM, int BN, int WARPS_M, int WARPS_N, int NS>
__global__ void __launch_bounds__(WARPS_M* WARPS_N * 64) conv3halo_kernel(const IGemmParams p, const int tiles_m,
                                                                           const int tiles_n, const int chunks_per_split) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NT = WARPS_M * WARPS_N * 64;
  constexpr int RPP = NT / 8;
  constexpr int PB = BN / RPP;                         // weight DMA pieces per thread per tap
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int HPMAX = (BM / 64 + 2) * 66;            // halo pixels at W = 64 (the largest for W in {16, 32, 64})
  constexpr int AHP = (HPMAX + RPP - 1) / RPP;         // halo DMA pieces per thread per chunk
  constexpr int HALO_BYTES = AHP * RPP * 128;
  constexpr int BSTAGE = BN * 128;
  constexpr int LDS_BYTES = 2 * HALO_BYTES + NS * BSTAGE;
  // The weight stream is what needs depth: every block reads every (chunk, tap) weight tile exactly once, and the blocks
  // of an XCD walk the taps in step, so most weight tiles are first touches of that XCD's L2 (HBM / Infinity-Cache
  // latency, ~1 us).  NS weight stages = NS - 1 taps of look-ahead; the halo of the NEXT chunk must be complete NS - 2
  // taps before the chunk switch, so it is issued at taps 0 .. LASTA.
  constexpr int LASTA = 10 - NS;
  constexpr int PA = (AHP + LASTA) / (LASTA + 1);      // halo pieces of the NEXT chunk issued per tap (taps 0 .. LASTA)
  constexpr int KS = BK / 16;
  constexpr int G = (TM * TN >= 4) ? 1 : 2;            // k-steps per pipeline unit (>= 4 MFMAs of cover)
  constexpr int U = KS / G;
  constexpr int MPU = G * TM * TN;
  static_assert(PB >= 1 && TM >= 1 && TN >= 1 && RPP % 16 == 0 && NS >= 2 && NS <= 9, "tile/wave shape");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

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
  const int nch = (p.c0 + p.c1 + p.c2) / BK;
  const int c_begin = split * chunks_per_split;
  const int c_end = min(nch, c_begin + chunks_per_split);
  if (c_begin >= c_end) return;

  const int tid = threadIdx.x;
  SDMI_STAMP(dbg_t0);
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int cpos = tid & 7, lrow = tid >> 3;
  const int gch = cpos ^ ((lrow >> 1) & 7);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wm = wave / WARPS_N, wn = wave - wm * WARPS_N;
  const int l31 = lane & 31, lg = lane >> 5;
  const int rsw = (l31 >> 1) & 7;

  // tile geometry: BM <= H*W: TH = BM / W rows of one image; BM > H*W: BM / (H*W) whole images (each with its own halo)
  const int W = p.Wout, H = p.Hout, HW = H * W, W2 = W + 2, ld = p.lda0;
  const int bimg = m0 / HW;
  const int THI = p.halo_thi;                     // output rows per image inside the tile
  const int HPI = (THI + 2) * W2;                 // halo pixels per image
  const int y0 = p.halo_ipt > 1 ? 0 : (m0 - bimg * HW) >> p.log2w;
  const int HP = p.halo_ipt * HPI;
  constexpr int OOB = (int)0x80000000;

  // per-lane source byte offset of every halo piece (constant over the chunks: the chunk moves the scalar offset)
  int hvoff[AHP];
#pragma unroll
  for (int q = 0; q < AHP; ++q) {
    const int hp = q * RPP + lrow;
    const int ip = fast_div(hp, p.magic_hpi), hr = hp - ip * HPI;
    const int hy = fast_div(hr, p.magic_w2), hx = hr - hy * W2;
    const int y = y0 + hy - 1, x = hx - 1;
    const bool valid = hp < HP && y >= 0 && y < H && x >= 0 && x < W;
    hvoff[q] = valid ? ((((bimg + ip) * H + y) * W + x) * ld + gch * 8) * 2 : OOB;
  }
  int b_off[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int n = min(n0 + i * RPP + lrow, p.N - 1);
    b_off[i] = (n * p.K + gch * 8) * 2;
  }
  // halo row of tap (0, 0) for the rows of this lane's MFMA tiles
  int hr0[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int ml = wm * WTM + i * 32 + l31;
    const int ip = ml >> p.log2_tpi, mr = ml & ((1 << p.log2_tpi) - 1);     // image inside the tile, pixel inside the image part
    hr0[i] = ip * HPI + (mr >> p.log2w) * W2 + (mr & (W - 1));
  }
  const int b_lds = 2 * HALO_BYTES + (wn * WTN + l31) * 128;

  const char* const srcA0 = (const char*)p.a0; const char* const srcA1 = (const char*)p.a1;
  const char* const srcA2 = (const char*)p.a2;
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, OOB, 0x00020000);
  const int pc0 = p.c0, pc01 = p.c0 + p.c1;
  struct ChunkSrc { __amdgpu_buffer_rsrc_t rsrc; int soff; };
  auto chunk_src = [&](int c) {
    const int cin0 = c * BK;
    const char* src; int coff;
    if (cin0 < pc0) { src = srcA0; coff = cin0; }
    else if (cin0 < pc01) { src = srcA1; coff = cin0 - pc0; }
    else { src = srcA2; coff = cin0 - pc01; }
    ChunkSrc r; r.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, OOB, 0x00020000); r.soff = coff * 2;
    return r;
  };
  auto issue_halo = [&](const ChunkSrc& cs, int hbuf, int q) {
    auto dst = (__attribute__((address_space(3))) void*)(smem + hbuf * HALO_BYTES + (q * RPP + wave_u * 8) * 128);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(cs.rsrc, dst, 16, hvoff[q], cs.soff, 0, 0);
  };
  auto issue_b = [&](int kt, int stage, int q) {
    auto dst = (__attribute__((address_space(3))) void*)(smem + 2 * HALO_BYTES + stage * BSTAGE + (q * RPP + wave_u * 8) * 128);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, dst, 16, b_off[q], kt * (BK * 2), 0, SDMI_W_AUX);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // LDS byte address (k-step 0) of the A fragment rows of MFMA tile i for tap (ky, kx) in halo buffer hbuf;
  // k-step ks is that address ^ (ks << 5) (the 16-byte chunk index is (2 ks + lg) ^ swizzle(row))
  auto a_base = [&](int i, int ky, int kx, int hbuf) -> int {
    const int rowt = hr0[i] + ky * W2 + kx;
    return hbuf * HALO_BYTES + ((rowt << 7) | ((lg ^ ((rowt >> 1) & 7)) << 4));
  };
  auto read_frags = [&](const int (&ab)[TM], int bstage, int ks, f16x8 (&a)[TM], f16x8 (&b)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = *(const f16x8*)(smem + (ab[i] ^ (ks << 5)));
    const unsigned char* st = smem + bstage * BSTAGE + b_lds + (((ks * 2 + lg) ^ rsw) << 4);
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = *(const f16x8*)(st + j * 32 * 128);
  };
  auto mfma_step = [&](const f16x8 (&a)[TM], const f16x8 (&b)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
  };

  // DMA pieces this thread issues at tap t (t taken mod 9): halo pieces [a_lo, a_hi) of the next chunk, then PB weight
  // pieces.  What may still be in flight when tile kt + 1 is needed = everything issued in the last NS - 2 taps.
  auto a_lo = [](int t) { return t <= LASTA ? (t * PA < AHP ? t * PA : AHP) : AHP; };
  auto a_hi = [](int t) { return t <= LASTA ? ((t + 1) * PA < AHP ? (t + 1) * PA : AHP) : AHP; };
  auto in_flight_ok = [&](int t) {
    int n = 0;
    for (int d = 0; d < NS - 2; ++d) { const int tt = (t - d + 18) % 9; n += a_hi(tt) - a_lo(tt) + PB; }
    return n;
  };

  // ---- prologue: the first halo, then the last NS - 1 taps of a virtual previous chunk (their halo pieces re-issue
  // piece 0: same bytes, same issue counts as the steady state, so the vmcnt literals hold from the first tap on) ----
  const int kt_first = c_begin * 9, kt_last = c_end * 9 - 1;
  {
    const ChunkSrc cs = chunk_src(c_begin);
#pragma unroll
    for (int q = 0; q < AHP; ++q) issue_halo(cs, 0, q);
#pragma unroll
    for (int s2 = 0; s2 < NS - 1; ++s2) {
      const int vt = 9 - (NS - 1) + s2;
#pragma unroll
      for (int e = a_lo(vt); e < a_hi(vt); ++e) issue_halo(cs, 0, 0);
#pragma unroll
      for (int q = 0; q < PB; ++q) issue_b(min(kt_first + s2, kt_last), s2, q);
    }
  }
  wait_vmcnt_n(in_flight_ok(8));
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  SDMI_STAMP(dbg_t1);
  f16x8 fa[2][G][TM], fb[2][G][TN];
  int ab[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) ab[i] = a_base(i, 0, 0, 0);
#pragma unroll
  for (int g = 0; g < G; ++g) read_frags(ab, 0, g, fa[0][g], fb[0][g]);

  int cur = 0, nxt = NS - 1, hb = 0;
  for (int c = c_begin; c < c_end; ++c) {
    // the next chunk's halo streams in during taps 0..7 (the last chunk of the split reloads itself: same issue counts,
    // so every vmcnt literal below stays valid)
    const ChunkSrc csn = chunk_src(min(c + 1, c_end - 1));
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kt = c * 9 + tap;
      const int tap1 = tap == 8 ? 0 : tap + 1;
      const int hb1 = tap == 8 ? (hb ^ 1) : hb;
      const int cur1 = (cur + 1 == NS) ? 0 : cur + 1;
      int ab1[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) ab1[i] = a_base(i, tap1 / 3, tap1 % 3, hb1);
      const int alo = a_lo(tap);
      const int na = a_hi(tap) - alo;                   // compile-time after unrolling
      const int npieces = na + PB;
      const int ppu = (npieces + U - 2) / (U - 1);
      const int bt = min(kt + NS - 1, kt_last);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (u + 1 < U) {
#pragma unroll
          for (int g = 0; g < G; ++g) read_frags(ab, cur, (u + 1) * G + g, fa[(u + 1) & 1][g], fb[(u + 1) & 1][g]);
#pragma unroll
          for (int e = u * ppu; e < (u + 1) * ppu && e < npieces; ++e) {
            if (e < na) issue_halo(csn, hb ^ 1, alo + e);         // halo pieces first: older than this tap's weights
            else issue_b(bt, nxt, e - na);
          }
        } else {
          // allowed in flight: what the last NS - 2 taps issued.  Weight tile kt + 1 -- and, at tap 8, the whole next halo
          // (issued at taps <= LASTA) -- has landed for this wave; the barrier makes it everybody's, and tells everybody
          // this tile's LDS reads are done
          wait_vmcnt_n(in_flight_ok(tap));
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
          for (int g = 0; g < G; ++g) read_frags(ab1, cur1, g, fa[0][g], fb[0][g]);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) mfma_step(fa[u & 1][g], fb[u & 1][g]);
        __builtin_amdgcn_sched_group_barrier(0x100, G * (TM + TN), 0);
#pragma unroll
        for (int e = 0; e < MPU; ++e) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (u + 1 < U && e < ppu && u * ppu + e < npieces) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) ab[i] = ab1[i];
      cur = cur1;
      nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
      hb = hb1;
    }
  }
  wait_vmcnt<0>();
  SDMI_STAMP(dbg_t2);
  igemm_epilogue<BM, BN, WARPS_M, WARPS_N, LDS_BYTES>(p, acc, m0, n0, split, tile_m, tile_n, smem);
#ifdef SDMI_IGEMM_TIMING
  if (p.dbg_times && tid == 0) {
    long long* d = p.dbg_times + 6 * (size_t)blockIdx.x;
    d[0] = dbg_t0; d[1] = dbg_t1; d[2] = dbg_t2; d[4] = (long long)__builtin_readcyclecounter();
  }
#endif
#endif  // __HIP_DEVICE_COMPILE__
}


#ifdef SDMI_EXPERIMENTS      // (the GroupNorm-folding halo conv: bit-identical, measured slower at every site in round 3, profiles/gn_fold_r03.txt)
// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// Schedule of the halo pieces of conv3halo_gn_kernel inside the nine taps of a channel chunk (all compile-time):
// piece q is loaded at tap LT(q) (taps 0 .. 6; with 8 or 9 pieces two at taps 0 and 1) and converted LAG taps later.  Order of
// the vector-memory issues inside a tap: [register loads of the tap's pieces (gamma / beta in front of piece 0)] [PB weight pieces].
//
// Counted waits, per DESTINATION CLASS.  The kernel has two kinds of loads in flight: LDS-DMA (weights) and loads into registers
// (the fp32 halo).  Round 3 measured (profiles/gn_fold_r03.txt) that ONE in-order queue across both is not what the hardware
// retires: with "everything younger than X may be outstanding" as the vmcnt literal, weight tiles were read before they had
// landed whenever the activations were L2-hot and the weights HBM-cold (inside a UNet call; never in a stand-alone repetition),
// and draining the queue at every wait (SDMI_GN_SAFE=1) cured it.  The literals below therefore count only the younger
// operations of the SAME class: total outstanding <= n implies outstanding-of-that-class <= n, and within a class retirement is
// in order (the LDS-DMA-only kernels rely on exactly that).  Also safe under a single in-order queue (the literal only shrinks).
template <int AHP, int PB, int NS, int LAG>
struct HaloGnSched {
  static constexpr int LT(int q) { return AHP <= 7 ? q : (q < 4 ? q / 2 : q - 2); }
  static constexpr int loads_at(int t) {               // register loads issued at tap t (0 .. 8)
    int n = 0;
    for (int q = 0; q < AHP; ++q) n += (LT(q) == t) ? 2 : 0;
    return n + (t == 0 ? 4 : 0);
  }
  static constexpr int vmem_at(int t) { return loads_at(((t % 9) + 9) % 9) + PB; }
  // register loads that may be outstanding when piece q is converted at tap LT(q) + LAG: the later pieces of its own tap, the
  // loads of the taps in between and of the conversion tap (issued in front of the conversions)
  static constexpr int after_piece(int q) {
    int n = 0;
    for (int q2 = q + 1; q2 < AHP; ++q2) n += (LT(q2) == LT(q)) ? 2 : 0;
    for (int t = LT(q) + 1; t <= LT(q) + LAG; ++t) n += loads_at(t);
    return n;
  }
  // LDS-DMA pieces that may be outstanding when weight tile kt + 1 is needed at the end of tap t: those of the last NS - 2 taps
  // (that tile's pieces were the DMA issues of tap t - (NS - 2)); the last chunk's place-holder issues are DMA pieces too and
  // only make the real count larger than the literal
  static constexpr int in_flight_ok(int) { return PB * (NS - 2); }
};

// ---- halo-staged 3x3 convolution with GroupNorm(32) + SiLU folded into the staging --------------------------------------
// ResBlock._forward, openaimodel.py:263-266,273-275: h = conv3x3(SiLU(GroupNorm32(x))) -- `in_layers` / `out_layers`
// (openaimodel.py:201-204,225-231; GroupNorm32 = util.py:199-216: fp32 statistics, eps 1e-5).  The stand-alone path runs the
// normalisation as its own launch (norm.hip gn_apply_kernel: fp32 stream in, fp16 operand out) and this convolution reads that
// operand by LDS-DMA.  Here the convolution reads the fp32 stream ITSELF: every thread loads 8 channels of a halo pixel into
// registers, applies (x - mean) * rstd * gamma + beta and SiLU -- the same arithmetic, gn_apply_elem, so the fp16 operand is
// bit-identical -- and writes the fp16 row into the halo tile in LDS, from where the nine taps read their A fragments exactly
// as in conv3halo_kernel.  One launch and one 16-bit round trip of the activation through HBM less per convolution; the
// price is that every N-tile (and every halo overlap) repeats the normalisation of its input pixels.
//   * statistics: complete before the launch (emitted by the producers' epilogues or the statistics kernel); every workgroup
//     folds the (sample, group) accumulators of the images it touches into an LDS table {mean, rstd} once.
//   * pipeline: the next chunk's halo is fetched in batches of BP pieces (2 x 16 bytes per thread and piece) at fixed taps,
//     consumed (normalise + ds_write_b128) a few taps later -- the latency of the fp32 loads runs under 2-3 taps of MFMAs; the
//     weight ring is the DMA ring of conv3halo_kernel, and its counted waits include the register loads (vmcnt retires in order).
//   * optional raw split-fp16 copy of the input (hi | lo, the operand of the ResBlock's 1x1 skip convolution,
//     openaimodel.py:241): written by the tile_n == 0 workgroups for the pixels they own.
template <int BM, int BN, int WARPS_M, int WARPS_N, int NS, bool RAW>
__global__ void __launch_bounds__(WARPS_M* WARPS_N * 64) conv3halo_gn_kernel(const IGemmParams p, const int tiles_m,
                                                                              const int tiles_n, const int chunks_per_split) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NT = WARPS_M * WARPS_N * 64;
  constexpr int RPP = NT / 8;
  constexpr int PB = BN / RPP;                         // weight DMA pieces per thread per tap
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int HPMAX = (BM / 64 + 2) * 66;
  constexpr int AHP = (HPMAX + RPP - 1) / RPP;         // halo pieces (pixels) per thread per chunk
  constexpr int HROWS = AHP * RPP;                     // rows of a halo buffer
  constexpr int HALO_BYTES = HROWS * 128;
  constexpr int BSTAGE = BN * 128;
  constexpr int LDS_BYTES = 2 * HALO_BYTES + NS * BSTAGE;
  // the last 8 rows of halo buffer 1 hold the {mean, rstd} table (up to 4 images x 32 groups), the row in front of them takes the
  // writes of the pieces beyond the halo (so that every piece stores unconditionally); the launcher checks HP <= HROWS - 9
  constexpr int TAB_OFF = 2 * HALO_BYTES - 8 * 128;
  constexpr int DUMP_ROW = HROWS - 9;
  constexpr int LAG = 2;                               // taps between the loads of a piece and its conversion
  constexpr int NSLOT = AHP <= 7 ? 3 : 5;              // staging slots (8 registers each): pieces loaded at taps t, t - 1, t - 2 are live
  static_assert(AHP <= 9, "piece schedule");
  constexpr int KS = BK / 16;
  constexpr int G = (TM * TN >= 4) ? 1 : 2;
  constexpr int U = KS / G;
  constexpr int MPU = G * TM * TN;
  static_assert(PB >= 1 && TM >= 1 && TN >= 1 && RPP % 16 == 0 && NS >= 3 && NS <= 9, "tile/wave shape");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

  const int nblk = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tiles_mn = tiles_m * tiles_n;
  const int split = wgid / tiles_mn;
  const int tmn = wgid - split * tiles_mn;
  int tile_m, tile_n;
  if (p.tile_n_fastest) { tile_m = tmn / tiles_n; tile_n = tmn - tile_m * tiles_n; }
  else { tile_n = tmn / tiles_m; tile_m = tmn - tile_n * tiles_m; }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int Cin = p.c0 + p.c1;
  const int nch = Cin / BK;
  const int c_begin = split * chunks_per_split;
  const int c_end = min(nch, c_begin + chunks_per_split);
  if (c_begin >= c_end) return;

  const int tid = threadIdx.x;
  SDMI_STAMP(dbg_t0);
#ifdef SDMI_GN_POISON
  // bisecting build: the whole LDS starts as fp16 NaNs -- a fragment read that runs ahead of its tile produces NaNs instead of
  // whatever the previous kernel on this CU left there (in a repeated test: the same tile)
  for (int e = tid; e < LDS_BYTES / 16; e += NT) *(u32x4*)(smem + e * 16) = u32x4{0x7e007e00u, 0x7e007e00u, 0x7e007e00u, 0x7e007e00u};
  __syncthreads();
#endif
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int cpos = tid & 7, lrow = tid >> 3;
  const int gch = cpos ^ ((lrow >> 1) & 7);            // weight DMA: global chunk that lands at (row, cpos)
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wm = wave / WARPS_N, wn = wave - wm * WARPS_N;
  const int l31 = lane & 31, lg = lane >> 5;
  const int rsw = (l31 >> 1) & 7;

  const int W = p.Wout, H = p.Hout, HW = H * W, W2 = W + 2;
  const int bimg = m0 / HW;
  const int THI = p.halo_thi;
  const int HPI = (THI + 2) * W2;
  const int y0 = p.halo_ipt > 1 ? 0 : (m0 - bimg * HW) >> p.log2w;
  const int HP = p.halo_ipt * HPI;
  constexpr int OOB = (int)0x80000000;

  // per piece: the source pixel (-1: outside the image = zero padding, also for the rows beyond the halo), the LDS row it writes,
  // and packed bits {this tile owns the pixel, image in tile}
  int hpix[AHP];
  unsigned meta = 0;                                   // 3 bits per piece: bit 0 = own, bits 1..2 = image in tile
  unsigned beyond = 0;                                 // bit q: piece q lies beyond the halo (its write goes to the dump row)
#pragma unroll
  for (int q = 0; q < AHP; ++q) {
    const int hp = q * RPP + lrow;
    const int ip = fast_div(hp, p.magic_hpi), hr = hp - ip * HPI;
    const int hy = fast_div(hr, p.magic_w2), hx = hr - hy * W2;
    const int y = y0 + hy - 1, x = hx - 1;
    const bool valid = hp < HP && y >= 0 && y < H && x >= 0 && x < W;
    hpix[q] = valid ? ((bimg + ip) * H + y) * W + x : -1;
    const bool own = valid && hy >= 1 && hy <= THI;
    meta |= ((own ? 1u : 0u) | ((unsigned)(ip & 3) << 1)) << (3 * q);
    beyond |= (hp >= HP ? 1u : 0u) << q;
  }
  int b_off[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int n = min(n0 + i * RPP + lrow, p.N - 1);
    b_off[i] = (n * p.K + gch * 8) * 2;
  }
  int hr0[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int ml = wm * WTM + i * 32 + l31;
    const int ip = ml >> p.log2_tpi, mr = ml & ((1 << p.log2_tpi) - 1);
    hr0[i] = ip * HPI + (mr >> p.log2w) * W2 + (mr & (W - 1));
  }
  const int b_lds = 2 * HALO_BYTES + (wn * WTN + l31) * 128;
  // halo write address of piece q = lds_w0 + hbuf * HALO_BYTES + q * RPP * 128 (RPP % 16 == 0: the swizzle does not depend on q)
  const int lds_w0 = lrow * 128 + ((cpos ^ ((lrow >> 1) & 7)) << 4);
  const int lds_dump = DUMP_ROW * 128 + (lane & 7) * 16;

  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, OOB, 0x00020000);
  // fp32 sources / gamma / beta as raw descriptor words for the hand-issued loads below (base, stride 0, num_records, flags)
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  auto make_desc = [](const void* ptr, unsigned bytes) {
    const unsigned long long a = (unsigned long long)ptr;
    return i32x4{(int)(unsigned)a, (int)(unsigned)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
  };
  const i32x4 d_x0 = make_desc(p.xf0, 0x80000000u), d_x1 = make_desc(p.xf1 ? (const void*)p.xf1 : (const void*)p.xf0, 0x80000000u);
  const i32x4 d_ga = make_desc(p.gn_in_gamma, (unsigned)Cin * 4u), d_be = make_desc(p.gn_in_beta, (unsigned)Cin * 4u);
  const int pc0 = p.c0, pc1 = p.c1;
  const bool emit_raw = RAW && n0 == 0;               // (every split stages its own channel chunks)
  const int silu = p.gn_in_silu;
  const bool gn_safe = p.gn_safe != 0;                 // debugging: every counted wait drains the queue (SDMI_GN_SAFE=1)
  const unsigned long long magic_cpg = p.magic_cpg_in;
  const int cpg_in = Cin / 32;

  // fp32 source of a chunk: descriptor, row pitch (elements), first channel inside that source
  struct ChunkSrc { i32x4 desc; int ld, coff, cin0; };
  auto chunk_src = [&](int c) {
    ChunkSrc r;
    r.cin0 = c * BK;
    const bool first = r.cin0 < pc0;
    r.desc = first ? d_x0 : d_x1;
    r.ld = first ? pc0 : pc1;
    r.coff = first ? r.cin0 : r.cin0 - pc0;
    return r;
  };
  auto issue_b = [&](int kt, int stage, int q) {
    auto dst = (__attribute__((address_space(3))) void*)(smem + 2 * HALO_BYTES + stage * BSTAGE + (q * RPP + wave_u * 8) * 128);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, dst, 16, b_off[q], kt * (BK * 2), 0, SDMI_W_AUX);
  };

  // Hand-issued register loads: the compiler does not see them (beside LDS-DMA it would drain the whole ring -- vmcnt(0) -- in
  // front of their first use); their completion is counted by hand like the DMA pieces (vmcnt retires in order).
  // Two 16-byte loads fetch the 8 channels of one pixel; s_nop 4 covers a descriptor SGPR written just before.
  f32x4 st[NSLOT][2];
  f32x4 gam[2], bet[2];
  auto ld32 = [&](f32x4& lo, f32x4& hi, const i32x4& desc, int voff) {
#ifdef SDMI_GN_VISIBLE
    // bisecting build (-DSDMI_GN_VISIBLE): loads the compiler can see -- it waits for them itself (and drains the DMA ring doing so)
    const unsigned long long a = (unsigned long long)(unsigned)desc[0] | ((unsigned long long)((unsigned)desc[1] & 0xffffu) << 32);
    const bool in = (unsigned)voff < (unsigned)desc[2];
    const f32x4* src = (const f32x4*)((const char*)a + (in ? voff : 0));
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    lo = in ? __builtin_nontemporal_load(src) : z; hi = in ? __builtin_nontemporal_load(src + 1) : z;
#elif defined(SDMI_GN_POISON)
    // bisecting build (-DSDMI_GN_POISON): the destination registers hold NaNs until the load lands -- a conversion that runs ahead
    // of its data produces NaNs instead of whatever the registers held before (in a repeated test: the same data)
    const float qn = __builtin_nanf("");
    lo = f32x4{qn, qn, qn, qn}; hi = lo;
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, 0 offen\n\tbuffer_load_dwordx4 %1, %2, %3, 0 offen offset:16"
                 : "+&v"(lo), "+&v"(hi) : "v"(voff), "s"(desc) : "memory");
#else
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, 0 offen\n\tbuffer_load_dwordx4 %1, %2, %3, 0 offen offset:16"
                 : "=&v"(lo), "=&v"(hi) : "v"(voff), "s"(desc) : "memory");
#endif
  };
  auto load_piece = [&](const ChunkSrc& cs, int q) {
    const int voff = hpix[q] >= 0 ? (hpix[q] * cs.ld + cs.coff + cpos * 8) * 4 : OOB;
    ld32(st[q % NSLOT][0], st[q % NSLOT][1], cs.desc, voff);
  };
  auto load_affine = [&](const ChunkSrc& cs) {
    const int co = (cs.cin0 + cpos * 8) * 4;
    ld32(gam[0], gam[1], d_ga, co);
    ld32(bet[0], bet[1], d_be, co);
  };
  // wait until at most `n` vector-memory operations are outstanding and tie the staged registers of piece q to the wait
  auto tie_piece = [&](int q, bool affine) {
    asm volatile("" : "+v"(st[q % NSLOT][0]), "+v"(st[q % NSLOT][1]));
    if (affine) asm volatile("" : "+v"(gam[0]), "+v"(gam[1]), "+v"(bet[0]), "+v"(bet[1]));
  };
  // normalise + SiLU + round to fp16 (gn_apply_elem: the stand-alone kernel's arithmetic) and write the halo row of piece q
  auto store_piece = [&](const ChunkSrc& cs, int hbuf, int q, const f32x4& s_lo, const f32x4& s_hi) {
    const int c0 = cs.cin0 + cpos * 8;                                  // first of this thread's 8 channels (concat index)
    const int g0 = fast_div(c0, magic_cpg);
    const int nfirst = (g0 + 1) * cpg_in - c0;                          // channels of the octet in group g0 (cpg >= 8: at most two groups)
    const unsigned mq = (meta >> (3 * q)) & 7u;
    const int ip = (int)(mq >> 1);
    // (hand-issued LDS reads: in front of a compiler-visible ds_read of this array hipcc drains every pending LDS-DMA, vmcnt(0))
    float2 mr0, mr1;
    {
      const unsigned a0 = (unsigned)(TAB_OFF + (ip * 32 + g0) * 8), a1 = (unsigned)(TAB_OFF + (ip * 32 + min(g0 + 1, 31)) * 8);
      const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
      asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(mr0), "=&v"(mr1) : "v"(lbase + a0), "v"(lbase + a1) : "memory");
    }
    const bool valid = hpix[q] >= 0;
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = (j < 4 ? s_lo : s_hi)[j & 3];
      const bool second = j >= nfirst;
      const float t = gn_apply_elem(v, second ? mr1.x : mr0.x, second ? mr1.y : mr0.y, gam[j >> 2][j & 3], bet[j >> 2][j & 3], silu);
      o[j] = valid ? (f16)t : (f16)0.f;                                 // the convolution pads the NORMALISED activation with zeros
    }
    const int dst = ((beyond >> q) & 1u) ? lds_dump : (q * (RPP * 128) + lds_w0);
    *(f16x8*)(smem + hbuf * HALO_BYTES + dst) = o;
    if constexpr (RAW) {
      if (emit_raw && (mq & 1u)) {
        f16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = (j < 4 ? s_lo : s_hi)[j & 3];
          hi[j] = (f16)v; lo[j] = (f16)(v - (float)hi[j]);
        }
        const size_t ro = (size_t)hpix[q] * Cin + c0;
        *(f16x8*)(p.raw_hi + ro) = hi;
        if (p.raw_lo) *(f16x8*)(p.raw_lo + ro) = lo;
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto a_base = [&](int i, int ky, int kx, int hbuf) -> int {
    const int rowt = hr0[i] + ky * W2 + kx;
    return hbuf * HALO_BYTES + ((rowt << 7) | ((lg ^ ((rowt >> 1) & 7)) << 4));
  };
  auto read_frags = [&](const int (&ab)[TM], int bstage, int ks, f16x8 (&a)[TM], f16x8 (&b)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = *(const f16x8*)(smem + (ab[i] ^ (ks << 5)));
    const unsigned char* stp = smem + bstage * BSTAGE + b_lds + (((ks * 2 + lg) ^ rsw) << 4);
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = *(const f16x8*)(stp + j * 32 * 128);
  };
  auto mfma_step = [&](const f16x8 (&a)[TM], const f16x8 (&b)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
  };

  using Sched = HaloGnSched<AHP, PB, NS, LAG>;
  static_assert(Sched::LT(AHP - 1) + LAG <= 8, "the last piece is converted inside the chunk");

  // ---- prologue: everything the workgroup needs first is REQUESTED up front and travels together -- the first NS - 1 weight tiles
  // (LDS-DMA), gamma / beta and every piece of the first chunk's halo (a staging array that only lives here: the accumulators are
  // not live yet), the statistics accumulators -- then one wait, the {mean, rstd} table, the conversions.  (The loop starts with
  // nothing outstanding: fewer operations in flight than its counted waits allow is always safe.)
  const int kt_first = c_begin * 9, kt_last = c_end * 9 - 1;
  {
#pragma unroll
    for (int s2 = 0; s2 < NS - 1; ++s2)
#pragma unroll
      for (int q = 0; q < PB; ++q) issue_b(min(kt_first + s2, kt_last), s2, q);
    const ChunkSrc cs = chunk_src(c_begin);
    load_affine(cs);
    f32x4 pst[AHP][2];
    static_for<0, AHP>([&](auto qc) {
      const int voff = hpix[qc] >= 0 ? (hpix[qc] * cs.ld + cs.coff + cpos * 8) * 4 : OOB;
      ld32(pst[qc][0], pst[qc][1], cs.desc, voff);
    });
    // the {mean, rstd} table of the images this tile touches (the statistics are complete: fold the slots once)
  {
    const int cpg = Cin / 32;
    const double nel = (double)cpg * (double)HW;
    float* const tab = (float*)(smem + TAB_OFF);
    for (int e = tid; e < p.halo_ipt * 32 * GN_SLOTS; e += NT) {         // 8 consecutive lanes fold one (image, group)
      const int ig = e >> 3, sub = e & 7;
      const int ip = ig >> 5, g = ig & 31;
      const long long* src = p.gn_in_acc + ((size_t)((bimg + ip) * 32 + g) * GN_SLOTS + sub) * GN_STRIDE;
      long long s = src[0], sl = src[1], ss = src[2], ssl = src[3];
#pragma unroll
      for (int o = GN_SLOTS / 2; o >= 1; o >>= 1) {
        s += __shfl_xor(s, o); sl += __shfl_xor(sl, o); ss += __shfl_xor(ss, o); ssl += __shfl_xor(ssl, o);
      }
      if (sub == 0) {
        const double m = gn_acc_value(s, sl) / nel;
        double var = gn_acc_value(ss, ssl) / nel - m * m;
        if (var < 0.0) var = 0.0;
        tab[ig * 2] = (float)m;
        tab[ig * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.gn_in_eps));
      }
    }
  }
    wait_vmcnt<0>();
    asm volatile("" : "+v"(gam[0]), "+v"(gam[1]), "+v"(bet[0]), "+v"(bet[1]));
    auto tie2 = [](f32x4& a, f32x4& b) { asm volatile("" : "+v"(a), "+v"(b)); };
    static_for<0, AHP>([&](auto qc) { tie2(pst[qc][0], pst[qc][1]); });
    __syncthreads();                                   // the table is complete
    static_for<0, AHP>([&](auto qc) { store_piece(cs, 0, qc, pst[qc][0], pst[qc][1]); });
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  SDMI_STAMP(dbg_t1);
  f16x8 fa[2][G][TM], fb[2][G][TN];
  int ab[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) ab[i] = a_base(i, 0, 0, 0);
#pragma unroll
  for (int g = 0; g < G; ++g) read_frags(ab, 0, g, fa[0][g], fb[0][g]);

  int cur = 0, nxt = NS - 1, hb = 0;
  // One channel chunk = nine taps.  The next chunk's halo is staged during this chunk's taps (MORE); the last chunk of the split
  // issues the same NUMBER of operations (re-issues of a weight piece) so that every vmcnt literal stays valid, and converts
  // nothing.  Two copies of the body instead of a branch inside it: the conversions share basic blocks with the MFMAs.
  auto chunk_body = [&](int c, auto more_c) {
    constexpr bool more = decltype(more_c)::value;
    const ChunkSrc csn = chunk_src(min(c + 1, c_end - 1));
    static_for<0, 9>([&](auto tap_c) {
      constexpr int tap = decltype(tap_c)::value;
      const int kt = c * 9 + tap;
      constexpr int tap1 = tap == 8 ? 0 : tap + 1;
      const int hb1 = tap == 8 ? (hb ^ 1) : hb;
      const int cur1 = (cur + 1 == NS) ? 0 : cur + 1;
      int ab1[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) ab1[i] = a_base(i, tap1 / 3, tap1 % 3, hb1);
      const int bt = min(kt + NS - 1, kt_last);
      if constexpr (more) {
        if constexpr (tap == 0) load_affine(csn);
        static_for<0, AHP>([&](auto qc) { if constexpr (Sched::LT(qc) == tap) load_piece(csn, qc); });
        static_for<0, AHP>([&](auto qc) {
          if constexpr (Sched::LT(qc) + LAG == tap) {
            wait_vmcnt<Sched::after_piece(qc)>();
            if (gn_safe) wait_vmcnt<0>();
            tie_piece(qc, Sched::LT(qc) == 0);
            store_piece(csn, hb ^ 1, qc, st[qc % NSLOT][0], st[qc % NSLOT][1]);
#ifdef SDMI_GN_XBAR
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // bisecting build: a full barrier behind every conversion
#endif
          }
        });
      } else {
#pragma unroll
        for (int e = 0; e < Sched::loads_at(tap); ++e) issue_b(bt, nxt, 0);   // (overwritten by this tap's real piece 0 below: in order)
      }
      constexpr int ppu = (PB + U - 2) / (U - 1);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (u + 1 < U) {
#pragma unroll
          for (int g = 0; g < G; ++g) read_frags(ab, cur, (u + 1) * G + g, fa[(u + 1) & 1][g], fb[(u + 1) & 1][g]);
#pragma unroll
          for (int e = u * ppu; e < (u + 1) * ppu && e < PB; ++e) issue_b(bt, nxt, e);
        } else {
          wait_vmcnt<Sched::in_flight_ok(tap)>();
          if (gn_safe) wait_vmcnt<0>();
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
          for (int g = 0; g < G; ++g) read_frags(ab1, cur1, g, fa[0][g], fb[0][g]);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) mfma_step(fa[u & 1][g], fb[u & 1][g]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) ab[i] = ab1[i];
      cur = cur1;
      nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
      hb = hb1;
    });
  };
  for (int c = c_begin; c + 1 < c_end; ++c) chunk_body(c, std::true_type{});
  chunk_body(c_end - 1, std::false_type{});
  wait_vmcnt<0>();
  SDMI_STAMP(dbg_t2);
  igemm_epilogue<BM, BN, WARPS_M, WARPS_N, LDS_BYTES>(p, acc, m0, n0, split, tile_m, tile_n, smem);
#ifdef SDMI_IGEMM_TIMING
  if (p.dbg_times && tid == 0) {
    long long* d = p.dbg_times + 6 * (size_t)blockIdx.x;
    d[0] = dbg_t0; d[1] = dbg_t1; d[2] = dbg_t2; d[4] = (long long)__builtin_readcyclecounter();
  }
#endif
#endif  // __HIP_DEVICE_COMPILE__
}

#endif  // SDMI_EXPERIMENTS
}  // namespace

// halo-staged 3x3 convolution: supported iff stride 1, pad 1, no upsampling, power-of-two width 16..64 and tiles of whole
// image rows that do not straddle samples
bool halo_supported(const IGemmParams& p, int bm) {
  const int W = p.Wout, HW = p.Hout * p.Wout;
  if (!(p.ksize == 3 && p.stride == 1 && p.pad == 1 && !p.up && p.Hin == p.Hout && p.Win == p.Wout && W >= 8 && W <= 64 &&
        (W & (W - 1)) == 0 && bm % W == 0))
    return false;
  const int cap = (((bm / 64 + 2) * 66 + bm / 4 - 1) / (bm / 4)) * (bm / 4);    // halo rows an LDS buffer holds (AHP * RPP)
  if (bm <= HW) return HW % bm == 0 && (bm / W + 2) * (W + 2) <= cap;
  return bm % HW == 0 && p.M % bm == 0 && (bm / HW) * (p.Hout + 2) * (W + 2) <= cap;    // whole images per tile
}

template <int BM, int BN, int WARPS_M, int WARPS_N, int NS>
int launch_halo_cfg(const IGemmParams& p, int splitk, hipStream_t stream) {
  SDMI_CHECK(halo_supported(p, BM), "halo-staged conv tile requested for an unsupported shape");
  const int tiles_m = cdiv(p.M, BM), tiles_n = cdiv(p.N, BN);
  const int nch = (p.c0 + p.c1 + p.c2) / BK;
  const int chunks_per_split = cdiv(nch, splitk);
  const int nsplit = cdiv(nch, chunks_per_split);
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
  q.magic_w2 = div_magic(p.Wout + 2);
  q.log2w = 0;
  while ((1 << q.log2w) < p.Wout) ++q.log2w;
  {
    const int HW = p.Hout * p.Wout;
    q.halo_ipt = BM <= HW ? 1 : BM / HW;
    q.halo_thi = BM <= HW ? BM / p.Wout : p.Hout;
    q.magic_hpi = div_magic((q.halo_thi + 2) * (p.Wout + 2));
    const int tpi = q.halo_thi * p.Wout;             // output pixels per image part: a power of two when halo_ipt > 1
    q.log2_tpi = 0;
    while ((1 << q.log2_tpi) < tpi) ++q.log2_tpi;
    if (q.halo_ipt == 1) q.log2_tpi = 30;            // one image: every row of the tile belongs to part 0
  }
  for (int t = 0; t < p.gn_n; ++t) q.gn_magic[t] = div_magic(p.gn_cpg[t]);
  dim3 grid(tiles_m * tiles_n * nsplit), block(WARPS_M * WARPS_N * 64);
  static const int by_shape = env_int("SDMI_PROF_SHAPES", 0);
  std::string pname = std::string("conv3halo_") + std::to_string(BM) + "x" + std::to_string(BN) + "w" +
                      std::to_string(WARPS_M * WARPS_N) + "s" + std::to_string(NS);
  if (by_shape && prof_enabled())
    pname += "_M" + std::to_string(p.M) + "_N" + std::to_string(p.N) + "_K" + std::to_string(p.K) + "_s" + std::to_string(nsplit);
  const double src_pix = (double)p.B * p.Hin * p.Win;
  ProfScope ps(pname.c_str(), 2.0 * p.M * (double)p.N * p.K,
               src_pix * (p.c0 + p.c1) * 2.0 + (double)p.N * p.K * 2.0 + (double)p.M * p.N * ((p.out_f32 ? 4.0 : 0.0) + (p.out_f16 ? 2.0 : 0.0)) +
                   (p.residual ? (double)p.M * p.N * 4.0 : 0.0),
               stream);
  SDMI_LAUNCH((conv3halo_kernel<BM, BN, WARPS_M, WARPS_N, NS>), grid, block, 0, stream, q, tiles_m, tiles_n, chunks_per_split);
  SDMI_HIP_OK(hipGetLastError());
  ps.end();
  if (nsplit > 1 && !q.splitk_fused) return launch_splitk_reduce(q, nsplit, stream);
  if (q.ln_out) return launch_layernorm(q.out_f32, q.ln_gamma, q.ln_beta, q.ln_out, q.M, q.N, q.ln_eps, stream);
  return 0;
}

// May the GroupNorm-folding kernel take this convolution?  (the halo geometry of the plain kernel, at least 8 channels per group
// -- an octet of channels then spans at most two groups --, at most 4 images per tile and 8 spare rows in the halo buffer for
// the {mean, rstd} table)
bool halo_gn_supported(const IGemmParams& p, int bm) {
#ifndef SDMI_EXPERIMENTS
  return false;
#endif
  if (!halo_supported(p, bm)) return false;
  const int Cin = p.c0 + p.c1, W = p.Wout, HW = p.Hout * p.Wout;
  if (p.c2 != 0 || Cin % 64 != 0 || p.c0 % 64 != 0 || (Cin / 32) < 8) return false;
  const int ipt = bm <= HW ? 1 : bm / HW;
  const int thi = bm <= HW ? bm / W : p.Hout;
  const int cap = (((bm / 64 + 2) * 66 + bm / 4 - 1) / (bm / 4)) * (bm / 4);
  return ipt <= 4 && ipt * (thi + 2) * (W + 2) <= cap - 9 &&
         (int64_t)p.B * HW * std::max(p.c0, p.c1) * 4 < ((int64_t)1 << 31) - 65536;
}

#ifdef SDMI_EXPERIMENTS
template <int BM, int BN, int WARPS_M, int WARPS_N, int NS>
int launch_halo_gn_cfg(const IGemmParams& p, int splitk, hipStream_t stream) {
  SDMI_CHECK(halo_gn_supported(p, BM), "GroupNorm-folding halo conv requested for an unsupported shape");
  SDMI_CHECK(p.xf0 && p.gn_in_acc && p.gn_in_gamma && p.gn_in_beta && (p.c1 == 0 || p.xf1), "GroupNorm-folding halo conv: missing pointer");
  const int tiles_m = cdiv(p.M, BM), tiles_n = cdiv(p.N, BN);
  const int nch = (p.c0 + p.c1) / BK;
  const int chunks_per_split = cdiv(nch, splitk);
  const int nsplit = cdiv(nch, chunks_per_split);
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
  q.magic_w2 = div_magic(p.Wout + 2);
  q.magic_cpg_in = div_magic((p.c0 + p.c1) / 32);
  static const int env_safe = env_int("SDMI_GN_SAFE", 0);
  q.gn_safe = env_safe;
  q.log2w = 0;
  while ((1 << q.log2w) < p.Wout) ++q.log2w;
  {
    const int HW = p.Hout * p.Wout;
    q.halo_ipt = BM <= HW ? 1 : BM / HW;
    q.halo_thi = BM <= HW ? BM / p.Wout : p.Hout;
    q.magic_hpi = div_magic((q.halo_thi + 2) * (p.Wout + 2));
    const int tpi = q.halo_thi * p.Wout;
    q.log2_tpi = 0;
    while ((1 << q.log2_tpi) < tpi) ++q.log2_tpi;
    if (q.halo_ipt == 1) q.log2_tpi = 30;
  }
  for (int t = 0; t < p.gn_n; ++t) q.gn_magic[t] = div_magic(p.gn_cpg[t]);
  dim3 grid(tiles_m * tiles_n * nsplit), block(WARPS_M * WARPS_N * 64);
  static const int by_shape = env_int("SDMI_PROF_SHAPES", 0);
  std::string pname = std::string("conv3halo_gn_") + std::to_string(BM) + "x" + std::to_string(BN) + "w" +
                      std::to_string(WARPS_M * WARPS_N) + "s" + std::to_string(NS);
  if (by_shape && prof_enabled())
    pname += "_M" + std::to_string(p.M) + "_N" + std::to_string(p.N) + "_K" + std::to_string(p.K) + "_s" + std::to_string(nsplit);
  const double src_pix = (double)p.B * p.Hin * p.Win;
  ProfScope ps(pname.c_str(), 2.0 * p.M * (double)p.N * p.K,
               src_pix * (p.c0 + p.c1) * 4.0 + (double)p.N * p.K * 2.0 + (double)p.M * p.N * ((p.out_f32 ? 4.0 : 0.0) + (p.out_f16 ? 2.0 : 0.0)) +
                   (p.residual ? (double)p.M * p.N * 4.0 : 0.0),
               stream);
  if (p.raw_hi) SDMI_LAUNCH((conv3halo_gn_kernel<BM, BN, WARPS_M, WARPS_N, NS, true>), grid, block, 0, stream, q, tiles_m, tiles_n, chunks_per_split);
  else SDMI_LAUNCH((conv3halo_gn_kernel<BM, BN, WARPS_M, WARPS_N, NS, false>), grid, block, 0, stream, q, tiles_m, tiles_n, chunks_per_split);
  SDMI_HIP_OK(hipGetLastError());
  ps.end();
  if (nsplit > 1 && !q.splitk_fused) return launch_splitk_reduce(q, nsplit, stream);
  if (q.ln_out) return launch_layernorm(q.out_f32, q.ln_gamma, q.ln_beta, q.ln_out, q.M, q.N, q.ln_eps, stream);
  return 0;
}

#endif  // SDMI_EXPERIMENTS

bool gn_fold_conv_supported(int B, int H, int W, int c0, int c1, int N) {
#ifndef SDMI_EXPERIMENTS
  return false;      // (product build: the kernel is not compiled in)
#else
  // Opt-in (SDMI_FUSE_GN_CONV=1).  Same-box A/B, round 3 (profiles/gn_fold_r03.txt): 6.97 ms per UNet call with the folding kernel
  // at its heuristic sites, 8.04 ms with it at every site, against 6.50 ms for GroupNorm-apply launches + the tuned LDS-DMA
  // convolutions: every one of the N / BN column tiles of a convolution repeats the normalisation (+ SiLU: two quarter-rate
  // transcendentals per element) of its input halo, 5 ... 20 times the work of the stand-alone kernel, and reads the stream as
  // fp32 through registers instead of fp16 by DMA.
  static const int env_on = env_int("SDMI_FUSE_GN_CONV", 0);
  if (!env_on) return false;
  IGemmParams p;
  p.B = B; p.Hin = p.Hout = H; p.Win = p.Wout = W; p.ksize = 3; p.stride = 1; p.pad = 1; p.up = 0;
  p.c0 = c0; p.c1 = c1; p.M = B * H * W; p.N = N; p.K = 9 * (c0 + c1);
  for (int bm : {256, 128})
    if (bm <= std::max(p.M, 128) && halo_gn_supported(p, bm)) return true;
  return false;
#endif
}

// tile ids 14 .. 17 of the table in igemm.hip (kTiles); p.xf0 != NULL selects the GroupNorm-folding kernel
int launch_halo_gn_tile(int tile, const IGemmParams& p, int splitk, hipStream_t stream) {
#ifndef SDMI_EXPERIMENTS
  return fail("the GroupNorm-folding halo conv (conv3halo_gn_kernel) is an experiment: build with SDMI_CXXFLAGS=-DSDMI_EXPERIMENTS");
#else
  switch (tile) {
    case 14: return launch_halo_gn_cfg<256, 64, 4, 2, 5>(p, splitk, stream);
    case 15: return launch_halo_gn_cfg<256, 128, 4, 2, 3>(p, splitk, stream);
    case 16: return launch_halo_gn_cfg<128, 64, 2, 2, 8>(p, splitk, stream);
    case 17: return launch_halo_gn_cfg<128, 128, 2, 2, 5>(p, splitk, stream);
    default: return fail("not a halo-staged conv tile id");
  }
#endif
}

// tile ids 14 .. 17 of the table in igemm.hip (kTiles)
int launch_halo_tile(int tile, const IGemmParams& p, int splitk, hipStream_t stream) {
  switch (tile) {
    case 14: return launch_halo_cfg<256, 64, 4, 2, 5>(p, splitk, stream);
    case 15: return launch_halo_cfg<256, 128, 4, 2, 3>(p, splitk, stream);
    case 16: return launch_halo_cfg<128, 64, 2, 2, 8>(p, splitk, stream);
    case 17: return launch_halo_cfg<128, 128, 2, 2, 5>(p, splitk, stream);
    default: return fail("not a halo-staged conv tile id");
  }
}

}  // namespace sdmi
