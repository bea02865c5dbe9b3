// Row-strip chain kernels for gfx950 (MI355X): several dependent GEMMs of a BasicTransformerBlock / SpatialTransformer as ONE launch.
//
//   ff_tail_kernel:  x_out = x_in + proj_out( t + FF(norm3(t)) )        FF(y) = (y Wv^T + bv) * gelu(y Wg^T + bg) Wf^T + bf
//     ldm/modules/attention.py:58-64 (GEGLU, FeedForward.net), :214 (x = ff(norm3(x)) + x), :258-261 (proj_out, + x_in)
//
// at the 64 x 64 level of SD v1 (M = B * 4096 token rows of C = 320 channels), where the multi-launch path runs three GEMMs
// (GEGLU 8192 x 2560 x 320, FF-out 8192 x 320 x 1280, proj_out 8192 x 320 x 320 as 3-pass split-fp16) that each spend most of
// their 14 ... 43 us on a launch boundary, a prologue fill and an epilogue that writes an intermediate the next launch reads
// back (21 MB of GEGLU output, 10 MB of token stream, 10 MB of split-fp16 operands).
//
// Design ("row-strip chains"): a workgroup OWNS 32 token rows for the whole chain -- 8192 / 32 = 256
// workgroups = one per CU.  Every operand that is indexed by rows stays on the CU: the A operand of each GEMM is a 32 x K fp16
// strip in LDS (the LayerNorm-folded rows, then the GEGLU output chunk, then the split-fp16 halves of the FF output).  What streams
// is the WEIGHTS, all 2.8 MB of them, through one LDS-DMA ring that never drains between GEMMs.  The unit of the ring is 160 weight
// rows x 64 k (20 KB); five COMPUTE waves side by side own one 32 x 32 MFMA tile of it each (wave w = packed columns [64 w, 64 w + 64)
// of every 320-column pass: even unit = its first 32 columns, odd unit = its second 32 -- a GEGLU value / gate pair).
//
// A sixth wave is the LOADER: it alone issues every LDS-DMA of the ring (20 instructions of 1 KB per unit, ~20 cycles each) and alone
// waits for them (one in-order vmcnt queue, counted waits are exact); the compute waves never touch vector memory between the
// prologue and the final epilogue, so nothing they do -- MFMA chains, the GEGLU arithmetic -- ever sits behind a blocked DMA issue
// (the first version let all five waves issue their share: every wave was stalled ~80 cycles per instruction on the shared address
// path and stream and compute serialised: 700 cycles per unit where the stream alone takes 500 and the MFMAs alone 300;
// profiles/ff_tail_r05.txt).  One raw s_barrier per unit is the whole protocol: arriving, the loader has seen unit g land and the
// compute waves have finished reading unit g - 1, whose stage the loader then refills with unit g + NS - 1.
//
// The flat schedule of units (kFtTab):   P0(0) P1(0) | P0(1) P1(1) FF(0) | P0(2) P1(2) FF(1) | P0(3) P1(3) FF(2) | FF(3) | proj_out
// P0 / P1(h) = the two 320-column GEGLU passes of hidden chunk h (5 k-tiles x 2 halves each), FF(h) = the FF-out partial over
// hidden chunk h (all 320 output columns, k = 320 h ... 320 h + 319: k ascending over the launch -- the same fp32 additions in the
// same order as the unsplit GEMMs this replaces, outputs bit-identical to them).  The GEGLU arithmetic of a pass (LayerNorm-fold
// correction, erf GELU, value * gate: ~45 VALU instructions per output, 16 outputs per lane) is software-pipelined INTO the next
// block of units, two accumulator rows per unit: the pass's accumulators stay in registers while the next pass runs into a second
// set, the fp16 results land in one of two hidden-chunk strips, and FF(h) runs one block later, when both passes of chunk h are
// complete.  (Run after the pass as one piece the arithmetic stalled the ring for 5800 cycles eight times: 27 % of the kernel.)
//
// Register-destined loads (the token-stream rows of the FF-out residual) are issued by the compute waves, whose vmcnt then counts
// nothing else: register loads and LDS-DMA do not retire through one in-order queue (profiles/gn_fold_r03.txt), here they never
// share one.
#include <utility>

#include "igemm_dev.h"

namespace sdmi {
namespace {

constexpr int RC_ROWS = 32;                    // token rows per workgroup
constexpr int RC_NWC = 5;                      // compute waves (threads 0 .. 319)
constexpr int RC_NTC = RC_NWC * 64;
// loader waves: 1 or 2 (template argument NLD; wave RC_NWC + l issues pieces p = l mod NLD).  Four loaders = nine waves = three per SIMD
// leave 168 VGPRs per wave: the compute waves spill (measured 96 us against 63, profiles/ff_tail_r05.txt)
constexpr int RC_UROWS = 160;                  // weight rows per unit (one 32-row MFMA tile per compute wave)
constexpr int RC_UNIT = RC_UROWS * 128;        // 20 KB: 160 rows x 64 k fp16
constexpr int RC_UPIECES = RC_UNIT / 1024;     // LDS-DMA instructions per unit (20, all issued by the loader wave)
constexpr int RC_NS = 4;                       // ring depth (3, 4 and 5 measured the same)
constexpr int RC_NLD_DEFAULT = 2;
// The PREFETCH wave (template argument PF; gnconv.hip has the measurements): inside a UNet call the weights are HBM-cold, the workgroups of
// an XCD stream the same unit at the same time and a CU keeps only ~256 lines in flight, so first touches of the L2 (~1300 cycles against
// ~700 for a hit) set the pace of the ring.  The workgroups of an XCD (slot = blockIdx / 8 of gridDim / 8) share the job: behind the barrier
// of unit g each touches its 1 / nslots of the 160 lines of unit g + RC_PFD (one dword per 32 bytes, into a dead corner of LDS, never waited
// for): together they have pulled the whole unit into their L2 before anybody streams it, at ~5 lines per workgroup and unit.
#ifndef SDMI_RC_PFD
#define SDMI_RC_PFD 24
#endif
constexpr int RC_PFD = SDMI_RC_PFD;                    // (build-time A/B: tools/build_variant.sh rowchain.hip -DSDMI_RC_PFD=n; 12 / 24 / 48 measured)
#if defined(__HIP_DEVICE_COMPILE__)
// this lane's line of a unit for the prefetch wave: row offset inside the unit's 160 weight rows (64 w + [0, 32) of compute wave w) and the
// 32-byte sector, or -1
__device__ __forceinline__ int rc_pf_row(int lane, int* sector) {
  const int ntiles = gridDim.x, bid = blockIdx.x;
  const bool ok = (ntiles & 7) == 0;
  const int nslots = ok ? max(ntiles >> 3, 10) : 1, slot = ok ? (bid >> 3) : 1;
  const int idx = slot + (lane >> 2) * nslots;
  *sector = (lane & 3) * 32;
  return (slot < nslots && idx < 160) ? 64 * (idx >> 5) + (idx & 31) : -1;
}
#endif

// ---- the flat schedule of weight units of ff_tail_kernel<320>, in consumption order ------------------------------------------
constexpr int FT_C = 320, FT_KT = FT_C / 64, FT_HID = 4 * FT_C, FT_NCHUNK = 4;
constexpr int FT_BLK = 2 * FT_KT;                                          // units per block (10)
constexpr int FT_NU = (3 * FT_NCHUNK) * FT_BLK + 2 * FT_BLK;              // 12 GEGLU / FF-out blocks + proj_out {hi, lo} pairs (140)
constexpr int FT_NUH = FT_NU + FT_BLK;                                    // ... behind the out-projection of attn2 (HEAD launches: 150)
struct FtUnitTab {
  int soff[FT_NUH];     // source byte offset of the unit inside its weight matrix: row0 * row pitch + k offset
  int sel[FT_NUH];      // 0 GEGLU weights (row pitch C), 1 FF-out (4C), 2 proj_out split-fp16 (3C), 3 attn2's to_out (C)
  int aux[FT_NUH];      // at the barrier of unit g: >= 0: issue the {cs, d} column terms of that hidden chunk; -2: prefetch the x_in rows
};
// entries 0 ... FT_BLK - 1: the out-projection of attn2 (HEAD launches start there, the others at FT_BLK)
constexpr FtUnitTab ft_make_tab() {
  FtUnitTab t{};
  int u = 0;
  for (int i = 0; i < FT_NUH; ++i) t.aux[i] = -1;
  for (int kt = 0; kt < FT_KT; ++kt)
    for (int half = 0; half < 2; ++half) { t.soff[u] = (32 * half) * (FT_C * 2) + kt * 128; t.sel[u] = 3; ++u; }
  for (int h = 0; h <= FT_NCHUNK; ++h) {
    if (h < FT_NCHUNK) {
      for (int pass = 0; pass < 2; ++pass) {
        // the next chunk's column terms go into the buffer whose last reader was the carried arithmetic of P1(h - 1), inside P0(h)
        if (pass == 1 && h + 1 < FT_NCHUNK) t.aux[u] = h + 1;
        for (int kt = 0; kt < FT_KT; ++kt)
          for (int half = 0; half < 2; ++half) { t.soff[u] = (2 * FT_C * h + FT_C * pass + 32 * half) * (FT_C * 2) + kt * 128; t.sel[u] = 0; ++u; }
      }
    } else {
      t.aux[u] = -2;
    }
    if (h >= 1)
      for (int kt = 0; kt < FT_KT; ++kt)
        for (int half = 0; half < 2; ++half) { t.soff[u] = (32 * half) * (FT_HID * 2) + (FT_C * (h - 1) + 64 * kt) * 2; t.sel[u] = 1; ++u; }
  }
  for (int kt = 0; kt < FT_KT; ++kt)
    for (int half = 0; half < 2; ++half)
      for (int lo = 0; lo < 2; ++lo) { t.soff[u] = (32 * half) * (3 * FT_C * 2) + (64 * kt + (lo ? 2 * FT_C : 0)) * 2; t.sel[u] = 2; ++u; }
  return t;
}
__device__ const FtUnitTab kFtTab = ft_make_tab();

#if defined(__HIP_DEVICE_COMPILE__)
// position of element (row, col) of a [32 rows][K] fp16 operand strip in LDS: k-tile-major blocks of [32][128 B], the 16-byte
// chunks of a row XOR-swizzled with (row >> 1) & 7 -- the layout the LDS-DMA of a row-major global strip produces (source-side
// swizzle) and the MFMA fragment reads expect
__device__ __forceinline__ int strip_off(int row, int col) {
  return (col >> 6) * (RC_ROWS * 128) + row * 128 + (((((col & 63) >> 3)) ^ ((row >> 1) & 7)) << 4) + (col & 7) * 2;
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): loops whose index must be a constant (accumulator registers)
template <int... I, class F>
__device__ __forceinline__ void rc_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void rc_static_for(F&& f) { rc_static_for_impl(std::make_integer_sequence<int, N>{}, f); }
#endif

// ABL (timing build only, -DSDMI_RC_TIMING; WRONG results): 1 = the compute waves only keep the barriers (the stream alone),
// 2 = the loader only keeps the barriers (the compute side alone), 3 = no GEGLU arithmetic
// HEAD: the out-projection of attn2 in front (FfTailParams::a16 ...): t += a16 Wo^T + bo in place, the LayerNorm-folded GEGLU operand strip
// fp16(gamma3 * t) and the row statistics stay in LDS -- st_head_kernel's KIND 1 first stage (the same bits as the igemm launch it replaces:
// bias, residual = t, out_f32 = t, f16_scale, lnp_out)                 attention.py:213 (x = attn2(norm2(x), context) + x: to_out, :191-192)
template <int C, int NLD, int ABL = 0, int NS = RC_NS, bool HEAD = false, bool PF = false>
__global__ void __launch_bounds__(RC_NTC + 64 * NLD + (PF ? 64 : 0)) ff_tail_kernel(const FfTailParams rp) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(C == FT_C, "unit geometry: five waves x 64 columns = C; the unit table is generated for it");
  constexpr int KT = C / 64;                          // k-tiles of a K = C GEMM piece (5)
  constexpr int XBYTES = RC_ROWS * C * 2;             // one operand strip (20 KB)
  constexpr int HID = 4 * C, NCHUNK = HID / C;        // hidden dimension walked in chunks of C (4)
  constexpr int NU = HEAD ? FT_NUH : FT_NU;
  constexpr int TB = HEAD ? 0 : FT_BLK;               // first entry of kFtTab of this launch
  constexpr int AUXB = 2 * (2 * C) * 4;               // {cs, d} of one chunk's 2C packed GEGLU columns (5 KB)
  constexpr int RING = NS * RC_UNIT;
  constexpr int OFF_XA = RING, OFF_XG = OFF_XA + XBYTES, OFF_AUX = OFF_XG + 2 * XBYTES, OFF_TAB = OFF_AUX + 2 * AUXB;
  constexpr int OFF_LNP = OFF_TAB + RC_ROWS * 8;      // (HEAD) [row][C / 32] {sum, sum of squares} of the token stream's 32-column blocks
  constexpr int OFF_PF = OFF_LNP + (HEAD ? RC_ROWS * (C / 32) * 8 : 0);       // 256 B nobody reads (the prefetch wave's destination)
  constexpr int LDS_TOTAL = OFF_PF + (PF ? 256 : 0);
  constexpr int LSTR = 64;                            // (HEAD) row pitch (floats) of the out-projection's epilogue slabs: five 32 x 64 fp32 = the two GEGLU strips
  static_assert(RC_NWC * 32 * LSTR * 4 <= 2 * XBYTES, "the epilogue slabs live in the (still unused) hidden-chunk strips");
  static_assert(NLD == 1 || NLD == 2, "loader waves");
  constexpr int LPIECES = RC_UPIECES / NLD;           // LDS-DMA instructions per unit and loader wave
  constexpr int LD_WAIT = LPIECES * (NS - 2);         // a loader: its pieces of the NS - 2 youngest units may still be in flight
  static_assert(AUXB == 5 * 1024 && XBYTES == 4 * RC_NTC * 16, "piece counts");
  static_assert(NS >= 3 && LD_WAIT + LPIECES <= 63 && LD_WAIT <= 48 && LD_WAIT >= 5, "vmcnt is a 6-bit counter");
  static_assert(RING >= RC_NWC * 32 * 68 * 4, "the final epilogue turns its slabs through the ring");
#ifdef SDMI_RC_TIMING
  constexpr int OFF_DBG = LDS_TOTAL;                  // 128 cycle stamps of wave 0 (s_memtime), dumped at the end
  static_assert(LDS_TOTAL + 1024 <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_TOTAL + 1024];
  int n_stamp = 0;
#define RC_STAMP() do { if (threadIdx.x == 0 && n_stamp < 128) ((long long*)(smem + OFF_DBG))[n_stamp] = (long long)__builtin_readcyclecounter(); ++n_stamp; } while (0)
#else
  static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_TOTAL];
#define RC_STAMP() do { } while (0)
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int m0 = blockIdx.x * RC_ROWS;
  const IGemmParams& ep = rp.epi;
  constexpr int OOB = (int)0x80000000;

  if (PF && wave_u == RC_NWC + NLD) {
    // =============================== the prefetch wave (see RC_PFD) ================================================================
    const __amdgpu_buffer_rsrc_t rs_gg = __builtin_amdgcn_make_buffer_rsrc((void*)rp.wgg, 0, OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ff = __builtin_amdgcn_make_buffer_rsrc((void*)rp.wff2, 0, OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_po = __builtin_amdgcn_make_buffer_rsrc((void*)rp.wpo, 0, OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wo = __builtin_amdgcn_make_buffer_rsrc((void*)(HEAD ? rp.wo : rp.wgg), 0, OOB, 0x00020000);
    int sector;
    const int prow = rc_pf_row(lane, &sector);
    const int v_c = prow >= 0 ? prow * (C * 2) + sector : OOB, v_h = prow >= 0 ? prow * (HID * 2) + sector : OOB,
              v_3 = prow >= 0 ? prow * (3 * C * 2) + sector : OOB;
    auto touch = [&](int u) {
      if (u >= NU) return;
      const int soff = __builtin_amdgcn_readfirstlane(kFtTab.soff[TB + u]);
      const int sel = __builtin_amdgcn_readfirstlane(kFtTab.sel[TB + u]);
      const __amdgpu_buffer_rsrc_t rs = sel == 0 ? rs_gg : (sel == 1 ? rs_ff : (sel == 2 ? rs_po : rs_wo));
      const int vo = (sel == 0 || sel == 3) ? v_c : (sel == 1 ? v_h : v_3);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + OFF_PF), 4, vo, soff, 0, 0);
    };
    for (int u = NS - 1; u < RC_PFD; ++u) touch(u);
    for (int g = 0; g < NU; ++g) {                      // (the loaders' barriers, one for one)
      if (HEAD && g == FT_BLK) asm volatile("s_barrier" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      touch(g + RC_PFD);
    }
    wait_vmcnt<0>();
    asm volatile("s_barrier" ::: "memory");
    return;
  }
  if (wave_u >= RC_NWC) {
    // =============================== the loader waves ==============================================================================
    const int lw = wave_u - RC_NWC;                     // this loader's pieces: p = lw, lw + NLD, ... (loader 0 also brings the small operands)
    const __amdgpu_buffer_rsrc_t rs_aux = __builtin_amdgcn_make_buffer_rsrc((void*)rp.csd, 0, OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_gg = __builtin_amdgcn_make_buffer_rsrc((void*)rp.wgg, 0, OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ff = __builtin_amdgcn_make_buffer_rsrc((void*)rp.wff2, 0, OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_po = __builtin_amdgcn_make_buffer_rsrc((void*)rp.wpo, 0, OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)ep.residual, 0, OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wo = __builtin_amdgcn_make_buffer_rsrc((void*)(HEAD ? rp.wo : rp.wgg), 0, OOB, 0x00020000);
    // piece p of a unit = LDS rows 8 p ... 8 p + 7 (lane -> row 8 p + lane / 8, 16-byte position lane % 8): weight row
    // 64 (p / 4) + 8 (p % 4) + lane / 8 of the unit (tile slot p / 4 = the compute wave that owns it), source chunk (lane % 8) ^ swizzle
    // of the LDS row = (lane % 8) ^ ((4 (p % 2) + lane / 16) % 8): two voffset variants, everything else is scalar
    const int l8 = lane >> 3, cpos = lane & 7;
    const int g16_0 = (cpos ^ ((l8 >> 1) & 7)) << 4, g16_1 = (cpos ^ ((4 + (l8 >> 1)) & 7)) << 4;
    auto issue_unit = [&](int soff, int sel, int stage) {
      const __amdgpu_buffer_rsrc_t rs = sel == 0 ? rs_gg : (sel == 1 ? rs_ff : (sel == 2 ? rs_po : rs_wo));
      const int ldw2 = (sel == 0 || sel == 3) ? C * 2 : (sel == 1 ? HID * 2 : 3 * C * 2);
      const int v0 = l8 * ldw2 + g16_0, v1 = l8 * ldw2 + g16_1;
      const int vv = (lw & 1) ? v1 : v0;                // (NLD = 2, 4: the parity of a loader's pieces is its own)
#pragma unroll
      for (int pp = 0; pp < LPIECES; ++pp) {
        const int p = pp * NLD + lw;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + stage * RC_UNIT + p * 1024), 16,
                                                 NLD == 1 ? ((p & 1) ? v1 : v0) : vv, soff + (64 * (p >> 2) + 8 * (p & 3)) * ldw2, 0, SDMI_W_AUX);
      }
    };
    auto issue_aux = [&](int h) {                       // {cs, d} of hidden chunk h: 5 KB into the buffer of its parity
#pragma unroll
      for (int p = 0; p < 5; ++p)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_aux, (__attribute__((address_space(3))) void*)(smem + OFF_AUX + (h & 1) * AUXB + p * 1024), 16,
                                                 lane * 16, h * AUXB + p * 1024, 0, 0);
    };
    auto prefetch_x = [&]() {
      // one dword of each 128-byte line of the strip's x_in rows (the final epilogue's residual: written several launches ago) into a
      // dead corner of LDS, so that the epilogue's loads find the lines in the L2
#pragma unroll
      for (int p = 0; p < 5; ++p) {
        const int t = p * 64 + lane;                    // 320 lines: row t / 10, line t % 10
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(smem + OFF_AUX + p * 256), 4,
                                                 ((m0 + t / 10) * ep.ldr + (t % 10) * 32) * 4, 0, 0, 0);
      }
    };
    auto desc = [&](int u, int& soff, int& sel) {        // (past the end: the last unit again -- in bounds, never consumed)
      u = min(u, NU - 1);
      soff = __builtin_amdgcn_readfirstlane(kFtTab.soff[TB + u]);
      sel = __builtin_amdgcn_readfirstlane(kFtTab.sel[TB + u]);
    };
    if constexpr (ABL != 2) {
      if (lw == 0) issue_aux(0);
#pragma unroll
      for (int s = 0; s < NS - 1; ++s) {
        if (s == 2) wait_vmcnt<LD_WAIT>();              // (one loader: 5 + 40 in flight -- make room in the 6-bit counter)
        issue_unit(kFtTab.soff[TB + s], kFtTab.sel[TB + s], s);
      }
    }
    int nxt = NS - 1, d_soff, d_sel, d_aux;
    desc(NS - 1, d_soff, d_sel);
    d_aux = __builtin_amdgcn_readfirstlane(kFtTab.aux[TB]);
    for (int g = 0; g < NU; ++g) {
      if (HEAD && g == FT_BLK) asm volatile("s_barrier" ::: "memory");  // X1: the out-projection's operand strip is dead (its epilogue slabs go there)
      if constexpr (ABL != 2) wait_vmcnt<LD_WAIT>();    // unit g has landed
      asm volatile("s_barrier" ::: "memory");           // ... and the compute waves are done with unit g - 1
      if constexpr (ABL != 2) {
        if (d_aux != -1 && lw == 0) {                   // (wave-uniform; 4 times per launch)
          wait_vmcnt<LD_WAIT - 5>();
          if (d_aux >= 0) issue_aux(d_aux); else prefetch_x();
        }
        issue_unit(d_soff, d_sel, nxt);
      }
      desc(g + NS, d_soff, d_sel);
      d_aux = __builtin_amdgcn_readfirstlane(kFtTab.aux[TB + min(g + 1, NU - 1)]);
      nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
    }
    wait_vmcnt<0>();                                    // the surplus refills: the ring becomes the epilogue's scratch
    asm volatile("s_barrier" ::: "memory");
    return;                                             // (a barrier only waits for the waves that are still alive)
  }

  // ================================= the five compute waves ==========================================================================
  const int l31 = lane & 31, lg = lane >> 5;
  RC_STAMP();
  // LayerNorm row partials (register loads) and this wave's share of the LayerNorm-folded operand strip (LDS-DMA): the only vector
  // memory the compute waves touch before the FF-out residual
  // (HEAD: the operand strip of the out-projection -- the cross-attention output rows -- into hidden-chunk strip 0, and the token-stream rows,
  // bias and norm3 weight of the 16-byte epilogue: lane -> row 4 q + rl of a slab pass, columns nw + c4 ... + 3)
  IGemmParams lq;                                       // (lnf_request / lnf_finish read these four fields)
  lq.lnf_part = rp.lnp; lq.lnf_npart = C / 32; lq.M = ep.M; lq.lnf_eps = rp.ln_eps;
  float2 lnf_pv[LNF_MAXP];
  // (an opaque copy of the lane id: the same row / column expressions appear in the final epilogue 140 units later, and values shared
  // with it would be carried -- spilled -- across the whole main loop)
  int lane_h = lane;
  asm volatile("" : "+v"(lane_h));
  const int rl = lane_h >> 4, c4 = (lane_h & 15) * 4, nw = (tid - lane_h);
  f32x4 resv[8], colv, g4;
  if constexpr (!HEAD) { if (tid < RC_ROWS) lnf_request(lq, m0 + tid, lnf_pv); }
  {
    const __amdgpu_buffer_rsrc_t rs_ln = __builtin_amdgcn_make_buffer_rsrc((void*)(HEAD ? rp.a16 : rp.ln), 0, OOB, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = i * RC_NTC + tid;                   // 16 bytes at strip offset 16 q: k-tile q / 256, row (q / 8) % 32, position q % 8
      const int kt = q >> 8, row = (q >> 3) & 31;
      const int gch = (q & 7) ^ ((row >> 1) & 7);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_ln, (__attribute__((address_space(3))) void*)(smem + (HEAD ? OFF_XG : OFF_XA) + (i * RC_NTC + wave_u * 64) * 16), 16,
                                               ((m0 + row) * C + kt * 64 + gch * 8) * 2, 0, 0, 0);
    }
  }
  if constexpr (HEAD) {
#pragma unroll
    for (int q = 0; q < 8; ++q) resv[q] = *(const f32x4*)(rp.t + (size_t)(m0 + q * 4 + rl) * C + nw + c4);
    colv = *(const f32x4*)(rp.bo + nw + c4);
    g4 = *(const f32x4*)(rp.ln_gamma + nw + c4);
  }
  wait_vmcnt<0>();                                      // (both kinds: the only full drain of these waves before the residual)
  float2* const tab = (float2*)(smem + OFF_TAB);        // {mean, rstd} of the strip's rows
  if constexpr (!HEAD) {
    if (tid < RC_ROWS) {
      float mu, rs_;
      lnf_finish(lq, lnf_pv, &mu, &rs_);
      tab[tid] = float2{mu, rs_};
    }
  }
  RC_STAMP();

  const int rsw = (l31 >> 1) & 7;
  const int a_frag = l31 * 128, b_frag = (wave * 32 + l31) * 128;
  int cur = 0;
  auto unit_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto unit_end = [&]() { cur = (cur + 1 == NS) ? 0 : cur + 1; };
  auto frag = [&](int base, int ks) { return *(const f16x8*)(smem + base + (((ks * 2 + lg) ^ rsw) << 4)); };
  // One block of ten units = the five k-tiles of a K = C GEMM piece, two column halves each: acc0 / acc1 += A[32 x 64 k] * W_unit^T
  // (the A fragments are read once per k-tile).  carry(u): the arithmetic carried by unit u of the block (u is a compile-time
  // constant: accumulator registers are addressed by it).
  auto block = [&](int a_base, f32x16& acc0, f32x16& acc1, auto&& carry) {
    rc_static_for<KT>([&](auto ktc) {
      constexpr int kt = decltype(ktc)::value;
      f16x8 fa[4], fb[4];
      unit_sync();
      if constexpr (ABL != 1) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fa[ks] = frag(a_base + kt * (RC_ROWS * 128) + a_frag, ks);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb[ks] = frag(cur * RC_UNIT + b_frag, ks);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks], fb[ks], acc0, 0, 0, 0);
        carry(std::integral_constant<int, 2 * kt>{});
      }
      unit_end();
      unit_sync();
      if constexpr (ABL != 1) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb[ks] = frag(cur * RC_UNIT + b_frag, ks);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks], fb[ks], acc1, 0, 0, 0);
        carry(std::integral_constant<int, 2 * kt + 1>{});
      }
      unit_end();
      RC_STAMP();
    });
  };
  auto no_carry = [](auto) {};
  // The GEGLU arithmetic of one finished pass, two accumulator rows per unit (units 0 ... 7 of the carrying block), in the accumulator
  // layout (lane = column, registers = rows): LayerNorm fold (igemm_epilogue's expression), value * gelu(gate), fp16 into the
  // hidden-chunk strip of chunk h: column 160 pass + 32 wave + l31.  av / ag: the value / gate accumulators of this wave.
  auto geglu_rows = [&](auto uc, const f32x16& av, const f32x16& ag, int h, int pass) {
    constexpr int u = decltype(uc)::value;
    if constexpr (u < 8 && ABL != 3) {
      const float* const aux = (const float*)(smem + OFF_AUX + (h & 1) * AUXB);
      const int pc = C * pass + 64 * wave + l31;        // packed column of the value tile inside the chunk; gate = pc + 32
      const float cs_v = aux[pc], cs_g = aux[pc + 32], d_v = aux[2 * C + pc], d_g = aux[2 * C + pc + 32];
      const int oc = (C / 2) * pass + 32 * wave + l31;
      const int xg = OFF_XG + (h & 1) * XBYTES;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        constexpr int r0 = 2 * u;
        const int r = r0 + i;
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lg;
        const float2 mr = tab[row];
        const float val = fmaf(mr.y, av[r] - mr.x * cs_v, d_v);
        const float gate = fmaf(mr.y, ag[r] - mr.x * cs_g, d_g);
        *(f16*)(smem + xg + strip_off(row, oc)) = (f16)(val * gelu_erf(gate));
      }
    }
  };
  auto zero2 = [](f32x16 (&a)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) a[j][r] = 0.f;
  };

  f32x16 accA[2], accB[2], acc2[2];                     // GEGLU pass 0 / pass 1 of the running chunk, FF-out (all hidden chunks)
  if constexpr (HEAD) {
    // ---- t += a16 Wo^T + bo (st_head_kernel KIND 1: the same block, the same 16-byte epilogue) ----
    zero2(accA);
    block(OFF_XG, accA[0], accA[1], no_carry);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // X1: every wave is done reading the operand strip
    if constexpr (ABL != 1) {
      float* const wl = (float*)(smem + OFF_XG) + wave * (32 * LSTR);
      float2* const lnp = (float2*)(smem + OFF_LNP);
      slab_put<2, LSTR>(wl, accA, l31, lg);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int row = q * 4 + rl;
        f32x4 v = *(const f32x4*)(wl + row * LSTR + c4) + colv;
        v = v + resv[q];
        SDMI_ST_F32X4(const_cast<float*>(rp.t), (size_t)(m0 + row) * C + nw + c4, v);
        const f32x4 vs = v * g4;
        *(f16x4*)(smem + OFF_XA + strip_off(row, nw + c4)) = f16x4{(f16)vs[0], (f16)vs[1], (f16)vs[2], (f16)vs[3]};
        float s1 = (v[0] + v[1]) + (v[2] + v[3]);
        float s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        s1 = sum8_dpp(s1); s2 = sum8_dpp(s2);
        if ((lane & 7) == 0) lnp[row * (C / 32) + ((nw + c4) >> 5)] = float2{s1, s2};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    RC_STAMP();
  }
  // (HEAD) the {mean, rstd} table of the strip's rows from the LDS partials (lnf_finish's fold, block order), behind the first unit's barrier:
  // every wave's partials are visible there, and the first reader is the arithmetic carried by P1(0)
  auto ln_table = [&]() {
    if (tid < RC_ROWS) {
      const float2* pp = (const float2*)(smem + OFF_LNP) + tid * (C / 32);
      float2 pv[LNF_MAXP];
#pragma unroll
      for (int j = 0; j < LNF_MAXP; ++j) pv[j] = j < C / 32 ? pp[j] : float2{0.f, 0.f};
      float mu, rs_;
      lnf_finish(lq, pv, &mu, &rs_);
      tab[tid] = float2{mu, rs_};
    }
  };
  zero2(acc2); zero2(accB);
  for (int h = 0; h < NCHUNK; ++h) {
    // P0(h), carrying the arithmetic of P1(h - 1)
    zero2(accA);
    block(OFF_XA, accA[0], accA[1], [&](auto uc) {
      if constexpr (HEAD && decltype(uc)::value == 0) { if (h == 0) ln_table(); }
      if (h > 0) geglu_rows(uc, accB[0], accB[1], h - 1, 1);
    });
    // P1(h), carrying the arithmetic of P0(h)
    zero2(accB);
    block(OFF_XA, accB[0], accB[1], [&](auto uc) { geglu_rows(uc, accA[0], accA[1], h, 0); });
    // FF(h - 1): both passes of chunk h - 1 are complete (the second one since P0(h)); the last one carries P1(3)
    if (h > 0) block(OFF_XG + ((h - 1) & 1) * XBYTES, acc2[0], acc2[1], [&](auto uc) { if (h == NCHUNK - 1) geglu_rows(uc, accB[0], accB[1], h, 1); });
  }
  // FF(3); its residual rows (the fp32 token stream, accumulator layout) and bias are requested in front of it: nothing else of these
  // waves is in the vmcnt queue, the drain behind the block is exact and free
  float T[2][16], bff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = wave * 64 + j * 32 + l31;
    bff[j] = rp.bff2[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) T[j][r] = rp.t[(size_t)(m0 + (r & 3) + 8 * (r >> 2) + 4 * lg) * C + col];
  }
  block(OFF_XG + ((NCHUNK - 1) & 1) * XBYTES, acc2[0], acc2[1], no_carry);
  wait_vmcnt<0>();
  // ---- FF-out epilogue: t' = (acc + bias) + t (the launch's expression), then the split-fp16 operand of proj_out: hi over the
  // LayerNorm strip (last read in P1(3)), lo over hidden-chunk strip 0 (last read by FF(2)) ----
  if constexpr (ABL != 1) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wave * 64 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lg;
        const float v = acc2[j][r] + bff[j] + T[j][r];
        const f16 hi = (f16)v;
        const int o = strip_off(row, col);
        *(f16*)(smem + OFF_XA + o) = hi;
        *(f16*)(smem + OFF_XG + o) = (f16)(v - (float)hi);
      }
    }
  }
  RC_STAMP();
  // ---- proj_out as split-fp16: per (k-tile, half) the hi unit only parks its fragments, the lo unit runs the three products of
  // every k-step in gemm_split16_kernel's order (a_hi w_hi, a_lo w_hi, a_hi w_lo) ----
  f32x16 acc[1][2];
  zero2(acc[0]);
  rc_static_for<KT>([&](auto ktc) {
    constexpr int kt = decltype(ktc)::value;
    f16x8 ah[4], al[4];
    rc_static_for<2>([&](auto hc) {
      constexpr int half = decltype(hc)::value;
      f16x8 bh[4], bl[4];
      unit_sync();
      if constexpr (ABL != 1) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bh[ks] = frag(cur * RC_UNIT + b_frag, ks);
        if constexpr (half == 0) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            ah[ks] = frag(OFF_XA + kt * (RC_ROWS * 128) + a_frag, ks);
            al[ks] = frag(OFF_XG + kt * (RC_ROWS * 128) + a_frag, ks);
          }
        }
      }
      unit_end();
      unit_sync();
      if constexpr (ABL != 1) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bl[ks] = frag(cur * RC_UNIT + b_frag, ks);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          acc[0][half] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh[ks], acc[0][half], 0, 0, 0);
          acc[0][half] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh[ks], acc[0][half], 0, 0, 0);
          acc[0][half] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl[ks], acc[0][half], 0, 0, 0);
        }
      }
      unit_end();
    });
    RC_STAMP();
  });
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the loader has drained its queue: the ring is free
  RC_STAMP();
  // the shared GEMM epilogue on a 32 x 320 tile of five 64-column waves: bias + x_in, fp32 out (+ fp16 copy), GroupNorm statistics
  igemm_epilogue<RC_ROWS, C, 1, RC_NWC, RING>(ep, acc, m0, 0, 0, (int)blockIdx.x, 0, smem);
#ifdef SDMI_RC_TIMING
  RC_STAMP();
  __syncthreads();
  if (rp.dbg && tid < 128) rp.dbg[(size_t)blockIdx.x * 128 + tid] = tid < n_stamp ? ((const long long*)(smem + OFF_DBG))[tid] : 0;
#endif
#endif  // __HIP_DEVICE_COMPILE__
}


// =====================================================================================================================================
// st_head_kernel: the HEAD of a SpatialTransformer as one launch (StHeadParams, common.h)
//     t = proj_in(GroupNorm(x)) + b          attention.py:254-255 (norm, proj_in: a 1x1 conv = a dense GEMM over NHWC rows)
//     q | k | v = norm1(t) Wqkv^T            attention.py:212, 170-176 (attn1's projections; LayerNorm folded as in igemm_epilogue)
// Same skeleton as ff_tail_kernel (32 token rows per workgroup, five compute waves x 64 columns, loader waves streaming 20 KB weight
// units through the LDS-DMA ring).  The flat schedule: proj_in as split-fp16 {w_hi, w_lo} unit pairs (20 units), then q, k, v
// (10 units each).  What each stage replaces, with the same arithmetic in the same order (outputs are the same bits):
//   prologue   = gn_apply_kernel: the strip's fp32 rows are normalised ONCE into the split-fp16 operand strips {hi, lo} in LDS
//   proj_in    = gemm_split16_kernel (a_hi w_hi, a_lo w_hi, a_hi w_lo per k-step) + igemm_epilogue's 16-byte plain path: bias, fp32
//                token stream to memory, fp16(gamma1 * t) into the q|k|v operand strip, LayerNorm partials per 32-column block
//   q, k, v    = igemm_kernel + the LayerNorm-fold correction + the per-head scatter (q, k rows through the LDS slabs, v^T from the
//                accumulator registers: four consecutive tokens per lane)
//
// KIND 1, the MIDDLE of a BasicTransformerBlock (same kernel, `if constexpr`):   t += attn1_out Wo^T + b;  q2 = norm2(t) Wq^T
//     attention.py:212 (x = attn1(norm1(x)) + x: CrossAttention.to_out, :191-192), :213 + :170 (attn2.to_q over norm2(x))
// -- the out-projection of the self-attention (operand strip by LDS-DMA, residual = the token stream, updated in place) and the query
// projection of the cross-attention (10 + 10 units) instead of two GEMM launches.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#if defined(__HIP_DEVICE_COMPILE__)
// max over the two 32-lane halves of the wave, in every lane (attn.hip: v_permlane32_swap, a VALU instruction)
__device__ __forceinline__ void rc_swap_halves(float& a, float& b) {      // a.hi <-> b.lo
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float rc_max_across_halves(float x) {
  float a = x, b = x;
  rc_swap_halves(a, b);
  return fmaxf(a, b);
}
#endif
constexpr int SH_NU = 2 * FT_BLK + 3 * FT_BLK;                           // head: 50 units; middle: the first 20
struct ShUnitTab { int soff[2][SH_NU]; int sel[2][SH_NU]; };
constexpr ShUnitTab sh_make_tab() {
  ShUnitTab t{};
  int u = 0;
  for (int kt = 0; kt < FT_KT; ++kt)
    for (int half = 0; half < 2; ++half)
      for (int lo = 0; lo < 2; ++lo) { t.soff[0][u] = (32 * half) * (3 * FT_C * 2) + (64 * kt + (lo ? 2 * FT_C : 0)) * 2; t.sel[0][u] = 0; ++u; }
  for (int b = 0; b < 3; ++b)
    for (int kt = 0; kt < FT_KT; ++kt)
      for (int half = 0; half < 2; ++half) { t.soff[0][u] = (FT_C * b + 32 * half) * (FT_C * 2) + kt * 128; t.sel[0][u] = 1; ++u; }
  u = 0;
  for (int b = 0; b < 2; ++b)                           // middle: Wo [C][C], then Wq [C][C]
    for (int kt = 0; kt < FT_KT; ++kt)
      for (int half = 0; half < 2; ++half) { t.soff[1][u] = (32 * half) * (FT_C * 2) + kt * 128; t.sel[1][u] = b; ++u; }
  for (; u < SH_NU; ++u) { t.soff[1][u] = t.soff[1][2 * FT_BLK - 1]; t.sel[1][u] = 1; }
  return t;
}
__device__ const ShUnitTab kShTab = sh_make_tab();

//
// CTX (KIND 1 only): the CROSS-ATTENTION behind to_q inside the same launch (attention.py:213, 170-193 with the cached context K / V^T):
// q stays in LDS (fp16, the bits the scatter would have stored), compute wave w runs heads w and w + 5 for the strip's 32 query rows --
// attn_dma_kernel's loop on the same values in the same order (64-key tiles in the permuted row order, S^T = K Q^T, running max with the
// wave-wide rescale test, packed-fma exp2, P^T fed back from the accumulator registers, the softmax denominator as row D of O^T through a
// ones row, IEEE division at the end), with the K / V^T fragments read straight from memory (16-byte buffer loads through the same
// descriptors: the same bytes the LDS-DMA image would hold, zeros out of range) -- the attention output rows are the same bits as the
// attention launch's.  d = 40 (C = 320, 8 heads), at most 128 context keys.
template <int C, int NLD, int KIND, int NS = RC_NS, bool PF = false, bool CTX = false>
__global__ void __launch_bounds__(RC_NTC + 64 * NLD + (PF ? 64 : 0)) st_head_kernel(const StHeadParams rp) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(C == FT_C, "unit geometry: five waves x 64 columns = C");
  static_assert(NLD == 1 || NLD == 2, "loader waves");
  constexpr int KT = C / 64;
  constexpr int XBYTES = RC_ROWS * C * 2;             // one operand strip (20 KB)
  constexpr int NU = KIND == 0 ? SH_NU : 2 * FT_BLK;    // units of the launch
  constexpr int NU1 = KIND == 0 ? 2 * FT_BLK : FT_BLK;  // ... of its first GEMM (behind it: barrier X1, the token-stream epilogue)
  constexpr int RING = NS * RC_UNIT;
  constexpr int OFF_XH = RING, OFF_XL = OFF_XH + XBYTES, OFF_XN = OFF_XL + XBYTES;
  constexpr int OFF_GTAB = OFF_XN + XBYTES;           // {mean, rstd} of the sample's 32 GroupNorm groups
  constexpr int OFF_LTAB = OFF_GTAB + 32 * 8;         // {mean, rstd} of the strip's rows (norm1)
  constexpr int OFF_LNP = OFF_LTAB + RC_ROWS * 8;     // [row][C / 32] {sum, sum of squares} of the token stream's 32-column blocks
  constexpr int OFF_PF = OFF_LNP + RC_ROWS * (C / 32) * 8;        // 256 B nobody reads (the prefetch wave's destination)
  constexpr int LDS_TOTAL = OFF_PF + (PF ? 256 : 0);
  constexpr int LSTR = 64;                            // row pitch (floats) of the epilogue slabs: five 32 x 64 fp32 slabs = the two proj_in strips
  static_assert(RC_NWC * 32 * LSTR * 4 <= 2 * XBYTES, "the epilogue slabs live in the (dead) proj_in operand strips");
  constexpr int LPIECES = RC_UPIECES / NLD, LD_WAIT = LPIECES * (NS - 2);
  static_assert(NS >= 3 && LD_WAIT + LPIECES <= 63 && LD_WAIT <= 48, "vmcnt is a 6-bit counter");
#ifdef SDMI_RC_TIMING
  constexpr int OFF_DBG = LDS_TOTAL;
  static_assert(LDS_TOTAL + 1024 <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_TOTAL + 1024];
  int n_stamp = 0;
#else
  static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_TOTAL];
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int m0 = blockIdx.x * RC_ROWS;
  constexpr int OOB = (int)0x80000000;

  if (PF && wave_u == RC_NWC + NLD) {
    // =============================== the prefetch wave (see RC_PFD) ================================================================
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)rp.w_in, 0, OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_qkv = __builtin_amdgcn_make_buffer_rsrc((void*)rp.wqkv, 0, OOB, 0x00020000);
    int sector;
    const int prow = rc_pf_row(lane, &sector);
    const int v_c = prow >= 0 ? prow * (C * 2) + sector : OOB, v_3 = prow >= 0 ? prow * (3 * C * 2) + sector : OOB;
    auto touch = [&](int u) {
      if (u >= NU) return;
      const int soff = __builtin_amdgcn_readfirstlane(kShTab.soff[KIND][u]);
      const int sel = __builtin_amdgcn_readfirstlane(kShTab.sel[KIND][u]);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(sel == 0 ? rs_in : rs_qkv, (__attribute__((address_space(3))) void*)(smem + OFF_PF), 4,
                                               (KIND == 0 && sel == 0) ? v_3 : v_c, soff, 0, 0);
    };
    for (int u = NS - 1; u < RC_PFD; ++u) touch(u);
    asm volatile("s_barrier" ::: "memory");             // X0
    for (int g = 0; g < NU; ++g) {                      // (the loaders' barriers, one for one)
      if (g == NU1) asm volatile("s_barrier" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      touch(g + RC_PFD);
    }
    wait_vmcnt<0>();
    asm volatile("s_barrier" ::: "memory");
    return;
  }
  if (wave_u >= RC_NWC) {
    // =============================== the loader waves (see ff_tail_kernel) =========================================================
    const int lw = wave_u - RC_NWC;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)rp.w_in, 0, OOB, 0x00020000);      // (middle: Wo [C][C])
    const __amdgpu_buffer_rsrc_t rs_qkv = __builtin_amdgcn_make_buffer_rsrc((void*)rp.wqkv, 0, OOB, 0x00020000);
    const int l8 = lane >> 3, cpos = lane & 7;
    const int g16_0 = (cpos ^ ((l8 >> 1) & 7)) << 4, g16_1 = (cpos ^ ((4 + (l8 >> 1)) & 7)) << 4;
    auto issue_unit = [&](int soff, int sel, int stage) {
      const __amdgpu_buffer_rsrc_t rs = sel == 0 ? rs_in : rs_qkv;
      const int ldw2 = (KIND == 0 && sel == 0) ? 3 * C * 2 : C * 2;
      const int v0 = l8 * ldw2 + g16_0, v1 = l8 * ldw2 + g16_1;
      const int vv = (lw & 1) ? v1 : v0;
#pragma unroll
      for (int pp = 0; pp < LPIECES; ++pp) {
        const int p = pp * NLD + lw;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + stage * RC_UNIT + p * 1024), 16,
                                                 NLD == 1 ? ((p & 1) ? v1 : v0) : vv, soff + (64 * (p >> 2) + 8 * (p & 3)) * ldw2, 0, SDMI_W_AUX);
      }
    };
    auto desc = [&](int u, int& soff, int& sel) {
      u = min(u, NU - 1);
      soff = __builtin_amdgcn_readfirstlane(kShTab.soff[KIND][u]);
      sel = __builtin_amdgcn_readfirstlane(kShTab.sel[KIND][u]);
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue_unit(kShTab.soff[KIND][s], kShTab.sel[KIND][s], s);
    asm volatile("s_barrier" ::: "memory");             // X0: the compute waves' GroupNorm table
    int nxt = NS - 1, d_soff, d_sel;
    desc(NS - 1, d_soff, d_sel);
    for (int g = 0; g < NU; ++g) {
      if (g == NU1) asm volatile("s_barrier" ::: "memory");             // X1: the first GEMM's operand strips are dead (the epilogue slabs go there)
      wait_vmcnt<LD_WAIT>();                            // unit g has landed
      asm volatile("s_barrier" ::: "memory");           // ... and the compute waves are done with unit g - 1
      issue_unit(d_soff, d_sel, nxt);
      desc(g + NS, d_soff, d_sel);
      nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
    }
    wait_vmcnt<0>();
    asm volatile("s_barrier" ::: "memory");
    return;
  }

  // ================================= the five compute waves ==========================================================================
#ifdef SDMI_RC_TIMING
#define SH_STAMP() do { if (threadIdx.x == 0 && n_stamp < 128) ((long long*)(smem + OFF_DBG))[n_stamp] = (long long)__builtin_readcyclecounter(); ++n_stamp; } while (0)
#else
#define SH_STAMP() do { } while (0)
#endif
  SH_STAMP();
  const int l31 = lane & 31, lg = lane >> 5;
  const int bsample = m0 / rp.ntok;                    // (32 | ntok: the strip lies inside one sample)
  const int tok0 = m0 - bsample * rp.ntok;
  // ---- prologue: the strip's fp32 rows (thread -> rows xr + 8 i, channels 8 c8 ... 8 c8 + 7), gamma / beta of those channels, the
  // LayerNorm-fold column terms of this lane's q | k | v columns; then the sample's GroupNorm table ----
  const int xr = tid / 40, c8 = tid - xr * 40;
  const int rl = lane >> 4, c4 = (lane & 15) * 4;      // the 16-byte epilogues: lane -> row q * 4 + rl of a slab pass, columns nw + c4 ... + 3
  const int nw = wave * 64;
  f32x4 xv[4][2], gv[2], bv[2];
  f32x4 resv[8];                                        // middle: the token-stream rows the first epilogue adds (and overwrites)
  constexpr int NB = KIND == 0 ? 3 : 1;                 // LayerNorm-folding GEMMs behind the token stream
  float fcs[NB][2], fdn[NB][2];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = C * b + 64 * wave + 32 * j + l31;
      fcs[b][j] = rp.lnf_cs[n]; fdn[b][j] = rp.lnf_d[n];
    }
  float2* const gtab = (float2*)(smem + OFF_GTAB);
  float2* const ltab = (float2*)(smem + OFF_LTAB);
  if constexpr (KIND == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* src = rp.x + (size_t)(m0 + xr + 8 * i) * C + c8 * 8;
      xv[i][0] = *(const f32x4*)src; xv[i][1] = *(const f32x4*)(src + 4);
    }
    gv[0] = *(const f32x4*)(rp.gn_gamma + c8 * 8); gv[1] = *(const f32x4*)(rp.gn_gamma + c8 * 8 + 4);
    bv[0] = *(const f32x4*)(rp.gn_beta + c8 * 8); bv[1] = *(const f32x4*)(rp.gn_beta + c8 * 8 + 4);
  } else {
    // the fp16 operand strip (the attention output rows) by LDS-DMA into strip_off's layout (ff_tail_kernel's prologue), the residual rows
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)rp.a16, 0, OOB, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = i * RC_NTC + tid;                   // 16 bytes at strip offset 16 q: k-tile q / 256, row (q / 8) % 32, position q % 8
      const int kt = q >> 8, row = (q >> 3) & 31;
      const int gch = (q & 7) ^ ((row >> 1) & 7);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(smem + OFF_XH + (i * RC_NTC + wave_u * 64) * 16), 16,
                                               ((m0 + row) * C + kt * 64 + gch * 8) * 2, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) resv[q] = *(const f32x4*)(rp.t + (size_t)(m0 + q * 4 + rl) * C + nw + c4);
  }
  if (KIND == 0 && tid < 256) {
    // (norm.hip gn_fold: 8 consecutive lanes fold the 8 slots of a group)
    const int g = tid >> 3, sub = tid & 7;
    const long long* src = rp.gn_acc + ((size_t)(bsample * 32 + g) * GN_SLOTS + sub) * GN_STRIDE;
    long long a = src[0], al = src[1], q = src[2], ql = src[3];
#pragma unroll
    for (int o = GN_SLOTS / 2; o >= 1; o >>= 1) {
      a += __shfl_xor(a, o); al += __shfl_xor(al, o); q += __shfl_xor(q, o); ql += __shfl_xor(ql, o);
    }
    if (sub == 0) {
      float m, r;
      gn_mean_rstd(a, al, q, ql, (double)(C / 32) * (double)rp.ntok, rp.gn_eps, &m, &r);
      gtab[g] = float2{m, r};
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");        // X0
  if constexpr (KIND == 0) {
    // normalise (gn_apply_kernel's expression), split into hi | lo (lo_half), write the two operand strips in strip_off's layout
    constexpr int cpg = C / 32;
    const int c0 = c8 * 8;
    const int g0 = c0 / cpg;
    const int nfirst = (g0 + 1) * cpg - c0;             // channels of the octet in group g0 (cpg >= 8: at most two groups)
    const float2 ga = gtab[g0], gb = gtab[min(g0 + 1, 31)];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = xr + 8 * i;
      f16x8 hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool second = j >= nfirst;
        const float y = gn_apply_elem(xv[i][j >> 2][j & 3], second ? gb.x : ga.x, second ? gb.y : ga.y, gv[j >> 2][j & 3], bv[j >> 2][j & 3], 0);
        hi[j] = (f16)y; lo[j] = (f16)(y - (float)hi[j]);
      }
      const int off = (c8 >> 3) * (RC_ROWS * 128) + row * 128 + (((c8 & 7) ^ ((row >> 1) & 7)) << 4);
      *(f16x8*)(smem + OFF_XH + off) = hi;
      *(f16x8*)(smem + OFF_XL + off) = lo;
    }
  }
  SH_STAMP();

  const int rsw = (l31 >> 1) & 7;
  const int a_frag = l31 * 128, b_frag = (wave * 32 + l31) * 128;
  int cur = 0;
  auto unit_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto unit_end = [&]() { cur = (cur + 1 == NS) ? 0 : cur + 1; };
  auto frag = [&](int base, int ks) { return *(const f16x8*)(smem + base + (((ks * 2 + lg) ^ rsw) << 4)); };
  auto zero2 = [](f32x16 (&a)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) a[j][r] = 0.f;
  };

  // ---- proj_in as split-fp16: per (k-tile, half) the hi unit parks its fragments, the lo unit runs the three products of every k-step
  // in gemm_split16_kernel's order (a_hi w_hi, a_lo w_hi, a_hi w_lo) ----
  // One block of ten units = the five k-tiles of a K = C GEMM over an operand strip, two column halves each.  first(): run once behind
  // the first unit's barrier (the LayerNorm table of the rows: every wave's partials are visible there).
  auto block = [&](f32x16 (&a)[2], int a_base, auto&& first) {
    rc_static_for<KT>([&](auto ktc) {
      constexpr int kt = decltype(ktc)::value;
      f16x8 fa[4], fb[4];
      unit_sync();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fa[ks] = frag(a_base + kt * (RC_ROWS * 128) + a_frag, ks);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fb[ks] = frag(cur * RC_UNIT + b_frag, ks);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks], fb[ks], a[0], 0, 0, 0);
      if constexpr (kt == 0) first();
      unit_end();
      unit_sync();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fb[ks] = frag(cur * RC_UNIT + b_frag, ks);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks], fb[ks], a[1], 0, 0, 0);
      unit_end();
    });
  };
  auto nothing = []() {};
  f32x16 acc[2];
  zero2(acc);
  if constexpr (KIND == 1) block(acc, OFF_XH, nothing);                 // the out-projection (plain fp16 operands)
  if constexpr (KIND == 0)
  rc_static_for<KT>([&](auto ktc) {
    constexpr int kt = decltype(ktc)::value;
    f16x8 ah[4], al[4];
    rc_static_for<2>([&](auto hc) {
      constexpr int half = decltype(hc)::value;
      f16x8 bh[4], bl[4];
      unit_sync();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bh[ks] = frag(cur * RC_UNIT + b_frag, ks);
      if constexpr (half == 0) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          ah[ks] = frag(OFF_XH + kt * (RC_ROWS * 128) + a_frag, ks);
          al[ks] = frag(OFF_XL + kt * (RC_ROWS * 128) + a_frag, ks);
        }
      }
      unit_end();
      unit_sync();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bl[ks] = frag(cur * RC_UNIT + b_frag, ks);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        acc[half] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh[ks], acc[half], 0, 0, 0);
        acc[half] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh[ks], acc[half], 0, 0, 0);
        acc[half] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl[ks], acc[half], 0, 0, 0);
      }
      unit_end();
    });
  });
  SH_STAMP();
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // X1: every wave is done reading the proj_in strips
  // ---- proj_in epilogue (igemm_epilogue's 16-byte plain path on this wave's 32 x 64 slab): t = acc + bias -> memory (fp32), fp16(gamma1 * t)
  // -> the q | k | v operand strip, {sum, sum of squares} per 32-column block -> the LayerNorm partial table ----
  float* const wl = (float*)(smem + OFF_XH) + wave * (32 * LSTR);
  {
    const f32x4 colv = *(const f32x4*)(rp.b_in + nw + c4);
    const f32x4 g4 = *(const f32x4*)(rp.ln_gamma + nw + c4);
    slab_put<2, LSTR>(wl, acc, l31, lg);
    float2* const lnp = (float2*)(smem + OFF_LNP);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = q * 4 + rl;
      f32x4 v = *(const f32x4*)(wl + row * LSTR + c4) + colv;
      if constexpr (KIND == 1) v = v + resv[q];
      SDMI_ST_F32X4(rp.t, (size_t)(m0 + row) * C + nw + c4, v);
      const f32x4 vs = v * g4;
      *(f16x4*)(smem + OFF_XN + strip_off(row, nw + c4)) = f16x4{(f16)vs[0], (f16)vs[1], (f16)vs[2], (f16)vs[3]};
      float s1 = (v[0] + v[1]) + (v[2] + v[3]);
      float s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      s1 = sum8_dpp(s1); s2 = sum8_dpp(s2);
      if ((lane & 7) == 0) lnp[row * (C / 32) + ((nw + c4) >> 5)] = float2{s1, s2};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  SH_STAMP();

  auto ln_table = [&]() {
    if (tid < RC_ROWS) {                                // (lnf_finish's fold of the block partials, in block order)
      IGemmParams lq;
      lq.lnf_npart = C / 32; lq.lnf_eps = rp.ln_eps;
      const float2* pp = (const float2*)(smem + OFF_LNP) + tid * (C / 32);
      float2 pv[LNF_MAXP];
#pragma unroll
      for (int j = 0; j < LNF_MAXP; ++j) pv[j] = j < C / 32 ? pp[j] : float2{0.f, 0.f};
      float mu, rs_;
      lnf_finish(lq, pv, &mu, &rs_);
      ltab[tid] = float2{mu, rs_};
    }
  };
  // LayerNorm-fold correction of a finished accumulator pair (igemm_epilogue's expression)
  auto fold = [&](f32x16 (&a)[2], const float (&cs_)[2], const float (&dn_)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float2 mr = ltab[(r & 3) + 8 * (r >> 2) + 4 * lg];
        a[j][r] = fmaf(mr.y, a[j][r] - mr.x * cs_[j], dn_[j]);
      }
  };
  // q / k: [B * heads][ntok][dh] rows through the wave's LDS slab -- a lane stores 4 consecutive dd of a token
  auto rows_out = [&](f32x16 (&a)[2], f16* dst) {
    slab_put<2, LSTR>(wl, a, l31, lg);
    const int n = nw + c4;
    const int head = n / rp.dh, dd = n - head * rp.dh;
    f16* const base = dst + (((size_t)bsample * rp.heads + head) * rp.ntok + tok0) * rp.dh + dd;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = q * 4 + rl;
      const f32x4 v = *(const f32x4*)(wl + row * LSTR + c4);
      SDMI_ST(f16x4, base + (size_t)row * rp.dh, (f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]}));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  zero2(acc);
  block(acc, OFF_XN, ln_table);                         // q
  SH_STAMP();
  if constexpr (KIND == 1 && CTX) {
    static_assert(C == 320, "head dim 40: eight heads of five 16-byte chunks");
    constexpr int D = 40, DKS = 3, DVT = 2, KVT = 64;
    const float sc = rp.ctx_scale * 1.4426950408889634f;
    const int nt = (rp.ctx_nkv + KVT - 1) / KVT;                        // 1 or 2 (the launcher checked nkv <= 128)
    const int nheads = (rp.heads - wave + RC_NWC - 1) / RC_NWC;         // this wave's heads: wave, wave + 5
    const int nsteps = nheads * nt;                                     // (head, tile) steps: at most 4
    // the fragments of step i = (head wave + 5 (i / nt), tile i % nt): K rows in the permuted order (bits 2 and 3 of the row swapped),
    // V^T rows d (row D = ones, beyond: zeros).  Two register sets: step i + 2 is requested while step i + 1 is computed.
    auto load_frags = [&](int i, u32x4 (&kf)[2][DKS], u32x4 (&vf)[2][2][DVT]) {
      const int head = wave + RC_NWC * (nt == 2 ? (i >> 1) : i), t = nt == 2 ? (i & 1) : 0;
      const int bh = bsample * rp.heads + head;
      const __amdgpu_buffer_rsrc_t rsrc_k = __builtin_amdgcn_make_buffer_rsrc((void*)(rp.ctx_k + (size_t)bh * rp.ctx_nkv * D), 0, rp.ctx_nkv * D * 2, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc((void*)(rp.ctx_vt + (size_t)bh * D * rp.ctx_nkv_pad), 0, D * rp.ctx_nkv_pad * 2, 0x00020000);
#pragma unroll
      for (int kvb = 0; kvb < 2; ++kvb) {
        const int row = kvb * 32 + l31;
        const int key = t * KVT + ((row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1));
#pragma unroll
        for (int ks = 0; ks < DKS; ++ks) kf[kvb][ks] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_k, key * (D * 2) + (ks * 2 + lg) * 16, 0, 0);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int dt = 0; dt < DVT; ++dt) {
            const int d = dt * 32 + l31;
            const unsigned one2 = d == D ? 0x3C003C00u : 0u;
            u32x4 v = {one2, one2, one2, one2};
            if (d < D) v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, d * (rp.ctx_nkv_pad * 2) + t * (KVT * 2) + (4 * kvb + 2 * s2 + lg) * 16, 0, 0);
            vf[kvb][s2][dt] = v;
          }
      }
    };
    f16x8 qf[DKS];
    f32x16 o[DVT];
    float m_run = -1e30f;
    auto step = [&](int i, const u32x4 (&kf)[2][DKS], const u32x4 (&vf)[2][2][DVT]) {
      const int head = wave + RC_NWC * (nt == 2 ? (i >> 1) : i), t = nt == 2 ? (i & 1) : 0;
      if (t == 0) {
        // Q^T fragments: lane (q = l31, g = lg) holds q[q][40 head + 16 ks + 8 g .. + 8]; zero beyond D
#pragma unroll
        for (int ks = 0; ks < DKS; ++ks) {
          const int dcol = ks * 16 + lg * 8;
          f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
          if (dcol < D) v = *(const f16x8*)(smem + OFF_XN + strip_off(l31, head * D + dcol));
          qf[ks] = v;
        }
#pragma unroll
        for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
        m_run = -1e30f;
      }
      // ---- S^T = K Q^T (two 32-key blocks) ----
      f32x16 sa[2];
#pragma unroll
      for (int kvb = 0; kvb < 2; ++kvb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sa[kvb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < DKS; ++ks)
          sa[kvb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kf[kvb][ks]), qf[ks], sa[kvb], 0, 0, 0);
      }
      const int kv0 = t * KVT;
      if (kv0 + KVT > rp.ctx_nkv) {
#pragma unroll
        for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kv = kv0 + kvb * 32 + (r & 3) + 4 * ((r >> 2) & 1) + 8 * lg + 16 * (r >> 3);
            if (kv >= rp.ctx_nkv) sa[kvb][r] = -1e30f;
          }
      }
      float mx = -1e30f;
#pragma unroll
      for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sa[kvb][r]);
      mx = rc_max_across_halves(mx);
      const float m_new = fmaxf(m_run, mx * sc);
      if (__any(m_new > m_run)) {
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      }
      const f32x2 sc2 = {sc, sc}, nm2 = {-m_run, -m_run};
#pragma unroll
      for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 sv = {sa[kvb][r], sa[kvb][r + 1]};
          const f32x2 e = __builtin_elementwise_fma(sv, sc2, nm2);
          sa[kvb][r] = __builtin_amdgcn_exp2f(e[0]);
          sa[kvb][r + 1] = __builtin_amdgcn_exp2f(e[1]);
        }
      // ---- O^T += V^T P^T ----
#pragma unroll
      for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          f16x8 pf;
#pragma unroll
          for (int e = 0; e < 8; ++e) pf[e] = (f16)sa[kvb][8 * s2 + e];
#pragma unroll
          for (int dt = 0; dt < DVT; ++dt)
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, vf[kvb][s2][dt]), pf, o[dt], 0, 0, 0);
        }
      if (t == nt - 1) {
        // the denominator: row D of O^T = tile D / 32, local row D % 32 = 8: register (8 & 3) + 4 (8 >> 3) = 4 of lane half (8 >> 2) & 1 = 0
        float la = o[D / 32][4], lb = la;
        rc_swap_halves(la, lb);
        const float inv = 1.0f / la;
        f16* orow = rp.ao_out + (size_t)(m0 + l31) * C + (size_t)head * D;
#pragma unroll
        for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int dd = dt * 32 + 8 * r4 + 4 * lg;
            if (dd < D) {
              const f16x4 v = {(f16)(o[dt][r4 * 4 + 0] * inv), (f16)(o[dt][r4 * 4 + 1] * inv), (f16)(o[dt][r4 * 4 + 2] * inv), (f16)(o[dt][r4 * 4 + 3] * inv)};
              SDMI_ST(f16x4, orow + dd, v);
            }
          }
      }
    };
    u32x4 kfa[2][DKS], vfa[2][2][DVT], kfb[2][DKS], vfb[2][2][DVT];
    if (nsteps > 0) load_frags(0, kfa, vfa);          // (K / V^T do not depend on q: requested in front of the fold and the q strip)
    if (nsteps > 1) load_frags(1, kfb, vfb);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // (pairs with the loaders' last barrier: they are gone after it)
    fold(acc, fcs[0], fdn[0]);
    // q -> fp16 strip in LDS (over the dead out-projection operand strip), in strip_off's layout: the values rows_out would have stored
    slab_put<2, LSTR>(wl, acc, l31, lg);
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) {
      const int row = qq * 4 + rl;
      const f32x4 v = *(const f32x4*)(wl + row * LSTR + c4);
      *(f16x4*)(smem + OFF_XN + strip_off(row, nw + c4)) = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // a head's 40 columns come from two waves
    if (nsteps > 0) { step(0, kfa, vfa); if (nsteps > 2) load_frags(2, kfa, vfa); }
    if (nsteps > 1) { step(1, kfb, vfb); if (nsteps > 3) load_frags(3, kfb, vfb); }
    if (nsteps > 2) step(2, kfa, vfa);
    if (nsteps > 3) step(3, kfb, vfb);
    return;
  }
  if constexpr (KIND == 1) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // (pairs with the loaders' last barrier)
    fold(acc, fcs[0], fdn[0]);
    rows_out(acc, rp.q);
#ifdef SDMI_RC_TIMING
    SH_STAMP();
    if (rp.dbg && tid < 64) rp.dbg[(size_t)blockIdx.x * 128 + tid] = tid < n_stamp ? ((const long long*)(smem + OFF_DBG))[tid] : 0;
#endif
    return;
  }
  fold(acc, fcs[0], fdn[0]);
  rows_out(acc, rp.q);
  SH_STAMP();
  zero2(acc);
  block(acc, OFF_XN, nothing);                          // k
  SH_STAMP();
  fold(acc, fcs[NB > 1 ? 1 : 0], fdn[NB > 1 ? 1 : 0]);
  rows_out(acc, rp.k);
  SH_STAMP();
  zero2(acc);
  block(acc, OFF_XN, nothing);                          // v
  SH_STAMP();
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (pairs with the loaders' last barrier)
  fold(acc, fcs[NB > 2 ? 2 : 0], fdn[NB > 2 ? 2 : 0]);
  // v^T: [B * heads][dh][ntok_pad] straight from the accumulator registers (lane = column, registers 4 r4 ... 4 r4 + 3 = 4 consecutive tokens)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = nw + j * 32 + l31;
    const int head = n / rp.dh, dd = n - head * rp.dh;
    f16* const base = rp.vt + (((size_t)bsample * rp.heads + head) * rp.dh + dd) * rp.ntok_pad + tok0 + 4 * lg;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4)
      *(f16x4*)(base + 8 * r4) = f16x4{(f16)acc[j][r4 * 4 + 0], (f16)acc[j][r4 * 4 + 1], (f16)acc[j][r4 * 4 + 2], (f16)acc[j][r4 * 4 + 3]};
  }
#ifdef SDMI_RC_TIMING
  SH_STAMP();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (rp.dbg && tid < 64) rp.dbg[(size_t)blockIdx.x * 128 + tid] = tid < n_stamp ? ((const long long*)(smem + OFF_DBG))[tid] : 0;
#endif
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace

// May the tail of a SpatialTransformer (GEGLU -> FF-out -> proj_out) run as one ff_tail_kernel launch?  C channels, M token rows,
// hw rows per sample.
bool ff_tail_supported(int C, int M, int hw) { return C == 320 && M % RC_ROWS == 0 && hw % RC_ROWS == 0 && (int64_t)M * C * 4 < ((int64_t)1 << 31); }

#ifdef SDMI_RC_TIMING
// timing build only: where the next launches dump their stamps ([workgroup][128] cycle counts of wave 0) and which ablation they run
static long long* g_rc_dbg = nullptr;
static int g_rc_abl = 0;
extern "C" int sdmi_k_ff_tail_dbg(void* stamps, int abl) { g_rc_dbg = (long long*)stamps; g_rc_abl = abl; return 0; }
#endif

int launch_ff_tail(const FfTailParams& p, hipStream_t stream) {
  const IGemmParams& e = p.epi;
  const int C = e.N;
  SDMI_CHECK(ff_tail_supported(C, e.M, e.Hout * e.Wout), "ff_tail: C = 320, rows (per sample) multiples of 32");
  const bool head = p.a16 != nullptr;
  SDMI_CHECK((head || (p.ln && p.lnp)) && p.csd && p.wgg && p.wff2 && p.bff2 && p.t && p.wpo, "ff_tail: null operand");
  SDMI_CHECK(!head || (p.wo && p.bo && p.ln_gamma), "ff_tail: the out-projection in front needs wo, bo and the norm3 weight");
  SDMI_CHECK(e.mode == EPI_PLAIN && e.K == C && e.M == e.B * e.Hout * e.Wout && e.out_f32 && e.ldo % 4 == 0, "ff_tail: proj_out descriptor");
  SDMI_CHECK(e.residual && e.ldr >= C, "ff_tail: proj_out adds the SpatialTransformer's input (residual)");
  SDMI_CHECK(!e.lnf_part && !e.lnp_out && !e.f16_scale && !e.ln_out && !e.out_lo && !e.rowvec, "ff_tail: plain proj_out epilogue only");
  if (e.gn_n) {
    SDMI_CHECK(e.gn_n <= 2 && (e.Hout * e.Wout) % 32 == 0, "GroupNorm statistics need Hout*Wout % 32 == 0");
    for (int t = 0; t < e.gn_n; ++t)
      SDMI_CHECK(e.gn_acc[t] && e.gn_cpg[t] >= 2 && (e.gn_cbase[t] + e.N + e.gn_cpg[t] - 1) / e.gn_cpg[t] <= 32, "bad GroupNorm statistics target");
  }
  FfTailParams q = p;
  q.epi.splitk = 1; q.epi.splitk_fused = 0; q.epi.slab_tiled = 0;
  q.epi.epi_vec = epi_vec_ok(e);
  SDMI_CHECK((int64_t)e.M < (int64_t)65536 * e.Hout * e.Wout, "fast_div_hw: at most 65535 samples per launch");
  q.epi.magic_hw = div_magic_hw(e.Hout * e.Wout);
  q.epi.magic_w = div_magic(e.Wout);
  for (int t = 0; t < e.gn_n; ++t) q.epi.gn_magic[t] = div_magic(e.gn_cpg[t]);
  const double M = e.M;
  // algorithmic work of the three reference ops (one fp16 read of each operand / weight, the fp32 stream in and out)
  ProfScope ps(head ? "st_tail_32x320w5" : "ff_tail_32x320w5", 2.0 * M * (8.0 * C * C + 4.0 * C * C + (double)C * C + (head ? (double)C * C : 0.0)),
               M * C * (2.0 + 4.0 + 4.0 + 4.0 + (head ? 4.0 : 0.0) + (e.out_f16 ? 2.0 : 0.0)) + (head ? 14.0 : 13.0) * C * C * 2.0, stream,
               2.0 * M * (8.0 * C * C + 4.0 * C * C + 3.0 * C * C + (head ? (double)C * C : 0.0)));
  const dim3 grid(e.M / RC_ROWS);
  const int nld_env = env_int("SDMI_FF_TAIL_LD", RC_NLD_DEFAULT);      // loader waves (read per launch: A/B)
#ifdef SDMI_RC_TIMING
  q.dbg = g_rc_dbg;
  const int abl = g_rc_abl % 10, nld = g_rc_abl >= 10 ? g_rc_abl / 10 : nld_env;      // 10 a + x: ablation x with a loader waves
  SDMI_CHECK(!head, "ff_tail timing build: the out-projection in front is not instantiated");
#define RC_LAUNCH(NLD, ABL) SDMI_LAUNCH((ff_tail_kernel<320, NLD, ABL>), grid, dim3(RC_NTC + 64 * NLD), 0, stream, q)
#define RC_ABL(NLD) switch (abl) { case 1: RC_LAUNCH(NLD, 1); break; case 2: RC_LAUNCH(NLD, 2); break; case 3: RC_LAUNCH(NLD, 3); break; default: RC_LAUNCH(NLD, 0); break; }
  if (nld == 1) { RC_ABL(1) } else { RC_ABL(2) }
#else
  const int pf = env_int("SDMI_CHAIN_PF", 1);           // the prefetch wave (read per launch: A/B)
  if (nld_env == 1) {
    if (head) SDMI_LAUNCH((ff_tail_kernel<320, 1, 0, RC_NS, true>), grid, dim3(RC_NTC + 64), 0, stream, q);
    else SDMI_LAUNCH((ff_tail_kernel<320, 1>), grid, dim3(RC_NTC + 64), 0, stream, q);
  } else if (pf) {
    if (head) SDMI_LAUNCH((ff_tail_kernel<320, 2, 0, RC_NS, true, true>), grid, dim3(RC_NTC + 192), 0, stream, q);
    else SDMI_LAUNCH((ff_tail_kernel<320, 2, 0, RC_NS, false, true>), grid, dim3(RC_NTC + 192), 0, stream, q);
  } else {
    if (head) SDMI_LAUNCH((ff_tail_kernel<320, 2, 0, RC_NS, true>), grid, dim3(RC_NTC + 128), 0, stream, q);
    else SDMI_LAUNCH((ff_tail_kernel<320, 2>), grid, dim3(RC_NTC + 128), 0, stream, q);
  }
#endif
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}


// May the head of a SpatialTransformer (GroupNorm-apply -> proj_in -> q | k | v) run as one st_head_kernel launch?
bool st_head_supported(int C, int M, int ntok, int ntok_pad, int heads, int dh) {
  return C == 320 && M % RC_ROWS == 0 && ntok % RC_ROWS == 0 && M % ntok == 0 && heads * dh == C && dh % 4 == 0 && ntok % 4 == 0 &&
         ntok_pad % 4 == 0 && ntok_pad >= ntok && (int64_t)M * C * 4 < ((int64_t)1 << 31);
}

#ifdef SDMI_RC_TIMING
static long long* g_sh_dbg = nullptr;
extern "C" int sdmi_k_st_head_dbg(void* stamps) { g_sh_dbg = (long long*)stamps; return 0; }
#endif

int launch_st_head(const StHeadParams& p, hipStream_t stream) {
  SDMI_CHECK(st_head_supported(p.C, p.M, p.ntok, p.ntok_pad, p.heads, p.dh), "st_head: C = 320, rows (per sample) multiples of 32, heads * dh = C, dh % 4 = 0");
  SDMI_CHECK(p.x && p.gn_acc && p.gn_gamma && p.gn_beta && p.w_in && p.b_in && p.t && p.ln_gamma && p.wqkv && p.lnf_cs && p.lnf_d && p.q && p.k && p.vt,
             "st_head: null operand");
  SDMI_CHECK(p.B * p.ntok == p.M, "st_head: M = B * ntok");
  const double M = p.M, C = p.C;
  // algorithmic work of the reference ops (proj_in once, q | k | v), one read of x and the weights, t / q / k / v written once
  ProfScope ps("st_head_32x320w5", 2.0 * M * (C * C + 3.0 * C * C), M * C * (4.0 + 4.0 + 3 * 2.0) + 4.0 * C * C * 2.0, stream,
               2.0 * M * (3.0 * C * C + 3.0 * C * C));
  StHeadParams q = p;
  const dim3 grid(p.M / RC_ROWS);
  const int nld = env_int("SDMI_FF_TAIL_LD", RC_NLD_DEFAULT);          // loader waves (read per launch: A/B)
#ifdef SDMI_RC_TIMING
  q.dbg = g_sh_dbg;
#endif
  if (nld == 1) SDMI_LAUNCH((st_head_kernel<320, 1, 0>), grid, dim3(RC_NTC + 64), 0, stream, q);
  else if (env_int("SDMI_CHAIN_PF", 1)) SDMI_LAUNCH((st_head_kernel<320, 2, 0, RC_NS, true>), grid, dim3(RC_NTC + 192), 0, stream, q);
  else SDMI_LAUNCH((st_head_kernel<320, 2, 0>), grid, dim3(RC_NTC + 128), 0, stream, q);
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}

// the middle of a BasicTransformerBlock: t += a16 Wo^T + b (in place), q = norm2(t) Wq^T (StHeadParams: a16, w_in = Wo [C][C], b_in, t,
// ln_gamma = norm2 weight, wqkv = Wq [C][C], lnf_cs / lnf_d [C], q)
int launch_st_mid(const StHeadParams& p, hipStream_t stream) {
  SDMI_CHECK(st_head_supported(p.C, p.M, p.ntok, p.ntok_pad > 0 ? p.ntok_pad : p.ntok, p.heads, p.dh), "st_mid: C = 320, rows (per sample) multiples of 32, heads * dh = C, dh % 4 = 0");
  const bool ctx = p.ctx_k != nullptr;
  SDMI_CHECK(p.a16 && p.w_in && p.b_in && p.t && p.ln_gamma && p.wqkv && p.lnf_cs && p.lnf_d && (ctx || p.q), "st_mid: null operand");
  SDMI_CHECK(!ctx || (p.ctx_vt && p.ao_out && p.dh == 40 && p.ctx_nkv >= 1 && p.ctx_nkv <= 128 && p.ctx_nkv_pad % 8 == 0 && p.ctx_nkv_pad >= p.ctx_nkv),
             "st_mid with the cross-attention inside: head dim 40, 1 .. 128 context keys, V^T rows padded to a multiple of 8");
  SDMI_CHECK(p.B * p.ntok == p.M, "st_mid: M = B * ntok");
  const double M = p.M, C = p.C;
  // (with the cross-attention inside: + 4 B h N Nkv d flops of the attention product)
  ProfScope ps(ctx ? "st_mid_ctx_32x320w5" : "st_mid_32x320w5", 2.0 * M * (2.0 * C * C) + (ctx ? 4.0 * M * p.ctx_nkv * C : 0.0),
               M * C * (2.0 + 4.0 + 4.0 + 2.0) + 2.0 * C * C * 2.0, stream);
  StHeadParams q = p;
  const dim3 grid(p.M / RC_ROWS);
  const int nld = env_int("SDMI_FF_TAIL_LD", RC_NLD_DEFAULT);
#ifdef SDMI_RC_TIMING
  q.dbg = g_sh_dbg;
#endif
  if (ctx) SDMI_LAUNCH((st_head_kernel<320, 2, 1, RC_NS, true, true>), grid, dim3(RC_NTC + 192), 0, stream, q);
  else if (nld == 1) SDMI_LAUNCH((st_head_kernel<320, 1, 1>), grid, dim3(RC_NTC + 64), 0, stream, q);
  else if (env_int("SDMI_CHAIN_PF", 1)) SDMI_LAUNCH((st_head_kernel<320, 2, 1, RC_NS, true>), grid, dim3(RC_NTC + 192), 0, stream, q);
  else SDMI_LAUNCH((st_head_kernel<320, 2, 1>), grid, dim3(RC_NTC + 128), 0, stream, q);
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}

}  // namespace sdmi
