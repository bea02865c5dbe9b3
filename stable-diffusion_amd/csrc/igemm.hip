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

#include "common.h"
#include "prof.h"

namespace sdmi {
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

int launch_splitk_reduce(const IGemmParams& p, int nsplit, hipStream_t stream);

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

namespace {

constexpr int BK = 64;

// Per-workgroup phase stamps (s_memtime at kernel entry, k-loop entry, k-loop exit, kernel exit) exist only in a build with
// -DSDMI_IGEMM_TIMING (SDMI_CXXFLAGS=-DSDMI_IGEMM_TIMING SDMI_LIB_OUT=... python stable-diffusion_amd/build.py; tools/igemm_timing.py):
// the product library carries no trace of them.
#ifdef SDMI_IGEMM_TIMING
#define SDMI_STAMP(name) const long long name = p.dbg_times ? (long long)__builtin_readcyclecounter() : 0
#else
#define SDMI_STAMP(name)
#endif

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

// counted wait: the immediate must be a literal; `n` is a compile-time constant at every call site (a template argument,
// or a value that is constant after loop unrolling), so the switch folds to the one s_waitcnt
__device__ __forceinline__ void wait_vmcnt_n(int n) {
  switch (n) {
#define SDMI_VMCNT_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break
    SDMI_VMCNT_CASE(0); SDMI_VMCNT_CASE(1); SDMI_VMCNT_CASE(2); SDMI_VMCNT_CASE(3); SDMI_VMCNT_CASE(4);
    SDMI_VMCNT_CASE(5); SDMI_VMCNT_CASE(6); SDMI_VMCNT_CASE(7); SDMI_VMCNT_CASE(8); SDMI_VMCNT_CASE(9);
    SDMI_VMCNT_CASE(10); SDMI_VMCNT_CASE(11); SDMI_VMCNT_CASE(12); SDMI_VMCNT_CASE(13); SDMI_VMCNT_CASE(14);
    SDMI_VMCNT_CASE(15); SDMI_VMCNT_CASE(16); SDMI_VMCNT_CASE(17); SDMI_VMCNT_CASE(18); SDMI_VMCNT_CASE(19);
    SDMI_VMCNT_CASE(20); SDMI_VMCNT_CASE(21); SDMI_VMCNT_CASE(22); SDMI_VMCNT_CASE(23); SDMI_VMCNT_CASE(24);
    SDMI_VMCNT_CASE(25); SDMI_VMCNT_CASE(26); SDMI_VMCNT_CASE(27); SDMI_VMCNT_CASE(28); SDMI_VMCNT_CASE(29);
    SDMI_VMCNT_CASE(30); SDMI_VMCNT_CASE(31); SDMI_VMCNT_CASE(32); SDMI_VMCNT_CASE(33); SDMI_VMCNT_CASE(34);
    SDMI_VMCNT_CASE(35); SDMI_VMCNT_CASE(36); SDMI_VMCNT_CASE(37); SDMI_VMCNT_CASE(38); SDMI_VMCNT_CASE(39);
    SDMI_VMCNT_CASE(40); SDMI_VMCNT_CASE(41); SDMI_VMCNT_CASE(42); SDMI_VMCNT_CASE(43); SDMI_VMCNT_CASE(44);
    SDMI_VMCNT_CASE(45); SDMI_VMCNT_CASE(46); SDMI_VMCNT_CASE(47); SDMI_VMCNT_CASE(48);
#undef SDMI_VMCNT_CASE
    default: __builtin_trap();      // (vmcnt is a 6-bit field: 63 outstanding at most)
  }
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 48, "add the literal");
  wait_vmcnt_n(N);
}

// Kernel kinds: the gather of the implicit A matrix differs, so each is its own instantiation (no runtime branches and no
// dead per-row state in the k-loop).
enum : int { KIND_1X1 = 0, KIND_3X3 = 1, KIND_3X3_UP = 2 };

// floor(m / d) for 0 <= m, m * d < 2^40, with magic = ceil(2^40 / d) (host computed): the per-row (batch, y, x) split of
// the prologue without the ~40-instruction integer division sequences
__device__ __forceinline__ int fast_div(int m, unsigned long long magic) {
  return (int)(((unsigned long long)(unsigned)m * magic) >> 40);
}

#if defined(__HIP_DEVICE_COMPILE__)
// ---- accumulator slabs through LDS (the 16-byte epilogues) -------------------------------------------------------------
// A wave owns one LDS region of 32 rows x LSTR floats.  slab_put writes the wave's TN 32x32 MFMA accumulator tiles of one
// 32-row slab in the C/D register layout (col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)); afterwards lane l
// reads 16 bytes at row q * RPI + l / LPR, column 4 * (l % LPR): WTN / 4 lanes cover a row, a wave instruction covers RPI
// whole rows (128- or 256-byte runs in memory).  Only the owning wave touches its region, and a wave's LDS operations execute
// in order, so a drained lgkmcnt (plus a compiler barrier) is all the synchronisation the turn-around needs.
template <int WTN> constexpr int SLAB_LPR = WTN / 4;       // lanes per row
template <int WTN> constexpr int SLAB_RPI = 64 / (WTN / 4);  // rows per wave instruction
template <int WTN> constexpr int SLAB_NPASS = 32 / (64 / (WTN / 4));
// row pitch in floats: 16-byte aligned rows with 4 banks of skew where the LDS allows it (not the 2-stage 128x128 8-wave tile)
template <int NWAVES, int WTN, int LDS_BYTES>
constexpr int SLAB_LSTR = (NWAVES * 32 * (WTN + 4) * 4 <= LDS_BYTES) ? WTN + 4 : WTN;
template <int NWAVES, int WTN, int LDS_BYTES>
__device__ __forceinline__ float* slab_base(unsigned char* smem, int wave) {
  static_assert(WTN == 32 || WTN == 64, "lane mapping of the 16-byte epilogue");
  static_assert(NWAVES * 32 * SLAB_LSTR<NWAVES, WTN, LDS_BYTES> * 4 <= LDS_BYTES, "LDS too small for the epilogue slabs");
  return (float*)smem + wave * (32 * SLAB_LSTR<NWAVES, WTN, LDS_BYTES>);
}
template <int TN, int LSTR>
__device__ __forceinline__ void slab_put(float* wl, const f32x16 (&a)[TN], int l31, int lg) {
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) wl[((r & 3) + 8 * (r >> 2) + 4 * lg) * LSTR + j * 32 + l31] = a[j][r];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
template <int TN, int LSTR>
__device__ __forceinline__ void slab_get(const float* wl, f32x16 (&a)[TN], int l31, int lg) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) a[j][r] = wl[((r & 3) + 8 * (r >> 2) + 4 * lg) * LSTR + j * 32 + l31];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// The epilogue shared by the GEMM kernels (generic implicit GEMM and the halo-staged 3x3 convolution): accumulators of the
// wave's TM x TN MFMA tiles -> bias / time-embedding row vector / residual / fp32 + fp16 (+ split-fp16 low half) stores,
// GEGLU, per-head q / k / v^T scatter, split-K slabs, GroupNorm statistics.  `smem` = the block's LDS (free at this point:
// every LDS-DMA of the block has landed and is no longer read), LDS_BYTES its size.
template <int BM, int BN, int WARPS_M, int WARPS_N, int LDS_BYTES>
__device__ __forceinline__ void igemm_epilogue(const IGemmParams& p_arg, f32x16 (&acc)[BM / WARPS_M / 32][BN / WARPS_N / 32],
                                               const int m0, const int n0, const int split, const int tile_m,
                                               const int tile_n, unsigned char* smem) {
#ifdef SDMI_IGEMM_TIMING
  IGemmParams p = p_arg;                                  // timing build: epilogue ablations (wrong results, time only)
  if (p.dbg_abl & 1) p.residual = nullptr;
  if (p.dbg_abl & 4) { p.out_f32 = nullptr; p.out_f16 = nullptr; p.out_lo = nullptr; }
#else
  const IGemmParams& p = p_arg;
#endif
  constexpr int NT = WARPS_M * WARPS_N * 64;
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int LSTR = SLAB_LSTR<WARPS_M * WARPS_N, WTN, LDS_BYTES>;       // row pitch of the 16-byte epilogues' LDS slabs
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WARPS_N, wn = wave - wm * WARPS_N;
  const int l31 = lane & 31, lg = lane >> 5;
  const int HWout = p.Hout * p.Wout;
  // C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int mw = m0 + wm * WTM, nw = n0 + wn * WTN;
  // Full interior tiles take a branch-free path: all residual loads of a 32-row slab are issued back to back
  // (independent), column terms are hoisted, and no per-element bounds checks split the stores into dependent
  // load -> wait -> store chains (those chains were ~70 % of the short-K kernels' time, profiles/ablate2_r01.txt).
  const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.N);
  if (p.splitk > 1 && p.splitk_fused) {
    // ---- fused split-K reduction: accumulators to this split's slab in register order (16 bytes per lane, a wave writes
    // 1 KB runs), ticket; all but the last block of the tile are done.  The last one re-reads every split's slab IN INDEX
    // ORDER (its own included: the sum does not depend on which block came last) and falls through to the ordinary
    // epilogue.  The blocks of a tile run on different XCDs, whose L2s are not coherent with each other: the slab stores
    // and loads carry the agent-scope bit (sc1: performed at the memory side), which orders them against the ticket
    // with plain s_waitcnt -- an agent-scope release / acquire FENCE instead writes back / invalidates the whole L2
    // per wave and cost ~60 us per GEMM (profiles/splitk_fused_r02.txt).
    constexpr int SC1 = 16;                                // buffer cache-policy bit: agent scope
    const int tile_lin = tile_m * ((p.N + BN - 1) / BN) + tile_n;
    const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.splitk_ws + (size_t)tile_lin * p.splitk * (BM * BN)), 0, p.splitk * (BM * BN) * 4, 0x00020000);
    const int my_off = (split * (BM * BN) + tid * 4) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const f32x4 v = {acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ws, my_off + ((i * TN + j) * 4 + r4) * (NT * 16), 0, SC1);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's slab stores are performed ...
    __syncthreads();                                       // ... every wave's (and the LDS is free: all are out of the k-loop)
    if (tid == 0) *(volatile int*)smem = __hip_atomic_fetch_add(p.splitk_cnt + tile_lin, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = *(volatile int*)smem;
    if (ticket != p.splitk - 1) return;
    if (tid == 0) __hip_atomic_store(p.splitk_cnt + tile_lin, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int s = 0; s < p.splitk; ++s) {
      const int off = (s * (BM * BN) + tid * 4) * 4;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ws, off + ((i * TN + j) * 4 + r4) * (NT * 16), 0, SC1));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][4 * r4 + e] += v[e];
          }
    }
    __syncthreads();                       // smem[0] is reused below
  }
  const bool unfused_split = p.splitk > 1 && !p.splitk_fused;
  if (p.mode == EPI_PLAIN) {
    const bool atomic = unfused_split;     // unfused split-K: raw partial sums go to this split's slab
    float* slab = atomic ? (p.splitk_ws + (size_t)split * p.M * p.N) : nullptr;
    const int b_first = m0 / HWout;
    const bool one_batch = ((m0 + BM - 1) / HWout == b_first);
    if (full && one_batch && p.epi_vec && !atomic) {
      // ---- 16-byte epilogue: every wave turns its 32 x WTN accumulator slabs through its own LDS region (the tile buffers are
      // free now) so that a lane owns 4 CONSECUTIVE columns of a row: one dwordx4 residual load, one dwordx4 fp32 store and
      // one 8-byte fp16 store per 4 values instead of a dword / short access each -- the same bytes in a quarter of the
      // vector-memory instructions.  The arithmetic is the scalar path's, value by value ((acc + column term) + residual):
      // results are bit-identical.  Measured (profiles/epilogue_16byte_r02.txt): -3 ... -12 % epilogue cycles here, -45 ... -70 %
      // on the q / k scatter below; the split-K slab stores and the GEGLU epilogue got SLOWER through the LDS turn (stores
      // without loads in front of them are fire-and-forget either way) and keep their register-layout stores.
      __syncthreads();                                     // every wave's LDS-DMA has landed and nobody reads the tiles any more
      float* const wl = slab_base<WARPS_M * WARPS_N, WTN, LDS_BYTES>(smem, wave);
      const int rl = lane / SLAB_LPR<WTN>, c4 = (lane % SLAB_LPR<WTN>) * 4;
      f32x4 colv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) colv = *(const f32x4*)(p.bias + nw + c4);
      if (p.rowvec) colv += *(const f32x4*)(p.rowvec + (size_t)b_first * p.ld_rowvec + nw + c4);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        slab_put<TN, LSTR>(wl, acc[i], l31, lg);
        constexpr int NP = SLAB_NPASS<WTN>, RPI = SLAB_RPI<WTN>;
        f32x4 resv[NP];
        if (p.residual) {
#pragma unroll
          for (int q = 0; q < NP; ++q)
            resv[q] = *(const f32x4*)(p.residual + (size_t)(mw + i * 32 + q * RPI + rl) * p.ldr + nw + c4);
        } else {
#pragma unroll
          for (int q = 0; q < NP; ++q) resv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const int row = q * RPI + rl;
          float* const lp = wl + row * LSTR + c4;
          const f32x4 v = *(const f32x4*)lp + colv + resv[q];
          const size_t ro = (size_t)(mw + i * 32 + row) * p.ldo + nw + c4;
          if (p.out_f32) *(f32x4*)(p.out_f32 + ro) = v;
          const f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
          if (p.out_f16) *(f16x4*)(p.out_f16 + ro) = h;
          if (p.out_lo)
            *(f16x4*)(p.out_lo + ro) = f16x4{(f16)(v[0] - (float)h[0]), (f16)(v[1] - (float)h[1]), (f16)(v[2] - (float)h[2]),
                                            (f16)(v[3] - (float)h[3])};
          if (p.gn_n > 0) *(f32x4*)lp = v;                 // final values back for the statistics below
        }
        if (p.gn_n > 0) slab_get<TN, LSTR>(wl, acc[i], l31, lg);
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // slab reads done before the next slab_put overwrites them
      }
    } else if (full && one_batch) {
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
              acc[i][j][r] = v;                       // final value, kept for the GroupNorm statistics below
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
              acc[i][j][r] = v;
              if (p.out_f32) p.out_f32[(size_t)m * p.ldo + n] = v;
              if (p.out_f16) p.out_f16[(size_t)m * p.ldo + n] = (f16)v;
              if (p.out_lo) p.out_lo[(size_t)m * p.ldo + n] = (f16)(v - (float)(f16)v);
            }
          }
        }
      }
    }
    // ---- GroupNorm statistics of the finished output, for the GroupNorm(s) that will read it (up to two: the next
    // layer's, and the skip-concat's of an output block): {sum, sum of squares} per (sample, group) of this tile, added as
    // fixed-point int64 to the consumer's accumulators -- the same words norm.hip's statistics kernel fills, so that
    // kernel (one launch per GroupNorm) is not needed.  Integer adds are associative: bit-reproducible.  The waves of the
    // block first combine in LDS (the tile buffers are free now), so the block issues ONE global atomic set per
    // (sample, group) it touched: the global adds, not the arithmetic, are what statistics cost.
    // Needs Hout*Wout % 32 == 0 (a 32-row MFMA tile lies inside one sample); the executor checks it.
#ifdef SDMI_IGEMM_TIMING
    if (p.dbg_times && threadIdx.x == 0) p.dbg_times[6 * (size_t)blockIdx.x + 3] = (long long)__builtin_readcyclecounter();
    if (p.dbg_abl & 2) return;
#endif
    if (p.gn_n > 0 && !atomic) {
      constexpr int GNB = BM / 32;                        // samples a tile can touch (Hout*Wout >= 32)
      unsigned long long* lacc = (unsigned long long*)smem;                  // [target][sample in tile][group][GN_WORDS]
      static_assert(2 * GNB * 32 * GN_WORDS * 8 <= LDS_BYTES, "LDS too small for the statistics accumulators");
      __syncthreads();                                    // every wave's LDS-DMA has landed (wait_vmcnt<0> above) and is unread
      for (int e = tid; e < 2 * GNB * 32 * GN_WORDS; e += NT) lacc[e] = 0ull;
      __syncthreads();
      const int b_tile = m0 / HWout;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = nw + j * 32 + l31;
        const bool nvalid = n < p.N;
        auto flush = [&](int b, float s1, float s2) {
          if (!nvalid) { s1 = 0.f; s2 = 0.f; }
          s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);         // the two half-waves hold disjoint rows of a column
          for (int t = 0; t < p.gn_n; ++t) {
            const int gid = nvalid ? fast_div(p.gn_cbase[t] + n, p.gn_magic[t]) : -1;
            float a1 = s1, a2 = s2;
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {                   // segmented sum over runs of equal group id
              const float t1 = __shfl_down(a1, off, 32), t2 = __shfl_down(a2, off, 32);
              const int tg = __shfl_down(gid, off, 32);
              if (l31 + off < 32 && tg == gid) { a1 += t1; a2 += t2; }
            }
            const int gprev = __shfl_up(gid, 1, 32);
            if (lg == 0 && gid >= 0 && (l31 == 0 || gprev != gid)) {
              unsigned long long* dst = lacc + ((size_t)(t * GNB + (b - b_tile)) * 32 + gid) * GN_WORDS;
              gn_acc_add(dst, a1);
              gn_acc_add(dst + 2, a2);
            }
          }
        };
        float s1 = 0.f, s2 = 0.f;
        int bcur = -1;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int mt = mw + i * 32;                  // wave-uniform
          if (mt < p.M) {
            const int bi = mt / HWout;
            if (bcur >= 0 && bi != bcur) { flush(bcur, s1, s2); s1 = 0.f; s2 = 0.f; }
            bcur = bi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int m = mt + (r & 3) + 8 * (r >> 2) + 4 * lg;
              if (m < p.M) { const float v = acc[i][j][r]; s1 += v; s2 += v * v; }
            }
          }
        }
        if (bcur >= 0) flush(bcur, s1, s2);
      }
      __syncthreads();
      const int slot = (tile_m + tile_n) & (GN_SLOTS - 1);
      for (int e = tid; e < p.gn_n * GNB * 32 * GN_WORDS; e += NT) {
        const unsigned long long w = lacc[e];
        if (w == 0ull) continue;
        const int word = e % GN_WORDS, g = (e / GN_WORDS) % 32, bl = (e / (GN_WORDS * 32)) % GNB, t = e / (GN_WORDS * 32 * GNB);
        if (b_tile + bl >= p.B) continue;
        atomicAdd((unsigned long long*)p.gn_acc[t] + ((size_t)((b_tile + bl) * 32 + g) * GN_SLOTS + slot) * GN_STRIDE + word, w);
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
    if (unfused_split) {   // raw partial tile to this split's slab; splitk_reduce_heads_kernel scatters the sum
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
    if (p.bias) {                          // q/k/v projections with a bias (CLIP text model); the UNet's have none
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = nw + j * 32 + l31;
        const float bv = n < p.N ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += bv;
      }
    }
    // row-layout segments (q, k: [token][dh]) through the LDS slabs: a lane stores 4 consecutive dd of a token as one 8-byte
    // quad instead of 4 shorts.  The launcher checked segC % 32 == 0 (a 32-column block lies in one segment) and dh % 4 == 0.
    // Transposed segments (v^T) keep the register path below: there a lane already owns 4 consecutive tokens.
    const bool vecq = full && p.epi_vec;
    if (vecq) {
      __syncthreads();
      float* const wl = slab_base<WARPS_M * WARPS_N, WTN, LDS_BYTES>(smem, wave);
      const int rl = lane / SLAB_LPR<WTN>, c4 = (lane % SLAB_LPR<WTN>) * 4;
      const int n = nw + c4;
      const int seg = n / p.segC;
      const int c = n - seg * p.segC;
      const int head = c / p.dh;
      const int dd = c - head * p.dh;
      f16* const dst = p.seg_dst[seg];
      const bool rowseg = p.seg_kind[seg] == 0;
      bool any_row = false;
#pragma unroll
      for (int j = 0; j < TN; ++j) any_row |= p.seg_kind[(nw + j * 32) / p.segC] == 0;     // wave-uniform
      if (any_row) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          slab_put<TN, LSTR>(wl, acc[i], l31, lg);
          if (rowseg) {
#pragma unroll
            for (int q = 0; q < SLAB_NPASS<WTN>; ++q) {
              const int row = q * SLAB_RPI<WTN> + rl;
              const int m = mw + i * 32 + row;
              const int b = m / p.ntok;
              const int tok = m - b * p.ntok;
              const f32x4 a = *(const f32x4*)(wl + row * LSTR + c4);
              *(f16x4*)(dst + (((size_t)b * p.heads + head) * p.ntok + tok) * p.dh + dd) = f16x4{(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]};
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slab reads done before the next slab_put overwrites them
        }
      }
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
      if (vecq && kind == 0) continue;       // stored above
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

#endif  // __HIP_DEVICE_COMPILE__

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
    const int b = fast_div(m, p.magic_hw);
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
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, dst, 16, b_off[q - A_PASSES], c.b_soff, 0, 0);
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
  igemm_epilogue<BM, BN, WARPS_M, WARPS_N, NS * STAGE_BYTES>(p, acc, m0, n0, split, tile_m, tile_n, smem);
#ifdef SDMI_IGEMM_TIMING
  if (p.dbg_times && tid == 0) {        // (where a workgroup's time goes; blocks that return early in the epilogue are not stamped)
    long long* d = p.dbg_times + 6 * (size_t)blockIdx.x;      // d[3] = after the output stores (written inside the epilogue)
    d[0] = dbg_t0; d[1] = dbg_t1; d[2] = dbg_t2; d[4] = (long long)__builtin_readcyclecounter();
  }
#endif
#endif  // __HIP_DEVICE_COMPILE__
}

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
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, dst, 16, b_off[q], kt * (BK * 2), 0, 0);
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
    if (p.out_f32) *(f32x4*)(p.out_f32 + (size_t)m * p.ldo + n) = v;
    if (p.out_f16) *(f16x4*)(p.out_f16 + (size_t)m * p.ldo + n) = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
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

static unsigned long long div_magic(int d) {      // ceil(2^40 / d), see fast_div
  const unsigned long long one = 1ull << 40;
  return (one + (unsigned long long)d - 1) / (unsigned long long)d;
}

// see the kernels' tile numbering: true = an XCD owns rows of A (M > N), false = it owns weight panels
static int tile_order_n_fastest(const IGemmParams& p) {
  static const int env_order = env_int("SDMI_TILE_ORDER", 0);      // 0 auto, 1 always M fastest (round-1 order), 2 always N fastest
  if (env_order == 1) return 0;
  if (env_order == 2) return 1;
  return p.M > p.N ? 1 : 0;
}

// split-K slabs a (tile, split) choice needs, in floats: register-order slabs of whole tiles when the reduction is fused
// into the GEMM (see igemm_epilogue), [split][M][N] for the separate reduce kernel
static bool splitk_fusable(const IGemmParams& p, int bm, int bn) {
  // default off: same-box A/B (profiles/splitk_fused_r02.txt) has the separate reduce kernel ahead, 3.23 vs 3.16 images/s
  static const int env_fused = env_int("SDMI_SPLITK_FUSED", 0);
  return env_fused && p.splitk_cnt && (int64_t)cdiv(p.M, bm) * cdiv(p.N, bn) <= p.splitk_cnt_ints;
}
static int64_t splitk_ws_need(const IGemmParams& p, int bm, int bn, int nsplit) {
  if (nsplit <= 1) return 0;
  if (splitk_fusable(p, bm, bn)) return (int64_t)nsplit * cdiv(p.M, bm) * bm * cdiv(p.N, bn) * bn;
  return (int64_t)nsplit * p.M * p.N;
}

// May this launch use the 16-byte epilogues (igemm_epilogue: unsplit plain mode, q / k of the per-head scatter)?  They need
// 16-byte aligned fp32 rows and 8-byte aligned fp16 rows at every multiple-of-4 column, and for the scatter 32-column blocks
// that lie inside one segment.
// SDMI_EPI_VEC=0 keeps the dword / short epilogues (A/B; the results are bit-identical).
static int epi_vec_ok(const IGemmParams& p) {
  if (!env_int("SDMI_EPI_VEC", 1) || p.N % 4) return 0;      // (read per launch: the tests flip it between two calls)
  auto al = [](const void* q, uintptr_t a) { return ((uintptr_t)q & (a - 1)) == 0; };
  if (p.mode == EPI_PLAIN)
    return p.ldo % 4 == 0 && al(p.out_f32, 16) && al(p.out_f16, 8) && al(p.out_lo, 8) && al(p.bias, 16) &&
           al(p.rowvec, 16) && p.ld_rowvec % 4 == 0 && al(p.residual, 16) && p.ldr % 4 == 0;
  if (p.mode == EPI_GEGLU) return 0;
  return p.segC % 32 == 0 && p.dh % 4 == 0 && al(p.bias, 16) && al(p.seg_dst[0], 8) && al(p.seg_dst[1], 8) && al(p.seg_dst[2], 8);
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
  q.epi_vec = epi_vec_ok(p);
  SDMI_CHECK(splitk_ws_need(p, BM, BN, nsplit) <= p.splitk_ws_floats, "split-K workspace too small");
  q.magic_hw = div_magic(p.Hout * p.Wout);
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
    if (dma) hipLaunchKernelGGL((igemm_kernel<BM, BN, WARPS_M, WARPS_N, true, NS, K_>), grid, block, 0, stream, q,   \
                                tiles_m, tiles_n, kt_per_split);                                                    \
    else hipLaunchKernelGGL((igemm_kernel<BM, BN, WARPS_M, WARPS_N, false, 2, K_>), grid, block, 0, stream, q,       \
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

int launch_splitk_reduce(const IGemmParams& p, int nsplit, hipStream_t stream) {
  SDMI_CHECK(nsplit >= 1 && nsplit <= 16 && p.N % 4 == 0 && p.splitk_ws, "splitk_reduce: bad arguments");
  if (p.mode == EPI_HEADS) {
    const int64_t total_h = (int64_t)p.M * (p.N / 4);
    ProfScope psh("splitk_reduce", 0.0, (double)p.M * p.N * (4.0 * nsplit + 2.0), stream);
    hipLaunchKernelGGL(splitk_reduce_heads_kernel, dim3((unsigned)((total_h + 255) / 256)), dim3(256), 0, stream, p, nsplit);
    SDMI_HIP_OK(hipGetLastError());
    return 0;
  }
  // rows per block: 32 (one row per thread) up to 256, doubling while the grid keeps >= 1024 blocks (the kernel is a
  // latency-bound stream of nsplit 16-byte loads per thread: it wants every CU busy); with GroupNorm statistics it must
  // also divide the sample's row count, so a strip never straddles two samples
  int rpb = 32;
  const int hw = p.Hout * p.Wout;
  if (p.gn_n > 0) SDMI_CHECK(hw % 32 == 0, "GroupNorm statistics need Hout*Wout % 32 == 0");
  while (rpb < 256 && (int64_t)cdiv(p.N / 4, 8) * cdiv(p.M, rpb * 2) >= 1024 && (p.gn_n == 0 || hw % (rpb * 2) == 0)) rpb *= 2;
  ProfScope ps2("splitk_reduce", 0.0, (double)p.M * p.N * 4.0 * (nsplit + 1), stream);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv(p.N / 4, 8), (unsigned)cdiv(p.M, rpb)), dim3(256), 0, stream, p, nsplit,
                     rpb);
  SDMI_HIP_OK(hipGetLastError());
  ps2.end();
  if (p.ln_out) return launch_layernorm(p.out_f32, p.ln_gamma, p.ln_beta, p.ln_out, p.M, p.N, p.ln_eps, stream);
  return 0;
}

// halo-staged 3x3 convolution: supported iff stride 1, pad 1, no upsampling, power-of-two width 16..64 and tiles of whole
// image rows that do not straddle samples
static bool halo_supported(const IGemmParams& p, int bm) {
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
  q.epi_vec = epi_vec_ok(p);
  SDMI_CHECK(splitk_ws_need(p, BM, BN, nsplit) <= p.splitk_ws_floats, "split-K workspace too small");
  q.magic_hw = div_magic(p.Hout * p.Wout);
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
  hipLaunchKernelGGL((conv3halo_kernel<BM, BN, WARPS_M, WARPS_N, NS>), grid, block, 0, stream, q, tiles_m, tiles_n, chunks_per_split);
  SDMI_HIP_OK(hipGetLastError());
  ps.end();
  if (nsplit > 1 && !q.splitk_fused) return launch_splitk_reduce(q, nsplit, stream);
  if (q.ln_out) return launch_layernorm(q.out_f32, q.ln_gamma, q.ln_beta, q.ln_out, q.M, q.N, q.ln_eps, stream);
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
};
static inline bool tile_is_halo(int t) { return t >= 14 && t <= 17; }
static inline bool tile_tn_even(int t) { return (kTiles[t].bn / kTiles[t].wn / 32) % 2 == 0; }

static int launch_tile(int tile, const IGemmParams& p, bool dma, int splitk, hipStream_t stream) {
  switch (tile) {            // tile ids: see include/sdmi.h (sdmi_igemm_desc.tile)
    case 0: return launch_cfg<128, 128, 2, 2, 2>(p, dma, splitk, stream);
    case 1: return launch_cfg<128, 64, 2, 2, 2>(p, dma, splitk, stream);
    case 2: return launch_cfg<64, 64, 2, 2, 2>(p, dma, splitk, stream);
    case 3: return launch_cfg<256, 128, 4, 2, 2>(p, dma, splitk, stream);
    case 4: return launch_cfg<128, 64, 2, 2, 3>(p, dma, splitk, stream);
    case 5: return launch_cfg<64, 64, 2, 2, 3>(p, dma, splitk, stream);
    case 6: return launch_cfg<256, 128, 4, 2, 3>(p, dma, splitk, stream);
    case 7: return launch_cfg<128, 128, 2, 2, 3>(p, dma, splitk, stream);
    case 8: return launch_cfg<64, 128, 2, 2, 3>(p, dma, splitk, stream);
    case 9: return launch_cfg<128, 128, 4, 2, 3>(p, dma, splitk, stream);
    case 10: return launch_cfg<64, 64, 2, 2, 4>(p, dma, splitk, stream);
    case 11: return launch_cfg<128, 256, 2, 4, 2>(p, dma, splitk, stream);
    case 12: return launch_cfg<64, 256, 1, 4, 3>(p, dma, splitk, stream);
    case 13: return launch_cfg<256, 64, 4, 1, 3>(p, dma, splitk, stream);
    case 14: return launch_halo_cfg<256, 64, 4, 2, 5>(p, splitk, stream);
    case 15: return launch_halo_cfg<256, 128, 4, 2, 3>(p, splitk, stream);
    case 16: return launch_halo_cfg<128, 64, 2, 2, 8>(p, splitk, stream);
    case 17: return launch_halo_cfg<128, 128, 2, 2, 5>(p, splitk, stream);
    case 18: return launch_cfg<64, 64, 2, 2, 8>(p, dma, splitk, stream);
    case 19: return launch_cfg<64, 128, 2, 2, 6>(p, dma, splitk, stream);
    case 20: return launch_cfg<128, 64, 2, 2, 6>(p, dma, splitk, stream);
    case 21: return launch_cfg<128, 128, 4, 2, 4>(p, dma, splitk, stream);
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
                 &k.splitk_req, &c.tile, &c.splitk, &c.us) >= 10 && c.tile >= 0 && c.tile < SDMI_NUM_TILES && c.splitk >= 1 &&
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
    // never chosen by any of the round-2 collection runs (profiles/tune_candidates_r02.txt): the 2-stage twins of the
    // 3-stage tiles, 128x128 / 256x128 with 2 stages, and the 64x256 / 256x64 4-wave tiles -- fewer candidates = more
    // samples per candidate
    if ((c.ns == 2 && t != 11 && !tile_is_halo(t)) || t == 12 || t == 13) continue;
    if (p.mode == EPI_GEGLU && !tile_tn_even(t)) continue;
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
  SDMI_CHECK(p.lda0 % 8 == 0 && (p.a1 == nullptr || p.lda1 % 8 == 0), "A row pitch must be a multiple of 8 halves");
  SDMI_CHECK(p.zero_page != nullptr, "zero page missing");
  SDMI_CHECK(p.M == p.B * p.Hout * p.Wout, "M != B * Hout * Wout");
  SDMI_CHECK(p.c1 == 0 || p.a1 != nullptr, "second A source missing");
  SDMI_CHECK(p.c2 == 0 || (p.a2 != nullptr && p.lda2 % 8 == 0), "third A source missing");
  SDMI_CHECK((p.c1 == 0 || p.lda1 == p.lda0) && (p.c2 == 0 || p.lda2 == p.lda0), "all A sources must share one row pitch");
  SDMI_CHECK(!p.up || (p.ksize == 3 && p.stride == 1), "upsample folding needs a 3x3 stride-1 conv");
  if (p.mode == EPI_GEGLU) SDMI_CHECK(p.N % 64 == 0 && p.out_f16 != nullptr, "GEGLU needs N % 64 == 0 and an fp16 output");
  if (p.ln_out)
    SDMI_CHECK(p.mode == EPI_PLAIN && p.out_f32 && p.ldo == p.N && p.N % 4 == 0 && p.N <= 2560 && p.ln_gamma && p.ln_beta,
               "LayerNorm post-op needs plain mode, an fp32 output with ldo == N <= 2560, gamma and beta");
  if (p.out_lo) SDMI_CHECK(p.mode == EPI_PLAIN && p.ldo % 4 == 0, "out_lo needs plain mode");
  if (p.gn_n) {
    SDMI_CHECK(p.mode == EPI_PLAIN && p.gn_n <= 2 && (p.Hout * p.Wout) % 32 == 0, "GroupNorm statistics need plain mode and Hout*Wout % 32 == 0");
    SDMI_CHECK(p.N % 4 == 0, "GroupNorm statistics: N % 4 == 0");
    for (int t = 0; t < p.gn_n; ++t)
      SDMI_CHECK(p.gn_acc[t] && p.gn_cpg[t] >= 2 && (p.gn_cbase[t] + p.N + p.gn_cpg[t] - 1) / p.gn_cpg[t] <= 32, "bad GroupNorm statistics target");
  }
  if (p.mode == EPI_HEADS) SDMI_CHECK(p.segC > 0 && p.dh > 0 && p.N % p.segC == 0 && p.N / p.segC <= 3, "bad head scatter");

  SDMI_CHECK((int64_t)p.B * p.Hin * p.Win * p.lda0 * 2 + (int64_t)(p.Win + 1) * p.lda0 * 2 < ((int64_t)1 << 31) - 65536 &&
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
  TuneKey tkey{p.M, p.N, p.K, p.ksize, p.stride, p.up, p.mode, splitk};
  int tcand = -1;
  if (tile < 0) {
    std::lock_guard<std::mutex> lk(g_tuner.mu);
    g_tuner.ensure_loaded();
    if (g_tuner.collecting) {
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
      if (it != g_tuner.table.end()) {
        const int tt = it->second.tile, sk = it->second.splitk;
        const bool halo = tile_is_halo(tt);
        const bool ws_ok = sk == 1 || (can_split && splitk_ws_need(p, kTiles[tt].bm, kTiles[tt].bn, sk) <= p.splitk_ws_floats);
        const bool split_ok = splitk != 0 ? (sk == splitk && ws_ok)                      // the caller pinned the split
                                          : (sk == 1 || (ws_ok && nkt / sk >= 4 && (halo || (sk != 5 && sk != 10)) &&
                                                         (!halo || (nkt / 9) % sk == 0)));
        if (split_ok && (p.mode != EPI_GEGLU || tile_tn_even(tt)) && (!halo || halo_supported(p, kTiles[tt].bm))) {
          tile = tt; splitk = sk;
        }
      }
    }
  }
  if (p.mode == EPI_GEGLU && tile >= 0 && !tile_tn_even(tile)) tile = 0;   // GEGLU pairs 32-col tiles inside a wave
  if (tile < 0) {
    // heuristic for shapes the table does not know (round-1 sweep: the many-block 64x64 tile except for >= 25 GFLOP)
    const double gflop = 2.0 * p.M * (double)p.N * p.K * 1e-9;
    if (p.mode == EPI_GEGLU) tile = 0;
    else if (p.mode == EPI_HEADS) tile = 2;
    else tile = gflop >= 25.0 ? 3 : 5;
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
