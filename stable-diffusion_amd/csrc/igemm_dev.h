// Device code and host helpers shared by the GEMM translation units (igemm.hip: the generic implicit-GEMM kernel, the split-K
// reduce, tile / split-K selection and the tuning table; conv3halo.hip: the halo-staged 3x3 convolutions): the k-tile size, the
// counted vmcnt waits, the epilogue every kernel ends with, and the launch-side policy helpers.  Everything here has internal
// linkage (each translation unit instantiates what it uses).
#pragma once
#include "common.h"
#include "prof.h"

namespace sdmi {
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

int launch_splitk_reduce(const IGemmParams& p, int nsplit, hipStream_t stream);
// the generic kernel's tile shapes, instantiated in three translation units (igemm_t0 / t1 / t2.hip)
int launch_generic_tile_g0(int tile, const IGemmParams& p, bool dma, int splitk, hipStream_t stream);
int launch_generic_tile_g1(int tile, const IGemmParams& p, bool dma, int splitk, hipStream_t stream);
int launch_generic_tile_g2(int tile, const IGemmParams& p, bool dma, int splitk, hipStream_t stream);

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
// 1 / (1 + p z) is the hardware reciprocal (v_rcp_f32, 1 ulp: the IEEE division sequence was 10 of the 37 VALU instructions per GEGLU
// output, and the GEGLU arithmetic is what the row-strip chain kernel's compute waves are bound by); the sign is one v_bfi_b32.
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * __expf(-z * z);
  const float erf_v = __builtin_copysignf(erf_abs, x);
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
// the row -> sample split m / (Hout * Wout): the dividend is a ROW INDEX (up to B * Hout * Wout), so m * d reaches B * (Hout*Wout)^2 --
// 1.4e12 > 2^40 for four 768 x 768 first-stage maps, where the 2^40 form returned sample B for the last pixels of the last sample.
// 2^48 form (magic = ceil(2^48 / d), div_magic_hw): exact for m * d < 2^48; the product m * magic stays below 2^64 because the
// quotient (a sample index) is < 2^16.
__device__ __forceinline__ int fast_div_hw(int m, unsigned long long magic) {
  return (int)(((unsigned long long)(unsigned)m * magic) >> 48);
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

// Row statistics of a GEMM that folds the LayerNorm of its input rows (IGemmParams::lnf_*): the producer left {sum, sum of
// squares} per 32-column block.  lnf_request: thread i < BM asks for the partials of row m at the very top of the kernel (plain
// loads, up to LNF_MAXP blocks = 1280 channels); lnf_finish folds them (block order, fp32: deterministic) once the operand
// prologue has been issued -- the first version fetched and folded (fp64) between the DMA issue and the first barrier, two
// dependent round trips + a double-precision rsqrt on every workgroup's critical path: +7 us on a 640-workgroup GEMM.
constexpr int LNF_MAXP = 40;                            // 1280 channels
__device__ __forceinline__ void lnf_request(const IGemmParams& p, int m, float2 (&pv)[LNF_MAXP]) {
  const float2* src = (const float2*)p.lnf_part + m;
#pragma unroll
  for (int j = 0; j < 20; ++j) pv[j] = src[(size_t)min(j, p.lnf_npart - 1) * p.M];
  if (p.lnf_npart > 20) {                                // (wave-uniform: the second half only for > 640 channels)
#pragma unroll
    for (int j = 20; j < LNF_MAXP; ++j) pv[j] = src[(size_t)min(j, p.lnf_npart - 1) * p.M];
  } else {
#pragma unroll
    for (int j = 20; j < LNF_MAXP; ++j) pv[j] = float2{0.f, 0.f};
  }
}
__device__ __forceinline__ void lnf_finish(const IGemmParams& p, const float2 (&pv)[LNF_MAXP], float* mean, float* rstd) {
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int j = 0; j < LNF_MAXP; ++j)
    if (j < p.lnf_npart) { s += pv[j].x; q += pv[j].y; }
  const float inv_c = 1.0f / (32.0f * (float)p.lnf_npart);
  const float mu = s * inv_c;
  const float var = fmaxf(q * inv_c - mu * mu, 0.f);
  *mean = mu;
  *rstd = 1.0f / sqrtf(var + p.lnf_eps);
}
// sum over aligned groups of 8 lanes, in every lane of the group (three DPP adds: quad_perm xor 1, xor 2, row_half_mirror)
__device__ __forceinline__ float sum8_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  return v;
}

// The epilogue shared by the GEMM kernels (generic implicit GEMM and the halo-staged 3x3 convolution): accumulators of the
// wave's TM x TN MFMA tiles -> bias / time-embedding row vector / residual / fp32 + fp16 (+ split-fp16 low half) stores,
// GEGLU, per-head q / k / v^T scatter, split-K slabs, GroupNorm statistics.  `smem` = the block's LDS (free at this point:
// every LDS-DMA of the block has landed and is no longer read), LDS_BYTES its size.
template <int BM, int BN, int WARPS_M, int WARPS_N, int LDS_BYTES>
__device__ __forceinline__ void igemm_epilogue(const IGemmParams& p_arg, f32x16 (&acc)[BM / WARPS_M / 32][BN / WARPS_N / 32],
                                               const int m0, const int n0, const int split, const int tile_m,
                                               const int tile_n, unsigned char* smem, const float lnf_mean = 0.f,
                                               const float lnf_rstd = 1.f) {
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
    // The slab reads are memory-side round trips (~2 us each way): SB splits' worth of them are requested back to back (up to 16
    // quads = 64 VGPRs in flight) and only then added, still in split order -- one round trip per batch instead of one per split
    // (the first version of this loop waited per split: +28 us on a 12-way split, profiles/splitk_fused_r02.txt).
    constexpr int QPS = TM * TN * 4;                       // 16-byte quads of one split's slab per lane
    constexpr int SB = QPS >= 16 ? 1 : 16 / QPS;           // splits per batch
    for (int s0 = 0; s0 < p.splitk; s0 += SB) {
      f32x4 v[SB][QPS];
#pragma unroll
      for (int ss = 0; ss < SB; ++ss) {
        const int off = (min(s0 + ss, p.splitk - 1) * (BM * BN) + tid * 4) * 4;     // (past the last split: a repeat that is not added)
#pragma unroll
        for (int e = 0; e < QPS; ++e)
          v[ss][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ws, off + e * (NT * 16), 0, SC1));
      }
#pragma unroll
      for (int ss = 0; ss < SB; ++ss) {
        if (s0 + ss < p.splitk) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * r4 + e] += v[ss][(i * TN + j) * 4 + r4][e];
        }
      }
    }
    __syncthreads();                       // smem[0] is reused below
  }
  if (p.splitk > 1 && !p.splitk_fused && p.slab_tiled) {
    // ---- unfused split-K, register-order slabs (IGemmParams::slab_tiled): this split's whole tile, 16 bytes per lane and store,
    // written through (sc1: the slab is read by another kernel on other XCDs; nothing here reads it back) -- a quarter of the
    // store instructions of the row-major slab, 1 KB runs per wave instruction.  splitk_reduce_tiled_kernel sums the splits in index
    // order: the same fp32 additions in the same order as the row-major reduction, i.e. the same output bits.
    constexpr int SC1 = 16;
    const int tiles_n_all = (p.N + BN - 1) / BN, ntiles = ((p.M + BM - 1) / BM) * tiles_n_all;
    const int tile_lin = tile_m * tiles_n_all + tile_n;
    const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.splitk_ws + ((size_t)split * ntiles + tile_lin) * (size_t)(BM * BN)), 0, BM * BN * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const f32x4 v = {acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ws, (tid * 4 + ((i * TN + j) * 4 + r4) * (NT * 4)) * 4, 0, SC1);
        }
    return;
  }
  if (p.lnf_part) {
    // ---- LayerNorm of the A rows folded into this GEMM (IGemmParams::lnf_*): thread i < BM brought {mean, rstd} of row m0 + i
    // (lnf_row_stats, kernel prologue); through an LDS table every lane picks up the rows of its accumulator registers ----
    __syncthreads();                                       // every wave's LDS-DMA has landed, nobody reads the tiles any more
    float2* const tab = (float2*)smem;
    if (tid < BM) tab[tid] = float2{lnf_mean, lnf_rstd};
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = min(nw + j * 32 + l31, p.N - 1);
      const float cs = p.lnf_cs[n], dn = p.lnf_d[n];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float2 mr = tab[wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg];
          acc[i][j][r] = fmaf(mr.y, acc[i][j][r] - mr.x * cs, dn);
        }
    }
    __syncthreads();                                       // the table is read: the LDS is free for the slabs / statistics below
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
      f32x4 g4 = {1.f, 1.f, 1.f, 1.f};                     // scale of the fp16 copy (a LayerNorm's gamma when its consumer folds it)
      if (p.f16_scale) g4 = *(const f32x4*)(p.f16_scale + nw + c4);
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
          if (p.out_f32) SDMI_ST_F32X4(p.out_f32, ro, v);
          const f32x4 vs = v * g4;                         // (g4 = 1 without a scale: exact)
          const f16x4 h = {(f16)vs[0], (f16)vs[1], (f16)vs[2], (f16)vs[3]};
          if (p.out_f16) SDMI_ST_F16X4(p.out_f16, ro, h);
          if (p.out_lo)
            *(f16x4*)(p.out_lo + ro) = f16x4{(f16)(v[0] - (float)h[0]), (f16)(v[1] - (float)h[1]), (f16)(v[2] - (float)h[2]),
                                            (f16)(v[3] - (float)h[3])};
          if (p.lnp_out) {
            // row statistics of the finished values for the LayerNorm that reads them: {sum, sum of squares} over this row's
            // 32-column block = the 8 lanes that hold it (fixed DPP add tree: deterministic), one float2 per (block, row)
            float s1 = (v[0] + v[1]) + (v[2] + v[3]);
            float s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            s1 = sum8_dpp(s1); s2 = sum8_dpp(s2);
            if ((lane & 7) == 0)
              *(float2*)(p.lnp_out + 2 * ((size_t)((nw + c4) >> 5) * p.M + (size_t)(mw + i * 32 + row))) = float2{s1, s2};
          }
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
              SDMI_ST(f16, p.out_f16 + (size_t)m * p.ldo + oc, (f16)(val * gelu_erf(gate)));
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
              SDMI_ST(f16x4, dst + (((size_t)b * p.heads + head) * p.ntok + tok) * p.dh + dd, (f16x4{(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]}));
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

static unsigned long long div_magic(int d) {      // ceil(2^40 / d), see fast_div
  const unsigned long long one = 1ull << 40;
  return (one + (unsigned long long)d - 1) / (unsigned long long)d;
}
static unsigned long long div_magic_hw(int d) {   // ceil(2^48 / d), see fast_div_hw (IGemmParams::magic_hw)
  const unsigned long long one = 1ull << 48;
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
// into the GEMM (see igemm_epilogue) or the slabs are tiled (IGemmParams::slab_tiled), [split][M][N] otherwise
static bool splitk_fusable(const IGemmParams& p, int bm, int bn) {
  // default off: same-box A/B (profiles/splitk_fused_r02.txt) has the separate reduce kernel ahead, 3.23 vs 3.16 images/s
#ifdef SDMI_EXPERIMENTS
  static const int env_fused = env_int("SDMI_SPLITK_FUSED", 0);
#else
  constexpr int env_fused = 0;
#endif
  return env_fused && p.splitk_cnt && (int64_t)cdiv(p.M, bm) * cdiv(p.N, bn) <= p.splitk_cnt_ints;
}
// rows per block of splitk_reduce_kernel: 32 (one row per thread) up to 256, doubling while the grid keeps >= 1024 blocks
static int reduce_rows_per_block(const IGemmParams& p) {
  int rpb = 32;
  const int hw = p.Hout * p.Wout;
  while (rpb < 256 && (int64_t)cdiv(p.N / 4, 8) * cdiv(p.M, rpb * 2) >= 1024 && (p.gn_n == 0 || hw % (rpb * 2) == 0)) rpb *= 2;
  return rpb;
}
// may the reduction of this split GEMM apply the consuming GroupNorm itself (IGemmParams::pgn_*)?  Returns the quads per thread
// of splitk_reduce_gn_kernel (1 / 3 / 5) or 0.  Opt-in (SDMI_REDUCE_GN=1): bit-identical and 15 launches fewer, but one workgroup
// per (sample, group) = 64 workgroups read what the plain reduction spreads over the chip -- same-box A/Bs (profiles/splitk_slabs_r04.txt):
// -0.01 ms per UNet call against the row-major reduce + GroupNorm-apply launches on a fast box, +0.04 ... +0.17 ms against the
// register-order reduce + GroupNorm-apply launches (mid / slow box: 24 us per launch there).
static int reduce_gn_maxq(const IGemmParams& p, int nsplit) {
#ifndef SDMI_EXPERIMENTS
  return 0;                                         // (product build: splitk_reduce_gn_kernel is not compiled in)
#endif
  const int on = env_int("SDMI_REDUCE_GN", 0);      // (read per launch: the tests flip it between two forwards)
  const int hw = p.Hout * p.Wout;
  if (!on || !p.pgn_out || !p.pgn_gamma || !p.pgn_beta || p.mode != EPI_PLAIN || nsplit < 2 || nsplit > 16) return 0;
  if (p.N % 128 || p.M != p.B * hw || p.out_f16 || p.out_lo || p.ln_out || p.lnp_out) return 0;
  // Taken only where the two-launch path gets that GroupNorm's statistics from this very reduction (the executor attached it as the
  // one statistics target: Hout*Wout % 32 == 0, ...): the result is then the same bits (see the kernel), i.e. this is a launch-count
  // optimisation with no numerical footprint.  Elsewhere (maps of < 32 pixels) the statistics kernel + apply launches stay.
  if (p.gn_n != 1 || p.gn_cbase[0] != 0 || p.gn_cpg[0] != p.N / 32 || hw % 32) return 0;
  if (p.pgn_keep_f32 && (!p.out_f32 || p.ldo % 4)) return 0;
  if (p.ldr % 4 || p.ld_rowvec % 4) return 0;
  if (reduce_rows_per_block(p) != 32) return 0;         // (the statistics are then the two-launch path's, bit for bit: see the kernel)
  const int64_t quads = (int64_t)hw * (p.N / 128);
  return quads <= 1024 ? 1 : quads <= 3 * 1024 ? 3 : quads <= 5 * 1024 ? 5 : 0;
}
// Register-order ("tiled") slabs for the unfused split-K (IGemmParams::slab_tiled): every tile shape of the generic / halo /
// split-fp16 kernels (power-of-two geometry; not the five-wave tile), plain or per-head epilogue.  With GroupNorm statistics only
// where the row-major reduction forms one partial per quad (32 rows per block): the statistics words are then the same integers.
// The reduction that applies the GroupNorm itself reads either layout.  SDMI_SLAB_TILED=0: never (A/B).
static bool slab_tiled_ok(const IGemmParams& p, int bm, int bn, int nsplit) {
  const int on = env_int("SDMI_SLAB_TILED", 1);      // (read per launch: the tests flip it)
  if (!on || nsplit <= 1 || splitk_fusable(p, bm, bn) || p.mode == EPI_GEGLU || p.N % 4) return false;
  if ((bm & (bm - 1)) || (bn & (bn - 1))) return false;
  if (p.mode == EPI_PLAIN && p.gn_n > 0 && reduce_rows_per_block(p) != 32) return false;
  // the tiled branch of igemm_epilogue returns before the LayerNorm-fold correction, and splitk_reduce_tiled_kernel knows neither
  // f16_scale nor lnp_out: those launches are pinned to split 1 by launch_igemm -- keep the dependency here as well
  if (p.lnf_part || p.lnp_out || p.f16_scale) return false;
  // whole padded tiles must fit the workspace; a caller that sized it as splitk * M * N (the row-major contract) keeps that layout
  if ((int64_t)nsplit * cdiv(p.M, bm) * bm * cdiv(p.N, bn) * bn > p.splitk_ws_floats) return false;
  return true;
}
static void slab_layout(IGemmParams& q, int bm, int bn, int wm, int wn, int nsplit) {
  q.slab_tiled = slab_tiled_ok(q, bm, bn, nsplit) ? 1 : 0;
  q.slab_bm = bm; q.slab_bn = bn; q.slab_wm = wm; q.slab_wn = wn;
  auto lg2 = [](int v) { int s = 0; while ((1 << s) < v) ++s; return s; };
  q.slab_sh_qpt = lg2(bm * bn / 4); q.slab_sh_nt = lg2(wm * wn * 64); q.slab_sh_tn = lg2(bn / wn / 32); q.slab_sh_wn = lg2(wn);
  // (power-of-two geometry: slab_tiled_ok; the threads per block and waves per row of every such tile are powers of two as well)
  if (q.slab_tiled && (((wm * wn) & (wm * wn - 1)) || (wn & (wn - 1)))) q.slab_tiled = 0;
}
static int64_t splitk_ws_need(const IGemmParams& p, int bm, int bn, int nsplit) {
  if (nsplit <= 1) return 0;
  if (splitk_fusable(p, bm, bn) || slab_tiled_ok(p, bm, bn, nsplit)) return (int64_t)nsplit * cdiv(p.M, bm) * bm * cdiv(p.N, bn) * bn;
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

}  // namespace

// the five-wave 64 x 160 tile (igemm5.hip): tile id 22
int launch_igemm5_tile(int tile, const IGemmParams& p, int splitk, hipStream_t stream);
// halo-staged 3x3 convolution tiles (conv3halo.hip)
bool halo_supported(const IGemmParams& p, int bm);
int launch_halo_tile(int tile, const IGemmParams& p, int splitk, hipStream_t stream);
// split-fp16 dense GEMM family (gemm_split16.hip): the tile ids of kTiles it instantiates
bool split16_tile_supported(int tile);

int launch_split16_tile(int tile, const IGemmParams& p, int splitk, hipStream_t stream);
// ... with GroupNorm(32) + SiLU of the fp32 input folded into the staging (IGemmParams::xf0 / gn_in_*)
bool halo_gn_supported(const IGemmParams& p, int bm);
int launch_halo_gn_tile(int tile, const IGemmParams& p, int splitk, hipStream_t stream);

}  // namespace sdmi
