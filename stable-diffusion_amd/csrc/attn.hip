// Fused softmax(Q K^T * scale) V for the SpatialTransformer self- and cross-attention
// (reference: CrossAttention.forward, ldm/modules/attention.py:170-193).
// The reference materialises sim[(b h), N, Nkv]; this kernel keeps it in registers (online softmax).
//
// gfx950 design:
//   * one wave owns 32 query rows; NW waves per workgroup share the K / V^T tiles (64 keys) staged in LDS,
//     double buffered, register-staged (loads for tile t+1 are in flight while tile t is computed).
//   * "swapped" product S^T = K Q^T on v_mfma_f32_32x32x16_f16: the accumulator layout then gives every lane
//     one query column (q = lane & 31) and 16 of the 32 key rows, so the row max / row sum are in-lane
//     reductions plus one cross-half exchange, and the O rescale factor is a per-lane scalar.
//   * O^T = V^T P^T: the fp16-rounded probabilities are fed back as the MFMA B operand *straight from the
//     accumulator registers*; the k-slot order of the contraction is permuted to match (lane half g, slot e
//     <-> key 16*s + 8*(e>>2) + 4*g + (e&3)) and V^T fragments are read from LDS in that same order with
//     two ds_read_b64 -- no cross-lane shuffles, no P round trip through LDS.
//   * V is consumed transposed ([d][key], key-contiguous) -- the QKV projection epilogue (igemm EPI_HEADS)
//     writes it that way, so no transpose happens here.
//   * fp32 scores, max, sum and output accumulation; exp via v_exp_f32 on log2e-prescaled scores.
#include <type_traits>

#include "common.h"
#include "prof.h"

namespace sdmi {
namespace {

constexpr int KVT = 64;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_dma() {          // counted s_waitcnt vmcnt(N): the immediate must be a literal
  static_assert(N >= 0 && N <= 44, "add the literal");
  switch (N) {
#define SDMI_VM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break
    SDMI_VM(0); SDMI_VM(1); SDMI_VM(2); SDMI_VM(3); SDMI_VM(4); SDMI_VM(5); SDMI_VM(6); SDMI_VM(7); SDMI_VM(8); SDMI_VM(9);
    SDMI_VM(10); SDMI_VM(11); SDMI_VM(12); SDMI_VM(13); SDMI_VM(14); SDMI_VM(15); SDMI_VM(16); SDMI_VM(17); SDMI_VM(18);
    SDMI_VM(19); SDMI_VM(20); SDMI_VM(21); SDMI_VM(22); SDMI_VM(23); SDMI_VM(24); SDMI_VM(25); SDMI_VM(26); SDMI_VM(27);
    SDMI_VM(28); SDMI_VM(29); SDMI_VM(30); SDMI_VM(31); SDMI_VM(32); SDMI_VM(33); SDMI_VM(34); SDMI_VM(35); SDMI_VM(36);
    SDMI_VM(37); SDMI_VM(38); SDMI_VM(39); SDMI_VM(40); SDMI_VM(41); SDMI_VM(42); SDMI_VM(43); SDMI_VM(44);
#undef SDMI_VM
  }
}

template <int D, int NW, bool CAUSAL>
__global__ void __launch_bounds__(NW * 64) attn_kernel(const AttnParams p) {
  constexpr int DKS = (D + 15) / 16;   // k-steps of 16 over the head dim (QK^T)
  constexpr int DVT = (D + 31) / 32;   // 32-row tiles over the head dim (PV)
  constexpr int NT = NW * 64;
  constexpr int KSTRIDE = DKS * 32 + 16;          // bytes; odd multiple of 16 -> conflict-free ds_read_b128
  constexpr int VSTRIDE = KVT * 2 + 8;            // bytes; 34 dwords -> conflict-free ds_read_b64
  constexpr int KBYTES = KVT * KSTRIDE;
  constexpr int VBYTES = DVT * 32 * VSTRIDE;
  constexpr int STAGE = KBYTES + VBYTES;
  constexpr int KCH = KVT * DKS * 2;              // 16-B chunks in a K tile
  constexpr int VCH = DVT * 32 * (KVT / 8);       // 16-B chunks in a V^T tile
  constexpr int KIT = (KCH + NT - 1) / NT;
  constexpr int VIT = (VCH + NT - 1) / NT;

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lg = lane >> 5;
  const int bh = blockIdx.y;
  const int q0 = blockIdx.x * (32 * NW) + wave * 32;
  const f16* Qg = p.q + (size_t)bh * p.nq * D;
  const f16* Kg = p.k + (size_t)bh * p.nkv * D;
  const f16* Vg = p.vt + (size_t)bh * D * p.nkv_pad;

  // Q^T fragments (MFMA B operand): lane (q = l31, g = lg) holds Q[q][16*ks + 8*g .. +8]
  f16x8 qf[DKS];
#pragma unroll
  for (int ks = 0; ks < DKS; ++ks) {
    const int dcol = ks * 16 + lg * 8;
    f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (q0 + l31 < p.nq && dcol < D) v = *(const f16x8*)(Qg + (size_t)(q0 + l31) * D + dcol);
    qf[ks] = v;
  }

  f16x8 kreg[KIT], vreg[VIT];
  auto load_tile = [&](int t) {
    const int kv0 = t * KVT;
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int c = tid + it * NT;
      f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (c < KCH) {
        const int row = c / (DKS * 2), col = c - row * (DKS * 2);
        if (kv0 + row < p.nkv && col * 8 < D) v = *(const f16x8*)(Kg + (size_t)(kv0 + row) * D + col * 8);
      }
      kreg[it] = v;
    }
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int c = tid + it * NT;
      f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (c < VCH) {
        const int row = c / (KVT / 8), col = c - row * (KVT / 8);
        if (row < D && kv0 + col * 8 < p.nkv_pad) v = *(const f16x8*)(Vg + (size_t)row * p.nkv_pad + kv0 + col * 8);
      }
      vreg[it] = v;
    }
  };
  auto store_tile = [&](int stage) {
    unsigned char* Ks = smem + stage * STAGE;
    unsigned char* Vs = Ks + KBYTES;
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int c = tid + it * NT;
      if (c < KCH) {
        const int row = c / (DKS * 2), col = c - row * (DKS * 2);
        *(f16x8*)(Ks + row * KSTRIDE + col * 16) = kreg[it];
      }
    }
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int c = tid + it * NT;
      if (c < VCH) {
        const int row = c / (KVT / 8), col = c - row * (KVT / 8);
        unsigned char* d = Vs + row * VSTRIDE + col * 16;
        const f16x8 v = vreg[it];
        *(f16x4*)(d) = f16x4{v[0], v[1], v[2], v[3]};
        *(f16x4*)(d + 8) = f16x4{v[4], v[5], v[6], v[7]};
      }
    }
  };

  f32x16 o[DVT];
#pragma unroll
  for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float sc = p.scale * 1.4426950408889634f;   // scores are compared / exponentiated in log2 units

  const int nt = (p.nkv + KVT - 1) / KVT;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    const bool more = (t + 1 < nt);
    if (more) load_tile(t + 1);
    const unsigned char* Ks = smem + cur * STAGE;
    const unsigned char* Vs = Ks + KBYTES;

    // ---- S^T = K Q^T (two 32-key blocks) ----
    f32x16 s[KVT / 32];
#pragma unroll
    for (int kvb = 0; kvb < KVT / 32; ++kvb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kvb][r] = 0.f;
      const unsigned char* kp = Ks + (kvb * 32 + l31) * KSTRIDE + lg * 16;
#pragma unroll
      for (int ks = 0; ks < DKS; ++ks) {
        const f16x8 a = *(const f16x8*)(kp + ks * 32);
        s[kvb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[ks], s[kvb], 0, 0, 0);
      }
    }
    // ---- mask keys beyond nkv (last tile only; every tile when causal) ----
    // A real, wave-uniform branch: if-converted into 64 compares + selects per tile it was a third of the loop's
    // instructions (round-1 ISA) although only the last tile of a non-causal call has anything to mask.
    const int kv0 = t * KVT;
    if (CAUSAL || kv0 + KVT > p.nkv) {
      asm volatile("; masked tile" ::: "memory");                // (keeps the compiler from speculating the block)
      const int qlim = CAUSAL ? (q0 + l31) : 0x7fffffff;         // causal: this lane's query sees keys <= its own index
#pragma unroll
      for (int kvb = 0; kvb < KVT / 32; ++kvb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + kvb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
          if (kv >= p.nkv || kv > qlim) s[kvb][r] = -1e30f;
        }
    }
    // ---- online softmax (per query = per lane column; halves lg = 0/1 hold disjoint keys) ----
    float mx = -1e30f;
#pragma unroll
    for (int kvb = 0; kvb < KVT / 32; ++kvb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kvb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx * sc);
    // The running maximum stops growing after the first few tiles: when it did not move for ANY query of this wave the
    // rescale factor is exactly 2^0 = 1 for every lane, and skipping the multiplies is bit-identical.
    if (__any(m_new > m_run)) {
      asm volatile("; rescale" ::: "memory");
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    float psum = 0.f;
#pragma unroll
    for (int kvb = 0; kvb < KVT / 32; ++kvb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(s[kvb][r], sc, -m_run));
        s[kvb][r] = pv;
        psum += pv;
      }
    l_run += psum;

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int kvb = 0; kvb < KVT / 32; ++kvb) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        f16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (f16)s[kvb][8 * s2 + e];
        const unsigned char* vp = Vs + l31 * VSTRIDE + (kvb * 32 + 16 * s2 + 4 * lg) * 2;
#pragma unroll
        for (int dt = 0; dt < DVT; ++dt) {
          const f16x4 lo = *(const f16x4*)(vp + dt * 32 * VSTRIDE);
          const f16x4 hi = *(const f16x4*)(vp + dt * 32 * VSTRIDE + 16);
          const f16x8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pf, o[dt], 0, 0, 0);
        }
      }
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- normalise and store: O[b][q][head*D + dd] ----
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  const int q = q0 + l31;
  if (q < p.nq) {
    const int b = bh / p.heads, head = bh - b * p.heads;
    f16* orow = p.out + ((size_t)b * p.nq + q) * ((size_t)p.heads * D) + (size_t)head * D;
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int dd = dt * 32 + 8 * r4 + 4 * lg;
        if (dd < D) {
          f16x4 v = {(f16)(o[dt][r4 * 4 + 0] * inv), (f16)(o[dt][r4 * 4 + 1] * inv), (f16)(o[dt][r4 * 4 + 2] * inv),
                     (f16)(o[dt][r4 * 4 + 3] * inv)};
          *(f16x4*)(orow + dd) = v;
        }
      }
  }
}

// Workgroups are dealt to the 8 XCDs round-robin by linear id, and each XCD has its own 4 MB L2.  With the natural
// (q tile, head) numbering every XCD sees every head's K / V^T (SD-v1 64x64 level: 16 heads x 656 KB = 10.5 MB, more than
// one L2), and the tiles stream from the fabric instead.  Renumber so that one XCD owns whole heads.
__device__ __forceinline__ void head_of_block(int& bh, int& qt) {
  const unsigned nqt = gridDim.x, total = gridDim.x * gridDim.y;
  unsigned lin = blockIdx.x + nqt * blockIdx.y;
  if ((total & 7) == 0) lin = (lin & 7) * (total >> 3) + (lin >> 3);
  bh = (int)(lin / nqt);
  qt = (int)(lin - (unsigned)bh * nqt);
}

// max / sum over the two 32-lane halves of the wave, in every lane.  v_permlane32_swap is a VALU instruction; the
// ds_bpermute that __shfl_xor(x, 32) turns into queues behind the wave's outstanding LDS fragment reads (and its
// lgkmcnt(0) waits for them).  Inline asm: this compiler returns the first result twice from the builtin.
__device__ __forceinline__ void swap_halves(float& a, float& b) {      // a.hi <-> b.lo
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float max_across_halves(float x) {
  float a = x, b = x;
  swap_halves(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float sum_across_halves(float x) {
  float a = x, b = x;
  swap_halves(a, b);
  return a + b;
}

// ---- LDS-DMA variant ---------------------------------------------------------------------------------------------------
// Same algorithm as attn_kernel, re-cut around what actually bounds it: the per-score VALU work (v_exp_f32 is quarter rate:
// 32 scores x 16 cycles per 64-key tile and wave exceed the 14 MFMAs x 32 cycles of d = 40), not the matrix cores.
//   * the K / V^T tiles arrive by LDS-DMA into an NS-deep ring: no VALU / VGPRs spent on staging, NS - 1 tiles in flight;
//   * the keys of a tile sit in LDS in a permuted order (bits 2 and 3 of the row swapped), which makes the eight keys a lane
//     owns in a P^T fragment CONTIGUOUS in the V^T row: one ds_read_b128 per fragment instead of two b64 + a repack;
//   * scale / subtract and the row sum run as packed fp32 (v_pk_fma_f32, v_pk_add_f32).  The register-staged
// loop above spends ~0.4 us of MFMA + softmax per 64-key tile and then waits for the NEXT tile's global loads, which were
// issued only one tile earlier: every iteration exposes the L2 latency (wait_any 44 % of the wave cycles, MFMA busy 26 %,
// profiles/pmc_sq_by_kernel_r02.txt).  Here NS - 1 tiles are in flight, retired by a counted vmcnt.
//   LDS image of a stage (rows of 128 B, 16-byte chunks XOR-swizzled with (row >> 1) & 7 on the DMA *source* side, like
//   the GEMM tiles): K as ceil(D / 64) sub-tiles [64 keys][64 halves of d]; V^T as DVT sub-tiles [32 d-rows][64 keys].
//   Rows / columns past the tensors are out-of-range buffer offsets (zeros) or finite neighbouring data that the zero pad
//   of Q, the score mask and the zero columns of V^T (nkv .. nkv_pad) neutralise.
// ABL != 0: timing-only ablations (wrong results) that price one ingredient of the loop at a time (tools/attn_ablate.py):
//   1 no per-tile wait + barrier, 2 no DMA issue, 3 exp2 replaced by a move, 4 no PV MFMAs, 5 no QK^T MFMAs,
//   6 K / V^T fragments read from LDS once (first tile) only
// KVS = 2 (round 6): the KEY range is split inside the workgroup.  NW = 16 waves: waves 0..7 and 8..15 serve the SAME 256 queries, group g over
// keys [g * nkv / 2, (g + 1) * nkv / 2) with its own LDS-DMA ring; at the end group 1 hands {O^T, running maximum} to group 0 through LDS and
// group 0 merges the two softmax partials (O = 2^(m0 - m) O0 + 2^(m1 - m) O1; the denominator is a row of O^T, see ONES) and stores.  Why: with one
// 8-wave workgroup per CU a SIMD holds two waves and 39 % of their cycles are waits (ds_read / barrier / vmcnt, profiles/pmc_by_kernel_r06.txt)
// that nothing overlaps; 4 waves per SIMD do overlap them, and a grid of 256 workgroups cannot be doubled along the queries without
// re-staging K / V per workgroup (round 2: 4-wave workgroups at two per CU, 107 us).  The per-wave instruction stream is unchanged.
// ROT (KVS = 2 only): key group 1 runs the three blocks of a tile ROTATED -- softmax(t), P V(t), then Q K^T of tile t + 1 -- while group 0 keeps
// Q K^T(t), softmax(t), P V(t): the four waves of a SIMD no longer all want the VALU (the softmax: 60 % of a tile) in the same phase.  One barrier
// per tile as before; group 1 needs tile t + 1 landed one interval earlier (one LDS-DMA tile less in flight for that group).  Same values, same
// per-wave arithmetic: bit-identical to the unrotated key split.
// TPB = 2 (round 6): ONE barrier per TWO key tiles.  The ring of four stages holds the pair being read and the pair in flight; the waves of a
// group may drift up to two tiles apart, so the four waves of a SIMD stop marching through matrix and VALU phases in lock-step.  Same values, same
// per-wave arithmetic, same bits.
template <int D, int NW, int NS, bool CAUSAL, int ABL = 0, int KVS = 1, bool ROT = false, int TPB = 1>
__global__ void __launch_bounds__(NW * 64) attn_dma_kernel(const AttnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int DKS = (D + 15) / 16;   // k-steps of 16 over the head dim (QK^T)
  constexpr int DVT = (D + 31) / 32;   // 32-row tiles over the head dim (PV)
  constexpr int NCH = (D + 63) / 64;   // 64-half chunks of a K row
  constexpr int KROWS = NCH * 64, VROWS = DVT * 32;
  constexpr int STAGE = (KROWS + VROWS) * 128;
  // The V^T tile has VROWS = 32 * DVT rows in LDS; when D is not a multiple of 32 (d = 40, 80) the rows D .. VROWS-1 are
  // padding.  They are written once (never DMA'd), row D with ONES: O^T row D = sum_k P[q][k] is then the softmax
  // denominator, accumulated by the MFMAs (from the same fp16-rounded P as the numerator, rescaled with O) -- the loop
  // carries no row-sum instructions.
  constexpr bool ONES = VROWS > D;
  constexpr int PK = NCH * 8, PV = ONES ? D / 8 : DVT * 4, PT = PK + PV;   // DMA pieces (8 rows x 128 B each) per tile
  static_assert(D % 8 == 0, "a DMA piece is 8 rows");
  constexpr int NWQ = NW / KVS;                                    // query waves (waves per key group)
  static_assert(KVS == 1 || (KVS == 2 && !CAUSAL && NW % 2 == 0), "key split: two groups, no causal mask");
  static_assert(!ROT || (KVS == 2 && NS >= 4 && ABL == 0), "rotated key group: needs the key split and a ring of four stages");
  static_assert(TPB == 1 || (TPB == 2 && NS == 4 && !ROT && ABL == 0), "two tiles per barrier: a ring of exactly four stages");
  constexpr int PPW = (PT + NWQ - 1) / NWQ;                        // per wave (the surplus re-issues the last piece)
  static_assert(NS >= 2 && KVS * NS * STAGE <= 160 * 1024, "LDS budget");
  static_assert(KVS == 1 || NWQ * (DVT * 16 + 2) * 64 * 4 <= KVS * NS * STAGE, "the hand-over of the key groups lives in the (drained) rings");

  __shared__ __attribute__((aligned(16))) unsigned char smem_all[KVS * NS * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = KVS == 1 ? 0 : wave_all / NWQ;                   // key group of this wave
  const int wave = KVS == 1 ? wave_all : wave_all - grp * NWQ;     // query slice inside the workgroup
  unsigned char* const smem = smem_all + grp * (NS * STAGE);       // this group's ring
  const int l31 = lane & 31, lg = lane >> 5;
  int bh, qt;
  head_of_block(bh, qt);
  const int q0 = qt * (32 * NWQ) + wave * 32;
  const f16* Qg = p.q + (size_t)bh * p.nq * D;
  constexpr int OOB = (int)0x80000000;
  const __amdgpu_buffer_rsrc_t rsrc_k =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.k + (size_t)bh * p.nkv * D), 0, p.nkv * D * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_v =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.vt + (size_t)bh * D * p.nkv_pad), 0, D * p.nkv_pad * 2, 0x00020000);

  // Q^T fragments (MFMA B operand): lane (q = l31, g = lg) holds Q[q][16*ks + 8*g .. +8]; zero beyond D / nq
  f16x8 qf[DKS];
#pragma unroll
  for (int ks = 0; ks < DKS; ++ks) {
    const int dcol = ks * 16 + lg * 8;
    f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (q0 + l31 < p.nq && dcol < D) v = *(const f16x8*)(Qg + (size_t)(q0 + l31) * D + dcol);
    qf[ks] = v;
  }

  // this wave's DMA pieces: q = wave + j * NW (clamped): per-lane byte offset at tile 0, per-tile increment, LDS row
  int pv_off[PPW], pv_step[PPW], pv_row[PPW];
  bool pv_isk[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int q = min(wave + j * NWQ, PT - 1);
    const int r8 = lane >> 3, cp = lane & 7;
    if (q < PK) {
      const int c = q >> 3, row = (q & 7) * 8 + r8;               // LDS row; it holds key perm(row) of the tile
      const int key = (row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1);
      const int gch = cp ^ ((row >> 1) & 7);
      pv_off[j] = key * (D * 2) + c * 128 + gch * 16;
      pv_step[j] = KVT * D * 2;
      pv_row[j] = c * 64 + (q & 7) * 8;
      pv_isk[j] = true;
    } else {
      const int qv = q - PK, dt = qv >> 2, row = (qv & 3) * 8 + r8;       // row inside the 32-row sub-tile
      const int d = dt * 32 + row;
      const int gch = cp ^ ((row >> 1) & 7);
      pv_off[j] = d < D ? d * (p.nkv_pad * 2) + gch * 16 : OOB;
      pv_step[j] = d < D ? KVT * 2 : 0;
      pv_row[j] = KROWS + dt * 32 + (qv & 3) * 8;
      pv_isk[j] = false;
    }
  }
  auto issue_tile = [&](int stage) {      // advances the per-lane offsets: call once per tile, in order
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      auto dst = (__attribute__((address_space(3))) void*)(smem + stage * STAGE + pv_row[j] * 128);
      if (pv_isk[j]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_k, dst, 16, pv_off[j], 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_v, dst, 16, pv_off[j], 0, 0, 0);
      // (offsets past the tensor end stay out of range: they only grow; the OOB marker has a zero step)
      pv_off[j] += pv_step[j];
    }
  };

  f32x16 o[DVT];
#pragma unroll
  for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  if (ONES) {
    for (int e = tid; e < KVS * NS * (VROWS - D) * 8; e += NW * 64) {     // 16-byte chunks of the padding rows of every stage (of every ring)
      const int st = e / ((VROWS - D) * 8), rc = e - st * ((VROWS - D) * 8), row = D + (rc >> 3);
      const unsigned one2 = row == D ? 0x3C003C00u : 0u;                   // fp16 1.0 pairs
      *(u32x4*)(smem_all + st * STAGE + (KROWS + row) * 128 + (rc & 7) * 16) = u32x4{one2, one2, one2, one2};
    }
  }
  const float sc = p.scale * 1.4426950408889634f;   // scores are compared / exponentiated in log2 units

  const int nt_all = (p.nkv + KVT - 1) / KVT;
  const int nt = nt_all / KVS;                                 // (the launcher takes KVS = 2 only for an even tile count)
  const int t_first = grp * nt;                                // this group's first key tile
  if (KVS > 1) {
#pragma unroll
    for (int j = 0; j < PPW; ++j) pv_off[j] += t_first * pv_step[j];       // (the out-of-range marker has a zero step)
  }
#pragma unroll
  for (int s2 = 0; s2 < (TPB == 2 ? 2 : NS - 1); ++s2) issue_tile(s2);          // (tiles past nt read out of range: zeros, never consumed)
  const bool rot = ROT && grp == 1;                           // (wave-uniform)
  if (TPB == 2) wait_dma<0>();
  else if (rot) wait_dma<PPW*(NS - 3 > 0 ? NS - 3 : 0)>(); else wait_dma<PPW*(NS - 2)>();
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  const int ksw = (l31 >> 1) & 7;
  int cur = 0, nxt = NS - 1;
  f32x16 s[KVT / 32];
  // S^T = K Q^T of the tile in ring stage `stage` (two 32-key blocks)
  auto qk_tile = [&](int stage) {
    const unsigned char* Kq = smem + (ABL == 6 ? 0 : stage) * STAGE;
    // (round 6: the first k-step takes the constant 0 as its C operand -- an inline constant of the MFMA -- instead of 32 v_mov_b32 per tile that
    // zeroed the score accumulators: 18 % of the loop's VALU issue at d = 40; the same sums, the same bits)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kvb = 0; kvb < KVT / 32; ++kvb) {
      if (ABL == 5) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kvb][r] = 0.f;
      }
#pragma unroll
      for (int ks = 0; ks < DKS; ++ks) {
        const unsigned char* kp = Kq + ((ks >> 2) * 64 + kvb * 32 + l31) * 128 + ((((ks & 3) * 2 + lg) ^ ksw) << 4);
        const f16x8 a = *(const f16x8*)kp;
        if (ABL == 5) { s[kvb][ks] += (float)a[0]; continue; }
        s[kvb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[ks], ks == 0 ? zero16 : s[kvb], 0, 0, 0);
      }
    }
  };
  if (rot) qk_tile(0);                                        // (group 1 enters the loop with tile 0's scores)
  // Two waves share a SIMD's VALU issue by priority, then age: the second-dispatched half of an 8-wave workgroup loses every
  // arbitration (MI355X_MICROARCH.md "two waves per SIMD", item 4).  One static s_setprio for that half, no per-phase flips.
  if (NW == 8 && p.prio && wave >= 4) __builtin_amdgcn_s_setprio(1);
  for (int t = 0; t < nt; ++t) {
    if (TPB == 2) {
      if ((t & 1) == 0) { issue_tile((cur + 2) & 3); issue_tile((cur + 3) & 3); }     // the next pair, into the stages of the pair before this one
    } else if (ABL != 2) issue_tile(nxt);
    const unsigned char* Ks = smem + (ABL == 6 ? 0 : cur) * STAGE;
    const unsigned char* Vs = Ks + KROWS * 128;

    // ---- S^T = K Q^T (two 32-key blocks) ----
    if (!rot) qk_tile(cur);
    const int kv0 = (t_first + t) * KVT;
    if (CAUSAL || kv0 + KVT > p.nkv) {
      asm volatile("; masked tile" ::: "memory");
      const int qlim = CAUSAL ? (q0 + l31) : 0x7fffffff;
#pragma unroll
      for (int kvb = 0; kvb < KVT / 32; ++kvb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + kvb * 32 + (r & 3) + 4 * ((r >> 2) & 1) + 8 * lg + 16 * (r >> 3);   // (permuted rows)
          if (kv >= p.nkv || kv > qlim) s[kvb][r] = -1e30f;
        }
    }
    float mx = -1e30f;
#pragma unroll
    for (int kvb = 0; kvb < KVT / 32; ++kvb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kvb][r]);
    mx = max_across_halves(mx);
    const float m_new = fmaxf(m_run, mx * sc);
    if (__any(m_new > m_run)) {
      asm volatile("; rescale" ::: "memory");
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      if (!ONES) l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    f32x2 psum2 = {0.f, 0.f};
    const f32x2 sc2 = {sc, sc}, nm2 = {-m_run, -m_run};
#pragma unroll
    for (int kvb = 0; kvb < KVT / 32; ++kvb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 sv = {s[kvb][r], s[kvb][r + 1]};
        const f32x2 e = __builtin_elementwise_fma(sv, sc2, nm2);
        const f32x2 pv = {ABL == 3 ? e[0] : __builtin_amdgcn_exp2f(e[0]), ABL == 3 ? e[1] : __builtin_amdgcn_exp2f(e[1])};
        s[kvb][r] = pv[0];
        s[kvb][r + 1] = pv[1];
        if (!ONES) psum2 += pv;
      }
    if (!ONES) l_run += psum2[0] + psum2[1];

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int kvb = 0; kvb < KVT / 32; ++kvb) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        f16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (f16)s[kvb][8 * s2 + e];
        const int ch = 4 * kvb + 2 * s2;        // + lg: the 16-byte chunk with this lane's eight keys (see the row permutation)
#pragma unroll
        for (int dt = 0; dt < DVT; ++dt) {
          const f16x8 a = *(const f16x8*)(Vs + (dt * 32 + l31) * 128 + (((ch + lg) ^ ksw) << 4));
          if (ABL == 4) { o[dt][kvb * 2 + s2] += (float)a[0] * (float)pf[0]; continue; }
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pf, o[dt], 0, 0, 0);
        }
      }
    }
    if (rot && t + 1 < nt) qk_tile((cur + 1 == NS) ? 0 : cur + 1);      // the rotated group: next tile's scores behind this tile's P V
    if (TPB == 2) {
      if ((t & 1) || t + 1 == nt) {                        // behind the second tile of a pair (or a lone last tile)
        wait_dma<0>();                                     // the next pair has landed
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    } else if (ABL != 1) {
      if (rot) wait_dma<PPW*(NS - 3 > 0 ? NS - 3 : 0)>();  // (rotated group: tile t + 2 has to be there for the next interval's Q K^T)
      else wait_dma<PPW*(NS - 2)>();                       // this wave's pieces of tile t + 1 have landed
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... everybody's; and tile t is fully read
    }
    cur = (cur + 1 == NS) ? 0 : cur + 1;
    nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
  }
  wait_dma<0>();

  if constexpr (KVS == 2) {
    // ---- merge the two key groups: group 1 parks {O^T, m, l} in LDS (lane-contiguous: no bank conflicts), group 0 folds them in ----
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // every DMA of both rings has landed, every fragment is read: the rings are free
    float* const xch = (float*)smem_all + (size_t)wave * ((DVT * 16 + 2) * 64) + lane;
    if (grp == 1) {
#pragma unroll
      for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(dt * 16 + r) * 64] = o[dt][r];
      xch[(DVT * 16) * 64] = m_run;
      xch[(DVT * 16 + 1) * 64] = l_run;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (grp == 1) return;
    const float m1 = xch[(DVT * 16) * 64], l1 = xch[(DVT * 16 + 1) * 64];
    const float m = fmaxf(m_run, m1);
    const float a0 = __builtin_amdgcn_exp2f(m_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = a0 * o[dt][r] + a1 * xch[(dt * 16 + r) * 64];
    l_run = a0 * l_run + a1 * l1;
  }

  float l_tot;
  if (ONES) {            // row D of O^T: tile D / 32, register and lane half of local row D % 32
    constexpr int rl = D % 32, r_l = (rl & 3) + 4 * (rl >> 3), lg_l = (rl >> 2) & 1;
    float a = o[D / 32][r_l], b = a;
    swap_halves(a, b);                                   // a = {a.lo, b.lo}, b = {a.hi, b.hi}
    l_tot = lg_l == 0 ? a : b;
  } else {
    l_tot = sum_across_halves(l_run);
  }
  const float inv = 1.0f / l_tot;
  const int q = q0 + l31;
  if (q < p.nq) {
    const int b = bh / p.heads, head = bh - b * p.heads;
    f16* orow = p.out + ((size_t)b * p.nq + q) * ((size_t)p.heads * D) + (size_t)head * D;
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int dd = dt * 32 + 8 * r4 + 4 * lg;
        if (dd < D) {
          f16x4 v = {(f16)(o[dt][r4 * 4 + 0] * inv), (f16)(o[dt][r4 * 4 + 1] * inv), (f16)(o[dt][r4 * 4 + 2] * inv),
                     (f16)(o[dt][r4 * 4 + 3] * inv)};
          SDMI_ST(f16x4, orow + dd, v);
        }
      }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

#ifdef SDMI_EXPERIMENTS      // (bit-identical, measured 12 % slower at d = 40 in round 4: profiles/experiments_r04.txt)
// ---- ping-pong variant of attn_dma_kernel for 8-wave workgroups (two waves per SIMD) ------------------------------------------
// Same arithmetic, operand layouts and LDS-DMA ring as attn_dma_kernel -- per wave the very same instruction sequence on the same
// values, so the results are bit-identical -- but the two waves of a SIMD no longer run in lock-step.  attn_dma_kernel has one
// barrier per key tile: all eight waves do QK^T (matrix pipe), then all do the softmax (VALU: 32 v_exp_f32 + 16 cvt + 16 max3 + 20
// fma per lane and tile), then all do PV (matrix pipe) -- the ablations of round 2 (profiles/attn_variants_r02.txt) show every
// ingredient costing exactly its own pipe time: zero overlap between the two waves that share a SIMD's matrix pipe and VALU issue.
// Here a key tile is TWO steps with a barrier each, and the halves of the workgroup (waves 0-3 / 4-7: wave i and i + 4 share a
// SIMD) run them in opposite order:
//     step 2t     : half A  [PV(t-1), QK^T(t)]  (14 MFMAs)     half B  softmax(t-1)          (VALU)
//     step 2t + 1 : half A  softmax(t)          (VALU)          half B  [PV(t-1), QK^T(t)]    (14 MFMAs)
// so each SIMD always has one wave in its matrix block and one in its VALU block (MI355X_MICROARCH.md "two waves per SIMD": the
// matrix pipe and the VALU of a SIMD run concurrently for two different waves; a rendezvous pays when the paired intervals are
// complementary).  P(t) stays in registers from softmax(t) to PV(t) one step later; tile t's ring slot is read until half B's
// PV(t) at step 2t + 3, so the ring runs NS - 2 tiles ahead instead of NS - 1.
template <int D, int NS>
__global__ void __launch_bounds__(512) attn_pp_kernel(const AttnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = 8;
  constexpr int DKS = (D + 15) / 16;
  constexpr int DVT = (D + 31) / 32;
  constexpr int NCH = (D + 63) / 64;
  constexpr int KROWS = NCH * 64, VROWS = DVT * 32;
  constexpr int STAGE = (KROWS + VROWS) * 128;
  constexpr bool ONES = VROWS > D;
  constexpr int PK = NCH * 8, PV = ONES ? D / 8 : DVT * 4, PT = PK + PV;
  static_assert(D % 8 == 0, "a DMA piece is 8 rows");
  constexpr int PPW = (PT + NW - 1) / NW;
  static_assert(NS >= 4 && NS * STAGE <= 160 * 1024, "LDS budget (the ring runs NS - 2 tiles ahead)");

  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                         // half of the workgroup: 0 = matrix block on even steps, 1 = on odd steps
  const int l31 = lane & 31, lg = lane >> 5;
  int bh, qt;
  head_of_block(bh, qt);
  const int q0 = qt * (32 * NW) + wave * 32;
  const f16* Qg = p.q + (size_t)bh * p.nq * D;
  constexpr int OOB = (int)0x80000000;
  const __amdgpu_buffer_rsrc_t rsrc_k =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.k + (size_t)bh * p.nkv * D), 0, p.nkv * D * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_v =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.vt + (size_t)bh * D * p.nkv_pad), 0, D * p.nkv_pad * 2, 0x00020000);

  f16x8 qf[DKS];
#pragma unroll
  for (int ks = 0; ks < DKS; ++ks) {
    const int dcol = ks * 16 + lg * 8;
    f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (q0 + l31 < p.nq && dcol < D) v = *(const f16x8*)(Qg + (size_t)(q0 + l31) * D + dcol);
    qf[ks] = v;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the Q fragments are register loads: drained before any LDS-DMA is in flight)

  int pv_off[PPW], pv_step[PPW], pv_row[PPW];
  bool pv_isk[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int q = min(wave + j * NW, PT - 1);
    const int r8 = lane >> 3, cp = lane & 7;
    if (q < PK) {
      const int c = q >> 3, row = (q & 7) * 8 + r8;
      const int key = (row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1);
      const int gch = cp ^ ((row >> 1) & 7);
      pv_off[j] = key * (D * 2) + c * 128 + gch * 16;
      pv_step[j] = KVT * D * 2;
      pv_row[j] = c * 64 + (q & 7) * 8;
      pv_isk[j] = true;
    } else {
      const int qv = q - PK, dt = qv >> 2, row = (qv & 3) * 8 + r8;
      const int d = dt * 32 + row;
      const int gch = cp ^ ((row >> 1) & 7);
      pv_off[j] = d < D ? d * (p.nkv_pad * 2) + gch * 16 : OOB;
      pv_step[j] = d < D ? KVT * 2 : 0;
      pv_row[j] = KROWS + dt * 32 + (qv & 3) * 8;
      pv_isk[j] = false;
    }
  }
  auto issue_tile = [&](int stage) {
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      auto dst = (__attribute__((address_space(3))) void*)(smem + stage * STAGE + pv_row[j] * 128);
      if (pv_isk[j]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_k, dst, 16, pv_off[j], 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_v, dst, 16, pv_off[j], 0, 0, 0);
      pv_off[j] += pv_step[j];
    }
  };

  f32x16 o[DVT];
#pragma unroll
  for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  if (ONES) {
    for (int e = tid; e < NS * (VROWS - D) * 8; e += NW * 64) {
      const int st = e / ((VROWS - D) * 8), rc = e - st * ((VROWS - D) * 8), row = D + (rc >> 3);
      const unsigned one2 = row == D ? 0x3C003C00u : 0u;
      *(u32x4*)(smem + st * STAGE + (KROWS + row) * 128 + (rc & 7) * 16) = u32x4{one2, one2, one2, one2};
    }
  }
  const float sc = p.scale * 1.4426950408889634f;
  const int nt = (p.nkv + KVT - 1) / KVT;
  const int ksw = (l31 >> 1) & 7;

  // the three blocks of a tile (the same code as attn_dma_kernel's loop body, cut at the two hand-overs)
  f32x16 s[KVT / 32];                                // scores, then fp32 probabilities of the tile in flight
  f16x8 pf[KVT / 32][2];                             // ... as fp16 B fragments, from softmax(t) to PV(t)
  auto qk = [&](int t) __attribute__((always_inline)) {
    const unsigned char* Ks = smem + (t % NS) * STAGE;
#pragma unroll
    for (int kvb = 0; kvb < KVT / 32; ++kvb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kvb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < DKS; ++ks) {
        const unsigned char* kp = Ks + ((ks >> 2) * 64 + kvb * 32 + l31) * 128 + ((((ks & 3) * 2 + lg) ^ ksw) << 4);
        const f16x8 a = *(const f16x8*)kp;
        s[kvb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[ks], s[kvb], 0, 0, 0);
      }
    }
  };
  auto softmax = [&](int t) __attribute__((always_inline)) {
    const int kv0 = t * KVT;
    if (kv0 + KVT > p.nkv) {
      asm volatile("; masked tile" ::: "memory");
#pragma unroll
      for (int kvb = 0; kvb < KVT / 32; ++kvb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + kvb * 32 + (r & 3) + 4 * ((r >> 2) & 1) + 8 * lg + 16 * (r >> 3);
          if (kv >= p.nkv) s[kvb][r] = -1e30f;
        }
    }
    float mx = -1e30f;
#pragma unroll
    for (int kvb = 0; kvb < KVT / 32; ++kvb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kvb][r]);
    mx = max_across_halves(mx);
    const float m_new = fmaxf(m_run, mx * sc);
    if (__any(m_new > m_run)) {
      asm volatile("; rescale" ::: "memory");
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      if (!ONES) l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    f32x2 psum2 = {0.f, 0.f};
    const f32x2 sc2 = {sc, sc}, nm2 = {-m_run, -m_run};
#pragma unroll
    for (int kvb = 0; kvb < KVT / 32; ++kvb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 sv = {s[kvb][r], s[kvb][r + 1]};
        const f32x2 e = __builtin_elementwise_fma(sv, sc2, nm2);
        const f32x2 pv = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
        s[kvb][r] = pv[0];
        s[kvb][r + 1] = pv[1];
        if (!ONES) psum2 += pv;
      }
    if (!ONES) l_run += psum2[0] + psum2[1];
#pragma unroll
    for (int kvb = 0; kvb < KVT / 32; ++kvb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[kvb][s2][e] = (f16)s[kvb][8 * s2 + e];
  };
  auto pvmul = [&](int t) __attribute__((always_inline)) {
    const unsigned char* Vs = smem + (t % NS) * STAGE + KROWS * 128;
#pragma unroll
    for (int kvb = 0; kvb < KVT / 32; ++kvb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int ch = 4 * kvb + 2 * s2;
#pragma unroll
        for (int dt = 0; dt < DVT; ++dt) {
          const f16x8 a = *(const f16x8*)(Vs + (dt * 32 + l31) * 128 + (((ch + lg) ^ ksw) << 4));
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pf[kvb][s2], o[dt], 0, 0, 0);
        }
      }
  };

  // ring: tiles 0 .. NS - 3 up front; tile t + NS - 2 is requested at step 2 t (its slot held tile t - 2, last read by half B's
  // PV(t - 2) at step 2 t - 1); tile t must have landed for everybody when step 2 t begins
#pragma unroll
  for (int s2 = 0; s2 < NS - 2; ++s2) issue_tile(s2);
  wait_dma<PPW*(NS - 3)>();                          // tile 0
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // One loop per half, chosen once: inside it the blocks follow each other in a fixed order and update s / pf / o in place (a
  // single loop that picks the block by the parity of a step counter made the compiler copy the 80 accumulator registers around
  // at every join: 184 v_mov per step, 145 us instead of 81).  Both halves execute the same barriers: two per key tile.
  auto run_half = [&](auto grp_c) __attribute__((always_inline)) {
    constexpr int G = decltype(grp_c)::value;
    for (int t = 0; t <= nt; ++t) {
      issue_tile((t + NS - 2) % NS);                 // (tiles past nt read out of range: zeros, never consumed)
      if constexpr (G == 0) {                        // step 2 t
        if (t >= 1) pvmul(t - 1);
        if (t < nt) qk(t);
      } else {
        if (t >= 1) softmax(t - 1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if constexpr (G == 0) {                        // step 2 t + 1
        if (t < nt) softmax(t);
      } else {
        if (t >= 1) pvmul(t - 1);
        if (t < nt) qk(t);
      }
      wait_dma<PPW*(NS - 3)>();                      // tile t + 1, needed from the next step on, has landed for this wave
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  };
  if (grp == 0) run_half(std::integral_constant<int, 0>{});
  else run_half(std::integral_constant<int, 1>{});
  wait_dma<0>();

  float l_tot;
  if (ONES) {
    constexpr int rl = D % 32, r_l = (rl & 3) + 4 * (rl >> 3), lg_l = (rl >> 2) & 1;
    float a = o[D / 32][r_l], b = a;
    swap_halves(a, b);
    l_tot = lg_l == 0 ? a : b;
  } else {
    l_tot = sum_across_halves(l_run);
  }
  const float inv = 1.0f / l_tot;
  const int q = q0 + l31;
  if (q < p.nq) {
    const int b = bh / p.heads, head = bh - b * p.heads;
    f16* orow = p.out + ((size_t)b * p.nq + q) * ((size_t)p.heads * D) + (size_t)head * D;
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int dd = dt * 32 + 8 * r4 + 4 * lg;
        if (dd < D) {
          f16x4 v = {(f16)(o[dt][r4 * 4 + 0] * inv), (f16)(o[dt][r4 * 4 + 1] * inv), (f16)(o[dt][r4 * 4 + 2] * inv),
                     (f16)(o[dt][r4 * 4 + 3] * inv)};
          SDMI_ST(f16x4, orow + dd, v);
        }
      }
  }
#endif  // __HIP_DEVICE_COMPILE__
}
#endif  // SDMI_EXPERIMENTS

// A/B knobs of the experiments build (SDMI_CXXFLAGS=-DSDMI_EXPERIMENTS): the product library reads none of them
#ifdef SDMI_EXPERIMENTS
#define SDMI_EXP_ENV(name, def) (getenv(name) ? atoi(getenv(name)) : (def))
#else
#define SDMI_EXP_ENV(name, def) (def)
#endif

#ifndef SDMI_ATTN_TPB_DEFAULT
#define SDMI_ATTN_TPB_DEFAULT 1
#endif
#ifndef SDMI_ATTN_ROT_DEFAULT
#define SDMI_ATTN_ROT_DEFAULT 0
#endif
#ifndef SDMI_ATTN_KVS_DEFAULT
#define SDMI_ATTN_KVS_DEFAULT 1
#endif
template <int D>
int launch_d(const AttnParams& p, hipStream_t stream) {
  // Long sequences (> 1024 queries): 8 waves share each K/V tile.  Short ones (the 32x32 / 16x16 / 8x8 levels) are
  // latency bound with few workgroups: at most 4 waves per workgroup so there are more of them.
  int nw = p.nw;
  if (nw <= 0) {
    const int slices = cdiv(p.nq, 32);
    // (same-box A/B, profiles/ab_*_r01.txt: 4 waves beat 8 up to 1024 queries -- more, smaller workgroups)
    nw = (slices >= 8 && p.nq > 1024) ? 8 : (slices >= 4 ? 4 : 2);
    static const int env_small = SDMI_EXP_ENV("SDMI_ATTN_NW_LE1K", 0);   // A/B knobs
    static const int env_big = SDMI_EXP_ENV("SDMI_ATTN_NW_GT1K", 0);
    if (env_small > 0 && p.nq <= 1024) nw = env_small;
    if (env_big > 0 && p.nq > 1024) nw = env_big;
  }
  dim3 grid(cdiv(p.nq, 32 * nw), p.BH);
  static const std::string pname_long = std::string("attn_d") + std::to_string(D) + "_self";
  static const std::string pname_short = std::string("attn_d") + std::to_string(D) + "_ctx";
  ProfScope ps((p.nkv >= 256 ? pname_long : pname_short).c_str(), 4.0 * p.BH * (double)p.nq * p.nkv * D, 2.0 * p.BH * D * (2.0 * p.nq + 2.0 * p.nkv), stream);
  static const int use_v1 = SDMI_EXP_ENV("SDMI_ATTN_V1", 0);     // A/B: the register-staged kernel (the product build keeps it for unaligned K / V only)
  constexpr int DNS = (D > 128) ? 3 : 4;                        // LDS-DMA ring depth (D = 160: 3 x 44 KB)
  // round 6: the key range split over two 8-wave groups of ONE 16-wave workgroup (attn_dma_kernel KVS = 2): d = 40 self-attention of the
  // 64 x 64 level (an even number of 64-key tiles, 16 .. 32 of them per group: 2048 .. 4096 keys).  Same-box A/B (profiles/attn_kvsplit_r06.txt): 78.0 -> 73.9 us
  // per launch at 4096 keys, UNet call -0.019 ms at 64 x 64 but +0.06 ms at 96 x 96 (9216 keys: the 8-wave kernel stays there).
  // SDMI_ATTN_KVS=0 restores the 8-wave kernel everywhere, =2 forces the split for every even tile count >= 32 (A/B; read per launch)
  const int kvs_env = getenv("SDMI_ATTN_KVS") ? atoi(getenv("SDMI_ATTN_KVS")) : SDMI_ATTN_KVS_DEFAULT;
  // (experiments build, read per launch) one barrier per two key tiles / the second key group's tile blocks rotated
  const bool tpb2 = SDMI_EXP_ENV("SDMI_ATTN_TPB", SDMI_ATTN_TPB_DEFAULT) == 2;
  const bool kvs_rot = SDMI_EXP_ENV("SDMI_ATTN_ROT", SDMI_ATTN_ROT_DEFAULT) != 0;
  (void)tpb2; (void)kvs_rot;
  const bool kvs2 = D == 40 && nw == 8 && !p.causal && !p.pingpong && p.nkv % (2 * KVT) == 0 && p.nkv >= 32 * KVT && kvs_env != 0 && (p.nkv <= 64 * KVT || kvs_env == 2);
  if (!use_v1 && !p.causal && (p.nkv * D) % 8 == 0) {
    // timing-only ablations (WRONG results: each removes one ingredient of the loop, tools/attn_ablate.py) exist only in a build with
    // -DSDMI_ABLATE (SDMI_CXXFLAGS=-DSDMI_ABLATE SDMI_LIB_OUT=libsdmi_ablate.so python stable-diffusion_amd/build.py): no environment
    // variable can make the product library compute something else
#ifdef SDMI_ABLATE
    static const int abl = getenv("SDMI_ATTN_ABL") ? atoi(getenv("SDMI_ATTN_ABL")) : 0;
#else
    constexpr int abl = 0;
#endif
    if (D == 40 && nw == 8 && abl) {
#ifdef SDMI_ABLATE
      if constexpr (D == 40) {
        switch (abl) {
          case 1: SDMI_LAUNCH((attn_dma_kernel<D, 8, DNS, false, 1>), grid, dim3(512), 0, stream, p); break;
          case 2: SDMI_LAUNCH((attn_dma_kernel<D, 8, DNS, false, 2>), grid, dim3(512), 0, stream, p); break;
          case 3: SDMI_LAUNCH((attn_dma_kernel<D, 8, DNS, false, 3>), grid, dim3(512), 0, stream, p); break;
          case 4: SDMI_LAUNCH((attn_dma_kernel<D, 8, DNS, false, 4>), grid, dim3(512), 0, stream, p); break;
          case 5: SDMI_LAUNCH((attn_dma_kernel<D, 8, DNS, false, 5>), grid, dim3(512), 0, stream, p); break;
          default: SDMI_LAUNCH((attn_dma_kernel<D, 8, DNS, false, 6>), grid, dim3(512), 0, stream, p); break;
        }
      }
#endif
#ifdef SDMI_EXPERIMENTS
    } else if (nw == 8 && p.pingpong && DNS >= 4) {
      if constexpr (DNS >= 4) SDMI_LAUNCH((attn_pp_kernel<D, DNS>), grid, dim3(512), 0, stream, p);
#endif
    } else if (nw == 8 && kvs2) {
      if constexpr (D == 40) {
#ifdef SDMI_EXPERIMENTS      // (both bit-identical to the plain key split and measured slower: profiles/attn_kvsplit_r06.txt)
        if (kvs_rot) SDMI_LAUNCH((attn_dma_kernel<D, 16, DNS, false, 0, 2, true>), grid, dim3(1024), 0, stream, p);
        else if (tpb2) SDMI_LAUNCH((attn_dma_kernel<D, 16, DNS, false, 0, 2, false, 2>), grid, dim3(1024), 0, stream, p);
        else
#endif
        SDMI_LAUNCH((attn_dma_kernel<D, 16, DNS, false, 0, 2>), grid, dim3(1024), 0, stream, p);
      }
#ifdef SDMI_EXPERIMENTS
    } else if (nw == 8 && tpb2 && D == 40) {
      if constexpr (D == 40) SDMI_LAUNCH((attn_dma_kernel<D, 8, DNS, false, 0, 1, false, 2>), grid, dim3(512), 0, stream, p);
#endif
    } else if (nw == 8) SDMI_LAUNCH((attn_dma_kernel<D, 8, DNS, false>), grid, dim3(512), 0, stream, p);
    else if (nw == 4) SDMI_LAUNCH((attn_dma_kernel<D, 4, DNS, false>), grid, dim3(256), 0, stream, p);
    else SDMI_LAUNCH((attn_dma_kernel<D, 2, DNS, false>), grid, dim3(128), 0, stream, p);
  } else if (p.causal) {          // the text encoder's 77-token self-attention: one configuration is enough
    if constexpr (D == 32 || D == 64 || D == 128) SDMI_LAUNCH((attn_kernel<D, 2, true>), dim3(cdiv(p.nq, 64), p.BH), dim3(128), 0, stream, p);
    else return fail("causal attention is instantiated for head dims 32 / 64 / 128");
  } else if (nw == 8) SDMI_LAUNCH((attn_kernel<D, 8, false>), grid, dim3(512), 0, stream, p);
  else if (nw == 4) SDMI_LAUNCH((attn_kernel<D, 4, false>), grid, dim3(256), 0, stream, p);
  else SDMI_LAUNCH((attn_kernel<D, 2, false>), grid, dim3(128), 0, stream, p);
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}

}  // namespace

static int launch_attention_impl(const AttnParams& p, hipStream_t stream);
int launch_attention(const AttnParams& p, hipStream_t stream) {
  const int rc = launch_attention_impl(p, stream);
  if (rc == 0 && range_check_enabled()) return range_scan("attention output", p.out, (int64_t)p.BH * p.nq * p.d, stream);
  return rc;
}
static int launch_attention_impl(const AttnParams& p_in, hipStream_t stream) {
  AttnParams p = p_in;
  static const int env_prio = SDMI_EXP_ENV("SDMI_ATTN_PRIO", 0);      // A/B knob (bit-identical)
  p.prio = env_prio;
  p.pingpong = SDMI_EXP_ENV("SDMI_ATTN_PP", 0);       // (read per launch: the tests flip it) 8-wave launches on attn_pp_kernel
  SDMI_CHECK(p.BH > 0 && p.nq > 0 && p.nkv > 0 && p.heads > 0 && p.BH % p.heads == 0, "bad attention shape");
  SDMI_CHECK(p.nkv_pad % 8 == 0 && p.nkv_pad >= p.nkv, "nkv_pad must be a multiple of 8 and >= nkv");
  SDMI_CHECK(!p.causal || p.nq == p.nkv, "causal attention needs nq == nkv");
  switch (p.d) {
    case 32: return launch_d<32>(p, stream);
    case 40: return launch_d<40>(p, stream);
    case 64: return launch_d<64>(p, stream);
    case 80: return launch_d<80>(p, stream);
    case 128: return launch_d<128>(p, stream);
    case 160: return launch_d<160>(p, stream);
    default: return fail("attention head dim " + std::to_string(p.d) + " not instantiated (32/40/64/80/128/160)");
  }
}

}  // namespace sdmi
