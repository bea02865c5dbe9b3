// ResBlock in_layers / out_layers as ONE launch for gfx950 (MI355X):   conv3x3( SiLU( GroupNorm32(x) ) ) + bias (+ emb row, + residual)
//     ldm/modules/diffusionmodules/openaimodel.py:201-204, 225-231 (in_layers / out_layers), :246-261 (_forward); util.py:199-216 (GroupNorm32)
//
// The north-star fusion with a tile that owns ALL output columns.  The earlier GroupNorm-folding convolution (conv3halo_gn_kernel) lost
// because every one of its N / 64 column tiles normalised its own copy of the input halo (5x at N = 320) through registers.  Here a
// workgroup OWNS 32 output pixels (half an image row at W = 64) for all N = 320 output channels -- M / 32 = 256 workgroups = one per CU
// at the 64 x 64 level -- so the halo (3 rows x 34 pixels) is read as fp32 ONCE per workgroup, normalised + SiLU'd once into LDS
// (gn_apply_elem: the GroupNorm-apply kernel's expression, the same fp16 operand bits), and every column reuses it.  What streams is the
// WEIGHTS (9 taps x Cin x 320 fp16 = 1.8 MB at Cin = 320), through the row-strip chain kernels' LDS-DMA ring (rowchain.hip): units of
// 160 weight rows x 64 k (20 KB), loader waves that issue nothing else, five compute waves x 64 output columns (two 32 x 32 MFMA tiles),
// one raw s_barrier per unit.
//
// k order = the packed conv weights' (pack_conv_kernel: 64-channel chunk major, the nine taps of a chunk adjacent): per chunk c the taps
// 0 .. 8, per tap the four k-steps -- conv3halo_kernel's order, so the accumulators (and the outputs) are the bits of GroupNorm-apply +
// conv3halo / igemm launches with splitk = 1.  Only ONE chunk of the halo is resident (104 pixel rows x 128 B, double buffered): chunk
// c + 1 is loaded (fp32, registers) at tap 0 of chunk c and normalised into the other buffer at taps 3 / 5 / 7, under the weight stream
// (the compute waves' MFMAs take ~130 of a unit's ~600 cycles); the prologue pays for chunk 0 only.
//
// The epilogue is the shared GEMM epilogue (igemm_dev.h) on a 32 x 320 tile: bias, the time-embedding row vector, the residual, fp32 out,
// optional fp16 copy, the GroupNorm statistics of the consumers.
#include <utility>

#include "igemm_dev.h"

namespace sdmi {
namespace {

constexpr int GC_ROWS = 32;                    // output pixels per workgroup
constexpr int GC_N = 320;                      // output channels (all of them)
constexpr int GC_NWC = 5;                      // compute waves
constexpr int GC_NTC = GC_NWC * 64;
constexpr int GC_UNIT = 160 * 128;             // 20 KB: 160 weight rows x 64 k fp16
constexpr int GC_UPIECES = GC_UNIT / 1024;     // LDS-DMA instructions per unit
constexpr int GC_NS = 4;                       // ring depth
constexpr int GC_HW2 = GC_ROWS + 2;            // halo pixels per image row (34)
constexpr int GC_HP = 3 * GC_HW2;              // halo pixels (102)
constexpr int GC_HBYTES = 104 * 128;           // one halo chunk buffer: pixel rows of 64 channels
constexpr int GC_ITEMS = 3;                    // (pixel, channel octet) items per compute thread and chunk: 102 x 8 = 816 <= 3 x 320
constexpr int GC_MAXC = 960;                   // input channels (gamma / beta tables in LDS)

#if defined(__HIP_DEVICE_COMPILE__)
template <int... I, class F>
__device__ __forceinline__ void gc_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void gc_static_for(F&& f) { gc_static_for_impl(std::make_integer_sequence<int, N>{}, f); }
#endif

// PF: one more wave that only PREFETCHES.  Inside a UNet call the weights are HBM-cold; the workgroups of an XCD stream the same unit at the
// same time, a CU keeps only ~256 lines in flight, and a first touch of the L2 takes ~1300 cycles where a hit takes ~700: 800 cycles per
// unit against 430 with hot weights (thread-0 stamps, profiles/gn_conv3_r05.txt).  The workgroups of an XCD (slot = blockIdx / 8 of
// gridDim / 8) share the job: behind the barrier of unit g each touches its 1 / nslots of the 160 lines of unit g + GC_PFD -- one dword per
// 32 bytes, into a dead corner of LDS, never waited for -- so that together they have pulled the whole unit into their L2 ~GC_PFD units
// before anybody streams it, at 5 lines per workgroup and unit.
constexpr int GC_PFD = 24;
template <int NLD, int NS = GC_NS, bool PF = false>
__global__ void __launch_bounds__(GC_NTC + 64 * NLD + (PF ? 64 : 0)) gn_conv3_kernel(const GnConvParams rp) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(NLD == 1 || NLD == 2, "loader waves");
  constexpr int RING = NS * GC_UNIT;
  constexpr int OFF_H = RING;                         // two halo chunk buffers
  constexpr int OFF_GTAB = OFF_H + 2 * GC_HBYTES;     // {mean, rstd} of the sample's 32 groups
  constexpr int OFF_GB = OFF_GTAB + 32 * 8;           // gamma [Cin] | beta [Cin] fp32
  constexpr int OFF_PF = OFF_GB + 2 * GC_MAXC * 4;    // 256 B nobody reads (the prefetch wave's destination)
  constexpr int LDS_TOTAL = OFF_PF + 256;
  static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
  static_assert(RING >= GC_NWC * 32 * 68 * 4, "the epilogue turns its slabs through the ring");
  constexpr int LPIECES = GC_UPIECES / NLD, LD_WAIT = LPIECES * (NS - 2);
  static_assert(NS >= 3 && LD_WAIT + LPIECES <= 63 && LD_WAIT <= 48, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_TOTAL];
#ifdef SDMI_RC_TIMING
  long long stamp[16];                                // timing build: cycle stamps of thread 0 (entry, tables, chunk 0 ready, every chunk, ring drained, end)
  int n_stamp = 0;
#define GC_STAMP() do { if (n_stamp < 16) stamp[n_stamp] = (long long)__builtin_readcyclecounter(); ++n_stamp; } while (0)
#else
#define GC_STAMP() do { } while (0)
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const IGemmParams& ep = rp.epi;
  const int Cin = rp.c0 + rp.c1;
  const int nch = Cin >> 6;                           // 64-channel chunks
  const int NU = nch * 18;                            // weight units: chunk x tap x column half
  constexpr int OOB = (int)0x80000000;
  // tile numbering: an XCD (workgroups are dealt round-robin by linear id) owns a contiguous range of tiles = image rows, so a halo row
  // is fetched from the fabric by (mostly) one L2 instead of three
  const int ntiles = gridDim.x, bid = blockIdx.x;
  const int tile = (ntiles & 7) == 0 ? (bid & 7) * (ntiles >> 3) + (bid >> 3) : bid;
  const int m0 = tile * GC_ROWS;

  if (PF && wave_u == GC_NWC + NLD) {
    // =============================== the prefetch wave =============================================================================
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)rp.w, 0, OOB, 0x00020000);
    const int ldw2 = ep.K * 2;
    // the 160 lines of a unit: line idx -> weight row 64 (idx / 32) + idx % 32 (+ 32 half), 128 bytes at k offset (g / 2) * 128;
    // this workgroup's lines: idx = slot, slot + nslots, ... (up to 16 of them), lane -> (line lane / 4, 32-byte sector lane % 4)
    const int nslots = (ntiles & 7) == 0 ? max(ntiles >> 3, 10) : 160, slot = (ntiles & 7) == 0 ? (bid >> 3) : 160;
    const int idx = slot + (lane >> 2) * nslots;
    const int voff = (slot < nslots && idx < 160) ? (64 * (idx >> 5) + (idx & 31)) * ldw2 + (lane & 3) * 32 : OOB;
    auto touch = [&](int g) {
      if (g >= NU) return;
      const int soff = __builtin_amdgcn_readfirstlane((g & 1) * 32 * ldw2 + (g >> 1) * 128);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(smem + OFF_PF), 4, voff, soff, 0, 0);
    };
    // ... and, 16 units before the end, one dword of each 128-byte line of the residual rows the epilogue adds (written several launches
    // ago): the epilogue's loads then find them in the L2 (rowchain.hip: ff_tail_kernel's prefetch_x)
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)(ep.residual ? ep.residual : ep.out_f32), 0, OOB, 0x00020000);
    auto touch_residual = [&]() {
      if (!ep.residual) return;
#pragma unroll
      for (int p = 0; p < 5; ++p) {
        const int t = p * 64 + lane;                    // 320 lines: row t / 10, line t % 10
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_r, (__attribute__((address_space(3))) void*)(smem + OFF_PF), 4,
                                                 ((m0 + t / 10) * ep.ldr + (t % 10) * 32) * 4, 0, 0, 0);
      }
    };
    for (int g = NS - 1; g < GC_PFD; ++g) touch(g);
    asm volatile("s_barrier" ::: "memory");             // X0
    for (int g = 0; g < NU; ++g) {
      asm volatile("s_barrier" ::: "memory");
      touch(g + GC_PFD);
      if (g == NU - 16) touch_residual();
    }
    wait_vmcnt<0>();
    asm volatile("s_barrier" ::: "memory");
    return;
  }
  if (wave_u >= GC_NWC) {
    // =============================== the loader waves (rowchain.hip: ff_tail_kernel) ================================================
    const int lw = wave_u - GC_NWC;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)rp.w, 0, OOB, 0x00020000);
    const int l8 = lane >> 3, cpos = lane & 7;
    const int g16_0 = (cpos ^ ((l8 >> 1) & 7)) << 4, g16_1 = (cpos ^ ((4 + (l8 >> 1)) & 7)) << 4;
    const int ldw2 = ep.K * 2;                          // weight row pitch in bytes (K = 9 Cin)
    const int v0 = l8 * ldw2 + g16_0, v1 = l8 * ldw2 + g16_1;
    const int vv = (lw & 1) ? v1 : v0;
    // unit g = (chunk * 9 + tap) * 2 + half: weight rows 64 w + 32 half + [0, 32) of compute wave w, k = (chunk * 9 + tap) * 64 + [0, 64)
    auto issue_unit = [&](int g, int stage) {
      g = min(g, NU - 1);                               // (past the end: the last unit again -- in bounds, never consumed)
      const int soff = __builtin_amdgcn_readfirstlane((g & 1) * 32 * ldw2 + (g >> 1) * 128);
#pragma unroll
      for (int pp = 0; pp < LPIECES; ++pp) {
        const int p = pp * NLD + lw;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(smem + stage * GC_UNIT + p * 1024), 16,
                                                 NLD == 1 ? ((p & 1) ? v1 : v0) : vv, soff + (64 * (p >> 2) + 8 * (p & 3)) * ldw2, 0, SDMI_W_AUX);
      }
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue_unit(s, s);
    asm volatile("s_barrier" ::: "memory");             // X0: the compute waves' GroupNorm / gamma / beta tables
    int nxt = NS - 1;
    for (int g = 0; g < NU; ++g) {
      wait_vmcnt<LD_WAIT>();                            // unit g has landed
      asm volatile("s_barrier" ::: "memory");           // ... and the compute waves are done with unit g - 1
      issue_unit(g + NS - 1, nxt);
      nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
    }
    wait_vmcnt<0>();
    asm volatile("s_barrier" ::: "memory");
    return;
  }

  // ================================= the five compute waves ==========================================================================
  GC_STAMP();
  const int l31 = lane & 31, lg = lane >> 5;
  const int W = ep.Wout, H = ep.Hout, HW = H * W;
  const int bsample = m0 / HW;
  const int rem = m0 - bsample * HW;
  const int y0 = rem / W, x0 = rem - y0 * W;          // the tile: pixels (y0, x0 .. x0 + 31) of sample bsample
  const int cpg = Cin >> 5;
  float2* const gtab = (float2*)(smem + OFF_GTAB);
  // ---- prologue: gamma / beta -> LDS, the sample's GroupNorm table (norm.hip gn_fold: 8 consecutive lanes fold the 8 slots of a group) ----
  for (int i = tid; i < (Cin >> 2); i += GC_NTC) {
    *(f32x4*)(smem + OFF_GB + i * 16) = *(const f32x4*)(rp.gn_gamma + i * 4);
    *(f32x4*)(smem + OFF_GB + Cin * 4 + i * 16) = *(const f32x4*)(rp.gn_beta + i * 4);
  }
  if (tid < 256) {
    const int g = tid >> 3, sub = tid & 7;
    const long long* src = rp.gn_acc + ((size_t)(bsample * 32 + g) * GN_SLOTS + sub) * GN_STRIDE;
    long long a = src[0], al = src[1], q = src[2], ql = src[3];
#pragma unroll
    for (int o = GN_SLOTS / 2; o >= 1; o >>= 1) {
      a += __shfl_xor(a, o); al += __shfl_xor(al, o); q += __shfl_xor(q, o); ql += __shfl_xor(ql, o);
    }
    if (sub == 0) {
      float m, r;
      gn_mean_rstd(a, al, q, ql, (double)cpg * (double)HW, rp.gn_eps, &m, &r);
      gtab[g] = float2{m, r};
    }
  }
  // this thread's items of a chunk: item it = tid + 320 i -> halo pixel it / 8 (row hy = pixel / 34, column hx = pixel % 34), channel
  // octet it % 8 = tid % 8 (320 % 8 == 0: the same octet for all three); out-of-image pixels are the convolution's zero padding
  const int oct = tid & 7;
  int poff[GC_ITEMS];                                   // pixel index inside the sample's [HW] rows, or -1
  int hrow[GC_ITEMS];                                   // LDS byte offset inside a halo buffer, or -1 (no such item)
#pragma unroll
  for (int i = 0; i < GC_ITEMS; ++i) {
    const int it = tid + GC_NTC * i;
    const int px = it >> 3;
    const int hy = px / GC_HW2, hx = px - hy * GC_HW2;
    const int y = y0 + hy - 1, x = x0 + hx - 1;
    const bool item = px < GC_HP;
    poff[i] = (item && y >= 0 && y < H && x >= 0 && x < W) ? (bsample * HW + y * W + x) : -1;
    hrow[i] = item ? px * 128 + ((oct ^ ((px >> 1) & 7)) << 4) : -1;
  }
  // Two register sets: chunk c + 2 is requested at tap 0 of chunk c and chunk c + 1 (requested a whole chunk = 18 units earlier) is
  // normalised at taps 3 / 5 / 7 -- inside a UNet call x comes from the Infinity Cache / HBM, and with one chunk of look-ahead (three
  // taps between request and use) every workgroup sat in front of its loads: 51 us per launch against 36 us with hot operands
  f32x4 xa[GC_ITEMS][2], xb[GC_ITEMS][2];
  auto load_chunk = [&](int c, f32x4 (&xv)[GC_ITEMS][2]) {      // the fp32 halo of chunk c (this thread's items) into registers
    const int ch0 = c * 64 + oct * 8;
    const float* src; int ld, co;
    if (ch0 < rp.c0) { src = rp.x0; ld = rp.c0; co = ch0; } else { src = rp.x1; ld = rp.c1; co = ch0 - rp.c0; }
#pragma unroll
    for (int i = 0; i < GC_ITEMS; ++i) {
      if (poff[i] >= 0) {
        const float* s = src + (size_t)poff[i] * ld + co;
        xv[i][0] = *(const f32x4*)s; xv[i][1] = *(const f32x4*)(s + 4);
      } else {
        xv[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; xv[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto convert_item = [&](int c, int i, int buf, const f32x4 (&xv)[GC_ITEMS][2]) {      // normalise + SiLU (gn_apply_kernel's expression) -> fp16 -> the halo buffer
    if (hrow[i] < 0) return;
    f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
    if (poff[i] >= 0) {
      const int ch0 = c * 64 + oct * 8;
      const int g0 = ch0 / cpg;
      const int nfirst = (g0 + 1) * cpg - ch0;          // channels of the octet in group g0 (cpg >= 8: at most two groups)
      const float2 ga = gtab[g0], gb = gtab[min(g0 + 1, 31)];
      const f32x4 gm0 = *(const f32x4*)(smem + OFF_GB + ch0 * 4), gm1 = *(const f32x4*)(smem + OFF_GB + ch0 * 4 + 16);
      const f32x4 bt0 = *(const f32x4*)(smem + OFF_GB + (Cin + ch0) * 4), bt1 = *(const f32x4*)(smem + OFF_GB + (Cin + ch0) * 4 + 16);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool second = j >= nfirst;
        const float gm = j < 4 ? gm0[j & 3] : gm1[j & 3], bt = j < 4 ? bt0[j & 3] : bt1[j & 3];
        o[j] = (f16)gn_apply_elem(xv[i][j >> 2][j & 3], second ? gb.x : ga.x, second ? gb.y : ga.y, gm, bt, 1);
      }
    }
    *(f16x8*)(smem + OFF_H + buf * GC_HBYTES + hrow[i]) = o;
  };
  load_chunk(0, xa);
  if (nch > 1) load_chunk(1, xb);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // X0: the tables are everybody's (tid < 256 waited for its accumulator words)
  GC_STAMP();
#pragma unroll
  for (int i = 0; i < GC_ITEMS; ++i) convert_item(0, i, 0, xa);
  GC_STAMP();

  const int rsw = (l31 >> 1) & 7;
  const int b_frag = (wave * 32 + l31) * 128;
  int cur = 0;
  auto unit_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto unit_end = [&]() { cur = (cur + 1 == NS) ? 0 : cur + 1; };
  auto fragb = [&](int ks) { return *(const f16x8*)(smem + cur * GC_UNIT + b_frag + (((ks * 2 + lg) ^ rsw) << 4)); };

  f32x16 acc[1][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

  // one chunk: 18 units; xload <- chunk c + 2 (free: it held chunk c, normalised during chunk c - 1), xconv = chunk c + 1
  auto chunk = [&](int c, f32x4 (&xload)[GC_ITEMS][2], const f32x4 (&xconv)[GC_ITEMS][2]) {
    const int hb = OFF_H + (c & 1) * GC_HBYTES;
    const bool more = c + 1 < nch, more2 = c + 2 < nch;  // (wave-uniform)
    gc_static_for<9>([&](auto tc) {
      constexpr int tap = decltype(tc)::value;
      // A fragment rows of tap (ky, kx): halo pixel ky * 34 + kx + l31; the 16-byte chunk index is (2 ks + lg) ^ swizzle(row)
      const int rowt = (tap / 3) * GC_HW2 + (tap % 3) + l31;
      const int ab = hb + ((rowt << 7) | ((lg ^ ((rowt >> 1) & 7)) << 4));
      f16x8 fa[4], fb[4];
      unit_sync();
      if constexpr (tap == 0) { if (more2) load_chunk(c + 2, xload); }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fa[ks] = *(const f16x8*)(smem + (ab ^ (ks << 5)));
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fb[ks] = fragb(ks);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks], fb[ks], acc[0][0], 0, 0, 0);
      if constexpr (tap == 3 || tap == 5 || tap == 7) { if (more) convert_item(c + 1, (tap - 3) / 2, (c + 1) & 1, xconv); }
      unit_end();
      unit_sync();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fb[ks] = fragb(ks);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks], fb[ks], acc[0][1], 0, 0, 0);
      unit_end();
    });
  };
  for (int c = 0; c < nch; c += 2) {
    chunk(c, xa, xb);
    GC_STAMP();
    if (c + 1 < nch) { chunk(c + 1, xb, xa); GC_STAMP(); }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the loaders have drained their queue: the ring is free
  GC_STAMP();
  igemm_epilogue<GC_ROWS, GC_N, 1, GC_NWC, RING>(ep, acc, m0, 0, 0, tile, 0, smem);
#ifdef SDMI_RC_TIMING
  GC_STAMP();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GC_STAMP();
  if (rp.dbg && tid == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) rp.dbg[(size_t)blockIdx.x * 16 + i] = i < n_stamp ? stamp[i] : 0;
  }
#endif
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace

// May conv3x3(SiLU(GroupNorm32(cat(x0, x1)))) with `Cout` output channels over B x H x W pixels run as one gn_conv3_kernel launch?
bool gn_conv3_supported(int B, int H, int W, int c0, int c1, int Cout) {
  const int Cin = c0 + c1;
  return Cout == GC_N && W % GC_ROWS == 0 && Cin % 64 == 0 && c0 % 64 == 0 && Cin >= 256 && Cin <= GC_MAXC && Cin % 32 == 0 &&
         (int64_t)B * H * W * Cout * 4 < ((int64_t)1 << 31) && (int64_t)B * H * W * Cin * 4 < ((int64_t)1 << 31);
}

#ifdef SDMI_RC_TIMING
static long long* g_gc_dbg = nullptr;
static int g_gc_launch = 0;
extern "C" int sdmi_k_gn_conv3_dbg(void* stamps) { g_gc_dbg = (long long*)stamps; g_gc_launch = 0; return 0; }
#endif

int launch_gn_conv3(const GnConvParams& p, hipStream_t stream) {
  const IGemmParams& e = p.epi;
  SDMI_CHECK(gn_conv3_supported(e.B, e.Hout, e.Wout, p.c0, p.c1, e.N), "gn_conv3: N = 320, W % 32 = 0, 256 <= Cin <= 960 in 64-channel chunks");
  SDMI_CHECK(p.x0 && (p.c1 == 0 || p.x1) && p.gn_acc && p.gn_gamma && p.gn_beta && p.w, "gn_conv3: null operand");
  SDMI_CHECK(e.mode == EPI_PLAIN && e.ksize == 3 && e.stride == 1 && !e.up && e.Hin == e.Hout && e.Win == e.Wout && e.K == 9 * (p.c0 + p.c1) &&
             e.M == e.B * e.Hout * e.Wout && e.out_f32 && e.ldo % 4 == 0, "gn_conv3: a stride-1 3x3 convolution descriptor");
  SDMI_CHECK(!e.lnf_part && !e.lnp_out && !e.f16_scale && !e.ln_out && !e.out_lo, "gn_conv3: plain epilogue only");
  if (e.gn_n) {
    SDMI_CHECK(e.gn_n <= 2 && (e.Hout * e.Wout) % 32 == 0, "GroupNorm statistics need Hout*Wout % 32 == 0");
    for (int t = 0; t < e.gn_n; ++t)
      SDMI_CHECK(e.gn_acc[t] && e.gn_cpg[t] >= 2 && (e.gn_cbase[t] + e.N + e.gn_cpg[t] - 1) / e.gn_cpg[t] <= 32, "bad GroupNorm statistics target");
  }
  GnConvParams q = p;
#ifdef SDMI_RC_TIMING
  q.dbg = g_gc_dbg ? g_gc_dbg + (size_t)((g_gc_launch++) % 16) * 4096 * 16 : nullptr;     // [launch % 16][workgroup <= 4096][16 stamps]
#endif
  q.epi.splitk = 1; q.epi.splitk_fused = 0; q.epi.slab_tiled = 0;
  q.epi.epi_vec = epi_vec_ok(e);
  SDMI_CHECK((int64_t)e.M < (int64_t)65536 * e.Hout * e.Wout, "fast_div_hw: at most 65535 samples per launch");
  q.epi.magic_hw = div_magic_hw(e.Hout * e.Wout);
  q.epi.magic_w = div_magic(e.Wout);
  for (int t = 0; t < e.gn_n; ++t) q.epi.gn_magic[t] = div_magic(e.gn_cpg[t]);
  const double M = e.M, Cin = p.c0 + p.c1;
  char name[96];
  snprintf(name, sizeof name, "gnconv3_32x320w5_M%d_N%d_K%d", e.M, e.N, e.K);
  // algorithmic work of the reference ops: the convolution's flops; x read once (fp32), the weights once, the output written once
  ProfScope ps(name, 2.0 * M * e.N * e.K, M * Cin * 4.0 + (double)e.N * e.K * 2.0 + M * e.N * (4.0 + (e.residual ? 4.0 : 0.0) + (e.out_f16 ? 2.0 : 0.0)), stream);
  const dim3 grid(e.M / GC_ROWS);
  // ring depth and the prefetch wave (read per launch: A/B)
  const int ns = env_int("SDMI_GN_CONV_NS", 4), pf = env_int("SDMI_GN_CONV_PF", 1);
  if (ns == 6) {
    if (pf) SDMI_LAUNCH((gn_conv3_kernel<2, 6, true>), grid, dim3(GC_NTC + 192), 0, stream, q);
    else SDMI_LAUNCH((gn_conv3_kernel<2, 6, false>), grid, dim3(GC_NTC + 128), 0, stream, q);
  } else {
    if (pf) SDMI_LAUNCH((gn_conv3_kernel<2, 4, true>), grid, dim3(GC_NTC + 192), 0, stream, q);
    else SDMI_LAUNCH((gn_conv3_kernel<2, 4, false>), grid, dim3(GC_NTC + 128), 0, stream, q);
  }
  SDMI_HIP_OK(hipGetLastError());
  return 0;
}

}  // namespace sdmi
