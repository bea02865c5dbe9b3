// Fused GroupNorm(32) + SiLU + 3x3 convolution (stride 1, pad 1) for the ResBlock convs at the high-resolution
// levels -- `in_layers` / `out_layers` of ResBlock._forward (ldm/modules/diffusionmodules/openaimodel.py:201-204,
// 225-231, 263-275): conv3x3(SiLU(GroupNorm32(x))) + bias (+ time-embedding row vector) (+ residual).
//
// Why a second conv kernel: the implicit-GEMM kernel (igemm.hip) re-reads every input pixel once per tap and per
// N-tile, and on MI355X that kernel family is bound by L2->LDS bytes in flight (profiles/ablate_r01.txt).  Here a block
// owns an 8x16 patch of output pixels; per 64-channel chunk it stages the 10x18 *halo* patch of the input ONCE,
// applying the GroupNorm affine + SiLU on the way (fp32 stream in, fp16 MFMA operand out -- rounded once), and all
// nine taps read their A fragments from that LDS tile at shifted pixel offsets.  A traffic drops ~6x and the separate
// normalise pass (fp16 activation write + re-read) disappears.  Weights stream per (chunk, tap) through LDS-DMA
// (global_load_lds_dwordx4), double buffered, exactly like igemm.hip; K order is chunk-major (see pack_conv_kernel).
//
//   block  = 256 threads = 4 waves (2 x 2), output tile 128 pixels (8 x 16) x 64 channels, wave tile 64 x 32
//   MFMA   = v_mfma_f32_32x32x16_f16; a 32-row MFMA tile = 2 image rows of 16 pixels
//   LDS    = A halo 192 rows x 128 B (180 used) + weight ring 6 x 64 x 128 B = 72 KB  -> 2 blocks / CU
//   A tile is XOR-swizzled per halo pixel hp with ((hp >> 1) & 7) on 16-byte chunks (same scheme as igemm.hip)
#include "common.h"
#include "prof.h"

namespace sdmi {
namespace {

constexpr int TH = 8, TW = 16, HW2 = TW + 2, HROWS = (TH + 2) * (TW + 2);   // 180 halo pixels
constexpr int AROWS = 192;                                                    // padded to 6 passes of 32 rows
constexpr int BN = 64;
constexpr int A_STAGE = AROWS * 128, B_STAGE = BN * 128;
constexpr int RB = 6;                                                         // weight-tile ring depth (LDS-DMA)

__global__ void __launch_bounds__(256) conv3gn_kernel(const Conv3GnParams p, const int tiles_x, const int tiles_y,
                                                      const int tiles_n, const int chunks_per_split) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[A_STAGE + RB * B_STAGE];   // 24 + 48 KB -> 2 blocks / CU
  unsigned char* const Asm = smem;
  unsigned char* const Bsm = smem + A_STAGE;

  // ---- tile assignment (XCD-aware remap as in igemm.hip; speed only) ---------------------------------------------
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tiles_m = p.B * tiles_y * tiles_x;
  const int tiles_mn = tiles_m * tiles_n;
  const int split = wgid / tiles_mn;
  const int tmn = wgid - split * tiles_mn;
  const int tile_n = tmn / tiles_m;
  int tm = tmn - tile_n * tiles_m;
  const int b = tm / (tiles_y * tiles_x);
  tm -= b * tiles_y * tiles_x;
  const int tyi = tm / tiles_x, txi = tm - tyi * tiles_x;
  const int y0 = tyi * TH, x0 = txi * TW, n0 = tile_n * BN;
  const int Cin = p.c0 + p.c1;
  const int nchunks = Cin / 64;
  const int ch_begin = split * chunks_per_split;
  const int ch_end = min(nchunks, ch_begin + chunks_per_split);
  if (ch_begin >= ch_end) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int l31 = lane & 31, lg = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- A staging: thread -> (halo row = tid >> 3 (+32 per pass), channel octet = tid & 7) -----------------------------
  const int oct = tid & 7;
  int a_src[6];        // pixel index b*H*W + y*W + x of the halo row, or -1 outside the image
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int hp = (tid >> 3) + i * 32;
    int v = -1;
    if (hp < HROWS) {
      const int hy = hp / HW2, hx = hp - hy * HW2;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      if (y >= 0 && y < p.H && x >= 0 && x < p.W) v = (b * p.H + y) * p.W + x;
    }
    a_src[i] = v;
  }
  const int cpg = Cin / 32;
  // fold this batch row's GroupNorm accumulators (8 lanes per group) into an LDS {mean, rstd} table
  __shared__ float s_tab[64];
  float* const s_stats = s_tab;
  {
    const long long* src = p.acc + ((size_t)(b * 32 + (tid >> 3)) * GN_SLOTS + (tid & 7)) * GN_STRIDE;
    long long s = src[0], sl = src[1], ss = src[2], ssl = src[3];
#pragma unroll
    for (int o = GN_SLOTS / 2; o >= 1; o >>= 1) {
      s += __shfl_xor(s, o); sl += __shfl_xor(sl, o); ss += __shfl_xor(ss, o); ssl += __shfl_xor(ssl, o);
    }
    if ((tid & 7) == 0) {
      const double n = (double)cpg * (double)p.H * (double)p.W;
      const double m = gn_acc_value(s, sl) / n;
      double var = gn_acc_value(ss, ssl) / n - m * m;
      if (var < 0.0) var = 0.0;
      s_stats[(tid >> 3) * 2] = (float)m;
      s_stats[(tid >> 3) * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
  }
  __syncthreads();
  f32x4 stage[6][2];                       // raw fp32 input of the next chunk (12 x 16 B in flight per thread)
  float scale[8], shift[8];                // y = x * scale + shift  (GroupNorm affine folded)

  auto load_chunk = [&](int chunk) {       // issue the global loads of chunk's halo patch into registers
    const int c = chunk * 64 + oct * 8;    // first of this thread's 8 channels (concat index)
    const float* src; int ldc, coff;
    if (c < p.c0) { src = p.x0; ldc = p.c0; coff = c; } else { src = p.x1; ldc = p.c1; coff = c - p.c0; }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (a_src[i] >= 0) {
        const float* g = src + (size_t)a_src[i] * ldc + coff;
        stage[i][0] = *(const f32x4*)g;
        stage[i][1] = *(const f32x4*)(g + 4);
      } else {
        stage[i][0] = f32x4{0, 0, 0, 0}; stage[i][1] = f32x4{0, 0, 0, 0};
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cc = c + j;
      const int g = cc / cpg;
      const float rs = s_tab[g * 2 + 1] * p.gamma[cc];
      scale[j] = rs;
      shift[j] = p.beta[cc] - s_tab[g * 2] * rs;
    }
  };
  auto store_chunk = [&]() {               // normalise + SiLU + fp16 -> LDS halo tile (zeros outside the image)
    unsigned char* As = Asm;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int hp = (tid >> 3) + i * 32;
      f16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = stage[i][j >> 2][j & 3];
        float t = x * scale[j] + shift[j];
        t = t * __builtin_amdgcn_rcpf(1.0f + __expf(-t));     // SiLU; 1-ulp reciprocal, the result is rounded to fp16 anyway
        o[j] = (a_src[i] >= 0) ? (f16)t : (f16)0.f;
      }
      *(f16x8*)(As + hp * 128 + ((oct ^ ((hp >> 1) & 7)) << 4)) = o;
    }
  };

  // ---- B staging (weights): LDS-DMA, rows n0 + (tid >> 3) + 32 i, chunk position tid & 7 ------------------------------
  const int cpos = tid & 7, lrow = tid >> 3;
  const int gch = cpos ^ ((lrow >> 1) & 7);
  const int K = 9 * Cin;
  int b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) b_off[i] = (n0 + i * 32 + lrow) * K + gch * 8;       // N % 64 == 0: always valid
  auto issue_b = [&](int it, int stg) {    // it = chunk * 9 + tap = k-tile index in the packed weight
    const f16* wk = p.w + (size_t)it * 64;
    unsigned char* Bs = Bsm + stg * B_STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wk + b_off[i]),
                                       (__attribute__((address_space(3))) void*)(Bs + (i * 32 + wave_u * 8) * 128), 16, 0, 0);
  };

  // ---- MFMA fragment addressing -----------------------------------------------------------------------------------
  // A: MFMA row tile mi = wm*2 + i covers tile rows ty = 2*mi + (l31 >> 4), tx = l31 & 15
  int hp0[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) hp0[i] = (2 * (wm * 2 + i) + (l31 >> 4)) * HW2 + (l31 & 15);
  const int brow = wn * 32 + l31;
  const int bsw = (l31 >> 1) & 7;

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // ---- main loop over k-tiles it = chunk * 9 + tap -----------------------------------------------------------------------
  // Weights: ring of RB LDS tiles filled by LDS-DMA, RB-1 tiles in flight, retired by a counted s_waitcnt vmcnt(N)
  // across a raw s_barrier (the per-tap MFMA work, 8 MFMAs per wave, is far shorter than the L2 latency).
  // Input: ONE halo stage; the next chunk's patch is loaded into registers at tap 0 (in flight during 9 taps of MFMAs)
  // and normalised + written to LDS after tap 8, between two barriers.
  const int it_begin = ch_begin * 9, it_end = ch_end * 9;
  load_chunk(ch_begin);
#pragma unroll
  for (int s0 = 0; s0 < RB - 1; ++s0) issue_b(min(it_begin + s0, it_end - 1), s0);
  store_chunk();
  int tap = 0, chunk = ch_begin, slot = 0, nslot = RB - 1;
  for (int it = it_begin; it < it_end; ++it) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");            // 2 DMA per thread per tile, RB-2 = 4 newer tiles may fly
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (!(p.debug & 1)) issue_b(min(it + RB - 1, it_end - 1), nslot);   // past the end: harmless reload into a free slot
    const bool more_chunks = (chunk + 1 < ch_end);
    if (tap == 0 && more_chunks && !(p.debug & 4)) load_chunk(chunk + 1);
    const int ky = tap / 3, kx = tap - ky * 3;
    const unsigned char* As = Asm;
    const unsigned char* Bs = Bsm + slot * B_STAGE + brow * 128;
    int hp[2], asw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { hp[i] = hp0[i] + ky * HW2 + kx; asw[i] = (hp[i] >> 1) & 7; }
    if (!(p.debug & 2))
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = ks * 2 + lg;
      const f16x8 bf = *(const f16x8*)(Bs + ((c ^ bsw) << 4));
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f16x8 af = *(const f16x8*)(As + hp[i] * 128 + ((c ^ asw[i]) << 4));
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[i], 0, 0, 0);
      }
    }
    slot = (slot + 1 == RB) ? 0 : slot + 1;
    nslot = (nslot + 1 == RB) ? 0 : nslot + 1;
    if (++tap == 9) {
      tap = 0; ++chunk;
      if (more_chunks) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done reading the halo tile
        if (!(p.debug & 4)) store_chunk();                                  // published by the next iteration's barrier
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue (tiles are always full: H % 8 == 0, W % 16 == 0, N % 64 == 0) -----------------------------------------
  const int n = n0 + wn * 32 + l31;
  const bool slab_mode = p.splitk > 1;
  float colv = 0.f;
  if (!slab_mode) {
    if (p.bias) colv = p.bias[n];
    if (p.rowvec) colv += p.rowvec[(size_t)b * p.ld_rowvec + n];
  }
  float* slab = slab_mode ? (p.splitk_ws + (size_t)split * ((size_t)p.B * p.H * p.W) * p.N) : nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    size_t row[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ml = (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;     // row inside the 128-pixel tile
      row[r] = (size_t)((b * p.H + y0 + (ml >> 4)) * p.W + x0 + (ml & 15));
    }
    if (slab_mode) {
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[row[r] * p.N + n] = acc[i][r];
    } else {
      float resv[16];
      if (p.residual) {
#pragma unroll
        for (int r = 0; r < 16; ++r) resv[r] = p.residual[row[r] * p.ldr + n];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) resv[r] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) p.out[row[r] * p.ldo + n] = acc[i][r] + colv + resv[r];
    }
  }
}

}  // namespace

bool conv3gn_supported(int B, int H, int W, int c0, int c1, int N) {
  const int Cin = c0 + c1;
  return H % TH == 0 && W % TW == 0 && Cin % 64 == 0 && c0 % 64 == 0 && N % BN == 0 && (Cin / 32) >= 2 &&
         (int64_t)B * H * W * std::max(c0, std::max(c1, 1)) < ((int64_t)1 << 31) && (int64_t)N * 9 * Cin < ((int64_t)1 << 31);
}

int launch_conv3gn(const Conv3GnParams& p, hipStream_t stream) {
  SDMI_CHECK(conv3gn_supported(p.B, p.H, p.W, p.c0, p.c1, p.N), "conv3gn: unsupported shape");
  SDMI_CHECK(p.x0 && p.acc && p.gamma && p.beta && p.w && p.out, "conv3gn: missing pointer");
  SDMI_CHECK(p.c1 == 0 || p.x1 != nullptr, "conv3gn: second source missing");
  const int tiles_x = p.W / TW, tiles_y = p.H / TH, tiles_n = p.N / BN;
  const int nchunks = (p.c0 + p.c1) / 64;
  const long blocks = (long)p.B * tiles_x * tiles_y * tiles_n;
  const int64_t MN = (int64_t)p.B * p.H * p.W * p.N;
  int splitk = p.splitk;
  if (splitk <= 0) {          // auto: >= ~320 blocks, >= 3 chunks (27 k-tiles) per split
    splitk = 1;
    while (blocks * splitk < 320 && nchunks / (splitk * 2) >= 3 && splitk < 16 && p.splitk_ws &&
           (int64_t)(splitk * 2) * MN <= p.splitk_ws_floats)
      splitk *= 2;
  }
  if (splitk > 1) SDMI_CHECK(p.splitk_ws && (int64_t)splitk * MN <= p.splitk_ws_floats, "conv3gn: split-K workspace too small");
  const int cps = cdiv(nchunks, splitk);
  const int nsplit = cdiv(nchunks, cps);
  Conv3GnParams q = p;
  q.splitk = nsplit;
  static const char* abl = getenv("SDMI_CONV3GN_ABLATE");
  q.debug = abl ? atoi(abl) : 0;
  {
    const double M = (double)p.B * p.H * p.W, Cin = p.c0 + p.c1;
    ProfScope ps("conv3gn_8x16x64", 2.0 * M * p.N * 9.0 * Cin, M * Cin * 4.0 + (double)p.N * 9.0 * Cin * 2.0 + M * p.N * 4.0 +
                 (p.residual ? M * p.N * 4.0 : 0.0), stream);
    hipLaunchKernelGGL(conv3gn_kernel, dim3((unsigned)(blocks * nsplit)), dim3(256), 0, stream, q, tiles_x, tiles_y, tiles_n, cps);
    SDMI_HIP_OK(hipGetLastError());
  }
  if (nsplit > 1) {
    IGemmParams r;             // reuse the deterministic slab reduction of igemm.hip
    r.M = p.B * p.H * p.W; r.N = p.N; r.Hout = p.H; r.Wout = p.W; r.B = p.B;
    r.bias = p.bias; r.rowvec = p.rowvec; r.ld_rowvec = p.ld_rowvec; r.residual = p.residual; r.ldr = p.ldr;
    r.out_f32 = p.out; r.ldo = p.ldo; r.splitk_ws = p.splitk_ws; r.splitk_ws_floats = p.splitk_ws_floats;
    return launch_splitk_reduce(r, nsplit, stream);
  }
  return 0;
}

}  // namespace sdmi
