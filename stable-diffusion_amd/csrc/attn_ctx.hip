// Cross-attention with the query projection inside the kernel (gfx950 / MI355X):
//
//     out = softmax( (x Wq^T) K^T * scale ) V            CrossAttention.forward with context, ldm/modules/attention.py:170-193
//
// where x is the (LayerNorm'ed) token stream [rows][C], Wq the to_q weight [C][C] (attention.py:161) and K / V^T the per-prompt
// cached projections of the 77 context tokens (sdmi_unet_cache_context).  Rounds 1-2 ran this as TWO launches -- the to_q GEMM with
// the per-head scatter epilogue, then the flash-attention kernel over two 64-key tiles: 16 + 16 launches of 10-15 us per UNet call
// for 0.3 % of its FLOPs, every one of them a latency chain (profiles/unet_call_by_shape_r03.txt; a launch costs ~5 us before it
// does anything).  One launch here:
//   * a workgroup owns one head and NW x 32 queries; K and V^T of that head (<= 128 keys) are requested by LDS-DMA first of all and
//     land while the projection runs (weight rows of the head through a double-buffered LDS-DMA stage, the token row of each lane
//     straight into registers: 16 bytes per lane and k-step, nothing to share between waves);
//   * q^T = Wq_head x^T on v_mfma_f32_32x32x16_f16 (A = weight rows = head dims, B = token rows): the accumulator layout then holds
//     one QUERY per lane and 16 head dims per 32-dim tile in its registers -- i.e. q^T is already the B operand of S^T = K q^T once
//     the contraction index of that product is taken in the accumulator's register order (registers 8h .. 8h+7 of dim tile t <-> dims
//     32t + 16h + 4g + {0..3, 8..11}); the K fragments are read from LDS in the same order (two ds_read_b64), so q never leaves the
//     registers: no per-head q buffer, no scatter epilogue, no second launch;
//   * all keys are present at once: plain softmax (one max, one sum), P fed back from the accumulator registers as the B operand of
//     O^T = V^T P^T with the same permuted contraction (as attn.hip);
//   * optional LayerNorm fold (IGemmParams::lnf_*): x = fp16(gamma * t) and q = rstd (acc - mean cs) + d per query row, exactly what
//     the to_q GEMM's epilogue did.
// Operand tiles in LDS are [rows][128 B] with the 16-byte chunks XOR-swizzled by (row >> 1) & 7 on the DMA source side, as everywhere.
// Only LDS-DMA is in flight inside the loops; the few register loads of the prologue are issued behind the DMA requests and drained
// with vmcnt(0) (the two kinds do not retire through one in-order queue, profiles/gn_fold_r03.txt).
#include "igemm_dev.h"

#ifdef SDMI_EXPERIMENTS      // (to_q inside the cross-attention kernel: measured slower in round 3, profiles/experiments_r03.txt)
namespace sdmi {
namespace {

typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));

template <int D, int NW>
__global__ void __launch_bounds__(NW * 64) attn_ctx_kernel(const AttnCtxParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int DVT = (D + 31) / 32;                 // 32-dim tiles of the head
  constexpr int NCH = (D + 63) / 64;                 // 64-half chunks of a K row
  constexpr int KEYS = 128;                          // padded key count (two V^T sub-tiles of 64 keys)
  constexpr int KB_MAX = KEYS / 32;
  constexpr int K_BYTES = NCH * KEYS * 128;          // K: NCH sub-tiles [128 keys][64 halves]
  constexpr int V_BYTES = 2 * DVT * 32 * 128;        // V^T: 2 sub-tiles [DVT * 32 dims][64 keys]
  constexpr int WROWS = DVT * 32;
  constexpr int STAGE = WROWS * 128;                 // weight rows (head dims) of one 64-channel chunk
  // Weight chunks are double buffered: chunk kt + 1 is requested while chunk kt is multiplied; every chunk costs one L2 round trip
  // (~0.8 us: 5 / 10 / 20 of them for C = 320 / 640 / 1280 -- 21.6 / 24.0 / 38.8 us per launch against 25.4 / 22.6 / 25.7 us for the
  // two launches this replaces, pass N).  Requesting whole PHASES of 3-5 chunks at once (one wait per phase) was measured WORSE
  // (33 / 43 / 69 us on a slower box): the larger LDS footprint leaves one workgroup per CU and every request of every workgroup
  // lands at kernel start.
  constexpr int PH = 2;
  // {cs, d} of this head's dims (LayerNorm fold): in the rows D .. DVT * 32 - 1 of V^T sub-tile 0 where the head dim leaves some
  // (they are never requested: the O^T rows they would produce are not stored), behind the stages otherwise (d = 160)
  constexpr bool TAB_IN_V = (DVT * 32 - D) * 128 >= 2 * DVT * 32 * 4;
  constexpr int TAB_OFF = TAB_IN_V ? K_BYTES + D * 128 : K_BYTES + V_BYTES + PH * STAGE;
  constexpr int LDS_BYTES = K_BYTES + V_BYTES + PH * STAGE + (TAB_IN_V ? 0 : 2 * DVT * 32 * 4);
  static_assert(D % 8 == 0 && LDS_BYTES <= 160 * 1024, "LDS budget");
  constexpr int OOB = (int)0x80000000;

  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
  unsigned char* const Ks = smem;
  unsigned char* const Vs = smem + K_BYTES;
  unsigned char* const Ws = smem + K_BYTES + V_BYTES;
  float* const tab = (float*)(smem + TAB_OFF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lg = lane >> 5;
  const int r8 = lane >> 3, cp = lane & 7;
  const int bh = blockIdx.y, b = bh / p.heads, head = bh - b * p.heads;
  const int q0 = blockIdx.x * (NW * 32);             // first query of the workgroup
  const int C = p.C;

  // ---- 1. LDS-DMA requests: K, V^T of this head, then the weight rows of the first channel chunk (octets of 8 rows dealt over the waves) ----
  const __amdgpu_buffer_rsrc_t rsrc_k =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.k + (size_t)bh * p.nkv * D), 0, p.nkv * D * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_v =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.vt + (size_t)bh * D * p.nkv_pad), 0, D * p.nkv_pad * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.wq + (size_t)head * D * C), 0, D * C * 2, 0x00020000);
  {
    constexpr int KOCT = NCH * KEYS / 8;             // K octets: (sub-tile c, 8 keys)
#pragma unroll
    for (int o0 = 0; o0 < KOCT; o0 += NW) {
      const int o = min(o0 + wave, KOCT - 1);
      const int c = o / (KEYS / 8), ro = o - c * (KEYS / 8), row = ro * 8 + r8;       // key
      const int gch = cp ^ ((row >> 1) & 7);
      const int d0 = c * 64 + gch * 8;
      const int voff = (row < p.nkv && d0 < D) ? (row * D + d0) * 2 : OOB;
      auto dst = (__attribute__((address_space(3))) void*)(Ks + (c * KEYS + ro * 8) * 128);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_k, dst, 16, voff, 0, 0, 0);
    }
    constexpr int VOPS = (D + 7) / 8;                // V^T octets per sub-tile that hold head dims (the rest is never read)
    constexpr int VOCT = 2 * VOPS;
#pragma unroll
    for (int o0 = 0; o0 < VOCT; o0 += NW) {
      const int o = min(o0 + wave, VOCT - 1);
      const int sub = o / VOPS, ro = o - sub * VOPS, row = ro * 8 + r8;               // dim
      const int gch = cp ^ ((row >> 1) & 7);
      const int key0 = sub * 64 + gch * 8;
      const int voff = (row < D && key0 < p.nkv_pad) ? (row * p.nkv_pad + key0) * 2 : OOB;
      auto dst = (__attribute__((address_space(3))) void*)(Vs + (sub * DVT * 32 + ro * 8) * 128);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_v, dst, 16, voff, 0, 0, 0);
    }
  }
  auto issue_w = [&](int kt, int st) {               // weight rows (head dims) [0, WROWS) x channels 64 kt .. 64 kt + 63
    constexpr int WOCT = WROWS / 8;
#pragma unroll
    for (int o0 = 0; o0 < WOCT; o0 += NW) {
      const int o = min(o0 + wave, WOCT - 1);
      const int dd = o * 8 + r8;
      const int gch = cp ^ ((dd >> 1) & 7);
      const int voff = dd < D ? (dd * C + gch * 8) * 2 : OOB;
      auto dst = (__attribute__((address_space(3))) void*)(Ws + st * STAGE + o * (8 * 128));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, dst, 16, voff, kt * 128, 0, 0);
    }
  };
  const int nkt = C / 64;
  issue_w(0, 0);

  // ---- 2. register loads, BEHIND every DMA request above and drained by the vmcnt(0) of the loop: this lane's token row
  // (B operand of the projection: lane (q, g) holds x[q][16 ks + 8 g .. + 8], 16 bytes, straight from memory -- every wave has
  // its own queries, nothing to share), the LayerNorm statistics of its query, {cs, d} of the head's dims ----
  const int q_lane = q0 + wave * 32 + l31;
  const f16* const xrow = p.x + ((size_t)b * p.nq + min(q_lane, p.nq - 1)) * C + lg * 8;
  f16x8 xb[2][4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xb[0][ks] = *(const f16x8*)(xrow + ks * 16);
  float mean = 0.f, rstd = 1.f;
  const bool fold = p.lnf_part != nullptr;
  if (fold) {
    const int m = b * p.nq + min(q_lane, p.nq - 1);
    const float2* src = (const float2*)p.lnf_part + m;
    float sx = 0.f, sq = 0.f;
    float2 pv[20];
#pragma unroll
    for (int j = 0; j < 20; ++j) pv[j] = src[(size_t)min(j, p.lnf_npart - 1) * p.M];
#pragma unroll
    for (int j = 0; j < 20; ++j)
      if (j < p.lnf_npart) { sx += pv[j].x; sq += pv[j].y; }
    const float inv_c = 1.0f / (32.0f * (float)p.lnf_npart);
    mean = sx * inv_c;
    const float var = fmaxf(sq * inv_c - mean * mean, 0.f);
    rstd = 1.0f / sqrtf(var + p.lnf_eps);
    for (int e = tid; e < DVT * 32; e += NW * 64) {
      const bool in = e < D;
      tab[e] = in ? p.lnf_cs[head * D + e] : 0.f;
      tab[DVT * 32 + e] = in ? p.lnf_d[head * D + e] : 0.f;
    }
  }

  // ---- 3. q^T = Wq_head x^T: one 64-channel chunk per iteration; weights double buffered in LDS, token row in registers ----
  f32x16 qacc[DVT];
#pragma unroll
  for (int t = 0; t < DVT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) qacc[t][r] = 0.f;
  const int rsw = (l31 >> 1) & 7;
  auto chunk = [&](int kt, const f16x8 (&xc)[4], f16x8 (&xn)[4]) {
    wait_vmcnt<0>();                                 // everything this wave requested so far has landed (chunk kt; first time: K, V^T too)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... everybody's; chunk kt - 1 is fully read
    if (kt + 1 < nkt) {
      issue_w(kt + 1, (kt + 1) & 1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xn[ks] = *(const f16x8*)(xrow + (kt + 1) * 64 + ks * 16);
    }
    const unsigned char* st = Ws + (kt & 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int ch = (((ks * 2 + lg) ^ rsw) << 4);
#pragma unroll
      for (int t = 0; t < DVT; ++t) {
        const f16x8 wa = *(const f16x8*)(st + (t * 32 + l31) * 128 + ch);
        qacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, xc[ks], qacc[t], 0, 0, 0);
      }
    }
  };
  for (int kt = 0; kt < nkt; kt += 2) {              // (C / 64 is odd for C = 320: the second half is guarded)
    chunk(kt, xb[0], xb[1]);
    if (kt + 1 < nkt) chunk(kt + 1, xb[1], xb[0]);
  }
  // (K, V^T landed before the first barrier of the loop; the {cs, d} table was written before it too)

  // ---- 4. q^T as fp16 B fragments, in the accumulator's register order: qf[t][h] = registers 8h .. 8h+7 of dim tile t ----
  const float sc = p.scale * 1.4426950408889634f;
  f16x8 qf[DVT][2];
#pragma unroll
  for (int t = 0; t < DVT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = qacc[t][r];
      if (fold) {
        const int dd = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
        v = fmaf(rstd, v - mean * tab[dd], tab[DVT * 32 + dd]);
      }
      qf[t][r >> 3][r & 7] = (f16)v;
    }

  // ---- 5. S^T = K q^T over all key blocks (contraction in the permuted order: lane half g, slot j <-> dim 32t + 16h + 4g + (j & 3) + 8 (j >> 2)) ----
  const int nkb = (p.nkv + 31) >> 5;                 // 32-key blocks that hold keys (<= KB_MAX)
  f32x16 s[KB_MAX];
#pragma unroll
  for (int kb = 0; kb < KB_MAX; ++kb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
    if (kb < nkb) {
      const int key = kb * 32 + l31;
      const int ksw = (key >> 1) & 7;
#pragma unroll
      for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          constexpr int DPAD = ((D + 15) / 16) * 16;
          const int d0 = t * 32 + h * 16;            // dims d0 + 4g + {0..3} and d0 + 4g + 8 + {0..3}
          if (d0 < DPAD) {                           // (compile-time after unrolling: skip a k-step entirely beyond the head dim)
            const int c = d0 >> 6, chk = (d0 & 63) >> 3;            // 16-byte chunk of dims d0 .. d0 + 7; the next one holds d0 + 8 ..
            const unsigned char* row = Ks + (c * KEYS + key) * 128 + lg * 8;
            const f16x4v lo = *(const f16x4v*)(row + ((chk ^ ksw) << 4));
            const f16x4v hi = *(const f16x4v*)(row + (((chk + 1) ^ ksw) << 4));
            const f16x8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[t][h], s[kb], 0, 0, 0);
          }
        }
    }
  }
  // ---- 6. softmax over the keys of this lane's query (registers hold keys kb * 32 + (r & 3) + 8 (r >> 2) + 4 g) ----
  float mx = -1e30f;
#pragma unroll
  for (int kb = 0; kb < KB_MAX; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
      if (key >= p.nkv) s[kb][r] = -1e30f;
      mx = fmaxf(mx, s[kb][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const float msc = mx * sc;
  float psum = 0.f;
  f16x8 pf[KB_MAX][2];
#pragma unroll
  for (int kb = 0; kb < KB_MAX; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const f16 ph = (f16)__builtin_amdgcn_exp2f(fmaf(s[kb][r], sc, -msc));       // (masked keys: exp2(-huge) = 0)
      pf[kb][r >> 3][r & 7] = ph;
      psum += (float)ph;                             // the denominator sums the same rounded probabilities the numerator uses
    }
  psum += __shfl_xor(psum, 32);

  // ---- 7. O^T = V^T P^T (contraction over keys in the same permuted order) ----
  f32x16 o[DVT];
#pragma unroll
  for (int t = 0; t < DVT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
#pragma unroll
  for (int kb = 0; kb < KB_MAX; ++kb) {
    if (kb < nkb) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k0 = kb * 32 + h * 16;             // keys k0 + 4g + {0..3} and k0 + 4g + 8 + {0..3}
        const int sub = k0 >> 6, chk = (k0 & 63) >> 3;
#pragma unroll
        for (int t = 0; t < DVT; ++t) {
          const int drow = t * 32 + l31;
          const int vsw = (drow >> 1) & 7;
          const unsigned char* row = Vs + (sub * DVT * 32 + drow) * 128 + lg * 8;
          const f16x4v lo = *(const f16x4v*)(row + ((chk ^ vsw) << 4));
          const f16x4v hi = *(const f16x4v*)(row + (((chk + 1) ^ vsw) << 4));
          const f16x8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pf[kb][h], o[t], 0, 0, 0);
        }
      }
    }
  }
  // ---- 8. normalise and store: out[b][q][head * D + dd] ----
  const float inv = 1.0f / psum;
  if (q_lane < p.nq) {
    f16* orow = p.out + ((size_t)b * p.nq + q_lane) * ((size_t)p.heads * D) + (size_t)head * D;
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int dd = t * 32 + 8 * r4 + 4 * lg;
        if (dd < D) {
          const f16x4v v = {(f16)(o[t][r4 * 4 + 0] * inv), (f16)(o[t][r4 * 4 + 1] * inv), (f16)(o[t][r4 * 4 + 2] * inv),
                            (f16)(o[t][r4 * 4 + 3] * inv)};
          *(f16x4v*)(orow + dd) = v;
        }
      }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

template <int D>
int launch_ctx_d(const AttnCtxParams& p, hipStream_t stream) {
  // waves per workgroup (32 queries each): 4 while that leaves >= 256 workgroups, else 2 while >= 128, else 1 -- these launches are
  // latency chains, more workgroups = more CUs working on them
  static const std::string pname = std::string("attn_ctx_q_d") + std::to_string(D);
  ProfScope ps(pname.c_str(), 2.0 * p.BH * (double)p.nq * D * p.C + 4.0 * p.BH * (double)p.nq * p.nkv * D,
               2.0 * ((double)p.BH / p.heads * p.nq * p.C + (double)p.C * p.C + (double)p.BH * D * 2.0 * p.nkv + (double)p.BH * p.nq * D), stream);
  static const int env_nw = env_int("SDMI_ATTN_CTX_NW", 0);       // A/B knob
  int nw = (long)cdiv(p.nq, 128) * p.BH >= 256 ? 4 : ((long)cdiv(p.nq, 64) * p.BH >= 128 ? 2 : 1);
  if (env_nw == 1 || env_nw == 2 || env_nw == 4) nw = env_nw;
  if (nw == 4) SDMI_LAUNCH((attn_ctx_kernel<D, 4>), dim3(cdiv(p.nq, 128), p.BH), dim3(256), 0, stream, p);
  else if (nw == 2) SDMI_LAUNCH((attn_ctx_kernel<D, 2>), dim3(cdiv(p.nq, 64), p.BH), dim3(128), 0, stream, p);
  else SDMI_LAUNCH((attn_ctx_kernel<D, 1>), dim3(cdiv(p.nq, 32), p.BH), dim3(64), 0, stream, p);
  SDMI_HIP_OK(hipGetLastError());
  ps.end();
  return 0;
}

}  // namespace

bool attention_ctx_supported(int d, int C, int nkv) { return (d == 40 || d == 80 || d == 160) && C % 64 == 0 && nkv >= 1 && nkv <= 128; }

int launch_attention_ctx(const AttnCtxParams& p, hipStream_t stream) {
  SDMI_CHECK(p.x && p.wq && p.k && p.vt && p.out, "attention_ctx: missing pointer");
  SDMI_CHECK(p.BH > 0 && p.heads > 0 && p.BH % p.heads == 0 && p.nq > 0 && p.C == p.heads * p.d, "attention_ctx: bad shape");
  SDMI_CHECK(attention_ctx_supported(p.d, p.C, p.nkv) && p.nkv_pad % 8 == 0 && p.nkv_pad >= p.nkv, "attention_ctx: head dim 40 / 80 / 160, <= 128 keys");
  SDMI_CHECK((int64_t)p.nq * p.C * 2 < ((int64_t)1 << 31) - 65536, "attention_ctx: token stream too large for 31-bit offsets");
  if (p.lnf_part) SDMI_CHECK(p.lnf_cs && p.lnf_d && p.lnf_npart * 32 == p.C && p.lnf_npart <= 20 && p.M > 0, "attention_ctx: bad LayerNorm fold");
  int rc;
  switch (p.d) {
    case 40: rc = launch_ctx_d<40>(p, stream); break;
    case 80: rc = launch_ctx_d<80>(p, stream); break;
    default: rc = launch_ctx_d<160>(p, stream); break;
  }
  if (rc == 0 && range_check_enabled()) return range_scan("cross-attention output", p.out, (int64_t)p.BH * p.nq * p.d, stream);
  return rc;
}

}  // namespace sdmi
#else
namespace sdmi {
bool attention_ctx_supported(int, int, int) { return false; }
int launch_attention_ctx(const AttnCtxParams&, hipStream_t) {
  return fail("attention with the to_q projection inside (attn_ctx.hip) is an experiment: build with SDMI_CXXFLAGS=-DSDMI_EXPERIMENTS");
}
}  // namespace sdmi
#endif
