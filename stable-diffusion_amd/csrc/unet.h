// UNet executor state (see unet.cpp).
#pragma once
#include <stdlib.h>

#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/sdmi.h"
#include "common.h"

namespace sdmi {

// bump allocator over a caller-provided workspace; `dry` only measures
struct Arena {
  char* base = nullptr; size_t cap = 0, off = 0, peak = 0; bool dry = false; bool overflow = false;
  void* alloc(size_t bytes) {
    const size_t a = (off + 255) & ~(size_t)255;
    off = a + bytes;
    peak = std::max(peak, off);
    if (dry) return (void*)(uintptr_t)(a + 4096);   // fake non-null address; never dereferenced
    if (off > cap) { overflow = true; return base; }
    return base + a;
  }
};

// an fp32 NHWC activation.  `id` / `stats` serve the GroupNorm-statistics fusion: an activation produced by an igemm
// epilogue (or its split-K reduce) can have the {sum, sumsq} of its consumers' GroupNorm groups accumulated there.
// `can_f16`: the producer is an igemm launch that can also store the fp16 rounding of its output (see want_f16 below).
struct Act { float* p = nullptr; int C = 0, H = 0, W = 0; int id = -1; bool stats = false; bool can_f16 = false; };

// one GroupNorm that will read an activation: accumulator region index, the activation's first channel inside that
// GroupNorm's (concatenated) input, channels per group
struct GnTarget { int gn_idx = 0, cbase = 0, cpg = 0; };
struct GnPlan {                       // built by the dry pass of a forward, consumed by the real pass
  std::vector<std::vector<GnTarget>> targets;    // per activation id
  std::vector<char> fused;                       // per GroupNorm call: statistics come from the producers' epilogues
  // per activation id: a Downsample / Upsample convolution will read it as an fp16 MFMA operand -- the producing GEMM's
  // epilogue (or its split-K reduce) stores that copy beside the fp32 output instead of a cast launch in front of the conv
  std::vector<char> want_f16;
  void clear() { targets.clear(); fused.clear(); want_f16.clear(); }
};

// state and helpers shared by the executors (UNet forward, first-stage encode / decode): arenas, GroupNorm accumulator
// regions, split-K slabs and the igemm / GroupNorm launch wrappers
struct FwdBase {
  hipStream_t s = nullptr; bool dry = false; int B = 1;
  const f16* zero = nullptr;      // zero page for out-of-image conv taps
  bool precise_1x1 = true;        // 1x1 convs on the residual stream as 3-pass split-fp16 GEMMs
  Arena persist, scratch;
  long long* gn_acc = nullptr;   // GN_MAX_CALLS regions of gn_acc_words(B) int64, zeroed once per forward
  int gn_calls = 0;
  float* splitk_ws = nullptr; int64_t splitk_ws_floats = 0;
  int* splitk_cnt = nullptr;                          // one int per output tile of a split-K GEMM (igemm.hip)
  static constexpr int SPLITK_CNT_INTS = 8192;
  // side stream (optional): work that is independent of the main chain for a while -- the ResBlock's 1x1 skip convolution
  // runs beside conv1 / GroupNorm 2 -- is enqueued there between two events; it has its own split-K slabs
  hipStream_t side = nullptr;
  hipEvent_t* side_ev = nullptr; int side_nev = 0, side_next = 0;
  float* splitk_ws2 = nullptr; int64_t splitk_ws2_floats = 0;
  int rc = 0;
  GnPlan* plan = nullptr;         // null: GroupNorm statistics always by the statistics kernel (first stage, text encoder)
  int n_acts = 0;

  // register an activation; `by_igemm`: its producer is an igemm epilogue / split-K reduce that can emit statistics
  Act make_act(float* ptr, int C, int H, int W, bool by_igemm) {
    Act a; a.p = ptr; a.C = C; a.H = H; a.W = W; a.id = n_acts++;
    a.stats = by_igemm && plan != nullptr && (H * W) % 32 == 0 && H * W >= 1024 / C + 2 && C % 4 == 0 && C / 32 >= 2;
    if (plan && dry && (int)plan->targets.size() < n_acts) plan->targets.resize(n_acts);
    if (plan && dry && (int)plan->want_f16.size() < n_acts) plan->want_f16.resize(n_acts, 0);
    return a;
  }
  // fp16 copy of activation `a` for a resampling convolution (openaimodel.py:116-118,150-153).  Dry pass: the CONSUMER
  // (want_f16_copy) records the wish and accounts the bytes; real pass: the PRODUCER's GEMM (attach_f16_copy) allocates the
  // copy and stores it from its epilogue, the consumer picks it up (f16_copy) -- same rounding of the same fp32 values as
  // the cast kernel it replaces.  The persist arena never rewinds, so the different allocation order of the two passes
  // does not change its size.
  std::vector<f16*> f16_of;
  void attach_f16_copy(IGemmParams& p, Act& a) {
    a.can_f16 = plan != nullptr && p.out_f16 == nullptr && p.mode == EPI_PLAIN;
    if (dry || !a.can_f16 || a.id < 0 || a.id >= (int)plan->want_f16.size() || !plan->want_f16[a.id]) return;
    f16* c = P<f16>((size_t)B * a.H * a.W * a.C);
    if ((int)f16_of.size() <= a.id) f16_of.resize(a.id + 1, nullptr);
    f16_of[a.id] = c;
    p.out_f16 = c;
  }
  bool want_f16_copy(const Act& a) {          // dry pass, consumer side; true: the copy will exist in the real pass
    if (!dry || !plan || !a.can_f16 || a.id < 0 || a.id >= (int)plan->want_f16.size()) return false;
    if (const char* e = getenv("SDMI_F16_COPY")) { if (atoi(e) == 0) return false; }   // A/B knob: 0 = cast launches (bit-identical)
    plan->want_f16[a.id] = 1;
    (void)P<f16>((size_t)B * a.H * a.W * a.C);
    return true;
  }
  const f16* f16_copy(const Act& a) const {   // real pass, consumer side
    return (!dry && a.id >= 0 && a.id < (int)f16_of.size()) ? f16_of[a.id] : nullptr;
  }
  // attach the statistics targets of activation `a` (all GroupNorms that will read it and rely on fused statistics)
  void attach_gn_targets(IGemmParams& p, const Act& a) {
    p.gn_n = 0;
    if (dry || !plan || !a.stats || a.id < 0 || a.id >= (int)plan->targets.size()) return;
    for (const GnTarget& t : plan->targets[a.id]) {
      if (p.gn_n >= 2) { ok(fail("internal: more than two GroupNorm consumers of one activation")); return; }
      p.gn_acc[p.gn_n] = gn_acc + (size_t)t.gn_idx * gn_acc_words(B);
      p.gn_cpg[p.gn_n] = t.cpg; p.gn_cbase[p.gn_n] = t.cbase;
      ++p.gn_n;
    }
  }

  template <class T> T* P(size_t n) { return (T*)persist.alloc(n * sizeof(T)); }
  template <class T> T* S(size_t n) { return (T*)scratch.alloc(n * sizeof(T)); }
  void ok(int r) { if (r && !rc) rc = r; }
  long long* next_gn_acc() {
    if (gn_calls >= GN_MAX_CALLS) { ok(fail("more GroupNorm calls than accumulator regions")); return gn_acc; }
    return gn_acc + (size_t)(gn_calls++) * gn_acc_words(B);
  }
  long long* last_gn_acc() const { return gn_acc + (size_t)(gn_calls - 1) * gn_acc_words(B); }
  // allocate + zero the accumulator regions and the split-K slabs (call once per pass, before any layer)
  int begin_pass(int64_t splitk_floats, bool with_gn = true) {
    gn_calls = 0;
    gn_acc = nullptr;
    // the split-K tile counters sit behind the accumulators so that one memset clears both (the counters reset themselves;
    // clearing them per pass only guards against a pass that was aborted mid-kernel)
    const size_t acc_words = with_gn ? (size_t)GN_MAX_CALLS * gn_acc_words(B) : 0;
    long long* blk = P<long long>(acc_words + SPLITK_CNT_INTS / 2);
    gn_acc = with_gn ? blk : nullptr;
    splitk_cnt = (int*)(blk + acc_words);
    if (!dry) SDMI_HIP_OK(memset_async(blk, 0, (acc_words + SPLITK_CNT_INTS / 2) * sizeof(long long), s));
    splitk_ws_floats = splitk_floats;
    splitk_ws = P<float>((size_t)splitk_floats);
    splitk_ws2_floats = splitk_floats / 2;               // (always: the workspace size must not depend on the side stream)
    splitk_ws2 = P<float>((size_t)splitk_ws2_floats);
    return 0;
  }
  // fork: the side stream continues from this point of the main stream; join: the main stream waits for the side stream
  hipEvent_t next_side_event() { hipEvent_t e = side_ev[side_next]; side_next = (side_next + 1) % side_nev; return e; }
  void fork_side() {
    if (dry || rc || !side) return;
    hipEvent_t e = next_side_event();
    if (hipEventRecord(e, s) != hipSuccess || hipStreamWaitEvent(side, e, 0) != hipSuccess) ok(fail("side stream fork failed"));
  }
  void join_side() {
    if (dry || rc || !side) return;
    hipEvent_t e = next_side_event();
    if (hipEventRecord(e, side) != hipSuccess || hipStreamWaitEvent(s, e, 0) != hipSuccess) ok(fail("side stream join failed"));
  }
  void gemm_side(IGemmParams& p) {        // like gemm(), on the side stream (falls back to the main stream without one)
    if (!side) { gemm(p); return; }
    p.zero_page = zero;
    p.splitk_ws = splitk_ws2; p.splitk_ws_floats = splitk_ws2_floats;
    p.splitk_cnt = nullptr; p.splitk_cnt_ints = 0;
    if (!dry && !rc) ok(launch_igemm(p, IGemmTune(), side));
  }

  void gemm(IGemmParams& p) {
    p.zero_page = zero;
    p.splitk_ws = splitk_ws; p.splitk_ws_floats = splitk_ws_floats;
    p.splitk_cnt = splitk_cnt; p.splitk_cnt_ints = SPLITK_CNT_INTS;
    if (!dry && !rc) ok(launch_igemm(p, IGemmTune(), s));
  }
  // dense [M][K] x W[N][K]^T
  IGemmParams dense(const f16* a, int M, int K, const f16* w, int N, int rows_per_batch) {
    IGemmParams p;
    p.a0 = a; p.c0 = K; p.lda0 = K;
    p.B = M / rows_per_batch; p.Hin = p.Hout = rows_per_batch; p.Win = p.Wout = 1;
    p.ksize = 1; p.w = w; p.M = M; p.N = N; p.K = K; p.splitk = 0;
    return p;
  }
  IGemmParams conv3(const f16* a, int C, int Hin, int Win, int Hout, int Wout, int stride, int up, const f16* w, int N) {
    IGemmParams p;
    p.a0 = a; p.c0 = C; p.lda0 = C;
    p.B = B; p.Hin = Hin; p.Win = Win; p.Hout = Hout; p.Wout = Wout;
    p.ksize = 3; p.stride = stride; p.up = up; p.w = w; p.M = B * Hout * Wout; p.N = N; p.K = 9 * C; p.splitk = 0;
    return p;
  }
  // 1x1 conv on split-fp16 operands (packed W_SPLIT3 weights [N][3K] = [hi | hi | lo]) when precise: the split-fp16 GEMM family
  // (gemm_split16.hip: four operand tiles per 64-channel chunk, three MFMAs per fragment pair); SDMI_SPLIT16_KERNEL=0 = the
  // rounds-1/2 formulation, one K-concatenated GEMM A' = [hi | lo | hi] through the generic kernel (A/B)
  IGemmParams dense1x1(const f16* hi, const f16* lo, int M, int K, const f16* w, int N, int rows_per_batch, bool precise) {
    IGemmParams p = dense(hi, M, K, w, N, rows_per_batch);
    if (precise) {
      static const bool family = !(getenv("SDMI_SPLIT16_KERNEL") && atoi(getenv("SDMI_SPLIT16_KERNEL")) == 0);
      if (family && K % 64 == 0) {
        p.a1 = lo; p.lda1 = K; p.split16 = 1; p.ldw = 3 * K;
      } else {
        p.a1 = lo; p.c1 = K; p.lda1 = K; p.a2 = hi; p.c2 = K; p.lda2 = K; p.K = 3 * K; p.k_alg = K;
      }
    }
    return p;
  }
  // stats_only: fill (or, when the producers' epilogues did, just locate) the statistics accumulators of this GroupNorm and
  // launch no apply kernel -- the consuming convolution normalises while it stages its input (IGemmParams::gn_in_acc).
  // Returns the accumulator region.
  // already_applied: the producing GEMM's split-K reduction normalised its output itself (IGemmParams::pgn_*): this call only
  // keeps the plan's bookkeeping (accumulator region, call index) in step and launches nothing.
  long long* groupnorm(const Act& x0, const Act* x1, const float* gamma, const float* beta, float eps, int silu, f16* o16,
                       float* o32, f16* raw, f16* o16_lo = nullptr, f16* raw_lo = nullptr, bool stats_only = false,
                       bool already_applied = false) {
    GroupNormParams g;
    g.x0 = x0.p; g.c0 = x0.C;
    if (x1) { g.x1 = x1->p; g.c1 = x1->C; }
    g.B = B; g.HW = x0.H * x0.W; g.gamma = gamma; g.beta = beta; g.eps = eps; g.silu = silu;
    g.out_f16 = o16; g.out_f32 = o32; g.raw_f16 = raw; g.out_lo = o16_lo; g.raw_lo = raw_lo;
    g.stats_only = stats_only ? 1 : 0;
    const int idx = gn_calls;
    g.acc = next_gn_acc();
    if (plan) {
      const int cpg = (g.c0 + g.c1) / 32;
      if (dry) {
        // statistics are fused iff EVERY source of this GroupNorm comes from a statistics-capable producer
        const bool f = x0.stats && (!x1 || x1->stats) && x0.id >= 0 && (!x1 || x1->id >= 0);
        if ((int)plan->fused.size() <= idx) plan->fused.resize(idx + 1, 0);
        plan->fused[idx] = f ? 1 : 0;
        if (f) {
          plan->targets[x0.id].push_back({idx, 0, cpg});
          if (x1) plan->targets[x1->id].push_back({idx, g.c0, cpg});
        }
      } else {
        g.skip_stats = (idx < (int)plan->fused.size() && plan->fused[idx]) ? 1 : 0;
      }
    }
    if (!dry && !rc && !(stats_only && g.skip_stats) && !already_applied) ok(launch_groupnorm(g, s));
    return g.acc;
  }
};

// host or device fp32 pointer -> device pointer (staged through a temporary device buffer when it is host memory)
struct DevStage {
  const float* dptr = nullptr; float* staged = nullptr;
  int acquire(const float* ptr, int64_t numel, hipStream_t stream);
  int release(hipStream_t stream);
};

enum LayerKind { L_CONV_IN, L_RES, L_ATTN, L_DOWN, L_UP };
enum WKind { W_F32, W_F32_ROWS, W_CONV, W_CONV_OUT, W_ROWS16, W_GEGLU_W, W_GEGLU_B, W_SPLIT3, W_SPLIT3_ROWS, W_CONV_SPLIT3 };

struct TBlock {   // BasicTransformerBlock (ldm/modules/attention.py:196-215)
  f16* wqkv = nullptr;   // [3C][C]   attn1 to_q | to_k | to_v
  f16* wo1 = nullptr; float* bo1 = nullptr;
  f16* wq2 = nullptr;    // [C][C]
  f16* wkv2 = nullptr;   // attn2 to_k | to_v: [2C][3 context_dim] split-fp16 {hi | hi | lo} (round 6; [2C][context_dim] in the experiments build with SDMI_PRECISE_KV=0)
  f16* wo2 = nullptr; float* bo2 = nullptr;
  f16* wgg = nullptr; float* bgg = nullptr;     // GEGLU proj, rows interleaved (value32 | gate32)
  f16* wff2 = nullptr; float* bff2 = nullptr;
  float* ln[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  f16* ck = nullptr; f16* cvt = nullptr;        // cached cross-attention K [B*h][L][d] and V^T [B*h][d][Lpad]
  // LayerNorm folded into the consuming GEMM (IGemmParams::lnf_cs / lnf_d; computed by finalize() from the packed weights):
  // {cs, d} of attn1 q|k|v with norm1, attn2 to_q with norm2, the GEGLU projection with norm3 (packed column order, bias inside d)
  float* lnf[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // the GEGLU projection's {cs, d} regrouped per hidden chunk of C for the row-strip chain (rowchain.hip, FfTailParams::csd):
  // [4][cs of 2C packed columns | d of the same 2C]; filled by finalize() where the chain geometry exists (C = 320)
  float* lnf_csd = nullptr;
};

struct Layer {
  LayerKind kind = L_RES;
  std::string prefix;
  int cin = 0, cout = 0, heads = 0, dh = 0, emb_off = 0, attn_index = -1;
  // Precision allocation by measured sensitivity (round 6; DESIGN.md section 2, oracle/fp16_floor.py):
  //   p1x1     the layer's 1x1 convs on the residual stream (skip_connection / proj_in / proj_out) as 3-pass split-fp16 -- the two upper levels
  //            (downsample factors 1 and 2): single-pass fp16 there would add 87 % / 15 % to the eps error variance, at downsample factor 4
  //            3.5 %, at 8 and in the middle block 0.1 % -- those run single-pass (-0.08 ms per UNet call);
  //   precise3 the ResBlock's two 3x3 convs as 3-pass split-fp16 -- the LAST ResBlock only, which alone carries 26 - 30 % of that variance
  bool p1x1 = true, precise3 = false;
  f16* w16[3] = {nullptr, nullptr, nullptr};
  float* w32[1] = {nullptr};
  float* f32[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  std::vector<TBlock> tb;
};

struct WeightSlot {
  std::string key;
  std::vector<int64_t> shape;
  WKind kind = W_F32;
  void** dst = nullptr; int row0 = 0, ld = 0; void** dst2 = nullptr;
  bool set = false;
};

class UNet {
 public:
  UNet() = default;
  ~UNet();
  UNet(const UNet&) = delete;
  UNet& operator=(const UNet&) = delete;

  int build(const sdmi_unet_cfg& cfg);
  int set_weight(const char* key, const float* ptr, const int64_t* shape, int ndim, hipStream_t stream);
  int finalize();
  // packed-weight blob: every packed device buffer behind a header that pins cfg / ABI
  int packed_layout(std::vector<std::pair<void**, size_t>>* bufs, int64_t* total) const;
  int export_packed(void* host_buf, int64_t bytes, hipStream_t stream);
  int import_packed(const void* host_buf, int64_t bytes, hipStream_t stream);
  // dry = size only; ctx_only = just the cross-attention K/V of every SpatialTransformer
  int run(const float* x, const int64_t* t_i64, const float* t_f32, const float* ctx, float* eps_out, int B, int H, int W,
          int Lctx, void* workspace, int64_t ws_bytes, hipStream_t stream, bool dry, bool ctx_only, int64_t* bytes_needed);

  // emb_all rows of a list of integer timesteps, computed in one batch (include/sdmi.h: sdmi_unet_cache_timesteps);
  // hint: the next run()'s rows all have timestep t
  int cache_timesteps(const int64_t* t_host, int n, hipStream_t stream);
  // grow-only capacity of the cross-attention K / V^T caches (allocates; never called by run() on the forward path)
  int reserve_ctx_cache(int B, int Lctx);
  int hint_timestep(int64_t t);

  const std::vector<WeightSlot>& slots() const { return slots_; }

  sdmi_unet_cfg cfg_{};
  int te_ = 0, emb_total_ = 0, n_attn_ = 0;
  f16* zero_ = nullptr;
  // 1x1 convs on the residual stream (skip_connection, proj_in, proj_out) run as 3-pass split-fp16 GEMMs
  // (a_hi*w_hi + a_lo*w_hi + a_hi*w_lo): ~22-bit operands for 5 % of the FLOPs (DESIGN.md "precision")
  bool precise_1x1_ = true;
  bool precise_last_res_ = true;   // the last ResBlock's 3x3 convs as 3-pass split-fp16 (round 6; Layer::precise3)
  int precise_1x1_max_ds_ = 4;     // stream 1x1 convs are split-fp16 at downsample factors below this (round 6; Layer::p1x1)
  bool precise_kv_ = true;     // context K / V projections as 3-pass split-fp16 (round 6; see UNet::build)
  // ResBlock convs fold the GroupNorm + SiLU of their input into their halo staging (conv3halo.hip, conv3halo_gn_kernel) wherever
  // gn_fold_conv_supported() says so; SDMI_FUSE_GN_CONV=0 restores the GroupNorm-apply launches (A/B).

 private:
  friend struct Fwd;
  void expect(const std::string& key, std::vector<int64_t> shape, WKind kind, void** dst, int row0 = 0, int ld = 0,
              void** dst2 = nullptr);
  int dev_alloc(void** dst, size_t bytes);
  size_t slot_bytes(const WeightSlot& s) const;
  int ensure_ctx_cache(int B, int Lctx, bool may_grow);
  GnPlan gn_plan_;           // GroupNorm-statistics fusion plan of the current forward (rebuilt by its dry pass)
  bool side_stream_ = false;    // SDMI_SIDE_STREAM=1: ResBlock skip convolutions on a side stream (measured 3 % slower)
  hipStream_t side_ = nullptr; hipEvent_t side_ev_[32] = {};
  bool fuse_gn_stats_ = true;   // SDMI_FUSE_GN_STATS=0: every GroupNorm runs its own statistics kernel (A/B, debugging)
  // The LayerNorms of a BasicTransformerBlock (attention.py:211-215) folded into the GEMMs that read them, where the producing
  // GEMM is not split (>= ln_fold_min_rows_ token rows; the fold pins its producers to split 1): no LayerNorm launch, one fp32 read of the token stream less per site.
  // SDMI_LN_FOLD=0 restores the launches (A/B); SDMI_LN_FOLD_MIN_ROWS moves the threshold.
  bool ln_fold_ = true; int ln_fold_min_rows_ = 512;
  // SpatialTransformer tails (GEGLU -> FF-out -> proj_out) as one row-strip chain launch where the geometry fits (the 64 x 64 level of
  // SD v1: C = 320); rides on the LayerNorm fold.  SDMI_FF_TAIL=0 restores the three launches (bit-identical outputs; A/B).
  bool ff_tail_ = true;
  // ... and SpatialTransformer heads (GroupNorm-apply -> proj_in -> q | k | v of the first transformer block) likewise.  SDMI_ST_HEAD=0
  // restores the three launches (bit-identical outputs; A/B).
  bool st_head_ = true;

  // ---- launch tapes (tape.h): the launch list of a (shape, workspace, mode, knobs) recorded once and replayed
  struct TapeKey {
    int B = 0, H = 0, W = 0, Lctx = 0, mode = 0;         // mode: 0 timestep-table row (hinted), 1 int64 timesteps, 2 fp32 timesteps
    const void* ws = nullptr; int64_t ws_bytes = 0; bool have_ctx = false;
    uint64_t env = 0, weights_gen = 0, ctx_gen = 0;
    bool operator==(const TapeKey& o) const {
      return B == o.B && H == o.H && W == o.W && Lctx == o.Lctx && mode == o.mode && ws == o.ws && ws_bytes == o.ws_bytes &&
             have_ctx == o.have_ctx && env == o.env && weights_gen == o.weights_gen && ctx_gen == o.ctx_gen;
    }
  };
  std::vector<std::pair<TapeKey, std::unique_ptr<Tape>>> tapes_;     // (most recently used last; at most kMaxTapes)
  static constexpr size_t kMaxTapes = 12;
  uint64_t weights_gen_ = 0, ctx_gen_ = 0;
 public:
  uint64_t tape_hits_ = 0, tape_records_ = 0;            // (sdmi_unet_tape_stats: tests / bench)
 private:

  std::vector<std::vector<Layer>> input_blocks_, output_blocks_;
  std::vector<Layer> middle_;
  std::vector<WeightSlot> slots_;
  std::map<std::string, int> slot_index_;
  std::vector<void*> owned_;
  float *te_w0_ = nullptr, *te_b0_ = nullptr, *te_w2_ = nullptr, *te_b2_ = nullptr;
  float *emb_w_ = nullptr, *emb_b_ = nullptr;       // concatenated emb_layers [emb_total][te], [emb_total]
  float *out_gamma_ = nullptr, *out_beta_ = nullptr, *out_w_ = nullptr, *out_b_ = nullptr;
  bool finalized_ = false;
  int ctx_B_ = 0, ctx_L_ = 0; bool ctx_valid_ = false;
  int64_t ctx_cap_ = 0;      // K / V^T cache capacity in padded context rows (B * round_up(Lctx, 8))
  // timestep table: [n][emb_total] fp32 rows + scratch for one chunk of 8 timesteps (grow-only device buffer)
  float* emb_tab_ = nullptr; size_t emb_tab_floats_ = 0;
  int64_t* emb_tab_tdev_ = nullptr; size_t emb_tab_tcap_ = 0;
  std::vector<int64_t> emb_tab_t_;        // timesteps of the table rows (empty: no table)
  std::vector<int64_t> emb_tab_src_;      // host source of the last upload (kept alive for the asynchronous copy)
  int emb_hint_row_ = -1;
  void drop_timestep_table() { emb_tab_t_.clear(); emb_hint_row_ = -1; }
};

}  // namespace sdmi
