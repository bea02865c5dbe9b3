// Launch tape (round 6): the executor's host cost per UNet call was 11 us per launch, of which ~3.5 us is the HIP launch itself.
// The rest -- a dry pass that sizes the arenas and derives the fusion plan, tuning-table lookups, descriptor fills, environment reads -- produces
// the SAME launch list every time a (B, H, W, Lctx) shape comes back with the same workspace, weights and knobs.  So the list is recorded once
// (kernel pointer, grid, block, the filled parameter structs with their arena pointers, the memsets) and replayed as a flat loop of
// hipLaunchKernel; only the pointers that belong to the CALLER change between calls (x, eps_out, timesteps, context, the row of the timestep
// table): their positions inside the recorded parameter bytes are found once, by scanning for 8-byte words that point into those ranges,
// and patched before a replay.
//
// Every kernel launch of the library goes through SDMI_LAUNCH (= hipLaunchKernelGGL + the recording hook), every stream memset of a forward
// through sdmi::memset_async.  Recording is thread-local and only active inside UNet::run.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <vector>

namespace sdmi {

struct Tape {
  struct Op {
    int kind = 0;                       // 0 kernel launch, 1 memset
    const void* fn = nullptr; dim3 grid, block; unsigned shmem = 0; uint32_t first_arg = 0, nargs = 0;
    void* ptr = nullptr; int value = 0; size_t bytes = 0;
  };
  struct Reloc { uint32_t off; int which; };          // an 8-byte word of `blob` that points into caller range `which`
  enum { R_X = 0, R_OUT, R_T, R_CTX, R_EMB, R_COUNT };
  std::vector<Op> ops;
  std::vector<unsigned char> blob;                    // the parameter bytes of every launch, each argument at its natural alignment
  std::vector<uint32_t> arg_off;                      // per argument: offset into blob
  std::vector<uint32_t> arg_size;
  std::vector<Reloc> relocs;
  uintptr_t base[R_COUNT] = {0, 0, 0, 0, 0};          // the caller ranges the blob currently points into
  size_t span[R_COUNT] = {0, 0, 0, 0, 0};
  int64_t bytes_needed = 0;
  bool sets_ctx_valid = false;

  void clear() { ops.clear(); blob.clear(); arg_off.clear(); arg_size.clear(); relocs.clear(); }
  void push_arg(const void* src, size_t size, size_t align) {
    size_t off = (blob.size() + align - 1) / align * align;
    blob.resize(off + size);
    memcpy(blob.data() + off, src, size);
    arg_off.push_back((uint32_t)off); arg_size.push_back((uint32_t)size);
  }
  // find the words that point into the caller ranges (once, right after recording)
  void find_relocs() {
    relocs.clear();
    for (size_t a = 0; a < arg_off.size(); ++a) {
      if (arg_size[a] < 8) continue;
      const uint32_t o0 = (arg_off[a] + 7u) & ~7u;
      for (uint32_t o = o0; o + 8 <= arg_off[a] + arg_size[a]; o += 8) {
        uint64_t v;
        memcpy(&v, blob.data() + o, 8);
        for (int r = 0; r < R_COUNT; ++r)
          if (span[r] && v >= base[r] && v < base[r] + span[r]) { relocs.push_back({o, r}); break; }
      }
    }
  }
  // point the blob at new caller ranges
  void retarget(const uintptr_t (&nb)[R_COUNT]) {
    bool same = true;
    for (int r = 0; r < R_COUNT; ++r) same = same && (nb[r] == base[r] || !span[r]);
    if (same) return;
    for (const Reloc& rl : relocs) {
      uint64_t v;
      memcpy(&v, blob.data() + rl.off, 8);
      v = v - base[rl.which] + nb[rl.which];
      memcpy(blob.data() + rl.off, &v, 8);
    }
    for (int r = 0; r < R_COUNT; ++r) if (span[r]) base[r] = nb[r];
  }
  int replay(hipStream_t s) {
    void* args[64];
    for (const Op& op : ops) {
      hipError_t e;
      if (op.kind == 1) {
        e = hipMemsetAsync(op.ptr, op.value, op.bytes, s);
      } else {
        for (uint32_t i = 0; i < op.nargs; ++i) args[i] = blob.data() + arg_off[op.first_arg + i];
        e = hipLaunchKernel(op.fn, op.grid, op.block, args, op.shmem, s);
      }
      if (e != hipSuccess) return (int)e;
    }
    return 0;
  }
};

extern thread_local Tape* g_tape_rec;        // non-null while UNet::run records

template <typename... KArgs>
inline void tape_push_launch(Tape* t, const void* fn, dim3 g, dim3 b, size_t shm, const KArgs&... a) {
  Tape::Op op;
  op.kind = 0; op.fn = fn; op.grid = g; op.block = b; op.shmem = (unsigned)shm;
  op.first_arg = (uint32_t)t->arg_off.size(); op.nargs = (uint32_t)sizeof...(KArgs);
  (t->push_arg(&a, sizeof(KArgs), alignof(KArgs) < 8 ? 8 : alignof(KArgs)), ...);
  t->ops.push_back(op);
}

#if defined(__HIP__) || defined(__HIPCC__)
// hipLaunchKernelGGL + the recording hook.  The arguments are converted to the kernel's parameter types first, so the recorded bytes are
// exactly what the launch passes.
template <typename... KArgs, typename... Args>
inline void launch_rec(void (*kernel)(KArgs...), dim3 g, dim3 b, size_t shm, hipStream_t s, Args&&... a) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
  static_assert(sizeof...(KArgs) <= 64, "Tape::replay's argument array");
  if (g_tape_rec) {
    [&](const KArgs&... conv) { tape_push_launch<KArgs...>(g_tape_rec, (const void*)kernel, g, b, shm, conv...); }(static_cast<KArgs>(a)...);
  }
  hipLaunchKernelGGL(kernel, g, b, shm, s, static_cast<KArgs>(a)...);
}

#endif

inline hipError_t memset_async(void* ptr, int value, size_t bytes, hipStream_t s) {
  if (g_tape_rec) {
    Tape::Op op;
    op.kind = 1; op.ptr = ptr; op.value = value; op.bytes = bytes;
    g_tape_rec->ops.push_back(op);
  }
  return hipMemsetAsync(ptr, value, bytes, s);
}

}  // namespace sdmi

#define SDMI_LAUNCH(kernel, grid, block, shmem, stream, ...) ::sdmi::launch_rec(kernel, grid, block, shmem, stream, __VA_ARGS__)
