// Host-side plumbing shared by every entry point: thread-local error text, launch check, version.
#include <stdarg.h>
#include <stdlib.h>

#include "pd_common.h"

namespace pd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const Switches& switches() {
  static const Switches sw = [] {   // C++11: initialised once, thread-safe
    Switches v;
    auto num = [](const char* name) { const char* e = getenv(name); return e ? atoi(e) : -1; };
    v.no_rowpair = getenv("PD_NO_ROWPAIR") != nullptr;
    v.pp_rows_off = num("PD_PP_ROWS") == 0;
    v.pp_seg_off = num("PD_PP_SEG") == 0 || v.pp_rows_off;
    v.pp_chain_off = num("PD_PP_CHAIN") == 0;
    v.row_waves = num("PD_ROW_WAVES") > 0 ? num("PD_ROW_WAVES") : 0;
    v.uni_chunk = num("PD_UNI_CHUNK") > 0 ? num("PD_UNI_CHUNK") : 0;
    v.fwd_stream = num("PD_FWD_STREAM") != 0;
    return v;
  }();
  return sw;
}

static int current_device_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  return (dev >= 0 && dev < kMaxDevices) ? dev : kMaxDevices - 1;   // (ordinals beyond the table share its last slot's limit)
}

size_t device_lds_bytes() {
  static std::atomic<size_t> bytes[kMaxDevices];   // zero-initialised: 0 = not asked yet
  const int dev = current_device_slot();
  size_t have = bytes[dev].load(std::memory_order_relaxed);
  if (have) return have;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || n <= 0)
    n = 160 * 1024;  // no device to ask (host-side size queries on a box without a GPU): the target's value, gfx950
  (void)hipGetLastError();
  bytes[dev].store((size_t)n, std::memory_order_relaxed);   // (two threads racing here store the same value)
  return (size_t)n;
}

int grant_dynamic_lds(const void* kernel, size_t bytes, LdsGrant* grant, const char* what) {
  constexpr size_t kDefault = 64 * 1024;
  if (bytes <= kDefault) return PD_OK;
  const int dev = current_device_slot();
  if (bytes <= grant->granted[dev].load(std::memory_order_acquire)) return PD_OK;
  std::lock_guard<std::mutex> lock(grant->slow);
  if (bytes <= grant->granted[dev].load(std::memory_order_relaxed)) return PD_OK;   // somebody else raised it meanwhile
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    (void)hipGetLastError();
    set_error("%s: %zu bytes of LDS per workgroup are not available on this device", what, bytes);
    return PD_ERR_UNSUPPORTED;
  }
  grant->granted[dev].store(bytes, std::memory_order_release);   // only ever raised: a concurrent launch with a smaller need stays valid
  return PD_OK;
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return PD_OK;
  set_error("%s: %s", what, hipGetErrorString(e));
  return PD_ERR_LAUNCH;
}

}  // namespace pd

extern "C" int pd_experiments(void) {
#ifdef PD_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}
extern "C" int pd_build_flags(void) {
  int f = 0;
#ifdef PD_EXPERIMENTS
  f |= 1;
#endif
#ifdef PD_DIAGNOSTICS
  f |= 2;
#endif
  return f;
}
extern "C" int pd_version(void) { (void)pd::switches(); return 260; /* 0.2.6: pd_post_process (+ _workspace_floats), pd_pp_combine, pd_plane_levels_fwd/bwd, PD_PP_SEG (segment-form post-process warps), homography row products in torch.matmul's rounding order; 0.2.5: pd_build_flags (timing-ablation switches need -DPD_DIAGNOSTICS), pd_plane_sweep_bwd_tail, pd_sweep_bwd_tail_fuses, pd_sweep_bwd_plane_adds, pd_sweep_auto_row_eps, PD_BWD_PLANE_ZEROED, PD_IMPL_EXACT_ROWS, PD_ROW_EPS; 0.2.4: pd_uniform_fwd_pair / pd_uniform_bwd_pair (pd_sweep_view), pd_plade_tail_*, PD_PH_MEAN_ZEROED; 0.2.3: segment-stream forward (pd_plane_sweep_fwdstream.hip; PD_IMPL_ROWS1 selects the plane-group forward too), pd_source_hash; 0.2.2: gather backward for per-plane homographies (pd_debug_gather_flags), pd_uniform_gather_pair + PD_BWD_DEFER_GATHER, packed wide-row context; 0.2.1: row-stream backward, PD_IMPL_UNIFORM_DIRECT, pd_experiments, environment switches read once; 0.2.0: PD_HOMO_UNIFORM, PD_BWD_ACCUMULATE, pd_homography_matrices_*, pd_masked_photometric_*, pd_crop_grid; 0.1.1: fused mean of ph_map */ }
extern "C" const char* pd_last_error(void) { return pd::g_err; }

// What this binary was compiled from.  The marker string is also what __graft_entry__.build() looks for in the file's bytes
// to decide whether the library on disk matches the source tree (no side file, no dlopen of a stale library).
#ifndef PD_SRC_HASH
#define PD_SRC_HASH "unknown"
#endif
static const char kSourceHashMarker[] = "PD_SRC_HASH=" PD_SRC_HASH;
extern "C" const char* pd_source_hash(void) { return kSourceHashMarker + 12; }

// Diagnostics: fill the LDS of (as good as) every CU with NaNs, so that a kernel that reads shared memory it never wrote
// shows up as NaN results instead of depending on what the previous tenant left there (tests/test_gpu_parity.py).
namespace pd {
__global__ __launch_bounds__(256) void poison_lds_kernel(int* __restrict__ sink) {
  __shared__ float junk[16 * 1024];   // 64 KB: two workgroups fill a CU's 160 KB almost completely
  for (int i = threadIdx.x; i < 16 * 1024; i += 256) junk[i] = __int_as_float(0x7fc00000 | i);
  __syncthreads();
  if (sink && __float_as_int(junk[(threadIdx.x * 61) & (16 * 1024 - 1)]) == 1) *sink = 1;   // keep the stores alive
}
}  // namespace pd

namespace pd {
__global__ __launch_bounds__(256) void count_lds_nans_kernel(int* __restrict__ count) {
  __shared__ float junk[8 * 1024];   // 32 KB, read WITHOUT having been written
  int n = 0;
  if (count == nullptr)   // never true: it only keeps the compiler from treating the reads below as reads of `undef`
    for (int i = threadIdx.x; i < 8 * 1024; i += 256) junk[i] = 0.0f;
  __syncthreads();
  for (int i = threadIdx.x; i < 8 * 1024; i += 256) {
    const float v = junk[i];
    n += (v != v) ? 1 : 0;
  }
  if (n) atomicAdd(count, n);
}
}  // namespace pd

/* test of the test: how many NaNs 2048 workgroups find in 32 KB of LDS they never wrote (*d_count += that) */
extern "C" int pd_debug_count_lds_nans(int* d_count, pd_stream_t stream) {
  pd::count_lds_nans_kernel<<<256 * 8, 256, 0, (hipStream_t)stream>>>(d_count);
  return pd::check_launch("count_lds_nans_kernel");
}

extern "C" int pd_debug_poison_lds(pd_stream_t stream) {
  pd::poison_lds_kernel<<<256 * 8, 256, 0, (hipStream_t)stream>>>(nullptr);
  return pd::check_launch("poison_lds_kernel");
}
