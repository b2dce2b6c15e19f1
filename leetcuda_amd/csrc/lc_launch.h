// lc_launch.h — pieces shared by the translation units of libleetcuda_amd.so (lc_abi.hip + the tu_*.hip files that
// hold the compile-heavy literal-AGPR kernels, built in parallel by leetcuda_amd/build.py).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/lc_abi.h"
#include "lc_common.h"

namespace lc {

// Launch status of the kernel just submitted.  hipGetLastError() also returns (and clears) a sticky error left by an
// EARLIER, unrelated call of this thread; launchers call launch_guard() first so that such an error is reported as
// what it is (LC_ERR_LAUNCH before our kernel is even submitted) instead of being blamed on the new launch.
inline int launch_guard() { return hipPeekAtLastError() == hipSuccess ? LC_OK : LC_ERR_LAUNCH; }
inline int check_launch() { return hipGetLastError() == hipSuccess ? LC_OK : LC_ERR_LAUNCH; }

// The reference re-issues cudaFuncSetAttribute on every call (hgemm_mma_stage.cu:2284); here the attribute is
// set once per (kernel, device) and remembered.
template <typename KernelT>
int set_dyn_lds(KernelT kernel, int bytes) {
  static std::mutex mu;
  static std::vector<std::pair<const void*, int>> done;
  const void* fn = reinterpret_cast<const void*>(kernel);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return LC_ERR_DEVICE;
  std::lock_guard<std::mutex> g(mu);
  for (const auto& d : done)
    if (d.first == fn && d.second == dev) return LC_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
    return LC_ERR_LAUNCH;
  done.emplace_back(fn, dev);
  return LC_OK;
}

// Compute units of the CURRENT device (the tail-split rule of launch_mfma256 and both persistent launchers size their grids with
// it): looked up once per device ordinal, not once per process — a later device with another CU count gets its own figure.
inline int device_cu_count() {
  static std::mutex mu;
  static int cache[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  std::lock_guard<std::mutex> g(mu);
  if (cache[dev] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cache[dev] = v;
  }
  return cache[dev];
}

// Scratch workspace for the two launch sequences that need partial results in HBM (split-KV attention: attn_w4u.hip WALK 3;
// split-K border strips of the GEMM: hgemm_mfma128.hip): one cached buffer per (device, stream), grown on demand, at most 16 per device
// (least recently used evicted).  Why not hipMallocAsync per call: measured (profiles/r5a_attn_split.log) the allocation + free pair
// costs about as much as the whole attention of a (1,8,1024,128) problem.  Safety: successive users on ONE stream are ordered by the
// stream; different streams never share a buffer; the lease holds THE DEVICE's mutex until the caller has enqueued its last kernel, so
// two host threads enqueueing on the same stream cannot interleave their sequences — and (round 6, advisor) threads that drive
// DIFFERENT GPUs no longer serialise on one process-wide lock.  Growth is geometric (a regrow frees the old buffer, which waits for
// the device: a sequence of slowly growing shapes regrows O(log) times, not once per shape), requests beyond kWorkspaceCapBytes are
// refused (the callers then run their workspace-free form), and lc_workspace_release() (C-ABI) gives everything back.  hipMalloc /
// hipFree are illegal while THIS stream is being captured — callers check stream_is_capturing first and take their workspace-free
// path — and "potentially unsafe" while ANOTHER thread captures in the global mode (torch's default): the allocation runs under a
// relaxed thread capture mode (hipThreadExchangeStreamCaptureMode), the idiom of torch's own caching allocator, so a first-use
// allocation here cannot invalidate somebody else's capture.  Nothing is freed at exit.
struct WorkspaceLease {
  std::unique_lock<std::mutex> lock;
  void* ptr = nullptr;
};
constexpr size_t kWorkspaceCapBytes = (size_t)1 << 30;   // per buffer; the split rules stay far below (partials <= 256 MiB)
struct WorkspaceEntry { hipStream_t st; void* p; size_t bytes; unsigned long long tick; };
struct WorkspacePool {
  std::mutex mu;
  std::vector<WorkspaceEntry> entries;
  unsigned long long tick = 0;
};
inline WorkspacePool& workspace_pool(int dev) {
  static WorkspacePool pools[64];
  return pools[(dev >= 0 && dev < 64) ? dev : 0];
}
struct RelaxedCaptureMode {   // hipMalloc / hipFree while another thread's global-mode capture is open
  hipStreamCaptureMode prev = hipStreamCaptureModeRelaxed;
  bool ok;
  RelaxedCaptureMode() { ok = hipThreadExchangeStreamCaptureMode(&prev) == hipSuccess; if (!ok) (void)hipGetLastError(); }
  ~RelaxedCaptureMode() { if (ok) (void)hipThreadExchangeStreamCaptureMode(&prev); }
};
// A launch sequence that already holds this thread's lease (the K-padding path of lc_hgemm_f16 keeps padded operands in the workspace while the
// inner launch is enqueued) marks the thread: nested requests get no workspace — their callers run their workspace-free forms — and take no lock.
inline bool& workspace_held_by_this_thread() {
  static thread_local bool held = false;
  return held;
}
inline WorkspaceLease stream_workspace(hipStream_t st, size_t bytes) {
  WorkspaceLease lease;
  if (workspace_held_by_this_thread()) return lease;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return lease;
  }
  WorkspacePool& pool = workspace_pool(dev);
  lease.lock = std::unique_lock<std::mutex>(pool.mu);
  if (bytes == 0 || bytes > kWorkspaceCapBytes) return lease;
  bytes = (bytes + ((size_t)1 << 22) - 1) & ~(((size_t)1 << 22) - 1);   // 4 MiB granules: a slightly larger next shape re-uses the buffer
  WorkspaceEntry* e = nullptr;
  for (auto& x : pool.entries)
    if (x.st == st) e = &x;
  if (e && e->bytes >= bytes) {
    e->tick = ++pool.tick;
    lease.ptr = e->p;
    return lease;
  }
  RelaxedCaptureMode relaxed;
  if (!e && pool.entries.size() >= 16) {   // evict the least recently used buffer (hipFree waits for the device: no kernel still reads it)
    size_t lru = 0;
    for (size_t i = 1; i < pool.entries.size(); ++i)
      if (pool.entries[i].tick < pool.entries[lru].tick) lru = i;
    (void)hipFree(pool.entries[lru].p);
    pool.entries.erase(pool.entries.begin() + lru);
  }
  if (e) {
    size_t grown = e->bytes + e->bytes / 2;   // geometric: at least 1.5 x the old size
    grown = (grown + ((size_t)1 << 22) - 1) & ~(((size_t)1 << 22) - 1);
    if (grown > bytes && grown <= kWorkspaceCapBytes) bytes = grown;
    (void)hipFree(e->p);   // (device-synchronising: the stream's earlier users are done with it)
    e->p = nullptr;
    e->bytes = 0;
  }
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess || !p) {
    (void)hipGetLastError();
    if (e) pool.entries.erase(pool.entries.begin() + (e - pool.entries.data()));
    return lease;
  }
  if (e) {
    e->p = p;
    e->bytes = bytes;
    e->tick = ++pool.tick;
  } else {
    pool.entries.push_back(WorkspaceEntry{st, p, bytes, ++pool.tick});
  }
  lease.ptr = p;
  return lease;
}
// Gives every cached workspace buffer of every device back (lc_workspace_release; waits for the devices).  Returns the bytes freed.
inline size_t workspace_release_all() {
  size_t freed = 0;
  int cur = 0;
  const bool have_cur = hipGetDevice(&cur) == hipSuccess;
  RelaxedCaptureMode relaxed;
  for (int dev = 0; dev < 64; ++dev) {
    WorkspacePool& pool = workspace_pool(dev);
    std::lock_guard<std::mutex> g(pool.mu);
    if (pool.entries.empty()) continue;
    if (hipSetDevice(dev) != hipSuccess) {
      (void)hipGetLastError();
      continue;
    }
    for (auto& x : pool.entries) {
      (void)hipFree(x.p);
      freed += x.bytes;
    }
    pool.entries.clear();
  }
  if (have_cur) (void)hipSetDevice(cur);
  return freed;
}
inline size_t workspace_cached_bytes() {
  size_t n = 0;
  for (int dev = 0; dev < 64; ++dev) {
    WorkspacePool& pool = workspace_pool(dev);
    std::lock_guard<std::mutex> g(pool.mu);
    for (auto& x : pool.entries) n += x.bytes;
  }
  return n;
}
inline bool stream_is_capturing(hipStream_t st) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) {
    (void)hipGetLastError();
    return true;   // cannot tell: behave as if it were
  }
  return cap != hipStreamCaptureStatusNone;
}

// tuning globals (defined in lc_abi.hip, lc_tune_set)
// Every knob is a std::atomic<int> (relaxed loads / stores through the implicit conversions): lc_tune_set from one host thread
// while another launches is a data race on a plain int; a launch reads each knob ONCE into a local and decides from that.
using tune_t = std::atomic<int>;
extern tune_t g_tune_attn_ablate, g_tune_w4_abl, g_tune_hgemm_stamps;
extern tune_t g_tune_hgemm_mid, g_tune_hgemm_mid_ns;   // one-round kernel: tile / ring depth (lc_tune_set "hgemm_mid", "hgemm_mid_ns")
extern tune_t g_tune_attn_d512;   // D = 256 / 512 attention kernel choice (lc_tune_set "attn_d512")
extern tune_t g_tune_attn_bigd_stagger;   // attn_bigd4: the KV walk of XCD x starts x eighths in: 0 = auto (with the round-robin map), 1 = off, 2 = on
extern tune_t g_tune_attn_bigd_map;     // query-block map of attn_bigd4 / attn_bigd6: 0 = auto (D = 1024 round-robin over the XCDs, D = 512 XCD-contiguous), 1 = contiguous, 2 = round-robin
extern tune_t g_tune_hgemm_persist;   // 1 (default) = hgemm_w4y_kernel as a persistent workgroup per CU when the tiles divide evenly (tu_w4.hip)
extern tune_t g_tune_hgemm_stagger;   // K-loop stagger of hgemm_w4y_kernel: 0 = auto (by XCD, step K / 64 / 8), 1 << 27 (exactly) = off, else cx | cm << 4 | cn << 8 | step << 12 | mask << 20 with mask < 128 (hgemm_w4y.hip)
// the kernel argument of the K-loop stagger for a K walk of kt tiles: auto (knob 0) = by XCD, the eight start tiles spread evenly
// over the K range (L2 sharing inside an XCD stays intact, fabric bytes unchanged: profiles/r3q_hgemm_stagger_ab.log).  The mask
// field is 7 bits (20 .. 26); bit 27 alone is the "off" value — lc_tune_set refuses it in combination with anything else, so a
// large mask can no longer switch the stagger off by accident (round-3 advisor finding).
constexpr int STAGGER_OFF = 1 << 27;
inline int stagger_arg(int kt) {
  const int knob = g_tune_hgemm_stagger;
  if (knob == STAGGER_OFF) return 0;
  if (knob != 0) return knob;
  const int step = kt / 8;
  return 1 | (step < 1 ? 1 : step > 255 ? 255 : step) << 12 | 7 << 20;
}
extern tune_t g_tune_w4y_sched;   // schedule of hgemm_w4y_kernel's generated loop body (all of them compute the same bits)

// launchers living in their own translation units
// tu_w4.hip: LC_HGEMM_MFMA256W4 / W4S / W4B / W4C (M, N % 256 == 0, K % 64 == 0 checked by the caller)
int w4_effective_variant(int variant, bool b_kn, int N, int K);   // W4C / W4X / W4Y -> W4B when 32-bit DMA offsets could overflow
// nblk > 0: launch only the first nblk blocks (hgemm_w4y_kernel only; the caller hands the remaining raster ids to the 128-tile kernel)
// the mid-size kernel (hgemm_mid.hip, tu_mid.hip): (64 tmw) x (64 tnw) tiles, ns ring slots
int launch_hgemm_mid(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, bool b_kn, int tmw, int tnw, int ns, int pw,
                     hipStream_t st, float* part = nullptr, int ks = 1);
// 128 x 128 tiles of the mid-size kernel that may reach beyond M / N (hgemm_mid_edge_kernel): strips of C as hgemm_edge_kernel's launcher defines them
int launch_hgemm_mid_edge(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, bool b_kn, int tmw, int tnw, int ns, int Mi, int Ni, hipStream_t st);
// ... split-K of a whole ragged problem (tmw 1 / 2 = 64 / 128 x 128 tiles): part holds launch_hgemm_mid_edge_sk_floats(M, N, tmw, ks) floats
size_t launch_hgemm_mid_edge_sk_floats(int M, int N, int tmw, int ks);
int launch_hgemm_mid_edge_sk(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, bool b_kn, int tmw, int ks, float* part, hipStream_t st);
int launch_hgemm_mid_rem(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, bool b_kn, int tmw, int ns, int tiles_m256,
                         int tiles_n256, int pw256, int rem_base, int rem_tiles, hipStream_t st);   // the 256-tile kernel's ragged last round as 128 x 128 quadrants
int launch_w4_family(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int variant, bool b_kn,
                     int tiles_m, int tiles_n, int panel_w, int nblk, hipStream_t st);
// tu_valu.hip: the vector-ALU ladder (hgemm_valu.hip), rung = LC_HGEMM_VALU_*
int launch_valu_rung(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int rung, hipStream_t st);
void valu_rung_tile(int rung, int* tm, int* tn, int* tk);
const char* valu_rung_kernel_name(int rung);
// tu_attn_w4u_{d128,d128t,d64,d64t}.hip: THE merged-phase attention kernel (attn_w4u.hip), one unit per (head dim, V layout): N % 256 == 0;
// walk 0 = one 256-row query block per workgroup, 1 = persistent workgroup per CU with a static walk, 2 = persistent with a dynamic
// per-XCD block queue (falls back to 0 when there are no more blocks than CUs), 3 = split-KV: nsplit workgroups per query block +
// a combine kernel, partials in the stream's cached workspace (falls back to 0 while the stream is being captured)
int launch_attn_w4u_d128(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, int walk, int nsplit, hipStream_t st);
int launch_attn_w4u_d128t(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, int walk, int nsplit, hipStream_t st);   // V as [B,H,D,N]
int launch_attn_w4u_d64(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, int walk, int nsplit, hipStream_t st);
int launch_attn_w4u_d64t(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, int walk, int nsplit, hipStream_t st);    // V as [B,H,D,N]
int diag_attn_slowpath_u_d128(unsigned* out4, int reset);   // slow-path counters of each unit's kernels (host copy; resets when asked)
int diag_attn_slowpath_u_d128t(unsigned* out4, int reset);
int diag_attn_slowpath_u_d64(unsigned* out4, int reset);
int diag_attn_slowpath_u_d64t(unsigned* out4, int reset);
// tu_attn_w4i.hip: the generated merged-phase kernel (attn_w4i.hip: a phase = one generated asm statement): D in {32, 64, 96, 128},
// N % 256 == 0, V as [B,H,N,D]; the only merged-phase kernel for D = 96 / 32
int launch_attn_w4i(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, int D, int sched, hipStream_t st);
int diag_attn_slowpath_g(unsigned* out4, int reset);   // + the slow-path counters of the w4i kernels
// tu_attn_big.hip: full-width large-head-dim kernel, D in {256, 512}, N % 128 == 0, V as [B,H,N,D]; fp16 or bf16
int launch_attn_bigd2(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, int D, bool bf16,
                      hipStream_t st);
// tu_attn_big4.hip: D = 1024 (attn_bigd4.hip: two waves share 32 query rows, each owns 512 columns; N % 64 == 0, V as [B,H,N,D], fp16) and
// attn_bigd2's V-transposed instantiation (D = 256, N % 128 == 0, V as [B,H,D,N], fp16)
int launch_attn_bigd4(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, int span8, hipStream_t st);   // span8: DMA spread (eighths of a phase; 0 = default)
int launch_attn_bigd2_vt(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, int D, hipStream_t st);
// tu_attn_big6.hip: D = 512 on v_mfma_f32_16x16x32 (attn_bigd6.hip), N % 128 == 0, V as [B,H,N,D]; fp16 or bf16
int launch_attn_bigd6(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, bool bf16, hipStream_t st);
// tu_attn_big7.hip: D = 256, 64 query rows per wave on v_mfma_f32_16x16x32 (attn_bigd7.hip), N % 256 == 0, V as [B,H,N,D]; fp16 or bf16
int launch_attn_bigd7(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, bool bf16, hipStream_t st);
int launch_attn_bigd7_vt(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, hipStream_t st);   // V as [B,H,D,N], fp16
// tu_fp8.hip: fp8 e4m3 GEMM, mx = 1 (MX, 4 waves) / 2 (MX, 8 waves) / 0 (plain K = 16)
int launch_gemm_fp8(const uint8_t* A, const uint8_t* B, half_t* C, int M, int N, int K, float alpha, int tiles_m,
                    int tiles_n, int panel_w, int mx, hipStream_t st);
// tu_fp8k.hip: the K = 128 MX form (gemm_fp8_w4k.hip): unit scales / real E8M0 block scales packed by launch_mx_pack_scales
bool gemm_fp8_w4k_fits(int K);
int launch_gemm_fp8_w4k(const uint8_t* A, const uint8_t* B, half_t* C, int M, int N, int K, float alpha, int tiles_m, int tiles_n,
                        int panel_w, hipStream_t st);
int launch_gemm_mxfp8(const uint8_t* A, const uint32_t* PA, const uint8_t* B, const uint32_t* PB, half_t* C, int M, int N, int K,
                      float alpha, int tiles_m, int tiles_n, int panel_w, hipStream_t st);
int launch_mx_pack_scales(const uint8_t* S, uint32_t* P, int rows, int K, hipStream_t st);

}  // namespace lc
