// engine.hip - gfx950 kernels + executor + C ABI of the contraction engine.
//
// Replaces the stabilised pairwise loop of the reference
// (contractn/einsum.py:326-393) and stabilize() (einsum.py:89-107):
//
//   * every pairwise step C[b,m,n] = sum_k A[b,m,k] * B[b,k,n] runs as ONE kernel whose
//     loads/stores go through gather-offset tables (transpose/reshape fused into
//     the load; reference einsum.py:371-377 does tensordot + transpose copies);
//   * a label kept while shared (copy tensor / hyperedge, reference ctn.py:154-165)
//     is a batch index of that kernel - no identity tensor is ever materialised;
//   * stabilize() is fused: the epilogue divides by the producers' rescale
//     factors (applied lazily: (A/sA)(B/sB) == (A B)/sA/sB), stores the
//     un-normalised tile and emits one partial sum of |C| per workgroup.  The
//     consumer (or the finishing pass) reduces <= 64 partials with one wave in a
//     fixed order, so results are bit-reproducible run to run (no float atomics).
//
// Written for CDNA4 only: 64-lane waves, v_mfma_f32_32x32x2_f32, 160 KiB LDS/CU.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "plan.h"

#include "kernel_args.h"
#include "kernels_mfma.h"
#include "kernels_mfma_g.h"
#include "kernels_mfma_g64.h"
#include "kernels_mfma_h.h"
#include "kernels_mfma_ares.h"
#include "kernels_zip.h"
#include "kernels_zip64.h"
#include "kernels_zipl.h"
#include "kernels_sweep.h"
#include "kernels_mfma_lat.h"
#include "kernels_stream.h"

namespace ctn {

// ---------------------------------------------------------------------------
// executor
// ---------------------------------------------------------------------------
thread_local std::string g_err;

// Development switches: every environment variable the library looks at, read in ONE place, once per executor
// (ctn_exec_create) - never on the launch path.  None is needed in production ("Development switches").
struct DevSwitches {
  int mfma_g = 1;        // CTN_MFMA_G: 0 never use the large-tile LDS-DMA kernels, 1 when a launch fills the chip, 2 whenever eligible (tests)
  int graph = 1;         // CTN_GRAPH=0: every enqueue issues its launches one by one
  int mfma_bk = 0;       // CTN_MFMA_BK=16|32: force the k-tile depth of k_mfma_f32
  int group = 1;         // CTN_GROUP=0: never batch independent leaf steps into one launch (k_stream_group)
  int halve = 1;         // CTN_HALVE_TILES=0: never halve the tiles of an under-filled register-staged launch
  int lat_wg_per_cu_x2 = 2;  // CTN_LAT_WG_X2: twice the workgroups per CU up to which the one-launch latency form is taken
  int lat_max_t = 64;    // CTN_LAT_MAX_T=32: no 64 x 64 form of the one-launch latency kernel
  int splitk_fill_long = 2;  // CTN_SPLITK_FILL_LONG: K >= 1024 steps are split until 64 x 64 tiles give this many workgroups per CU
  int lat64_min_k = 512; // CTN_LAT64_MIN_K: least K for the 64 x 64 one-launch latency form when an operand is k-contiguous
  int splitk = -1;       // CTN_SPLITK: 0 disables the latency mode, 1 forces it for every eligible step (tests)
  int splitk_max = 0;    // CTN_SPLITK_MAX: tile-count threshold of the latency mode
  int lat = -1;          // CTN_LAT: 0 never use the one-launch latency form (k_mfma_f32_lat), 1 whenever the shape allows (tests)
  int hform = -1;        // CTN_H: 0 never use the one-tile-per-CU form (k_mfma_f32_h), 1 whenever the shape allows (tests)
  int sweep = -1;        // CTN_SWEEP: 0 never walk a chain of epilogue-summed steps in one launch (k_sweep_f32), 1 whenever one matches (tests)
  int dot_tr = 1;        // CTN_DOT_TR=0: full dots against a transposed tensor stay on k_dot_split's 4-byte gathers
  int zip = -1;          // CTN_ZIP: 0 never fuse a zipper's two GEMM steps into one launch (k_zip_f32), 1 whenever the pair matches
                         // (tests), 2 likewise with 64 values of u per workgroup (k_zip64_f32)
  int zipl = -1;         // CTN_ZIPL: 0 never run a zipper pair as one latency-form launch (k_zip_lat), 1 whenever the pair matches (tests)
  int zipl_max_r = 8;    // CTN_ZIPL_MAX_R: most networks in flight for which k_zip_lat is taken by default (100-site D = 256
                         // network, ms per pass, k_zip_lat / per-step launches: R = 1 1.40 / 2.05, 2: 1.58 / 3.3, 4: 2.1 / 3.5,
                         // 8: 4.1 / 4.4, 16: 8.5 / 7.0)
  int g_big_min_k = 1024;  // CTN_G_BIG_MIN_K: least K from which full long-K steps of any width take the 256 x 256 tiles
                           // (1536 until round 4; K = 1024, N = 1024: CP r = n = 1024 16.4 -> 15.9 ms, Tucker 15.8 -> 15.7 per mode product)
  int ares = -1;         // CTN_ARES: 0 never the resident-left-operand kernel (k_mfma_f32_ares), 1 whenever the step's shape allows (tests)
  int ares_ntw = 0;      // CTN_ARES_NTW: column tiles per workgroup of that kernel (tests)
  int g_big = 1;         // CTN_G_BIG=0: never the 256 x 256 tiles (experiments)
  int g_splitk = 1;      // CTN_G_SPLITK=0: no K split over workgroups on the large-tile kernel
  int zipl_mp = 0;       // CTN_ZIPL_MP=32|64: force the part of m1 a k_zip_lat workgroup owns (tests)
  bool g_no_asm = false; // CTN_G_NO_ASM: C++ inner loop instead of the hand-scheduled blocks
  const char* stamps = nullptr;  // CTN_DEBUG_STAMPS=<file> (make STAMPS=1 builds): dump in-kernel cycle stamps
  int stamp_step = -1;   // CTN_DEBUG_STAMP_STEP=<s>: stamp only this step
};
static DevSwitches read_dev_switches() {
  DevSwitches d;
  auto num = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
  d.mfma_g = num("CTN_MFMA_G", 1);
  d.graph = num("CTN_GRAPH", 1);
  const int bk = num("CTN_MFMA_BK", 0);
  d.mfma_bk = (bk == 16 || bk == 32) ? bk : 0;
  d.splitk = num("CTN_SPLITK", -1);
  d.splitk_max = num("CTN_SPLITK_MAX", 0);
  d.lat = num("CTN_LAT", -1);
  d.hform = num("CTN_H", -1);
  d.zip = num("CTN_ZIP", -1);
  d.zipl = num("CTN_ZIPL", -1);
  d.zipl_max_r = num("CTN_ZIPL_MAX_R", 8);
  d.zipl_mp = num("CTN_ZIPL_MP", 0);
  d.g_splitk = num("CTN_G_SPLITK", 1);
  d.g_big = num("CTN_G_BIG", 1);
  d.ares = num("CTN_ARES", -1);
  d.ares_ntw = num("CTN_ARES_NTW", 0);
  d.g_big_min_k = num("CTN_G_BIG_MIN_K", 1024);
  d.dot_tr = num("CTN_DOT_TR", 1);
  d.sweep = num("CTN_SWEEP", -1);
  d.lat64_min_k = num("CTN_LAT64_MIN_K", 512);
  d.splitk_fill_long = num("CTN_SPLITK_FILL_LONG", 2);
  d.lat_max_t = num("CTN_LAT_MAX_T", 64);
  d.lat_wg_per_cu_x2 = num("CTN_LAT_WG_X2", 2);
  d.halve = num("CTN_HALVE_TILES", 1);
  d.group = num("CTN_GROUP", 1);
  d.g_no_asm = getenv("CTN_G_NO_ASM") != nullptr;
  d.stamps = getenv("CTN_DEBUG_STAMPS");
  d.stamp_step = num("CTN_DEBUG_STAMP_STEP", -1);
  return d;
}

#define HIPCHECK(expr)                                                                   \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                         \
      return CTN_HIP_ERROR;                                                              \
    }                                                                                    \
  } while (0)

// Entry points select the executor's device and give the caller's current device back on return (a caller
// that works on another GPU - torch.cuda.current_device() - must not find it changed behind its back).
struct DeviceGuard {
  int prev = -1;
  hipError_t err;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
    err = prev == dev ? hipSuccess : hipSetDevice(dev);
    if (prev == dev) prev = -1;   // nothing to restore
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct Exec {
  const Plan* plan = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int R = 1;
  int n_tensors = 0;
  int n_cu = 256;
  DevSwitches sw;        // environment switches as they were when the executor was created
  int mfma_g = 1;        // sw.mfma_g (see exec_launch_steps)
  // the launch sequence as a hipGraph (CTN_GRAPH, see exec_launch_all): captured on the second enqueue,
  // replayed afterwards; tensors are reached through the device pointer table, so new operands need no update
  int use_graph = 1;
  bool graph_warm = false;          // one eager enqueue has run (code objects loaded, tiles recorded)
  bool graph_aligned = true;        // outs_aligned16 the graph was captured under
  hipGraphExec_t graph_exec = nullptr;
  // zipper pairs (kernels_zip.h): zip[s2] describes the fused launch of steps (s2 - 1, s2); zip_skip[s1] = the first
  // step of such a pair is never launched (its result only exists in the fused kernel's registers)
  struct ZipDesc { bool on = false; int64_t ldE = 0, ldXq = 0, ldXk = 0, ldYq = 0, ldYm = 0, ldC = 0; int Q = 0, U = 0, K1 = 0;
                   int zu = 128; };      // zu: values of u per workgroup (128: k_zip_f32, 64: k_zip64_f32)
  std::vector<ZipDesc> zip;
  std::vector<char> zip_skip;
  // the same pairs in their latency form (kernels_zipl.h): zl[s2] = the fused launch of steps (s2 - 1, s2), whose result
  // leaves as slabs; from_prev = its E is the slabs of the pair before, feeds_next = the next pair adds its slabs up (else
  // k_zip_slab_sum does); zl_skip[s1] like zip_skip
  struct ZipLat { bool on = false, from_prev = false, feeds_next = false; int buf = 0, mp = 32; ZipDesc d; };
  std::vector<ZipLat> zl;
  std::vector<char> zl_skip;
  float* d_zl_slab[2] = {nullptr, nullptr};   // [R][S][U][ZM] each, ping-pong along the chain
  // a sweep (kernels_sweep.h): a run of epilogue-summed GEMM steps, each on the result of the one before, walked by ONE
  // launch at the position of its last member; sweep_role[s] = 1 a member that is never launched, 2 the last member
  struct SweepDesc {
    bool on = false;
    std::vector<int> steps;          // the members' step indices, in chain order
    int64_t ldIn = 0, ldOut = 0, ldWl = 0, ldWp = 0, ldX = 0;
    int J = 0, M = 0, D = 0, Pd = 0;     // row blocks, rows, bond and physical dimension
  };
  SweepDesc sweep;
  std::vector<char> sweep_role;
  int32_t* d_sweep_ids = nullptr;      // [S][2] tensor ids (W_s, x_s)
  int64_t* d_sweep_off = nullptr;      // [S] step_off of the members
  int32_t* d_sweep_slots = nullptr;    // [S] step_partials of the members
  double* d_sweep_a = nullptr;         // [R][S][J]
  float* d_sweep_s = nullptr;          // [R][S][J]
  double* d_sweep_z = nullptr;         // [R][S]
  double* d_sweep_la = nullptr;        // [R][S][J] logs of d_sweep_a / d_sweep_s
  double* d_sweep_ls = nullptr;
  // steps with a small resident left operand against a very wide right one (kernels_mfma_ares.h): column tiles per workgroup
  // (bit 16: the left operand takes 16-byte loads), 0 = not taken
  std::vector<int> ares_ntw;
  // full dots of a tensor with a transposed one (k_dot_tr), found on the plan's tables when the executor is created
  struct DotTr { bool on = false; int Ka = 0, Kb = 0, x_is_a = 1; int64_t ldY = 0; };
  std::vector<DotTr> dot_tr;
  std::vector<int32_t> launched_tile;  // per step: (tile rows << 16 | tile columns) of the last enqueue's MFMA kernel, else 0
  char* d_ws = nullptr;
  int32_t* d_tables = nullptr;
  int64_t* d_tables64 = nullptr;    // batch / outer-group offsets (Plan::tables64)
  void** d_ptrs = nullptr;
  std::vector<void*> h_ptrs;
  bool ptrs_valid = false;
  double* d_partials = nullptr;
  double* d_scratch = nullptr;
  void* d_slab = nullptr;           // split-K partial tiles (latency mode), sized at creation
  void* d_slab2 = nullptr;          // 1/16 of it: where more than 16 slabs are folded 16 to 1 (k_splitk_fold)
  std::vector<int> step_partials;   // abs-sum partials per replica of every step (plan value unless split-K / latency form)
  std::vector<int64_t> step_off;    // d_partials: step s, replica r at (step_off[s] * R + r * step_partials[s]) doubles
  int64_t part_slots = 0;           // sum of the steps' slots
  int64_t* d_stepOff = nullptr;
  int32_t* d_stepSlots = nullptr;
  double* d_log = nullptr;
  double* d_resc = nullptr;
  double* d_logs = nullptr;
  ChainStep* d_chain = nullptr;
  unsigned long long* d_dbg = nullptr;  // CTN_DEBUG_STAMPS=<file>: stamps of the LAST MFMA launch
  size_t dbg_tiles = 0;
  char* h_pack = nullptr;       // pinned host bounce buffer for many-small-operand staging
  size_t h_pack_bytes = 0;
  bool outs_aligned16 = true;
  void* d_ones = nullptr;
  int32_t* d_stepP = nullptr;
  double* d_stepNumel = nullptr;
  char* d_stage_in = nullptr;
  char* d_stage_out = nullptr;
  // Rescale mode.  false = lazy (default): tile kernels multiply by 1 / (sA sB) in the epilogue, the
  // un-normalised tensor is what is stored.  true = eager: every intermediate is normalised in place right
  // after its step (k_renorm) and consumed with scale 1 - the reference's own order of operations.  An executor
  // switches to eager by itself, and repeats the contraction, when the scale registers of a finished run show
  // that a lazy product left the dtype's range (exec_scales_suspect); it then stays eager.
  bool eager_rescale = false;
  bool eager_forced = false;        // ctn_exec_set_rescale_mode(1): every run eager; otherwise lazy is tried first
  int eager_reruns = 0;
  int eager_streak = 0;             // consecutive runs that had to be repeated eagerly (see exec_new_run)
  std::vector<double> h_resc;       // host copy of the last run's per-step rescale factors [R][n_steps]
  // Grouped leaf steps: runs of consecutive streaming steps whose operands are all network inputs (independent of
  // each other and of everything before them) go out as ONE launch of k_stream_group; their arguments never
  // change, so they are built on the first enqueue and kept in device memory.
  struct LeafGroup { int len = 0, vw = 1, u = 1, blocks = 1, off = 0; bool ready = false; };
  std::vector<LeafGroup> groups;    // per step: len >= 2 at the head of a group, else 0
  StepArgs* h_group_args = nullptr; // pinned: the one-time upload may fall inside a stream capture
  StepArgs* d_group_args = nullptr;
  bool defer_finish = false;        // ctn_exec_set_finish_mode(1): the final division waits for ctn_exec_finish
  double* d_mult = nullptr;         // [R] factors of ctn_exec_finish
  char* d_merge = nullptr;          // ctn_exec_merge_scales: liveness flags + new registers, grown on demand
  size_t merge_bytes = 0;
  int timing_slots = 0;             // 0 = timing off
  int timing_runs = 0;              // enqueues recorded since timing was enabled
  std::vector<hipEvent_t> events;   // [slot][step][2]

  ~Exec() {
    DeviceGuard dg(device);
    for (void* p : {(void*)d_ws, (void*)d_tables, (void*)d_tables64, (void*)d_ptrs, (void*)d_partials, (void*)d_scratch, (void*)d_slab, (void*)d_slab2,
                    (void*)d_log, (void*)d_resc, (void*)d_logs, (void*)d_chain, d_ones, (void*)d_stepP, (void*)d_stepNumel,
                    (void*)d_stepOff, (void*)d_stepSlots,
                    (void*)d_stage_in, (void*)d_stage_out, (void*)d_group_args, (void*)d_sweep_ids, (void*)d_sweep_off,
                    (void*)d_sweep_slots, (void*)d_sweep_a, (void*)d_sweep_s, (void*)d_sweep_z, (void*)d_sweep_la, (void*)d_sweep_ls,
                    (void*)d_merge, (void*)d_zl_slab[0], (void*)d_zl_slab[1], (void*)d_mult})
      if (p) (void)hipFree(p);
    if (h_pack) (void)hipHostFree(h_pack);
    if (h_group_args) (void)hipHostFree(h_group_args);
    for (auto ev : events) (void)hipEventDestroy(ev);
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    if (own_stream && stream) (void)hipStreamDestroy(stream);
  }
};

template <int MA, int BK, int TN, int TM>
static void launch_mfma_b(int mb, dim3 grid, hipStream_t st, const StepArgs& a) {
  switch (mb) {
    case 1: hipLaunchKernelGGL((k_mfma_f32<MA, 1, BK, TN, TM>), grid, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((k_mfma_f32<MA, 2, BK, TN, TM>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((k_mfma_f32<MA, 0, BK, TN, TM>), grid, dim3(256), 0, st, a); break;
  }
}

template <int BK, int TN, int TM>
static void launch_mfma_a(int ma, int mb, dim3 grid, hipStream_t st, const StepArgs& a) {
  switch (ma) {
    case 1: launch_mfma_b<1, BK, TN, TM>(mb, grid, st, a); break;
    case 2: launch_mfma_b<2, BK, TN, TM>(mb, grid, st, a); break;
    case 3: launch_mfma_b<3, BK, TN, TM>(mb, grid, st, a); break;   // A = X (.) Y formed while staging (fused step)
    case 4: launch_mfma_b<4, BK, TN, TM>(mb, grid, st, a); break;   // ... with 16-byte accesses along the rows
    case 5: launch_mfma_b<5, BK, TN, TM>(mb, grid, st, a); break;   // ... along k
    default: launch_mfma_b<0, BK, TN, TM>(mb, grid, st, a); break;
  }
}

// epilogue-summed steps (planner pattern C): BK = 16, a plain A operand
template <int MA, int TN, int TM>
static void launch_mfma_epw_b(int mb, dim3 grid, hipStream_t st, const StepArgs& a) {
  switch (mb) {
    case 1: hipLaunchKernelGGL((k_mfma_f32<MA, 1, 16, TN, TM, true>), grid, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((k_mfma_f32<MA, 2, 16, TN, TM, true>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((k_mfma_f32<MA, 0, 16, TN, TM, true>), grid, dim3(256), 0, st, a); break;
  }
}
template <int TN, int TM>
static void launch_mfma_epw(int ma, int mb, dim3 grid, hipStream_t st, const StepArgs& a) {
  switch (ma) {
    case 1: launch_mfma_epw_b<1, TN, TM>(mb, grid, st, a); break;
    case 2: launch_mfma_epw_b<2, TN, TM>(mb, grid, st, a); break;
    default: launch_mfma_epw_b<0, TN, TM>(mb, grid, st, a); break;
  }
}

// k-tile depth.  BK = 16: 32 KiB of LDS + 167 registers => 3 workgroups (12 waves) per CU, which
// hides the per-tile prologue/epilogue best when K is short (MPS shapes: 108-118 TFLOP/s vs
// 103-117 with BK = 32); BK = 32 halves the barriers per flop and wins on long-K GEMMs
// (4096^3: 130 vs 115 TFLOP/s).  CTN_MFMA_BK=16|32 forces one of them (development knob).
static int mfma_bk(int K, const DevSwitches& sw) {
  if (sw.mfma_bk) return sw.mfma_bk;
  return K >= 2048 ? 32 : 16;
}

// tile_m = 64: skinny rows (M <= 64 against a huge N) - always BK = 16 (these steps have a short K)
static void launch_mfma(int ma, int mb, int tile_m, int tile_n, dim3 grid, hipStream_t st, const StepArgs& a, const DevSwitches& sw) {
  if (a.epw) {
    if (tile_m == 64) {
      if (tile_n == 64) launch_mfma_epw<64, 64>(ma, mb, grid, st, a);
      else launch_mfma_epw<128, 64>(ma, mb, grid, st, a);
    } else {
      if (tile_n == 64) launch_mfma_epw<64, 128>(ma, mb, grid, st, a);
      else launch_mfma_epw<128, 128>(ma, mb, grid, st, a);
    }
    return;
  }
  const bool bk16 = mfma_bk(a.K, sw) == 16;
  if (tile_m == 64) {
    if (tile_n == 64) launch_mfma_a<16, 64, 64>(ma, mb, grid, st, a);
    else launch_mfma_a<16, 128, 64>(ma, mb, grid, st, a);
  } else if (tile_n == 64) {
    if (bk16) launch_mfma_a<16, 64, 128>(ma, mb, grid, st, a);
    else launch_mfma_a<32, 64, 128>(ma, mb, grid, st, a);
  } else {
    if (bk16) launch_mfma_a<16, 128, 128>(ma, mb, grid, st, a);
    else launch_mfma_a<32, 128, 128>(ma, mb, grid, st, a);
  }
}

// Latency mode (see k_mfma_f32_sk): chosen when the 128-wide tiles of a step cannot occupy half the
// chip.  Returns the number of K splits (0 = use the throughput kernel).  CTN_SPLITK=0 disables,
// CTN_SPLITK=1 forces it for every eligible step (tests).
static bool h_forced(const Step& st, int dtype, const DevSwitches& sw);
static int splitk_splits(const Step& st, int R, int n_cu, int dtype, const DevSwitches& sw) {
  if (h_forced(st, dtype, sw)) return 0;
  const int mode = sw.splitk;
  const bool mfma = (dtype == CTN_F32 && st.kernel == CTN_KERNEL_MFMA_F32) || (dtype == CTN_F64 && st.kernel == CTN_KERNEL_MFMA_F64);
  if (mode == 0 || !mfma || st.collapse || st.K < 128 || st.modeA >= 3 || st.epw) return 0;
  const int max_tiles = sw.splitk_max;
  const int64_t limit = max_tiles > 0 ? max_tiles : n_cu / 2;
  if (mode != 1 && (int64_t)st.blocks * R > limit) return 0;
  const int64_t tiles64 = st.Bt * ((st.M + 63) / 64) * ((st.N + 63) / 64) * R;
  // 64 x 64 tiles that already give every CU a workgroup run un-split on the register-staged kernel: two K slabs plus
  // the reduce pass lose to it (7 replicas of 256 x 1024 x 256: 5.3 ms per 100-site network against 4.35 at 8) - a long
  // K (>= 1024) is still worth two slabs up to two workgroups per CU (16 x (256 x 256 x 1024): 7.02 -> 6.20 ms)
  if (tiles64 >= (int64_t)n_cu * (st.K >= 1024 ? sw.splitk_fill_long : 1) && mode != 1) return 0;
  int64_t S = (2 * (int64_t)n_cu + tiles64 - 1) / tiles64;   // aim at ~2 small workgroups per CU
  S = std::max<int64_t>(1, std::min<int64_t>(S, st.K / 64));
  // one split is no split: the register-staged kernel on halved tiles does the same work without slab and reduce pass
  // (8 replicas of a 256 x 1024 x 256 step; CTN_SPLITK=1 keeps it for the tests)
  if (S <= 1 && mode != 1) return 0;
  return (int)S;
}

// One-launch latency form (k_mfma_f32_lat): K split over the eight waves of a workgroup instead of over
// workgroups, no slabs and no reduce launch.  Returns the tile edge T (16, 32 or 64), 0 = not taken.
// Same launch-size condition as split-K (the step's 128-wide tiles cannot occupy half the chip); the largest tile
// that still gives every CU a workgroup (with one network in flight a 256 x 256 x 1024 step runs as 256 tiles of
// 16 x 16: on 64 tiles of 32 x 32 three quarters of the matrix pipes idle and the step takes 14 us instead of ~6);
// at most kLatMaxTiles tiles per replica (one abs-sum partial each), both k-offset tables in LDS.
constexpr int kLatMaxTiles = kMaxPartials;
static int lat_form(const Step& st, int R, int n_cu, int dtype, const DevSwitches& sw) {
  if (h_forced(st, dtype, sw)) return 0;           // tests of the one-tile-per-CU form (CTN_H=1)
  const bool f64 = dtype == CTN_F64 && st.kernel == CTN_KERNEL_MFMA_F64;
  if (sw.lat == 0 || !(f64 || (dtype == CTN_F32 && st.kernel == CTN_KERNEL_MFMA_F32)) || st.rhs < 0 || st.modeA >= 3 || st.epw) return 0;
  if (st.K > kLatMaxK || st.K < 32) return 0;
  if (sw.mfma_g >= 2 && (f64 ? st.tileN == 128 : st.tileM == 256)) return 0;   // tests that force the large-tile kernels
  if (sw.lat != 1 && ((int64_t)st.blocks * R > n_cu / 2 || st.K < 128)) return 0;
  auto tiles = [&](int T) { return st.Bt * ((st.M + T - 1) / T) * ((st.N + T - 1) / T); };
  const int64_t t16 = tiles(16), t32 = tiles(32), t64 = tiles(64);
  // ... and not more than one workgroup per CU (a second round of 512-thread workgroups doubles the step): beyond
  // that split-K / halved register-staged tiles win (100-site D = 256 network, R = 4: 3.49 vs 4.23 ms; R = 5: 5.33 vs
  // 4.61; R = 6: 5.55 vs 4.79; R = 8: 5.83 vs 4.37)
  auto fits = [&](int64_t t) { return t <= kLatMaxTiles && (sw.lat == 1 || t * R <= (int64_t)n_cu * sw.lat_wg_per_cu_x2 / 2); };
  if (f64) return fits(t16) ? 16 : 0;     // fp64: 16 x 16 tiles (v_mfma_f64_16x16x4_f64) only
  // (a k-contiguous operand reaches the fragment registers as 4-byte loads a row stride apart; with a short K split
  // eight ways there is nothing to pipeline them behind: 1024 x 1024 x 256, A k-contiguous, took 27.8 us on 256 tiles
  // of 64 x 64 against 23.7 us for the LDS-staged tiles with two K slabs - while the same tile count with both
  // operands row-contiguous, 4 x (256 x 1024 x 256), is 6 % faster here than there)
  const bool kcontig = st.modeA == 2 || st.modeB == 2;
  // (... and with K = 1024 the plain 64 x 64 tiles win: 16 replicas of 256 x 256 x 1024, 7.53 -> 7.03 ms per network)
  if (sw.lat_max_t >= 64 && t64 * R >= n_cu && fits(t64) && st.K <= 512 && (!kcontig || st.K >= sw.lat64_min_k)) return 64;
  if (t32 * R >= n_cu && fits(t32)) return 32;
  if (fits(t16)) return 16;
  if (fits(t32)) return 32;
  // half the CUs busy with 64 x 64 tiles still beats four K slabs plus a reduce pass (2 / 3 replicas of
  // 256 x 1024 x 256: 3.47 -> 3.28 / 3.62 -> 3.44 ms per 100-site network)
  if (sw.lat_max_t >= 64 && 2 * t64 * R >= n_cu && fits(t64) && st.K <= 256 && !kcontig) return 64;   // (a long K is better off split over workgroups)
  return 0;
}

// CTN_H=1 (tests): a step whose SHAPE admits the one-tile-per-CU form takes it whatever the launch size - the latency
// forms step aside (the remaining conditions - vector-storable C, not the last step - are checked by h_form itself)
static bool h_forced(const Step& st, int dtype, const DevSwitches& sw) {
  return sw.hform == 1 && dtype == CTN_F32 && st.kernel == CTN_KERNEL_MFMA_F32 && st.rhs >= 0 && !st.collapse &&
         st.modeA >= 1 && st.modeA <= 2 && st.modeB >= 1 && st.modeB <= 2 && st.K % 32 == 0 && st.K >= 64 && st.cvec &&
         (st.epw ? (st.epw_split && (st.epw == 2 || st.epw == 4)) : st.lhs2 < 0) &&
         st.Bt * ((st.M + 127) / 128) * ((st.N + 127) / 128) <= kMaxPartials;
}

// One-tile-per-CU form (k_mfma_f32_h, kernels_mfma_h.h): 128 x 128 tiles, K split over the two halves of an 8-wave
// workgroup.  Taken when the step's 128 x 128 tiles are about one round of one workgroup per CU - between three quarters
// of a chip and a whole one - which is where the register-staged kernel's 128 x 64 tiles run as a single round of two small
// workgroups per CU and the large-tile kernel would leave CUs idle (one MPS site applied to 4096 inputs: 256 tiles).
static bool h_form(const Plan& P, const Step& st, int R, int n_cu, int dtype, const DevSwitches& sw, bool c_vec) {
  if (sw.hform == 0 || dtype != CTN_F32 || st.kernel != CTN_KERNEL_MFMA_F32 || st.rhs < 0 || st.collapse) return false;
  if (st.modeA < 1 || st.modeA > 2 || st.modeB < 1 || st.modeB > 2) return false;      // LDS-DMA: 16-byte requests only
  if (st.epw ? !(st.epw_split && (st.epw == 2 || st.epw == 4)) : st.lhs2 >= 0) return false;
  if (st.K % 32 != 0 || st.K < 64 || !c_vec) return false;
  if (P.tensors[st.lhs].numel > (1LL << 30) || P.tensors[st.rhs].numel > (1LL << 30)) return false;   // 32-bit byte offsets
  const int64_t t128 = st.Bt * ((st.M + 127) / 128) * ((st.N + 127) / 128);
  if (t128 > kMaxPartials) return false;
  if (sw.hform == 1) return true;
  if (sw.mfma_g >= 2 && st.tileM == 256) return false;     // tests that force the large-tile kernels
  // (from three quarters of a chip of tiles: at half a chip - 8 networks of the headline in flight, 2048 inputs through
  // the batched MPS - the smaller tiles of the other forms keep more CUs busy: 27.1 vs 20.6 us per site at B = 2048)
  return 4 * t128 * R >= 3 * n_cu && t128 * R <= n_cu;
}

// Tile of the register-staged fp32 kernel for this step and replica count: the planner's 128 / 64 choice, halved
// (rows or columns, the larger extent first) while the launch has fewer than two workgroups per CU - one network with
// a batch leg, or a fused step whose output is a quarter of the GEMM it replaced - as long as the tile count still
// fits the step's partial-sum region.  Steps the planner made eligible for the large-tile kernel keep 128-unit
// counting (their fallback must agree with it), and so do steps that already go through the collapse pass.
static bool g_launch(const Step& st, int R, int n_cu, int use_g) {
  // does a launch of this (planner-eligible) step take the large-tile LDS-DMA kernel?  At least two of the big tiles
  // per CU and K >= 192 (below that the 128-tile kernel's 3-4 workgroups per CU hide the per-tile cost better), or a
  // long K (>= 1024) from 7/8 of a tile per CU; CTN_MFMA_G=2: whenever eligible (tests)
  if (!use_g || st.kernel != CTN_KERNEL_MFMA_F32 || st.tileM != 256) return false;
  const int64_t gtiles = st.Bt * ((st.M + 255) / 256) * ((st.N + st.tileN - 1) / st.tileN) * R;
  if (use_g >= 2 || (gtiles >= 2LL * n_cu && st.K >= 192)) return true;
  // ... below two tiles per CU a long K still pays, provided the tiles divide evenly over the CUs: 320 tiles are two
  // rounds on 64 CUs and one on the rest (100-site network: 160 replicas 42.8 ms against 39.7 on 128-wide tiles; 96
  // replicas = 192 tiles 25.2 against 23.1; 112 = 224 tiles 28.4 against 29.7, 128 = 256 tiles 29.0 against 29.1)
  const int64_t rounds = (gtiles + n_cu - 1) / n_cu;
  return st.K >= 1024 && 4 * gtiles >= 3LL * n_cu && 100 * gtiles >= 85 * rounds * n_cu;
}

// K split over workgroups ON the large-tile LDS-DMA kernel (256 x 128 tiles; k_mfma_f32_g with a.ks_S > 0, then the
// fixed-order reduce pass): a planner-eligible step whose tiles cannot fill the chip by themselves while K is long - the
// root GEMMs of a sliced 2D grid with a rank's few slices in flight (8 x (512 x 512 x 4096): 64 tiles; they ran as 512
// register-staged 64 x 64 tiles), a batch of 256 x 256 x 2^20 products.  Returns the number of splits (0 = not taken): about
// two workgroups per CU, every split at least 256 deep and a whole number of 16-deep k-tiles.
static int g_splitk(const Step& st, int R, int n_cu, int use_g, const DevSwitches& sw, bool c_vec) {
  if (!use_g || sw.g_splitk == 0 || st.kernel != CTN_KERNEL_MFMA_F32 || st.tileM != 256 || !c_vec || st.collapse || st.rhs < 0) return 0;
  if (st.K < 2048 || st.K % GK != 0) return 0;
  const int64_t gtiles = st.Bt * ((st.M + GM - 1) / GM) * ((st.N + st.tileN - 1) / st.tileN) * R;
  if (gtiles >= n_cu) return 0;
  int64_t S = (2LL * n_cu + gtiles - 1) / gtiles;
  S = std::min<int64_t>(std::min<int64_t>(S, 16), st.K / 256);   // (at most 16 slabs: one reduce pass, no folding)
  while (S > 1 && (st.K % S != 0 || (st.K / S) % GK != 0)) --S;
  return S >= 2 ? (int)S : 0;
}

static void plain_tiles(const Step& st, int R, int n_cu, int use_g, bool is_last, int* tm, int* tn, int halve = 1) {
  *tm = st.tileM == 64 ? 64 : kTileM;
  *tn = st.tileN;
  if (st.kernel != CTN_KERNEL_MFMA_F32 || st.collapse || !halve) return;
  // a step the planner made eligible for the large-tile kernel keeps 128-unit counting when that kernel will (or,
  // for the caller's possibly unaligned final buffer, may) take it
  if (st.tileM == 256 && (is_last || g_launch(st, R, n_cu, use_g))) return;
  auto tiles = [&](int a, int b) { return st.Bt * ((st.M + a - 1) / a) * ((st.N + b - 1) / b); };
  while (tiles(*tm, *tn) * R < 2LL * n_cu) {
    int a = *tm, b = *tn;
    // columns first: 128 x 64 tiles measured 62 vs 45 TFLOP/s for 64 x 128 on 4096 x 1024 x 256 (one batched-MPS site)
    if (b == 128 && st.N > 64) b = 64;
    else if (a == 128 && st.M > 64) a = 64;
    else break;
    if (tiles(a, b) > 4096) break;
    *tm = a; *tn = b;
  }
}

// the reduce pass of a split-K step: fold 16 to 1 while more than 16 slabs are left (ping-pong between the two
// slab buffers), then the final pass (rescale, C, abs-sum partials)
template <typename T>
static void launch_splitk_reduce(Exec* E, int partials, int R, const StepArgs& a, SplitKArgs sk);

// A streaming step with few output items and a long K (column sums, `ab,ab->b`: 4 workgroups walked K = 4096 one
// element at a time - 1.7 ms for 67 MB): split K over workgroups, partial sums through the split-K reduce pass
static int stream_splits(const Step& st, int R, int n_cu) {
  if (st.kernel != CTN_KERNEL_ELEMENT || st.K < 1024 || st.kvec) return 0;
  const int64_t wgs = (int64_t)st.blocks * R;
  if (wgs >= 2LL * n_cu) return 0;
  return (int)std::max<int64_t>(2, std::min<int64_t>(std::min<int64_t>(st.K / 128, 1024), (8LL * n_cu + wgs - 1) / wgs));
}

// k_stream variant of a plain streaming step: vector width along n and vectors per row lookup (4 only where it pays:
// short K - lookup-dominated - and enough rows left to fill the chip; long-K / small steps keep one for parallelism)
static void stream_variant(const Step& st, int R, int vw, int* u, int64_t* nq) {
  *nq = (st.Nv + vw - 1) / vw;
  const int64_t rows = st.H * st.L * (int64_t)R;
  *u = (*nq % 4 == 0 && st.K <= 16 && rows * (*nq / 4) >= (1 << 20)) ? 4 : 1;
}

template <typename T>
static void launch_stream_group(int vw, int u, dim3 g, hipStream_t st, const StepArgs* arr) {
  constexpr int VF = 16 / (int)sizeof(T);
  if (vw == VF) {
    if (u == 4) hipLaunchKernelGGL((k_stream_group<T, VF, 4>), g, dim3(256), 0, st, arr);
    else hipLaunchKernelGGL((k_stream_group<T, VF, 1>), g, dim3(256), 0, st, arr);
  } else {
    if (u == 4) hipLaunchKernelGGL((k_stream_group<T, 1, 4>), g, dim3(256), 0, st, arr);
    else hipLaunchKernelGGL((k_stream_group<T, 1, 1>), g, dim3(256), 0, st, arr);
  }
}

// Row-dot steps (one wave per output) with few outputs and a long K: split K over workgroups too
static int rowdot_splits(const Step& st, int R, int n_cu) {
  if (st.kernel != CTN_KERNEL_ROWDOT || st.K < 4096) return 0;
  const int64_t waves = st.H * st.L * st.Nv * (int64_t)R;
  if (waves >= 16LL * n_cu) return 0;
  return (int)std::max<int64_t>(2, std::min<int64_t>(std::min<int64_t>(st.K / 1024, 1024), (16LL * n_cu + waves - 1) / waves));
}

// Tiny output, huge K (CTN_KERNEL_DOT: at most 64 outputs): split K over workgroups as well, see k_dot_split
static int dot_splits(const Step& st) {
  if (st.kernel != CTN_KERNEL_DOT || st.K < 32768) return 0;
  return (int)std::min<int64_t>(1024, st.K / 8192);
}

template <int MA>
static void launch_sk_b(int mb, dim3 grid, hipStream_t st, const StepArgs& a, const SplitKArgs& sk) {
  switch (mb) {
    case 1: hipLaunchKernelGGL((k_mfma_f32_sk<MA, 1>), grid, dim3(256), 0, st, a, sk); break;
    case 2: hipLaunchKernelGGL((k_mfma_f32_sk<MA, 2>), grid, dim3(256), 0, st, a, sk); break;
    default: hipLaunchKernelGGL((k_mfma_f32_sk<MA, 0>), grid, dim3(256), 0, st, a, sk); break;
  }
}

static void launch_sk(int ma, int mb, dim3 grid, hipStream_t st, const StepArgs& a, const SplitKArgs& sk) {
  switch (ma) {
    case 1: launch_sk_b<1>(mb, grid, st, a, sk); break;
    case 2: launch_sk_b<2>(mb, grid, st, a, sk); break;
    default: launch_sk_b<0>(mb, grid, st, a, sk); break;
  }
}

template <int MA>
static void launch_sk64_b(int mb, dim3 grid, hipStream_t st, const StepArgs& a, const SplitKArgs& sk) {
  switch (mb) {
    case 1: hipLaunchKernelGGL((k_mfma_f64_sk<MA, 1>), grid, dim3(256), 0, st, a, sk); break;
    case 2: hipLaunchKernelGGL((k_mfma_f64_sk<MA, 2>), grid, dim3(256), 0, st, a, sk); break;
    default: hipLaunchKernelGGL((k_mfma_f64_sk<MA, 0>), grid, dim3(256), 0, st, a, sk); break;
  }
}

static void launch_sk64(int ma, int mb, dim3 grid, hipStream_t st, const StepArgs& a, const SplitKArgs& sk) {
  switch (ma) {
    case 1: launch_sk64_b<1>(mb, grid, st, a, sk); break;
    case 2: launch_sk64_b<2>(mb, grid, st, a, sk); break;
    default: launch_sk64_b<0>(mb, grid, st, a, sk); break;
  }
}

template <typename T>
static void launch_splitk_reduce(Exec* E, int partials, int R, const StepArgs& a, SplitKArgs sk) {
  T* bufs[2] = {(T*)E->d_slab, (T*)E->d_slab2};
  int cur = 0;
  while (sk.S > 16 && E->d_slab2) {
    const int S_out = (sk.S + 15) / 16;
    constexpr int V = 16 / sizeof(T);
    const dim3 grid((unsigned)((sk.numelC + 256 * V - 1) / (256 * V)), (unsigned)S_out, (unsigned)R);
    hipLaunchKernelGGL(k_splitk_fold<T>, grid, dim3(256), 0, E->stream, (const T*)bufs[cur], bufs[cur ^ 1],
                       sk.numelC, sk.S, S_out);
    cur ^= 1;
    sk.S = S_out;
  }
  sk.slab = bufs[cur];
  hipLaunchKernelGGL(k_splitk_reduce<T>, dim3(partials, R), dim3(256), 0, E->stream, a, sk);
}

// Do steps (s2 - 1, s2) form a zipper pair that k_zip_f32 can run as one launch?  Checked on the plan's own offset
// tables (every operand dense along its innermost index with uniform strides), so nothing about the network's labels
// is assumed: T = E . X with |m1| = 256 rows from E, columns (q, u) from X; E' = T . Y contracting (m1, q), |n2| = 256.
static bool zip_match(const Plan& P, int s2, Exec::ZipDesc* z, int u_mult = ZU) {
  if (s2 < 1 || s2 + 1 >= P.n_steps || P.dtype != CTN_F32) return false;
  const Step& a = P.steps[s2 - 1];
  const Step& b = P.steps[s2];
  auto plain = [&](const Step& st) {
    return st.kernel == CTN_KERNEL_MFMA_F32 && st.Bt == 1 && st.rhs >= 0 && st.lhs2 < 0 && !st.epw && st.modeA == 1 &&
           st.modeB == 1 && !st.collapse;
  };
  if (!plain(a) || !plain(b) || b.lhs != a.out || !b.cvec) return false;
  if (a.rhs >= P.n_inputs || b.rhs >= P.n_inputs) return false;            // X and Y: network inputs (scale 1)
  if (a.M != ZM || b.N != ZM || a.K % ZK != 0 || a.K < 2 * ZK || a.N % u_mult != 0) return false;
  const int32_t* T = P.tables.data();
  const int32_t *omA = T + a.t.omA, *okA = T + a.t.okA, *onB = T + a.t.onB, *okB = T + a.t.okB, *omC = T + a.t.omC, *onC = T + a.t.onC;
  const int64_t N1 = a.N;
  for (int i = 0; i < ZM; ++i) if (omA[i] != i || omC[i] != (int64_t)i * N1) return false;     // E rows dense, T = [m1][N1]
  for (int64_t n = 0; n < N1; ++n) if (onC[n] != n) return false;
  int64_t U = N1;
  for (int64_t n = 1; n < N1; ++n) if (onB[n] != onB[0] + n) { U = n; break; }
  if (onB[0] != 0 || U % u_mult != 0 || N1 % U != 0) return false;
  const int64_t Q = N1 / U, ldXq = Q > 1 ? onB[U] : 0;
  for (int64_t n = 0; n < N1; ++n) if (onB[n] != (n / U) * ldXq + n % U) return false;
  const int64_t ldE = okA[1] - okA[0], ldXk = okB[1] - okB[0];
  for (int64_t k = 0; k < a.K; ++k) if (okA[k] != k * ldE || okB[k] != k * ldXk) return false;
  if (ldE < ZM || b.M != U || b.K != (int64_t)ZM * Q) return false;
  const int32_t *omA2 = T + b.t.omA, *okA2 = T + b.t.okA, *onB2 = T + b.t.onB, *okB2 = T + b.t.okB, *omC2 = T + b.t.omC, *onC2 = T + b.t.onC;
  for (int64_t u = 0; u < U; ++u) if (omA2[u] != u) return false;                                   // T dense along u
  for (int i = 0; i < ZM; ++i) if (onB2[i] != i || onC2[i] != i) return false;
  const int64_t ldC = U > 1 ? omC2[1] - omC2[0] : ZM;
  for (int64_t u = 0; u < U; ++u) if (omC2[u] != u * ldC) return false;
  if (ldC < ZM) return false;
  // the contracted group of the second step enumerates (m1, q) in some order: read each entry's (m1, q) off T's offset
  int64_t ldYm = -1, ldYq = Q > 1 ? -1 : 0;
  for (int64_t k = 0; k < b.K; ++k) {
    const int64_t off = okA2[k], m1 = off / N1, q = (off % N1) / U;
    if (off % U != 0 || m1 >= ZM) return false;
    if (m1 == 1 && q == 0) ldYm = okB2[k];
    if (m1 == 0 && q == 1) ldYq = okB2[k];
  }
  if (ldYm < ZM || ldYq < 0) return false;
  std::vector<char> seen((size_t)b.K, 0);
  for (int64_t k = 0; k < b.K; ++k) {
    const int64_t off = okA2[k], m1 = off / N1, q = (off % N1) / U;
    if (okB2[k] != q * ldYq + m1 * ldYm || seen[(size_t)(m1 * Q + q)]) return false;
    seen[(size_t)(m1 * Q + q)] = 1;
  }
  z->on = true; z->ldE = ldE; z->ldXq = ldXq; z->ldXk = ldXk; z->ldYq = ldYq; z->ldYm = ldYm; z->ldC = ldC;
  z->Q = (int)Q; z->U = (int)U; z->K1 = (int)a.K;
  return true;
}

// Is step s an epilogue-summed GEMM step that k_sweep_f32 can take as one site of a sweep?  Checked on the plan's own
// offset tables: E row-major [b][l], W[l][p][r] with r unit-stride, x[b][p] with p unit-stride, result rows [b][r].
struct SweepShape { int64_t ldA = 0, ldC = 0, ldWl = 0, ldWp = 0, ldX = 0, M = 0; int D = 0, Pd = 0; };
static bool sweep_step_shape(const Plan& P, int s, SweepShape* sh) {
  const Step& st = P.steps[s];
  const int D = (int)st.K, Pd = st.epw;
  if (P.dtype != CTN_F32 || st.kernel != CTN_KERNEL_MFMA_F32 || (Pd != 2 && Pd != 4) || st.Bt != 1 ||
      (st.K != 64 && st.K != 128 && st.K != 256 && st.K != 512) || st.N != (int64_t)D * Pd || st.collapse || s + 1 >= P.n_steps)
    return false;
  if (st.rhs < 0 || st.rhs >= P.n_inputs || st.lhs2 < 0 || st.lhs2 >= P.n_inputs) return false;   // W, x: network inputs
  const int32_t* T = P.tables.data();
  const int32_t *omA = T + st.t.omA, *okA = T + st.t.okA, *onB = T + st.t.onB, *okB = T + st.t.okB, *omC = T + st.t.omC,
                *onC = T + st.t.onC, *omX = T + st.t.omA2;
  const int64_t M = st.M;
  const int64_t ldA = M > 1 ? omA[1] : D, ldC = M > 1 ? omC[1] : D, ldX = M > 1 ? omX[1] : Pd, ldWl = okB[1];
  for (int64_t m = 0; m < M; ++m)
    if (omA[m] != m * ldA || omC[m] != m * ldC || omX[m] != m * ldX) return false;
  for (int k = 0; k < D; ++k)
    if (okA[k] != k || okB[k] != (int64_t)k * ldWl) return false;
  int64_t ldWp = INT64_MAX;
  for (int n = 0; n < D * Pd; ++n)
    if (onB[n] >= D) ldWp = std::min<int64_t>(ldWp, onB[n]);
  if (ldWp == INT64_MAX) return false;
  std::vector<char> seen((size_t)D * Pd, 0);
  for (int n = 0; n < D * Pd; ++n) {
    const int64_t off = onB[n], pp = off / ldWp, rr = off % ldWp;
    if (off < 0 || pp >= Pd || rr >= D || onC[n] != rr || seen[(size_t)(pp * D + rr)]) return false;
    seen[(size_t)(pp * D + rr)] = 1;
  }
  if (ldA < D || ldC < D || ldA % 4 || ldC % 4 || ldX % Pd || ldX < Pd || ldWl % 4 || ldWp % 4 || ldWl < D) return false;
  if (((D + 12) * ldWl + 3 * ldWp + D) * 4 >= ((int64_t)1 << 31)) return false;   // a lane's offsets into a core: 32 bits
  sh->ldA = ldA; sh->ldC = ldC; sh->ldWl = ldWl; sh->ldWp = ldWp; sh->ldX = ldX; sh->M = M; sh->D = D; sh->Pd = Pd;
  return true;
}

// The longest run of such steps, each taking the result of the one before as its E.
static bool sweep_match(const Plan& P, Exec::SweepDesc* d) {
  std::vector<int> best;
  SweepShape best_first, best_last;
  std::vector<char> used((size_t)P.n_steps, 0);
  for (int s0 = 0; s0 < P.n_steps; ++s0) {
    SweepShape first, cur, last;
    if (used[s0] || !sweep_step_shape(P, s0, &first)) continue;
    std::vector<int> run{s0};
    last = first;
    for (int s = s0 + 1; s < P.n_steps; ++s) {
      // Only markers of absorbed steps (nothing is launched for them) may lie between two members: the ONE launch of
      // the run sits at its last member's position and reads the first member's input there, which the plan's arena
      // released right after the first member - any launched step in between (the other half of an interleaved
      // left / right chain, say) could have been given that region.  Such a step ends the run.
      if (P.steps[s].kernel == CTN_KERNEL_FUSED) continue;
      if (P.steps[s].lhs != P.steps[run.back()].out && P.steps[s].rhs != P.steps[run.back()].out &&
          P.steps[s].lhs2 != P.steps[run.back()].out)
        break;
      // the consumer of the run's last result: a member if it has the same shape and takes it as its E
      if (P.steps[s].lhs == P.steps[run.back()].out && sweep_step_shape(P, s, &cur) && cur.M == first.M && cur.D == first.D &&
          cur.Pd == first.Pd && cur.ldWl == first.ldWl && cur.ldWp == first.ldWp && cur.ldX == first.ldX && cur.ldA == last.ldC) {
        run.push_back(s);
        last = cur;
        continue;
      }
      break;
    }
    for (int s : run) used[s] = 1;
    if (run.size() > best.size()) { best = run; best_first = first; best_last = last; }
  }
  if (best.size() < 2) return false;
  {
    // the run's result must not land on its own input: blocks of rows start and finish at different times (a launch of
    // more than one round of workgroups), so a block's result rows may only cover ITS OWN input rows - the same region
    // with the same row stride - or nothing of the input at all
    const int idIn = P.steps[best.front()].lhs, idOut = P.steps[best.back()].out;
    const bool in_ws = idIn >= P.n_inputs, out_ws = idOut < P.n_inputs + P.n_steps - 1;
    if (in_ws && out_ws) {
      const int64_t es = P.elem_size();
      const int64_t a0 = P.tensors[idIn].ws_offset, a1 = a0 + ((best_first.M - 1) * best_first.ldA + best_first.D) * es;
      const int64_t b0 = P.tensors[idOut].ws_offset, b1 = b0 + ((best_first.M - 1) * best_last.ldC + best_first.D) * es;
      const bool overlap = a0 < b1 && b0 < a1;
      if (overlap && !(a0 == b0 && best_first.ldA == best_last.ldC)) return false;
    }
  }
  d->on = true; d->steps = best;
  d->ldIn = best_first.ldA; d->ldOut = best_last.ldC; d->ldWl = best_first.ldWl; d->ldWp = best_first.ldWp; d->ldX = best_first.ldX;
  d->J = (int)((best_first.M + SWR - 1) / SWR); d->M = (int)best_first.M; d->D = best_first.D; d->Pd = best_first.Pd;
  return true;
}

// Is this split dot step the sum over k = (a, b) of X[a Kb + b] Y[b ldY + a] - one operand contiguous in k, the other
// one its transpose (k_dot_tr)?  Checked on the plan's k tables.
static bool dot_tr_match(const Plan& P, const Step& st, Exec::DotTr* d) {
  if (st.kernel != CTN_KERNEL_DOT || st.K < 32768 || st.K >= (1LL << 31)) return false;
  const int32_t* T = P.tables.data();
  for (int x_is_a = 1; x_is_a >= 0; --x_is_a) {
    const int32_t* kx = T + (x_is_a ? st.t.okA : st.t.okB);
    const int32_t* ky = T + (x_is_a ? st.t.okB : st.t.okA);
    bool contig = true;
    for (int64_t k = 0; k < st.K && contig; ++k) contig = kx[k] == k;
    if (!contig || ky[0] != 0) continue;
    const int64_t ldY = ky[1];
    if (ldY < 32) continue;
    int64_t Kb = 0;
    for (int64_t k = 1; k < st.K; ++k) if (ky[k] == 1) { Kb = k; break; }
    if (Kb < 32 || st.K % Kb != 0) continue;
    const int64_t Ka = st.K / Kb;
    if (Ka % 32 != 0 || Kb % 32 != 0 || ldY < Ka) continue;
    bool ok = true;
    for (int64_t k = 0; k < st.K && ok; ++k) ok = ky[k] == (k % Kb) * ldY + k / Kb;
    if (!ok) continue;
    d->on = true; d->Ka = (int)Ka; d->Kb = (int)Kb; d->x_is_a = x_is_a; d->ldY = ldY;
    return true;
  }
  return false;
}

// Can step s run on k_mfma_f32_ares, and how many 128-column tiles should a workgroup walk?  Checked on the plan's own
// tables (0 = no): M = K = 256, N a multiple of 128, B vector-loadable along n or along k, every
// 128-column tile of C dense, more workgroup partials than slots (the step already goes through k_collapse), and
// enough tiles that workgroups of at least four fill the chip's last round.
static int ares_match(const Plan& P, int s, int R, int n_cu, const DevSwitches& sw) {
  const Step& st = P.steps[s];
  if (sw.ares == 0 || P.dtype != CTN_F32 || st.kernel != CTN_KERNEL_MFMA_F32 || st.rhs < 0 || st.lhs2 >= 0 || st.epw ||
      st.M != AR_M || st.K != AR_K || st.N % AR_TN != 0 || !st.collapse)
    return 0;
  if (st.modeB != 1 && st.modeB != 2) return 0;
  const int32_t* T = P.tables.data();
  const int32_t *onB = T + st.t.onB, *okB = T + st.t.okB, *onC = T + st.t.onC;
  for (int64_t n = 0; n < st.N; ++n)
    if (onC[n] != onC[n & ~(int64_t)(AR_TN - 1)] + (n & (AR_TN - 1))) return 0;
  if (st.modeB == 1) {
    for (int64_t n = 0; n < st.N; n += 4)
      if (onB[n + 1] != onB[n] + 1 || onB[n + 2] != onB[n] + 2 || onB[n + 3] != onB[n] + 3 || onB[n] % 4) return 0;
  } else {
    for (int k = 0; k < AR_K; k += 4)
      if (okB[k + 1] != okB[k] + 1 || okB[k + 2] != okB[k] + 2 || okB[k + 3] != okB[k] + 3 || okB[k] % 4) return 0;
    for (int64_t n = 0; n < st.N; ++n) if (onB[n] % 4) return 0;
  }
  const int64_t tiles = st.N / AR_TN;       // per batch entry: a workgroup's tiles are tiles of one matrix
  R *= (int)st.Bt;
  if (sw.ares != 1 && tiles * 2 * R < 2LL * n_cu) return 0;         // (the throughput regime of the large-tile kernel)
  if (sw.ares_ntw > 0) return tiles % sw.ares_ntw == 0 ? sw.ares_ntw : 0;
  // as many tiles per workgroup as still fill the last round of workgroups (a workgroup's load of A costs about a
  // quarter of a tile), at least 4
  for (int ntw : {64, 32, 16, 8, 4}) {
    if (tiles % ntw) continue;
    const int64_t wgs = tiles / ntw * R, rounds = (wgs + n_cu - 1) / n_cu;
    if (wgs * 100 >= rounds * n_cu * 94) return ntw;
  }
  return sw.ares == 1 && tiles % 4 == 0 ? 4 : 0;
}

// ... and can the resident operand be fetched with 16-byte loads along k?
static bool ares_avec(const Plan& P, int s) {
  const Step& st = P.steps[s];
  if (st.modeA != 2) return false;
  const int32_t* T = P.tables.data();
  const int32_t *omA = T + st.t.omA, *okA = T + st.t.okA, *obA = T + st.t.obA;
  if (obA[0] % 4 || P.tensors[st.lhs].numel % 4) return false;
  for (int k = 0; k < AR_K; ++k) if (okA[k] != okA[0] + k) return false;
  if (okA[0] % 4) return false;
  for (int m = 0; m < AR_M; ++m) if (omA[m] % 4) return false;
  return true;
}

static int exec_launch_steps(Exec* E) {
  const Plan& P = *E->plan;
  const int R = E->R;
  const bool chain = P.chain && E->d_chain != nullptr;
  if (chain) {
    const bool timed = E->timing_runs < E->timing_slots;
    const size_t ev0 = timed ? (size_t)E->timing_runs * P.n_steps * 2 : 0;
    if (timed) {
      HIPCHECK(hipEventRecord(E->events[ev0], E->stream));
    }
    if (P.dtype == CTN_F32)
      hipLaunchKernelGGL(k_chain<float>, dim3(R), dim3(256), 0, E->stream, (const ChainStep*)E->d_chain, P.n_steps,
                         (void* const*)E->d_ptrs, E->n_tensors, E->d_partials, R, P.min_norm, P.stabilize ? 1 : 0);
    else
      hipLaunchKernelGGL(k_chain<double>, dim3(R), dim3(256), 0, E->stream, (const ChainStep*)E->d_chain, P.n_steps,
                         (void* const*)E->d_ptrs, E->n_tensors, E->d_partials, R, P.min_norm, P.stabilize ? 1 : 0);
    if (timed) HIPCHECK(hipEventRecord(E->events[ev0 + 1], E->stream));  // whole walk = "step 0"
  }
  int group_skip = 0, collect_left = 0, collect_head = -1;
  for (int s = 0; s < P.n_steps && !chain; ++s) {
    const Step& st = P.steps[s];
    if (group_skip > 0) { --group_skip; continue; }   // went out with the head of its group
    if (collect_left == 0 && !E->groups.empty() && E->groups[s].len >= 2 && E->sw.group && !E->eager_rescale &&
        !(E->timing_runs < E->timing_slots)) {
      Exec::LeafGroup& G = E->groups[s];
      if (G.ready) {
        const dim3 g(G.blocks, R, G.len);
        if (P.dtype == CTN_F32) launch_stream_group<float>(G.vw, G.u, g, E->stream, E->d_group_args + G.off);
        else launch_stream_group<double>(G.vw, G.u, g, E->stream, E->d_group_args + G.off);
        group_skip = G.len - 1;
        continue;
      }
      collect_left = G.len; collect_head = s;   // first time: the steps' arguments are built below, not launched
    }
    if (!E->sweep_role.empty() && E->sweep_role[s] && !E->eager_rescale) {
      if ((int)E->launched_tile.size() != P.n_steps) E->launched_tile.assign(P.n_steps, 0);
      const bool timed_w = E->timing_runs < E->timing_slots;
      const size_t ew = timed_w ? ((size_t)E->timing_runs * P.n_steps + s) * 2 : 0;
      if (timed_w) HIPCHECK(hipEventRecord(E->events[ew], E->stream));
      if (E->sweep_role[s] == 1) {
        E->launched_tile[s] = (1 << 16) | 1;        // marker: absorbed into the next launched step
      } else {                                      // the last member: the whole chain, then its scale bookkeeping
        const Exec::SweepDesc& sd = E->sweep;
        const int S = (int)sd.steps.size();
        const Step& f0 = P.steps[sd.steps.front()];
        SweepArgs w{};
        w.ptrs = E->d_ptrs; w.n_tensors = E->n_tensors; w.site_ids = E->d_sweep_ids;
        w.idIn = f0.lhs; w.idOut = st.out; w.S = S; w.J = sd.J; w.M = sd.M;
        w.ldIn = sd.ldIn; w.ldOut = sd.ldOut; w.ldWl = sd.ldWl; w.ldWp = sd.ldWp; w.ldX = sd.ldX;
        w.partIn = nullptr; w.PIn = 0; w.strideIn = 0; w.numelIn = 1.0;
        if (f0.lhs >= P.n_inputs && P.stabilize && P.steps[P.tensors[f0.lhs].producer].kernel != CTN_KERNEL_FUSED) {
          const int ps = P.tensors[f0.lhs].producer;
          w.partIn = E->d_partials + (size_t)E->step_off[ps] * R;
          w.PIn = w.strideIn = E->step_partials[ps];
          w.numelIn = (double)P.tensors[f0.lhs].numel;
        }
        w.min_norm = P.min_norm;
        w.rec_a = E->d_sweep_a; w.rec_s = E->d_sweep_s; w.dbg = nullptr;
        E->launched_tile[s] = (SWR << 16) | (sd.D * sd.Pd);   // 16 rows x all columns (1024 at bond 256, d = 4) per workgroup, every site
        if (E->sw.stamps && (E->sw.stamp_step < 0 || E->sw.stamp_step == s)) {
          const size_t need = (size_t)sd.J * R;
          if (E->dbg_tiles < need) {
            if (E->d_dbg) (void)hipFree(E->d_dbg);
            HIPCHECK(hipMalloc((void**)&E->d_dbg, need * 64));
            E->dbg_tiles = need;
          }
          w.dbg = E->d_dbg;
          HIPCHECK(hipMemsetAsync(E->d_dbg, 0, E->dbg_tiles * 64, E->stream));
        }
        {
          const dim3 gs((unsigned)sd.J, (unsigned)R);
#define CTN_SWEEP_LAUNCH(DD)                                                                                   \
          do {                                                                                                 \
            if (sd.Pd == 4) hipLaunchKernelGGL((k_sweep_f32<DD, 4>), gs, dim3(DD == 64 ? 256 : 512), 0, E->stream, w);  \
            else hipLaunchKernelGGL((k_sweep_f32<DD, 2>), gs, dim3(DD == 64 ? 256 : 512), 0, E->stream, w);            \
          } while (0)
          if (sd.D == 64) CTN_SWEEP_LAUNCH(64);
          else if (sd.D == 128) CTN_SWEEP_LAUNCH(128);
          else if (sd.D == 256) CTN_SWEEP_LAUNCH(256);
          else CTN_SWEEP_LAUNCH(512);
#undef CTN_SWEEP_LAUNCH
        }
        const double numel = (double)P.tensors[st.out].numel;
        hipLaunchKernelGGL(k_sweep_logs, dim3((unsigned)S, (unsigned)R), dim3(256), 0, E->stream, (const double*)E->d_sweep_a,
                           (const float*)E->d_sweep_s, S, sd.J, E->d_sweep_la, E->d_sweep_ls);
        hipLaunchKernelGGL(k_sweep_z, dim3((unsigned)S, (unsigned)R), dim3(256), 0, E->stream, (const double*)E->d_sweep_la,
                           (const double*)E->d_sweep_ls, S, sd.J, numel, E->d_sweep_z);
        SweepFinish f{};
        f.ptrs = E->d_ptrs; f.n_tensors = E->n_tensors; f.idOut = st.out; f.S = S; f.J = sd.J; f.R = R; f.D = sd.D; f.M = sd.M;
        f.ldOut = sd.ldOut; f.Z = E->d_sweep_z; f.ls = E->d_sweep_ls; f.part_off = E->d_sweep_off;
        f.part_slots = E->d_sweep_slots; f.partials = E->d_partials; f.numel = numel;
        f.min_norm = P.stabilize ? P.min_norm : INFINITY;
        hipLaunchKernelGGL(k_sweep_finish, dim3((unsigned)sd.J, (unsigned)R), dim3(256), 0, E->stream, f);
      }
      if (timed_w) HIPCHECK(hipEventRecord(E->events[ew + 1], E->stream));
      continue;
    }
    if (!E->zl_skip.empty() && E->zl_skip[s]) {     // first step of a zipper pair in its latency form (k_zip_lat)
      if ((int)E->launched_tile.size() != P.n_steps) E->launched_tile.assign(P.n_steps, 0);
      E->launched_tile[s] = (1 << 16) | 1;          // marker: absorbed into the next launched step
      if (E->timing_runs < E->timing_slots) {
        const size_t e0 = ((size_t)E->timing_runs * P.n_steps + s) * 2;
        HIPCHECK(hipEventRecord(E->events[e0], E->stream));
        HIPCHECK(hipEventRecord(E->events[e0 + 1], E->stream));
      }
      continue;
    }
    if (!E->zl.empty() && E->zl[s].on) {
      const Exec::ZipLat& L = E->zl[s];
      const Exec::ZipDesc& zd = L.d;
      const Step& s1 = P.steps[s - 1];
      const int S = ZM / L.mp, nub = zd.U / 16;
      const bool eager = E->eager_rescale;
      ZipLatArgs z{};
      z.ptrs = E->d_ptrs; z.n_tensors = E->n_tensors;
      z.idE = s1.lhs; z.idX = s1.rhs; z.idY = st.rhs;
      z.slabs_in = (L.from_prev && !eager) ? E->d_zl_slab[L.buf ^ 1] : nullptr;
      z.slabs_out = E->d_zl_slab[L.buf];
      z.ldE = zd.ldE; z.ldXq = zd.ldXq; z.ldXk = zd.ldXk; z.ldYq = zd.ldYq; z.ldYm = zd.ldYm;
      z.U = zd.U; z.R = R;
      z.partE = nullptr; z.PE = 0; z.strideE = 0; z.numelE = 1.0;
      if (s1.lhs >= P.n_inputs && P.stabilize && !eager && P.steps[P.tensors[s1.lhs].producer].kernel != CTN_KERNEL_FUSED) {
        const int ps = P.tensors[s1.lhs].producer;
        z.partE = E->d_partials + (size_t)E->step_off[ps] * R;
        z.PE = z.strideE = E->step_partials[ps];
        z.numelE = (double)P.tensors[s1.lhs].numel;
      }
      z.min_norm = P.min_norm;
      z.partC = E->d_partials + (size_t)E->step_off[s] * R;
      z.partC_stride = E->step_partials[s];
      const bool timed_z = E->timing_runs < E->timing_slots;
      const size_t ez = timed_z ? ((size_t)E->timing_runs * P.n_steps + s) * 2 : 0;
      if (timed_z) HIPCHECK(hipEventRecord(E->events[ez], E->stream));
      if ((int)E->launched_tile.size() != P.n_steps) E->launched_tile.assign(P.n_steps, 0);
      E->launched_tile[s] = (L.mp << 16) | ZM;      // (m1 part, all n2) per workgroup, 16 values of u
      const dim3 gz((unsigned)((int64_t)R * nub * S));
      z.dbg = nullptr;
      if (E->sw.stamps && (E->sw.stamp_step < 0 || E->sw.stamp_step == s)) {
        const size_t need = (size_t)gz.x;
        if (E->dbg_tiles < need) {
          if (E->d_dbg) (void)hipFree(E->d_dbg);
          HIPCHECK(hipMalloc((void**)&E->d_dbg, need * 64));
          E->dbg_tiles = need;
        }
        z.dbg = E->d_dbg;
        HIPCHECK(hipMemsetAsync(E->d_dbg, 0, E->dbg_tiles * 64, E->stream));
      }
      if (zd.Q == 4 && L.mp == 32) hipLaunchKernelGGL((k_zip_lat<4, 32>), gz, dim3(512), 0, E->stream, z);
      else if (zd.Q == 4) hipLaunchKernelGGL((k_zip_lat<4, 64>), gz, dim3(512), 0, E->stream, z);
      else hipLaunchKernelGGL((k_zip_lat<2, 64>), gz, dim3(512), 0, E->stream, z);
      if (!L.feeds_next || eager) {                 // nobody adds these slabs up while loading them: do it here
        hipLaunchKernelGGL(k_zip_slab_sum, dim3((unsigned)E->step_partials[s], (unsigned)R), dim3(256), 0, E->stream,
                           (const float*)z.slabs_out, S, zd.U, (void* const*)E->d_ptrs, E->n_tensors, st.out, zd.ldC, z.partC,
                           z.partC_stride);
        if (eager && P.stabilize && s + 1 < P.n_steps) {
          const int64_t numel = P.tensors[st.out].numel;
          const dim3 g((unsigned)std::max<int64_t>(1, std::min<int64_t>((numel / 4 + 255) / 256, 2048)), R);
          hipLaunchKernelGGL(k_renorm<float>, g, dim3(256), 0, E->stream, (void* const*)E->d_ptrs, E->n_tensors, st.out, numel,
                             (const double*)z.partC, E->step_partials[s], E->step_partials[s], P.min_norm);
        }
      }
      if (timed_z) HIPCHECK(hipEventRecord(E->events[ez + 1], E->stream));
      continue;
    }
    if (!E->zip_skip.empty() && E->zip_skip[s]) {   // first step of a zipper pair: runs inside the next step's launch
      if ((int)E->launched_tile.size() != P.n_steps) E->launched_tile.assign(P.n_steps, 0);
      E->launched_tile[s] = (1 << 16) | 1;          // marker: absorbed into the next launched step
      if (E->timing_runs < E->timing_slots) {
        const size_t e0 = ((size_t)E->timing_runs * P.n_steps + s) * 2;
        HIPCHECK(hipEventRecord(E->events[e0], E->stream));
        HIPCHECK(hipEventRecord(E->events[e0 + 1], E->stream));
      }
      continue;
    }
    if (!E->zip.empty() && E->zip[s].on) {
      const Exec::ZipDesc& zd = E->zip[s];
      const Step& s1 = P.steps[s - 1];
      ZipArgs z{};
      z.ptrs = E->d_ptrs; z.n_tensors = E->n_tensors;
      z.idE = s1.lhs; z.idX = s1.rhs; z.idY = st.rhs; z.idC = st.out;
      z.ldE = zd.ldE; z.ldXq = zd.ldXq; z.ldXk = zd.ldXk; z.ldYq = zd.ldYq; z.ldYm = zd.ldYm; z.ldC = zd.ldC;
      z.K1 = zd.K1; z.Q = zd.Q; z.U = zd.U;
      z.partE = nullptr; z.PE = 0; z.strideE = 0; z.numelE = 1.0;
      if (s1.lhs >= P.n_inputs && P.stabilize && !E->eager_rescale && P.steps[P.tensors[s1.lhs].producer].kernel != CTN_KERNEL_FUSED) {
        const int ps = P.tensors[s1.lhs].producer;
        z.partE = E->d_partials + (size_t)E->step_off[ps] * R;
        z.PE = z.strideE = E->step_partials[ps];
        z.numelE = (double)P.tensors[s1.lhs].numel;
      }
      z.min_norm = P.min_norm;
      z.partC = E->d_partials + (size_t)E->step_off[s] * R;
      z.partC_stride = E->step_partials[s];
      z.R = R; z.dbg = nullptr;
      const bool timed_z = E->timing_runs < E->timing_slots;
      const size_t ez = timed_z ? ((size_t)E->timing_runs * P.n_steps + s) * 2 : 0;
      if (timed_z) HIPCHECK(hipEventRecord(E->events[ez], E->stream));
      if ((int)E->launched_tile.size() != P.n_steps) E->launched_tile.assign(P.n_steps, 0);
      E->launched_tile[s] = (512 << 16) | (zd.zu == 64 ? 128 : 256);   // the fused pair: 128 (or 64) values of u x all 256 n2 per workgroup
      const int per = zd.U / zd.zu;
      if (E->sw.stamps && (E->sw.stamp_step < 0 || E->sw.stamp_step == s)) {
        const size_t need = (size_t)per * R;
        if (E->dbg_tiles < need) {
          if (E->d_dbg) (void)hipFree(E->d_dbg);
          HIPCHECK(hipMalloc((void**)&E->d_dbg, need * 64));
          E->dbg_tiles = need;
        }
        z.dbg = E->d_dbg;
        HIPCHECK(hipMemsetAsync(E->d_dbg, 0, E->dbg_tiles * 64, E->stream));
      }
      if (zd.zu == 64) hipLaunchKernelGGL(k_zip64_f32, dim3((unsigned)((int64_t)per * R)), dim3(512), 0, E->stream, z);
      else hipLaunchKernelGGL(k_zip_f32, dim3((unsigned)((int64_t)per * R)), dim3(512), 0, E->stream, z);
      if (E->eager_rescale && P.stabilize && s + 1 < P.n_steps) {
        const int64_t numel = P.tensors[st.out].numel;
        const dim3 g((unsigned)std::max<int64_t>(1, std::min<int64_t>((numel / 4 + 255) / 256, 2048)), R);
        hipLaunchKernelGGL(k_renorm<float>, g, dim3(256), 0, E->stream, (void* const*)E->d_ptrs, E->n_tensors, st.out, numel,
                           (const double*)z.partC, E->step_partials[s], E->step_partials[s], P.min_norm);
      }
      if (timed_z) HIPCHECK(hipEventRecord(E->events[ez + 1], E->stream));
      continue;
    }
    if (st.kernel == CTN_KERNEL_FUSED) {   // formed on the fly inside its consumer: nothing to launch
      if (E->timing_runs < E->timing_slots) {
        const size_t e0 = ((size_t)E->timing_runs * P.n_steps + s) * 2;
        HIPCHECK(hipEventRecord(E->events[e0], E->stream));
        HIPCHECK(hipEventRecord(E->events[e0 + 1], E->stream));
      }
      continue;
    }
    StepArgs a;
    const int32_t* T = E->d_tables;
    const int64_t* T8 = E->d_tables64;
    a.obA = T8 + st.t.obA; a.obB = T8 + st.t.obB; a.obC = T8 + st.t.obC;
    a.omA = T + st.t.omA; a.omC = T + st.t.omC;
    a.onB = T + st.t.onB; a.onC = T + st.t.onC;
    a.okA = T + st.t.okA; a.okB = T + st.t.okB;
    a.ptrs = E->d_ptrs;
    auto part_of = [&](int id, const double** p, int32_t* cnt, int32_t* stride, double* numel) {
      *p = nullptr; *cnt = 0; *stride = 0; *numel = 1;
      if (id >= P.n_inputs && P.stabilize && !E->eager_rescale && P.steps[P.tensors[id].producer].kernel != CTN_KERNEL_FUSED) {
        const int ps = P.tensors[id].producer;
        *p = E->d_partials + (size_t)E->step_off[ps] * R;
        *cnt = *stride = E->step_partials[ps];
        *numel = (double)P.tensors[id].numel;
      }
    };
    part_of(st.lhs, &a.partA, &a.PA, &a.strideA, &a.numelA);
    part_of(st.rhs, &a.partB, &a.PB, &a.strideB, &a.numelB);
    part_of(st.lhs2, &a.partA2, &a.PA2, &a.strideA2, &a.numelA2);
    a.obA2 = T8 + st.t.obA2; a.omA2 = T + st.t.omA2; a.okA2 = T + st.t.okA2;
    a.idA2 = st.lhs2 >= 0 ? st.lhs2 : E->n_tensors - 1;
    a.krX = st.krX; a.krY = st.krY;
    a.epw = st.epw | (st.epw_split ? 0x100 : 0);
    double* part_dst = E->d_partials + (size_t)E->step_off[s] * R;
    const int part_stride = E->step_partials[s];    // slots per replica of this step's region
    a.partC = st.collapse ? E->d_scratch : part_dst;
    a.partC_stride = st.collapse ? st.blocks : part_stride;
    int collapse_blocks = st.blocks;   // partials written per replica when the step collapses (a launcher may retile)
    bool reduced = false;              // a K-split streaming / row-dot step: its reduce pass wrote the partials itself
    bool do_collapse = st.collapse;    // more workgroup partials than slots: through the scratch buffer and k_collapse
    a.min_norm = P.min_norm;
    a.Bt = (int32_t)st.Bt; a.M = (int32_t)st.M; a.N = (int32_t)st.N; a.K = (int32_t)st.K;
    a.idA = st.lhs; a.idB = st.rhs >= 0 ? st.rhs : E->n_tensors - 1; a.idC = st.out;
    a.n_tensors = E->n_tensors;
    const int row_tile = (st.kernel == CTN_KERNEL_MFMA_F32 && st.tileM == 64) ? 64 : kTileM;
    a.tiles_m = (int32_t)((st.M + row_tile - 1) / row_tile);
    a.tiles_n = (int32_t)((st.N + kTileN - 1) / kTileN);
    a.blocks_per_replica = st.blocks;
    a.R = R;
    a.c_vec = (st.cvec && (s + 1 < P.n_steps || E->outs_aligned16)) ? 1 : 0;
    a.dbg = nullptr;
    a.ks_slab = nullptr; a.ks_numelC = 0; a.ks_S = 0; a.ks_chunk = 0;
    a.ohA = T8 + st.t.ohA; a.ohB = T8 + st.t.ohB; a.ohC = T8 + st.t.ohC;
    a.olA = T + st.t.olA; a.olB = T + st.t.olB; a.olC = T + st.t.olC;
    a.H = (int32_t)st.H; a.L = (int32_t)st.L; a.Nv = (int32_t)st.Nv;
    a.sAn = (int32_t)st.sAn; a.sBn = (int32_t)st.sBn;
    a.dNq = make_fastdiv((st.Nv + st.vecw - 1) / st.vecw);
    a.dL = make_fastdiv(st.L);
    a.dNv = make_fastdiv(st.Nv);

    const bool timed = E->timing_runs < E->timing_slots;  // only the first `slots` enqueues are bracketed
    const size_t ev0 = timed ? ((size_t)E->timing_runs * P.n_steps + s) * 2 : 0;
    if (timed) HIPCHECK(hipEventRecord(E->events[ev0], E->stream));
    if ((int)E->launched_tile.size() != P.n_steps) E->launched_tile.assign(P.n_steps, 0);
    auto used_tile = [&](int tm, int tn) { E->launched_tile[s] = (tm << 16) | tn; };
    switch (st.kernel) {
      case CTN_KERNEL_MFMA_F32: {
        const int64_t total = (int64_t)st.blocks * R;
        if (total >= (1LL << 31)) { g_err = "grid too large"; return CTN_UNSUPPORTED; }
        if (const int ntw_av = (int)E->ares_ntw.size() == P.n_steps ? E->ares_ntw[s] : 0) {
          const int ntw = ntw_av & 0xffff, avec = ntw_av >> 16;
          // the left operand resident in registers, `ntw` column tiles per workgroup (k_mfma_f32_ares); one partial per
          // workgroup through the collapse pass the step has anyway
          const int wgs = (int)(st.Bt * (st.N / AR_TN / ntw));
          a.partC = E->d_scratch; a.partC_stride = wgs;
          do_collapse = true; collapse_blocks = wgs;
          used_tile(256, std::min(AR_TN * ntw, 32768));
          const dim3 ga((unsigned)((int64_t)wgs * R));
          if (st.modeB == 2) {
            if (avec) hipLaunchKernelGGL((k_mfma_f32_ares<2, 1>), ga, dim3(512), 0, E->stream, a, ntw);
            else hipLaunchKernelGGL((k_mfma_f32_ares<2, 0>), ga, dim3(512), 0, E->stream, a, ntw);
          } else {
            if (avec) hipLaunchKernelGGL((k_mfma_f32_ares<1, 1>), ga, dim3(512), 0, E->stream, a, ntw);
            else hipLaunchKernelGGL((k_mfma_f32_ares<1, 0>), ga, dim3(512), 0, E->stream, a, ntw);
          }
          break;
        }
        if (const int S = (E->d_slab && s + 1 < P.n_steps) ? g_splitk(st, R, E->n_cu, E->mfma_g, E->sw, st.cvec) : 0) {
          SplitKArgs sk{};
          sk.slab = E->d_slab;
          sk.numelC = P.tensors[st.out].numel;
          sk.S = S;
          sk.kchunk = (int32_t)(st.K / S);
          a.ks_slab = sk.slab; a.ks_numelC = sk.numelC; a.ks_S = S; a.ks_chunk = sk.kchunk;
          a.tiles_m = (int32_t)((st.M + GM - 1) / GM);
          a.tiles_n = (int32_t)((st.N + st.tileN - 1) / st.tileN);
          a.blocks_per_replica = (int32_t)(st.Bt * a.tiles_m * a.tiles_n * S);
          used_tile(256, 128);
          const dim3 gg((unsigned)((int64_t)a.blocks_per_replica * R));
          const bool use_asm = !E->sw.g_no_asm;          // (every split is whole k-tiles)
#define CTN_G_LAUNCH(AA, BB)                                                                             \
          do {                                                                                           \
            if (use_asm) hipLaunchKernelGGL((k_mfma_f32_g<4, 2, true, AA, BB>), gg, dim3(256), 0, E->stream, a); \
            else hipLaunchKernelGGL((k_mfma_f32_g<4, 2, false, AA, BB>), gg, dim3(256), 0, E->stream, a);        \
          } while (0)
          if (st.modeA == 2 && st.modeB == 1) CTN_G_LAUNCH(2, 1);
          else if (st.modeA == 1 && st.modeB == 2) CTN_G_LAUNCH(1, 2);
          else if (st.modeA == 2 || st.modeB == 2) CTN_G_LAUNCH(2, 2);
          else CTN_G_LAUNCH(1, 1);
#undef CTN_G_LAUNCH
          a.ks_slab = nullptr; a.ks_S = 0;
          a.partC = part_dst; a.partC_stride = part_stride; reduced = true;
          launch_splitk_reduce<float>(E, E->step_partials[s], R, a, sk);
          break;
        }
        if (const int T = lat_form(st, R, E->n_cu, P.dtype, E->sw)) {
          a.tiles_m = (int32_t)((st.M + T - 1) / T);
          a.tiles_n = (int32_t)((st.N + T - 1) / T);
          a.blocks_per_replica = (int32_t)(st.Bt * a.tiles_m * a.tiles_n);
          a.partC = part_dst; a.partC_stride = part_stride; reduced = true;   // one partial per tile, written directly
          const int ks = T == 64 ? 2 : 8, kp = T == 16 ? 4 : 2;
          const int kchunk = (int)(((st.K + ks - 1) / ks + kp - 1) / kp * kp);
          used_tile(T, T);
          const dim3 g((unsigned)((int64_t)a.blocks_per_replica * R));
          if (T == 16) hipLaunchKernelGGL((k_mfma_lat<16, float>), g, dim3(512), 0, E->stream, a, kchunk);
          else if (T == 32) hipLaunchKernelGGL((k_mfma_lat<32, float>), g, dim3(512), 0, E->stream, a, kchunk);
          else hipLaunchKernelGGL((k_mfma_lat<64, float>), g, dim3(512), 0, E->stream, a, kchunk);
          break;
        }
        if (const int S = E->d_slab ? splitk_splits(st, R, E->n_cu, P.dtype, E->sw) : 0) {
          SplitKArgs sk;
          sk.slab = E->d_slab;
          sk.numelC = P.tensors[st.out].numel;
          sk.S = S;
          sk.kchunk = (int32_t)(((st.K + S - 1) / S + 31) / 32 * 32);
          sk.S = (int32_t)((st.K + sk.kchunk - 1) / sk.kchunk);  // drop empty trailing splits
          sk.tiles_m = (int32_t)((st.M + 63) / 64);
          sk.tiles_n = (int32_t)((st.N + 63) / 64);
          sk.tiles_per_replica = (int32_t)(st.Bt * sk.tiles_m * sk.tiles_n);
          used_tile(64, 64);
          launch_sk(st.modeA, st.modeB, dim3((unsigned)((int64_t)sk.tiles_per_replica * sk.S * R)), E->stream, a, sk);
          launch_splitk_reduce<float>(E, E->step_partials[s], R, a, sk);
          break;
        }
        if (!g_launch(st, R, E->n_cu, E->mfma_g) && s + 1 < P.n_steps && h_form(P, st, R, E->n_cu, P.dtype, E->sw, a.c_vec != 0)) {
          a.tiles_m = (int32_t)((st.M + 127) / 128);
          a.tiles_n = (int32_t)((st.N + 127) / 128);
          a.blocks_per_replica = (int32_t)(st.Bt * a.tiles_m * a.tiles_n);
          a.partC = part_dst; a.partC_stride = part_stride; do_collapse = false;   // one partial per tile (<= kMaxPartials)
          if (E->sw.stamps && (E->sw.stamp_step < 0 || E->sw.stamp_step == s)) {
            const size_t need = (size_t)a.blocks_per_replica * R;
            if (E->dbg_tiles < need) {
              if (E->d_dbg) (void)hipFree(E->d_dbg);
              HIPCHECK(hipMalloc((void**)&E->d_dbg, need * 64));
              E->dbg_tiles = need;
            }
            a.dbg = E->d_dbg;
            HIPCHECK(hipMemsetAsync(E->d_dbg, 0, E->dbg_tiles * 64, E->stream));
          }
          used_tile(128, 128);
          const dim3 gh((unsigned)((int64_t)a.blocks_per_replica * R));
#define CTN_H_LAUNCH(AA, BB)                                                                              \
          do {                                                                                            \
            if (st.epw == 4) hipLaunchKernelGGL((k_mfma_f32_h<AA, BB, 4>), gh, dim3(512), 0, E->stream, a);      \
            else if (st.epw == 2) hipLaunchKernelGGL((k_mfma_f32_h<AA, BB, 2>), gh, dim3(512), 0, E->stream, a); \
            else hipLaunchKernelGGL((k_mfma_f32_h<AA, BB, 0>), gh, dim3(512), 0, E->stream, a);                  \
          } while (0)
          if (st.modeA == 2 && st.modeB == 2) CTN_H_LAUNCH(2, 2);
          else if (st.modeA == 2) CTN_H_LAUNCH(2, 1);
          else if (st.modeB == 2) CTN_H_LAUNCH(1, 2);
          else CTN_H_LAUNCH(1, 1);
#undef CTN_H_LAUNCH
          break;
        }
        a.tiles_n = (int32_t)((st.N + st.tileN - 1) / st.tileN);
        if (E->sw.stamps && (E->sw.stamp_step < 0 || E->sw.stamp_step == s)) {
          if (E->dbg_tiles < (size_t)total) {
            if (E->d_dbg) (void)hipFree(E->d_dbg);
            HIPCHECK(hipMalloc((void**)&E->d_dbg, (size_t)total * 64));
            E->dbg_tiles = (size_t)total;
          }
          a.dbg = E->d_dbg;
          HIPCHECK(hipMemsetAsync(E->d_dbg, 0, E->dbg_tiles * 64, E->stream));
        }
        // large-tile LDS-DMA variant (kernels_mfma_g.h) where the step's shape allows it.
        // CTN_MFMA_G (read when the executor is created): 0 = never, 1 = 256x128 tiles when the launch
        // fills the chip (default), 2 = whenever eligible (tests)
        const int use_g = E->mfma_g;
        static_assert(GM == 256 && GN == kTileN && GK == 16 && 2 * GK == 32, "planner eligibility rule (plan.cpp) assumes these");
        // ... and the launch has at least two of the big tiles per CU: with fewer, 128-row tiles spread the
        // same work over more CUs (measured: 2048^3 runs at 90 vs 56 TFLOP/s, 4096^3 at 125 vs 135)
        const int64_t gtiles = st.Bt * ((st.M + GM - 1) / GM) * a.tiles_n * R;
        const bool no_asm = E->sw.g_no_asm;
        const bool kcontig = st.modeA == 2 || st.modeB == 2;
        // ... and K >= 192: below that the 128-tile kernel's 3-4 workgroups per CU hide the per-tile cost
        // better (8192 x 8192 x K: K = 64 old +7 %, 128 +2 %, 192 equal, 256 large tiles +5 %)
        // ... and long K (>= 1024) already from 3/4 of a tile per CU: the per-tile cost amortises over the k loop
        // (256 x 256 x 1024 per replica, R = 96 / 128: 86.9 / 106.2 vs 80.3 / 95.8 TFLOP/s; R = 48 / 64 lose)
        if (g_launch(st, R, E->n_cu, use_g) && a.c_vec) {
          a.tiles_m = (int32_t)((st.M + GM - 1) / GM);
          // long-K steps on narrow outputs whose tiles are all full also exist as 256 x 256 tiles (8 waves, one
          // workgroup per CU): with N <= 512 each A tile is fetched half as often (256 x 256 x 1024 per replica:
          // 133.9 vs 130.3 TFLOP/s); K = 256 steps (116.0 vs 116.7) and wide outputs (8192 x 8192 x 768: 130.3 vs
          // 132.1) are better off with 256 x 128
          // - and so are very long K on any width (K = 2048 ... 8192: +1.5 ... +3 %)
          const bool big = use_g == 1 && E->sw.g_big && !kcontig && st.M % 256 == 0 && st.N % 256 == 0 && st.K % GK == 0 &&
                           ((st.K >= 512 && st.N <= 512) || st.K >= E->sw.g_big_min_k) && gtiles / 2 >= (int64_t)E->n_cu;
          if (big) {
            used_tile(256, 256);
            a.tiles_n = (int32_t)(st.N / 256);
            a.blocks_per_replica = (int32_t)(st.Bt * a.tiles_m * a.tiles_n);
            hipLaunchKernelGGL((k_mfma_f32_g<8, 2, true>), dim3((unsigned)((int64_t)a.blocks_per_replica * R)), dim3(512), 0,
                               E->stream, a);
            break;
          }
          {
            used_tile(256, 128);
            a.blocks_per_replica = (int32_t)(st.Bt * a.tiles_m * a.tiles_n);
            const dim3 gg((unsigned)((int64_t)a.blocks_per_replica * R));
            // hand-scheduled blocks for whole k-tiles, the C++ loop (which masks a ragged last k-tile) otherwise
            const bool use_asm = st.K % GK == 0 && !no_asm;
#define CTN_G_LAUNCH(AA, BB)                                                                             \
            do {                                                                                         \
              if (use_asm) hipLaunchKernelGGL((k_mfma_f32_g<4, 2, true, AA, BB>), gg, dim3(256), 0, E->stream, a); \
              else hipLaunchKernelGGL((k_mfma_f32_g<4, 2, false, AA, BB>), gg, dim3(256), 0, E->stream, a);        \
            } while (0)
            if (st.modeA == 2 && st.modeB == 1) CTN_G_LAUNCH(2, 1);
            else if (st.modeA == 1 && st.modeB == 2) CTN_G_LAUNCH(1, 2);
            else if (kcontig) CTN_G_LAUNCH(2, 2);
            else CTN_G_LAUNCH(1, 1);
#undef CTN_G_LAUNCH
          }
          break;
        }
        // register-staged tiles; halved (128 -> 64 rows / columns) while the launch is under-filled - see plain_tiles
        int tm = row_tile, tn = st.tileN;
        plain_tiles(st, R, E->n_cu, E->mfma_g, s + 1 == P.n_steps, &tm, &tn, E->sw.halve);
        a.tiles_m = (int32_t)((st.M + tm - 1) / tm);
        a.tiles_n = (int32_t)((st.N + tn - 1) / tn);
        a.blocks_per_replica = (int32_t)(st.Bt * a.tiles_m * a.tiles_n);
        if (a.blocks_per_replica > kMaxPartials) {             // (halved tiles may exceed the slots: collapse them)
          do_collapse = true;
          collapse_blocks = a.blocks_per_replica;
          a.partC = E->d_scratch;
          a.partC_stride = collapse_blocks;
        } else {
          a.partC_stride = part_stride;                        // == blocks_per_replica (ctn_exec_create used the same rule)
        }
        used_tile(tm, tn);
        launch_mfma(st.modeA, st.modeB, tm, tn, dim3((unsigned)((int64_t)a.blocks_per_replica * R)), E->stream, a, E->sw);
        break;
      }
      case CTN_KERNEL_MFMA_F64: {
        const int64_t total = (int64_t)st.blocks * R;
        if (total >= (1LL << 31)) { g_err = "grid too large"; return CTN_UNSUPPORTED; }
        if (const int T = lat_form(st, R, E->n_cu, P.dtype, E->sw)) {      // one-launch latency form, 16 x 16 tiles
          a.tiles_m = (int32_t)((st.M + T - 1) / T);
          a.tiles_n = (int32_t)((st.N + T - 1) / T);
          a.blocks_per_replica = (int32_t)(st.Bt * a.tiles_m * a.tiles_n);
          a.partC = part_dst; a.partC_stride = part_stride; reduced = true;
          const int kchunk = (int)(((st.K + 7) / 8 + 3) / 4 * 4);
          used_tile(T, T);
          hipLaunchKernelGGL((k_mfma_lat<16, double>), dim3((unsigned)((int64_t)a.blocks_per_replica * R)), dim3(512), 0,
                             E->stream, a, kchunk);
          break;
        }
        if (const int S = E->d_slab ? splitk_splits(st, R, E->n_cu, P.dtype, E->sw) : 0) {   // latency mode, as in fp32
          SplitKArgs sk;
          sk.slab = E->d_slab;
          sk.numelC = P.tensors[st.out].numel;
          sk.kchunk = (int32_t)(((st.K + S - 1) / S + 31) / 32 * 32);
          sk.S = (int32_t)((st.K + sk.kchunk - 1) / sk.kchunk);  // drop empty trailing splits
          sk.tiles_m = (int32_t)((st.M + 63) / 64);
          sk.tiles_n = (int32_t)((st.N + 63) / 64);
          sk.tiles_per_replica = (int32_t)(st.Bt * sk.tiles_m * sk.tiles_n);
          used_tile(64, 64);
          launch_sk64(st.modeA, st.modeB, dim3((unsigned)((int64_t)sk.tiles_per_replica * sk.S * R)), E->stream, a, sk);
          launch_splitk_reduce<double>(E, E->step_partials[s], R, a, sk);
          break;
        }
        a.tiles_m = (int32_t)((st.M + kTile64M - 1) / kTile64M);
        if (E->mfma_g && st.tileN == DN) {  // 128 x 128 LDS-DMA kernel, under the same launch-size rule as fp32
          const int64_t gtiles = st.Bt * a.tiles_m * ((st.N + DN - 1) / DN) * R;
          if (E->mfma_g >= 2 || gtiles >= 2LL * E->n_cu) {
            used_tile(128, 128);
            a.tiles_n = (int32_t)((st.N + DN - 1) / DN);
            a.blocks_per_replica = (int32_t)(st.Bt * a.tiles_m * a.tiles_n);
            const dim3 gg((unsigned)gtiles);
            if (st.modeA == 2 && st.modeB == 2) hipLaunchKernelGGL((k_mfma_f64_g<2, 2>), gg, dim3(256), 0, E->stream, a);
            else if (st.modeA == 2) hipLaunchKernelGGL((k_mfma_f64_g<2, 1>), gg, dim3(256), 0, E->stream, a);
            else if (st.modeB == 2) hipLaunchKernelGGL((k_mfma_f64_g<1, 2>), gg, dim3(256), 0, E->stream, a);
            else hipLaunchKernelGGL((k_mfma_f64_g<1, 1>), gg, dim3(256), 0, E->stream, a);
            break;
          }
        }
        used_tile(128, 64);
        a.tiles_n = (int32_t)((st.N + kTile64N - 1) / kTile64N);
        const dim3 g((unsigned)total), b(256);
#define CTN_F64(AA, BB) hipLaunchKernelGGL((k_mfma_f64<AA, BB>), g, b, 0, E->stream, a)
        switch (st.modeA * 3 + st.modeB) {
          case 0: CTN_F64(0, 0); break; case 1: CTN_F64(0, 1); break; case 2: CTN_F64(0, 2); break;
          case 3: CTN_F64(1, 0); break; case 4: CTN_F64(1, 1); break; case 5: CTN_F64(1, 2); break;
          case 6: CTN_F64(2, 0); break; case 7: CTN_F64(2, 1); break; default: CTN_F64(2, 2); break;
        }
#undef CTN_F64
        break;
      }
      case CTN_KERNEL_DOT:
        if (const int S0 = E->d_slab ? dot_splits(st) : 0) {
          SplitKArgs sk{};
          sk.slab = E->d_slab;
          sk.numelC = P.tensors[st.out].numel;
          sk.kchunk = (int32_t)(((st.K + S0 - 1) / S0 + 255) / 256 * 256);
          sk.S = (int32_t)((st.K + sk.kchunk - 1) / sk.kchunk);
          const dim3 g((unsigned)((int64_t)st.blocks * sk.S), R);
          const Exec::DotTr dt = (int)E->dot_tr.size() == P.n_steps ? E->dot_tr[s] : Exec::DotTr();
          if (P.dtype == CTN_F32) {
            if (dt.on) hipLaunchKernelGGL(k_dot_tr<float>, g, dim3(256), 0, E->stream, a, (float*)sk.slab, sk.numelC, sk.S, dt.Ka, dt.Kb, dt.ldY, dt.x_is_a);
            else hipLaunchKernelGGL(k_dot_split<float>, g, dim3(256), 0, E->stream, a, (float*)sk.slab, sk.numelC, sk.S, sk.kchunk);
            launch_splitk_reduce<float>(E, E->step_partials[s], R, a, sk);
          } else {
            if (dt.on) hipLaunchKernelGGL(k_dot_tr<double>, g, dim3(256), 0, E->stream, a, (double*)sk.slab, sk.numelC, sk.S, dt.Ka, dt.Kb, dt.ldY, dt.x_is_a);
            else hipLaunchKernelGGL(k_dot_split<double>, g, dim3(256), 0, E->stream, a, (double*)sk.slab, sk.numelC, sk.S, sk.kchunk);
            launch_splitk_reduce<double>(E, E->step_partials[s], R, a, sk);
          }
          break;
        }
        if (P.dtype == CTN_F32) hipLaunchKernelGGL(k_dot<float>, dim3(st.blocks, R), dim3(256), 0, E->stream, a);
        else hipLaunchKernelGGL(k_dot<double>, dim3(st.blocks, R), dim3(256), 0, E->stream, a);
        break;
      case CTN_KERNEL_ROWDOT: {
        const int ks = E->d_slab ? rowdot_splits(st, R, E->n_cu) : 0;
        SplitKArgs sk{};
        if (ks) {
          sk.slab = E->d_slab;
          sk.numelC = P.tensors[st.out].numel;
          sk.kchunk = (int32_t)(((st.K + ks - 1) / ks + 63) / 64 * 64);
          sk.S = (int32_t)((st.K + sk.kchunk - 1) / sk.kchunk);
          a.ks_slab = sk.slab; a.ks_numelC = sk.numelC; a.ks_S = sk.S; a.ks_chunk = sk.kchunk;
        }
        const dim3 g(st.blocks, R, ks ? sk.S : 1);
        if (P.dtype == CTN_F32) hipLaunchKernelGGL(k_rowdot<float>, g, dim3(256), 0, E->stream, a);
        else hipLaunchKernelGGL(k_rowdot<double>, g, dim3(256), 0, E->stream, a);
        if (ks) {
          a.partC = part_dst; a.partC_stride = part_stride; reduced = true;
          if (P.dtype == CTN_F32) launch_splitk_reduce<float>(E, E->step_partials[s], R, a, sk);
          else launch_splitk_reduce<double>(E, E->step_partials[s], R, a, sk);
        }
        break;
      }
      default: {
        if (st.kvec) {   // short unit-stride K of the left operand: one output per thread, 16-byte loads along k
          a.dNv = make_fastdiv(st.Nv);
          if (P.dtype == CTN_F32) hipLaunchKernelGGL(k_stream_kvec<float>, dim3(st.blocks, R), dim3(256), 0, E->stream, a);
          else hipLaunchKernelGGL(k_stream_kvec<double>, dim3(st.blocks, R), dim3(256), 0, E->stream, a);
          break;
        }
        // vector stores need a 16-byte aligned destination: the caller's final buffer may not be
        const int vw = (s + 1 == P.n_steps && !E->outs_aligned16) ? 1 : st.vecw;
        int u;
        int64_t nq;
        stream_variant(st, R, vw, &u, &nq);
        a.dNq = make_fastdiv(nq / u);
        if (collect_left > 0) {   // a grouped leaf step: keep the arguments; the last one uploads and launches the group
          Exec::LeafGroup& G = E->groups[collect_head];
          E->h_group_args[G.off + (s - collect_head)] = a;
          if (--collect_left == 0) {
            HIPCHECK(hipMemcpyAsync(E->d_group_args + G.off, E->h_group_args + G.off, sizeof(StepArgs) * G.len,
                                    hipMemcpyHostToDevice, E->stream));
            G.ready = true;
            const dim3 gg(G.blocks, R, G.len);
            if (P.dtype == CTN_F32) launch_stream_group<float>(G.vw, G.u, gg, E->stream, E->d_group_args + G.off);
            else launch_stream_group<double>(G.vw, G.u, gg, E->stream, E->d_group_args + G.off);
          }
          break;
        }
        const int ks = E->d_slab ? stream_splits(st, R, E->n_cu) : 0;
        SplitKArgs sk{};
        if (ks) {
          sk.slab = E->d_slab;
          sk.numelC = P.tensors[st.out].numel;
          sk.kchunk = (int32_t)((st.K + ks - 1) / ks);
          sk.S = (int32_t)((st.K + sk.kchunk - 1) / sk.kchunk);
          a.ks_slab = sk.slab; a.ks_numelC = sk.numelC; a.ks_S = sk.S; a.ks_chunk = sk.kchunk;
        }
        const dim3 g(st.blocks, R, ks ? sk.S : 1), b(256);
#define CTN_STREAM(TT, VV, UU) hipLaunchKernelGGL((k_stream<TT, VV, UU>), g, b, 0, E->stream, a)
        if (P.dtype == CTN_F32) {
          if (vw == 4) { if (u == 4) CTN_STREAM(float, 4, 4); else CTN_STREAM(float, 4, 1); }
          else { if (u == 4) CTN_STREAM(float, 1, 4); else CTN_STREAM(float, 1, 1); }
        } else {
          if (vw == 2) { if (u == 4) CTN_STREAM(double, 2, 4); else CTN_STREAM(double, 2, 1); }
          else { if (u == 4) CTN_STREAM(double, 1, 4); else CTN_STREAM(double, 1, 1); }
        }
#undef CTN_STREAM
        if (ks) {
          a.partC = part_dst; a.partC_stride = part_stride; reduced = true;
          if (P.dtype == CTN_F32) launch_splitk_reduce<float>(E, E->step_partials[s], R, a, sk);
          else launch_splitk_reduce<double>(E, E->step_partials[s], R, a, sk);
        }
        break;
      }
    }
    if (do_collapse && !reduced)
      hipLaunchKernelGGL(k_collapse, dim3(R), dim3(256), 0, E->stream, (const double*)E->d_scratch, collapse_blocks, part_dst);
    if (E->eager_rescale && P.stabilize && s + 1 < P.n_steps) {   // the final tensor is normalised by k_finalize
      const int64_t numel = P.tensors[st.out].numel;
      const int V = P.dtype == CTN_F64 ? 2 : 4;
      const dim3 g((unsigned)std::max<int64_t>(1, std::min<int64_t>((numel / V + 255) / 256, 2048)), R);
      if (P.dtype == CTN_F32)
        hipLaunchKernelGGL(k_renorm<float>, g, dim3(256), 0, E->stream, (void* const*)E->d_ptrs, E->n_tensors, st.out, numel,
                           (const double*)part_dst, E->step_partials[s], part_stride, P.min_norm);
      else
        hipLaunchKernelGGL(k_renorm<double>, g, dim3(256), 0, E->stream, (void* const*)E->d_ptrs, E->n_tensors, st.out, numel,
                           (const double*)part_dst, E->step_partials[s], part_stride, P.min_norm);
    }
    if (timed) HIPCHECK(hipEventRecord(E->events[ev0 + 1], E->stream));
  }
  FinalArgs f;
  f.ptrs = E->d_ptrs;
  f.partials = E->d_partials;
  f.stepOff = E->d_stepOff;
  f.stepSlots = E->d_stepSlots;
  f.stepP = E->d_stepP;
  f.stepNumel = E->d_stepNumel;
  f.log_scale = E->d_log;
  f.rescales = E->d_resc;
  f.min_norm = P.min_norm;
  f.out_numel = P.output().numel;
  f.n_steps = P.n_steps; f.R = R; f.id_out = P.n_inputs + P.n_steps - 1; f.n_tensors = E->n_tensors;
  f.stabilize = P.stabilize ? 1 : 0;
  f.defer = E->defer_finish ? 1 : 0;
  f.vec = E->outs_aligned16 ? 1 : 0;
  int fb = (int)std::min<int64_t>((P.output().numel / (f.vec ? (P.dtype == CTN_F64 ? 2 : 4) : 1) + 255) / 256, 4096);
  if (fb < 1) fb = 1;
  const dim3 sg((P.n_steps + 3) / 4, R);
  if (P.dtype == CTN_F32) {
    hipLaunchKernelGGL(k_scales<float>, sg, dim3(256), 0, E->stream, f, E->d_logs);
    hipLaunchKernelGGL(k_finalize<float>, dim3(fb, R), dim3(256), 0, E->stream, f, (const double*)E->d_logs);
  } else {
    hipLaunchKernelGGL(k_scales<double>, sg, dim3(256), 0, E->stream, f, E->d_logs);
    hipLaunchKernelGGL(k_finalize<double>, dim3(fb, R), dim3(256), 0, E->stream, f, (const double*)E->d_logs);
  }
  HIPCHECK(hipGetLastError());
  if (E->timing_runs < E->timing_slots) E->timing_runs++;
  return CTN_OK;
}

// One enqueue of the whole path.  A path is hundreds of launches whose arguments never change (operands are
// reached through the device pointer table), and with one small or mid-size network in flight the host's
// launch rate, not the device, bounds the walk (100-site MPS, D = 32 ... 256: 1.3 - 2.3 ms of enqueue for
// 1.6 - 2.6 ms of wall time): from the third enqueue on the sequence is replayed as ONE hipGraph launch.
// Event-timed enqueues, the single-launch chain walk, debug stamping and streams that are themselves being
// captured stay eager.  CTN_GRAPH=0 disables.
static int exec_launch_all(Exec* E) {
  const Plan& P = *E->plan;
  const bool timed = E->timing_runs < E->timing_slots;
  const bool chain = P.chain && E->d_chain != nullptr;
  if (!E->use_graph || timed || chain || P.n_steps < 4 || E->eager_rescale || E->sw.stamps) return exec_launch_steps(E);
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(E->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return exec_launch_steps(E);
  }
  if (E->graph_exec && E->graph_aligned != E->outs_aligned16) {   // vector width of the last step changed
    (void)hipGraphExecDestroy(E->graph_exec);
    E->graph_exec = nullptr;
  }
  if (!E->graph_exec) {
    if (!E->graph_warm) { E->graph_warm = true; return exec_launch_steps(E); }
    hipGraph_t g = nullptr;
    if (hipStreamBeginCapture(E->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      (void)hipGetLastError();
      E->use_graph = 0;
      return exec_launch_steps(E);
    }
    const int rc = exec_launch_steps(E);
    const hipError_t ec = hipStreamEndCapture(E->stream, &g);
    if (rc != CTN_OK || ec != hipSuccess || !g ||
        hipGraphInstantiate(&E->graph_exec, g, nullptr, nullptr, 0) != hipSuccess) {
      (void)hipGetLastError();
      if (g) (void)hipGraphDestroy(g);
      E->graph_exec = nullptr;
      E->use_graph = 0;                      // fall back to eager launches for good
      return rc != CTN_OK ? rc : exec_launch_steps(E);
    }
    (void)hipGraphDestroy(g);
    E->graph_aligned = E->outs_aligned16;
  }
  HIPCHECK(hipGraphLaunch(E->graph_exec, E->stream));
  return CTN_OK;
}

static int exec_set_pointers(Exec* E, const void* const* dev_inputs, void* const* dev_outs) {
  const Plan& P = *E->plan;
  const int nt = E->n_tensors;
  bool changed = !E->ptrs_valid;
  for (int r = 0; r < E->R; ++r) {
    void** row = E->h_ptrs.data() + (size_t)r * nt;
    for (int i = 0; i < P.n_inputs; ++i) {
      void* p = const_cast<void*>(dev_inputs[(size_t)r * P.n_inputs + i]);
      if (!p) { g_err = "null operand pointer"; return CTN_INVALID_ARG; }
      if ((uintptr_t)p % 16) { g_err = "operand pointers must be 16-byte aligned"; return CTN_INVALID_ARG; }
      if (row[i] != p) { row[i] = p; changed = true; }
    }
    void* o = dev_outs[r];
    if (!o) { g_err = "null output pointer"; return CTN_INVALID_ARG; }
    if ((uintptr_t)o % P.elem_size()) { g_err = "output pointers must be element aligned"; return CTN_INVALID_ARG; }
    if (r == 0) E->outs_aligned16 = true;
    if ((uintptr_t)o % 16) E->outs_aligned16 = false;
    const int id_out = P.n_inputs + P.n_steps - 1;
    if (row[id_out] != o) { row[id_out] = o; changed = true; }
  }
  if (changed) {
    HIPCHECK(hipMemcpyAsync(E->d_ptrs, E->h_ptrs.data(), E->h_ptrs.size() * sizeof(void*),
                            hipMemcpyHostToDevice, E->stream));
    E->ptrs_valid = true;
  }
  return CTN_OK;
}

// A new set of operands starts in the caller's mode: an executor that switched itself to eager rescaling for one
// extreme input (exec_fetch_checked) tries the lazy form - hipGraph replay, grouped leaf steps, no k_renorm passes -
// again on the next one, and after kEagerSticky such switches in a row it stays eager (a caller who keeps sending
// extreme operands should not pay two contractions each time).
constexpr int kEagerSticky = 3;
static void exec_new_run(Exec* E) {
  if (E->eager_forced) { E->eager_rescale = true; return; }
  if (E->eager_rescale && E->eager_streak < kEagerSticky) E->eager_rescale = false;
}

}  // namespace ctn

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
using namespace ctn;

struct ctn_plan { Plan p; };
struct ctn_exec { Exec e; };

extern "C" {

int ctn_version(void) { return CTN_ABI_VERSION; }

const char* ctn_last_error(void) { return g_err.c_str(); }

int ctn_device_count(int* count) {
  if (!count) { g_err = "count is NULL"; return CTN_INVALID_ARG; }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    *count = 0;
    g_err = std::string("no HIP device available: ") + hipGetErrorString(e);
    (void)hipGetLastError();
    return CTN_NO_DEVICE;
  }
  *count = n;
  return CTN_OK;
}

int ctn_plan_create(const ctn_plan_desc* desc, ctn_plan** out) {
  if (!desc || !out) { g_err = "NULL argument"; return CTN_INVALID_ARG; }
  *out = nullptr;
  ctn_plan* p = new (std::nothrow) ctn_plan();
  if (!p) { g_err = "out of host memory"; return CTN_OOM; }
  std::string err;
  int rc = CTN_OK;
  try {
    rc = build_plan(*desc, p->p, err);
  } catch (const std::bad_alloc&) {
    rc = CTN_OOM; err = "out of host memory while building the plan";
  } catch (const std::exception& ex) {
    rc = CTN_INVALID_ARG; err = ex.what();
  }
  if (rc != CTN_OK) { g_err = err; delete p; return rc; }
  *out = p;
  return CTN_OK;
}

void ctn_plan_destroy(ctn_plan* plan) { delete plan; }
int ctn_plan_dtype(const ctn_plan* plan) { return plan ? plan->p.dtype : CTN_INVALID_ARG; }
int ctn_plan_n_inputs(const ctn_plan* plan) { return plan ? plan->p.n_inputs : CTN_INVALID_ARG; }
int ctn_plan_n_steps(const ctn_plan* plan) { return plan ? plan->p.n_steps : CTN_INVALID_ARG; }
double ctn_plan_flops(const ctn_plan* plan) { return plan ? plan->p.flops : 0.0; }
int64_t ctn_plan_bytes_min(const ctn_plan* plan) { return plan ? plan->p.bytes_min : 0; }
int ctn_plan_out_ndim(const ctn_plan* plan) { return plan ? (int)plan->p.output().dims.size() : CTN_INVALID_ARG; }
int ctn_plan_out_dims(const ctn_plan* plan, int64_t* dims) {
  if (!plan || !dims) { g_err = "NULL argument"; return CTN_INVALID_ARG; }
  const auto& d = plan->p.output().dims;
  for (size_t i = 0; i < d.size(); ++i) dims[i] = d[i];
  return CTN_OK;
}
int ctn_plan_out_labels(const ctn_plan* plan, int32_t* labels) {
  if (!plan || !labels) { g_err = "NULL argument"; return CTN_INVALID_ARG; }
  const auto& l = plan->p.output().labels;
  for (size_t i = 0; i < l.size(); ++i) labels[i] = l[i];
  return CTN_OK;
}
int64_t ctn_plan_out_numel(const ctn_plan* plan) { return plan ? plan->p.output().numel : 0; }
int64_t ctn_plan_out_bytes(const ctn_plan* plan) {
  return plan ? plan->p.output().numel * (int64_t)plan->p.elem_size() : 0;
}

static int64_t exec_fixed_bytes(const Plan& P, int R) {
  const int nt = P.n_inputs + P.n_steps + 1;
  int64_t slots = 0;
  for (const Step& st : P.steps) slots += std::max(st.partials, kWaveOutputs);   // (a launcher may retile: upper bound)
  return (int64_t)P.tables.size() * 4 + (int64_t)P.tables64.size() * 8 + (int64_t)R * nt * 8 + slots * R * 8 +
         (int64_t)R * std::max<int64_t>(P.max_collapse_blocks, 1) * 8 + (int64_t)R * 8 +
         (int64_t)R * P.n_steps * 8 + 256 + (int64_t)P.n_steps * 12;
}

int64_t ctn_plan_workspace_bytes(const ctn_plan* plan, int replicas) {
  if (!plan || replicas < 1) return 0;
  return plan->p.ws_bytes_per_replica * replicas + exec_fixed_bytes(plan->p, replicas);
}

int ctn_plan_step_info(const ctn_plan* plan, int step, ctn_step_info* info) {
  if (!plan || !info || step < 0 || step >= plan->p.n_steps) { g_err = "invalid step query"; return CTN_INVALID_ARG; }
  const Step& s = plan->p.steps[step];
  info->kernel = s.kernel;
  info->swapped = s.swapped ? 1 : 0;
  info->batch = s.Bt; info->m = s.M; info->n = s.N; info->k = s.K;
  info->mode_a = s.modeA; info->mode_b = s.modeB;
  info->partials = s.partials; info->blocks = s.blocks;
  info->flops = s.flops;
  info->out_numel = plan->p.tensors[s.out].numel;
  info->tile_m = s.kernel == CTN_KERNEL_MFMA_F32 ? s.tileM : (s.kernel == CTN_KERNEL_MFMA_F64 ? kTile64M : 0);
  info->tile_n = (s.kernel == CTN_KERNEL_MFMA_F32 || s.kernel == CTN_KERNEL_MFMA_F64) ? s.tileN : 0;
  info->epilogue_sum = s.epw;
  info->reserved = 0;
  return CTN_OK;
}

int ctn_exec_create(const ctn_plan* plan, int device, void* stream, int replicas, ctn_exec** out) {
  if (!plan || !out || replicas < 1) { g_err = "invalid argument to ctn_exec_create"; return CTN_INVALID_ARG; }
  *out = nullptr;
  int ndev = 0;
  int rc = ctn_device_count(&ndev);
  if (rc != CTN_OK) return rc;
  if (device < 0 || device >= ndev) { g_err = "device index out of range"; return CTN_INVALID_ARG; }
  DeviceGuard dg(device);
  HIPCHECK(dg.err);
  int n_cu = 256;
  (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device);
  ctn_exec* x = new (std::nothrow) ctn_exec();
  if (!x) { g_err = "out of host memory"; return CTN_OOM; }
  Exec& E = x->e;
  const Plan& P = plan->p;
  E.plan = &P; E.device = device; E.R = replicas; E.n_cu = n_cu > 0 ? n_cu : 256;
  E.sw = read_dev_switches();
  E.mfma_g = E.sw.mfma_g;
  E.use_graph = E.sw.graph;
  E.n_tensors = P.n_inputs + P.n_steps + 1;
  auto fail = [&](int code) { delete x; return code; };
#define HIPCHECK_X(expr)                                                        \
  do {                                                                          \
    hipError_t e_ = (expr);                                                     \
    if (e_ != hipSuccess) {                                                     \
      g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                \
      return fail(e_ == hipErrorOutOfMemory ? CTN_OOM : CTN_HIP_ERROR);         \
    }                                                                           \
  } while (0)
  if (stream) { E.stream = (hipStream_t)stream; }
  else { HIPCHECK_X(hipStreamCreateWithFlags(&E.stream, hipStreamNonBlocking)); E.own_stream = true; }
  const size_t ws = (size_t)std::max<int64_t>(P.ws_bytes_per_replica, 256) * replicas;
  HIPCHECK_X(hipMalloc((void**)&E.d_ws, ws));
  HIPCHECK_X(hipMalloc((void**)&E.d_tables, std::max<size_t>(P.tables.size(), 4) * 4));
  HIPCHECK_X(hipMemcpy(E.d_tables, P.tables.data(), P.tables.size() * 4, hipMemcpyHostToDevice));
  HIPCHECK_X(hipMalloc((void**)&E.d_tables64, std::max<size_t>(P.tables64.size(), 2) * 8));
  HIPCHECK_X(hipMemcpy(E.d_tables64, P.tables64.data(), P.tables64.size() * 8, hipMemcpyHostToDevice));
  HIPCHECK_X(hipMalloc((void**)&E.d_ptrs, (size_t)replicas * E.n_tensors * sizeof(void*)));
  {
    size_t slab_elems = 0;   // split-K scratch: S slabs shaped like the step's output, per replica
    auto splits_of = [&](const Step& st) {
      if (P.dtype == CTN_F32)
        if (const int S = g_splitk(st, replicas, E.n_cu, E.mfma_g, E.sw, st.cvec)) return S;
      if (lat_form(st, replicas, E.n_cu, P.dtype, E.sw)) return 0;
      if (const int S = splitk_splits(st, replicas, E.n_cu, P.dtype, E.sw)) return S;
      if (const int S = dot_splits(st)) return S;
      if (const int S = stream_splits(st, replicas, E.n_cu)) return S;
      return rowdot_splits(st, replicas, E.n_cu);
    };
    for (const Step& st : P.steps)
      if (const int S = splits_of(st))
        slab_elems = std::max(slab_elems, (size_t)S * (size_t)P.tensors[st.out].numel * (size_t)replicas);
    if (slab_elems) HIPCHECK_X(hipMalloc((void**)&E.d_slab, slab_elems * (P.dtype == CTN_F64 ? 8 : 4)));
    size_t fold_elems = 0;   // a step with more than 16 slabs folds them 16 to 1 into this buffer and back
    for (const Step& st : P.steps)
      if (const int S = splits_of(st))
        if (S > 16)
          fold_elems = std::max(fold_elems, (size_t)((S + 15) / 16) * (size_t)P.tensors[st.out].numel * (size_t)replicas);
    if (fold_elems) HIPCHECK_X(hipMalloc((void**)&E.d_slab2, fold_elems * (P.dtype == CTN_F64 ? 8 : 4)));
  }
  // partial counts: plan value, one per tile in the latency form, and the split-K reduce pass spreads over up to
  // 64 workgroups per replica; every step gets a region of exactly that many slots per replica
  E.step_partials.resize(P.n_steps);
  E.step_off.resize(P.n_steps);
  int64_t scratch_need = std::max<int64_t>(P.max_collapse_blocks, 1);
  for (int s = 0; s < P.n_steps; ++s) {
    const Step& st = P.steps[s];
    E.step_partials[s] = std::max(st.partials, 1);
    if (P.chain) { E.step_off[s] = E.part_slots++; continue; }   // the chain walker: one slot per step, whatever the kernel kind
    if (E.d_slab && P.dtype == CTN_F32 && s + 1 < P.n_steps && g_splitk(st, replicas, E.n_cu, E.mfma_g, E.sw, st.cvec))
      E.step_partials[s] = (int)std::max<int64_t>(1, std::min<int64_t>(kWaveOutputs, P.tensors[st.out].numel / 1024));
    else if (const int T = lat_form(st, replicas, E.n_cu, P.dtype, E.sw))
      E.step_partials[s] = (int)(st.Bt * ((st.M + T - 1) / T) * ((st.N + T - 1) / T));
    else if (E.d_slab && (splitk_splits(st, replicas, E.n_cu, P.dtype, E.sw) || dot_splits(st) || stream_splits(st, replicas, E.n_cu) ||
                          rowdot_splits(st, replicas, E.n_cu)))
      E.step_partials[s] = (int)std::max<int64_t>(1, std::min<int64_t>(kWaveOutputs, P.tensors[st.out].numel / 1024));
    else if (!g_launch(st, replicas, E.n_cu, E.mfma_g) && s + 1 < P.n_steps &&
             h_form(P, st, replicas, E.n_cu, P.dtype, E.sw, st.cvec))
      E.step_partials[s] = (int)(st.Bt * ((st.M + 127) / 128) * ((st.N + 127) / 128));
    else if (st.kernel == CTN_KERNEL_MFMA_F32 && !st.collapse) {
      int tm, tn;
      plain_tiles(st, replicas, E.n_cu, E.mfma_g, s + 1 == P.n_steps, &tm, &tn, E.sw.halve);
      const int64_t tiles = st.Bt * ((st.M + tm - 1) / tm) * ((st.N + tn - 1) / tn);
      E.step_partials[s] = tiles > kMaxPartials ? 1 : (int)tiles;     // halved tiles beyond the slots: collapsed at launch
      scratch_need = std::max<int64_t>(scratch_need, tiles);
    }
    E.step_off[s] = E.part_slots;
    E.part_slots += E.step_partials[s];
  }
  // zipper pairs: at least one full round of workgroups (each is 134 MFLOP long), or CTN_ZIP=1
  if (!P.chain && E.sw.zip != 0) {
    E.zip.assign(P.n_steps, Exec::ZipDesc());
    E.zip_skip.assign(P.n_steps, 0);
    bool any = false;
    for (int s = 1; s + 1 < P.n_steps; ++s) {
      Exec::ZipDesc z;
      if (E.zip_skip[s - 1] || (s >= 2 && E.zip[s - 1].on)) continue;
      // 128 values of u per workgroup (k_zip_f32) when that fills the chip - or CTN_ZIP=1; else 64 (k_zip64_f32) when THAT
      // does: 64 ... 127 networks of |u| = 256 - or CTN_ZIP=2; fewer networks keep the two-launch / latency forms
      // ... and of the two the one whose workgroups fill their rounds better (one workgroup per CU and round: 96 networks
      // are 192 workgroups of 128 - refused - or 384 of 64 - a round and a half: 26.0 ms against 23.0 on the two-launch
      // forms - while 64 networks are exactly one round of 64: 13.5 ms against 15.9)
      auto fill = [&](int64_t wgs) { return (double)wgs / (double)(((wgs + E.n_cu - 1) / E.n_cu) * E.n_cu); };
      Exec::ZipDesc z128, z64;
      const bool m128 = E.sw.zip != 2 && zip_match(P, s, &z128) && (E.sw.zip == 1 || (int64_t)(z128.U / ZU) * replicas >= E.n_cu);
      const bool m64 = E.sw.zip != 1 && zip_match(P, s, &z64, Z6U) && z64.K1 % Z6K == 0 &&
                       (E.sw.zip == 2 || (int64_t)(z64.U / Z6U) * replicas >= E.n_cu);
      const double f128 = m128 ? fill((int64_t)(z128.U / ZU) * replicas) : 0.0, f64 = m64 ? fill((int64_t)(z64.U / Z6U) * replicas) : 0.0;
      bool ok = false;
      if (m128 && (E.sw.zip == 1 || f128 >= 0.9 || f128 >= f64)) { ok = true; z = z128; z.zu = ZU; }
      else if (m64 && (E.sw.zip == 2 || f64 >= 0.9)) { ok = true; z = z64; z.zu = Z6U; }
      else if (m128) { ok = true; z = z128; z.zu = ZU; }
      if (!ok || z.U / z.zu > kMaxPartials) continue;    // (one abs-sum partial per workgroup: the consumers add at most that many)
      E.zip[s] = z;
      E.zip_skip[s - 1] = 1;
      any = true;
      // one abs-sum partial per workgroup of the fused launch; the skipped step keeps its (never written, zero) slots
      E.part_slots -= E.step_partials[s];
      E.step_partials[s] = z.U / z.zu;
      E.part_slots += E.step_partials[s];
    }
    if (any) {   // regions moved: lay the offsets out again
      E.part_slots = 0;
      for (int s = 0; s < P.n_steps; ++s) { E.step_off[s] = E.part_slots; E.part_slots += E.step_partials[s]; }
    } else {
      E.zip.clear(); E.zip_skip.clear();
    }
  }
  // the same pairs in their latency form: a few networks in flight (or CTN_ZIPL=1), and no throughput-form pair taken
  if (!P.chain && P.dtype == CTN_F32 && E.zip.empty() && E.sw.zipl != 0 && (E.sw.zipl == 1 || replicas <= E.sw.zipl_max_r)) {
    E.zl.assign(P.n_steps, Exec::ZipLat());
    E.zl_skip.assign(P.n_steps, 0);
    bool any = false;
    int64_t slab_elems = 0;
    for (int s = 1; s + 1 < P.n_steps; ++s) {
      Exec::ZipDesc z;
      if (E.zl_skip[s - 1] || (s >= 2 && E.zl[s - 1].on) || !zip_match(P, s, &z, 16)) continue;
      if (z.K1 != ZM || (z.Q != 4 && z.Q != 2) || z.ldC % 4 || z.ldE % 4 || z.ldXk % 4 || z.ldXq % 4 || z.ldYm % 4 || z.ldYq % 4) continue;
      // the part of m1 a workgroup owns: 32 (8 slabs, 128 workgroups per network at |u| = 256) while that still fits ONE
      // round of workgroups, else 64 (4 slabs, half the workgroups, each with twice the work per byte it loads)
      int mp = 64;
      if (z.Q == 4 && (E.sw.zipl_mp == 32 || (E.sw.zipl_mp == 0 && (int64_t)replicas * (z.U / 16) * 8 <= E.n_cu))) mp = 32;
      const int S = ZM / mp;
      if ((z.U / 16) * S > kMaxPartials) continue;
      E.zl[s].on = true; E.zl[s].d = z; E.zl[s].mp = mp;
      E.zl_skip[s - 1] = 1;
      any = true;
      slab_elems = std::max<int64_t>(slab_elems, (int64_t)S * z.U * ZM);
      E.step_partials[s] = (z.U / 16) * S;        // one abs-sum partial per workgroup (of its slab piece)
    }
    if (any) {
      for (int s = 3; s + 1 < P.n_steps; ++s) {   // links of the chain: the pair (s - 1, s) takes the result of the pair (s - 3, s - 2)
        if (!E.zl[s].on || !E.zl[s - 2].on || P.steps[s - 1].lhs != P.steps[s - 2].out) continue;
        if (E.zl[s - 2].d.U != ZM || E.zl[s - 2].mp != E.zl[s].mp || E.zl[s - 2].d.ldC != E.zl[s].d.ldE) continue;
        E.zl[s].from_prev = true;
        E.zl[s].buf = E.zl[s - 2].buf ^ 1;
        E.zl[s - 2].feeds_next = true;
      }
      E.part_slots = 0;
      for (int s = 0; s < P.n_steps; ++s) { E.step_off[s] = E.part_slots; E.part_slots += E.step_partials[s]; }
      for (int i = 0; i < 2; ++i) HIPCHECK_X(hipMalloc((void**)&E.d_zl_slab[i], (size_t)replicas * slab_elems * 4));
    } else {
      E.zl.clear(); E.zl_skip.clear();
    }
  }
  // full dots against a transposed tensor
  if (!P.chain && E.sw.dot_tr) {
    E.dot_tr.assign(P.n_steps, Exec::DotTr());
    bool any = false;
    for (int s = 0; s < P.n_steps; ++s) any = dot_tr_match(P, P.steps[s], &E.dot_tr[s]) || any;
    if (!any) E.dot_tr.clear();
  }
  // steps with a resident left operand (k_mfma_f32_ares)
  if (!P.chain && E.sw.ares != 0 && P.dtype == CTN_F32) {
    E.ares_ntw.assign(P.n_steps, 0);
    bool any = false;
    for (int s = 0; s < P.n_steps; ++s) { E.ares_ntw[s] = ares_match(P, s, replicas, E.n_cu, E.sw);
      if (E.ares_ntw[s] && ares_avec(P, s)) E.ares_ntw[s] |= 1 << 16;
      any = any || E.ares_ntw[s];
    }
    if (!any) E.ares_ntw.clear();
  }
  // a sweep: at least half a chip of row blocks, or CTN_SWEEP=1
  if (!P.chain && E.sw.sweep != 0 && P.stabilize) {
    Exec::SweepDesc sd;
    // (bonds up to 128: the per-site launches are a few microseconds of latency each whatever the batch - always)
    if (sweep_match(P, &sd) && (int)sd.steps.size() <= kSweepMaxSites &&
        (E.sw.sweep == 1 || (sd.steps.size() >= 4 && (sd.D <= 128 || (int64_t)sd.J * replicas * 2 >= E.n_cu)))) {
      E.sweep = sd;
      const int S = (int)sd.steps.size();
      E.sweep_role.assign(P.n_steps, 0);
      std::vector<int32_t> ids((size_t)S * 2), slots((size_t)S);
      std::vector<int64_t> offs((size_t)S);
      for (int i = 0; i < S; ++i) {
        const int s = sd.steps[i];
        E.sweep_role[s] = i + 1 == S ? 2 : 1;
        ids[2 * i] = P.steps[s].rhs; ids[2 * i + 1] = P.steps[s].lhs2;
        offs[i] = E.step_off[s]; slots[i] = E.step_partials[s];
      }
      HIPCHECK_X(hipMalloc((void**)&E.d_sweep_ids, ids.size() * 4));
      HIPCHECK_X(hipMemcpy(E.d_sweep_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
      HIPCHECK_X(hipMalloc((void**)&E.d_sweep_off, offs.size() * 8));
      HIPCHECK_X(hipMemcpy(E.d_sweep_off, offs.data(), offs.size() * 8, hipMemcpyHostToDevice));
      HIPCHECK_X(hipMalloc((void**)&E.d_sweep_slots, slots.size() * 4));
      HIPCHECK_X(hipMemcpy(E.d_sweep_slots, slots.data(), slots.size() * 4, hipMemcpyHostToDevice));
      const size_t nrec = (size_t)replicas * S * sd.J;
      HIPCHECK_X(hipMalloc((void**)&E.d_sweep_a, nrec * 8));
      HIPCHECK_X(hipMalloc((void**)&E.d_sweep_s, nrec * 4));
      HIPCHECK_X(hipMalloc((void**)&E.d_sweep_z, (size_t)replicas * S * 8));
      HIPCHECK_X(hipMalloc((void**)&E.d_sweep_la, nrec * 8));
      HIPCHECK_X(hipMalloc((void**)&E.d_sweep_ls, nrec * 8));
    }
  }
  // leaf groups: runs of consecutive plain streaming steps on network inputs, same kernel variant (see Exec::LeafGroup)
  if (!P.chain && E.sw.group) {
    E.groups.assign(P.n_steps, Exec::LeafGroup());
    auto leaf = [&](int s, int* vw, int* u) {
      const Step& st = P.steps[s];
      if (st.kernel != CTN_KERNEL_ELEMENT || st.kvec || st.collapse || s + 1 >= P.n_steps || st.rhs < 0 ||
          st.lhs >= P.n_inputs || st.rhs >= P.n_inputs || st.lhs2 >= 0 || stream_splits(st, replicas, E.n_cu))
        return false;
      int64_t nq;
      *vw = st.vecw;
      stream_variant(st, replicas, *vw, u, &nq);
      return true;
    };
    int total = 0;
    for (int s = 0; s < P.n_steps;) {
      int vw, u;
      if (!leaf(s, &vw, &u)) { ++s; continue; }
      int e = s + 1, blocks = P.steps[s].blocks, vw2, u2;
      while (e < P.n_steps && e - s < 1024 && leaf(e, &vw2, &u2) && vw2 == vw && u2 == u) { blocks = std::max(blocks, P.steps[e].blocks); ++e; }
      if (e - s >= 2) {
        Exec::LeafGroup& G = E.groups[s];
        G.len = e - s; G.vw = vw; G.u = u; G.blocks = blocks; G.off = total;
        total += G.len;
      }
      s = e;
    }
    if (total) {
      HIPCHECK_X(hipHostMalloc((void**)&E.h_group_args, sizeof(StepArgs) * (size_t)total, hipHostMallocDefault));
      HIPCHECK_X(hipMalloc((void**)&E.d_group_args, sizeof(StepArgs) * (size_t)total));
    } else {
      E.groups.clear();
    }
  }
  HIPCHECK_X(hipMalloc((void**)&E.d_scratch, (size_t)replicas * scratch_need * 8));
  HIPCHECK_X(hipMalloc((void**)&E.d_partials, (size_t)E.part_slots * replicas * 8));
  HIPCHECK_X(hipMemset(E.d_partials, 0, (size_t)E.part_slots * replicas * 8));
  {
    std::vector<int32_t> slots(E.step_partials.begin(), E.step_partials.end());
    HIPCHECK_X(hipMalloc((void**)&E.d_stepOff, P.n_steps * 8));
    HIPCHECK_X(hipMalloc((void**)&E.d_stepSlots, P.n_steps * 4));
    HIPCHECK_X(hipMemcpy(E.d_stepOff, E.step_off.data(), P.n_steps * 8, hipMemcpyHostToDevice));
    HIPCHECK_X(hipMemcpy(E.d_stepSlots, slots.data(), P.n_steps * 4, hipMemcpyHostToDevice));
  }
  HIPCHECK_X(hipMalloc((void**)&E.d_log, (size_t)replicas * 8));
  HIPCHECK_X(hipMalloc((void**)&E.d_resc, (size_t)replicas * P.n_steps * 8));
  HIPCHECK_X(hipMalloc((void**)&E.d_logs, (size_t)replicas * P.n_steps * 8));
  HIPCHECK_X(hipMalloc(&E.d_ones, 256));
  {
    double one64 = 1.0; float one32 = 1.0f;
    if (P.dtype == CTN_F64) HIPCHECK_X(hipMemcpy(E.d_ones, &one64, 8, hipMemcpyHostToDevice));
    else HIPCHECK_X(hipMemcpy(E.d_ones, &one32, 4, hipMemcpyHostToDevice));
  }
  std::vector<int32_t> sp(P.n_steps);
  std::vector<double> sn(P.n_steps);
  for (int s = 0; s < P.n_steps; ++s) {
    sp[s] = P.steps[s].kernel == CTN_KERNEL_FUSED ? 0 : E.step_partials[s];   // no partials: rescale reported as 0.0
    sn[s] = (double)P.tensors[P.steps[s].out].numel;
  }
  HIPCHECK_X(hipMalloc((void**)&E.d_stepP, P.n_steps * 4));
  HIPCHECK_X(hipMalloc((void**)&E.d_stepNumel, P.n_steps * 8));
  HIPCHECK_X(hipMemcpy(E.d_stepP, sp.data(), P.n_steps * 4, hipMemcpyHostToDevice));
  HIPCHECK_X(hipMemcpy(E.d_stepNumel, sn.data(), P.n_steps * 8, hipMemcpyHostToDevice));
  if (P.chain) {
    std::vector<ChainStep> cs(P.n_steps);
    for (int s = 0; s < P.n_steps; ++s) {
      const Step& st = P.steps[s];
      ChainStep& c = cs[s];
      const int32_t* T = E.d_tables;
      const int64_t* T8 = E.d_tables64;
      c.obA = T8 + st.t.obA; c.obB = T8 + st.t.obB; c.obC = T8 + st.t.obC;
      c.omA = T + st.t.omA; c.omC = T + st.t.omC; c.onB = T + st.t.onB; c.onC = T + st.t.onC;
      c.okA = T + st.t.okA; c.okB = T + st.t.okB;
      c.numelC = (double)P.tensors[st.out].numel;
      c.Bt = (int32_t)st.Bt; c.M = (int32_t)st.M; c.N = (int32_t)st.N; c.K = (int32_t)st.K;
      c.idA = st.lhs; c.idB = st.rhs >= 0 ? st.rhs : E.n_tensors - 1; c.idC = st.out;
      c.prodA = st.lhs >= P.n_inputs ? P.tensors[st.lhs].producer : -1;
      c.prodB = st.rhs >= P.n_inputs ? P.tensors[st.rhs].producer : -1;
    }
    HIPCHECK_X(hipMalloc((void**)&E.d_chain, cs.size() * sizeof(ChainStep)));
    HIPCHECK_X(hipMemcpy(E.d_chain, cs.data(), cs.size() * sizeof(ChainStep), hipMemcpyHostToDevice));
  }
  // pointer table: intermediates and the ones-scalar are fixed for the executor's lifetime
  E.h_ptrs.assign((size_t)replicas * E.n_tensors, nullptr);
  for (int r = 0; r < replicas; ++r) {
    void** row = E.h_ptrs.data() + (size_t)r * E.n_tensors;
    for (int id = P.n_inputs; id < P.n_inputs + P.n_steps - 1; ++id)
      row[id] = E.d_ws + (size_t)r * P.ws_bytes_per_replica + P.tensors[id].ws_offset;
    row[E.n_tensors - 1] = E.d_ones;
  }
#undef HIPCHECK_X
  *out = x;
  return CTN_OK;
}

void ctn_exec_destroy(ctn_exec* exec) { delete exec; }

int ctn_exec_enqueue(ctn_exec* exec, const void* const* dev_inputs, void* const* dev_outs) {
  if (!exec || !dev_inputs || !dev_outs) { g_err = "NULL argument"; return CTN_INVALID_ARG; }
  Exec* E = &exec->e;
  DeviceGuard dg(E->device);
  HIPCHECK(dg.err);
  int rc = exec_set_pointers(E, dev_inputs, dev_outs);
  if (rc != CTN_OK) return rc;
  exec_new_run(E);
  return exec_launch_all(E);
}

int ctn_exec_synchronize(ctn_exec* exec) {
  if (!exec) { g_err = "NULL argument"; return CTN_INVALID_ARG; }
  HIPCHECK(hipStreamSynchronize(exec->e.stream));
  if (const char* path = exec->e.sw.stamps) {  // development only: dump the last MFMA launch's stamps
    Exec* E = &exec->e;
    if (E->d_dbg && E->dbg_tiles) {
      std::vector<unsigned long long> h(E->dbg_tiles * 8);
      HIPCHECK(hipMemcpy(h.data(), E->d_dbg, h.size() * 8, hipMemcpyDeviceToHost));
      if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
  }
  return CTN_OK;
}

// Did a lazy epilogue rescale leave the dtype's range in the run whose scale registers are in E->h_resc?
// A tile kernel accumulates on the STORED operands, A_hat sA and B_hat sB, and multiplies by 1 / (sA sB)
// afterwards, where the reference normalises each intermediate before the next product (einsum.py:387): the
// accumulators therefore carry about s_out sA sB, which can overflow (or sink into the subnormals) although
// every normalised quantity is fine - operands of magnitude ~1e13 do it in fp32.  All three factors are in the
// scale registers the run has just produced, so the test costs nothing on the device: non-finite anywhere, an
// accumulator magnitude beyond 2^+-100 (fp32; 2^+-900 fp64), or an all-zero output behind operand scales that
// could have flushed it.
static bool exec_scales_suspect(const Exec* E, const double* resc = nullptr, int replicas = -1) {
  const Plan& P = *E->plan;
  if (!P.stabilize || (P.chain && E->d_chain)) return false;   // the chain walker divides operands on load
  const double hi = std::ldexp(1.0, P.dtype == CTN_F64 ? 900 : 100), lo = 1.0 / hi;
  const double zhi = std::ldexp(1.0, P.dtype == CTN_F64 ? 500 : 60), zlo = 1.0 / zhi;
  if (!resc) resc = E->h_resc.data();
  if (replicas < 0) replicas = E->R;
  for (int r = 0; r < replicas; ++r) {
    const double* rs = resc + (size_t)r * P.n_steps;
    for (int s = 0; s < P.n_steps; ++s) {
      const Step& st = P.steps[s];
      auto scale_of = [&](int id) {
        if (id < P.n_inputs) return 1.0;
        const double v = rs[P.tensors[id].producer];
        return v == 0.0 ? 1.0 : v;
      };
      if (st.kernel == CTN_KERNEL_FUSED) continue;
      if (!E->zip_skip.empty() && E->zip_skip[s]) continue;          // runs inside the next step's launch
      if (!E->zl_skip.empty() && E->zl_skip[s]) continue;
      if (!E->sweep_role.empty() && E->sweep_role[s] && !E->eager_rescale) {   // a sweep keeps its products in range by itself
        if (!std::isfinite(rs[s])) return true;
        continue;
      }
      double sab = scale_of(st.lhs) * (st.rhs >= 0 ? scale_of(st.rhs) : 1.0) * (st.lhs2 >= 0 ? scale_of(st.lhs2) : 1.0);
      if ((!E->zip.empty() && E->zip[s].on) || (!E->zl.empty() && E->zl[s].on))   // the fused pair accumulates on E, X and Y as stored
        sab = scale_of(P.steps[s - 1].lhs) * scale_of(P.steps[s - 1].rhs) * scale_of(st.rhs);
      const double so = rs[s];
      if (!std::isfinite(so) || !std::isfinite(sab)) return true;
      if (so == 0.0) { if (sab > zhi || sab < zlo) return true; continue; }
      const double acc = so * sab;
      if (!(acc < hi) || !(acc > lo)) return true;
    }
  }
  return false;
}

// Wait for the stream, read the scale registers, and - in the default lazy mode - repeat the contraction in
// eager mode when they show that a lazy product left the dtype's range.  The operands are still in place: they
// are borrowed until the fetch.
static int exec_fetch_checked(Exec* E, double* log_scale) {
  const Plan& P = *E->plan;
  E->h_resc.resize((size_t)E->R * P.n_steps);
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (log_scale)
      HIPCHECK(hipMemcpyAsync(log_scale, E->d_log, (size_t)E->R * 8, hipMemcpyDeviceToHost, E->stream));
    HIPCHECK(hipMemcpyAsync(E->h_resc.data(), E->d_resc, E->h_resc.size() * 8, hipMemcpyDeviceToHost, E->stream));
    HIPCHECK(hipStreamSynchronize(E->stream));
    if (E->eager_rescale || !E->ptrs_valid) break;
    if (!exec_scales_suspect(E)) { E->eager_streak = 0; break; }
    E->eager_rescale = true;
    E->eager_reruns++;
    E->eager_streak++;
    const int rc = exec_launch_all(E);
    if (rc != CTN_OK) return rc;
  }
  return CTN_OK;
}

int ctn_exec_fetch(ctn_exec* exec, double* log_scale, double* step_rescales) {
  if (!exec) { g_err = "NULL argument"; return CTN_INVALID_ARG; }
  Exec* E = &exec->e;
  DeviceGuard dg(E->device);
  HIPCHECK(dg.err);
  const int rc = exec_fetch_checked(E, log_scale);
  if (rc != CTN_OK) return rc;
  if (step_rescales) memcpy(step_rescales, E->h_resc.data(), E->h_resc.size() * 8);
  return CTN_OK;
}

int ctn_exec_set_rescale_mode(ctn_exec* exec, int mode) {
  if (!exec || mode < 0 || mode > 1) { g_err = "rescale mode must be 0 (lazy, eager on demand) or 1 (eager)"; return CTN_INVALID_ARG; }
  Exec* E = &exec->e;
  const int prev = E->eager_rescale ? 1 : 0;
  E->eager_forced = mode == 1;
  E->eager_rescale = mode == 1;
  return prev;
}

int ctn_exec_eager_reruns(const ctn_exec* exec) { return exec ? exec->e.eager_reruns : CTN_INVALID_ARG; }

int ctn_exec_set_finish_mode(ctn_exec* exec, int mode) {
  if (!exec || mode < 0 || mode > 1) { g_err = "finish mode must be 0 (normalise at the end of every run) or 1 (deferred to ctn_exec_finish)"; return CTN_INVALID_ARG; }
  Exec* E = &exec->e;
  const int prev = E->defer_finish ? 1 : 0;
  if (prev != mode && E->graph_exec) {             // the captured k_finalize carries the old mode
    DeviceGuard dg(E->device);
    HIPCHECK(dg.err);
    HIPCHECK(hipStreamSynchronize(E->stream));
    (void)hipGraphExecDestroy(E->graph_exec);
    E->graph_exec = nullptr;
  }
  E->defer_finish = mode == 1;
  return prev;
}

int ctn_exec_finish(ctn_exec* exec, const double* mult) {
  if (!exec || !mult) { g_err = "NULL argument"; return CTN_INVALID_ARG; }
  Exec* E = &exec->e;
  const Plan& P = *E->plan;
  if (!E->defer_finish || !E->ptrs_valid) { g_err = "ctn_exec_finish: no deferred run to finish (ctn_exec_set_finish_mode(1), then enqueue)"; return CTN_INVALID_ARG; }
  DeviceGuard dg(E->device);
  HIPCHECK(dg.err);
  if (!E->d_mult) HIPCHECK(hipMalloc((void**)&E->d_mult, (size_t)E->R * 8));
  HIPCHECK(hipMemcpyAsync(E->d_mult, mult, (size_t)E->R * 8, hipMemcpyHostToDevice, E->stream));
  const int vec = E->outs_aligned16 ? 1 : 0;
  const int64_t numel = P.output().numel;
  const int V = vec ? (P.dtype == CTN_F64 ? 2 : 4) : 1;
  const dim3 g((unsigned)std::max<int64_t>(1, std::min<int64_t>((numel / V + 255) / 256, 4096)), (unsigned)E->R);
  const int id_out = P.n_inputs + P.n_steps - 1;
  if (P.dtype == CTN_F32)
    hipLaunchKernelGGL(k_finish<float>, g, dim3(256), 0, E->stream, (void* const*)E->d_ptrs, E->n_tensors, id_out, numel,
                       (const double*)E->d_resc, P.n_steps, P.stabilize ? 1 : 0, (const double*)E->d_mult, vec);
  else
    hipLaunchKernelGGL(k_finish<double>, g, dim3(256), 0, E->stream, (void* const*)E->d_ptrs, E->n_tensors, id_out, numel,
                       (const double*)E->d_resc, P.n_steps, P.stabilize ? 1 : 0, (const double*)E->d_mult, vec);
  HIPCHECK(hipGetLastError());
  return CTN_OK;
}

int ctn_exec_run(ctn_exec* exec, const void* const* inputs, int inputs_space, void* const* outs,
                 int outs_space, double* log_scale, double* step_rescales) {
  if (!exec || !inputs || !outs) { g_err = "NULL argument"; return CTN_INVALID_ARG; }
  Exec* E = &exec->e;
  const Plan& P = *E->plan;
  DeviceGuard dg(E->device);
  HIPCHECK(dg.err);
  const size_t es = P.elem_size();
  const int64_t out_bytes = P.output().numel * (int64_t)es;
  const int64_t out_slot = (out_bytes + kAlign - 1) / kAlign * kAlign;
  std::vector<const void*> din((size_t)E->R * P.n_inputs);
  std::vector<void*> dout(E->R);
  if (inputs_space == CTN_MEM_HOST) {
    if (!E->d_stage_in) {
      hipError_t e_ = hipMalloc((void**)&E->d_stage_in, (size_t)std::max<int64_t>(P.input_bytes_per_replica, 256) * E->R);
      if (e_ != hipSuccess) { g_err = "hipMalloc(input staging) failed"; return e_ == hipErrorOutOfMemory ? CTN_OOM : CTN_HIP_ERROR; }
    }
    // many small operands: pack them on the host and issue ONE copy (1001 tiny copies cost ms)
    const size_t total_in = (size_t)P.input_bytes_per_replica * E->R;
    const bool packed = total_in <= (8u << 20) && (size_t)P.n_inputs * E->R > 8;
    if (packed && E->h_pack_bytes < total_in) {
      if (E->h_pack) (void)hipHostFree(E->h_pack);
      E->h_pack = nullptr; E->h_pack_bytes = 0;
      if (hipHostMalloc((void**)&E->h_pack, total_in, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); }
      else E->h_pack_bytes = total_in;
    }
    const bool use_pack = packed && E->h_pack;
    if (use_pack) HIPCHECK(hipStreamSynchronize(E->stream));  // previous copy out of h_pack has finished
    for (int r = 0; r < E->R; ++r)
      for (int i = 0; i < P.n_inputs; ++i) {
        const void* src = inputs[(size_t)r * P.n_inputs + i];
        if (!src) { g_err = "null operand pointer"; return CTN_INVALID_ARG; }
        const size_t off = (size_t)r * P.input_bytes_per_replica + P.input_offsets[i];
        char* dst = E->d_stage_in + off;
        if (use_pack) memcpy(E->h_pack + off, src, (size_t)P.tensors[i].numel * es);
        else HIPCHECK(hipMemcpyAsync(dst, src, (size_t)P.tensors[i].numel * es, hipMemcpyHostToDevice, E->stream));
        din[(size_t)r * P.n_inputs + i] = dst;
      }
    if (use_pack) HIPCHECK(hipMemcpyAsync(E->d_stage_in, E->h_pack, total_in, hipMemcpyHostToDevice, E->stream));
  } else {
    for (size_t i = 0; i < din.size(); ++i) din[i] = inputs[i];
  }
  if (outs_space == CTN_MEM_HOST) {
    if (!E->d_stage_out) {
      hipError_t e_ = hipMalloc((void**)&E->d_stage_out, (size_t)out_slot * E->R);
      if (e_ != hipSuccess) { g_err = "hipMalloc(output staging) failed"; return e_ == hipErrorOutOfMemory ? CTN_OOM : CTN_HIP_ERROR; }
    }
    for (int r = 0; r < E->R; ++r) dout[r] = E->d_stage_out + (size_t)r * out_slot;
  } else {
    for (int r = 0; r < E->R; ++r) dout[r] = outs[r];
  }
  int rc = exec_set_pointers(E, din.data(), dout.data());
  if (rc != CTN_OK) return rc;
  exec_new_run(E);
  rc = exec_launch_all(E);
  if (rc != CTN_OK) return rc;
  rc = exec_fetch_checked(E, log_scale);      // may repeat the contraction in eager-rescale mode
  if (rc != CTN_OK) return rc;
  if (step_rescales) memcpy(step_rescales, E->h_resc.data(), E->h_resc.size() * 8);
  if (outs_space == CTN_MEM_HOST) {
    for (int r = 0; r < E->R; ++r) {
      if (!outs[r]) { g_err = "null output pointer"; return CTN_INVALID_ARG; }
      HIPCHECK(hipMemcpyAsync(outs[r], dout[r], (size_t)out_bytes, hipMemcpyDeviceToHost, E->stream));
    }
    HIPCHECK(hipStreamSynchronize(E->stream));
  }
  return CTN_OK;
}

int ctn_exec_snapshot_scales(ctn_exec* exec, double* dev_log_dst, int n, double* host_rescales) {
  if (!exec || n < 0 || n > exec->e.R) { g_err = "invalid argument to ctn_exec_snapshot_scales"; return CTN_INVALID_ARG; }
  Exec* E = &exec->e;
  DeviceGuard dg(E->device);
  HIPCHECK(dg.err);
  if (dev_log_dst && n)
    HIPCHECK(hipMemcpyAsync(dev_log_dst, E->d_log, (size_t)n * 8, hipMemcpyDeviceToDevice, E->stream));
  if (host_rescales && n)
    HIPCHECK(hipMemcpyAsync(host_rescales, E->d_resc, (size_t)n * E->plan->n_steps * 8, hipMemcpyDeviceToHost, E->stream));
  return CTN_OK;
}

int ctn_exec_scales_suspect(const ctn_exec* exec, const double* host_rescales, int replicas) {
  if (!exec || !host_rescales || replicas < 0) { g_err = "invalid argument to ctn_exec_scales_suspect"; return CTN_INVALID_ARG; }
  if (exec->e.eager_rescale) return 0;     // an eager run normalises every intermediate: nothing to suspect
  return exec_scales_suspect(&exec->e, host_rescales, replicas) ? 1 : 0;
}

int ctn_exec_report_suspect(ctn_exec* exec, int suspect) {
  if (!exec) { g_err = "NULL argument"; return CTN_INVALID_ARG; }
  Exec* E = &exec->e;
  if (E->eager_forced) return E->eager_streak;
  E->eager_streak = suspect ? E->eager_streak + 1 : (E->eager_streak >= kEagerSticky ? E->eager_streak : 0);
  if (E->eager_streak >= kEagerSticky) E->eager_rescale = true;     // exec_new_run keeps it from now on
  return E->eager_streak;
}

int ctn_exec_combine_split(ctn_exec* exec, int t_dtype, const void* t, int64_t t_stride, const double* c,
                           int64_t c_stride, int n, int64_t numel, double* out_packed) {
  if (!exec || !t || !c || !out_packed || n < 1 || numel < 1) { g_err = "invalid argument to ctn_exec_combine_split"; return CTN_INVALID_ARG; }
  if (n > kCombineMaxParts) { g_err = "ctn_exec_combine_split: more than 4096 parts"; return CTN_UNSUPPORTED; }
  if (t_dtype != CTN_F32 && t_dtype != CTN_F64) { g_err = "ctn_exec_combine_split: dtype must be f32 or f64"; return CTN_UNSUPPORTED; }
  Exec* E = &exec->e;
  DeviceGuard dg(E->device);
  HIPCHECK(dg.err);
  if (t_dtype == CTN_F32)
    hipLaunchKernelGGL(k_combine_split<float>, dim3(1), dim3(256), 0, E->stream, (const float*)t, t_stride, c, c_stride, n,
                       numel, E->plan->min_norm, out_packed);
  else
    hipLaunchKernelGGL(k_combine_split<double>, dim3(1), dim3(256), 0, E->stream, (const double*)t, t_stride, c, c_stride, n,
                       numel, E->plan->min_norm, out_packed);
  HIPCHECK(hipGetLastError());
  return CTN_OK;
}

int ctn_exec_add_scales(ctn_exec* exec, double* dst, const double* own, int n, int n_kids, const double* const* kid_scales,
                        const int64_t* const* kid_index) {
  if (!exec || !dst || !own || n < 0 || n_kids < 0 || (n_kids && (!kid_scales || !kid_index))) {
    g_err = "invalid argument to ctn_exec_add_scales";
    return CTN_INVALID_ARG;
  }
  if (n == 0) return CTN_OK;
  Exec* E = &exec->e;
  DeviceGuard dg(E->device);
  HIPCHECK(dg.err);
  const double* src = own;
  int done = 0;
  do {                                           // (more than 8 stages below one: several launches, dst carried along)
    ScalesAddArgs a{};
    a.dst = dst; a.own = src; a.n = n;
    a.n_kids = std::min(kScalesAddMaxKids, n_kids - done);
    for (int j = 0; j < a.n_kids; ++j) {
      if (!kid_scales[done + j] || !kid_index[done + j]) { g_err = "ctn_exec_add_scales: NULL child array"; return CTN_INVALID_ARG; }
      a.kid[j] = kid_scales[done + j]; a.idx[j] = kid_index[done + j];
    }
    hipLaunchKernelGGL(k_scales_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, E->stream, a);
    done += a.n_kids;
    src = dst;
  } while (done < n_kids);
  HIPCHECK(hipGetLastError());
  return CTN_OK;
}

int ctn_exec_merge_scales(ctn_exec* exec, int t_dtype, void* buf, int64_t stride, int64_t numel, double* scales, int n,
                          int ndim, const int32_t* extents, const int32_t* merged) {
  if (!exec || !buf || !scales || n < 1 || numel < 1 || stride < numel || ndim < 1 || ndim > kMergeMaxDims || !extents || !merged) {
    g_err = "invalid argument to ctn_exec_merge_scales";
    return CTN_INVALID_ARG;
  }
  if (t_dtype != CTN_F32 && t_dtype != CTN_F64) { g_err = "ctn_exec_merge_scales: dtype must be f32 or f64"; return CTN_UNSUPPORTED; }
  const int esz = t_dtype == CTN_F32 ? 4 : 8;
  if ((uintptr_t)buf % 16 || (stride * esz) % 16) { g_err = "ctn_exec_merge_scales: evaluations must start on 16-byte boundaries"; return CTN_INVALID_ARG; }
  int64_t cells = 1;
  for (int d = 0; d < ndim; ++d) { if (extents[d] < 1) { g_err = "ctn_exec_merge_scales: empty axis"; return CTN_INVALID_ARG; } cells *= extents[d]; }
  if (cells != n || n > 65535) { g_err = "ctn_exec_merge_scales: the grid of evaluations does not match n (<= 65535)"; return CTN_INVALID_ARG; }
  Exec* E = &exec->e;
  DeviceGuard dg(E->device);
  HIPCHECK(dg.err);
  const size_t need = (size_t)n * 16;
  if (E->merge_bytes < need) {
    if (E->d_merge) { HIPCHECK(hipStreamSynchronize(E->stream)); (void)hipFree(E->d_merge); E->d_merge = nullptr; E->merge_bytes = 0; }
    hipError_t e_ = hipMalloc((void**)&E->d_merge, need);
    if (e_ != hipSuccess) { g_err = "hipMalloc(merge scratch) failed"; return e_ == hipErrorOutOfMemory ? CTN_OOM : CTN_HIP_ERROR; }
    E->merge_bytes = need;
  }
  double* cum_new = (double*)E->d_merge;
  int32_t* flags = (int32_t*)(E->d_merge + (size_t)n * 8);
  HIPCHECK(hipMemsetAsync(flags, 0, (size_t)n * 4, E->stream));
  // an evaluation is cut into chunks of whole 16-byte vectors; enough workgroups to fill the chip, at most 4096 per evaluation
  const int64_t per_wg = 256 * 16;
  int64_t chunks = std::min<int64_t>(4096, std::max<int64_t>(1, (numel + per_wg - 1) / per_wg));
  chunks = std::max<int64_t>(1, std::min<int64_t>(chunks, std::max<int64_t>(1, (8LL * E->n_cu + n - 1) / n)));
  const int64_t chunk = ((numel + chunks - 1) / chunks + 1023) / 1024 * 1024;
  chunks = (numel + chunk - 1) / chunk;
  const dim3 grid((unsigned)chunks, (unsigned)n);
  MergeArgs a{};
  a.buf = buf; a.stride = stride; a.numel = numel; a.chunk = chunk; a.cum = scales; a.cum_new = cum_new; a.flags = flags;
  a.n = n; a.ndim = ndim;
  for (int d = 0; d < ndim; ++d) { a.ext[d] = extents[d]; a.merged[d] = merged[d] ? 1 : 0; }
  if (t_dtype == CTN_F32) {
    hipLaunchKernelGGL(k_merge_live<float>, grid, dim3(256), 0, E->stream, (const float*)buf, stride, numel, chunk, flags);
    hipLaunchKernelGGL(k_merge_scale<float>, grid, dim3(256), 0, E->stream, a);
  } else {
    hipLaunchKernelGGL(k_merge_live<double>, grid, dim3(256), 0, E->stream, (const double*)buf, stride, numel, chunk, flags);
    hipLaunchKernelGGL(k_merge_scale<double>, grid, dim3(256), 0, E->stream, a);
  }
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipMemcpyAsync(scales, cum_new, (size_t)n * 8, hipMemcpyDeviceToDevice, E->stream));
  return CTN_OK;
}

int ctn_exec_set_timing(ctn_exec* exec, int slots) {
  if (!exec || slots < 0) { g_err = "invalid argument"; return CTN_INVALID_ARG; }
  Exec* E = &exec->e;
  DeviceGuard dg(E->device);
  HIPCHECK(dg.err);
  HIPCHECK(hipStreamSynchronize(E->stream));
  const size_t need = (size_t)slots * E->plan->n_steps * 2;
  while (E->events.size() < need) {
    hipEvent_t ev;
    HIPCHECK(hipEventCreate(&ev));
    E->events.push_back(ev);
  }
  E->timing_slots = slots;
  E->timing_runs = 0;
  return CTN_OK;
}

int ctn_exec_step_tile(const ctn_exec* exec, int step, int32_t* tile_m, int32_t* tile_n) {
  if (!exec || !tile_m || !tile_n || step < 0 || step >= exec->e.plan->n_steps) { g_err = "invalid step query"; return CTN_INVALID_ARG; }
  const int32_t v = step < (int)exec->e.launched_tile.size() ? exec->e.launched_tile[step] : 0;
  *tile_m = v >> 16;
  *tile_n = v & 0xFFFF;
  return CTN_OK;
}

int ctn_exec_step_ms(ctn_exec* exec, float* ms) {
  if (!exec || !ms) { g_err = "NULL argument"; return CTN_INVALID_ARG; }
  Exec* E = &exec->e;
  const int used = std::min(E->timing_runs, E->timing_slots);
  if (used < 1) { g_err = "no timed run recorded: call ctn_exec_set_timing(slots) then enqueue"; return CTN_INVALID_ARG; }
  HIPCHECK(hipStreamSynchronize(E->stream));
  const int ns = E->plan->n_steps;
  const bool chain = E->plan->chain && E->d_chain != nullptr;
  for (int s = 0; s < ns; ++s) {
    double acc = 0;
    if (chain && s > 0) { ms[s] = 0.f; continue; }  // the persistent walker is one launch, timed as step 0
    for (int slot = 0; slot < used; ++slot) {
      float t = 0;
      const size_t ev0 = ((size_t)slot * ns + s) * 2;
      HIPCHECK(hipEventElapsedTime(&t, E->events[ev0], E->events[ev0 + 1]));
      acc += t;
    }
    ms[s] = (float)(acc / used);
  }
  return CTN_OK;
}

}  // extern "C"
