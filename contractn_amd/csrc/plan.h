// plan.h - host-side lowering of a pairwise contraction DAG to kernel launches.
//
// Pure host code (no HIP): every step of the reference's contraction list
// (reference contractn/einsum.py:341-391) is classified label by label into
// batch / M / N / K groups, intermediates get a layout chosen for the consumer,
// and gather-offset tables are built so the kernels never see strings, strides
// or transposes (the reference's `_tensordot` + lazy `_transpose`,
// einsum.py:371-377, collapse into table-driven loads/stores).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "ctn_abi.h"

namespace ctn {

constexpr int kMaxPartials = 512;   // abs-sum partial slots per (step, replica) at most: EVERY workgroup of the consumer adds
                                   // them (8 per lane, fixed order), so their number is paid workgroups x slots times -
                                   // 4096 slots cost a 512-workgroup GEMM 12 us; a step with more workgroups than this
                                   // goes through k_collapse (one 5 us launch, one slot)
constexpr int kWaveOutputs = 64;   // "a handful of outputs": k_dot takes at most this many, the split-K reduce pass spreads over as many workgroups
constexpr int kTileM = 128;        // MFMA f32 workgroup tile
constexpr int kTileN = 128;
constexpr int kTile64M = 128;      // MFMA f64 workgroup tile: 128 x 64
constexpr int kTile64N = 64;
constexpr bool kEnableMfmaF64 = true;
constexpr int kPadK = 32;          // k-offset tables are padded to this multiple
constexpr int kStreamMaxBlocks = 4096;  // streaming kernels: grid cap (16 workgroups per CU), threads stride
constexpr int kChainMaxSteps = 4096;   // rescale factors of all steps live in LDS
constexpr int64_t kChainMaxOut = 4096;  // per-step output elements
constexpr int64_t kChainMaxWork = 1 << 16;  // per-step multiply-adds
constexpr int64_t kAlign = 256;    // byte alignment of workspace tensors

struct Tensor {
  std::vector<int32_t> labels;
  std::vector<int64_t> dims;
  std::vector<int64_t> strides;  // in elements
  int64_t numel = 1;
  bool is_input = false;
  int producer = -1;      // step that writes it (-1 for inputs)
  int64_t ws_offset = -1; // byte offset inside one replica's workspace (intermediates)
};

// offsets of one step's tables: m / n / k / lo tables in int32 entries of Plan::tables; the BATCH tables (ob*) and the
// streaming kernels' outer-group tables (oh*) in int64 entries of Plan::tables64 - those carry the large strides: a
// tensor of 2^31 elements or more is addressed as (64-bit batch / outer offset) + (32-bit offsets inside one batch
// entry), the planner moving outer free labels into the batch group until every 32-bit table fits (plan.cpp)
struct TableRefs {
  int64_t obA = 0, obB = 0, obC = 0, omA = 0, omC = 0, onB = 0, onC = 0, okA = 0, okB = 0;
  int64_t obA2 = 0, omA2 = 0, okA2 = 0;  // fused steps (modeA == 3): the second tensor of the A side
  int64_t ohA = 0, ohB = 0, ohC = 0, olA = 0, olB = 0, olC = 0;  // streaming kernels: (hi, lo) output groups
};

struct Step {
  int lhs = -1, rhs = -1, out = -1;  // tensor ids after the optional operand swap; rhs -1 = unary
  int lhs2 = -1;         // fused steps: A = tensors[lhs] (.) tensors[lhs2], formed on the fly (modeA 3, 4, 5)
  int krX = 0, krY = 0;  // modeA 4 / 5: how each factor is read along the vector direction (0 gather, 1 float4, 2 broadcast)
  int epw = 0;           // 2 / 4: the step is a GEMM whose innermost column label (this extent) is re-weighted by
                         // tensors[lhs2] and summed in the epilogue ("bl,plr->bpr" then "bpr,bp->br" as one step)
  bool epw_split = false;  // ... with the columns ordered (.., u_hi, p, u_lo), |u_lo| = 4: a lane's four columns are four
                           // consecutive u of one p (vector accesses); the sum over p runs across adjacent lanes
  bool swapped = false;
  int kernel = CTN_KERNEL_ELEMENT;
  int64_t Bt = 1, M = 1, N = 1, K = 1;
  bool has_k = false;  // false: pure product (no summed label)
  int modeA = 0, modeB = 0;
  // streaming decomposition of the output index space (element / row-dot kernels): (hi, lo, n),
  // n runs along C's unit-stride label, sAn/sBn are the operands' strides along it (0 = broadcast)
  int64_t H = 1, L = 1, Nv = 1, sAn = 0, sBn = 0;
  int vecw = 1;          // output elements per thread (16-byte vectors when > 1)
  int kvec = 0;          // streaming step whose left operand is unit-stride along a short K: 16-byte loads along k
  bool chain_ok = false; // small enough for the persistent chain walker
  int tileN = kTileN;    // MFMA f32 column tile: 128, or 64 when that wastes less padding
  int tileM = kTileM;    // MFMA f32 row tile: 128, or 256 = the large-tile LDS-DMA kernel (kernels_mfma_g.h)
  bool cvec = false;   // 16-byte vector stores of C are valid (unit-stride column label, aligned strides)
  int blocks = 1;      // workgroups per replica
  int partials = 1;    // partial abs-sums per replica after the optional collapse pass
  bool collapse = false;
  double flops = 0;
  TableRefs t;
};

struct Plan {
  int dtype = CTN_F32;
  int n_inputs = 0;
  int n_steps = 0;
  bool stabilize = true;
  bool free_out_order = false;   // the last step's result takes the engine's natural layout (ctn_plan_desc.stabilize bit 1)
  double min_norm = 1e-7;
  std::vector<Tensor> tensors;  // n_inputs inputs, then one per step
  std::vector<Step> steps;
  std::vector<int32_t> tables;
  std::vector<int64_t> tables64;  // batch (ob*) and outer-group (oh*) offset tables, see TableRefs
  int64_t ws_bytes_per_replica = 0;
  int64_t input_bytes_per_replica = 0;  // staging size when operands arrive as host pointers
  std::vector<int64_t> input_offsets;   // byte offset of each input inside the staging block
  int64_t max_collapse_blocks = 0;
  bool chain = false;  // every step is tiny: one persistent workgroup per replica walks the whole DAG
  double flops = 0;
  int64_t bytes_min = 0;

  size_t elem_size() const { return dtype == CTN_F64 ? 8 : 4; }
  const Tensor& output() const { return tensors.back(); }
};

// Build a plan; on failure returns a negative ctn_status and fills `err`.
int build_plan(const ctn_plan_desc& d, Plan& plan, std::string& err);

}  // namespace ctn
