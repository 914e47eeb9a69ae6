// kernel_args.h - kernel argument structs and device helpers shared by every kernel
// Part of the gfx950 contraction engine (see engine.hip for the overview).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "plan.h"

namespace ctn {

// ---------------------------------------------------------------------------
// kernel arguments
// ---------------------------------------------------------------------------
// n / d for 0 <= n < 2^31 by multiply-shift (host-precomputed): exact, ~5 instructions instead of
// the ~40 (32-bit) / ~100 (64-bit) of a hardware-less integer division.
struct FastDiv {
  uint64_t M;
  uint32_t d;
  int32_t k;
  __device__ __forceinline__ uint32_t div(uint32_t n) const { return (uint32_t)(((uint64_t)n * M) >> k); }
};
static FastDiv make_fastdiv(int64_t d64) {
  FastDiv f;
  const uint32_t d = (uint32_t)std::max<int64_t>(d64, 1);
  int lg = 0;
  while ((1ull << lg) < d) ++lg;
  f.d = d;
  f.k = 31 + lg;
  f.M = ((1ull << f.k) / d) + 1;   // n * M < 2^31 * 2^(32) fits in 64 bits; exact for n < 2^31
  if (d == 1) { f.M = 1; f.k = 0; }
  return f;
}

struct StepArgs {
  const int64_t *obA, *obB, *obC;   // batch offsets: 64-bit (tensors of 2^31 elements and more, see plan.h TableRefs)
  const int32_t *omA, *omC, *onB, *onC, *okA, *okB;
  void* const* ptrs;    // [R][n_tensors] base pointer of every tensor of every replica
  const double* partA;  // [R][strideA] abs-sum partials of A's producer step (PA of them are valid), nullptr for inputs
  const double* partB;
  double* partC;        // [R][partC_stride] where this step's partials go
  double numelA, numelB;
  double min_norm;
  int32_t Bt, M, N, K;
  int32_t idA, idB, idC, n_tensors;
  int32_t PA, PB;
  int32_t strideA, strideB;  // slots per replica in partA / partB (>= PA / PB)
  int32_t partC_stride;
  int32_t tiles_m, tiles_n;
  int32_t blocks_per_replica;
  int32_t R;
  int32_t c_vec;  // float4 stores of C allowed
  // streaming kernels: output index = (hi, lo, n); n along C's unit-stride label
  const int64_t *ohA, *ohB, *ohC;   // outer group: 64-bit
  const int32_t *olA, *olB, *olC;
  int32_t H, L, Nv, sAn, sBn;
  FastDiv dNq, dL, dNv;  // divisors: vectors per row, lo extent, n extent
  unsigned long long* dbg;  // CTN_STAMPS builds only: 4 cycle stamps per MFMA tile (else unused, null)
  // K split of a streaming step with few outputs and a long K (k_stream, blockIdx.z = split): raw partial sums
  // go to slab s of the split-K scratch instead of C; 0 splits = the plain form
  void* ks_slab;
  int64_t ks_numelC;
  int32_t ks_S, ks_chunk;
  // fused element-wise product as the A operand ("KR", mode_a = 3): A[m][k] = X[..] * Y[..]; X is described by the
  // A fields above, Y by these
  const int64_t* obA2;
  const int32_t *omA2, *okA2;
  const double* partA2;
  double numelA2;
  int32_t idA2, PA2, strideA2;
  int32_t krX, krY;   // modes 4 / 5: how each factor is read along the vector direction (0 gather, 1 float4, 2 broadcast)
  // epilogue-summed GEMM (planner pattern C): extent (2 / 4) of the innermost column label that tensor idA2 -
  // offsets of the rows in omA2, the label itself unit-stride - re-weights and sums on the way out; 0 = plain step
  int32_t epw;        // | 0x100: columns ordered (.., u_hi, p, u_lo), see Step::epw_split
};

struct FinalArgs {
  void* const* ptrs;
  const double* partials;   // step s, replica r: partials + stepOff[s] * R + r * stepSlots[s]
  const int64_t* stepOff;   // [n_steps] offset of each step's region, in doubles per replica
  const int32_t* stepSlots; // [n_steps] slots per replica of each step's region
  const int32_t* stepP;     // [n_steps] partial count of each step (<= its slots)
  const double* stepNumel;  // [n_steps] numel of each step's output
  double* log_scale;        // [R]
  double* rescales;         // [R][n_steps]
  double min_norm;
  int64_t out_numel;
  int32_t n_steps, R, id_out, n_tensors, stabilize;
  int32_t defer;            // 1: k_finalize leaves the final tensor un-normalised (ctn_exec_finish does it, with the caller's factor)
  int32_t vec;              // 1: the final buffers are 16-byte aligned: 16-byte accesses
};

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
// Rescale factor of a tensor from its producer's partial sums (reference
// einsum.py:97-102: norm = sum|T|, rescale = norm / numel, applied iff
// norm > min_norm).  All 64 lanes of the calling wave must be active.
template <typename T>
__device__ __forceinline__ T producer_scale(const double* part, int P, int stride, double numel, double min_norm,
                                            int r, bool* cond_out = nullptr) {
  if (part == nullptr) {
    if (cond_out) *cond_out = false;
    return (T)1;
  }
  const int lane = threadIdx.x & 63;
  // lane l adds partials l, l + 64, ... (P <= kMaxPartials: at most 8 loads), then the xor butterfly: a fixed order
  const double* __restrict__ pr = part + (size_t)r * stride;
  double v = 0.0;
  if (P <= 64) {
    if (lane < P) v = pr[lane];
  } else {
    // all of a lane's (at most 8) partials requested at once - one memory latency, not one per round - and added in
    // index order as a plain loop would (an absent one reads the last slot and counts as +0.0: abs-sums are >= 0)
    const int n4 = P <= 256 ? 4 : 8;
    double d[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = j < n4 ? pr[min(lane + 64 * j, P - 1)] : 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) v += (lane + 64 * j < P) ? d[j] : 0.0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const T norm = (T)v;
  const bool cond = norm > (T)min_norm;
  if (cond_out) *cond_out = cond;
  return cond ? norm / (T)numel : (T)1;
}

// Sum over the workgroup in a fixed order; result valid in every thread.
__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double t = 0;
  const int nw = (blockDim.x + 63) >> 6;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}


}  // namespace ctn
