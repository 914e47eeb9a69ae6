// kernels_mfma_lat.h - fp32 MFMA GEMM step for ONE network in flight: one launch, K split inside the workgroup
// Part of the gfx950 contraction engine (see engine.hip for the overview).
#pragma once
#include <type_traits>

#include "kernels_mfma.h"

namespace ctn {

// ---------------------------------------------------------------------------
// K-mfma-f32-lat: LATENCY form of a GEMM step whose launch cannot fill the chip (a single mid-size network, a few
// replicas: the loop of reference einsum.py:341-391 is a chain of dependent steps, so what counts is the time of
// ONE step, not the rate).  The split-K form (k_mfma_f32_sk + k_splitk_reduce) spreads K over workgroups and
// needs a second launch to add the slabs - two dependent launches per step.  Here the K range is split over the
// EIGHT WAVES of one 512-thread workgroup instead, and the tile is chosen so small that the step still has a
// workgroup for every CU:
//
//   T = 16:  16 x 16 output tile (one v_mfma_f32_16x16x4_f32 accumulator), every wave one eighth of K
//   T = 32:  32 x 32 output tile (one v_mfma_f32_32x32x2_f32 accumulator), every wave one eighth of K
//   T = 64:  64 x 64 output tile, four quadrant waves x two K halves (short K, wide outputs)
//
// A tile is exactly one accumulator block per wave, and with K split over the waves NO operand element is used by
// two waves: fragments go straight from global memory (L2: the operands of such a step are a few MB at most) into
// the MFMA operand registers, table-driven like every other kernel - no operand staging in LDS, no barrier in the
// k loop.  LDS holds the k-offset tables (copied once, so a k-step's two table look-ups are LDS reads instead of
// another dependent global round trip) and, after the loop, the partial accumulators, which are added in wave
// order (fixed: bit-reproducible), rescaled lazily like every tile kernel, stored, and summed into ONE abs-sum
// partial per tile (at most kMaxPartials tiles per replica: consumers add the partials with one wave as usual).
// ---------------------------------------------------------------------------
constexpr int kLatMaxK = 4096;      // k-offset tables of both operands in LDS: 2 x 16 KiB

typedef float f32x4v __attribute__((ext_vector_type(4)));

typedef double f64x4v __attribute__((ext_vector_type(4)));

// F = float: T = 16 / 32 / 64 as above.  F = double: T = 16 only (v_mfma_f64_16x16x4_f64; the fp64 accumulator
// keeps row (lane >> 4) + 4 * reg, not 4 * (lane >> 4) + reg like the fp32 one).
template <int T, typename F = float>
__global__ __launch_bounds__(512) void k_mfma_lat(StepArgs a, int kchunk) {
  static_assert(T == 16 || T == 32 || T == 64, "tile edge");
  static_assert(std::is_same<F, float>::value || T == 16, "fp64: 16 x 16 tiles only");
  constexpr bool F64 = std::is_same<F, double>::value;
  constexpr int MB = T == 16 ? 16 : 32;        // MFMA block edge
  constexpr int KP = T == 16 ? 4 : 2;          // k per MFMA (16x16x4 / 32x32x2)
  constexpr int NA = MB * MB / 64;             // accumulator registers per lane (4 / 16)
  constexpr int Q = T / MB;                    // blocks per tile edge (1, 1, 2)
  constexpr int KS = 8 / (Q * Q);              // K splits (waves per block)
  constexpr int EPT = (T * T + 511) / 512;     // output elements per thread (T = 16: threads 0..255 one each)
  __shared__ int s_okA[kLatMaxK], s_okB[kLatMaxK];
  __shared__ __attribute__((aligned(16))) F s_part[8][MB * MB];       // one partial block per wave
  __shared__ double red[8];

  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int lr = lane % MB, kq = lane / MB;       // row / column inside the block, k slot inside an MFMA step
  const int pid = blockIdx.x;
  const int r = pid / a.blocks_per_replica;        // here: T x T tiles per replica
  const int t = pid - r * a.blocks_per_replica;
  const int tiles_mn = a.tiles_m * a.tiles_n;
  const int b = t / tiles_mn;
  const int tt = t - b * tiles_mn;
  const int m0 = (tt / a.tiles_n) * T, n0 = (tt % a.tiles_n) * T;
  const int q = w % (Q * Q), ks = w / (Q * Q);     // block and K split of this wave
  const int qm = (q / Q) * MB, qn = (q % Q) * MB;

  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const F* __restrict__ A = (const F*)tp[a.idA] + a.obA[b];
  const F* __restrict__ B = (const F*)tp[a.idB] + a.obB[b];
  F* __restrict__ C = (F*)tp[a.idC] + a.obC[b];

  // ONE round trip for everything that does not depend on other loads - requested back to back, first used at the
  // table copy below: the producers' abs-sum partials (reduced to the rescale factors after the loop, like
  // k_mfma_f32_g), this lane's row / column offsets, the C offsets of the elements this thread will store, and the
  // k-offset tables on their way to LDS.
  double pva = 0.0, pvb = 0.0;
  if (a.partA) {
    const double* __restrict__ pr = a.partA + (size_t)r * a.strideA;
    pva = pr[min(lane, a.PA - 1)];
    if (a.PA > 64)   // rare (a producer with more than 64 workgroups per replica): only this branch waits for the first load
      for (int i = lane + 64; i < a.PA; i += 64) pva += pr[i];
  }
  if (a.partB) {
    const double* __restrict__ pr = a.partB + (size_t)r * a.strideB;
    pvb = pr[min(lane, a.PB - 1)];
    if (a.PB > 64)
      for (int i = lane + 64; i < a.PB; i += 64) pvb += pr[i];
  }
  // rows / columns beyond M / N read padded table entries (in bounds) and are dropped at the store
  int offA = a.omA[m0 + qm + lr], offB = a.onB[n0 + qn + lr];
  int offc[EPT];
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int idx = (tid + 512 * i) % (T * T);
    offc[i] = a.omC[m0 + idx / T] + a.onC[n0 + idx % T];
  }
  for (int k = tid; k < a.K; k += 512) { s_okA[k] = a.okA[k]; s_okB[k] = a.okB[k]; }
  asm volatile("" : "+v"(offA), "+v"(offB), "+v"(pva), "+v"(pvb));
#pragma unroll
  for (int i = 0; i < EPT; ++i) asm volatile("" : "+v"(offc[i]));
  __syncthreads();

  typedef typename std::conditional<F64, f64x4v, typename std::conditional<T == 16, f32x4v, f32x16>::type>::type acc_t;
  acc_t acc;
#pragma unroll
  for (int e = 0; e < NA; ++e) acc[e] = 0;
  auto mfma = [&](F x, F y) {
    if constexpr (F64) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);
    else if constexpr (T == 16) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
  };
  const int kbeg = ks * kchunk, kend = min(a.K, kbeg + kchunk);   // kchunk is a multiple of KP; a wave may own nothing
  const F* __restrict__ pa = A + offA;
  const F* __restrict__ pb = B + offB;
  // Rounds of U MFMA steps, software-pipelined over two register sets: the 2U gathers of round r + 1 (their table
  // look-ups are LDS reads) are all in flight while the U MFMAs of round r issue.  U = 32: a wave's share of a
  // K = 1024 step is one or two rounds, i.e. ALL its loads are requested before its first MFMA - one L2 round trip
  // per step instead of one per round.
  constexpr int U = F64 ? 16 : 32;     // (fp64 operands take two registers each)
  F xa0[U], xb0[U], xa1[U], xb1[U];
  auto LOAD = [&](F (&ya)[U], F (&yb)[U], int k0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ya[u] = pa[s_okA[k0 + KP * u + kq]];
      yb[u] = pb[s_okB[k0 + KP * u + kq]];
    }
  };
  auto MMA = [&](const F (&ya)[U], const F (&yb)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) mfma(ya[u], yb[u]);
  };
  const int nround = kend > kbeg ? (kend - kbeg) / (KP * U) : 0;
  int rd = 0;
  if (nround > 0) LOAD(xa0, xb0, kbeg);
  for (; rd + 2 <= nround; rd += 2) {
    LOAD(xa1, xb1, kbeg + KP * U * (rd + 1));
    __builtin_amdgcn_sched_barrier(0);
    MMA(xa0, xb0);
    __builtin_amdgcn_sched_barrier(0);
    if (rd + 2 < nround) LOAD(xa0, xb0, kbeg + KP * U * (rd + 2));
    __builtin_amdgcn_sched_barrier(0);
    MMA(xa1, xb1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (rd < nround) MMA(xa0, xb0);
  int k = kbeg + KP * U * nround;
  // leftover of fewer than U steps: groups of 4 steps with their 8 gathers in flight together
  for (; k + 4 * KP <= kend; k += 4 * KP) {
    F ya[4], yb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { ya[u] = pa[s_okA[k + KP * u + kq]]; yb[u] = pb[s_okB[k + KP * u + kq]]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) mfma(ya[u], yb[u]);
  }
  for (; k < kend; k += KP) {                                      // tail: masked k (a zero pair adds nothing)
    const int kk = k + kq;
    const bool in = kk < kend;
    const F xa = in ? pa[s_okA[kk]] : (F)0;
    const F xb = in ? pb[s_okB[kk]] : (F)0;
    mfma(xa, xb);
  }

  // the wave's partial block, [row][col] with MB-float rows.  32 x 32: register e of lane (col, h) is row
  // (e&3)+8(e>>2)+4h; 16 x 16: register e of lane (col, g) is row 4g+e
#pragma unroll
  for (int e = 0; e < NA; ++e) {
    const int row = F64 ? kq + 4 * e : (T == 16 ? 4 * kq + e : (e & 3) + 8 * (e >> 2) + 4 * kq);
    s_part[w][row * MB + lr] = acc[e];
  }
  __syncthreads();

  // the rescale factors from the partials requested at the start: exactly producer_scale<float>()
  pva = lane < a.PA ? pva : 0.0;
  pvb = lane < a.PB ? pvb : 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { pva += __shfl_xor(pva, o, 64); pvb += __shfl_xor(pvb, o, 64); }
  const F nA = (F)pva, nB = (F)pvb;
  const F scA = (a.partA && nA > (F)a.min_norm) ? nA / (F)a.numelA : (F)1;
  const F scB = (a.partB && nB > (F)a.min_norm) ? nB / (F)a.numelB : (F)1;
  // every thread finishes up to EPT elements: the KS partials of its block in wave order, lazy rescale, store
  const F iA = (F)1 / scA, iB = (F)1 / scB;
  F asum = 0;
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int idx = tid + 512 * i;                 // element of the tile, column fastest
    if (idx < T * T) {
      const int row = idx / T, col = idx % T;
      const int qq = (row / MB) * Q + (col / MB);
      const int off = (row % MB) * MB + (col % MB);
      F v = s_part[qq][off];
#pragma unroll
      for (int s = 1; s < KS; ++s) v += s_part[qq + s * Q * Q][off];
      v = (v * iA) * iB;
      if (m0 + row < a.M && n0 + col < a.N) {
        C[offc[i]] = v;
        asum += fabs(v);
      }
    }
  }
  // fixed-order sum over the workgroup (8 waves)
  double vsum = (double)asum;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) vsum += __shfl_xor(vsum, o, 64);
  if (lane == 0) red[w] = vsum;
  __syncthreads();
  if (tid == 0) {
    double tot = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) tot += red[i];
    a.partC[(size_t)r * a.partC_stride + t] = tot;
  }
}

}  // namespace ctn
