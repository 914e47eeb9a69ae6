// kernels_stream.h - HBM-bound kernels: streaming hyperedge gather-multiply, row-dot, dot, collapse, finishing passes, chain walker
// Part of the gfx950 contraction engine (see engine.hip for the overview).
#pragma once
#include "kernel_args.h"

namespace ctn {

// ---------------------------------------------------------------------------
// K-element ("stream"): copy-tensor / hyperedge products, Khatri-Rao, Hadamard, traces, small-K
// steps.  HBM-bound gather-multiply: the output index space is (hi, lo, n) with n running along C's
// unit-stride label; one thread produces V consecutive n (one 16-byte store), consecutive lanes
// consecutive vectors, so C is written - and every operand that is unit-stride along n is read -
// in full cache lines; an operand that does not carry the label is a per-thread broadcast.  No
// identity tensor exists anywhere: a copy tensor is only the shared (hi/lo/n) index.
// Operands are divided by their producer's rescale on load, exactly like the reference's
// stabilize() output feeding the next step: bit-identical when the K sum is exact.
// ---------------------------------------------------------------------------
template <typename T, int V>
struct VecOf;
template <> struct VecOf<float, 4> { typedef float4 type; };
template <> struct VecOf<double, 2> { typedef double2 type; };
template <typename T> struct VecOf<T, 1> { typedef T type; };

template <typename T, int V>
__device__ __forceinline__ void load_vec(const T* __restrict__ p, int stride_n, T (&out)[V]) {
  if constexpr (V == 1) {
    out[0] = p[0];
  } else {
    if (stride_n == 0) {
      const T x = p[0];
#pragma unroll
      for (int v = 0; v < V; ++v) out[v] = x;
    } else if (stride_n != 1) {   // strided along n: a gather of V elements (the other operand and C still use vectors)
#pragma unroll
      for (int v = 0; v < V; ++v) out[v] = p[(int64_t)v * stride_n];
    } else {
      typedef typename VecOf<T, V>::type VT;
      const VT x = *reinterpret_cast<const VT*>(p);
      const T* e = reinterpret_cast<const T*>(&x);
#pragma unroll
      for (int v = 0; v < V; ++v) out[v] = e[v];
    }
  }
}

// Item loop of k_stream.  DA / DB (does operand A / B need dividing by its producer's rescale?) are
// template flags chosen ONCE per workgroup, so the loop body has no per-element branches and all
// 2U loads of an item are issued back to back before any arithmetic.
template <typename T, int V, int U, bool DA, bool DB>
__device__ __forceinline__ double stream_items(const StepArgs& a, const T* __restrict__ A, const T* __restrict__ B,
                                               T* __restrict__ C, T sA, T sB, int kbeg = 0, int kend = -1,
                                               uint32_t nblocks = 0) {
  const uint32_t nq_per = (uint32_t)((a.Nv + V - 1) / V);
  const uint32_t S = nq_per / U;                 // U divides nq_per (checked on the host)
  const uint32_t items = (uint32_t)a.H * (uint32_t)a.L * S;  // < 2^31
  const FastDiv dq = a.dNq;                      // divisor S
  const bool kone = a.K == 1;                    // pure product: k-offset tables hold a single 0
  const int64_t stepA = (int64_t)S * V * a.sAn, stepB = (int64_t)S * V * a.sBn;   // (64-bit: tensors of 2^31 elements and more)
  double absv = 0;
  const uint32_t stride = (nblocks ? nblocks : gridDim.x) * 256u;   // (a grouped launch is as wide as its widest step)
  for (uint32_t it = blockIdx.x * 256u + threadIdx.x; it < items; it += stride) {
    const uint32_t row = dq.div(it);
    const int c0 = (int)(it - row * S) * V;
    const int h = (int)a.dL.div(row);
    const int l = (int)(row - (uint32_t)h * (uint32_t)a.L);
    const T* pa = A + a.ohA[h] + a.olA[l] + (int64_t)c0 * a.sAn;
    const T* pb = B + a.ohB[h] + a.olB[l] + (int64_t)c0 * a.sBn;
    T* pc = C + (size_t)row * a.Nv + c0;         // C is contiguous in (hi, lo, n) order
    T acc[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < V; ++v) acc[u][v] = 0;
    const int nk = kone ? 1 : (kend >= 0 ? kend : a.K);
    for (int k = kbeg; k < nk; ++k) {
      const int ka = kone ? 0 : a.okA[k], kb = kone ? 0 : a.okB[k];
      T av[U][V], bv[U][V];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        load_vec<T, V>(pa + ka + u * stepA, a.sAn, av[u]);
        load_vec<T, V>(pb + kb + u * stepB, a.sBn, bv[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const T x = DA ? av[u][v] / sA : av[u][v];
          const T y = DB ? bv[u][v] / sB : bv[u][v];
          acc[u][v] = fma(x, y, acc[u][v]);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (V == 1) {
        pc[u * S] = acc[u][0];
      } else {
        typedef typename VecOf<T, V>::type VT;
        VT o;
        T* e = reinterpret_cast<T*>(&o);
#pragma unroll
        for (int v = 0; v < V; ++v) e[v] = acc[u][v];
        *reinterpret_cast<VT*>(pc + u * S * V) = o;
      }
      T part = 0;
#pragma unroll
      for (int v = 0; v < V; ++v) part += fabs(acc[u][v]);
      absv += (double)part;
    }
  }
  return absv;
}

// one item = U vectors of V elements of one output row (hi, lo): columns c, c + S, ..., c + (U-1) S
// with S = vectors per row / U, so the row's table lookups and broadcast operands are paid once
// per U*V outputs while every store instruction of a wave still covers a contiguous segment.
template <typename T, int V, int U>
__global__ __launch_bounds__(256) void k_stream(StepArgs a) {
  __shared__ double red[4];
  const int r = blockIdx.y;
  const T sA = producer_scale<T>(a.partA, a.PA, a.strideA, a.numelA, a.min_norm, r);
  const T sB = producer_scale<T>(a.partB, a.PB, a.strideB, a.numelB, a.min_norm, r);
  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const T* __restrict__ A = (const T*)tp[a.idA];
  const T* __restrict__ B = (const T*)tp[a.idB];
  T* __restrict__ C = (T*)tp[a.idC];
  if (a.ks_S > 0) {   // K split (few outputs, long K): this workgroup's k-range, un-scaled, into its slab
    const int s = blockIdx.z;
    T* slab = (T*)a.ks_slab + ((size_t)r * a.ks_S + s) * a.ks_numelC;
    (void)stream_items<T, V, U, false, false>(a, A, B, slab, (T)1, (T)1, s * a.ks_chunk, min(a.K, (s + 1) * a.ks_chunk));
    return;
  }
  // x / 1 == x exactly, so the division is skipped for network inputs and un-rescaled tensors
  const bool da = sA != (T)1, db = sB != (T)1;
  double absv;
  if (!da && !db) absv = stream_items<T, V, U, false, false>(a, A, B, C, sA, sB);
  else if (da && !db) absv = stream_items<T, V, U, true, false>(a, A, B, C, sA, sB);
  else if (!da && db) absv = stream_items<T, V, U, false, true>(a, A, B, C, sA, sB);
  else absv = stream_items<T, V, U, true, true>(a, A, B, C, sA, sB);
  const double tot = block_sum(absv, red);
  if (threadIdx.x == 0) a.partC[(size_t)r * a.partC_stride + blockIdx.x] = tot;
}

// Several INDEPENDENT streaming steps in one launch (blockIdx.z = the step; their arguments in device memory): the
// physical-leg absorptions at the leaves of a PEPS / MPS contraction tree are dozens of few-kilobyte products whose
// launches (5.5 us each in the trace) outweigh their work.  Same arithmetic as k_stream step by step.
template <typename T, int V, int U>
__global__ __launch_bounds__(256) void k_stream_group(const StepArgs* __restrict__ steps) {
  __shared__ double red[4];
  const StepArgs& a = steps[blockIdx.z];
  if ((int)blockIdx.x >= a.blocks_per_replica) return;   // uniform per workgroup: narrower than the widest step
  const int r = blockIdx.y;
  const T sA = producer_scale<T>(a.partA, a.PA, a.strideA, a.numelA, a.min_norm, r);
  const T sB = producer_scale<T>(a.partB, a.PB, a.strideB, a.numelB, a.min_norm, r);
  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const T* __restrict__ A = (const T*)tp[a.idA];
  const T* __restrict__ B = (const T*)tp[a.idB];
  T* __restrict__ C = (T*)tp[a.idC];
  const bool da = sA != (T)1, db = sB != (T)1;
  const uint32_t nb = (uint32_t)a.blocks_per_replica;
  double absv;
  if (!da && !db) absv = stream_items<T, V, U, false, false>(a, A, B, C, sA, sB, 0, -1, nb);
  else if (da && !db) absv = stream_items<T, V, U, true, false>(a, A, B, C, sA, sB, 0, -1, nb);
  else if (!da && db) absv = stream_items<T, V, U, false, true>(a, A, B, C, sA, sB, 0, -1, nb);
  else absv = stream_items<T, V, U, true, true>(a, A, B, C, sA, sB, 0, -1, nb);
  const double tot = block_sum(absv, red);
  if (threadIdx.x == 0) a.partC[(size_t)r * a.partC_stride + blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------
// K-stream-kvec: the left operand is unit-stride along a short K (4 .. 64 elements) while the output's unit-stride
// index strides it (`abk,k->ab`, a vector or small tensor applied to the innermost leg): one output per thread, its
// K operand elements fetched as 16-byte vectors (adjacent threads read adjacent K-runs), the other operand through
// its k table (small: cache hits).  Same operations in the same order as k_stream (divide on load, fma over
// ascending k), so the results are bit-identical to it.  4096 x 4096 x 16: 2.4 ms -> see DESIGN.md.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_stream_kvec(StepArgs a) {
  constexpr int V = 16 / sizeof(T);
  typedef typename VecOf<T, V>::type VT;
  __shared__ double red[4];
  const int r = blockIdx.y;
  const T sA = producer_scale<T>(a.partA, a.PA, a.strideA, a.numelA, a.min_norm, r);
  const T sB = producer_scale<T>(a.partB, a.PB, a.strideB, a.numelB, a.min_norm, r);
  const bool da = sA != (T)1, db = sB != (T)1;
  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const T* __restrict__ A = (const T*)tp[a.idA];
  const T* __restrict__ B = (const T*)tp[a.idB];
  T* __restrict__ C = (T*)tp[a.idC];
  const uint32_t outs = (uint32_t)a.H * (uint32_t)a.L * (uint32_t)a.Nv;
  double absv = 0;
  const uint32_t stride = gridDim.x * 256u;          // <= 2^20, outs < 2^31: o + 3 * stride cannot wrap
  uint32_t o = blockIdx.x * 256u + threadIdx.x;
  // four outputs of a thread at a time (its rounds o, o + stride, ...): their operand vectors are requested
  // together - with one output per round a thread has a single 16-byte load in flight (20 MB step: 1.7 TB/s)
  constexpr int U = 4;
  for (; o + (U - 1) * stride < outs; o += U * stride) {
    const T* pa[U];
    const T* pb[U];
    size_t co[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t ou = o + u * stride;
      const uint32_t row = a.dNv.div(ou);
      const int n = (int)(ou - row * (uint32_t)a.Nv);
      const int h = (int)a.dL.div(row);
      const int l = (int)(row - (uint32_t)h * (uint32_t)a.L);
      pa[u] = A + a.ohA[h] + a.olA[l] + (int64_t)n * a.sAn;
      pb[u] = B + a.ohB[h] + a.olB[l] + (int64_t)n * a.sBn;
      co[u] = (size_t)row * a.Nv + n;
    }
    T acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = 0;
    for (int k = 0; k < a.K; k += V) {
      VT xv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) xv[u] = *reinterpret_cast<const VT*>(pa[u] + k);
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int kb = a.okB[k + v];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const T x = reinterpret_cast<const T*>(&xv[u])[v];
          const T y = pb[u][kb];
          acc[u] = fma(da ? x / sA : x, db ? y / sB : y, acc[u]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      C[co[u]] = acc[u];
      absv += (double)fabs(acc[u]);
    }
  }
  for (; o < outs; o += stride) {
    const uint32_t row = a.dNv.div(o);
    const int n = (int)(o - row * (uint32_t)a.Nv);
    const int h = (int)a.dL.div(row);
    const int l = (int)(row - (uint32_t)h * (uint32_t)a.L);
    const T* pa = A + a.ohA[h] + a.olA[l] + (int64_t)n * a.sAn;
    const T* pb = B + a.ohB[h] + a.olB[l] + (int64_t)n * a.sBn;
    T acc = 0;
    for (int k = 0; k < a.K; k += V) {
      const VT xv = *reinterpret_cast<const VT*>(pa + k);
      const T* x = reinterpret_cast<const T*>(&xv);
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const T y = pb[a.okB[k + v]];
        acc = fma(da ? x[v] / sA : x[v], db ? y / sB : y, acc);
      }
    }
    C[(size_t)row * a.Nv + n] = acc;
    absv += (double)fabs(acc);
  }
  const double tot = block_sum(absv, red);
  if (threadIdx.x == 0) a.partC[(size_t)r * a.partC_stride + blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------
// K-rowdot: one WAVE per output element, the 64 lanes stride a unit-stride K (coalesced
// 256-byte reads), xor-butterfly reduction.  GEMV / batched-dot shaped steps.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_rowdot(StepArgs a) {
  __shared__ double red[4];
  const int r = blockIdx.y;
  const T sA = producer_scale<T>(a.partA, a.PA, a.strideA, a.numelA, a.min_norm, r);
  const T sB = producer_scale<T>(a.partB, a.PB, a.strideB, a.numelB, a.min_norm, r);
  const bool divA = sA != (T)1, divB = sB != (T)1;
  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const T* __restrict__ A = (const T*)tp[a.idA];
  const T* __restrict__ B = (const T*)tp[a.idB];
  T* __restrict__ C = (T*)tp[a.idC];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t outs = (uint32_t)a.H * (uint32_t)a.L * (uint32_t)a.Nv;
  if (a.ks_S > 0) {
    // K split (few outputs, each with a long K: 100 x 4,194,304 took 100 waves 29.9 ms): blockIdx.z = k-range, raw
    // partial sums into slab s (laid out like C), four independent chains per lane; the split-K reduce pass adds
    // the slabs in a fixed order, rescales and emits the abs-sum
    const int s = blockIdx.z;
    T* __restrict__ slab = (T*)a.ks_slab + ((size_t)r * a.ks_S + s) * a.ks_numelC;
    const int k0 = s * a.ks_chunk, k1 = min(a.K, k0 + a.ks_chunk);
    for (uint32_t o = blockIdx.x * 4u + w; o < outs; o += gridDim.x * 4u) {
      const uint32_t o2 = a.dNv.div(o);
      const int n = (int)(o - o2 * (uint32_t)a.Nv);
      const int h = (int)a.dL.div(o2);
      const int l = (int)(o2 - (uint32_t)h * (uint32_t)a.L);
      const T* pa = A + a.ohA[h] + a.olA[l] + (int64_t)n * a.sAn;
      const T* pb = B + a.ohB[h] + a.olB[l] + (int64_t)n * a.sBn;
      T acc[4] = {0, 0, 0, 0};
      int k = k0 + lane;
      for (; k + 192 < k1; k += 256) {
        T x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { x[u] = pa[a.okA[k + 64 * u]]; y[u] = pb[a.okB[k + 64 * u]]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = fma(x[u], y[u], acc[u]);
      }
      for (; k < k1; k += 64) acc[0] = fma(pa[a.okA[k]], pb[a.okB[k]], acc[0]);
      double v = ((double)acc[0] + (double)acc[1]) + ((double)acc[2] + (double)acc[3]);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) slab[a.ohC[h] + a.olC[l] + n] = (T)v;
    }
    return;
  }
  double mine = 0;
  for (uint32_t o = blockIdx.x * 4u + w; o < outs; o += gridDim.x * 4u) {  // wave-uniform
    const uint32_t o2 = a.dNv.div(o);
    const int n = (int)(o - o2 * (uint32_t)a.Nv);
    const int h = (int)a.dL.div(o2);
    const int l = (int)(o2 - (uint32_t)h * (uint32_t)a.L);
    const T* pa = A + a.ohA[h] + a.olA[l] + (int64_t)n * a.sAn;
    const T* pb = B + a.ohB[h] + a.olB[l] + (int64_t)n * a.sBn;
    T acc0 = 0, acc1 = 0;
    int k = lane;
    for (; k + 64 < a.K; k += 128) {
      const T x0 = pa[a.okA[k]], y0 = pb[a.okB[k]];
      const T x1 = pa[a.okA[k + 64]], y1 = pb[a.okB[k + 64]];
      acc0 = fma(divA ? x0 / sA : x0, divB ? y0 / sB : y0, acc0);
      acc1 = fma(divA ? x1 / sA : x1, divB ? y1 / sB : y1, acc1);
    }
    if (k < a.K) {
      const T x0 = pa[a.okA[k]], y0 = pb[a.okB[k]];
      acc0 = fma(divA ? x0 / sA : x0, divB ? y0 / sB : y0, acc0);
    }
    double v = (double)acc0 + (double)acc1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const T res = (T)v;
    if (lane == 0) C[a.ohC[h] + a.olC[l] + n] = res;
    mine += (double)fabs(res);
  }
  if (lane == 0) red[w] = mine;
  __syncthreads();
  if (threadIdx.x == 0)
    a.partC[(size_t)r * a.partC_stride + blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// ---------------------------------------------------------------------------
// K-dot: one workgroup per output element, K split over 256 lanes.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_dot(StepArgs a) {
  __shared__ double red[4];
  const int r = blockIdx.y;
  const T sA = producer_scale<T>(a.partA, a.PA, a.strideA, a.numelA, a.min_norm, r);
  const T sB = producer_scale<T>(a.partB, a.PB, a.strideB, a.numelB, a.min_norm, r);
  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const T* __restrict__ A = (const T*)tp[a.idA];
  const T* __restrict__ B = (const T*)tp[a.idB];
  T* __restrict__ C = (T*)tp[a.idC];
  const int o = blockIdx.x;
  const int n = o % a.N;
  const int q = o / a.N;
  const int m = q % a.M;
  const int b = q / a.M;
  const T* pa = A + a.obA[b] + a.omA[m];
  const T* pb = B + a.obB[b] + a.onB[n];
  T acc = 0;
  for (int k = threadIdx.x; k < a.K; k += 256) acc = fma(pa[a.okA[k]] / sA, pb[a.okB[k]] / sB, acc);
  const T tot = (T)block_sum((double)acc, red);
  if (threadIdx.x == 0) {
    const T v = tot;
    C[a.obC[b] + a.omC[m] + a.onC[n]] = v;
    a.partC[(size_t)r * a.partC_stride + o] = (double)fabs(v);
  }
}

// K-dot-split: the same outputs when K is huge (the inner product that closes a network: 1 output, 16.7 M
// terms took one workgroup 36.7 ms): workgroup (o, s) sums k-range s of output o - raw products, four
// independent chains per thread - into slab s of the split-K scratch (laid out like C); the split-K reduce
// pass (k_splitk_fold / k_splitk_reduce) adds the slabs in a fixed order, rescales and emits the abs-sum.
template <typename T>
__global__ __launch_bounds__(256) void k_dot_split(StepArgs a, T* __restrict__ slab, int64_t numelC, int S, int kchunk) {
  __shared__ double red[4];
  const int r = blockIdx.y;
  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const T* __restrict__ A = (const T*)tp[a.idA];
  const T* __restrict__ B = (const T*)tp[a.idB];
  const int o = blockIdx.x / S, s = blockIdx.x - o * S;
  const int n = o % a.N;
  const int q = o / a.N;
  const int m = q % a.M;
  const int b = q / a.M;
  const T* pa = A + a.obA[b] + a.omA[m];
  const T* pb = B + a.obB[b] + a.onB[n];
  const int k0 = s * kchunk, k1 = min(a.K, k0 + kchunk);
  T acc[4] = {0, 0, 0, 0};
  int k = k0 + threadIdx.x;
  for (; k + 768 < k1; k += 1024) {
    T x[4], y[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { x[u] = pa[a.okA[k + 256 * u]]; y[u] = pb[a.okB[k + 256 * u]]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = fma(x[u], y[u], acc[u]);
  }
  for (; k < k1; k += 256) acc[0] = fma(pa[a.okA[k]], pb[a.okB[k]], acc[0]);
  const double mine = ((double)acc[0] + (double)acc[1]) + ((double)acc[2] + (double)acc[3]);
  const T tot = (T)block_sum(mine, red);
  if (threadIdx.x == 0)
    slab[((size_t)r * S + s) * numelC + a.obC[b] + a.omC[m] + a.onC[n]] = tot;
}

// k_dot_tr: a full dot product of two tensors one of which is stored TRANSPOSED relative to the other - the two boundary
// halves of a sliced 2D grid closed by one 262144-long sum (`_tensordot` with every index contracted, reference
// einsum.py:371): k = (a, b), X contiguous in k (offset a Kb + b), Y[b ldY + a].  k_dot_split gathers Y four bytes at a
// time, every lane another cache line (0.8 TB/s); here a workgroup takes 32 x 32 tiles of (a, b): both operands are
// read along their own unit-stride index - 128-byte rows - and Y's tile turns round in LDS.  grid (outputs x S, R);
// split s takes tiles s, s + S, ...; the partial sums go through the split-K reduce pass like k_dot_split's.
template <typename T>
__global__ __launch_bounds__(256) void k_dot_tr(StepArgs a, T* __restrict__ slab, int64_t numelC, int S, int Ka, int Kb,
                                                int64_t ldY, int x_is_a) {
  __shared__ double red[4];
  __shared__ T tile[32][33];
  const int r = blockIdx.y;
  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const int o = blockIdx.x / S, s = blockIdx.x - o * S;
  const int n = o % a.N;
  const int q = o / a.N;
  const int m = q % a.M;
  const int b = q / a.M;
  const T* pa = (const T*)tp[a.idA] + a.obA[b] + a.omA[m];
  const T* pb = (const T*)tp[a.idB] + a.obB[b] + a.onB[n];
  const T* __restrict__ X = x_is_a ? pa : pb;       // contiguous in k
  const T* __restrict__ Y = x_is_a ? pb : pa;       // transposed
  const int j = threadIdx.x & 31, i0 = threadIdx.x >> 5;
  const int tb = Kb / 32, tiles = (Ka / 32) * tb;
  T acc = 0;
  for (int t = s; t < tiles; t += S) {
    const int a0 = (t / tb) * 32, b0 = (t % tb) * 32;
    T x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 8 * u;
      x[u] = X[(int64_t)(a0 + i) * Kb + b0 + j];
      tile[i][j] = Y[(int64_t)(b0 + i) * ldY + a0 + j];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = fma(x[u], tile[j][i0 + 8 * u], acc);
    __syncthreads();
  }
  const T tot = (T)block_sum((double)acc, red);
  if (threadIdx.x == 0)
    slab[((size_t)r * S + s) * numelC + a.obC[b] + a.omC[m] + a.onC[n]] = tot;
}

// Collapse more per-workgroup partials than a consumer wave adds (kMaxPartials) into one, in a fixed order;
// the step's region then has a single slot per replica.
__global__ __launch_bounds__(256) void k_collapse(const double* scratch, int blocks, double* part) {
  __shared__ double red[4];
  const int r = blockIdx.x;
  const double* src = scratch + (size_t)r * blocks;
  double v = 0;
  for (int i = threadIdx.x; i < blocks; i += 256) v += src[i];
  const double tot = block_sum(v, red);
  if (threadIdx.x == 0) part[r] = tot;
}

// ---------------------------------------------------------------------------
// Finishing passes.  k_scales: one wave per (step, replica) turns the step's partials into its
// rescale factor (0.0 = not rescaled) and log(rescale) evaluated in the tensor dtype
// (reference einsum.py:97-106).  k_finalize: normalise the final tensor and sum the logs.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_scales(FinalArgs f, double* __restrict__ logs) {
  const int r = blockIdx.y;
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= f.n_steps) return;  // whole wave exits together
  bool cond = false;
  T sc = (T)1;
  if (f.stabilize)
    sc = producer_scale<T>(f.partials + (size_t)f.stepOff[s] * f.R, f.stepP[s], f.stepSlots[s], f.stepNumel[s],
                           f.min_norm, r, &cond);
  if ((threadIdx.x & 63) == 0) {
    f.rescales[(size_t)r * f.n_steps + s] = cond ? (double)sc : 0.0;
    logs[(size_t)r * f.n_steps + s] = cond ? (double)log(sc) : 0.0;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_finalize(FinalArgs f, const double* __restrict__ logs) {
  __shared__ double red[4];
  const int r = blockIdx.y;
  if (blockIdx.x == 0) {
    double v = 0;
    for (int s = threadIdx.x; s < f.n_steps; s += 256) v += logs[(size_t)r * f.n_steps + s];
    const double tot = block_sum(v, red);
    if (threadIdx.x == 0) f.log_scale[r] = tot;
  }
  if (!f.stabilize || f.defer) return;
  const double rl = f.rescales[(size_t)r * f.n_steps + f.n_steps - 1];
  if (rl == 0.0) return;  // last step was not rescaled (norm <= min_norm): tensor unchanged
  const T s_last = (T)rl;
  T* out = (T*)f.ptrs[(size_t)r * f.n_tensors + f.id_out];
  constexpr int V = 16 / (int)sizeof(T);
  typedef T vecT __attribute__((ext_vector_type(V)));
  const int64_t nv = f.vec ? f.out_numel / V : 0;            // (a 4 GiB result: one read and one write of it, at full width)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
    vecT v = reinterpret_cast<vecT*>(out)[i];
#pragma unroll
    for (int j = 0; j < V; ++j) v[j] = v[j] / s_last;
    reinterpret_cast<vecT*>(out)[i] = v;
  }
  for (int64_t i = nv * V + (int64_t)blockIdx.x * 256 + threadIdx.x; i < f.out_numel; i += (int64_t)gridDim.x * 256)
    out[i] = out[i] / s_last;
}

// k_finish: the deferred form of that division with destabilize() (reference einsum.py:110-114) behind it in the same
// pass: out = (out / rescale_last) * mult[r], both roundings as the reference's two operations make them - the final
// tensor is read and written once instead of twice (ctn_exec_set_finish_mode / ctn_exec_finish).
template <typename T>
__global__ __launch_bounds__(256) void k_finish(void* const* __restrict__ ptrs, int n_tensors, int id_out, int64_t numel,
                                                const double* __restrict__ rescales, int n_steps, int stabilize,
                                                const double* __restrict__ mult, int vec) {
  const int r = blockIdx.y;
  const double rl = stabilize ? rescales[(size_t)r * n_steps + n_steps - 1] : 0.0;
  const bool div = rl != 0.0;
  const T s_last = div ? (T)rl : (T)1, m = (T)mult[r];
  T* out = (T*)ptrs[(size_t)r * n_tensors + id_out];
  constexpr int V = 16 / (int)sizeof(T);
  typedef T vecT __attribute__((ext_vector_type(V)));
  const int64_t nv = vec ? numel / V : 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
    vecT v = reinterpret_cast<vecT*>(out)[i];
#pragma unroll
    for (int j = 0; j < V; ++j) { const T q = div ? v[j] / s_last : v[j]; v[j] = q * m; }
    reinterpret_cast<vecT*>(out)[i] = v;
  }
  for (int64_t i = nv * V + (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
    const T q = div ? out[i] / s_last : out[i];
    out[i] = q * m;
  }
}

// ---------------------------------------------------------------------------
// k_renorm: the reference's stabilize() applied to a stored intermediate, literally (einsum.py:97-106:
// T <- T / rescale when sum|T| > min_norm).  Only the EAGER rescale mode launches it (engine.hip): there every
// intermediate is normalised in place right after its step and consumers take it with scale 1, so no product
// ever sees an un-normalised operand - what the lazy epilogue rescale of the tile kernels cannot promise when
// operand magnitudes are extreme (sA * sB * sum beyond the dtype's range).
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_renorm(void* const* __restrict__ ptrs, int n_tensors, int id, int64_t numel,
                                                const double* __restrict__ part, int P, int stride, double min_norm) {
  const int r = blockIdx.y;
  bool cond = false;
  const T sc = producer_scale<T>(part, P, stride, (double)numel, min_norm, r, &cond);
  if (!cond) return;
  T* __restrict__ x = (T*)ptrs[(size_t)r * n_tensors + id];
  constexpr int V = 16 / sizeof(T);
  typedef typename VecOf<T, V>::type VT;
  const int64_t nv = numel / V;            // workspace tensors start on 256-byte boundaries
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
    VT v = reinterpret_cast<VT*>(x)[i];
    T* e = reinterpret_cast<T*>(&v);
#pragma unroll
    for (int j = 0; j < V; ++j) e[j] = e[j] / sc;
    reinterpret_cast<VT*>(x)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(numel - nv * V)) x[nv * V + threadIdx.x] = x[nv * V + threadIdx.x] / sc;
}

// ---------------------------------------------------------------------------
// k_combine_split: sum of n tensors given in split format, without leaving it - the join of index slices
//  and of the ranks' partial results after the one all_gather:
//     sum_i T_i e^{c_i} = e^{c*} sum_i T_i e^{c_i - c*},  c* = max c_i over the parts that are not exact zeros,
// re-stabilised exactly like a contraction step (reference einsum.py:89-107: norm = sum|T|, rescale = norm / numel,
// applied iff norm > min_norm).  Part i is t + i * t_stride (numel elements of TIN), its scale c[i * c_stride];
// the result is PACKED: out[0 .. numel) = T_hat as doubles, out[numel] = c - the buffer that crosses the links.
// One workgroup (the payloads are small: a closed network's amplitude, a few thousand elements at most - larger
// open results take the reduce-scatter route of dist.SlicedContraction.run_device); every sum runs in a fixed
// order, so the result is bit-reproducible.  No part live: T_hat = 0, c = 0.
// ---------------------------------------------------------------------------
constexpr int kCombineMaxParts = 4096;
template <typename TIN>
__global__ __launch_bounds__(256) void k_combine_split(const TIN* __restrict__ t, int64_t t_stride,
                                                       const double* __restrict__ c, int64_t c_stride, int n,
                                                       int64_t numel, double min_norm, double* __restrict__ out) {
  __shared__ double red[4];
  __shared__ double w[kCombineMaxParts];     // e^{c_i - c*}, 0 for an exact-zero part
  __shared__ double cmax[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // liveness (any element != 0) and the largest scale among the live parts: one wave per part, lanes along it
  double best = -INFINITY;
  for (int i = wv; i < n; i += 4) {
    const TIN* __restrict__ p = t + (size_t)i * t_stride;
    bool nz = false;
    for (int64_t j = lane; j < numel && !nz; j += 64) nz = p[j] != (TIN)0;
    const bool live = __ballot(nz) != 0ull;
    const double ci = c[(size_t)i * c_stride];
    if (lane == 0) w[i] = live ? ci : -INFINITY;
    if (live) best = fmax(best, ci);
  }
  if (lane == 0) cmax[wv] = best;
  __syncthreads();
  const double c_star = fmax(fmax(cmax[0], cmax[1]), fmax(cmax[2], cmax[3]));
  if (c_star == -INFINITY) {               // every part is an exact zero
    for (int64_t j = threadIdx.x; j <= numel; j += 256) out[j] = 0.0;
    return;
  }
  for (int i = threadIdx.x; i < n; i += 256) w[i] = w[i] == -INFINITY ? 0.0 : exp(w[i] - c_star);
  __syncthreads();
  double absv = 0.0;
  for (int64_t j = threadIdx.x; j < numel; j += 256) {
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
      const double wi = w[i];
      if (wi != 0.0) acc += (double)t[(size_t)i * t_stride + j] * wi;
    }
    out[j] = acc;
    absv += fabs(acc);
  }
  const double norm = block_sum(absv, red);
  double c_out = c_star;
  if (norm > min_norm) {
    const double rescale = norm / (double)numel;
    for (int64_t j = threadIdx.x; j < numel; j += 256) out[j] = out[j] / rescale;   // own elements: no barrier needed
    c_out = c_star + log(rescale);
  }
  if (threadIdx.x == 0) out[numel] = c_out;
}

// ---------------------------------------------------------------------------
// Scale bookkeeping of a sliced contraction run in stages (dist.StagedSlicedContraction; no reference counterpart -
// the reference is a single process on one unsliced network): the results of a lower stage are operands of the stages
// above, and their log-scale registers ride along.
//
// k_scales_add: dst[i] = own[i] + sum_j kid_j[idx_j[i]], added left to right - the register of evaluation i of a stage
// plus those of the evaluations of the stages below that it consumed.
// ---------------------------------------------------------------------------
constexpr int kScalesAddMaxKids = 8;
struct ScalesAddArgs {
  double* dst;
  const double* own;
  int32_t n, n_kids;
  const double* kid[kScalesAddMaxKids];
  const int64_t* idx[kScalesAddMaxKids];
};
__global__ __launch_bounds__(256) void k_scales_add(ScalesAddArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  double v = a.own[i];
  for (int j = 0; j < a.n_kids; ++j) v += a.kid[j][a.idx[j][i]];
  a.dst[i] = v;
}

// k_merge_live / k_merge_scale: the n evaluations of a stage lie on a grid over the sliced labels the stage depends on
// (`ext[0..ndim)`, row-major); along the axes flagged in `merged` they are about to become ONE strided operand of the
// root stage, so every group of evaluations that differ only along those axes is brought to its common scale - the
// largest register among the members that are not exact zeros (the liveness rule of k_combine_split: the WHOLE tensor
// is looked at; a live tensor is recognised by whichever workgroup sees its first non-zero, an all-zero one costs a
// full scan) - by multiplying member i with e^{c_i - top}; the new register of every member is `top` (0 when the whole
// group is zero).  Exact zeros are left alone.  `cum_new` is a separate array: members of a group read each other's
// old registers.
template <typename T>
__global__ __launch_bounds__(256) void k_merge_live(const T* __restrict__ buf, int64_t stride, int64_t numel, int64_t chunk,
                                                    int32_t* __restrict__ flags) {
  const int i = blockIdx.y;
  if (__builtin_nontemporal_load(flags + i)) return;      // somebody has already found a non-zero
  const T* __restrict__ p = buf + (size_t)i * stride;
  const int64_t j0 = (int64_t)blockIdx.x * chunk, j1 = min(numel, j0 + chunk);
  bool nz = false;
  for (int64_t j = j0 + threadIdx.x; j < j1; j += 256) {
    nz = p[j] != (T)0;
    if (__ballot(nz) != 0ull) { nz = true; break; }
  }
  if (nz && (threadIdx.x & 63) == 0) flags[i] = 1;
}

constexpr int kMergeMaxDims = 8;
struct MergeArgs {
  void* buf;
  int64_t stride, numel, chunk;
  const double* cum;
  double* cum_new;
  const int32_t* flags;
  int32_t n, ndim;
  int32_t ext[kMergeMaxDims], merged[kMergeMaxDims];
};
template <typename T>
__global__ __launch_bounds__(256) void k_merge_scale(MergeArgs a) {
  __shared__ double s_top;
  const int i = blockIdx.y;
  if (threadIdx.x == 0) {
    int coord[kMergeMaxDims], rem = i;
    for (int d = a.ndim - 1; d >= 0; --d) { coord[d] = rem % a.ext[d]; rem /= a.ext[d]; }
    int members = 1;
    for (int d = 0; d < a.ndim; ++d) if (a.merged[d]) members *= a.ext[d];
    double top = -INFINITY;
    for (int mbr = 0; mbr < members; ++mbr) {
      int r2 = mbr, idx = 0;
      int c2[kMergeMaxDims];
      for (int d = a.ndim - 1; d >= 0; --d) {
        if (a.merged[d]) { c2[d] = r2 % a.ext[d]; r2 /= a.ext[d]; } else c2[d] = coord[d];
      }
      for (int d = 0; d < a.ndim; ++d) idx = idx * a.ext[d] + c2[d];
      if (a.flags[idx]) top = fmax(top, a.cum[idx]);
    }
    s_top = top == -INFINITY ? 0.0 : top;
    if (blockIdx.x == 0) a.cum_new[i] = s_top;
  }
  __syncthreads();
  if (!a.flags[i]) return;                                   // an exact zero stays one
  const T f = (T)exp(a.cum[i] - s_top);
  if (f == (T)1) return;
  T* __restrict__ p = (T*)a.buf + (size_t)i * a.stride;
  const int64_t j0 = (int64_t)blockIdx.x * a.chunk, j1 = min(a.numel, j0 + a.chunk);
  constexpr int V = 16 / (int)sizeof(T);                     // (stride and chunk are multiples of V: 16-byte accesses)
  typedef T vecT __attribute__((ext_vector_type(V)));
  const int64_t jv = j0 + (j1 - j0) / V * V;
  for (int64_t j = j0 + (int64_t)threadIdx.x * V; j < jv; j += 256 * V) {
    vecT v = *reinterpret_cast<vecT*>(p + j);
    v *= f;
    *reinterpret_cast<vecT*>(p + j) = v;
  }
  for (int64_t j = jv + threadIdx.x; j < j1; j += 256) p[j] *= f;
}

// ---------------------------------------------------------------------------
// K-chain: persistent small-tensor DAG walker.  One workgroup per replica executes EVERY step of
// the plan in order (reference loop einsum.py:341-391) - no per-step launch, rescale factors of
// all produced tensors kept in LDS.  Same arithmetic as k_element (operands divided by their
// producer's rescale on load), so results are bit-identical to the per-step path.
// ---------------------------------------------------------------------------
struct ChainStep {
  const int64_t *obA, *obB, *obC;
  const int32_t *omA, *omC, *onB, *onC, *okA, *okB;
  double numelC;
  int32_t Bt, M, N, K;
  int32_t idA, idB, idC;
  int32_t prodA, prodB;  // producing step of each operand, -1 for inputs
};

template <typename T>
__global__ __launch_bounds__(256) void k_chain(const ChainStep* __restrict__ steps, int n_steps,
                                               void* const* ptrs, int n_tensors, double* partials,
                                               int R, double min_norm, int stabilize) {
  constexpr int kStage = 24576 / sizeof(T);   // operand elements staged per step (24 KB; sc[] takes up to 32)
  __shared__ double red[4];
  __shared__ T sc[kChainMaxSteps];
  __shared__ T stage[kStage];
  const int r = blockIdx.x;
  void* const* tp = ptrs + (size_t)r * n_tensors;
  for (int s = 0; s < n_steps; ++s) {
    const ChainStep d = steps[s];
    const T sA = d.prodA >= 0 ? sc[d.prodA] : (T)1;
    const T sB = d.prodB >= 0 ? sc[d.prodB] : (T)1;
    const T* __restrict__ A = (const T*)tp[d.idA];
    const T* __restrict__ B = (const T*)tp[d.idB];
    T* __restrict__ C = (T*)tp[d.idC];
    const int total = d.Bt * d.M * d.N;
    const int na = d.Bt * d.M * d.K, nb = d.Bt * d.N * d.K;
    double absv = 0;
    if (d.K >= 4 && na + nb <= kStage) {   // a real K loop to amortise the staging pass
      // normalise each operand element ONCE into dense LDS images ([b][m][k] and [b][k][n]); the MAC loop
      // then has neither divisions nor table look-ups (same operations in the same order as below)
      T* la = stage;
      T* lb = stage + na;
      for (int i = threadIdx.x; i < na; i += 256) {
        const int k = i % d.K, q = i / d.K;
        la[i] = A[d.obA[q / d.M] + d.omA[q % d.M] + d.okA[k]] / sA;
      }
      for (int i = threadIdx.x; i < nb; i += 256) {
        const int n = i % d.N, q = i / d.N;
        lb[i] = B[d.obB[q / d.K] + d.onB[n] + d.okB[q % d.K]] / sB;
      }
      __syncthreads();
      for (int o = threadIdx.x; o < total; o += 256) {
        const int n = o % d.N;
        const int q = o / d.N;
        const int m = q % d.M;
        const int b = q / d.M;
        const T* pa = la + (b * d.M + m) * d.K;
        const T* pb = lb + b * d.K * d.N + n;
        T acc = 0;
        for (int k = 0; k < d.K; ++k) acc = fma(pa[k], pb[k * d.N], acc);
        C[d.obC[b] + d.omC[m] + d.onC[n]] = acc;
        absv += (double)fabs(acc);
      }
    } else {
      for (int o = threadIdx.x; o < total; o += 256) {
        const int n = o % d.N;
        const int q = o / d.N;
        const int m = q % d.M;
        const int b = q / d.M;
        const T* pa = A + d.obA[b] + d.omA[m];
        const T* pb = B + d.obB[b] + d.onB[n];
        T acc = 0;
        for (int k = 0; k < d.K; ++k) acc = fma(pa[d.okA[k]] / sA, pb[d.okB[k]] / sB, acc);
        C[d.obC[b] + d.omC[m] + d.onC[n]] = acc;
        absv += (double)fabs(acc);
      }
    }
    // the barriers inside block_sum also order this step's stores before the next step's loads
    const double tot = block_sum(absv, red);
    if (threadIdx.x == 0) {
      partials[(size_t)s * R + r] = tot;   // chain plans: one slot per step and replica
      const T norm = (T)tot;
      sc[s] = (stabilize && norm > (T)min_norm) ? norm / (T)d.numelC : (T)1;
    }
    __syncthreads();
  }
}


}  // namespace ctn
