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
  const int32_t *obA, *obB, *obC, *omA, *omC, *onB, *onC, *okA, *okB;
  void* const* ptrs;    // [R][n_tensors] base pointer of every tensor of every replica
  const double* partA;  // [R][64] abs-sum partials of A's producer step, nullptr for inputs
  const double* partB;
  double* partC;        // [R][partC_stride] where this step's partials go
  double numelA, numelB;
  double min_norm;
  int32_t Bt, M, N, K;
  int32_t idA, idB, idC, n_tensors;
  int32_t PA, PB;
  int32_t partC_stride;
  int32_t tiles_m, tiles_n;
  int32_t blocks_per_replica;
  int32_t R;
  int32_t c_vec;  // float4 stores of C allowed
  // streaming kernels: output index = (hi, lo, n); n along C's unit-stride label
  const int32_t *ohA, *ohB, *ohC, *olA, *olB, *olC;
  int32_t H, L, Nv, sAn, sBn;
  FastDiv dNq, dL, dNv;  // divisors: vectors per row, lo extent, n extent
  unsigned long long* dbg;  // CTN_STAMPS builds only: 4 cycle stamps per MFMA tile (else unused, null)
};

struct FinalArgs {
  void* const* ptrs;
  const double* partials;   // [n_steps][R][64]
  const int32_t* stepP;     // [n_steps] partial count of each step
  const double* stepNumel;  // [n_steps] numel of each step's output
  double* log_scale;        // [R]
  double* rescales;         // [R][n_steps]
  double min_norm;
  int64_t out_numel;
  int32_t n_steps, R, id_out, n_tensors, stabilize;
};

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
// Rescale factor of a tensor from its producer's partial sums (reference
// einsum.py:97-102: norm = sum|T|, rescale = norm / numel, applied iff
// norm > min_norm).  All 64 lanes of the calling wave must be active.
template <typename T>
__device__ __forceinline__ T producer_scale(const double* part, int P, double numel, double min_norm,
                                            int r, bool* cond_out = nullptr) {
  if (part == nullptr) {
    if (cond_out) *cond_out = false;
    return (T)1;
  }
  const int lane = threadIdx.x & 63;
  double v = lane < P ? part[(size_t)r * kMaxPartials + lane] : 0.0;
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
    } else {
      typedef typename VecOf<T, V>::type VT;
      const VT x = *reinterpret_cast<const VT*>(p);
      const T* e = reinterpret_cast<const T*>(&x);
#pragma unroll
      for (int v = 0; v < V; ++v) out[v] = e[v];
    }
  }
}

template <typename T, int V, int U>
__global__ __launch_bounds__(256) void k_stream(StepArgs a) {
  __shared__ double red[4];
  const int r = blockIdx.y;
  const T sA = producer_scale<T>(a.partA, a.PA, a.numelA, a.min_norm, r);
  const T sB = producer_scale<T>(a.partB, a.PB, a.numelB, a.min_norm, r);
  const bool divA = sA != (T)1, divB = sB != (T)1;
  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const T* __restrict__ A = (const T*)tp[a.idA];
  const T* __restrict__ B = (const T*)tp[a.idB];
  T* __restrict__ C = (T*)tp[a.idC];
  // one item = U vectors of V elements of one output row (hi, lo): columns c, c + S, ..., c + (U-1) S
  // with S = vectors per row / U, so the row's table lookups and broadcast operands are paid once
  // per U*V outputs while every store instruction of a wave still covers a contiguous segment.
  const uint32_t nq_per = (uint32_t)((a.Nv + V - 1) / V);
  const uint32_t S = nq_per / U;                 // U divides nq_per (checked on the host)
  const uint32_t items = (uint32_t)a.H * (uint32_t)a.L * S;  // < 2^31
  const FastDiv dq = a.dNq;                      // divisor S
  const bool kone = a.K == 1;                    // pure product: k-offset tables hold a single 0
  double absv = 0;
  const uint32_t stride = gridDim.x * 256u;
  for (uint32_t it = blockIdx.x * 256u + threadIdx.x; it < items; it += stride) {
    const uint32_t row = dq.div(it);
    const int c0 = (int)(it - row * S) * V;
    const int h = (int)a.dL.div(row);
    const int l = (int)(row - (uint32_t)h * (uint32_t)a.L);
    const T* pa = A + a.ohA[h] + a.olA[l] + c0 * a.sAn;
    const T* pb = B + a.ohB[h] + a.olB[l] + c0 * a.sBn;
    T* pc = C + (size_t)row * a.Nv + c0;         // C is contiguous in (hi, lo, n) order
    T acc[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < V; ++v) acc[u][v] = 0;
    const int stepA = (int)S * V * a.sAn, stepB = (int)S * V * a.sBn;
    for (int k = 0; k < a.K; ++k) {
      const int ka = kone ? 0 : a.okA[k], kb = kone ? 0 : a.okB[k];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        T av[V], bv[V];
        load_vec<T, V>(pa + ka + u * stepA, a.sAn, av);
        load_vec<T, V>(pb + kb + u * stepB, a.sBn, bv);
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const T x = divA ? av[v] / sA : av[v];
          const T y = divB ? bv[v] / sB : bv[v];
          acc[u][v] = fma(x, y, acc[u][v]);
        }
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
  const T sA = producer_scale<T>(a.partA, a.PA, a.numelA, a.min_norm, r);
  const T sB = producer_scale<T>(a.partB, a.PB, a.numelB, a.min_norm, r);
  const bool divA = sA != (T)1, divB = sB != (T)1;
  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const T* __restrict__ A = (const T*)tp[a.idA];
  const T* __restrict__ B = (const T*)tp[a.idB];
  T* __restrict__ C = (T*)tp[a.idC];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t outs = (uint32_t)a.H * (uint32_t)a.L * (uint32_t)a.Nv;
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
  const T sA = producer_scale<T>(a.partA, a.PA, a.numelA, a.min_norm, r);
  const T sB = producer_scale<T>(a.partB, a.PB, a.numelB, a.min_norm, r);
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

// Collapse > 64 per-workgroup partials into one, in a fixed order.
__global__ __launch_bounds__(256) void k_collapse(const double* scratch, int blocks, double* part) {
  __shared__ double red[4];
  const int r = blockIdx.x;
  const double* src = scratch + (size_t)r * blocks;
  double v = 0;
  for (int i = threadIdx.x; i < blocks; i += 256) v += src[i];
  const double tot = block_sum(v, red);
  if (threadIdx.x == 0) part[(size_t)r * kMaxPartials] = tot;
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
    sc = producer_scale<T>(f.partials + (size_t)s * f.R * kMaxPartials, f.stepP[s], f.stepNumel[s],
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
  if (!f.stabilize) return;
  const double rl = f.rescales[(size_t)r * f.n_steps + f.n_steps - 1];
  if (rl == 0.0) return;  // last step was not rescaled (norm <= min_norm): tensor unchanged
  const T s_last = (T)rl;
  T* out = (T*)f.ptrs[(size_t)r * f.n_tensors + f.id_out];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < f.out_numel; i += (int64_t)gridDim.x * 256)
    out[i] = out[i] / s_last;
}

// ---------------------------------------------------------------------------
// K-mfma-f32: 128 x TN workgroup tile (TN = 128 or 64), 4 waves (2x2), each wave
// 64 x TN/2 = 2 x TN/64 v_mfma_f32_32x32x2_f32 accumulators, register-staged
// double-buffered LDS, one barrier per k-tile.  TN = 64 serves skinny products
// (boundary absorptions of 2D grids: N = 64) where a 128-wide tile would be half masked.
//
// MODE (per operand): 0 scalar gather, 1 float4 along the free index (LDS image
// [k][rows]), 2 float4 along k (LDS image [rows][BK+1], odd row length => conflict-free
// ds_read_b32 for the MFMA fragment: lane l reads row l&31, k = 2*kk + (l>>5)).
//
// Latency structure: the k-offset table entries of tile t+2 are requested while
// tile t+1's data loads are in flight and tile t is being multiplied, so no
// load ever waits on a table lookup; global loads are unconditional (padded
// tables keep every address in bounds) and masked when written to LDS; LDS fragment
// reads run one k-step ahead of the MFMAs that consume them.
// ---------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = kTileM;

// Stages one ROWS x BK operand tile per k-step: global -> registers -> LDS.
template <int MODE, int BK, int ROWS>
struct TileLoader {
  static constexpr int NV = ROWS * BK / 256;               // floats staged per thread
  static constexpr int VPR = ROWS / 4;                     // mode 1: float4 per k-row
  static constexpr int RPP = 256 / VPR;                    // mode 1: k-rows covered per pass
  static constexpr int KPP = 256 / ROWS;                   // mode 0: k-rows covered per pass
  static constexpr int NT = MODE == 1 ? BK / RPP : (MODE == 2 ? 1 : NV);  // table entries per tile
  static constexpr int NM = MODE == 2 ? NV / 4 : 1;        // hoisted free-index offsets
  static constexpr int LDK = BK + 1;
  static constexpr int kSize = MODE == 2 ? ROWS * LDK : BK * ROWS;
  static_assert(NV >= 4 && NT >= 1 && NM >= 1, "tile too small for 256 threads");

  float v[NV];
  int kofs[NT];   // k-offset table entries of the NEXT tile to load
  int offm[NM];
  bool okm[NM];

  __device__ __forceinline__ void init(const int32_t* __restrict__ om, int m0, int M, int tid) {
    if (MODE == 1) {
      const int gm = m0 + (tid % VPR) * 4;
      offm[0] = om[gm];
      okm[0] = gm < M;
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        const int gm = m0 + ((tid + i * 256) / (BK / 4));
        offm[i] = om[gm];
        okm[i] = gm < M;
      }
    } else {
      const int gm = m0 + (tid % ROWS);
      offm[0] = om[gm];
      okm[0] = gm < M;
    }
  }

  // request the table entries this thread needs for the tile starting at k0 (table is padded)
  __device__ __forceinline__ void tab(const int32_t* __restrict__ ok, int k0, int tid) {
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < NT; ++i) kofs[i] = ok[k0 + tid / VPR + RPP * i];
    } else if (MODE == 2) {
      kofs[0] = ok[k0 + (tid % (BK / 4)) * 4];
    } else {
#pragma unroll
      for (int i = 0; i < NT; ++i) kofs[i] = ok[k0 + tid / ROWS + KPP * i];
    }
  }

  // issue the global loads of the tile using the entries fetched by the previous tab().
  // Unconditional: padded tables keep every address inside the tensor; out-of-range rows and
  // k are zeroed later, in store(), so nothing here waits on the data.
  __device__ __forceinline__ void load(const float* __restrict__ base) {
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const float4 x = *reinterpret_cast<const float4*>(base + offm[0] + kofs[i]);
        v[4 * i + 0] = x.x; v[4 * i + 1] = x.y; v[4 * i + 2] = x.z; v[4 * i + 3] = x.w;
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        const float4 x = *reinterpret_cast<const float4*>(base + offm[i] + kofs[0]);
        v[4 * i + 0] = x.x; v[4 * i + 1] = x.y; v[4 * i + 2] = x.z; v[4 * i + 3] = x.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NT; ++i) v[i] = base[offm[0] + kofs[i]];
    }
  }

  // write the staged tile (loaded from k0) into its LDS image; FULL skips the bounds masks
  template <bool FULL>
  __device__ __forceinline__ void store(float* __restrict__ s, int k0, int K, int tid) const {
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int kr = tid / VPR + RPP * i;
        const bool in = FULL || (okm[0] && (k0 + kr) < K);
        *reinterpret_cast<float4*>(s + kr * ROWS + (tid % VPR) * 4) =
            make_float4(in ? v[4 * i] : 0.f, in ? v[4 * i + 1] : 0.f, in ? v[4 * i + 2] : 0.f,
                        in ? v[4 * i + 3] : 0.f);
      }
    } else if (MODE == 2) {
      const bool kin = (k0 + (tid % (BK / 4)) * 4) < K;
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        const bool in = FULL || (okm[i] && kin);
        float* d = s + ((tid + i * 256) / (BK / 4)) * LDK + (tid % (BK / 4)) * 4;
        d[0] = in ? v[4 * i] : 0.f; d[1] = in ? v[4 * i + 1] : 0.f;
        d[2] = in ? v[4 * i + 2] : 0.f; d[3] = in ? v[4 * i + 3] : 0.f;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int kr = tid / ROWS + KPP * i;
        const bool in = FULL || (okm[0] && (k0 + kr) < K);
        s[kr * ROWS + (tid % ROWS)] = in ? v[i] : 0.f;
      }
    }
  }

  // LDS index of element (row, k) of the tile image
  static __device__ __forceinline__ int idx(int row, int k) {
    return MODE == 2 ? row * LDK + k : k * ROWS + row;
  }
};

template <int MA, int MB, int BK, int TN, bool FULL>
__device__ __forceinline__ void mfma_mainloop(TileLoader<MA, BK, BM>& la, TileLoader<MB, BK, TN>& lb,
                                              const float* __restrict__ A, const float* __restrict__ B,
                                              const int32_t* __restrict__ okA, const int32_t* __restrict__ okB,
                                              int K, float* sA, float* sB, f32x16 (&acc)[2][TN / 64], int tid,
                                              unsigned long long* dbg1) {
  using LA = TileLoader<MA, BK, BM>;
  using LB = TileLoader<MB, BK, TN>;
  constexpr int SZA = LA::kSize, SZB = LB::kSize;
  constexpr int NJ = TN / 64;  // 32-wide column blocks per wave
  const int lane = tid & 63, w = tid >> 6;
  const int wm = (w >> 1) * 64, wn = (w & 1) * (TN / 2);
  const int l31 = lane & 31, h = lane >> 5;

  const int nkt = (K + BK - 1) / BK;
  la.tab(okA, 0, tid);
  lb.tab(okB, 0, tid);
  la.load(A);
  lb.load(B);
  la.tab(okA, BK, tid);
  lb.tab(okB, BK, tid);
  la.template store<FULL>(sA, 0, K, tid);
  lb.template store<FULL>(sB, 0, K, tid);
  __syncthreads();
#ifdef CTN_STAMPS
  if (dbg1 && tid == 0) *dbg1 = __builtin_amdgcn_s_memtime();
#endif

  // per-lane LDS fragment bases (element indices)
  const int fa0 = LA::idx(wm + l31, h), fa1 = LA::idx(wm + 32 + l31, h);
  int fbx[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) fbx[j] = LB::idx(wn + j * 32 + l31, h);
  constexpr int stepA = MA == 2 ? 2 : 2 * BM;  // advance of the fragment index per k-step (k += 2)
  constexpr int stepB = MB == 2 ? 2 : 2 * TN;

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nkt;
    if (more) {
      la.load(A);
      lb.load(B);
      la.tab(okA, (kt + 2) * BK, tid);
      lb.tab(okB, (kt + 2) * BK, tid);
    }
    __builtin_amdgcn_sched_barrier(0);  // global loads stay in front of the MFMA phase
    const float* cA = sA + cur * SZA;
    const float* cB = sB + cur * SZB;
    float fa[2][2], fb[2][NJ];
    fa[0][0] = cA[fa0]; fa[0][1] = cA[fa1];
#pragma unroll
    for (int j = 0; j < NJ; ++j) fb[0][j] = cB[fbx[j]];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int c = kk & 1, nx = c ^ 1;
      if (kk + 1 < BK / 2) {
        fa[nx][0] = cA[fa0 + (kk + 1) * stepA]; fa[nx][1] = cA[fa1 + (kk + 1) * stepA];
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[nx][j] = cB[fbx[j] + (kk + 1) * stepB];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i], fb[c][j], acc[i][j], 0, 0, 0);
      // pin the interleave: the LDS reads of step kk+1 issue ahead of the MFMAs of step kk
      __builtin_amdgcn_sched_group_barrier(0x100, 2 + NJ, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NJ, 0);
    }
    __builtin_amdgcn_sched_barrier(0);  // the staged tile is consumed only after the MFMA phase
    if (more) {
      la.template store<FULL>(sA + (cur ^ 1) * SZA, (kt + 1) * BK, K, tid);
      lb.template store<FULL>(sB + (cur ^ 1) * SZB, (kt + 1) * BK, K, tid);
    }
    __syncthreads();
  }
}

// second launch-bound argument = waves per SIMD the register allocator must leave room for:
// BK = 16 is sized for 3 workgroups per CU (<= 168 registers), BK = 32 for 2
template <int MA, int MB, int BK, int TN>
__global__ __launch_bounds__(256, (BK == 16 ? 3 : 2)) void k_mfma_f32(StepArgs a) {
  using LA = TileLoader<MA, BK, BM>;
  using LB = TileLoader<MB, BK, TN>;
  constexpr int SZA = LA::kSize, SZB = LB::kSize;
  constexpr int NJ = TN / 64;
  // one LDS object: [A buf0][A buf1][B buf0][B buf1][omC 128][onC TN][red 4 doubles]
  __shared__ __attribute__((aligned(16))) float smem[2 * SZA + 2 * SZB + BM + TN + 8];
  float* sA = smem;
  float* sB = smem + 2 * SZA;
  int* s_omC = reinterpret_cast<int*>(smem + 2 * SZA + 2 * SZB);
  int* s_onC = s_omC + BM;
  double* red = reinterpret_cast<double*>(s_onC + TN);

  const int tid = threadIdx.x;
  // XCD-aware remap: workgroups are dealt round-robin over the 8 XCDs, so give each
  // XCD a contiguous range of tiles (one replica's tiles share that XCD's L2).
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3, q8 = nwg >> 3, r8 = nwg & 7;
  const int pid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const int r = pid / a.blocks_per_replica;
  const int t = pid - r * a.blocks_per_replica;
  const int tiles_mn = a.tiles_m * a.tiles_n;
  const int b = t / tiles_mn;
  const int tt = t - b * tiles_mn;
  const int m0 = (tt / a.tiles_n) * BM;
  const int n0 = (tt % a.tiles_n) * TN;

#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) a.dbg[(size_t)pid * 4 + 0] = __builtin_amdgcn_s_memtime();
#endif
  const float scA = producer_scale<float>(a.partA, a.PA, a.numelA, a.min_norm, r);
  const float scB = producer_scale<float>(a.partB, a.PB, a.numelB, a.min_norm, r);

  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const float* __restrict__ A = (const float*)tp[a.idA] + a.obA[b];
  const float* __restrict__ B = (const float*)tp[a.idB] + a.obB[b];
  float* __restrict__ C = (float*)tp[a.idC] + a.obC[b];

  if (tid < BM) s_omC[tid] = a.omC[m0 + tid];
  else if (tid - BM < TN) s_onC[tid - BM] = a.onC[n0 + tid - BM];

  LA la;
  LB lb;
  la.init(a.omA, m0, a.M, tid);
  lb.init(a.onB, n0, a.N, tid);

  const int lane = tid & 63, w = tid >> 6;
  const int wm = (w >> 1) * 64, wn = (w & 1) * (TN / 2);
  const int l31 = lane & 31, h = lane >> 5;

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#ifdef CTN_STAMPS
  unsigned long long* stamp1 = a.dbg ? a.dbg + (size_t)pid * 4 + 1 : nullptr;
#else
  unsigned long long* stamp1 = nullptr;
#endif
  // FULL: the tile lies completely inside M x N and K is a multiple of BK -> no masking
  const bool full = (m0 + BM <= a.M) && (n0 + TN <= a.N) && (a.K % BK == 0);
  if (full) mfma_mainloop<MA, MB, BK, TN, true>(la, lb, A, B, a.okA, a.okB, a.K, sA, sB, acc, tid, stamp1);
  else mfma_mainloop<MA, MB, BK, TN, false>(la, lb, A, B, a.okA, a.okB, a.K, sA, sB, acc, tid, stamp1);
#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) a.dbg[(size_t)pid * 4 + 2] = __builtin_amdgcn_s_memtime();
#endif

  // epilogue: lazy rescale, store through the C offset tables, abs-sum partial
  const float iA = 1.0f / scA, iB = 1.0f / scB;
  float asum = 0.f;
  {
    // Each WAVE stages its own 64 x TN/2 accumulator block through its quarter of the (now idle)
    // operand buffers, 32 rows at a time, and stores whole 16-byte row segments (4-8 rows of
    // 128-256 contiguous bytes per store instruction).  No workgroup barrier is involved: LDS
    // operations of one wave execute in order, so the write -> read hand-off is wave-local.
    constexpr int WT = TN / 2;                       // columns owned by a wave
    constexpr int LDSW = ((2 * SZA + 2 * SZB) / 4) & ~3;  // floats of LDS per wave (16-byte aligned)
    static_assert(LDSW >= 32 * WT, "per-wave staging area too small");
    constexpr int VW = WT / 4;                       // 16-byte vectors per row
    constexpr int RPI = 64 / VW;                     // rows covered by one wave-wide vector access
    float* wC = smem + w * LDSW;                     // [32][WT]
    const int c4 = (lane % VW) * 4;
    const int gcol = wn + c4;
    const bool cin = n0 + gcol < a.N;  // N % 4 == 0 whenever c_vec, otherwise checked per element
    const int offn = s_onC[gcol];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          wC[((e & 3) + 8 * (e >> 2) + 4 * h) * WT + j * 32 + l31] = (acc[i][j][e] * iA) * iB;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 32 / RPI; ++it) {
        const int lrow = it * RPI + lane / VW;
        const int row = wm + i * 32 + lrow;
        const float4 v = *reinterpret_cast<const float4*>(wC + lrow * WT + c4);
        if (m0 + row < a.M && cin) {
          float* dst = C + s_omC[row];
          if (a.c_vec) {
            *reinterpret_cast<float4*>(dst + offn) = v;
            asum += (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w));
          } else {
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (n0 + gcol + q < a.N) {
                dst[s_onC[gcol + q]] = vv[q];
                asum += fabsf(vv[q]);
              }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  const double tot = block_sum((double)asum, red);
  if (tid == 0) a.partC[(size_t)r * a.partC_stride + t] = tot;
#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) a.dbg[(size_t)pid * 4 + 3] = __builtin_amdgcn_s_memtime();
#endif
}

// ---------------------------------------------------------------------------
// K-mfma-f64: 64x64 workgroup tile, 4 waves (2x2), each wave 32x32 = 2x2
// v_mfma_f64_16x16x4_f64 accumulators, BK = 16, table-driven gather loads coalesced along the
// free index, LDS image [k][80] (row stride = 640 B = 32 banks mod 64: the two k rows a 32-lane
// group reads fall on disjoint bank halves).  f64 C/D map (NOT the f32 one): col = lane & 15,
// row = (lane >> 4) + 4 * reg.
// ---------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_mfma_f64(StepArgs a) {
  constexpr int T64 = kTile64, BK = 16, LD = 80, SZ = BK * LD;
  __shared__ __attribute__((aligned(16))) double smem[4 * SZ + 8];
  __shared__ int s_omC[T64], s_onC[T64];
  double* sA = smem;
  double* sB = smem + 2 * SZ;
  double* red = smem + 4 * SZ;

  const int tid = threadIdx.x;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3, q8 = nwg >> 3, r8 = nwg & 7;
  const int pid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const int r = pid / a.blocks_per_replica;
  const int t = pid - r * a.blocks_per_replica;
  const int tiles_mn = a.tiles_m * a.tiles_n;
  const int b = t / tiles_mn;
  const int tt = t - b * tiles_mn;
  const int m0 = (tt / a.tiles_n) * T64;
  const int n0 = (tt % a.tiles_n) * T64;

  const double scA = producer_scale<double>(a.partA, a.PA, a.numelA, a.min_norm, r);
  const double scB = producer_scale<double>(a.partB, a.PB, a.numelB, a.min_norm, r);
  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const double* __restrict__ A = (const double*)tp[a.idA] + a.obA[b];
  const double* __restrict__ B = (const double*)tp[a.idB] + a.obB[b];
  double* __restrict__ C = (double*)tp[a.idC] + a.obC[b];

  if (tid < T64) s_omC[tid] = a.omC[m0 + tid];
  else if (tid < 2 * T64) s_onC[tid - T64] = a.onC[n0 + tid - T64];

  // staging: element (free = tid & 63, k = (tid >> 6) + 4 i), i < 4, for both operands
  const int fr = tid & 63, kr = tid >> 6;
  const int offa = a.omA[m0 + fr], offb = a.onB[n0 + fr];
  const bool ina = m0 + fr < a.M, inb = n0 + fr < a.N;
  double va[4], vb[4];
  int ka[4], kb[4];
  auto tab = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { ka[i] = a.okA[k0 + kr + 4 * i]; kb[i] = a.okB[k0 + kr + 4 * i]; }
  };
  auto load = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) { va[i] = A[offa + ka[i]]; vb[i] = B[offb + kb[i]]; }
  };
  auto store = [&](double* dA, double* dB, int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool kin = k0 + kr + 4 * i < a.K;
      dA[(kr + 4 * i) * LD + fr] = (ina && kin) ? va[i] : 0.0;
      dB[(kr + 4 * i) * LD + fr] = (inb && kin) ? vb[i] : 0.0;
    }
  };

  const int lane = tid & 63, w = tid >> 6;
  const int wm = (w >> 1) * 32, wn = (w & 1) * 32;
  const int l15 = lane & 15, q = lane >> 4;
  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0;

  const int nkt = (a.K + BK - 1) / BK;
  tab(0);
  load();
  tab(BK);
  store(sA, sB, 0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nkt;
    if (more) {
      load();
      tab((kt + 2) * BK);
    }
    __builtin_amdgcn_sched_barrier(0);
    const double* cA = sA + cur * SZ;
    const double* cB = sB + cur * SZ;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const int k = kk * 4 + q;
      const double a0 = cA[k * LD + wm + l15], a1 = cA[k * LD + wm + 16 + l15];
      const double b0 = cB[k * LD + wn + l15], b1 = cB[k * LD + wn + 16 + l15];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) store(sA + (cur ^ 1) * SZ, sB + (cur ^ 1) * SZ, (kt + 1) * BK);
    __syncthreads();
  }

  // epilogue: operands' rescale factors divide the accumulator (division, as in the reference)
  double asum = 0.0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wn + j * 16 + l15;
      const bool cin = n0 + col < a.N;
      const int offn = s_onC[col];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = wm + i * 16 + q + 4 * e;
        if (cin && m0 + row < a.M) {
          const double v = (acc[i][j][e] / scA) / scB;
          C[s_omC[row] + offn] = v;
          asum += fabs(v);
        }
      }
    }
  const double tot = block_sum(asum, red);
  if (tid == 0) a.partC[(size_t)r * a.partC_stride + t] = tot;
}

// ---------------------------------------------------------------------------
// K-chain: persistent small-tensor DAG walker.  One workgroup per replica executes EVERY step of
// the plan in order (reference loop einsum.py:341-391) - no per-step launch, rescale factors of
// all produced tensors kept in LDS.  Same arithmetic as k_element (operands divided by their
// producer's rescale on load), so results are bit-identical to the per-step path.
// ---------------------------------------------------------------------------
struct ChainStep {
  const int32_t *obA, *obB, *obC, *omA, *omC, *onB, *onC, *okA, *okB;
  double numelC;
  int32_t Bt, M, N, K;
  int32_t idA, idB, idC;
  int32_t prodA, prodB;  // producing step of each operand, -1 for inputs
};

template <typename T>
__global__ __launch_bounds__(256) void k_chain(const ChainStep* __restrict__ steps, int n_steps,
                                               void* const* ptrs, int n_tensors, double* partials,
                                               int R, double min_norm, int stabilize) {
  __shared__ double red[4];
  __shared__ T sc[kChainMaxSteps];
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
    double absv = 0;
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
    // the barriers inside block_sum also order this step's stores before the next step's loads
    const double tot = block_sum(absv, red);
    if (threadIdx.x == 0) {
      partials[((size_t)s * R + r) * kMaxPartials] = tot;
      const T norm = (T)tot;
      sc[s] = (stabilize && norm > (T)min_norm) ? norm / (T)d.numelC : (T)1;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// executor
// ---------------------------------------------------------------------------
thread_local std::string g_err;

#define HIPCHECK(expr)                                                                   \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                         \
      return CTN_HIP_ERROR;                                                              \
    }                                                                                    \
  } while (0)

struct Exec {
  const Plan* plan = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int R = 1;
  int n_tensors = 0;
  int n_cu = 256;
  char* d_ws = nullptr;
  int32_t* d_tables = nullptr;
  void** d_ptrs = nullptr;
  std::vector<void*> h_ptrs;
  bool ptrs_valid = false;
  double* d_partials = nullptr;
  double* d_scratch = nullptr;
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
  int timing_slots = 0;             // 0 = timing off
  int timing_runs = 0;              // enqueues recorded since timing was enabled
  std::vector<hipEvent_t> events;   // [slot][step][2]

  ~Exec() {
    (void)hipSetDevice(device);
    for (void* p : {(void*)d_ws, (void*)d_tables, (void*)d_ptrs, (void*)d_partials, (void*)d_scratch,
                    (void*)d_log, (void*)d_resc, (void*)d_logs, (void*)d_chain, d_ones, (void*)d_stepP, (void*)d_stepNumel,
                    (void*)d_stage_in, (void*)d_stage_out})
      if (p) (void)hipFree(p);
    if (h_pack) (void)hipHostFree(h_pack);
    for (auto ev : events) (void)hipEventDestroy(ev);
    if (own_stream && stream) (void)hipStreamDestroy(stream);
  }
};

template <int MA, int BK, int TN>
static void launch_mfma_b(int mb, dim3 grid, hipStream_t st, const StepArgs& a) {
  switch (mb) {
    case 1: hipLaunchKernelGGL((k_mfma_f32<MA, 1, BK, TN>), grid, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((k_mfma_f32<MA, 2, BK, TN>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((k_mfma_f32<MA, 0, BK, TN>), grid, dim3(256), 0, st, a); break;
  }
}

template <int BK, int TN>
static void launch_mfma_a(int ma, int mb, dim3 grid, hipStream_t st, const StepArgs& a) {
  switch (ma) {
    case 1: launch_mfma_b<1, BK, TN>(mb, grid, st, a); break;
    case 2: launch_mfma_b<2, BK, TN>(mb, grid, st, a); break;
    default: launch_mfma_b<0, BK, TN>(mb, grid, st, a); break;
  }
}

// k-tile depth.  BK = 16: 32 KiB of LDS + 167 registers => 3 workgroups (12 waves) per CU, which
// hides the per-tile prologue/epilogue best when K is short (MPS shapes: 108-118 TFLOP/s vs
// 103-117 with BK = 32); BK = 32 halves the barriers per flop and wins on long-K GEMMs
// (4096^3: 130 vs 115 TFLOP/s).  CTN_MFMA_BK=16|32 forces one of them (development knob).
static int mfma_bk(int K) {
  static int forced = [] {
    const char* e = getenv("CTN_MFMA_BK");
    const int v = e ? atoi(e) : 0;
    return (v == 16 || v == 32) ? v : 0;
  }();
  if (forced) return forced;
  return K >= 2048 ? 32 : 16;
}

static void launch_mfma(int ma, int mb, int tile_n, dim3 grid, hipStream_t st, const StepArgs& a) {
  const bool bk16 = mfma_bk(a.K) == 16;
  if (tile_n == 64) {
    if (bk16) launch_mfma_a<16, 64>(ma, mb, grid, st, a);
    else launch_mfma_a<32, 64>(ma, mb, grid, st, a);
  } else {
    if (bk16) launch_mfma_a<16, 128>(ma, mb, grid, st, a);
    else launch_mfma_a<32, 128>(ma, mb, grid, st, a);
  }
}

static int exec_launch_all(Exec* E) {
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
  for (int s = 0; s < P.n_steps && !chain; ++s) {
    const Step& st = P.steps[s];
    StepArgs a;
    const int32_t* T = E->d_tables;
    a.obA = T + st.t.obA; a.obB = T + st.t.obB; a.obC = T + st.t.obC;
    a.omA = T + st.t.omA; a.omC = T + st.t.omC;
    a.onB = T + st.t.onB; a.onC = T + st.t.onC;
    a.okA = T + st.t.okA; a.okB = T + st.t.okB;
    a.ptrs = E->d_ptrs;
    auto part_of = [&](int id, const double** p, int32_t* cnt, double* numel) {
      *p = nullptr; *cnt = 0; *numel = 1;
      if (id >= P.n_inputs && P.stabilize) {
        const int ps = P.tensors[id].producer;
        *p = E->d_partials + (size_t)ps * R * kMaxPartials;
        *cnt = P.steps[ps].partials;
        *numel = (double)P.tensors[id].numel;
      }
    };
    part_of(st.lhs, &a.partA, &a.PA, &a.numelA);
    part_of(st.rhs, &a.partB, &a.PB, &a.numelB);
    double* part_dst = E->d_partials + (size_t)s * R * kMaxPartials;
    a.partC = st.collapse ? E->d_scratch : part_dst;
    a.partC_stride = st.collapse ? st.blocks : kMaxPartials;
    a.min_norm = P.min_norm;
    a.Bt = (int32_t)st.Bt; a.M = (int32_t)st.M; a.N = (int32_t)st.N; a.K = (int32_t)st.K;
    a.idA = st.lhs; a.idB = st.rhs >= 0 ? st.rhs : E->n_tensors - 1; a.idC = st.out;
    a.n_tensors = E->n_tensors;
    a.tiles_m = (int32_t)((st.M + kTileM - 1) / kTileM);
    a.tiles_n = (int32_t)((st.N + kTileN - 1) / kTileN);
    a.blocks_per_replica = st.blocks;
    a.R = R;
    a.c_vec = (st.cvec && (s + 1 < P.n_steps || E->outs_aligned16)) ? 1 : 0;
    a.dbg = nullptr;
    a.ohA = T + st.t.ohA; a.ohB = T + st.t.ohB; a.ohC = T + st.t.ohC;
    a.olA = T + st.t.olA; a.olB = T + st.t.olB; a.olC = T + st.t.olC;
    a.H = (int32_t)st.H; a.L = (int32_t)st.L; a.Nv = (int32_t)st.Nv;
    a.sAn = (int32_t)st.sAn; a.sBn = (int32_t)st.sBn;
    a.dNq = make_fastdiv((st.Nv + st.vecw - 1) / st.vecw);
    a.dL = make_fastdiv(st.L);
    a.dNv = make_fastdiv(st.Nv);

    const bool timed = E->timing_runs < E->timing_slots;  // only the first `slots` enqueues are bracketed
    const size_t ev0 = timed ? ((size_t)E->timing_runs * P.n_steps + s) * 2 : 0;
    if (timed) HIPCHECK(hipEventRecord(E->events[ev0], E->stream));
    switch (st.kernel) {
      case CTN_KERNEL_MFMA_F32: {
        const int64_t total = (int64_t)st.blocks * R;
        if (total >= (1LL << 31)) { g_err = "grid too large"; return CTN_UNSUPPORTED; }
        a.tiles_n = (int32_t)((st.N + st.tileN - 1) / st.tileN);
        if (getenv("CTN_DEBUG_STAMPS")) {
          if (E->dbg_tiles < (size_t)total) {
            if (E->d_dbg) (void)hipFree(E->d_dbg);
            HIPCHECK(hipMalloc((void**)&E->d_dbg, (size_t)total * 32));
            E->dbg_tiles = (size_t)total;
          }
          a.dbg = E->d_dbg;
        }
        launch_mfma(st.modeA, st.modeB, st.tileN, dim3((unsigned)total), E->stream, a);
        break;
      }
      case CTN_KERNEL_MFMA_F64: {
        const int64_t total = (int64_t)st.blocks * R;
        if (total >= (1LL << 31)) { g_err = "grid too large"; return CTN_UNSUPPORTED; }
        a.tiles_m = (int32_t)((st.M + kTile64 - 1) / kTile64);
        a.tiles_n = (int32_t)((st.N + kTile64 - 1) / kTile64);
        hipLaunchKernelGGL(k_mfma_f64, dim3((unsigned)total), dim3(256), 0, E->stream, a);
        break;
      }
      case CTN_KERNEL_DOT:
        if (P.dtype == CTN_F32) hipLaunchKernelGGL(k_dot<float>, dim3(st.blocks, R), dim3(256), 0, E->stream, a);
        else hipLaunchKernelGGL(k_dot<double>, dim3(st.blocks, R), dim3(256), 0, E->stream, a);
        break;
      case CTN_KERNEL_ROWDOT:
        if (P.dtype == CTN_F32) hipLaunchKernelGGL(k_rowdot<float>, dim3(st.blocks, R), dim3(256), 0, E->stream, a);
        else hipLaunchKernelGGL(k_rowdot<double>, dim3(st.blocks, R), dim3(256), 0, E->stream, a);
        break;
      default: {
        // vector stores need a 16-byte aligned destination: the caller's final buffer may not be
        const int vw = (s + 1 == P.n_steps && !E->outs_aligned16) ? 1 : st.vecw;
        const int64_t nq = (st.Nv + vw - 1) / vw;
        // 4 vectors per row lookup only where it pays: short K (lookup-dominated) and enough rows
        // left to fill the chip; long-K / small steps keep one vector per thread for parallelism
        const int64_t rows = st.H * st.L * (int64_t)R;
        const int u = (nq % 4 == 0 && st.K <= 16 && rows * (nq / 4) >= (1 << 20)) ? 4 : 1;
        a.dNq = make_fastdiv(nq / u);
        const dim3 g(st.blocks, R), b(256);
#define CTN_STREAM(TT, VV, UU) hipLaunchKernelGGL((k_stream<TT, VV, UU>), g, b, 0, E->stream, a)
        if (P.dtype == CTN_F32) {
          if (vw == 4) { if (u == 4) CTN_STREAM(float, 4, 4); else CTN_STREAM(float, 4, 1); }
          else { if (u == 4) CTN_STREAM(float, 1, 4); else CTN_STREAM(float, 1, 1); }
        } else {
          if (vw == 2) { if (u == 4) CTN_STREAM(double, 2, 4); else CTN_STREAM(double, 2, 1); }
          else { if (u == 4) CTN_STREAM(double, 1, 4); else CTN_STREAM(double, 1, 1); }
        }
#undef CTN_STREAM
        break;
      }
    }
    if (st.collapse)
      hipLaunchKernelGGL(k_collapse, dim3(R), dim3(256), 0, E->stream, (const double*)E->d_scratch, st.blocks, part_dst);
    if (timed) HIPCHECK(hipEventRecord(E->events[ev0 + 1], E->stream));
  }
  FinalArgs f;
  f.ptrs = E->d_ptrs;
  f.partials = E->d_partials;
  f.stepP = E->d_stepP;
  f.stepNumel = E->d_stepNumel;
  f.log_scale = E->d_log;
  f.rescales = E->d_resc;
  f.min_norm = P.min_norm;
  f.out_numel = P.output().numel;
  f.n_steps = P.n_steps; f.R = R; f.id_out = P.n_inputs + P.n_steps - 1; f.n_tensors = E->n_tensors;
  f.stabilize = P.stabilize ? 1 : 0;
  int fb = (int)std::min<int64_t>((P.output().numel + 255) / 256, 1024);
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
int64_t ctn_plan_out_numel(const ctn_plan* plan) { return plan ? plan->p.output().numel : 0; }
int64_t ctn_plan_out_bytes(const ctn_plan* plan) {
  return plan ? plan->p.output().numel * (int64_t)plan->p.elem_size() : 0;
}

static int64_t exec_fixed_bytes(const Plan& P, int R) {
  const int nt = P.n_inputs + P.n_steps + 1;
  return (int64_t)P.tables.size() * 4 + (int64_t)R * nt * 8 + (int64_t)P.n_steps * R * kMaxPartials * 8 +
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
  return CTN_OK;
}

int ctn_exec_create(const ctn_plan* plan, int device, void* stream, int replicas, ctn_exec** out) {
  if (!plan || !out || replicas < 1) { g_err = "invalid argument to ctn_exec_create"; return CTN_INVALID_ARG; }
  *out = nullptr;
  int ndev = 0;
  int rc = ctn_device_count(&ndev);
  if (rc != CTN_OK) return rc;
  if (device < 0 || device >= ndev) { g_err = "device index out of range"; return CTN_INVALID_ARG; }
  HIPCHECK(hipSetDevice(device));
  int n_cu = 256;
  (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device);
  ctn_exec* x = new (std::nothrow) ctn_exec();
  if (!x) { g_err = "out of host memory"; return CTN_OOM; }
  Exec& E = x->e;
  const Plan& P = plan->p;
  E.plan = &P; E.device = device; E.R = replicas; E.n_cu = n_cu > 0 ? n_cu : 256;
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
  HIPCHECK_X(hipMalloc((void**)&E.d_ptrs, (size_t)replicas * E.n_tensors * sizeof(void*)));
  HIPCHECK_X(hipMalloc((void**)&E.d_partials, (size_t)P.n_steps * replicas * kMaxPartials * 8));
  HIPCHECK_X(hipMemset(E.d_partials, 0, (size_t)P.n_steps * replicas * kMaxPartials * 8));
  HIPCHECK_X(hipMalloc((void**)&E.d_scratch, (size_t)replicas * std::max<int64_t>(P.max_collapse_blocks, 1) * 8));
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
  for (int s = 0; s < P.n_steps; ++s) { sp[s] = P.steps[s].partials; sn[s] = (double)P.tensors[P.steps[s].out].numel; }
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
      c.obA = T + st.t.obA; c.obB = T + st.t.obB; c.obC = T + st.t.obC;
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
  HIPCHECK(hipSetDevice(E->device));
  int rc = exec_set_pointers(E, dev_inputs, dev_outs);
  if (rc != CTN_OK) return rc;
  return exec_launch_all(E);
}

int ctn_exec_synchronize(ctn_exec* exec) {
  if (!exec) { g_err = "NULL argument"; return CTN_INVALID_ARG; }
  HIPCHECK(hipStreamSynchronize(exec->e.stream));
  if (const char* path = getenv("CTN_DEBUG_STAMPS")) {  // development only: dump the last MFMA launch's stamps
    Exec* E = &exec->e;
    if (E->d_dbg && E->dbg_tiles) {
      std::vector<unsigned long long> h(E->dbg_tiles * 4);
      HIPCHECK(hipMemcpy(h.data(), E->d_dbg, h.size() * 8, hipMemcpyDeviceToHost));
      if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
  }
  return CTN_OK;
}

int ctn_exec_fetch(ctn_exec* exec, double* log_scale, double* step_rescales) {
  if (!exec) { g_err = "NULL argument"; return CTN_INVALID_ARG; }
  Exec* E = &exec->e;
  HIPCHECK(hipSetDevice(E->device));
  if (log_scale)
    HIPCHECK(hipMemcpyAsync(log_scale, E->d_log, (size_t)E->R * 8, hipMemcpyDeviceToHost, E->stream));
  if (step_rescales)
    HIPCHECK(hipMemcpyAsync(step_rescales, E->d_resc, (size_t)E->R * E->plan->n_steps * 8,
                            hipMemcpyDeviceToHost, E->stream));
  HIPCHECK(hipStreamSynchronize(E->stream));
  return CTN_OK;
}

int ctn_exec_run(ctn_exec* exec, const void* const* inputs, int inputs_space, void* const* outs,
                 int outs_space, double* log_scale, double* step_rescales) {
  if (!exec || !inputs || !outs) { g_err = "NULL argument"; return CTN_INVALID_ARG; }
  Exec* E = &exec->e;
  const Plan& P = *E->plan;
  HIPCHECK(hipSetDevice(E->device));
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
  rc = exec_launch_all(E);
  if (rc != CTN_OK) return rc;
  if (outs_space == CTN_MEM_HOST)
    for (int r = 0; r < E->R; ++r) {
      if (!outs[r]) { g_err = "null output pointer"; return CTN_INVALID_ARG; }
      HIPCHECK(hipMemcpyAsync(outs[r], dout[r], (size_t)out_bytes, hipMemcpyDeviceToHost, E->stream));
    }
  return ctn_exec_fetch(exec, log_scale, step_rescales);
}

int ctn_exec_set_timing(ctn_exec* exec, int slots) {
  if (!exec || slots < 0) { g_err = "invalid argument"; return CTN_INVALID_ARG; }
  Exec* E = &exec->e;
  HIPCHECK(hipSetDevice(E->device));
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
