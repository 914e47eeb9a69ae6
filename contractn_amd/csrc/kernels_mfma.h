// kernels_mfma.h - MFMA GEMM kernels (fp32 v_mfma_f32_32x32x2_f32, fp64 v_mfma_f64_16x16x4_f64) with table-driven gather loads
// Part of the gfx950 contraction engine (see engine.hip for the overview).
#pragma once
#include "kernel_args.h"
#ifndef CTN_EXP
#define CTN_EXP 0
#endif

namespace ctn {

// ---------------------------------------------------------------------------
// K-mfma-f32: TM x TN workgroup tile (TM, TN = 128 or 64), 4 waves (2x2), each wave
// TM/2 x TN/2 = TM/64 x TN/64 v_mfma_f32_32x32x2_f32 accumulators, register-staged
// double-buffered LDS, one barrier per k-tile.  TN = 64 / TM = 64 serve skinny products
// (boundary absorptions of 2D grids: N = 64 or M = 64 against a huge other extent, 16 flop/B) where a
// 128-wide tile would be half masked - these steps sit at the HBM ridge, and a half-masked tile makes them
// matrix-pipe bound instead.
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

// MODE 3 ("KR"): the A operand is the element-wise product of TWO tensors, X (.) Y, formed while the tile is staged -
// a Khatri-Rao / Hadamard / broadcast product (reference einsum.py:382-384: `ad,ac->acd`, `bl,bp->bpl`) that the
// planner fused into the GEMM consuming it, so the product tensor never exists in memory (CP with r = n = 1024:
// a 4 GiB intermediate; r = 4096: 2^32 elements, which no buffer of this engine could even hold).  Every A element
// is X[omX[m] + okX[k]] * Y[omY[m] + okY[k]]: two scalar gathers from small, cache-resident tensors (a label that
// an operand does not carry has stride 0 in its tables).  LDS image [k][rows], like mode 0.
template <int BK, int ROWS>
struct TileLoaderKR {
  static constexpr int NV = ROWS * BK / 256;
  static constexpr int KPP = 256 / ROWS;
  static constexpr int kSize = BK * ROWS;
  float v[NV], v2[NV];
  int kofs[NV], kofs2[NV];
  int offm, offm2;
  bool okm;
  const int32_t* __restrict__ ok2;
  const float* __restrict__ base2;

  __device__ __forceinline__ void init(const int32_t* __restrict__ om, int m0, int M, int tid) {
    const int gm = m0 + (tid % ROWS);
    offm = om[gm];
    okm = gm < M;
  }
  __device__ __forceinline__ void init2(const int32_t* __restrict__ om2, const int32_t* __restrict__ ok2_,
                                        const float* __restrict__ base2_, int m0, int tid) {
    offm2 = om2[m0 + (tid % ROWS)];
    ok2 = ok2_;
    base2 = base2_;
  }
  __device__ __forceinline__ void tab(const int32_t* __restrict__ ok, int k0, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      kofs[i] = ok[k0 + tid / ROWS + KPP * i];
      kofs2[i] = ok2[k0 + tid / ROWS + KPP * i];
    }
  }
  __device__ __forceinline__ void load(const float* __restrict__ base) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i] = base[offm + kofs[i]];
      v2[i] = base2[offm2 + kofs2[i]];
    }
  }
  template <bool FULL>
  __device__ __forceinline__ void store(float* __restrict__ s, int k0, int K, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int kr = tid / ROWS + KPP * i;
      const bool in = FULL || (okm && (k0 + kr) < K);
      s[kr * ROWS + (tid % ROWS)] = in ? v[i] * v2[i] : 0.f;
    }
  }
  static __device__ __forceinline__ int idx(int row, int k) { return k * ROWS + row; }
};

// MODES 4 / 5: the same on-the-fly product with 16-byte accesses where the factors allow it.  Along the vector
// direction - 4 consecutive rows (mode 4, LDS image [k][rows] like mode 1) or 4 consecutive k (mode 5, image
// [rows][BK+1] like mode 2) - each factor is, by its stride along the innermost label there: contiguous (kind 1: one
// float4 load), constant (kind 2: one scalar, broadcast - the factor does not carry that label), or anything else
// (kind 0: four scalar gathers).  The planner picks the direction in which at least one factor is contiguous and
// passes the kinds (StepArgs::krX / krY); they are wave-uniform, so the branches cost nothing.
//   CP   `ad,ac->acd . ae`: rows (c, d), d innermost: B[a, d] contiguous, A[a, c] constant        -> mode 4
//   MPS  `bl (.) bp . plr`: k = (p, l), l innermost: v[b, l] contiguous, x[b, p] constant          -> mode 5
template <int DIR, int BK, int ROWS>
struct TileLoaderKRV {
  static_assert(DIR == 1 || DIR == 2, "vector direction");
  static constexpr int NV = ROWS * BK / 256;               // floats staged per thread
  static constexpr int VPR = ROWS / 4;                     // DIR 1: float4 per k-row
  static constexpr int RPP = 256 / VPR;                    // DIR 1: k-rows covered per pass
  static constexpr int NT = DIR == 1 ? BK / RPP : 1;       // k-table positions per thread and tile
  static constexpr int NM = DIR == 2 ? NV / 4 : 1;         // row positions per thread
  static constexpr int LDK = BK + 1;
  static constexpr int kSize = DIR == 2 ? ROWS * LDK : BK * ROWS;
  float v[NV];
  // offsets of the 4 elements of a vector differ only for a gathered factor; [.][0] serves kinds 1 and 2
  int kx[NT][DIR == 2 ? 4 : 1], ky[NT][DIR == 2 ? 4 : 1];   // k-table entries of the NEXT tile
  int mx[NM][DIR == 1 ? 4 : 1], my[NM][DIR == 1 ? 4 : 1];   // row-table entries
  bool okm[NM];
  int kindX, kindY;
  const int32_t* __restrict__ ok2;
  const float* __restrict__ base2;

  __device__ __forceinline__ void init(const int32_t* __restrict__ om, int m0, int M, int tid) {
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      const int gm = DIR == 1 ? m0 + (tid % VPR) * 4 : m0 + ((tid + i * 256) / (BK / 4));
#pragma unroll
      for (int j = 0; j < (DIR == 1 ? 4 : 1); ++j) mx[i][j] = om[gm + j];
      okm[i] = gm < M;
    }
  }
  __device__ __forceinline__ void init2(const int32_t* __restrict__ om2, const int32_t* __restrict__ ok2_,
                                        const float* __restrict__ base2_, int m0, int tid, int kx_, int ky_) {
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      const int gm = DIR == 1 ? m0 + (tid % VPR) * 4 : m0 + ((tid + i * 256) / (BK / 4));
#pragma unroll
      for (int j = 0; j < (DIR == 1 ? 4 : 1); ++j) my[i][j] = om2[gm + j];
    }
    ok2 = ok2_;
    base2 = base2_;
    kindX = kx_;
    kindY = ky_;
  }
  __device__ __forceinline__ void tab(const int32_t* __restrict__ ok, int k0, int tid) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int kk = DIR == 1 ? k0 + tid / VPR + RPP * i : k0 + (tid % (BK / 4)) * 4;
#pragma unroll
      for (int j = 0; j < (DIR == 2 ? 4 : 1); ++j) { kx[i][j] = ok[kk + j]; ky[i][j] = ok2[kk + j]; }
    }
  }
  // four values of one factor: rows gm..gm+3 at one k (DIR 1) or k..k+3 of one row (DIR 2)
  template <typename OM, typename OK>
  static __device__ __forceinline__ void four(const float* __restrict__ base, int kind, const OM& om, const OK& ok, float (&o)[4]) {
    if (kind == 1) {
      const float4 x = *reinterpret_cast<const float4*>(base + om[0] + ok[0]);
      o[0] = x.x; o[1] = x.y; o[2] = x.z; o[3] = x.w;
    } else if (kind == 2) {
      const float x = base[om[0] + ok[0]];
      o[0] = o[1] = o[2] = o[3] = x;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = base[om[DIR == 1 ? j : 0] + ok[DIR == 2 ? j : 0]];
    }
  }
  __device__ __forceinline__ void load(const float* __restrict__ base) {
    constexpr int N = DIR == 1 ? NT : NM;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float a[4], b[4];
      four(base, kindX, mx[DIR == 1 ? 0 : i], kx[DIR == 1 ? i : 0], a);
      four(base2, kindY, my[DIR == 1 ? 0 : i], ky[DIR == 1 ? i : 0], b);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * i + j] = a[j] * b[j];
    }
  }
  template <bool FULL>
  __device__ __forceinline__ void store(float* __restrict__ s, int k0, int K, int tid) const {
    if (DIR == 1) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int kr = tid / VPR + RPP * i;
        const bool in = FULL || (okm[0] && (k0 + kr) < K);
        *reinterpret_cast<float4*>(s + kr * ROWS + (tid % VPR) * 4) =
            make_float4(in ? v[4 * i] : 0.f, in ? v[4 * i + 1] : 0.f, in ? v[4 * i + 2] : 0.f, in ? v[4 * i + 3] : 0.f);
      }
    } else {
      const bool kin = (k0 + (tid % (BK / 4)) * 4) < K;
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        const bool in = FULL || (okm[i] && kin);
        float* d = s + ((tid + i * 256) / (BK / 4)) * LDK + (tid % (BK / 4)) * 4;
        d[0] = in ? v[4 * i] : 0.f; d[1] = in ? v[4 * i + 1] : 0.f;
        d[2] = in ? v[4 * i + 2] : 0.f; d[3] = in ? v[4 * i + 3] : 0.f;
      }
    }
  }
  static __device__ __forceinline__ int idx(int row, int k) { return DIR == 2 ? row * LDK + k : k * ROWS + row; }
};

template <int MODE, int BK, int ROWS>
struct LoaderOf { typedef TileLoader<MODE, BK, ROWS> type; };
template <int BK, int ROWS>
struct LoaderOf<3, BK, ROWS> { typedef TileLoaderKR<BK, ROWS> type; };
template <int BK, int ROWS>
struct LoaderOf<4, BK, ROWS> { typedef TileLoaderKRV<1, BK, ROWS> type; };
template <int BK, int ROWS>
struct LoaderOf<5, BK, ROWS> { typedef TileLoaderKRV<2, BK, ROWS> type; };

template <int MA, int MB, int BK, int TN, bool FULL, int BM>
__device__ __forceinline__ void mfma_mainloop(typename LoaderOf<MA, BK, BM>::type& la, TileLoader<MB, BK, TN>& lb,
                                              const float* __restrict__ A, const float* __restrict__ B,
                                              const int32_t* __restrict__ okA, const int32_t* __restrict__ okB,
                                              int K, float* sA, float* sB, f32x16 (&acc)[BM / 64][TN / 64], int tid,
                                              unsigned long long* dbg1) {
  using LA = typename LoaderOf<MA, BK, BM>::type;
  using LB = TileLoader<MB, BK, TN>;
  constexpr int SZA = LA::kSize, SZB = LB::kSize;
  constexpr int NJ = TN / 64;  // 32-wide column blocks per wave
  constexpr int NI = BM / 64;  // 32-high row blocks per wave
  const int lane = tid & 63, w = tid >> 6;
  const int wm = (w >> 1) * (BM / 2), wn = (w & 1) * (TN / 2);
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
  int fax[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) fax[i] = LA::idx(wm + i * 32 + l31, h);
  int fbx[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) fbx[j] = LB::idx(wn + j * 32 + l31, h);
  constexpr int stepA = (MA == 2 || MA == 5) ? 2 : 2 * BM;  // advance of the fragment index per k-step (k += 2)
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
    float fa[2][NI], fb[2][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i) fa[0][i] = cA[fax[i]];
#pragma unroll
    for (int j = 0; j < NJ; ++j) fb[0][j] = cB[fbx[j]];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int c = kk & 1, nx = c ^ 1;
      if (kk + 1 < BK / 2) {
#pragma unroll
        for (int i = 0; i < NI; ++i) fa[nx][i] = cA[fax[i] + (kk + 1) * stepA];
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[nx][j] = cB[fbx[j] + (kk + 1) * stepB];
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i], fb[c][j], acc[i][j], 0, 0, 0);
      // pin the interleave: the LDS reads of step kk+1 issue ahead of the MFMAs of step kk
      __builtin_amdgcn_sched_group_barrier(0x100, NI + NJ, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NI * NJ, 0);
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
// EPW: the innermost column label (extent a.epw = 2 or 4, so the 4 columns a lane stores are whole groups of it) is
// re-weighted by a third tensor W[row][p] and summed on the way out (planner pattern C: `bpr,bp->br` absorbed into
// `bl,plr->bpr`); the C column table then repeats each output offset a.epw times.
template <int MA, int MB, int BK, int TN, int BM = kTileM, bool EPW = false>
__global__ __launch_bounds__(256, (BK == 16 ? 3 : 2)) void k_mfma_f32(StepArgs a) {
  static_assert((BM == 128 || BM == 64) && (TN == 128 || TN == 64), "tile shapes");
  static_assert(!EPW || MA <= 2, "epilogue-summed steps: plain A operand");
  using LA = typename LoaderOf<MA, BK, BM>::type;
  using LB = TileLoader<MB, BK, TN>;
  constexpr int SZA = LA::kSize, SZB = LB::kSize;
  constexpr int NJ = TN / 64;
  constexpr int NI = BM / 64;
  // one LDS object: [A buf0][A buf1][B buf0][B buf1][omC BM][onC TN][red 4 doubles]; the operand buffers double as
  // the epilogue's per-wave staging area (32 rows x TN/2 columns each), which for the 64-row tile is the larger of the two
  // (the 64-row tile stages 16 rows at a time: 25 KB instead of 33 KB of LDS, five workgroups per CU instead of four)
  constexpr int SR = BM == 64 ? 16 : 32;           // rows a wave stages per epilogue pass
  constexpr int BUF = (2 * SZA + 2 * SZB > 4 * SR * (TN / 2) ? 2 * SZA + 2 * SZB : 4 * SR * (TN / 2) + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float smem[BUF + BM + TN + 8 + (EPW ? 5 * BM : 0)];
  float* sA = smem;
  float* sB = smem + 2 * SZA;
  int* s_omC = reinterpret_cast<int*>(smem + BUF);
  int* s_onC = s_omC + BM;
  double* red = reinterpret_cast<double*>(s_onC + TN);
  int* s_omW = reinterpret_cast<int*>(smem + BUF + BM + TN + 8);   // EPW: offset of every row in the weight tensor
  float* s_W = smem + BUF + BM + TN + 8 + BM;                      // EPW: [BM][4] the rows' weights, rescaled

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
  if (a.dbg && tid == 0) a.dbg[(size_t)pid * 8 + 0] = __builtin_amdgcn_s_memtime();
#endif
  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const float* __restrict__ A = (const float*)tp[a.idA] + a.obA[b];
  const float* __restrict__ B = (const float*)tp[a.idB] + a.obB[b];
  float* __restrict__ C = (float*)tp[a.idC] + a.obC[b];

  if (tid < BM) s_omC[tid] = a.omC[m0 + tid];
  else if (tid - BM < TN) s_onC[tid - BM] = a.onC[n0 + tid - BM];
  float scW = 1.f;
  const float* __restrict__ W = nullptr;
  if constexpr (EPW) {
    if (tid < BM) s_omW[tid] = a.omA2[m0 + tid];
    scW = producer_scale<float>(a.partA2, a.PA2, a.strideA2, a.numelA2, a.min_norm, r);
    W = (const float*)tp[a.idA2];
  }

  LA la;
  LB lb;
  la.init(a.omA, m0, a.M, tid);
  if constexpr (MA == 3) la.init2(a.omA2, a.okA2, (const float*)tp[a.idA2] + a.obA2[b], m0, tid);
  if constexpr (MA >= 4) la.init2(a.omA2, a.okA2, (const float*)tp[a.idA2] + a.obA2[b], m0, tid, a.krX, a.krY);
  lb.init(a.onB, n0, a.N, tid);
  // the producers' partial sums are requested AFTER the offset tables, so one wait covers both (their reduction is a
  // wave butterfly: asked for first, its wait used to hold back the table loads - 512 partials, two rounds: ~1 us)
  float scA = producer_scale<float>(a.partA, a.PA, a.strideA, a.numelA, a.min_norm, r);
  const float scB = producer_scale<float>(a.partB, a.PB, a.strideB, a.numelB, a.min_norm, r);
  float scA2 = 1.f;
  if constexpr (MA >= 3) scA2 = producer_scale<float>(a.partA2, a.PA2, a.strideA2, a.numelA2, a.min_norm, r);

  const int lane = tid & 63, w = tid >> 6;
  const int wm = (w >> 1) * (BM / 2), wn = (w & 1) * (TN / 2);
  const int l31 = lane & 31, h = lane >> 5;

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#ifdef CTN_STAMPS
  unsigned long long* stamp1 = a.dbg ? a.dbg + (size_t)pid * 8 + 1 : nullptr;
#else
  unsigned long long* stamp1 = nullptr;
#endif
  // FULL: the tile lies completely inside M x N and K is a multiple of BK -> no masking
  const bool full = (m0 + BM <= a.M) && (n0 + TN <= a.N) && (a.K % BK == 0);
  if (full) mfma_mainloop<MA, MB, BK, TN, true, BM>(la, lb, A, B, a.okA, a.okB, a.K, sA, sB, acc, tid, stamp1);
  else mfma_mainloop<MA, MB, BK, TN, false, BM>(la, lb, A, B, a.okA, a.okB, a.K, sA, sB, acc, tid, stamp1);
#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) a.dbg[(size_t)pid * 8 + 2] = __builtin_amdgcn_s_memtime();
#endif

  // epilogue: lazy rescale, store through the C offset tables, abs-sum partial
  // (KR: both factors of the A operand carry a rescale; folded into one multiplier)
  const float iA = MA >= 3 ? (1.0f / scA) * (1.0f / scA2) : 1.0f / scA, iB = 1.0f / scB;
  float asum = 0.f;
  if constexpr (EPW) {
    // the tile's weights through LDS, fetched once (read from memory inside the store loop below, every one of its
    // eight rounds waited for its own loads: 4.2 us of epilogue on a 128 x 64 x 256 tile)
    if (tid < BM && m0 + tid < a.M) {
      const float* wp = W + s_omW[tid];
      const float iW = 1.0f / scW;
      const float w0 = wp[0], w1 = wp[1];
      float w2 = 0.f, w3 = 0.f;
      if ((a.epw & 0xFF) == 4) { w2 = wp[2]; w3 = wp[3]; }
      *reinterpret_cast<float4*>(s_W + 4 * tid) = make_float4(w0 * iW, w1 * iW, w2 * iW, w3 * iW);
    }
    __syncthreads();
  }
  {
    // Each WAVE stages its own 64 x TN/2 accumulator block through its quarter of the (now idle)
    // operand buffers, 32 rows at a time, and stores whole 16-byte row segments (4-8 rows of
    // 128-256 contiguous bytes per store instruction).  No workgroup barrier is involved: LDS
    // operations of one wave execute in order, so the write -> read hand-off is wave-local.
    constexpr int WT = TN / 2;                       // columns owned by a wave
    constexpr int LDSW = (BUF / 4) & ~3;             // floats of LDS per wave (16-byte aligned)
    static_assert(LDSW >= SR * WT, "per-wave staging area too small");
    constexpr int VW = WT / 4;                       // 16-byte vectors per row
    constexpr int RPI = 64 / VW;                     // rows covered by one wave-wide vector access
    float* wC = smem + w * LDSW;                     // [32][WT]
    const int c4 = (lane % VW) * 4;
    const int gcol = wn + c4;
    const bool cin = n0 + gcol < a.N;  // N % 4 == 0 whenever c_vec, otherwise checked per element
    const int offn = s_onC[gcol];
#pragma unroll
    for (int ih = 0; ih < NI * (32 / SR); ++ih) {
      const int i = ih / (32 / SR), half = ih % (32 / SR);   // accumulator registers e: rows (e & 3) + 8 (e >> 2) + 4 h
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = half * (SR / 2); e < half * (SR / 2) + SR / 2; ++e)
          wC[((e & 3) + 8 * (e >> 2) + 4 * h - half * SR) * WT + j * 32 + l31] = (acc[i][j][e] * iA) * iB;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < SR / RPI; ++it) {
        const int lrow = it * RPI + lane / VW;
        const int row = wm + i * 32 + half * SR + lrow;
        const float4 v = *reinterpret_cast<const float4*>(wC + lrow * WT + c4);
        if constexpr (EPW) {
          const int P = a.epw & 0xFF;
          if (a.epw & 0x100) {
            // columns ordered (.., u_hi, p, u_lo): this lane holds four consecutive u of ONE p, the other p of the
            // same u sit in the adjacent lanes - weight, then add across the 2 / 4 lanes; the p = 0 lane stores
            const bool in = m0 + row < a.M && cin;
            const int pidx = (gcol >> 2) & (P - 1);
            const float wq = in ? s_W[4 * row + pidx] : 0.f;
            float4 x = make_float4(v.x * wq, v.y * wq, v.z * wq, v.w * wq);
            x.x += __shfl_xor(x.x, 1, 64); x.y += __shfl_xor(x.y, 1, 64); x.z += __shfl_xor(x.z, 1, 64); x.w += __shfl_xor(x.w, 1, 64);
            if (P == 4) {
              x.x += __shfl_xor(x.x, 2, 64); x.y += __shfl_xor(x.y, 2, 64); x.z += __shfl_xor(x.z, 2, 64); x.w += __shfl_xor(x.w, 2, 64);
            }
            if (in && pidx == 0) {
              float* dst = C + s_omC[row];
              if (a.c_vec) {
                *reinterpret_cast<float4*>(dst + offn) = x;
              } else {
                dst[offn] = x.x; dst[s_onC[gcol + 1]] = x.y; dst[s_onC[gcol + 2]] = x.z; dst[s_onC[gcol + 3]] = x.w;
              }
              asum += (fabsf(x.x) + fabsf(x.y)) + (fabsf(x.z) + fabsf(x.w));
            }
          } else if (m0 + row < a.M && cin) {
            float* dst = C + s_omC[row];
            const float4 wv = *reinterpret_cast<const float4*>(s_W + 4 * row);
            if (P == 4) {
              const float o = (v.x * wv.x + v.y * wv.y) + (v.z * wv.z + v.w * wv.w);
              dst[offn] = o;
              asum += fabsf(o);
            } else {
              const float w0 = wv.x, w1 = wv.y;
              const float o0 = v.x * w0 + v.y * w1;
              dst[offn] = o0;
              asum += fabsf(o0);
              if (n0 + gcol + 2 < a.N) {
                const float o1 = v.z * w0 + v.w * w1;
                dst[s_onC[gcol + 2]] = o1;
                asum += fabsf(o1);
              }
            }
          }
        } else if (m0 + row < a.M && cin) {
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
#ifdef CTN_STAMPS
      if (a.dbg && tid == 0) a.dbg[(size_t)pid * 8 + 4 + i] = __builtin_amdgcn_s_memtime();
#endif
    }
  }
#ifdef CTN_STAMPS
  if (a.dbg && tid == 192) a.dbg[(size_t)pid * 8 + 6] = __builtin_amdgcn_s_memtime();
  if (a.dbg && tid == 0)
    a.dbg[(size_t)pid * 8 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                 ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
#endif
  const double tot = block_sum((double)asum, red);
  if (tid == 0) a.partC[(size_t)r * a.partC_stride + t] = tot;
#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) a.dbg[(size_t)pid * 8 + 3] = __builtin_amdgcn_s_memtime();
#endif
}

// ---------------------------------------------------------------------------
// K-mfma-f32-sk: LATENCY mode for launches that cannot fill the chip with 128-wide tiles (a single
// network or a few replicas: a 128x128x1024 tile is ~55 us of matrix-pipe time on ONE CU while
// 250 CUs idle).  64x64 tiles, 4 waves of 32x32 (one accumulator each), BK = 32, and the K range
// split S ways across workgroups; every split writes its un-scaled partial tile into slab s of a
// scratch buffer laid out like C, and k_splitk_reduce sums the S slabs in a FIXED order (bit-
// reproducible, no atomics), applies the lazy rescale and emits the abs-sum partials.
// ---------------------------------------------------------------------------
struct SplitKArgs {
  void* slab;         // [R][S][numelC] elements of the plan's dtype
  int64_t numelC;
  int32_t S, kchunk;  // splits and K elements per split (multiple of 32)
  int32_t tiles_m, tiles_n, tiles_per_replica;  // 64-wide tiles
};

template <int MA, int MB>
__global__ __launch_bounds__(256) void k_mfma_f32_sk(StepArgs a, SplitKArgs sk) {
  constexpr int T = 64, BK = 32;
  using LA = TileLoader<MA, BK, T>;
  using LB = TileLoader<MB, BK, T>;
  constexpr int SZA = LA::kSize, SZB = LB::kSize;
  __shared__ __attribute__((aligned(16))) float smem[2 * SZA + 2 * SZB];
  __shared__ int s_omC[T], s_onC[T];
  float* sA = smem;
  float* sB = smem + 2 * SZA;

  const int tid = threadIdx.x;
  const int per_rep = sk.tiles_per_replica * sk.S;
  const int pid = blockIdx.x;
  const int r = pid / per_rep;
  const int rem = pid - r * per_rep;
  const int t = rem / sk.S;
  const int s = rem - t * sk.S;
  const int tiles_mn = sk.tiles_m * sk.tiles_n;
  const int b = t / tiles_mn;
  const int tt = t - b * tiles_mn;
  const int m0 = (tt / sk.tiles_n) * T;
  const int n0 = (tt % sk.tiles_n) * T;
  const int kbeg = s * sk.kchunk;
  const int kend = min(a.K, kbeg + sk.kchunk);   // this split's K range; masks use kend

  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const float* __restrict__ A = (const float*)tp[a.idA] + a.obA[b];
  const float* __restrict__ B = (const float*)tp[a.idB] + a.obB[b];
  float* __restrict__ C = (float*)sk.slab + ((size_t)r * sk.S + s) * sk.numelC + a.obC[b];

  if (tid < T) s_omC[tid] = a.omC[m0 + tid];
  else if (tid < 2 * T) s_onC[tid - T] = a.onC[n0 + tid - T];

  LA la;
  LB lb;
  la.init(a.omA, m0, a.M, tid);
  lb.init(a.onB, n0, a.N, tid);

  const int lane = tid & 63, w = tid >> 6;
  const int wm = (w >> 1) * 32, wn = (w & 1) * 32;
  const int l31 = lane & 31, h = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;

  const int nkt = (kend - kbeg + BK - 1) / BK;   // >= 1 (host guarantees kbeg < K)
  la.tab(a.okA, kbeg, tid);
  lb.tab(a.okB, kbeg, tid);
  la.load(A);
  lb.load(B);
  la.tab(a.okA, kbeg + BK, tid);
  lb.tab(a.okB, kbeg + BK, tid);
  la.template store<false>(sA, kbeg, kend, tid);
  lb.template store<false>(sB, kbeg, kend, tid);
  __syncthreads();

  const int fa = LA::idx(wm + l31, h), fb = LB::idx(wn + l31, h);
  constexpr int stepA = MA == 2 ? 2 : 2 * T;
  constexpr int stepB = MB == 2 ? 2 : 2 * T;
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nkt;
    if (more) {
      la.load(A);
      lb.load(B);
      la.tab(a.okA, kbeg + (kt + 2) * BK, tid);
      lb.tab(a.okB, kbeg + (kt + 2) * BK, tid);
    }
    __builtin_amdgcn_sched_barrier(0);
    const float* cA = sA + cur * SZA;
    const float* cB = sB + cur * SZB;
    float xa[2], xb[2];
    xa[0] = cA[fa]; xb[0] = cB[fb];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int c = kk & 1, nx = c ^ 1;
      if (kk + 1 < BK / 2) {
        xa[nx] = cA[fa + (kk + 1) * stepA];
        xb[nx] = cB[fb + (kk + 1) * stepB];
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[c], xb[c], acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      la.template store<false>(sA + (cur ^ 1) * SZA, kbeg + (kt + 1) * BK, kend, tid);
      lb.template store<false>(sB + (cur ^ 1) * SZB, kbeg + (kt + 1) * BK, kend, tid);
    }
    __syncthreads();
  }

  const int col = wn + l31;
  if (n0 + col < a.N) {
    const int offn = s_onC[col];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wm + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (m0 + row < a.M) C[s_omC[row] + offn] = acc[e];
    }
  }
}

// Sum the S partial slabs in order, apply the producers' rescale, write C and the abs-sum partials
// (exactly P = the step's planned partial count of workgroups per replica, so consumers are unchanged).
// The pass is pure latency (a few MB from L2): a thread's SMAX x U 16-byte loads - every slab of U
// vectors - are all requested before the first add, one round trip instead of S x iterations of them;
// the adds stay in slab order.
template <typename T, int SMAX, int U>
__device__ __forceinline__ T splitk_reduce_span(const T* __restrict__ slab, T* __restrict__ C, int64_t numelC, int S,
                                                int64_t lo, int64_t hi, T iA, T iB) {
  constexpr int V = 16 / sizeof(T);
  struct alignas(16) Vec { T x[V]; };
  T asum = 0;
  for (int64_t i0 = lo + (int64_t)threadIdx.x * V; i0 < hi; i0 += (int64_t)256 * V * U) {
    Vec x[SMAX][U];
#pragma unroll
    for (int s = 0; s < SMAX; ++s)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + (int64_t)u * 256 * V;
        if (s < S && i < hi) x[s][u] = *reinterpret_cast<const Vec*>(slab + (size_t)s * numelC + i);
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * 256 * V;
      if (i < hi) {
        Vec v = x[0][u];
#pragma unroll
        for (int s = 1; s < SMAX; ++s)
          if (s < S) {
#pragma unroll
            for (int e = 0; e < V; ++e) v.x[e] += x[s][u].x[e];
          }
        T part = 0;
#pragma unroll
        for (int e = 0; e < V; ++e) { v.x[e] = (v.x[e] * iA) * iB; part += fabs(v.x[e]); }
        *reinterpret_cast<Vec*>(C + i) = v;
        asum += part;
      }
    }
  }
  return asum;
}

// More than 16 slabs (a huge K against a small output, e.g. 64 x 64 x 2,097,152 = 512 slabs): fold them
// 16 to 1 first - slab g of dst = slabs 16 g .. 16 g + 15 of src, summed in order, every (vector, group) its
// own thread with all 16 loads in flight - until at most 16 are left for k_splitk_reduce.  (One thread
// walking 512 slabs took 469 us of a 670 us step.)  grid (element spans, groups, R); any numelC.
template <typename T>
__global__ __launch_bounds__(256) void k_splitk_fold(const T* __restrict__ src, T* __restrict__ dst, int64_t numelC,
                                                     int S_in, int S_out) {
  constexpr int V = 16 / sizeof(T);
  struct alignas(16) Vec { T x[V]; };
  const int r = blockIdx.z, g = blockIdx.y;
  const int s0 = 16 * g, ns = min(16, S_in - s0);
  const T* __restrict__ in = src + ((size_t)r * S_in + s0) * numelC;
  T* __restrict__ out = dst + ((size_t)r * S_out + g) * numelC;
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V;
  if (i >= numelC) return;
  if ((numelC & (V - 1)) == 0) {
    Vec x[16];
#pragma unroll
    for (int s = 0; s < 16; ++s)
      if (s < ns) x[s] = *reinterpret_cast<const Vec*>(in + (size_t)s * numelC + i);
    Vec v = x[0];
#pragma unroll
    for (int s = 1; s < 16; ++s)
      if (s < ns) {
#pragma unroll
        for (int e = 0; e < V; ++e) v.x[e] += x[s].x[e];
      }
    *reinterpret_cast<Vec*>(out + i) = v;
  } else {
    for (int e = 0; e < V && i + e < numelC; ++e) {
      T v = in[i + e];
      for (int s = 1; s < ns; ++s) v += in[(size_t)s * numelC + i + e];
      out[i + e] = v;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_splitk_reduce(StepArgs a, SplitKArgs sk) {
  __shared__ double red[4];
  const int r = blockIdx.y;
  const T scA = producer_scale<T>(a.partA, a.PA, a.strideA, a.numelA, a.min_norm, r);
  const T scB = producer_scale<T>(a.partB, a.PB, a.strideB, a.numelB, a.min_norm, r);
  const T iA = (T)1 / scA, iB = (T)1 / scB;
  T* __restrict__ C = (T*)a.ptrs[(size_t)r * a.n_tensors + a.idC];
  const T* __restrict__ slab = (const T*)sk.slab + (size_t)r * sk.S * sk.numelC;
  const int64_t per = ((sk.numelC + gridDim.x - 1) / gridDim.x + 3) & ~(int64_t)3;  // multiple of 4
  const int64_t lo = (int64_t)blockIdx.x * per, hi = min(sk.numelC, lo + per);
  T asum = 0;
  if ((sk.numelC & 3) == 0 && sk.S <= 16) {  // 16-byte vectors (slabs and C are 256-byte aligned)
    if (sk.S <= 4) asum = splitk_reduce_span<T, 4, 4>(slab, C, sk.numelC, sk.S, lo, hi, iA, iB);
    else if (sk.S <= 8) asum = splitk_reduce_span<T, 8, 2>(slab, C, sk.numelC, sk.S, lo, hi, iA, iB);
    else asum = splitk_reduce_span<T, 16, 1>(slab, C, sk.numelC, sk.S, lo, hi, iA, iB);
  } else {
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
      T v = slab[i];
      for (int s = 1; s < sk.S; ++s) v += slab[(size_t)s * sk.numelC + i];
      v = (v * iA) * iB;
      C[i] = v;
      asum += fabs(v);
    }
  }
  const double tot = block_sum((double)asum, red);
  if (threadIdx.x == 0) a.partC[(size_t)r * a.partC_stride + blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------
// K-mfma-f64: 128x64 workgroup tile, 4 waves (2x2), each wave 64x32 = 4x2
// v_mfma_f64_16x16x4_f64 accumulators (64 registers), BK = 16, the same pipeline as the fp32
// kernel: k-offset table entries two tiles ahead, unconditional 16-byte loads where an operand is
// unit-stride (mode 1 along the free index, mode 2 along k; mode 0 = scalar gather), masks applied
// when the staged tile is written to LDS, fragment reads one k-step ahead of the MFMAs.
// LDS images are [k][rows + pad] with row strides of 144 / 80 doubles (= 32 banks mod 64), so the
// two k rows a 32-lane group reads fall on disjoint bank halves.
// f64 C/D map (NOT the f32 one): col = lane & 15, row = (lane >> 4) + 4 * reg.
// ---------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int MODE, int ROWS, int LD>
struct TileLoaderD {
  static constexpr int BK = 16;
  static constexpr int NV = ROWS * BK / 256;                 // doubles staged per thread
  static constexpr int VPR = ROWS / 2;                       // mode 1: double2 per k-row
  static constexpr int RPP = 256 / VPR;                      // mode 1: k-rows per pass
  static constexpr int KPP = 256 / ROWS;                     // mode 0: k-rows per pass
  static constexpr int NT = MODE == 1 ? BK / RPP : (MODE == 2 ? 1 : NV);
  static constexpr int NM = MODE == 2 ? NV / 2 : 1;
  static constexpr int kSize = BK * LD;

  double v[NV];
  int kofs[NT];
  int offm[NM];
  bool okm[NM];

  __device__ __forceinline__ void init(const int32_t* __restrict__ om, int m0, int M, int tid) {
    if (MODE == 1) {
      const int gm = m0 + (tid % VPR) * 2;
      offm[0] = om[gm]; okm[0] = gm < M;
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        const int gm = m0 + (tid + i * 256) / (BK / 2);
        offm[i] = om[gm]; okm[i] = gm < M;
      }
    } else {
      const int gm = m0 + tid % ROWS;
      offm[0] = om[gm]; okm[0] = gm < M;
    }
  }
  __device__ __forceinline__ void tab(const int32_t* __restrict__ ok, int k0, int tid) {
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < NT; ++i) kofs[i] = ok[k0 + tid / VPR + RPP * i];
    } else if (MODE == 2) {
      kofs[0] = ok[k0 + (tid % (BK / 2)) * 2];
    } else {
#pragma unroll
      for (int i = 0; i < NT; ++i) kofs[i] = ok[k0 + tid / ROWS + KPP * i];
    }
  }
  __device__ __forceinline__ void load(const double* __restrict__ base) {
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const double2 x = *reinterpret_cast<const double2*>(base + offm[0] + kofs[i]);
        v[2 * i] = x.x; v[2 * i + 1] = x.y;
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        const double2 x = *reinterpret_cast<const double2*>(base + offm[i] + kofs[0]);
        v[2 * i] = x.x; v[2 * i + 1] = x.y;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NT; ++i) v[i] = base[offm[0] + kofs[i]];
    }
  }
  __device__ __forceinline__ void store(double* __restrict__ s, int k0, int K, int tid) const {
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int kr = tid / VPR + RPP * i;
        const bool in = okm[0] && (k0 + kr) < K;
        *reinterpret_cast<double2*>(s + kr * LD + (tid % VPR) * 2) =
            make_double2(in ? v[2 * i] : 0.0, in ? v[2 * i + 1] : 0.0);
      }
    } else if (MODE == 2) {
      const int kc = (tid % (BK / 2)) * 2;
      const bool kin = (k0 + kc) < K;
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        const bool in = okm[i] && kin;
        const int row = (tid + i * 256) / (BK / 2);
        s[kc * LD + row] = in ? v[2 * i] : 0.0;
        s[(kc + 1) * LD + row] = in ? v[2 * i + 1] : 0.0;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int kr = tid / ROWS + KPP * i;
        s[kr * LD + tid % ROWS] = (okm[0] && (k0 + kr) < K) ? v[i] : 0.0;
      }
    }
  }
};

template <int MA, int MB>
__global__ __launch_bounds__(256, 2) void k_mfma_f64(StepArgs a) {
  constexpr int TM = 128, TN = 64, BK = 16, LDA = 144, LDB = 80;
  using LA = TileLoaderD<MA, TM, LDA>;
  using LB = TileLoaderD<MB, TN, LDB>;
  constexpr int SZA = LA::kSize, SZB = LB::kSize;
  __shared__ __attribute__((aligned(16))) double smem[2 * SZA + 2 * SZB + 8];
  __shared__ int s_omC[TM], s_onC[TN];
  double* sA = smem;
  double* sB = smem + 2 * SZA;
  double* red = smem + 2 * SZA + 2 * SZB;

  const int tid = threadIdx.x;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3, q8 = nwg >> 3, r8 = nwg & 7;
  const int pid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const int r = pid / a.blocks_per_replica;
  const int t = pid - r * a.blocks_per_replica;
  const int tiles_mn = a.tiles_m * a.tiles_n;
  const int b = t / tiles_mn;
  const int tt = t - b * tiles_mn;
  const int m0 = (tt / a.tiles_n) * TM;
  const int n0 = (tt % a.tiles_n) * TN;

  const double scA = producer_scale<double>(a.partA, a.PA, a.strideA, a.numelA, a.min_norm, r);
  const double scB = producer_scale<double>(a.partB, a.PB, a.strideB, a.numelB, a.min_norm, r);
  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const double* __restrict__ A = (const double*)tp[a.idA] + a.obA[b];
  const double* __restrict__ B = (const double*)tp[a.idB] + a.obB[b];
  double* __restrict__ C = (double*)tp[a.idC] + a.obC[b];

  if (tid < TM) s_omC[tid] = a.omC[m0 + tid];
  else if (tid < TM + TN) s_onC[tid - TM] = a.onC[n0 + tid - TM];

  LA la;
  LB lb;
  la.init(a.omA, m0, a.M, tid);
  lb.init(a.onB, n0, a.N, tid);

  const int lane = tid & 63, w = tid >> 6;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 32;
  const int l15 = lane & 15, q = lane >> 4;
  f64x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0;

  const int nkt = (a.K + BK - 1) / BK;
  la.tab(a.okA, 0, tid);
  lb.tab(a.okB, 0, tid);
  la.load(A);
  lb.load(B);
  la.tab(a.okA, BK, tid);
  lb.tab(a.okB, BK, tid);
  la.store(sA, 0, a.K, tid);
  lb.store(sB, 0, a.K, tid);
  __syncthreads();

  const int fa = q * LDA + wm + l15, fb = q * LDB + wn + l15;  // + (4 kk) * LD + 16 * tile
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nkt;
    if (more) {
      la.load(A);
      lb.load(B);
      la.tab(a.okA, (kt + 2) * BK, tid);
      lb.tab(a.okB, (kt + 2) * BK, tid);
    }
    __builtin_amdgcn_sched_barrier(0);
    const double* cA = sA + cur * SZA;
    const double* cB = sB + cur * SZB;
    double xa[2][4], xb[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) xa[0][i] = cA[fa + 16 * i];
#pragma unroll
    for (int j = 0; j < 2; ++j) xb[0][j] = cB[fb + 16 * j];
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const int c = kk & 1, nx = c ^ 1;
      if (kk + 1 < BK / 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xa[nx][i] = cA[fa + (kk + 1) * 4 * LDA + 16 * i];
#pragma unroll
        for (int j = 0; j < 2; ++j) xb[nx][j] = cB[fb + (kk + 1) * 4 * LDB + 16 * j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[c][i], xb[c][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      la.store(sA + (cur ^ 1) * SZA, (kt + 1) * BK, a.K, tid);
      lb.store(sB + (cur ^ 1) * SZB, (kt + 1) * BK, a.K, tid);
    }
    __syncthreads();
  }

  // epilogue: lazy rescale (reciprocal multiplies, as in the fp32 kernel), table-driven stores
  const double iA = 1.0 / scA, iB = 1.0 / scB;
  double asum = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wn + j * 16 + l15;
      const bool cin = n0 + col < a.N;
      const int offn = s_onC[col];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = wm + i * 16 + q + 4 * e;
        if (cin && m0 + row < a.M) {
          const double v = (acc[i][j][e] * iA) * iB;
          C[s_omC[row] + offn] = v;
          asum += fabs(v);
        }
      }
    }
  const double tot = block_sum(asum, red);
  if (tid == 0) a.partC[(size_t)r * a.partC_stride + t] = tot;
}



// ---------------------------------------------------------------------------
// K-mfma-f64-sk: the latency mode in fp64 (the dtype a NumPy user's single network arrives in): 64x64
// tiles, 4 waves of 32x32 = 2x2 v_mfma_f64_16x16x4_f64 accumulators, BK = 16, K split S ways; partial
// tiles go to slab s, k_splitk_reduce<double> sums them in a fixed order.
// ---------------------------------------------------------------------------
template <int MA, int MB>
__global__ __launch_bounds__(256) void k_mfma_f64_sk(StepArgs a, SplitKArgs sk) {
  constexpr int T = 64, BK = 16, LD = 80;
  using LA = TileLoaderD<MA, T, LD>;
  using LB = TileLoaderD<MB, T, LD>;
  constexpr int SZA = LA::kSize, SZB = LB::kSize;
  __shared__ __attribute__((aligned(16))) double smem[2 * SZA + 2 * SZB];
  __shared__ int s_omC[T], s_onC[T];
  double* sA = smem;
  double* sB = smem + 2 * SZA;

  const int tid = threadIdx.x;
  const int per_rep = sk.tiles_per_replica * sk.S;
  const int pid = blockIdx.x;
  const int r = pid / per_rep;
  const int rem = pid - r * per_rep;
  const int t = rem / sk.S;
  const int s = rem - t * sk.S;
  const int tiles_mn = sk.tiles_m * sk.tiles_n;
  const int b = t / tiles_mn;
  const int tt = t - b * tiles_mn;
  const int m0 = (tt / sk.tiles_n) * T;
  const int n0 = (tt % sk.tiles_n) * T;
  const int kbeg = s * sk.kchunk;
  const int kend = min(a.K, kbeg + sk.kchunk);

  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const double* __restrict__ A = (const double*)tp[a.idA] + a.obA[b];
  const double* __restrict__ B = (const double*)tp[a.idB] + a.obB[b];
  double* __restrict__ C = (double*)sk.slab + ((size_t)r * sk.S + s) * sk.numelC + a.obC[b];

  if (tid < T) s_omC[tid] = a.omC[m0 + tid];
  else if (tid < 2 * T) s_onC[tid - T] = a.onC[n0 + tid - T];

  LA la;
  LB lb;
  la.init(a.omA, m0, a.M, tid);
  lb.init(a.onB, n0, a.N, tid);

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

  const int nkt = (kend - kbeg + BK - 1) / BK;   // >= 1 (host guarantees kbeg < K)
  la.tab(a.okA, kbeg, tid);
  lb.tab(a.okB, kbeg, tid);
  la.load(A);
  lb.load(B);
  la.tab(a.okA, kbeg + BK, tid);
  lb.tab(a.okB, kbeg + BK, tid);
  la.store(sA, kbeg, kend, tid);
  lb.store(sB, kbeg, kend, tid);
  __syncthreads();

  const int fa = q * LD + wm + l15, fb = q * LD + wn + l15;  // + (4 kk) * LD + 16 * block
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nkt;
    if (more) {
      la.load(A);
      lb.load(B);
      la.tab(a.okA, kbeg + (kt + 2) * BK, tid);
      lb.tab(a.okB, kbeg + (kt + 2) * BK, tid);
    }
    __builtin_amdgcn_sched_barrier(0);
    const double* cA = sA + cur * SZA;
    const double* cB = sB + cur * SZB;
    double xa[2][2], xb[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { xa[0][i] = cA[fa + 16 * i]; xb[0][i] = cB[fb + 16 * i]; }
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const int c = kk & 1, nx = c ^ 1;
      if (kk + 1 < BK / 4) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          xa[nx][i] = cA[fa + (kk + 1) * 4 * LD + 16 * i];
          xb[nx][i] = cB[fb + (kk + 1) * 4 * LD + 16 * i];
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[c][i], xb[c][j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      la.store(sA + (cur ^ 1) * SZA, kbeg + (kt + 1) * BK, kend, tid);
      lb.store(sB + (cur ^ 1) * SZB, kbeg + (kt + 1) * BK, kend, tid);
    }
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = wn + 16 * j + l15;
    if (n0 + col < a.N) {
      const int offn = s_onC[col];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = wm + 16 * i + q + 4 * e;
          if (m0 + row < a.M) C[s_omC[row] + offn] = acc[i][j][e];
        }
    }
  }
}

}  // namespace ctn
