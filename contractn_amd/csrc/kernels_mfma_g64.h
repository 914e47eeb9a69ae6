// kernels_mfma_g64.h - fp64 MFMA GEMM fed by direct-to-LDS loads (LDS-DMA): the fp64 sibling of k_mfma_f32_g
// Part of the gfx950 contraction engine (see engine.hip for the overview).
#pragma once
#include "kernels_mfma_g.h"

namespace ctn {

// ---------------------------------------------------------------------------
// K-mfma-f64-g: 128 x 128 workgroup tile, 4 waves (2x2), each wave 64 x 64 = 4 x 4
// v_mfma_f64_16x16x4_f64 accumulators (128 registers; 8 fragment reads per 16 MFMAs where the
// 128 x 64 kernel has 6 per 8), BK = 8 (two MFMA k-steps).  Same feeding as the fp32 kernel: operand
// tiles go global -> LDS by global_load_lds_dwordx4 (16 bytes = 2 consecutive rows / columns per
// lane, one wave instruction = one k-row of 128 doubles) into a 3-stage ring; one s_waitcnt vmcnt +
// one raw s_barrier per k-tile, between its two k-steps; fragment read-ahead across k-tiles.
// k-rows are 132 doubles apart in LDS (an LDS-DMA destination is lane-linear only WITHIN one
// instruction, and one instruction is one k-row), which spreads the four k-rows that a fragment read
// touches over different banks.
//
// Residency: the 51.7 KB of LDS admit three workgroups per CU; the mode-1 instance is bounded to 168 registers
// for that (15 spilled, all outside the k-loop): MPS shapes at 256 replicas 55.3 / 57.9 -> 56.8 / 58.9 TFLOP/s, 8192^3
// 69.4 -> 71.4, the fp64 headline 2362 -> 2401 /s.  The k-contiguous instances spill into the loop at that bound
// (row-major 4096^3: 64.7 -> 55.4) and stay at two.
//
// Eligibility (planner): both operands unit-stride along their free index ("mode 1", pairs of
// doubles), K >= 16, C pair-storable; ragged M / N / K handled as in the fp32 kernel.
// ---------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int DM = 128, DN = 128, DK = 8, DST = 3;
constexpr int DROW = 132;                       // doubles between k-rows in LDS (128 + 4 pad)
constexpr int D_SZ = DK * DROW;                 // one operand tile
constexpr int D_STG = 2 * D_SZ;                 // A | B

__device__ __forceinline__ void glds16d(const double* g, double* lds) {
  __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)lds, 16, 0, 0);
}

// MA / MB = 2: the operand is unit-stride along k (a row-major A): requests along k (one lane = one row, 2
// consecutive k), LDS image [k / 2][rows][2] with the four chunks 260 doubles apart, and BOTH operands read in
// the permuted k order k(ks, q) = 2 q + ks (q = lane / 16), which puts a lane's two k-steps into one chunk.
template <int MA = 1, int MB = 1>
__global__ __launch_bounds__(256, (MA == 1 && MB == 1) ? 3 : 2) void k_mfma_f64_g(StepArgs a) {
  constexpr bool PERM = MA == 2 || MB == 2;
  constexpr int DCH = 260;   // doubles between the k-chunks of a k-contiguous operand's image (256 + 4 pad)
  // ONE LDS object: [stage 0 A|B][stage 1 A|B][stage 2 A|B][red 4 doubles]
  __shared__ __attribute__((aligned(16))) double smem[DST * D_STG + 4 + 128];
  double* red = smem + DST * D_STG;
  double* zeros = red + 4;   // 128 zeros: where the fragment reads of k-rows beyond a ragged K are pointed

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3, q8 = nwg >> 3, r8 = nwg & 7;
  const int pid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const int r = pid / a.blocks_per_replica;       // here: 128 x 128 tiles per replica
  const int t = pid - r * a.blocks_per_replica;
  const int tiles_mn = a.tiles_m * a.tiles_n;
  const int b = t / tiles_mn;
  const int tt = t - b * tiles_mn;
  const int tm = tt / a.tiles_n, tn = tt % a.tiles_n;
  const int m0 = tm * DM, n0 = tn * DN;

  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const double* __restrict__ A = (const double*)tp[a.idA] + a.obA[b];
  const double* __restrict__ B = (const double*)tp[a.idB] + a.obB[b];
  double* __restrict__ C = (double*)tp[a.idC] + a.obC[b];

  const int l15 = lane & 15, q = lane >> 4;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  if (tid < 128) zeros[tid] = 0.0;   // published by the first barrier

  // loader: wave w fills k-rows 2w, 2w+1 of both operand tiles; lane l brings rows / columns 2l, 2l+1.
  // (wave-uniform 64-bit base) + (per-lane unsigned 32-bit byte offset): scalar-base loads
  uint32_t offA = (uint32_t)a.omA[m0 + 2 * lane] * 8u;
  uint32_t offB = (uint32_t)a.onB[n0 + 2 * lane] * 8u;
  asm volatile("" : "+v"(offA), "+v"(offB));   // consumed before the first request (see k_mfma_f32_g)
  uint32_t offA2[2] = {0, 0}, offB2[2] = {0, 0};   // k-contiguous operand: lane = row / column 64 g + lane
  if constexpr (MA == 2) {
    offA2[0] = (uint32_t)a.omA[m0 + lane] * 8u; offA2[1] = (uint32_t)a.omA[m0 + 64 + lane] * 8u;
    asm volatile("" : "+v"(offA2[0]), "+v"(offA2[1]));
  }
  if constexpr (MB == 2) {
    offB2[0] = (uint32_t)a.onB[n0 + lane] * 8u; offB2[1] = (uint32_t)a.onB[n0 + 64 + lane] * 8u;
    asm volatile("" : "+v"(offB2[0]), "+v"(offB2[1]));
  }
  const char* const Ac = reinterpret_cast<const char*>(A);
  const char* const Bc = reinterpret_cast<const char*>(B);
  const_i32_ptr okA = (const_i32_ptr)(a.okA + 2 * w);
  const_i32_ptr okB = (const_i32_ptr)(a.okB + 2 * w);
  const int nkt = (a.K + DK - 1) / DK;
  const bool ktail = (a.K % DK) != 0;
  const int krem = a.K - (nkt - 1) * DK;

  int ka[2], kb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { ka[i] = okA[i]; kb[i] = okB[i]; }

  auto request = [&](int kt_next, int stage) {
    double* sa = smem + stage * D_STG + (2 * w) * DROW;
    double* sb = sa + D_SZ;
    uint32_t oA = offA, oB = offB;
    asm volatile("" : "+v"(oA), "+v"(oB));
    if constexpr (MA == 2) {   // wave w brings k-chunk w (k = 2w, 2w+1) of all 128 rows: 2 requests of 64 rows
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        uint32_t o = offA2[g];
        asm volatile("" : "+v"(o));
        glds16d(reinterpret_cast<const double*>(Ac + (int64_t)ka[0] * 8 + o), smem + stage * D_STG + w * DCH + 128 * g);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16d(reinterpret_cast<const double*>(Ac + (int64_t)ka[i] * 8 + oA), sa + i * DROW);
    }
    if constexpr (MB == 2) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        uint32_t o = offB2[g];
        asm volatile("" : "+v"(o));
        glds16d(reinterpret_cast<const double*>(Bc + (int64_t)kb[0] * 8 + o), smem + stage * D_STG + D_SZ + w * DCH + 128 * g);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16d(reinterpret_cast<const double*>(Bc + (int64_t)kb[i] * 8 + oB), sb + i * DROW);
    }
    const int k0 = kt_next * DK;  // the tables are padded past K
#pragma unroll
    for (int i = 0; i < 2; ++i) { ka[i] = okA[k0 + i]; kb[i] = okB[k0 + i]; }
  };

  f64x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0;

  request(1, 0);
  request(2, 1);                                   // nkt >= 2 is guaranteed by the planner
  __builtin_amdgcn_sched_barrier(0);
  // epilogue operands, requested behind the first two k-tiles (see k_mfma_f32_g): 2 + 8 vector loads
  double pva = 0.0, pvb = 0.0;
  // (lane l takes partials l, l + 64, ...: one load each for the usual <= 64 partials; the first index is clamped so
  // that every lane requests something and the masking happens after the loop, where the values are used)
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
  int offn[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) offn[j] = a.onC[n0 + wn + 16 * j + l15];
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0x0F78);              // vmcnt(8): k-tile 0 landed; k-tile 1 (4) + >= 4 table loads may be out
  __builtin_amdgcn_s_barrier();

  // this lane's fragment base (doubles, lane group q folded in), the step to the tile's second k-step and the
  // distance between 16-row fragment blocks, per operand layout:
  //   [k][DROW], plain order k = 4 ks + q      : q DROW + row,     + 4 DROW, 16
  //   [k][DROW], permuted order k = 2 q + ks   : 2 q DROW + row,   + DROW,   16
  //   [k / 2][rows][2] (k-contiguous operand)  : q DCH + 2 row,    + 1,      32
  const int fa0 = MA == 2 ? q * DCH + 2 * (wm + l15) : (PERM ? 2 * q : q) * DROW + wm + l15;
  const int fb0 = D_SZ + (MB == 2 ? q * DCH + 2 * (wn + l15) : (PERM ? 2 * q : q) * DROW + wn + l15);
  constexpr int ksA = MA == 2 ? 1 : (PERM ? DROW : 4 * DROW), ksB = MB == 2 ? 1 : (PERM ? DROW : 4 * DROW);
  constexpr int blkA = MA == 2 ? 32 : 16, blkB = MB == 2 ? 32 : 16;
  int st_cur = 0, st_nxt = 1, st_req = 2;
  double xa[2][4], xb[2][4];
  {
    const double* c = smem;
#pragma unroll
    for (int i = 0; i < 4; ++i) xa[0][i] = c[fa0 + blkA * i];
#pragma unroll
    for (int j = 0; j < 4; ++j) xb[0][j] = c[fb0 + blkB * j];
  }
  for (int kt = 0; kt < nkt; ++kt) {
    const double* cur = smem + st_cur * D_STG;
    const double* nxt = smem + st_nxt * D_STG;
    const bool tail_cur = ktail && kt == nkt - 1, tail_nxt = ktail && kt + 2 == nkt;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks & 1, nx = c ^ 1;
      // fragment read-ahead: second k-step of this k-tile, or first k-step of the next one.  Lanes whose
      // k-row lies beyond a ragged K read the zero block instead (LDS-DMA cannot mask; selecting the
      // ADDRESS keeps the reads free of any wait, unlike zeroing the registers afterwards)
      const double *pa, *pb;
      if (ks == 0) {
        pa = cur + fa0 + ksA; pb = cur + fb0 + ksB;
        if (tail_cur && (PERM ? 2 * q + 1 : 4 + q) >= krem) { pa = zeros; pb = zeros; }
      } else {
        pa = nxt + fa0; pb = nxt + fb0;
        if (tail_nxt && (PERM ? 2 * q : q) >= krem) { pa = zeros; pb = zeros; }
      }
      if (ks == 0 || kt + 1 < nkt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xa[nx][i] = pa[blkA * i];
#pragma unroll
        for (int j = 0; j < 4; ++j) xb[nx][j] = pb[blkB * j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[c][i], xb[c][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // 8 ds_read_b64 pair up into ds_read2_b64
      __builtin_amdgcn_sched_group_barrier(0x008, 15, 0);
      if (ks == 0) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): k-tile kt+1 - this wave's requests, a tile old
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nkt) request(kt + 3, st_req);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    st_cur = st_nxt;
    st_nxt = st_req;
    st_req = st_req == DST - 1 ? 0 : st_req + 1;
  }

  // epilogue: lazy rescale, table-driven 8-byte stores (an accumulator register holds rows q + 4e of a
  // 16-wide column block: 4 rows x 128 contiguous bytes per store instruction), abs-sum partial
  pva = lane < a.PA ? pva : 0.0;
  pvb = lane < a.PB ? pvb : 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { pva += __shfl_xor(pva, o, 64); pvb += __shfl_xor(pvb, o, 64); }
  const double scA = (a.partA && pva > a.min_norm) ? pva / a.numelA : 1.0;   // = producer_scale<double>()
  const double scB = (a.partB && pvb > a.min_norm) ? pvb / a.numelB : 1.0;
  const double iA = 1.0 / scA, iB = 1.0 / scB;
  double asum = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = wm + 16 * i + q + 4 * e;
      const bool rin = m0 + row < a.M;
      double* __restrict__ crow = C + a.omC[m0 + row];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (rin && n0 + wn + 16 * j + l15 < a.N) {
          const double v = (acc[i][j][e] * iA) * iB;
          crow[offn[j]] = v;
          asum += fabs(v);
        }
    }
  const double tot = block_sum(asum, red);
  if (tid == 0) {  // this tile covers up to two of the planner's 128 x 64 partial slots
    const int tm128 = (a.M + 127) / 128, tn64 = (a.N + 63) / 64;
    double* pc = a.partC + (size_t)r * a.partC_stride + (size_t)b * tm128 * tn64 + (size_t)tm * tn64;
    pc[2 * tn] = tot;
    if (2 * tn + 1 < tn64) pc[2 * tn + 1] = 0.0;
  }
}

}  // namespace ctn
