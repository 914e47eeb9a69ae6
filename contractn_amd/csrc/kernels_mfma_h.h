// kernels_mfma_h.h - fp32 MFMA GEMM for launches of about ONE 128 x 128 tile per CU: K split over the two halves of an
// 8-wave workgroup, operands by LDS-DMA.  Part of the gfx950 contraction engine (see engine.hip for the overview).
#pragma once
#include <type_traits>

#include "kernels_mfma_g.h"

namespace ctn {

// ---------------------------------------------------------------------------
// K-mfma-f32-h.  A step whose output is about 256 tiles of 128 x 128 - one MPS site applied to a batch of 4096 inputs
// (reference README Fig. 1d: `bl,plr->bpr` + `bpr,bp->br`, 4096 x 1024 x 256), the K = 256 step of a few networks in
// flight - runs as ONE round of workgroups: all tiles start and end together, nothing covers their prologues and
// epilogues, and the register-staged kernel's 128 x 64 tiles (2 per CU, 64 x 32 per wave: 3 LDS fragment reads per 2
// MFMAs, one barrier per 16 MFMAs) keep the matrix pipes at 0.69 of their rate in between.  The large-tile kernel
// (256 x 128, kernels_mfma_g.h) would leave half the CUs idle.  This form gives every CU one 128 x 128 tile and still
// two waves per SIMD with 64 x 64 per wave (4 reads per 4 MFMAs): the 8 waves are 2 K-HALVES x (2 x 2) sub-tiles -
// waves 0-3 sum k in [0, K/2), waves 4-7 k in [K/2, K), each half with its own operand tiles in the 3-stage LDS-DMA
// ring (global_load_lds_dwordx4, no staging registers), one raw s_barrier per k-tile in the middle of its MFMA phase -
// and the two partial accumulators of a sub-tile meet once, through LDS, after the loop: each partner hands over
// half of its 64 x 64 block and finishes the other half (rescale, store, abs-sum), so the epilogue is split over all
// 8 waves as well.  The sum order is fixed (first half + second half): bit-reproducible.
//
// MA / MB: 1 = operand unit-stride along its free index (LDS image [k][128]), 2 = along k (image [k/4][128][4], both
// operands then read in the permuted k order of kernels_mfma_g.h).  EPW: the epilogue-summed form (planner pattern
// C with the columns ordered (.., u_hi, p, u_lo), Step::epw_split): the accumulators are re-weighted by W[row][p] and
// summed over p - across the two lane halves and, for p = 4, two accumulator groups - on their way out.
// Launch conditions (engine.hip, h_form): K % 32 == 0, K >= 64, C vector-storable, operands of at most 2^30 elements.
// ---------------------------------------------------------------------------
constexpr int HT = 128, HK = 16, HST = 3;

template <int MA, int MB, int EPW>   // EPW: 0 plain, 2 / 4 = extent of the label summed in the epilogue
__global__ __launch_bounds__(512, 1) void k_mfma_f32_h(StepArgs a) {
  constexpr bool PERM = MA == 2 || MB == 2;
  constexpr int SZ = HK * HT;                 // one operand tile of one K half, in floats
  constexpr int STG = 4 * SZ;                 // a ring stage: [A half 0][B half 0][A half 1][B half 1]
  __shared__ __attribute__((aligned(16))) float smem[HST * STG + 16];
  double* red = reinterpret_cast<double*>(smem + HST * STG);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = w >> 2, wq = w & 3;
  const int wm = (wq >> 1) * 64, wn = (wq & 1) * 64;
  const int l31 = lane & 31, h = lane >> 5;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3, q8 = nwg >> 3, r8 = nwg & 7;
  const int pid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const int r = pid / a.blocks_per_replica;
  const int t = pid - r * a.blocks_per_replica;
  const int tiles_mn = a.tiles_m * a.tiles_n;
  const int b = t / tiles_mn;
  const int tt = t - b * tiles_mn;
  const int m0 = (tt / a.tiles_n) * HT, n0 = (tt % a.tiles_n) * HT;

#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) a.dbg[(size_t)pid * 8 + 0] = __builtin_amdgcn_s_memtime();
#endif
  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const float* __restrict__ A = (const float*)tp[a.idA] + a.obA[b];
  const float* __restrict__ B = (const float*)tp[a.idB] + a.obB[b];
  float* __restrict__ C = (float*)tp[a.idC] + a.obC[b];
  const char* const Ac = reinterpret_cast<const char*>(A);
  const char* const Bc = reinterpret_cast<const char*>(B);

  // loader: wave (kh, wq) brings k-rows 4 wq .. 4 wq + 3 of its half's A and B tiles: two requests per operand -
  // mode 1: lanes 0-31 one k-row (4 rows of the tile each), lanes 32-63 the next; mode 2: 64 rows, 4 consecutive k
  uint32_t offA[2], offB[2];
  if constexpr (MA == 2) {
    offA[0] = (uint32_t)a.omA[m0 + lane] * 4u; offA[1] = (uint32_t)a.omA[m0 + 64 + lane] * 4u;
  } else {
    offA[0] = offA[1] = (uint32_t)a.omA[m0 + 4 * l31] * 4u;
  }
  if constexpr (MB == 2) {
    offB[0] = (uint32_t)a.onB[n0 + lane] * 4u; offB[1] = (uint32_t)a.onB[n0 + 64 + lane] * 4u;
  } else {
    offB[0] = offB[1] = (uint32_t)a.onB[n0 + 4 * l31] * 4u;
  }
  asm volatile("" : "+v"(offA[0]), "+v"(offA[1]), "+v"(offB[0]), "+v"(offB[1]));   // used before the first LDS-DMA
  const int Kh = a.K >> 1;                    // K % 32 == 0: both halves are whole k-tiles
  const int nkt = Kh / HK;
  const_i32_ptr okA = (const_i32_ptr)(a.okA + kh * Kh + 4 * wq);
  const_i32_ptr okB = (const_i32_ptr)(a.okB + kh * Kh + 4 * wq);
  int ka[4], kb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { ka[i] = okA[i]; kb[i] = okB[i]; }

  auto request = [&](int kt_next, int stage) {   // this wave's four requests of one k-tile, then the next tile's table entries
    float* sa = smem + stage * STG + kh * 2 * SZ;
    float* sb = sa + SZ;
    if constexpr (MA == 2) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        uint32_t o = offA[q];
        asm volatile("" : "+v"(o));
        glds16(reinterpret_cast<const float*>(Ac + (int64_t)ka[0] * 4 + o), sa + (wq * HT + 64 * q) * 4);
      }
    } else {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int lo = min(ka[2 * p], ka[2 * p + 1]);
        const uint32_t d0 = (uint32_t)(ka[2 * p] - lo) * 4u, d1 = (uint32_t)(ka[2 * p + 1] - lo) * 4u;
        glds16(reinterpret_cast<const float*>(Ac + (int64_t)lo * 4 + (offA[0] + (h ? d1 : d0))), sa + (4 * wq + 2 * p) * HT);
      }
    }
    if constexpr (MB == 2) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        uint32_t o = offB[q];
        asm volatile("" : "+v"(o));
        glds16(reinterpret_cast<const float*>(Bc + (int64_t)kb[0] * 4 + o), sb + (wq * HT + 64 * q) * 4);
      }
    } else {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int lo = min(kb[2 * p], kb[2 * p + 1]);
        const uint32_t d0 = (uint32_t)(kb[2 * p] - lo) * 4u, d1 = (uint32_t)(kb[2 * p + 1] - lo) * 4u;
        glds16(reinterpret_cast<const float*>(Bc + (int64_t)lo * 4 + (offB[0] + (h ? d1 : d0))), sb + (4 * wq + 2 * p) * HT);
      }
    }
    const int k0 = kt_next * HK;               // (the tables are padded by 64 entries past K)
#pragma unroll
    for (int i = 0; i < 4; ++i) { ka[i] = okA[k0 + i]; kb[i] = okB[k0 + i]; }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  request(1, 0);
  request(2, 1);                               // nkt >= 2: the launcher takes this form for K >= 64 only
  __builtin_amdgcn_sched_barrier(0);
  // everything the epilogue needs, behind the first two k-tiles: the producers' partial sums, the C offsets of the
  // half block this wave will finish (the second one of the first K half, the first one of the second), the weights' rows
  const int jk = kh ^ 1;                       // the 32-column block of its 64 x 64 sub-tile that this wave keeps
  double pva = 0.0, pvb = 0.0, pvw = 0.0;
  if (a.partA) {
    const double* __restrict__ pr = a.partA + (size_t)r * a.strideA;
    pva = pr[min(lane, a.PA - 1)];
    if (a.PA > 64)
      for (int i = lane + 64; i < a.PA; i += 64) pva += pr[i];
  }
  if (a.partB) {
    const double* __restrict__ pr = a.partB + (size_t)r * a.strideB;
    pvb = pr[min(lane, a.PB - 1)];
    if (a.PB > 64)
      for (int i = lane + 64; i < a.PB; i += 64) pvb += pr[i];
  }
  if (EPW && a.partA2) {
    const double* __restrict__ pr = a.partA2 + (size_t)r * a.strideA2;
    pvw = pr[min(lane, a.PA2 - 1)];
    if (a.PA2 > 64)
      for (int i = lane + 64; i < a.PA2; i += 64) pvw += pr[i];
  }
  int offm[2], offn[4], offw[2] = {0, 0};
#pragma unroll
  for (int i = 0; i < 2; ++i) offm[i] = a.omC[m0 + wm + 32 * i + l31];
  constexpr int P = EPW;
  if constexpr (EPW) {
    // columns (.., u_hi, p, u_lo): a group of 4 P columns is one u_hi; this lane stores group(s) h (P = 4) or 2 h, 2 h + 1 (P = 2)
    offn[0] = a.onC[n0 + wn + 32 * jk + 16 * h];
    offn[1] = a.onC[n0 + wn + 32 * jk + 16 * h + 8];
    offn[2] = offn[3] = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) offw[i] = a.omA2[m0 + wm + 32 * i + l31];
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) offn[g] = a.onC[n0 + wn + 32 * jk + 8 * g + 4 * h];
  }
  __builtin_amdgcn_sched_barrier(0);
  // k-tile 0 has landed (this wave's share): everything but the youngest request group (4) and at least 4 of the
  // (at least 6) loads behind it - counted low on purpose: the compiler may merge two of those.
  // simm16 = vmcnt[3:0] | 7 << 4 | 15 << 8 | vmcnt[5:4] << 14
  __builtin_amdgcn_s_waitcnt(0x0F78);          // vmcnt(8)
  __builtin_amdgcn_s_barrier();
#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) a.dbg[(size_t)pid * 8 + 1] = __builtin_amdgcn_s_memtime();
#endif
  float4 wrow[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  if constexpr (EPW) {                         // the rows' weights (a network input [rows][P], P = 2 or 4)
    const float* __restrict__ W = (const float*)tp[a.idA2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float* wp = W + offw[i];
      if constexpr (P == 4) wrow[i] = *reinterpret_cast<const float4*>(wp);
      else { const float2 x = *reinterpret_cast<const float2*>(wp); wrow[i] = make_float4(x.x, x.y, 0.f, 0.f); }
    }
  }

  // main loop: as in k_mfma_f32_g - fragments one k-step ahead, the barrier in the middle of a k-tile's MFMA phase
  const float* const ring = smem + kh * 2 * SZ;
  const int fa0 = MA == 2 ? (h * HT + wm + l31) * 4 : ((PERM ? 4 * h : h) * HT + wm + l31);
  const int fb0 = MB == 2 ? (h * HT + wn + l31) * 4 : ((PERM ? 4 * h : h) * HT + wn + l31);
  constexpr int blkA = MA == 2 ? 128 : 32, blkB = MB == 2 ? 128 : 32;
  int st_cur = 0, st_nxt = 1, st_req = 2;
  float fa[2][2], fb[2][2];
  {
    const float* cA = ring + fa0;
    const float* cB = ring + SZ + fb0;
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[0][i] = cA[blkA * i];
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[0][j] = cB[blkB * j];
  }
  for (int kt = 0; kt < nkt; ++kt) {
    const float* cA = ring + st_cur * STG + fa0;
    const float* cB = ring + st_cur * STG + SZ + fb0;
    const float* nA = ring + st_nxt * STG + fa0;
    const float* nB = ring + st_nxt * STG + SZ + fb0;
#pragma unroll
    for (int kk = 0; kk < HK / 2; ++kk) {
      const int c = kk & 1, nx = c ^ 1;
      constexpr auto frag = [](int step, int mode) {
        const int j = step / 4, e = step % 4;
        return mode == 2 ? (2 * j * HT) * 4 + e : ((MA == 2 || MB == 2) ? 8 * j + e : 2 * step) * HT;
      };
      if (kk + 1 < HK / 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[nx][i] = cA[frag(kk + 1, MA) + blkA * i];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[nx][j] = cB[frag(kk + 1, MB) + blkB * j];
      } else if (kt + 1 < nkt) {               // first k-step of the next k-tile (published by this tile's barrier)
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[nx][i] = nA[blkA * i];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[nx][j] = nB[blkB * j];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c][j], fa[c][i], acc[i][j], 0, 0, 0);   // C^T blocks
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      if (kk == HK / 4 - 1) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xF70) /* vmcnt(0) */;   // k-tile kt+1: this wave's requests, a tile old
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nkt) request(kt + 3, st_req);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    st_cur = st_nxt;
    st_nxt = st_req;
    st_req = st_req == HST - 1 ? 0 : st_req + 1;
  }
#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) a.dbg[(size_t)pid * 8 + 2] = __builtin_amdgcn_s_memtime();
#endif

  // the two K halves of a sub-tile meet: each partner hands over one 32-column block and finishes the other
  f32x16 mine[2], give[2];
  if (kh == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { mine[i] = acc[i][1]; give[i] = acc[i][0]; }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) { mine[i] = acc[i][0]; give[i] = acc[i][1]; }
  }
  __builtin_amdgcn_s_barrier();                // every wave has read its last fragments: the ring is free
  {
    // [block i][register quad q][lane][4]: 16-byte LDS accesses, lane-linear (conflict-free)
    float4* xo = reinterpret_cast<float4*>(smem + w * 2048) + lane;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        xo[(i * 4 + q) * 64] = make_float4(give[i][4 * q], give[i][4 * q + 1], give[i][4 * q + 2], give[i][4 * q + 3]);
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0) only: nothing else is outstanding that the partner reads
  __builtin_amdgcn_s_barrier();
  {
    const float4* xi = reinterpret_cast<const float4*>(smem + (w ^ 4) * 2048) + lane;
    // fixed order: (first K half) + (second K half), whichever of the two waves adds
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 o = xi[(i * 4 + q) * 64];
        const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) mine[i][4 * q + e] = kh == 0 ? mine[i][4 * q + e] + ov[e] : ov[e] + mine[i][4 * q + e];
      }
  }
#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) a.dbg[(size_t)pid * 8 + 4] = __builtin_amdgcn_s_memtime();
#endif
  // epilogue on the kept 64 x 32 block: lazy rescale, 16-byte stores straight from the accumulators, abs-sum partial
  pva = lane < a.PA ? pva : 0.0;
  pvb = lane < a.PB ? pvb : 0.0;
  pvw = (EPW && lane < a.PA2) ? pvw : 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { pva += __shfl_xor(pva, o, 64); pvb += __shfl_xor(pvb, o, 64); pvw += __shfl_xor(pvw, o, 64); }
  const float nA_ = (float)pva, nB_ = (float)pvb, nW_ = (float)pvw;   // exactly producer_scale<float>()
  const float scA = (a.partA && nA_ > (float)a.min_norm) ? nA_ / (float)a.numelA : 1.f;
  const float scB = (a.partB && nB_ > (float)a.min_norm) ? nB_ / (float)a.numelB : 1.f;
  const float scW = (EPW && a.partA2 && nW_ > (float)a.min_norm) ? nW_ / (float)a.numelA2 : 1.f;
  const float iA = 1.0f / scA, iB = 1.0f / scB, iW = 1.0f / scW;
  float asum = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const bool rin = m0 + wm + 32 * i + l31 < a.M;
    float* __restrict__ row = C + offm[i];
    if constexpr (EPW) {
      const float4 wv = wrow[i];
      if constexpr (P == 4) {
        // group u = columns 16 u .. 16 u + 15 = p (0..3) x u_lo (0..3): this lane half holds p = h (registers 8 u ..) and
        // p = 2 + h (registers 8 u + 4 ..); the other two p sit in the other half of the wave
        const float w0 = (h ? wv.y : wv.x) * iW, w1 = (h ? wv.w : wv.z) * iW;
        float x[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            x[u][e] = ((mine[i][8 * u + e] * iA) * iB) * w0 + ((mine[i][8 * u + 4 + e] * iA) * iB) * w1;
            x[u][e] += __shfl_xor(x[u][e], 32, 64);
          }
        float4 v;                                // this lane half stores group h
        v.x = h ? x[1][0] : x[0][0]; v.y = h ? x[1][1] : x[0][1]; v.z = h ? x[1][2] : x[0][2]; v.w = h ? x[1][3] : x[0][3];
        if (rin && n0 + wn + 32 * jk + 16 * h < a.N) {
          *reinterpret_cast<float4*>(row + offn[0]) = v;
          asum += (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w));
        }
      } else {
        // P = 2: group u = columns 8 u .. 8 u + 7 = p (0, 1) x u_lo: this lane half holds p = h in registers 4 u ..
        const float w0 = (h ? wv.y : wv.x) * iW;
        float x[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            x[u][e] = ((mine[i][4 * u + e] * iA) * iB) * w0;
            x[u][e] += __shfl_xor(x[u][e], 32, 64);
          }
#pragma unroll
        for (int q = 0; q < 2; ++q) {            // this lane half stores groups 2 h + q
          float4 v;
          v.x = h ? x[2 + q][0] : x[q][0]; v.y = h ? x[2 + q][1] : x[q][1];
          v.z = h ? x[2 + q][2] : x[q][2]; v.w = h ? x[2 + q][3] : x[q][3];
          if (rin && n0 + wn + 32 * jk + 16 * h + 8 * q < a.N) {
            *reinterpret_cast<float4*>(row + offn[q]) = v;
            asum += (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w));
          }
        }
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v;
        v.x = (mine[i][4 * g + 0] * iA) * iB;
        v.y = (mine[i][4 * g + 1] * iA) * iB;
        v.z = (mine[i][4 * g + 2] * iA) * iB;
        v.w = (mine[i][4 * g + 3] * iA) * iB;
        if (rin && n0 + wn + 32 * jk + 8 * g + 4 * h < a.N) {
          *reinterpret_cast<float4*>(row + offn[g]) = v;
          asum += (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w));
        }
      }
    }
  }
#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) a.dbg[(size_t)pid * 8 + 5] = __builtin_amdgcn_s_memtime();
#endif
  // the tile's abs-sum: waves reduce, one LDS hand-off behind a raw barrier (a __syncthreads() would first wait for
  // this wave's stores to drain; they drain while the eight partials are added - in wave order, always the same)
  double part = (double)asum;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) red[w] = part;                // (red[] lies behind the ring: nobody else touches it)
  __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
  if (tid == 0) {
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) tot += red[i];
    a.partC[(size_t)r * a.partC_stride + t] = tot;
  }
#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) {
    a.dbg[(size_t)pid * 8 + 3] = __builtin_amdgcn_s_memtime();
    a.dbg[(size_t)pid * 8 + 6] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

}  // namespace ctn
