// kernels_zip64.h - k_zip_f32 for 32 ... 127 networks in flight: a workgroup owns 64 values of u instead of 128, so that
// the fused pair fills the chip from 64 networks (32 give half a chip).  Part of the gfx950 contraction engine.
#pragma once
#include "kernels_zip.h"

namespace ctn {

// ---------------------------------------------------------------------------
// K-zip64-f32.  The pair of kernels_zip.h,
//
//     T[m1, (q, u)] = sum_k1  E[k1, m1] * X[q, k1, u]          E'[u, n2] = sum_(m1, q)  T[m1, q, u] * Y[q, m1, n2]
//
// with the same mechanics (phase 1 in accumulators, those accumulators used directly as phase 2's MFMA operand, LDS-DMA
// ring, one raw barrier per tile in the middle of its MFMA phase, requests from the waves of one half of the SIMD pairs
// only) but another split of the eight waves: 2 u-blocks of 32 x FOUR quarters of m1.  A wave forms Tq[its 64 m1, its 32 u]
// (2 accumulator blocks; phase-1 tiles are 32 deep, so a tile is still 32 MFMAs per wave) and a partial E'[32 u, 256 n2] over
// its quarter (8 blocks; phase-2 tiles hold 8 rows of each quarter: 32 MFMAs per wave and tile); the four quarters'
// partial sums meet after the last q in three rounds through LDS, each wave finishing two of the eight n2 blocks (its own
// + the three others' in a fixed order).  k_zip_f32 needs |u| / 128 x R >= CUs workgroups - 128 networks at |u| = 256;
// below that the two-launch forms ran at 0.67 (R = 64) of the MFMA peak.  Half the work per workgroup means the prologue,
// the three hand-over rounds and the epilogue weigh twice as much as in k_zip_f32: that kernel stays the one for R >= 128.
//
// Conditions (engine.hip, zip_match with 64-wide u blocks): as k_zip_f32, K1 a multiple of 32.
// ---------------------------------------------------------------------------
constexpr int Z6U = 64, Z6K = 32, Z6R = 8;            // u per workgroup, phase-1 tile depth, phase-2 rows per quarter and tile
constexpr int Z6STG = Z6K * (ZM + Z6U);               // 10240 floats = 40 KiB: a phase-1 tile (a phase-2 tile takes 8192)
constexpr int Z6ST = 3;

__global__ __launch_bounds__(512, 1) void k_zip64_f32(ZipArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[Z6ST * Z6STG + 16];
  double* red = reinterpret_cast<double*>(smem + Z6ST * Z6STG);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = w >> 1, ub = w & 1;            // quarter of m1 (64 values), u-block of 32
  const int l31 = lane & 31, h = lane >> 5;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3, q8 = nwg >> 3, r8 = nwg & 7;
  const int pid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const int per = a.U / Z6U;                    // workgroups per replica
  const int r = pid / per;
  const int t_ = pid - r * per;
  const int u0 = t_ * Z6U;
  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const float* __restrict__ E = (const float*)tp[a.idE];
  const float* __restrict__ X = (const float*)tp[a.idX] + u0;
  const float* __restrict__ Y = (const float*)tp[a.idY];
  float* __restrict__ C = (float*)tp[a.idC];

  const int T1 = a.K1 / Z6K;                    // phase-1 tiles per q
  constexpr int T2 = (ZM / 4) / Z6R;            // phase-2 tiles per q: 8 rows of each quarter at a time
  const int TQ = T1 + T2, TT = a.Q * TQ;

  // requests: waves 0-3 (the first two quarters' waves: one per SIMD) issue them, each a quarter of every tile
  const int iw = w & 3;
  const float* const rE0 = E + (int64_t)(8 * iw) * a.ldE;
  const float* rE = rE0;
  const float* rX = X + (int64_t)(8 * iw) * a.ldXk;
  const float* rY = Y + (int64_t)(2 * iw) * a.ldYm;
  const int offE = 4 * lane, offX = (lane >> 4) * (int)a.ldXk + 4 * (lane & 15);
  const int64_t stepE = (int64_t)Z6K * a.ldE, stepX = (int64_t)Z6K * a.ldXk, stepY = (int64_t)Z6R * a.ldYm;
  const int64_t nextX = a.ldXq - (int64_t)a.K1 * a.ldXk, nextY = a.ldYq - (int64_t)(ZM / 4) * a.ldYm;
  const int64_t partY = (int64_t)(ZM / 4) * a.ldYm;
  int rq_s = 0, rq_left = TT;
  auto request_issue = [&](int stage) {
    float* st = smem + stage * Z6STG;
    if (rq_s < T1) {             // rows 8 iw .. 8 iw + 7 of E (1 KiB each) and of Xq (four rows of 64 floats per request)
#pragma unroll
      for (int i = 0; i < 8; ++i) glds16(rE + i * a.ldE + offE, st + (8 * iw + i) * ZM);
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16(rX + 4 * i * a.ldXk + offX, st + Z6K * ZM + (8 * iw + 4 * i) * Z6U);
    } else {                     // rows 2 iw, 2 iw + 1 of every quarter of Yq
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(rY + pt * partY + i * a.ldYm + offE, st + pt * (Z6R * ZM) + (2 * iw + i) * ZM);
    }
  };
  auto request_step = [&]() {                  // (plain selects: the running pointers stay in scalar registers)
    const bool p1 = rq_s < T1;
    --rq_left;
    ++rq_s;
    const bool wrap = rq_s == TQ;
    rq_s = wrap ? 0 : rq_s;
    rE = wrap ? rE0 : rE + (p1 ? stepE : 0);
    rX += (p1 ? stepX : 0) + (wrap ? nextX : 0);
    rY += (p1 ? 0 : stepY) + (wrap ? nextY : 0);
  };

  f32x16 acc1[2], acc2[8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc1[i][e] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;

#pragma unroll
  for (int i = 0; i < Z6ST - 1; ++i) {
    if (w < 4) request_issue(i);
    request_step();
  }
  double pve = 0.0;
  if (a.partE) {
    const double* __restrict__ pr = a.partE + (size_t)r * a.strideE;
    pve = pr[min(lane, a.PE - 1)];
    if (a.PE > 64)
      for (int i = lane + 64; i < a.PE; i += 64) pve += pr[i];
  }
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
  __builtin_amdgcn_s_barrier();

  int st_cur = 0, st_nxt = 1, st_req = Z6ST - 1;
  auto middle = [&]() {                        // the barrier of a tile, in the middle of its MFMA phase
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): tile t + 1 has landed - this wave's requests, a tile old
    __builtin_amdgcn_s_barrier();
    if (w < 4 && rq_left > 0) request_issue(st_req);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto advance = [&]() {
    st_req = st_cur;
    st_cur = st_nxt;
    st_nxt = st_nxt == Z6ST - 1 ? 0 : st_nxt + 1;
  };
  float fa[2][2], fb[2], fy[2][8];

  for (int q = 0; q < a.Q; ++q) {
    // ---- phase 1: Tq[m1 quarter kh, u-block ub] = sum_k1 E[k1][m1] Xq[k1][u] ---------------------------------
    for (int s = 0; s < T1; ++s) {
      const float* cA = smem + st_cur * Z6STG + h * ZM + kh * (ZM / 4) + l31;           // E image [k1][256]
      const float* cB = smem + st_cur * Z6STG + Z6K * ZM + h * Z6U + ub * 32 + l31;     // Xq image [k1][64]
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[0][i] = cA[32 * i];
      fb[0] = cB[0];
#pragma unroll
      for (int kk = 0; kk < Z6K / 2; ++kk) {
        const int c = kk & 1, nx = c ^ 1;
        if (kk + 1 < Z6K / 2) {
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[nx][i] = cA[2 * (kk + 1) * ZM + 32 * i];
          fb[nx] = cB[2 * (kk + 1) * Z6U];
        }
        if (kk == Z6K / 4 + 1) request_step();  // the cursor moves on in the shadow of this k-step's MFMAs
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc1[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i], fb[c], acc1[i], 0, 0, 0);   // D1[m1][u]
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x004, 12, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (kk == Z6K / 4) middle();
      }
      advance();
    }
    // ---- phase 2: E'[u-block ub, :] += sum over this quarter's m1 of Tq[m1][u] Yq[m1][n2] ----------------------
#pragma unroll
    for (int ms = 0; ms < T2; ++ms) {
      // rows 8 ms .. 8 ms + 7 of the quarter = register group g = ms % 4 of accumulator block ms / 4
      const float* cY = smem + st_cur * Z6STG + kh * (Z6R * ZM) + (4 * h) * ZM + l31;    // Yq image [8 rows][256] of this quarter
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) fy[0][nb] = cY[32 * nb];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {           // k-step e = kk: row 4 h + kk of the tile
        const int c = kk & 1, nx = c ^ 1;
        if (kk + 1 < 4) {
#pragma unroll
          for (int nb = 0; nb < 8; ++nb) fy[nx][nb] = cY[(kk + 1) * ZM + 32 * nb];
        }
        const float tq = acc1[ms / 4][4 * (ms % 4) + kk];
        if (kk == 2) request_step();
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
          acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fy[c][nb], tq, acc2[nb], 0, 0, 0);       // D2^T[n2][u]
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x004, 6, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        if (kk == 1) middle();
      }
      advance();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc1[i][e] = 0.f;
  }

  // ---- the four quarters meet: wave kh finishes the n2 blocks 2 kh, 2 kh + 1; in round j = 1, 2, 3 every wave hands the
  // two blocks of quarter (kh + j) % 4 over through LDS and adds what quarter (kh - j) % 4 left for it (a fixed order)
  f32x16 mine[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) mine[i] = kh == 0 ? acc2[i] : kh == 1 ? acc2[2 + i] : kh == 2 ? acc2[4 + i] : acc2[6 + i];
#pragma unroll
  for (int j = 1; j < 4; ++j) {
    const int to = (kh + j) & 3, from = (kh - j) & 3;
    __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): this wave's own LDS reads are done
    __builtin_amdgcn_s_barrier();              // ... and everybody's: the area is free
    float4* xo = reinterpret_cast<float4*>(smem + w * 2048) + lane;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const f32x16 gv = to == 0 ? acc2[i] : to == 1 ? acc2[2 + i] : to == 2 ? acc2[4 + i] : acc2[6 + i];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) xo[(i * 4 + qd) * 64] = make_float4(gv[4 * qd], gv[4 * qd + 1], gv[4 * qd + 2], gv[4 * qd + 3]);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    const float4* xi = reinterpret_cast<const float4*>(smem + (from * 2 + ub) * 2048) + lane;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 o = xi[(i * 4 + qd) * 64];
        mine[i][4 * qd + 0] += o.x; mine[i][4 * qd + 1] += o.y; mine[i][4 * qd + 2] += o.z; mine[i][4 * qd + 3] += o.w;
      }
  }

  // ---- epilogue: lazy rescale by E's producer (X, Y are inputs), 16-byte stores, abs-sum partial ---------------
  pve = lane < a.PE ? pve : 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pve += __shfl_xor(pve, o, 64);
  const float nE = (float)pve;
  const float scE = (a.partE && nE > (float)a.min_norm) ? nE / (float)a.numelE : 1.f;
  const float iE = 1.0f / scE;
  float asum = 0.f;
  float* __restrict__ row = C + (int64_t)(u0 + 32 * ub + l31) * a.ldC + 4 * h;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int nb = 2 * kh + i;                 // the n2 block this accumulator holds
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 v;
      v.x = mine[i][4 * g + 0] * iE; v.y = mine[i][4 * g + 1] * iE; v.z = mine[i][4 * g + 2] * iE; v.w = mine[i][4 * g + 3] * iE;
      *reinterpret_cast<float4*>(row + 32 * nb + 8 * g) = v;
      asum += (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w));
    }
  }
  double part = (double)asum;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) red[w] = part;
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
  if (tid == 0) {
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) tot += red[i];
    a.partC[(size_t)r * a.partC_stride + t_] = tot;
  }
}

}  // namespace ctn
