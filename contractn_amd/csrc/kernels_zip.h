// kernels_zip.h - two consecutive GEMM steps of a "zipper" as ONE launch: the intermediate never leaves the registers.
// Part of the gfx950 contraction engine (see engine.hip for the overview).
#pragma once
#include "kernels_mfma_g.h"

namespace ctn {

// ---------------------------------------------------------------------------
// K-zip-f32.  The overlap <phi|psi> of two MPS (BASELINE configs[2]; reference einsum.py:341-391 walks it as 199
// pairwise steps) alternates two GEMMs per site,
//
//     T[m1, (q, u)] = sum_k1  E[k1, m1] * X[q, k1, u]          (256 x 1024 x 256:  E . psi_i,  q = physical leg)
//     E'[u, n2]     = sum_(m1, q)  T[m1, q, u] * Y[q, m1, n2]   (256 x 256 x 1024:  T . phi_i)
//
// and as two launches writes and re-reads the 1 MiB T per site and network (44 % of the path's HBM bytes) and pays the
// K = 256 step's tile prologue and epilogue 16 k-tiles apart.  Here ONE workgroup owns a block of 128 values of u and
// walks q: phase 1 forms Tq[m1 = 0..255, u-block] = E^T Xq in accumulators, phase 2 multiplies those accumulators -
// used directly as MFMA operands, no LDS round trip - into E'[u-block, 0..255] += Tq^T Yq, and T never exists outside
// the register file.  One prologue and one epilogue per 134 MFLOP instead of one pair per 17.
//
// How an accumulator becomes an operand: a v_mfma_f32_32x32x2_f32 result block D[i][j] leaves lane (j = lane & 31,
// h = lane >> 5) with rows i = 8 g + 4 h + e in register 4 g + e.  Phase 1 computes D1[i = m1][j = u]; register (g, e)
// of lane (u, h) is then exactly the B-side fragment "column u, k = m1" of a k-step that pairs m1 = 8 g + e (lower
// lane half) with m1 = 8 g + 4 + e (upper half) - any pairing is fine as long as the other operand follows it, and the
// Y fragment is read from LDS row 8 g + 4 h + e accordingly.  Phase 2 accumulates D2^T[i = n2][j = u] (operands
// swapped, as in k_mfma_f32_g), so a lane ends up with 4 consecutive n2 of one row u per register quad: 16-byte stores.
//
// 8 waves = 4 u-blocks of 32 x 2 halves of m1: wave (ub, kh) forms Tq[m1 in half kh (4 blocks), u-block ub] in phase 1
// (4 MFMAs per k-step: 4 E fragments + 1 X fragment) and in phase 2 sums ITS 128 values of m1 into a partial
// E'[u-block ub, all 256 n2] (8 accumulators; 8 MFMAs per k-step: 8 Y fragments, the other operand from registers).
// The two halves' partial sums meet once, after the last q, through LDS (fixed order: first half + second half).
// Operand tiles arrive by LDS-DMA in a 3-stage ring (one raw s_barrier per 16-deep tile, in the middle of its MFMA
// phase); phase-1 tile: E 16 x 256 + Xq 16 x 128, phase-2 tile: Yq 16 rows of each m1 half x 256.
//
// Conditions (engine.hip, zip_match): |m1| = |n2| = 256, |u| a multiple of 128, K1 a multiple of 16 and >= 32, every
// operand dense along its innermost index with uniform strides, X and Y network inputs, fp32.
// The intermediate's rescale (reference einsum.py:387 after the first step) is not applied: (T / s) Y = (T Y) / s, the
// register reports 0 for that step and the magnitude moves into the second step's rescale - the same product.
// ---------------------------------------------------------------------------
struct ZipArgs {
  void* const* ptrs;        // [R][n_tensors]
  int32_t n_tensors, idE, idX, idY, idC;
  int64_t ldE;              // E[k1][m1]: elements between consecutive k1 (m1 unit-stride)
  int64_t ldXq, ldXk;       // X[q][k1][u]: strides of q and k1 (u unit-stride)
  int64_t ldYq, ldYm;       // Y[q][m1][n2]: strides of q and m1 (n2 unit-stride)
  int64_t ldC;              // E'[u][n2]: stride of u (n2 unit-stride)
  int32_t K1, Q, U;
  const double* partE;      // E's producer partials (nullptr: E is a network input)
  int32_t PE, strideE;
  double numelE, min_norm;
  double* partC;            // [R][partC_stride]: one partial per workgroup (U / 128 per replica)
  int32_t partC_stride, R;
  unsigned long long* dbg;  // CTN_STAMPS builds only
};

constexpr int ZM = 256, ZU = 128, ZK = 16, ZSTG = 8192;   // stage: 8192 floats = 32 KiB
constexpr int ZST = 3;     // ring depth (4 measured: the requests are never waited for - 768 cycles per workgroup - nothing to gain)

__global__ __launch_bounds__(512, 1) void k_zip_f32(ZipArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[ZST * ZSTG + 16];
  double* red = reinterpret_cast<double*>(smem + ZST * ZSTG);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = w >> 2, ub = w & 3;
  const int l31 = lane & 31, h = lane >> 5;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3, q8 = nwg >> 3, r8 = nwg & 7;
  const int pid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const int per = a.U / ZU;                    // workgroups per replica
  const int r = pid / per;
  const int t_ = pid - r * per;
  const int u0 = t_ * ZU;
#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) a.dbg[(size_t)pid * 8 + 0] = __builtin_amdgcn_s_memtime();
#endif
  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const float* __restrict__ E = (const float*)tp[a.idE];
  const float* __restrict__ X = (const float*)tp[a.idX] + u0;
  const float* __restrict__ Y = (const float*)tp[a.idY];
  float* __restrict__ C = (float*)tp[a.idC];

  const int T1 = a.K1 / ZK;                    // phase-1 tiles per q
  constexpr int T2 = (ZM / 2) / ZK;            // phase-2 tiles per q: 16 rows of each m1 half at a time
  const int TQ = T1 + T2, TT = a.Q * TQ;

  // the LDS-DMA requests of the next tile not yet asked for.  The cursor walks the tiles in order with running pointers -
  // a handful of scalar adds per request: both waves of a SIMD come out of the tile's barrier together, and whatever
  // they do before their next MFMA is matrix-pipe idle time (~100 scalar instructions of index arithmetic here cost
  // 8 % of the kernel).  The pointers are wave-uniform - scalar registers; a lane's own offset is a 32-bit number added
  // by the instruction.
  // Only the four waves of the first m1 half issue requests - each for both halves: the two waves of a SIMD (w, w + 4)
  // do not interleave, the older one runs ahead and parks at the tile's barrier (a third of its time); an LDS-DMA
  // instruction costs its wave ~60 cycles, and with every wave issuing its share right behind the barrier both waves
  // of a SIMD paid them at the same moment with no MFMA queued.  Now the younger wave goes straight on with its MFMAs.
  const float* const rE0 = E + (int64_t)(4 * ub) * a.ldE;
  const float* rE = rE0;
  const float* rX = X + (int64_t)(4 * ub) * a.ldXk;
  const float* rY = Y + (int64_t)(4 * ub) * a.ldYm;
  const int offE = 4 * lane, offX = h * (int)a.ldXk + 4 * l31;
  const int64_t stepE = (int64_t)ZK * a.ldE, stepX = (int64_t)ZK * a.ldXk, stepY = (int64_t)ZK * a.ldYm;
  const int64_t nextX = a.ldXq - (int64_t)a.K1 * a.ldXk, nextY = a.ldYq - (int64_t)(ZM / 2) * a.ldYm;
  const int64_t halfY = (int64_t)(ZM / 2) * a.ldYm;
  int rq_s = 0, rq_left = TT;
  auto request_issue = [&](int stage) {
    float* st = smem + stage * ZSTG;
    if (rq_s < T1) {             // rows 4 ub .. 4 ub + 3 of E (1 KiB each) and of Xq (two rows per request: lanes 0-31 / 32-63)
#pragma unroll
      for (int i = 0; i < 4; ++i) glds16(rE + i * a.ldE + offE, st + (4 * ub + i) * ZM);
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16(rX + 2 * i * a.ldXk + offX, st + 4096 + (4 * ub + 2 * i) * ZU);
    } else {                     // rows 4 ub .. 4 ub + 3 of both m1 halves of Yq
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(rY + hf * halfY + i * a.ldYm + offE, st + hf * 4096 + (4 * ub + i) * ZM);
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

  f32x16 acc1[4], acc2[8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc1[i][e] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;

#pragma unroll
  for (int i = 0; i < ZST - 1; ++i) {
    if (kh == 0) request_issue(i);
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
  __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): once per 134 MFLOP - no need to count
  __builtin_amdgcn_s_barrier();
#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) a.dbg[(size_t)pid * 8 + 1] = __builtin_amdgcn_s_memtime();
#endif

  // (measured without gain, LAB_NOTES R3: s_setprio 1 for waves 4-7; those waves taking the barrier at the END of a
  // tile so that the two waves of a SIMD run half a tile apart: -0.8 %)
  constexpr int bar_at = 3;
  int st_cur = 0, st_nxt = 1, st_req = ZST - 1, t = 0;
#ifdef CTN_STAMPS
  unsigned long long wait_vm = 0, wait_bar = 0;
#endif
  auto middle = [&]() {                        // the barrier of a tile, in the middle of its MFMA phase
    __builtin_amdgcn_sched_barrier(0);
#ifdef CTN_STAMPS
    const unsigned long long s0 = __builtin_amdgcn_s_memtime();
#endif
    // vmcnt(0): tile t + 1 has landed - this wave's requests, a tile old
    __builtin_amdgcn_s_waitcnt(0x0F70);
#ifdef CTN_STAMPS
    const unsigned long long s1 = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_s_barrier();
#ifdef CTN_STAMPS
    const unsigned long long s2 = __builtin_amdgcn_s_memtime();
    wait_vm += s1 - s0;
    wait_bar += s2 - s1;
#endif
    if (kh == 0 && rq_left > 0) request_issue(st_req);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto advance = [&]() {
    st_req = st_cur;
    st_cur = st_nxt;
    st_nxt = st_nxt == ZST - 1 ? 0 : st_nxt + 1;
    ++t;
  };
  // (a tile's first fragments read at the end of the tile before - the stage has landed by then: +-0, the other wave
  // of the SIMD covers that latency already)
  float fa[2][4], fb[2], fy[2][8];

  for (int q = 0; q < a.Q; ++q) {
    // ---- phase 1: Tq[m1 half kh, u-block ub] = sum_k1 E[k1][m1] Xq[k1][u] ------------------------------------
    for (int s = 0; s < T1; ++s) {
      const float* cA = smem + st_cur * ZSTG + h * ZM + kh * (ZM / 2) + l31;       // E image [k1][256]
      const float* cB = smem + st_cur * ZSTG + 4096 + h * ZU + ub * 32 + l31;      // Xq image [k1][128]
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[0][i] = cA[32 * i];
      fb[0] = cB[0];
#pragma unroll
      for (int kk = 0; kk < ZK / 2; ++kk) {
        const int c = kk & 1, nx = c ^ 1;
        if (kk + 1 < ZK / 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[nx][i] = cA[2 * (kk + 1) * ZM + 32 * i];
          fb[nx] = cB[2 * (kk + 1) * ZU];
        }
        if (kk == bar_at + 1) request_step();   // the cursor moves on in the shadow of this k-step's MFMAs
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc1[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i], fb[c], acc1[i], 0, 0, 0);   // D1[m1][u]
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x004, 12, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        if (kk == bar_at) middle();
      }
      advance();
    }
    // ---- phase 2: E'[u-block ub, :] += sum over this half's m1 of Tq[m1][u] Yq[m1][n2] -------------------------
#pragma unroll
    for (int ms = 0; ms < T2; ++ms) {
      // rows 16 ms .. 16 ms + 15 of the half = half of accumulator block ms / 2: its register groups g = 2 (ms & 1), + 1
      const float* cY = smem + st_cur * ZSTG + kh * 4096 + (4 * h) * ZM + l31;      // Yq image [16 rows][256]
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) fy[0][nb] = cY[32 * nb];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {           // k-step (g, e) = (kk / 4, kk % 4): row 8 (kk / 4) + 4 h + e of the tile
        const int c = kk & 1, nx = c ^ 1;
        if (kk + 1 < 8) {
#pragma unroll
          for (int nb = 0; nb < 8; ++nb) fy[nx][nb] = cY[(8 * ((kk + 1) / 4) + (kk + 1) % 4) * ZM + 32 * nb];
        }
        const float tq = acc1[ms / 2][4 * (2 * (ms & 1) + kk / 4) + kk % 4];
        if (kk == bar_at + 1) request_step();
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
        if (kk == bar_at) middle();
      }
      advance();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc1[i][e] = 0.f;
  }
#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) {
    a.dbg[(size_t)pid * 8 + 2] = __builtin_amdgcn_s_memtime();
    a.dbg[(size_t)pid * 8 + 5] = wait_vm;
    a.dbg[(size_t)pid * 8 + 6] = wait_bar;
  }
  if (a.dbg && tid == 448) a.dbg[(size_t)pid * 8 + 7] = wait_vm + wait_bar;
#endif

  // ---- the two m1 halves meet: each partner hands over four of its eight n2 blocks (two rounds of two through the
  // ring's LDS) and finishes the other four: first half + second half, whichever wave adds
  f32x16 mine[4], give[4];
  if (kh == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { mine[i] = acc2[i]; give[i] = acc2[4 + i]; }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) { mine[i] = acc2[4 + i]; give[i] = acc2[i]; }
  }
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): this wave's own LDS reads are done
    __builtin_amdgcn_s_barrier();              // ... and everybody's: the area is free
    float4* xo = reinterpret_cast<float4*>(smem + w * 2048) + lane;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
        xo[(i * 4 + qd) * 64] = make_float4(give[2 * round + i][4 * qd], give[2 * round + i][4 * qd + 1],
                                            give[2 * round + i][4 * qd + 2], give[2 * round + i][4 * qd + 3]);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    const float4* xi = reinterpret_cast<const float4*>(smem + (w ^ 4) * 2048) + lane;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 o = xi[(i * 4 + qd) * 64];
        const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float m = mine[2 * round + i][4 * qd + e];
          mine[2 * round + i][4 * qd + e] = kh == 0 ? m + ov[e] : ov[e] + m;
        }
      }
  }
#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) a.dbg[(size_t)pid * 8 + 4] = __builtin_amdgcn_s_memtime();
#endif

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
  for (int i = 0; i < 4; ++i) {
    const int nb = 4 * kh + i;                 // the n2 block this accumulator holds
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
#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) a.dbg[(size_t)pid * 8 + 3] = __builtin_amdgcn_s_memtime();
#endif
}

}  // namespace ctn
