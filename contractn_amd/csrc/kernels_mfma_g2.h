// kernels_mfma_g2.h - the 256 x 128 LDS-DMA kernel with a TWO-stage ring: three workgroups per CU
// Part of the gfx950 contraction engine (see engine.hip for the overview).
//
// k_mfma_f32_g<4, 2> keeps three 24 KB stages (72 KB: two workgroups per CU, two waves per SIMD).  On
// short-K steps (K = 256: 16 k-tiles per output tile) the tile boundary - first requests, rescale /
// abs-sum / stores of 128 accumulators - is a fifth of a tile's life, and with two co-resident workgroups
// that tend to run in step the matrix pipe idles there.  This variant trades ring depth for residency:
// two stages (48 KB), at most 168 registers, THREE workgroups per CU.  Protocol per k-tile: the whole
// tile's MFMAs (hand-scheduled block, no read-ahead into the next tile), then wait for this wave's
// requests of the next tile (issued a whole tile earlier), one raw barrier (everybody has left the stage,
// the next one is published), re-request into the stage just left, read the first fragments of the next tile.
// Mode-1 operands, K % 16 == 0, K >= 32 only (everything else stays on k_mfma_f32_g); the epilogue's
// operands are loaded after the loop (no registers held across it - the other two workgroups cover the wait).
#pragma once
#include "kernels_mfma_g.h"

namespace ctn {

__global__ __launch_bounds__(256, 3) void k_mfma_f32_g2(StepArgs a) {
  constexpr int NW = 4, NJ = 2, TNB = 128, NST = 2;
  constexpr int SZA = GK * GM, SZB = GK * TNB, STG = SZA + SZB;
  constexpr int RPW = GK / NW;
  __shared__ __attribute__((aligned(16))) float smem[NST * STG + 2 * NW];
  double* red = reinterpret_cast<double*>(smem + NST * STG);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3, q8 = nwg >> 3, r8 = nwg & 7;
  const int pid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const int r = pid / a.blocks_per_replica;
  const int t = pid - r * a.blocks_per_replica;
  const int tiles_mn = a.tiles_m * a.tiles_n;
  const int b = t / tiles_mn;
  const int tt = t - b * tiles_mn;
  const int tm = tt / a.tiles_n, tn = tt % a.tiles_n;
  const int m0 = tm * GM;
  const int n0 = tn * TNB;

  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const float* __restrict__ A = (const float*)tp[a.idA] + a.obA[b];
  const float* __restrict__ B = (const float*)tp[a.idB] + a.obB[b];
  float* __restrict__ C = (float*)tp[a.idC] + a.obC[b];

  const int l31 = lane & 31, h = lane >> 5;
  const int wm = (w / 2) * 128, wn = (w % 2) * 64;

  uint32_t offA = (uint32_t)a.omA[m0 + 4 * lane] * 4u;
  uint32_t offB = (uint32_t)a.onB[n0 + 4 * l31] * 4u;
  asm volatile("" : "+v"(offA), "+v"(offB));   // consumed before the first LDS-DMA (see k_mfma_f32_g)
  const char* const Ac = reinterpret_cast<const char*>(A);
  const char* const Bc = reinterpret_cast<const char*>(B);
  const_i32_ptr okA = (const_i32_ptr)(a.okA + RPW * w);
  const_i32_ptr okB = (const_i32_ptr)(a.okB + RPW * w);
  const int nkt = a.K / GK;

  int ka[RPW], kb[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) { ka[i] = okA[i]; kb[i] = okB[i]; }

  auto request = [&](int kt_next, int stage) {
    float* sa = smem + stage * STG + (RPW * w) * GM;
    float* sb = smem + stage * STG + SZA + (RPW * w) * TNB;
    uint32_t oA = offA;
    asm volatile("" : "+v"(oA));
#pragma unroll
    for (int i = 0; i < RPW; ++i)
      glds16(reinterpret_cast<const float*>(Ac + (int64_t)ka[i] * 4 + oA), sa + i * GM);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int lo = min(kb[2 * p], kb[2 * p + 1]);
      const uint32_t d0 = (uint32_t)(kb[2 * p] - lo) * 4u, d1 = (uint32_t)(kb[2 * p + 1] - lo) * 4u;
      glds16(reinterpret_cast<const float*>(Bc + (int64_t)lo * 4 + (offB + (h ? d1 : d0))), sb + 2 * p * TNB);
    }
    const int k0 = kt_next * GK;  // the tables are padded by 64 entries past K
#pragma unroll
    for (int i = 0; i < RPW; ++i) { ka[i] = okA[k0 + i]; kb[i] = okB[k0 + i]; }
  };

  f32x16 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  request(1, 0);
  request(2, 1);                                   // nkt >= 2 is guaranteed by the launcher
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0x0F76);              // vmcnt(6): k-tile 0 (this wave's share) has landed
  __builtin_amdgcn_s_barrier();

  const int fa0 = h * GM + wm + l31;
  const int fb0 = h * TNB + wn + l31;
  float fa[2][4], fb[2][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[1][i] = 0.f;
  fb[1][0] = fb[1][1] = 0.f;
#define CTN_G2_OPERANDS                                                                                         \
    [a00] "+v"(acc[0][0]), [a01] "+v"(acc[0][1]), [a10] "+v"(acc[1][0]), [a11] "+v"(acc[1][1]),                \
    [a20] "+v"(acc[2][0]), [a21] "+v"(acc[2][1]), [a30] "+v"(acc[3][0]), [a31] "+v"(acc[3][1]),                \
    [fa00] "+v"(fa[0][0]), [fa01] "+v"(fa[0][1]), [fa02] "+v"(fa[0][2]), [fa03] "+v"(fa[0][3]),                \
    [fa10] "+v"(fa[1][0]), [fa11] "+v"(fa[1][1]), [fa12] "+v"(fa[1][2]), [fa13] "+v"(fa[1][3]),                \
    [fb00] "+v"(fb[0][0]), [fb01] "+v"(fb[0][1]), [fb10] "+v"(fb[1][0]), [fb11] "+v"(fb[1][1])
  for (int kt = 0; kt < nkt; ++kt) {
    const int st = kt & 1;
    const float* cA = smem + st * STG + fa0;
    const float* cB = smem + st * STG + SZA + fb0;
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[0][i] = cA[32 * i];
#pragma unroll
    for (int j = 0; j < NJ; ++j) fb[0][j] = cB[32 * j];
    const unsigned vA = lds_addr(cA), vB = lds_addr(cB);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile(CTN_G_ASM_WHOLE_TILE_N128 : CTN_G2_OPERANDS : [vA] "v"(vA), [vB] "v"(vB) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xF70) /* vmcnt(0) */;  // k-tile kt+1: this wave's requests, a tile old
    __builtin_amdgcn_s_barrier();                       // everybody has left stage st; k-tile kt+1 is published
    if (kt + 2 < nkt) request(kt + 3, st);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef CTN_G2_OPERANDS

  // epilogue (as k_mfma_f32_g; its operands are fetched only now)
  double pva = 0.0, pvb = 0.0;
  if (a.partA) pva = lane < a.PA ? a.partA[(size_t)r * kMaxPartials + lane] : 0.0;
  if (a.partB) pvb = lane < a.PB ? a.partB[(size_t)r * kMaxPartials + lane] : 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { pva += __shfl_xor(pva, o, 64); pvb += __shfl_xor(pvb, o, 64); }
  const float nA = (float)pva, nB = (float)pvb;  // exactly producer_scale<float>()
  const float scA = (a.partA && nA > (float)a.min_norm) ? nA / (float)a.numelA : 1.f;
  const float scB = (a.partB && nB > (float)a.min_norm) ? nB / (float)a.numelB : 1.f;
  const float iA = 1.0f / scA, iB = 1.0f / scB;
  float asum = 0.f;
  const bool full = (m0 + GM <= a.M) && (n0 + TNB <= a.N);
  auto store_tile = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* __restrict__ row = C + a.omC[m0 + wm + 32 * i + l31];
      const bool rin = FULL || (m0 + wm + 32 * i + l31 < a.M);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v;
          v.x = (acc[i][j][4 * g + 0] * iA) * iB;
          v.y = (acc[i][j][4 * g + 1] * iA) * iB;
          v.z = (acc[i][j][4 * g + 2] * iA) * iB;
          v.w = (acc[i][j][4 * g + 3] * iA) * iB;
          if (FULL || (rin && n0 + wn + 32 * j + 8 * g + 4 * h < a.N)) {
            *reinterpret_cast<float4*>(row + a.onC[n0 + wn + 32 * j + 8 * g + 4 * h]) = v;
            asum += (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w));
          }
        }
    }
  };
  if (full) store_tile(std::true_type{});
  else store_tile(std::false_type{});
  const double tot = block_sum((double)asum, red);
  if (tid == 0) {  // this tile covers up to 2 of the planner's 128 x 128 partial slots
    const int tm128 = (a.M + 127) / 128, tn128 = (a.N + 127) / 128;
    double* pc = a.partC + (size_t)r * a.partC_stride + (size_t)b * tm128 * tn128;
#pragma unroll
    for (int dm = 0; dm < 2; ++dm) {
      const int sm = 2 * tm + dm;
      if (sm < tm128 && tn < tn128) pc[sm * tn128 + tn] = dm == 0 ? tot : 0.0;
    }
  }
}

}  // namespace ctn
