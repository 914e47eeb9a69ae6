// kernels_mfma_g.h - fp32 MFMA GEMM, large-tile variant fed by direct-to-LDS loads (LDS-DMA)
// Part of the gfx950 contraction engine (see engine.hip for the overview).
#pragma once
#include "kernels_mfma.h"

namespace ctn {

// ---------------------------------------------------------------------------
// K-mfma-f32-g: 256 x 128 workgroup tile, 4 waves (2x2), each wave 128 x 64 = 4 x 2
// v_mfma_f32_32x32x2_f32 accumulators (6 LDS fragment reads per 8 MFMAs instead of 4 per 4), BK = 16.
// Operand tiles go global -> LDS with global_load_lds_dwordx4 (no staging registers, no ds_write
// pass) into a 3-stage ring, two k-tiles ahead of the MFMAs; the only synchronisation per k-tile
// is one counted s_waitcnt vmcnt + one raw s_barrier (a __syncthreads() would drain the ring).
//
// Eligibility (checked by the launcher): both operands "mode 1" (unit stride along their free
// index, so 16 bytes per lane are 4 consecutive rows/columns and a wave instruction fills one
// lane-linear k-row of the LDS image), M % 256 == 0, N % 128 == 0, K % 16 == 0, K >= 32, C
// vector-storable.  Everything else stays on k_mfma_f32.
//
// The MFMA is issued with the operands swapped (B fragment as SrcA), i.e. it accumulates C^T
// blocks: a lane then holds 4 CONSECUTIVE columns of one row of C in 4 consecutive accumulator
// registers, so the epilogue stores 16 bytes per lane straight from the accumulators - no LDS
// staging, which keeps the ring alive (persistent variant) and the LDS budget at 72 KiB.
// ---------------------------------------------------------------------------
constexpr int GM = 256, GN = 128, GK = 16, GST = 3;
constexpr int G_SZA = GK * GM, G_SZB = GK * GN, G_STG = G_SZA + G_SZB;  // floats per ring stage (24 KiB)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
// k-offset tables are read through the constant address space: with a wave-uniform index the
// compiler then uses scalar loads (lgkmcnt), which keeps them out of the vmcnt queue the ring counts on
typedef const __attribute__((address_space(4))) int32_t* const_i32_ptr;

// one LDS-DMA wave instruction: lane l copies 16 bytes from its own global address to lds + 16 l
__device__ __forceinline__ void glds16(const float* g, float* lds) {
  __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)lds, 16, 0, 0);
}

__global__ __launch_bounds__(256, 2) void k_mfma_f32_g(StepArgs a) {
  // ONE LDS object: [stage 0 A|B][stage 1 A|B][stage 2 A|B][red 4 doubles]
  __shared__ __attribute__((aligned(16))) float smem[GST * G_STG + 8];
  double* red = reinterpret_cast<double*>(smem + GST * G_STG);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3, q8 = nwg >> 3, r8 = nwg & 7;
  const int pid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const int r = pid / a.blocks_per_replica;       // here: 256x128 tiles per replica
  const int t = pid - r * a.blocks_per_replica;
  const int tiles_mn = a.tiles_m * a.tiles_n;
  const int b = t / tiles_mn;
  const int tt = t - b * tiles_mn;
  const int m0 = (tt / a.tiles_n) * GM;
  const int n0 = (tt % a.tiles_n) * GN;

  const float scA = producer_scale<float>(a.partA, a.PA, a.numelA, a.min_norm, r);
  const float scB = producer_scale<float>(a.partB, a.PB, a.numelB, a.min_norm, r);

  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const float* __restrict__ A = (const float*)tp[a.idA] + a.obA[b];
  const float* __restrict__ B = (const float*)tp[a.idB] + a.obB[b];
  float* __restrict__ C = (float*)tp[a.idC] + a.obC[b];

  const int l31 = lane & 31, h = lane >> 5;
  const int wm = (w >> 1) * 128, wn = (w & 1) * 64;

  // epilogue addressing, fetched before any LDS-DMA is in flight
  int offm[4], offn[2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) offm[i] = a.omC[m0 + wm + 32 * i + l31];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) offn[j][g] = a.onC[n0 + wn + 32 * j + 8 * g + 4 * h];

  // loader: wave w fills k-rows 4w .. 4w+3 of every stage; A row = 64 lanes x 4 rows of the tile,
  // B row pair = lanes 0-31 -> k-row, lanes 32-63 -> the next one
  const float* __restrict__ Ab = A + a.omA[m0 + 4 * lane];
  const float* __restrict__ Bb = B + a.onB[n0 + 4 * l31];
  const_i32_ptr okA = (const_i32_ptr)(a.okA + 4 * w);
  const_i32_ptr okB = (const_i32_ptr)(a.okB + 4 * w);
  const int nkt = a.K / GK;

  int ka[4], kb[4];  // k-offset table entries of the next k-tile to request (wave-uniform)
#pragma unroll
  for (int i = 0; i < 4; ++i) { ka[i] = okA[i]; kb[i] = okB[i]; }

  auto request = [&](int kt_next, int stage) {  // issue the 6 LDS-DMA loads of one k-tile, then look up the next offsets
    float* sa = smem + stage * G_STG + (4 * w) * GM;
    float* sb = smem + stage * G_STG + G_SZA + (4 * w) * GN;
    glds16(Ab + ka[0], sa);
    glds16(Ab + ka[1], sa + GM);
    glds16(Ab + ka[2], sa + 2 * GM);
    glds16(Ab + ka[3], sa + 3 * GM);
    glds16(Bb + (h ? kb[1] : kb[0]), sb);
    glds16(Bb + (h ? kb[3] : kb[2]), sb + 2 * GN);
    const int k0 = kt_next * GK;  // the tables are padded by 64 entries past K
#pragma unroll
    for (int i = 0; i < 4; ++i) { ka[i] = okA[k0 + i]; kb[i] = okB[k0 + i]; }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  request(1, 0);
  request(2, 1);                                   // nkt >= 2 is guaranteed by the launcher
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // k-tile 0 has landed (this wave's share)
  __builtin_amdgcn_s_barrier();                    // ... and everybody else's

  const int fa0 = h * GM + wm + l31;
  const int fb0 = h * GN + wn + l31;
  int st_cur = 0, st_req = 2;
  for (int kt = 0; kt < nkt; ++kt) {
    const bool ahead = kt + 2 < nkt;
    if (ahead) request(kt + 3, st_req);
    __builtin_amdgcn_sched_barrier(0);
    const float* cA = smem + st_cur * G_STG + fa0;
    const float* cB = smem + st_cur * G_STG + G_SZA + fb0;
    float fa[2][4], fb[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[0][i] = cA[32 * i];
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[0][j] = cB[32 * j];
#pragma unroll
    for (int kk = 0; kk < GK / 2; ++kk) {
      const int c = kk & 1, nx = c ^ 1;
      if (kk + 1 < GK / 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[nx][i] = cA[(kk + 1) * 2 * GM + 32 * i];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[nx][j] = cB[(kk + 1) * 2 * GN + 32 * j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c][j], fa[c][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // retire k-tile kt+1 (all but the 6 youngest = k-tile kt+2's requests), then meet the others
    if (ahead) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    st_cur = st_cur == GST - 1 ? 0 : st_cur + 1;
    st_req = st_req == GST - 1 ? 0 : st_req + 1;
  }

  // epilogue: lazy rescale, 16-byte stores straight from the accumulators, abs-sum partial
  const float iA = 1.0f / scA, iB = 1.0f / scB;
  float asum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float* __restrict__ row = C + offm[i];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v;
        v.x = (acc[i][j][4 * g + 0] * iA) * iB;
        v.y = (acc[i][j][4 * g + 1] * iA) * iB;
        v.z = (acc[i][j][4 * g + 2] * iA) * iB;
        v.w = (acc[i][j][4 * g + 3] * iA) * iB;
        *reinterpret_cast<float4*>(row + offn[j][g]) = v;
        asum += (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w));
      }
  }
  const double tot = block_sum((double)asum, red);
  if (tid == 0) {  // this tile covers two of the planner's 128-row partial slots
    a.partC[(size_t)r * a.partC_stride + 2 * t] = tot;
    a.partC[(size_t)r * a.partC_stride + 2 * t + 1] = 0.0;
  }
}

}  // namespace ctn
