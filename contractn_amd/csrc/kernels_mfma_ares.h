// kernels_mfma_ares.h - fp32 MFMA GEMM step with a small LEFT operand (256 x 256) against a very wide right one: the left
// operand lives in registers for the whole workgroup, which walks several column tiles while only the right operand
// streams.  Part of the gfx950 contraction engine (see engine.hip for the overview).
#pragma once
#include "kernels_mfma_g.h"

namespace ctn {

// ---------------------------------------------------------------------------
// K-mfma-f32-ares ("A resident").  The boundary absorptions of a 2D grid at bond 16 (8 x 8 PEPS, D = 16: steps of
// 256 x 2^20 x 256 per slice, a fifth of the contraction; reference einsum.py:371-387 runs each as tensordot +
// transpose + stabilize) are GEMMs whose column tiles all share ONE 256 x 256 left operand, at 64 flop per byte of
// B + C.  On the large-tile kernel (k_mfma_f32_g<4,2>) every 256 x 128 tile is a workgroup of its own: prologue, 16
// k-tiles of A and B through LDS, 128 KB of stores - 0.66-0.72 of the MFMA peak where the same kernel reaches 0.89 on
// long-K steps.
//
// Here a workgroup (8 waves) keeps A in REGISTERS - wave w holds rows 32 w .. 32 w + 31 as the 128 A-side fragments of
// its MFMAs (v_mfma_f32_32x32x2_f32) - and walks `ntw` consecutive column tiles of 128: only B streams, through a
// 3-stage LDS-DMA ring of 32-deep k-tiles (all eight waves read the same B fragments), the ring running on across
// tile boundaries; a tile ends with its stores, nothing else.  Half the LDS-DMA traffic of the large-tile kernel (no
// A), no per-tile prologue, one abs-sum partial per workgroup.
//
// What the stores cost (256 x 2^22 x 256, LAB_NOTES R4.9): 138 TFLOP/s with the stores left out; 112 with 16-byte
// stores straight from transposed accumulators (a wave's store touches 32 rows, 32 bytes of each); 125 with the
// accumulators the right way up and dword stores, lanes along n (two full 128-byte lines per store) - this form.  Not
// a burst problem: a variant that spreads a finished 32-column block's stores over the MFMAs of the next one (whole K
// of a 32-column sub-tile per LDS slot, one accumulator block per wave) reached 108, and 130 without its stores.
//
// The k-steps run in the permuted order k(kk, h) = 8 (kk / 4) + 4 h + kk % 4 (lane half h), so that a lane's four
// consecutive k-steps are four consecutive k: a k-contiguous A is loaded with 16-byte loads (AV = 1; with 4-byte loads a
// wave's request touches 64 cache lines and the 256 KB of A cost a workgroup ~8 us), a k-contiguous B (MB = 2, image
// [k / 4][128][4]) is read from LDS eight bytes at a time.  MB = 1: B unit-stride along n (image [k][128]).
//
// The LDS reads of the B fragments are inline asm, a UNIT (two k-steps, eight MFMAs) ahead of their MFMAs: left to
// itself the compiler, short of registers, reads each pair of values right before its two MFMAs and the wave waits out
// the LDS latency 32 times per k-tile.  vmcnt counts loads, stores and LDS-DMA together in issue order: the barrier of
// a tile's first k-tile waits with vmcnt(63) - the k-tile it needs was requested BEFORE the previous tile's 64 stores
// per lane, of which all but the first stay in flight.
//
// Conditions (engine.hip, ares_match): fp32, M = K = 256 (any number of batch entries), N a multiple of 128, B in mode 1 or 2, C
// with every 128-column tile dense, more workgroup partials than slots (the collapse path: N >= 32768).
// ---------------------------------------------------------------------------
constexpr int AR_M = 256, AR_K = 256, AR_TN = 128, AR_KT = 32, AR_ST = 3;
constexpr int AR_STG = AR_KT * AR_TN;        // 4096 floats = 16 KiB per stage

typedef float ar_f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) float* ar_gout;   // global memory, said so: global_store, not flat_store

template <int OFF>
__device__ __forceinline__ float ar_lds32(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ ar_f32x2 ar_lds64(unsigned addr) {
  ar_f32x2 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// unit U of a k-tile (k-steps 2 U, 2 U + 1) for column block J; addr = the stage + the lane's own offset.
// MB = 1 (image [k][128], lane offset (4 h * 128 + l31) * 4): rows 8 (U / 2) + 2 (U & 1) (+ 4 h) and the next one;
// MB = 2 (image [k / 4][128][4], lane offset (h * 128 + l31) * 16): chunk 2 (U / 2) (+ h), elements 2 (U & 1) and
// the next one - one 8-byte read.
template <int MB, int U, int J>
__device__ __forceinline__ void ar_read_one(float (&b)[4][2], unsigned addr) {
  if constexpr (MB == 2) {
    const ar_f32x2 v = ar_lds64<(U / 2) * 4096 + J * 512 + (U & 1) * 8>(addr);
    b[J][0] = v[0]; b[J][1] = v[1];
  } else {
    b[J][0] = ar_lds32<(8 * (U / 2) + 2 * (U & 1)) * 512 + J * 128>(addr);
    b[J][1] = ar_lds32<(8 * (U / 2) + 2 * (U & 1) + 1) * 512 + J * 128>(addr);
  }
}
template <int MB, int U>
__device__ __forceinline__ void ar_read_unit(float (&b)[4][2], unsigned addr) {
  ar_read_one<MB, U, 0>(b, addr); ar_read_one<MB, U, 1>(b, addr);
  ar_read_one<MB, U, 2>(b, addr); ar_read_one<MB, U, 3>(b, addr);
}

template <int MB, int AV>
__global__ __launch_bounds__(512, 1) void k_mfma_f32_ares(StepArgs a, int ntw) {
  __shared__ __attribute__((aligned(16))) float smem[AR_ST * AR_STG];
  __shared__ int s_okA[AR_K], s_okB[AR_K], s_omC[AR_M];
  __shared__ double red[8];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3, q8 = nwg >> 3, r8 = nwg & 7;
  const int pid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const int per = (a.N / AR_TN) / ntw;          // workgroups per batch entry
  const int perR = per * a.Bt;                  // ... per replica
  const int r = pid / perR;
  const int rem = pid - r * perR;               // (the workgroup's partial-sum slot)
  const int bt = rem / per;
  const int t_ = rem - bt * per;
  const int n_first = t_ * ntw * AR_TN;         // first column of this workgroup

  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const float* __restrict__ A = (const float*)tp[a.idA] + a.obA[bt];
  const char* const Bc = reinterpret_cast<const char*>((const float*)tp[a.idB] + a.obB[bt]);
  float* __restrict__ C = (float*)tp[a.idC] + a.obC[bt];

  for (int k = tid; k < AR_K; k += 512) { s_okA[k] = a.okA[k]; s_okB[k] = a.okB[k]; s_omC[k] = a.omC[k]; }
  double pva = 0.0, pvb = 0.0;
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
  const int offAm = a.omA[32 * w + l31];
  __syncthreads();

  // A, once: lane (m = 32 w + l31, h) holds A[m][k(kk, h)] for the 128 k-steps
  float fa[AR_K / 2];
  if (AV) {                                     // AV: A's k is one unit-stride run (engine.hip, ares_avec)
    const float* __restrict__ Arow = A + offAm + s_okA[0] + 4 * h;
#pragma unroll
    for (int c = 0; c < AR_K / 8; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(Arow + 8 * c);
      fa[4 * c + 0] = v.x; fa[4 * c + 1] = v.y; fa[4 * c + 2] = v.z; fa[4 * c + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int kk = 0; kk < AR_K / 2; ++kk) {
      fa[kk] = A[offAm + s_okA[8 * (kk / 4) + 4 * h + kk % 4]];
      if (kk % 16 == 15) __builtin_amdgcn_sched_barrier(0);   // (all 128 addresses at once do not fit beside the 128 values)
    }
  }

  // B requests: waves 0-3 (one per SIMD) bring a k-tile each time, a quarter of it each.  The column offsets of a tile's
  // 16-byte pieces come from the table (a tile's columns need not be dense in B), looked up a tile ahead.
  const int total = ntw * (AR_K / AR_KT);       // k-tiles this workgroup walks
  int rq = 0;                                   // next k-tile to request (all waves count; waves 0-3 issue)
  int offB_cur, offB_nxt = 0, offB2_cur = 0, offB2_nxt = 0;
  auto lookup = [&](int tile, int& o1, int& o2) {
    const int n0 = n_first + min(tile, ntw - 1) * AR_TN;
    if (MB == 2) { o1 = a.onB[n0 + lane]; o2 = a.onB[n0 + 64 + lane]; }
    else { o1 = a.onB[n0 + 4 * l31]; o2 = 0; }
  };
  lookup(0, offB_cur, offB2_cur);
  lookup(1, offB_nxt, offB2_nxt);
  auto request = [&](int stage) {               // k-tile `rq` of the walk -> ring stage
    const int kt = rq % (AR_K / AR_KT);
    float* st = smem + stage * AR_STG;
    if (MB == 2) {                              // k chunks 2 w, 2 w + 1 of the k-tile: 64 columns each per request
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int kc = 2 * w + c;               // chunk inside the k-tile (8 chunks of 4)
        const int64_t kb = s_okB[kt * AR_KT + 4 * kc];
        glds16(reinterpret_cast<const float*>(Bc + (kb + offB_cur) * 4), st + (kc * AR_TN) * 4);
        glds16(reinterpret_cast<const float*>(Bc + (kb + offB2_cur) * 4), st + (kc * AR_TN + 64) * 4);
      }
    } else {                                    // rows 8 w .. 8 w + 7: two rows per request (lanes 0-31 / 32-63)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int row = 8 * w + 2 * p + h;
        const int64_t kb = s_okB[kt * AR_KT + row];
        glds16(reinterpret_cast<const float*>(Bc + (kb + offB_cur) * 4), st + (8 * w + 2 * p) * AR_TN);
      }
    }
  };
  auto request_step = [&]() {                   // after a request: the next k-tile; a new tile takes the looked-up offsets
    ++rq;
    if (rq % (AR_K / AR_KT) == 0) {
      offB_cur = offB_nxt; offB2_cur = offB2_nxt;
      lookup(rq / (AR_K / AR_KT) + 1, offB_nxt, offB2_nxt);
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

#pragma unroll
  for (int i = 0; i < AR_ST - 1; ++i) {
    if (w < 4) request(i);
    request_step();
  }
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): once per workgroup
  __builtin_amdgcn_s_barrier();

  pva = lane < a.PA ? pva : 0.0;
  pvb = lane < a.PB ? pvb : 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { pva += __shfl_xor(pva, o, 64); pvb += __shfl_xor(pvb, o, 64); }
  const float nA = (float)pva, nB = (float)pvb;
  const float scA = (a.partA && nA > (float)a.min_norm) ? nA / (float)a.numelA : 1.f;
  const float scB = (a.partB && nB > (float)a.min_norm) ? nB / (float)a.numelB : 1.f;
  const float iA = 1.0f / scA, iB = 1.0f / scB;

  int st_cur = 0, st_nxt = 1, st_req = AR_ST - 1;
  float asum = 0.f;
  float b[2][4][2];                             // B fragments: two units (two k-steps each) of the four column blocks
  const unsigned lane_off = MB == 2 ? (unsigned)((h * AR_TN + l31) * 16) : (unsigned)((4 * h * AR_TN + l31) * 4);
  ar_read_unit<MB, 0>(b[0], lds_addr(smem) + lane_off);
#define AR_UNIT(U, BUF, SRC)                                                                                         \
  {                                                                                                                  \
    __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0): unit U is in b[BUF] */                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kt * 16 + 2 * (U)], b[BUF][0][0], acc[0], 0, 0, 0);             \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    ar_read_unit<MB, ((U) + 1) & 7>(b[(BUF) ^ 1], SRC);                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    _Pragma("unroll") for (int j = 1; j < 4; ++j)                                                                    \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kt * 16 + 2 * (U)], b[BUF][j][0], acc[j], 0, 0, 0);           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                    \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kt * 16 + 2 * (U) + 1], b[BUF][j][1], acc[j], 0, 0, 0);       \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  }
  int offCn_nxt = a.onC[n_first];
  for (int tile = 0; tile < ntw; ++tile) {
    const int offCn = offCn_nxt;
#pragma unroll
    for (int kt = 0; kt < AR_K / AR_KT; ++kt) {
      const unsigned vB = lds_addr(smem + st_cur * AR_STG) + lane_off, vBn = lds_addr(smem + st_nxt * AR_STG) + lane_off;
      AR_UNIT(0, 0, vB) AR_UNIT(1, 1, vB) AR_UNIT(2, 0, vB) AR_UNIT(3, 1, vB)
      // the barrier of a k-tile, in the middle of its MFMA phase: the next k-tile has landed - this wave's requests, a
      // k-tile old; behind them only the last tile's stores
      if (kt == 0 && tile > 0) __builtin_amdgcn_s_waitcnt(0xCF7F);   // vmcnt(63)
      else __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0)
      __builtin_amdgcn_s_barrier();
      if (w < 4 && rq < total) request(st_req);
      request_step();
      // the next tile's offset in C, fetched here: a load between a tile's stores and the vmcnt(63) would be counted there
      if (kt == 1) offCn_nxt = a.onC[n_first + min(tile + 1, ntw - 1) * AR_TN];
      __builtin_amdgcn_sched_barrier(0);
      AR_UNIT(4, 0, vB) AR_UNIT(5, 1, vB) AR_UNIT(6, 0, vB) AR_UNIT(7, 1, vBn)   // (the last one reads ahead into the next k-tile)
      st_req = st_cur;
      st_cur = st_nxt;
      st_nxt = st_nxt == AR_ST - 1 ? 0 : st_nxt + 1;
    }
    // a column tile is complete: lazy rescale, dword stores (lane (l31, h) holds rows 8 g + 4 h + e of column l31 of
    // each 32 x 32 block: a store instruction writes two full lines), abs-sum, zero
    ar_gout colp = (ar_gout)(C + offCn + l31);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int om[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) om[e] = s_omC[32 * w + 8 * g + 4 * h + e];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = (acc[j][4 * g + e] * iA) * iB;
          colp[om[e] + 32 * j] = v[e];
        }
        asum += (fabsf(v[0]) + fabsf(v[1])) + (fabsf(v[2]) + fabsf(v[3]));
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  }
#undef AR_UNIT
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
    a.partC[(size_t)r * a.partC_stride + rem] = tot;
  }
}

}  // namespace ctn
