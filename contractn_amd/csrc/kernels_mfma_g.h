// kernels_mfma_g.h - fp32 MFMA GEMM, large-tile variant fed by direct-to-LDS loads (LDS-DMA)
// Part of the gfx950 contraction engine (see engine.hip for the overview).
#pragma once
#include <type_traits>

#include "kernels_mfma.h"
#include "kernels_mfma_g_asm.inc"

namespace ctn {

// ---------------------------------------------------------------------------
// K-mfma-f32-g: 256 x 128 workgroup tile, 4 waves (2x2), each wave 128 x 64 = 4 x 2
// v_mfma_f32_32x32x2_f32 accumulators (6 LDS fragment reads per 8 MFMAs instead of 4 per 4), BK = 16.
// Operand tiles go global -> LDS with global_load_lds_dwordx4 (no staging registers, no ds_write
// pass) into a 3-stage ring, up to two k-tiles ahead of the MFMAs; the only synchronisation per
// k-tile is one s_waitcnt vmcnt + one raw s_barrier in the MIDDLE of its MFMA phase (a
// __syncthreads() would drain the ring).
//
// Eligibility (decided by the planner, plan.cpp): both operands "mode 1" (unit stride along their
// free index, so 16 bytes per lane are 4 consecutive rows/columns and a wave instruction fills one
// lane-linear k-row of the LDS image), K >= 32, C vector-storable.  LDS-DMA cannot mask: padded tables
// keep every load in bounds, ragged M / N only produce rows and columns that the epilogue drops, and
// the rows of a ragged last k-tile are zeroed when they are read into fragments.  Everything else stays on k_mfma_f32.
//
// The MFMA is issued with the operands swapped (B fragment as SrcA), i.e. it accumulates C^T
// blocks: a lane then holds 4 CONSECUTIVE columns of one row of C in 4 consecutive accumulator
// registers, so the epilogue stores 16 bytes per lane straight from the accumulators - no LDS
// staging, LDS budget 72 KiB = two workgroups per CU.
// ---------------------------------------------------------------------------
constexpr int GM = 256, GN = 128, GK = 16, GST = 3;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
// k-offset tables are read through the constant address space: with a wave-uniform index the
// compiler then uses scalar loads (lgkmcnt), which keeps them out of the vmcnt queue the ring counts on
typedef const __attribute__((address_space(4))) int32_t* const_i32_ptr;

// LDS byte address of a __shared__ element (DS instruction operand)
__device__ __forceinline__ unsigned lds_addr(const float* p) { return (unsigned)(size_t)(lds_void_t*)p; }

// one LDS-DMA wave instruction: lane l copies 16 bytes from its own global address to lds + 16 l
__device__ __forceinline__ void glds16(const float* g, float* lds) {
  __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)lds, 16, 0, 0);
}

// <NW waves, NJ 32-wide column blocks per wave>:
//   <4, 2>: 256 x 128 tile, wave 128 x 64, 2 workgroups per CU
//   <8, 2>: 256 x 256 tile, 8 waves as 2 x 4 of 128 x 64, 1 workgroup per CU
// ASM: the k-steps run as the hand-scheduled blocks of kernels_mfma_g_asm.inc (<4, 2> and K % 16 == 0 only)
// MA / MB = 2: the operand is unit-stride along k instead ("mode 2", e.g. a row-major A): its 16-byte requests
// run along k (one lane = one row, 4 consecutive k), its LDS image is [k / 4][rows][4], and both operands
// are read in the permuted k order described in tools/gen_mfma_g_asm.py.  <4, 2, asm> only.
template <int NW, int NJ, bool ASM = false, int MA = 1, int MB = 1>
__global__ __launch_bounds__(NW * 64, (NW == 4 && NJ == 2) ? 2 : 1) void k_mfma_f32_g(StepArgs a) {
  static_assert(!ASM || NJ == 2, "asm blocks exist for the 128 x 64 wave tile only");
  static_assert((MA == 1 && MB == 1) || (NW == 4 && NJ == 2), "k-contiguous operands: the 256 x 128 tile only");
  constexpr bool PERM = MA == 2 || MB == 2;         // permuted k order (see the generator)
  constexpr int WNC = NW / 2;                       // waves along N (2 along M)
  constexpr int TNB = WNC * 32 * NJ;                // tile columns
  constexpr int NREQ = GK / NW + (TNB == 128 ? 2 : GK / NW);  // LDS-DMA instructions per wave and k-tile
  constexpr int SZA = GK * GM, SZB = GK * TNB, STG = SZA + SZB;
  constexpr int RPW = GK / NW;                      // k-rows of each operand tile loaded by one wave
  // ONE LDS object: [stage 0 A|B][stage 1 A|B][stage 2 A|B][red NW doubles]
  __shared__ __attribute__((aligned(16))) float smem[GST * STG + 2 * NW];
  double* red = reinterpret_cast<double*>(smem + GST * STG);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3, q8 = nwg >> 3, r8 = nwg & 7;
  const int pid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const int r = pid / a.blocks_per_replica;       // here: 256 x TNB tiles per replica (x K splits)
  int t = pid - r * a.blocks_per_replica;
  const int tiles_mn = a.tiles_m * a.tiles_n;
  // K split over workgroups (a.ks_S > 0; the launcher takes it when the step's tiles alone cannot fill the chip and K is
  // long): this workgroup multiplies k in [kbeg, kbeg + Kloc) and leaves its un-scaled tile in slab `sp`, shaped like C;
  // k_splitk_reduce adds the slabs in a fixed order, rescales, stores and writes the abs-sum partials
  const bool split = a.ks_S > 0;
  int sp = 0;
  if (split) { sp = t / (a.Bt * tiles_mn); t -= sp * (a.Bt * tiles_mn); }
  const int kbeg = sp * a.ks_chunk;
  const int Kloc = split ? min(a.K - kbeg, a.ks_chunk) : a.K;
  const int b = t / tiles_mn;
  const int tt = t - b * tiles_mn;
  const int tm = tt / a.tiles_n, tn = tt % a.tiles_n;
  const int m0 = tm * GM;
  const int n0 = tn * TNB;

#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) a.dbg[(size_t)pid * 8 + 0] = __builtin_amdgcn_s_memtime();
#endif
  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const float* __restrict__ A = (const float*)tp[a.idA] + a.obA[b];
  const float* __restrict__ B = (const float*)tp[a.idB] + a.obB[b];
  float* __restrict__ C = (split ? (float*)a.ks_slab + ((size_t)r * a.ks_S + sp) * a.ks_numelC : (float*)tp[a.idC]) + a.obC[b];

  const int l31 = lane & 31, h = lane >> 5;
  const int wm = (w / WNC) * 128, wn = (w % WNC) * (32 * NJ);


  // loader: wave w fills k-rows RPW*w .. RPW*w + RPW-1 of every stage.  A row = 64 lanes x 4 rows of
  // the tile; B row (256 wide) likewise, or (128 wide) lanes 0-31 -> k-row, lanes 32-63 -> the next
  // addresses are kept as (wave-uniform 64-bit base) + (per-lane unsigned 32-bit byte offset): the
  // scalar-base form of the load, so no vector-ALU address arithmetic per request (the planner only
  // sends operands of at most 2^30 elements here, so byte offsets fit)
  uint32_t offA = (uint32_t)a.omA[m0 + 4 * lane] * 4u;
  uint32_t offB = (uint32_t)a.onB[n0 + 4 * (TNB == 128 ? l31 : lane)] * 4u;
  // both table entries are "used" here, before the first LDS-DMA: a wait for an ordinary load that the
  // compiler places after a request can only be vmcnt(0) and would also wait for the request to land
  asm volatile("" : "+v"(offA), "+v"(offB));
  uint32_t offA2[MA == 2 ? 4 : 1] = {0}, offB2[MB == 2 ? 2 : 1] = {0};   // k-contiguous operand: lane = row 64 q + lane
  if constexpr (MA == 2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) offA2[q] = (uint32_t)a.omA[m0 + 64 * q + lane] * 4u;
    asm volatile("" : "+v"(offA2[0]), "+v"(offA2[1]), "+v"(offA2[2]), "+v"(offA2[3]));
  }
  if constexpr (MB == 2) {
#pragma unroll
    for (int q = 0; q < 2; ++q) offB2[q] = (uint32_t)a.onB[n0 + 64 * q + lane] * 4u;
    asm volatile("" : "+v"(offB2[0]), "+v"(offB2[1]));
  }
  const char* const Ac = reinterpret_cast<const char*>(A);
  const char* const Bc = reinterpret_cast<const char*>(B);
  // k-offset table entries: scalar loads through the constant address space (measured: fetching them
  // with wave-uniform vector loads next to the LDS-DMA requests costs 2.5 % on the headline)
  const_i32_ptr okA = (const_i32_ptr)(a.okA + kbeg + RPW * w);
  const_i32_ptr okB = (const_i32_ptr)(a.okB + kbeg + RPW * w);
  const int nkt = (Kloc + GK - 1) / GK;
  // ragged K: LDS-DMA cannot mask, so the last k-tile's rows beyond K hold in-bounds garbage (padded
  // k-tables); they are zeroed when read into fragments - both operands, so nothing can turn into NaN
  const bool ktail = (Kloc % GK) != 0;
  const int krem = Kloc - (nkt - 1) * GK;

  int ka[RPW], kb[RPW];  // k-offset table entries of the next k-tile to request (wave-uniform)
#pragma unroll
  for (int i = 0; i < RPW; ++i) { ka[i] = okA[i]; kb[i] = okB[i]; }

  auto request = [&](int kt_next, int stage) {  // issue this wave's LDS-DMA loads of one k-tile, then look up the next offsets
    float* sa = smem + stage * STG + (RPW * w) * GM;
    float* sb = smem + stage * STG + SZA + (RPW * w) * TNB;
    uint32_t oA = offA;
    asm volatile("" : "+v"(oA));  // opaque: keeps (uniform base + k offset) + lane offset from being re-associated
    if constexpr (MA == 2) {      // wave w brings k-chunk w (k = 4w .. 4w+3) of all 256 rows: 4 requests of 64 rows
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t o = offA2[q];
        asm volatile("" : "+v"(o));
        glds16(reinterpret_cast<const float*>(Ac + (int64_t)ka[0] * 4 + o), smem + stage * STG + (w * GM + 64 * q) * 4);
      }
    } else {
#pragma unroll
      for (int i = 0; i < RPW; ++i)
        glds16(reinterpret_cast<const float*>(Ac + (int64_t)ka[i] * 4 + oA), sa + i * GM);
    }
    if constexpr (MB == 2) {      // k-chunk w of all 128 columns: 2 requests of 64 columns
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        uint32_t o = offB2[q];
        asm volatile("" : "+v"(o));
        glds16(reinterpret_cast<const float*>(Bc + (int64_t)kb[0] * 4 + o), smem + stage * STG + SZA + (w * TNB + 64 * q) * 4);
      }
    } else if constexpr (TNB == 128) {
      // lanes 0-31 fetch k-row 2p, lanes 32-63 k-row 2p+1: the smaller of the two table entries is the
      // scalar base, the (non-negative) distance to the other one goes into that half's lane offset
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int lo = min(kb[2 * p], kb[2 * p + 1]);
        const uint32_t d0 = (uint32_t)(kb[2 * p] - lo) * 4u, d1 = (uint32_t)(kb[2 * p + 1] - lo) * 4u;
        glds16(reinterpret_cast<const float*>(Bc + (int64_t)lo * 4 + (offB + (h ? d1 : d0))), sb + 2 * p * TNB);
      }
    } else {
#pragma unroll
      for (int i = 0; i < RPW; ++i)
        glds16(reinterpret_cast<const float*>(Bc + (int64_t)kb[i] * 4 + offB), sb + i * TNB);
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
  // Everything the EPILOGUE needs is requested only now, behind the first two k-tiles, so that nothing
  // but the operand addressing stands between the start of the workgroup and its first LDS-DMA: the
  // producers' abs-sum partials (reduced to the rescale factors after the main loop; index clamped,
  // lanes >= P masked there) and the C offset tables - 12 + (0..2) vector loads.
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
  int offm[4], offn[NJ][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) offm[i] = a.omC[m0 + wm + 32 * i + l31];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) offn[j][g] = a.onC[n0 + wn + 32 * j + 8 * g + 4 * h];
  __builtin_amdgcn_sched_barrier(0);
  // k-tile 0 has landed (this wave's share): everything but the youngest request group and the (at
  // least 4 + 4 NJ) epilogue loads behind it.  simm16 = vmcnt[3:0] | 7 << 4 | 15 << 8 | vmcnt[5:4] << 14
  static_assert(NREQ + 4 + 4 * NJ == 18 || NREQ + 4 + 4 * NJ == 16 || NREQ + 4 + 4 * NJ == 28, "add the vmcnt immediate");
  if (NREQ + 4 + 4 * NJ == 18) __builtin_amdgcn_s_waitcnt(0x4F72);       // vmcnt(18)
  else if (NREQ + 4 + 4 * NJ == 16) __builtin_amdgcn_s_waitcnt(0x4F70);  // vmcnt(16)
  else __builtin_amdgcn_s_waitcnt(0x4F7C);                               // vmcnt(28)
  __builtin_amdgcn_s_barrier();                    // ... and everybody else's
#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) a.dbg[(size_t)pid * 8 + 1] = __builtin_amdgcn_s_memtime();
#endif

  // Main loop.  The barrier of a k-tile sits in the MIDDLE of its MFMA phase, between k-steps whose
  // fragments are already in registers: by then this wave's requests for the next k-tile (issued a
  // whole tile earlier) have landed, the barrier publishes them, and the requests for the k-tile
  // after that follow it (every wave has left the previous k-tile, so its ring stage is free).  The
  // fragment read-ahead therefore runs straight across the k-tile boundary and a wave never waits
  // on LDS or on its peers with an empty matrix pipe.
  // this lane's fragment base inside an operand tile, in floats (lane-half h folded in), and the distance
  // between 32-row fragment blocks: layout [k][rows] (k = 2 kk + h, or 8 (kk / 4) + 4 h + kk % 4 when
  // permuted), resp. [k / 4][rows][4] for a k-contiguous operand
  const int fa0 = MA == 2 ? (h * GM + wm + l31) * 4 : ((PERM ? 4 * h : h) * GM + wm + l31);
  const int fb0 = MB == 2 ? (h * TNB + wn + l31) * 4 : ((PERM ? 4 * h : h) * TNB + wn + l31);
  constexpr int blkA = MA == 2 ? 128 : 32, blkB = MB == 2 ? 128 : 32;
  int st_cur = 0, st_nxt = 1, st_req = 2;
  float fa[2][4], fb[2][NJ];
  {
    const float* cA = smem + fa0;
    const float* cB = smem + SZA + fb0;
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[0][i] = cA[blkA * i];
#pragma unroll
    for (int j = 0; j < NJ; ++j) fb[0][j] = cB[blkB * j];
  }
  // <4, 2> with whole k-tiles: the k-steps run as hand-scheduled blocks (tools/gen_mfma_g_asm.py):
  // single ds_read_b32 with immediate offsets (no address arithmetic on the vector ALU), the reads
  // of step s+1 behind the first MFMA of step s, one wait per step, at its end.  The barrier / LDS-DMA
  // section between the two halves of a k-tile stays in C++.
  if constexpr (ASM) {
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[1][i] = 0.f;
    fb[1][0] = fb[1][1] = 0.f;
#define CTN_G_ASM_OPERANDS                                                                                      \
    [a00] "+v"(acc[0][0]), [a01] "+v"(acc[0][1]), [a10] "+v"(acc[1][0]), [a11] "+v"(acc[1][1]),                \
    [a20] "+v"(acc[2][0]), [a21] "+v"(acc[2][1]), [a30] "+v"(acc[3][0]), [a31] "+v"(acc[3][1]),                \
    [fa00] "+v"(fa[0][0]), [fa01] "+v"(fa[0][1]), [fa02] "+v"(fa[0][2]), [fa03] "+v"(fa[0][3]),                \
    [fa10] "+v"(fa[1][0]), [fa11] "+v"(fa[1][1]), [fa12] "+v"(fa[1][2]), [fa13] "+v"(fa[1][3]),                \
    [fb00] "+v"(fb[0][0]), [fb01] "+v"(fb[0][1]), [fb10] "+v"(fb[1][0]), [fb11] "+v"(fb[1][1])
    for (int kt = 0; kt < nkt; ++kt) {
      const unsigned vA = lds_addr(smem + st_cur * STG + fa0), vB = lds_addr(smem + st_cur * STG + SZA + fb0);
      const unsigned vAn = lds_addr(smem + st_nxt * STG + fa0), vBn = lds_addr(smem + st_nxt * STG + SZA + fb0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (MA == 2 && MB == 1)
        asm volatile(CTN_G_ASM_FIRST_HALF_N128_A2B1 : CTN_G_ASM_OPERANDS : [vA] "v"(vA), [vB] "v"(vB) : "memory");
      else if constexpr (MA == 1 && MB == 2)
        asm volatile(CTN_G_ASM_FIRST_HALF_N128_A1B2 : CTN_G_ASM_OPERANDS : [vA] "v"(vA), [vB] "v"(vB) : "memory");
      else if constexpr (MA == 2 && MB == 2)
        asm volatile(CTN_G_ASM_FIRST_HALF_N128_A2B2 : CTN_G_ASM_OPERANDS : [vA] "v"(vA), [vB] "v"(vB) : "memory");
      else if constexpr (TNB == 128)
        asm volatile(CTN_G_ASM_FIRST_HALF_N128 : CTN_G_ASM_OPERANDS : [vA] "v"(vA), [vB] "v"(vB) : "memory");
      else
        asm volatile(CTN_G_ASM_FIRST_HALF_N256 : CTN_G_ASM_OPERANDS : [vA] "v"(vA), [vB] "v"(vB) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xF70) /* vmcnt(0) */;  // k-tile kt+1: this wave's requests, a tile old
      __builtin_amdgcn_s_barrier();
      if (kt + 2 < nkt) request(kt + 3, st_req);
      __builtin_amdgcn_sched_barrier(0);
      // (the last k-tile also reads "next-tile" fragments: in-bounds LDS, never used - one code path,
      // so the 128 accumulator registers stay pinned through the loop)
      if constexpr (MA == 2 && MB == 1)
        asm volatile(CTN_G_ASM_SECOND_HALF_NEXT_N128_A2B1 : CTN_G_ASM_OPERANDS
                     : [vA] "v"(vA), [vB] "v"(vB), [vAn] "v"(vAn), [vBn] "v"(vBn) : "memory");
      else if constexpr (MA == 1 && MB == 2)
        asm volatile(CTN_G_ASM_SECOND_HALF_NEXT_N128_A1B2 : CTN_G_ASM_OPERANDS
                     : [vA] "v"(vA), [vB] "v"(vB), [vAn] "v"(vAn), [vBn] "v"(vBn) : "memory");
      else if constexpr (MA == 2 && MB == 2)
        asm volatile(CTN_G_ASM_SECOND_HALF_NEXT_N128_A2B2 : CTN_G_ASM_OPERANDS
                     : [vA] "v"(vA), [vB] "v"(vB), [vAn] "v"(vAn), [vBn] "v"(vBn) : "memory");
      else if constexpr (TNB == 128)
        asm volatile(CTN_G_ASM_SECOND_HALF_NEXT_N128 : CTN_G_ASM_OPERANDS
                     : [vA] "v"(vA), [vB] "v"(vB), [vAn] "v"(vAn), [vBn] "v"(vBn) : "memory");
      else
        asm volatile(CTN_G_ASM_SECOND_HALF_NEXT_N256 : CTN_G_ASM_OPERANDS
                     : [vA] "v"(vA), [vB] "v"(vB), [vAn] "v"(vAn), [vBn] "v"(vBn) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      st_cur = st_nxt;
      st_nxt = st_req;
      st_req = st_req == GST - 1 ? 0 : st_req + 1;
    }
#undef CTN_G_ASM_OPERANDS
  } else
  for (int kt = 0; kt < nkt; ++kt) {
    const float* cA = smem + st_cur * STG + fa0;
    const float* cB = smem + st_cur * STG + SZA + fb0;
    const float* nA = smem + st_nxt * STG + fa0;
    const float* nB = smem + st_nxt * STG + SZA + fb0;
    const bool tail_cur = ktail && kt == nkt - 1, tail_nxt = ktail && kt + 2 == nkt;
#pragma unroll
    for (int kk = 0; kk < GK / 2; ++kk) {
      const int c = kk & 1, nx = c ^ 1;
      // fragment offsets of k-step kk + 1 off the lane's base and the k it covers, per layout (as in the generator)
      constexpr auto frag = [](int step, int rows, int mode) {
        const int j = step / 4, e = step % 4;
        return mode == 2 ? (2 * j * rows) * 4 + e : ((MA == 2 || MB == 2) ? 8 * j + e : 2 * step) * rows;
      };
      const int knext = PERM ? 8 * ((kk + 1) / 4) + 4 * h + (kk + 1) % 4 : 2 * (kk + 1) + h;
      if (kk + 1 < GK / 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[nx][i] = cA[frag(kk + 1, GM, MA) + blkA * i];
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[nx][j] = cB[frag(kk + 1, TNB, MB) + blkB * j];
        if (tail_cur && knext >= krem) {
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[nx][i] = 0.f;
#pragma unroll
          for (int j = 0; j < NJ; ++j) fb[nx][j] = 0.f;
        }
      } else if (kt + 1 < nkt) {  // first k-step of the next k-tile (published by this tile's barrier)
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[nx][i] = nA[blkA * i];
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[nx][j] = nB[blkB * j];
        if (tail_nxt && (PERM ? 4 * h : h) >= krem) {
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[nx][i] = 0.f;
#pragma unroll
          for (int j = 0; j < NJ; ++j) fb[nx][j] = 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c][j], fa[c][i], acc[i][j], 0, 0, 0);
      // issue order: the first MFMA of the step (its operand wait then only covers reads that are a
      // whole k-step old), the fragment reads of the next step (pairs merge into ds_read2_b32), the rest
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, (4 + NJ) / 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * NJ - 1, 0);
      if (kk == GK / 4 - 1) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xF70) /* vmcnt(0) */;  // k-tile kt+1: this wave's requests, a tile old
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nkt) request(kt + 3, st_req);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    st_cur = st_nxt;
    st_nxt = st_req;
    st_req = st_req == GST - 1 ? 0 : st_req + 1;
  }

#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) a.dbg[(size_t)pid * 8 + 2] = __builtin_amdgcn_s_memtime();
#endif
  // epilogue: lazy rescale, 16-byte stores straight from the accumulators, abs-sum partial.
  // Ragged tiles (M % 256, N % 128) mask whole 16-byte vectors: N % 4 == 0 whenever C is vector-storable.
  pva = lane < a.PA ? pva : 0.0;
  pvb = lane < a.PB ? pvb : 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { pva += __shfl_xor(pva, o, 64); pvb += __shfl_xor(pvb, o, 64); }
  const float nA = (float)pva, nB = (float)pvb;  // exactly producer_scale<float>()
  const float scA = (a.partA && nA > (float)a.min_norm) ? nA / (float)a.numelA : 1.f;
  const float scB = (a.partB && nB > (float)a.min_norm) ? nB / (float)a.numelB : 1.f;
  const float iA = split ? 1.0f : 1.0f / scA, iB = split ? 1.0f : 1.0f / scB;   // (x * 1 == x: a slab holds the raw sums)
  float asum = 0.f;
  const bool full = (m0 + GM <= a.M) && (n0 + TNB <= a.N);
  auto store_tile = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* __restrict__ row = C + offm[i];
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
            *reinterpret_cast<float4*>(row + offn[j][g]) = v;
            asum += (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w));
          }
        }
    }
  };
  if (full) store_tile(std::true_type{});
  else store_tile(std::false_type{});
#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) {
    a.dbg[(size_t)pid * 8 + 4] = __builtin_amdgcn_s_memtime();
    a.dbg[(size_t)pid * 8 + 5] = a.dbg[(size_t)pid * 8 + 4];
    a.dbg[(size_t)pid * 8 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                 ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
  }
  if (a.dbg && tid == 192) a.dbg[(size_t)pid * 8 + 6] = __builtin_amdgcn_s_memtime();
#endif
  const double tot = block_sum((double)asum, red);
#ifdef CTN_STAMPS
  if (a.dbg && tid == 0) a.dbg[(size_t)pid * 8 + 3] = __builtin_amdgcn_s_memtime();
#endif
  if (tid == 0 && !split) {  // this tile covers up to 2 x (TNB / 128) of the planner's 128 x 128 partial slots
    const int tm128 = (a.M + 127) / 128, tn128 = (a.N + 127) / 128;
    double* pc = a.partC + (size_t)r * a.partC_stride + (size_t)b * tm128 * tn128;
#pragma unroll
    for (int dm = 0; dm < 2; ++dm)
#pragma unroll
      for (int dn = 0; dn < TNB / 128; ++dn) {
        const int sm = 2 * tm + dm, sn = (TNB / 128) * tn + dn;
        if (sm < tm128 && sn < tn128) pc[sm * tn128 + sn] = (dm == 0 && dn == 0) ? tot : 0.0;
      }
  }
}

}  // namespace ctn
