// kernels_zipl.h - the two GEMM steps of a zipper site as ONE launch when only a FEW networks are in flight: the pair is
// cut over both the free index u AND the contracted index m1, partial results leave as slabs that the next pair adds up
// while it loads them.  Part of the gfx950 contraction engine (see engine.hip for the overview).
#pragma once
#include <type_traits>

#include "kernels_zip.h"

namespace ctn {

// ---------------------------------------------------------------------------
// K-zip-lat.  Same pair as k_zip_f32 (kernels_zip.h; reference einsum.py:341-391, two iterations of the loop),
//
//     T[m1, (q, u)] = sum_k1  E[k1, m1] * X[q, k1, u]          (256 x 1024 x 256)
//     E'[u, n2]     = sum_(m1, q)  T[m1, q, u] * Y[q, m1, n2]   (256 x 256 x 1024)
//
// for the regime `tn.contract()` itself lives in: ONE network (or a few) in flight.  There the chain is 196 dependent
// launches of 9-10 us each - k_mfma_lat, 16 x 16 / 32 x 32 tiles with K split over the waves of a workgroup - and what a
// launch costs is its dependent memory round trips, not its 134 MFLOP.  k_zip_f32 cannot help: it needs 128 networks to
// fill the chip, because all of m1 has to meet inside one workgroup.
//
// Here a workgroup owns 16 values of u AND a PART of m1 (MP = 32 of the 256): phase 1 forms T[m1-part, q, u-block] (no
// work is repeated: every element of T is formed exactly once), phase 2 multiplies it into a PARTIAL E'[u-block, all n2] -
// the sum over its own m1 only - and stores that into slab number `mp`.  The 256 / MP = 8 slabs of a result are never
// added up by a launch of their own: the NEXT pair's workgroups add them while they load their piece of E (8 x 16-byte
// loads instead of one - a few hundred KB per workgroup out of L2 / MALL), in slab order, so the sums are fixed-order and
// bit-reproducible.  One launch per SITE instead of two per site, 128 workgroups per network (U = 256).
//
// Inside the workgroup (8 waves) the pair runs on v_mfma_f32_16x16x4_f32 without any k-loop synchronisation.  Phase 1:
// a wave owns one q and one PART of k1 and forms all 16 x 16 blocks (m1 in [16 hh, 16 hh + 16)) of T for that q over its
// part (E from LDS, where the slab sums were left; X straight from global memory into the operand registers, requested up
// front - no X element is loaded by two waves); the parts of k1 become available one after the other, and a part's waves
// start while the later rows of E are still arriving.  D[i = m1][j = u] leaves lane (u, g) with m1 = 4 g + e in register
// e: the blocks go to LDS as [part][block][u][m1], a few KB in all, and are added in the order of the parts when read.
// Phase 2 cuts the OTHER way: wave w owns the columns n2 in [32 w, 32 w + 32) and sums over every block - its result is
// complete, nothing is added across waves.  The B operand of a k-step pairs lane group g with m1 = 4 g + e (one 16-byte
// LDS read per block), and the A operand follows that pairing: Y straight from global memory as 8-byte loads, lane (i, g)
// takes Y[q][m1 = 4 g + e][32 w + 2 i, + 1], row i of the two tiles t = 0, 1, tile t being the columns 32 w + 2 i + t.
// (The first version kept the register-resident accumulator of phase 1 as phase 2's operand, k_zip_f32's way; every wave
// then held a partial 16 x 256 block and the eight of them met through 128 KB of LDS: 3.5 of the workgroup's 12 us.)
//
// Stabilisation.  stabilize() (reference einsum.py:89-107) needs sum |E'| of the SUM of the slabs, which no workgroup of
// this launch knows.  It does know sum |slab piece|: the launch writes those as its abs-sum partials, so that the
// "rescale factor" every consumer derives from them (producer_scale, k_scales) is the triangle bound
// sum_s sum |slab_s| / numel >= sum |E'| / numel - a positive number of the right magnitude (within the number of slabs),
// which is all the lazy rescale needs: any divisor keeps (T_hat, c) the same product as long as the register takes its
// log, and the last tensor is normalised exactly by k_finalize.  Per-step factors therefore differ from the reference's
// own (include/ctn_abi.h, step_rescales); their product does not.  The last pair of a run has no successor to add its
// slabs: k_zip_slab_sum does (one small launch), and leaves the TRUE abs-sum partials for whatever step follows.  In the
// eager rescale mode every pair is followed by k_zip_slab_sum and k_renorm and reads a plain, normalised E.
//
// Conditions (engine.hip, zip_match with 16-wide u blocks): |m1| = |n2| = K1 = 256, |u| a multiple of 16, Q = 4 or 2
// (MP = 32 or 64: (MP / 16) x Q = 8 waves), operands dense along their innermost index, X and Y network inputs, fp32.
// ---------------------------------------------------------------------------
struct ZipLatArgs {
  void* const* ptrs;        // [R][n_tensors]
  int32_t n_tensors, idE, idX, idY;
  const float* slabs_in;    // nullptr: E is tensor idE (rows ldE apart); else [R][S][K1][ZM] slabs of the pair before
  float* slabs_out;         // [R][S][U][ZM]
  int64_t ldE;
  int64_t ldXq, ldXk;       // X[q][k1][u]
  int64_t ldYq, ldYm;       // Y[q][m1][n2]
  int32_t U, R;
  const double* partE;      // E's producer partials (nullptr: a network input, or eager mode)
  int32_t PE, strideE;
  double numelE, min_norm;
  double* partC;            // [R][partC_stride]: one partial per workgroup, (U / 16) * S per replica
  int32_t partC_stride;
  unsigned long long* dbg;  // CTN_STAMPS builds only
};

#ifdef CTN_STAMPS
#define ZL_STAMP(k) do { if (a.dbg && tid == 0) a.dbg[(size_t)pid * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ZL_STAMP(k) do { } while (0)
#endif

typedef float zl_f4 __attribute__((ext_vector_type(4)));
typedef float zl_f2 __attribute__((ext_vector_type(2)));

// a barrier that waits for this wave's LDS traffic only: requests to global memory stay in flight across it
__device__ __forceinline__ void zl_lds_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// (A form for two co-resident workgroups per CU - at most 128 registers, the slabs of a part of k1 requested only when the
// part before has been written, Y behind phase 1 - was measured for 3 ... 16 networks, LAB_NOTES R4.2: 116 bytes of scratch and
// late requests made it 2.3 x SLOWER than one workgroup per CU with everything in flight.)
template <int Q, int MP>
__global__ __launch_bounds__(512, 1) void k_zip_lat(ZipLatArgs a) {
  static_assert(8 % Q == 0 && ZM % MP == 0 && MP % 16 == 0, "a wave owns one q and one part of k1");
  constexpr int S = ZM / MP;                   // parts of m1 = slabs of a result
  constexpr int NBQ = MP / 16;                 // 16 x 16 blocks of T per q: every wave of that q forms all of them ...
  constexpr int NKP = 8 / Q;                   // ... over ITS part of k1: the waves of a q split the contracted index
  constexpr int NBLK = NBQ * Q;                // blocks of T per workgroup: (hh, q), hh slowest
  constexpr int LDE = MP + 16;                 // LDS row of the E piece: the four k rows of a fragment read hit different banks
  constexpr int LDT = 20;                      // LDS row of a T block [u][m1 = 0..15]: 16-byte accesses of 16 rows, no bank twice
  constexpr int KS = ZM / 4;                   // k-steps of phase 1 (K1 = ZM)
  constexpr int KSW = KS / NKP;                // ... of one wave
  constexpr int NPOS = ZM * (MP / 4) / 512;    // 16-byte pieces of the E piece per thread; piece p = rows [p, p + 1) * ZM / NPOS
  constexpr int PP = NPOS / NKP;               // pieces per part of k1
  constexpr int NY0 = NBLK <= 8 ? NBLK : NBLK / 2;   // blocks whose Y fragments are requested before phase 1 (the rest behind it)
  static_assert(NPOS % NKP == 0 && KS % NKP == 0, "parts of k1 are whole pieces");
  __shared__ __attribute__((aligned(16))) float sE[ZM * LDE];
  __shared__ __attribute__((aligned(16))) float sT[NKP * NBLK * 16 * LDT];
  __shared__ double red[8];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = w % Q, kp = w / Q;
  const int i16 = lane & 15, g = lane >> 4;
  // every XCD a contiguous range of work items; items are ordered (m1 part, replica, u block), so the workgroups of an
  // XCD share their E piece and their rows of Y through its L2
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3, q8 = nwg >> 3, r8 = nwg & 7;
  const int pid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const int nub = a.U / 16, per_mp = a.R * nub;
  const int mp = pid / per_mp;
  const int rem = pid - mp * per_mp;
  const int r = rem / nub;
  const int ub = rem - r * nub;
  const int u0 = 16 * ub;
  ZL_STAMP(0);

  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const bool from_slabs = a.slabs_in != nullptr;

  // ---- everything this workgroup reads is requested here, in the order it is needed -----------------------------
  // (1) the E piece [k1 = 0..255][m1 part], 16 bytes at a time, every slab of it; piece by piece, so that the rows of
  // the first part of k1 are complete first.  (The slabs' address is a kernel argument: these requests do not wait for
  // the pointer table.)
  zl_f4 sv[NPOS][S];
#define ZL_EREQUEST(P0, NP)                                                                                              \
  do {                                                                                                                    \
    if (from_slabs) {                                                                                                     \
      const float* __restrict__ Es_ = a.slabs_in + (size_t)r * S * ZM * ZM + mp * MP;                                     \
      _Pragma("unroll") for (int p = 0; p < (NP); ++p) {                                                                  \
        const int f = tid + 512 * ((P0) + p), row = f / (MP / 4), c4 = f % (MP / 4);                                      \
        _Pragma("unroll") for (int s_ = 0; s_ < S; ++s_)                                                                  \
          sv[p][s_] = *reinterpret_cast<const zl_f4*>(Es_ + (size_t)s_ * ZM * ZM + (int64_t)row * ZM + 4 * c4);           \
      }                                                                                                                   \
    } else {                                                                                                              \
      const float* __restrict__ Es_ = (const float*)tp[a.idE] + mp * MP;                                                  \
      _Pragma("unroll") for (int p = 0; p < (NP); ++p) {                                                                  \
        const int f = tid + 512 * ((P0) + p), row = f / (MP / 4), c4 = f % (MP / 4);                                      \
        sv[p][0] = *reinterpret_cast<const zl_f4*>(Es_ + (int64_t)row * a.ldE + 4 * c4);                                  \
      }                                                                                                                   \
    }                                                                                                                     \
  } while (0)
  ZL_EREQUEST(0, NPOS);                            // (the slabs of pieces P0 .. P0 + NP - 1 into sv[0 .. NP))
  const float* __restrict__ X = (const float*)tp[a.idX] + u0;
  const float* __restrict__ Y = (const float*)tp[a.idY] + (int64_t)(mp * MP) * a.ldYm + 32 * w + 2 * i16;
  // (2) this wave's X fragments: lane (u = i16, g) holds X[q][k1 = 4 s + g][u0 + u] of its k-steps s
  float xb[KSW];
  {
    const float* __restrict__ px = X + (int64_t)q * a.ldXq + (int64_t)(4 * kp * KSW + g) * a.ldXk + i16;
#pragma unroll
    for (int s = 0; s < KSW; ++s) xb[s] = px[(int64_t)(4 * s) * a.ldXk];
  }
  // (3) E's producer partials (the lazy rescale of the epilogue)
  double pve = 0.0;
  if (a.partE) {
    const double* __restrict__ pr = a.partE + (size_t)r * a.strideE;
    pve = pr[min(lane, a.PE - 1)];
    if (a.PE > 64)
      for (int i = lane + 64; i < a.PE; i += 64) pve += pr[i];
  }
  // (4) the Y fragments of phase 2, where this wave owns the columns n2 in [32 w, 32 w + 32) of EVERY block (hh, q'):
  // lane (i, g) takes Y[q'][m1 = 16 hh + 4 g + e][32 w + 2 i, + 1] - row i of the two tiles t = 0, 1, tile t being the
  // columns 32 w + 2 i + t.  Requested behind the first barrier of phase 1 (the second half of them behind phase 1).
  zl_f2 yv[NBLK][4];
  auto yrequest = [&](int blk) {
    const int hh = blk / Q, qq = blk % Q;
    const float* __restrict__ py = Y + (int64_t)qq * a.ldYq + (int64_t)(16 * hh + 4 * g) * a.ldYm;
#pragma unroll
    for (int e = 0; e < 4; ++e) yv[blk][e] = *reinterpret_cast<const zl_f2*>(py + (int64_t)e * a.ldYm);
  };
  ZL_STAMP(1);

  // ---- phase 1: T_block[m1 = 16 hh + i][u] += sum over this wave's k1 of E[k1][m1] Xq[k1][u], all NBQ blocks of its q.
  // The parts of k1 become available one after the other (slab sums -> LDS -> barrier); the waves of part kp start as soon
  // as THEIR rows are there, while the later rows are still on their way.
  zl_f4 acc1[NBQ];
#pragma unroll
  for (int b = 0; b < NBQ; ++b) acc1[b] = zl_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int part = 0; part < NKP; ++part) {
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
      const int p = part * PP + pp;
      const int f = tid + 512 * p, row = f / (MP / 4), c4 = f % (MP / 4);
      zl_f4 ev = sv[p][0];
      if (from_slabs) {
#pragma unroll
        for (int s = 1; s < S; ++s) ev += sv[p][s];              // slab order: fixed
      }
      *reinterpret_cast<zl_f4*>(sE + row * LDE + 4 * c4) = ev;
    }
    zl_lds_barrier();
    if (part == 0) {
      // the requests for Y go out HERE, behind the first part's barrier: the workgroup's loads are bound by the rate at
      // which the CU's memory pipeline takes requests (448 KB at ~60 B / clock), not by their latency - issued up front they
      // held back the first MFMA by 2 k cycles; now they are taken in the shadow of phase 1
#pragma unroll
      for (int blk = 0; blk < NY0; ++blk) yrequest(blk);
    }
    if (kp == part) {                              // (wave-uniform)
      const float* cA = sE + (4 * kp * KSW + g) * LDE + i16;
#pragma unroll
      for (int s = 0; s < KSW; ++s)
#pragma unroll
        for (int b = 0; b < NBQ; ++b)
          acc1[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(cA[4 * s * LDE + 16 * b], xb[s], acc1[b], 0, 0, 0);
    }
  }
  ZL_STAMP(2);
  // D[i = m1][j = u] leaves lane (u, g) with m1 = 4 g + e in register e: one 16-byte store per block into [kp][blk][u][m1]
#pragma unroll
  for (int b = 0; b < NBQ; ++b) {
    const int blk = b * Q + q;
    *reinterpret_cast<zl_f4*>(sT + ((kp * NBLK + blk) * 16 + i16) * LDT + 4 * g) = acc1[b];
  }
#pragma unroll
  for (int blk = NY0; blk < NBLK; ++blk) yrequest(blk);
  ZL_STAMP(3);
  zl_lds_barrier();
  ZL_STAMP(4);

  // ---- phase 2: E'^T[n2][u] = sum over the blocks of Y[q'][m1][n2] T[m1][q'][u] for this wave's 32 columns: the B operand
  // of a k-step pairs lane group g with m1 = 4 g + e - 16-byte LDS reads, the parts of k1 added in their order - and Y
  // was requested to match
  zl_f4 acc2[2] = {zl_f4{0.f, 0.f, 0.f, 0.f}, zl_f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) {
    zl_f4 tv = *reinterpret_cast<const zl_f4*>(sT + (blk * 16 + i16) * LDT + 4 * g);
#pragma unroll
    for (int k2 = 1; k2 < NKP; ++k2) tv += *reinterpret_cast<const zl_f4*>(sT + ((k2 * NBLK + blk) * 16 + i16) * LDT + 4 * g);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[blk][e][t], tv[e], acc2[t], 0, 0, 0);
  }
  ZL_STAMP(5);

  // ---- epilogue: lane (u, g') holds n2 = 32 w + 8 g' + 2 e' + t in acc2[t][e']; lazy rescale, 16-byte stores, abs-sum
  pve = lane < a.PE ? pve : 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pve += __shfl_xor(pve, o, 64);
  const float nE = (float)pve;
  const float scE = (a.partE && nE > (float)a.min_norm) ? nE / (float)a.numelE : 1.f;
  const float iE = 1.0f / scE;
  float* __restrict__ out = a.slabs_out + ((size_t)(r * S + mp) * a.U + u0 + i16) * ZM + 32 * w + 8 * g;
  zl_f4 v0 = zl_f4{acc2[0][0], acc2[1][0], acc2[0][1], acc2[1][1]} * iE;
  zl_f4 v1 = zl_f4{acc2[0][2], acc2[1][2], acc2[0][3], acc2[1][3]} * iE;
  *reinterpret_cast<zl_f4*>(out) = v0;
  *reinterpret_cast<zl_f4*>(out + 4) = v1;
  const float asum = ((fabsf(v0[0]) + fabsf(v0[1])) + (fabsf(v0[2]) + fabsf(v0[3]))) +
                     ((fabsf(v1[0]) + fabsf(v1[1])) + (fabsf(v1[2]) + fabsf(v1[3])));
  double part = (double)asum;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) red[w] = part;
  zl_lds_barrier();                              // (not __syncthreads: that would wait for the stores above to be acknowledged)
  if (tid == 0) {
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) tot += red[i];
    a.partC[(size_t)r * a.partC_stride + mp * nub + ub] = tot;
  }
  ZL_STAMP(6);
#undef ZL_EREQUEST
}

// The slabs of the LAST pair of a run (or of every pair, in the eager rescale mode) added up into the step's ordinary
// result tensor C[u][n2] (rows ldC apart), with its true abs-sum partials: `slots` workgroups per replica, each a range
// of rows; slab order, fixed.
__global__ __launch_bounds__(256) void k_zip_slab_sum(const float* __restrict__ slabs, int S, int U, void* const* ptrs,
                                                      int n_tensors, int idC, int64_t ldC, double* __restrict__ partC,
                                                      int partC_stride) {
  __shared__ double red[4];
  const int r = blockIdx.y, j = blockIdx.x, slots = gridDim.x;
  const int rows = (U + slots - 1) / slots;
  const int u_lo = j * rows, u_hi = min(U, u_lo + rows);
  float* __restrict__ C = (float*)ptrs[(size_t)r * n_tensors + idC];
  const float* __restrict__ sl = slabs + (size_t)r * S * U * ZM;
  float asum = 0.f;
  for (int f = threadIdx.x; f < (u_hi - u_lo) * (ZM / 4); f += 256) {
    const int u = u_lo + f / (ZM / 4), c4 = f % (ZM / 4);
    zl_f4 v = *reinterpret_cast<const zl_f4*>(sl + (size_t)u * ZM + 4 * c4);
    for (int s = 1; s < S; ++s) v += *reinterpret_cast<const zl_f4*>(sl + ((size_t)s * U + u) * ZM + 4 * c4);
    *reinterpret_cast<zl_f4*>(C + (int64_t)u * ldC + 4 * c4) = v;
    asum += (fabsf(v[0]) + fabsf(v[1])) + (fabsf(v[2]) + fabsf(v[3]));
  }
  const double tot = block_sum((double)asum, red);
  if (threadIdx.x == 0) partC[(size_t)r * partC_stride + j] = tot;
}

}  // namespace ctn
