// kernels_sweep.h - an MPS applied to a batch of inputs: every site of the chain inside ONE launch, a block of 16 inputs
// per workgroup.  Part of the gfx950 contraction engine (see engine.hip for the overview).
#pragma once
#include "kernels_mfma_g.h"

namespace ctn {

// ---------------------------------------------------------------------------
// K-sweep-f32.  BASELINE configs[3b] (reference README Fig. 1d: an MPS classifier on a batch hyperedge; the reference
// walks it as 2 pairwise steps per site, einsum.py:341-391): per site
//
//     C[b, (p, r)] = sum_l E[b, l] W_s[l, p, r]        E'[b, r] = sum_p x_s[b, p] C[b, (p, r)]
//
// (one epilogue-summed GEMM step of the plan, 4096 x 1024 x 256 for 4096 inputs at bond 256).  One launch per site is
// ONE round of 256 workgroups that start and end together: a quarter of its time is prologue, hand-over and epilogue
// with nothing to cover them, and the site launches depend on each other (LAB_NOTES R3.4).  But the rows b never meet:
// a block of rows can walk the WHOLE chain on its own.  Here a workgroup owns 16 inputs - 4096 inputs = 256
// workgroups, one per CU - and keeps their state E (16 x 256) in LDS from the first site to the last; the cores W_s
// (1 MiB each, the same for every workgroup: L2 hits) stream straight from global memory into MFMA operand registers,
// no LDS, no barrier inside a site.
//
// 8 waves = 4 ranges of 64 values of r x the 2 halves of l.  v_mfma_f32_16x16x4_f32, D[i][j = b]:
//   A operand: lane (i = lane & 15, kg = lane >> 4) holds W_s[l][p][64 w4 + 4 i + c] for c = 0..3 - one 16-byte load
//              (every CU reads every core from its XCD's L2: 8-byte requests reach 0.6 of the rate of 16-byte ones)
//              feeds the four accumulators (p, c): row i of accumulator c is r = 64 w4 + 4 i + c;
//   B operand: lane (j = b, kg) holds E[b][l];
//   k-step t = 0..3 of a group G of 16 values of l pairs lane group kg with l = 16 G + 4 kg + t, so that a lane's four
//   B operands of a group are ONE 16-byte LDS read (any pairing is fine as long as both operands follow it);
//   D: lane (b, g = lane >> 4) holds rows i = 4 g + e in register e: after the sum over p a lane has its half's
//   share of E'[b][64 w4 + 16 g + 0..15]; the two halves of l meet through LDS, each finishing eight of the sixteen.
// The cores are requested 4 k-steps (one loop body) ahead into a register queue that runs on across site
// boundaries; two barriers per site (E' complete; its abs-sum known), nothing but LDS traffic is waited for.
//
// Stabilisation.  The reference rescales by the mean |.| of the WHOLE tensor after every step (einsum.py:97-106),
// which no workgroup knows.  A workgroup rescales its own 16 rows by THEIR mean instead (any bounded scale keeps
// the products in range) and records, per site, the abs-sum a[s][j] of its un-rescaled result and the scale
// s[s][j] it applied.  With g[j][s] = sum_{i <= s} log s[i][j], the whole tensor's mean after site s is
// exp(Z_s), Z_s = log((1 / numel) sum_j a[s][j] exp(g[j][s - 1])) (k_sweep_z, a log-sum-exp per site), and the
// reference's rescale factors follow from the Z_s by its own recurrence (k_sweep_finish: rescale_s = exp(Z_s) / R_{s-1}
// when its norm exceeds min_norm, R_s = exp(Z_s) then, else R_s = R_{s-1}); k_sweep_finish writes them where the
// per-step launches would have left their abs-sums, and brings the last site's rows - stored by the sweep with each
// block's own scale - to the one scale the reference's stored tensor has.  Same numbers as the per-site launches up
// to rounding; four launches instead of one per site.
//
// Conditions (engine.hip, sweep_match): fp32, |l| = |r| = 64, 128, 256 or 512, |p| = 2 or 4, W_s and x_s network inputs
// with r and p unit-stride, E row-major.  (The description above is the bond-256, d = 4 instance.)
// ---------------------------------------------------------------------------
struct SweepArgs {
  void* const* ptrs;         // [R][n_tensors]
  int32_t n_tensors;
  const int32_t* site_ids;   // [S][2]: tensor ids of (W_s, x_s)
  int32_t idIn, idOut;       // the chain's input E [b][l] and its output E' [b][r]
  int32_t S, J, M;           // sites, row blocks (ceil(rows / 16)), rows
  int64_t ldIn, ldOut;       // row strides of input and output (elements)
  int64_t ldWl, ldWp;        // W_s[l][p][r]: strides of l and p (r unit-stride)
  int64_t ldX;               // x_s[b][p]: row stride (p unit-stride)
  const double* partIn;      // the input's producer partials (nullptr: a network input)
  int32_t PIn, strideIn;
  double numelIn, min_norm;
  double* rec_a;             // [R][S][J] abs-sum of a block's un-rescaled result
  float* rec_s;              // [R][S][J] the scale the block then applied (1 after the last site)
  unsigned long long* dbg;   // CTN_STAMPS builds only
};

constexpr int SWR = 16;     // inputs per workgroup
constexpr int SWQ = 4;      // k-steps of W in flight per wave (one loop body; 8 measured: the same - the CU's own load path is the limit)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) char* sw_gptr;   // global memory, said so: plain global_load / store
typedef __attribute__((address_space(1))) float* sw_gout;

// <D, P>: bond dimension (|l| = |r|: 64, 128, 256 or 512) and physical dimension (2 or 4).  Waves: D / 64 ranges of 64
// values of r x NL parts of l (8 waves; 4 for D = 64): D = 256 as described above (4 x 2), D = 512 8 x 1 (no hand-over),
// D = 128 2 x 4, D = 64 1 x 4.  The image rows are D + 4 floats apart (the 16-byte reads of 16 rows spread over the banks).
template <int D, int P>
__global__ __launch_bounds__(D == 64 ? 256 : 512, 1) void k_sweep_f32(SweepArgs a) {
  constexpr int NWV = D == 64 ? 4 : 8, NT = 64 * NWV;
  constexpr int NR = D / 64, NL = NWV / NR;   // ranges of r, parts of l
  constexpr int LW = D / NL, NG = LW / 16;    // values of l per wave; its groups of 16 per site
  constexpr int EK = 4 / NL;                  // of the four e (16-byte pieces of a lane's 16 values of r), a wave finishes EK
  constexpr int LD = D + 4;
  static_assert((D == 64 || D == 128 || D == 256 || D == 512) && (P == 2 || P == 4) && NG >= 1 && (NG & (NG - 1)) == 0, "shape");
  __shared__ __attribute__((aligned(16))) float img[2][SWR * LD];
  __shared__ __attribute__((aligned(16))) float xch[NL > 1 ? NWV * 4 * 64 * 4 : 4];   // the parts' hand-over: 4 x 16 bytes per lane
  __shared__ double red[NWV];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w4 = w % NR, kh = w / NR;
  const int i16 = lane & 15, kg = lane >> 4;
  const int j = blockIdx.x, r = blockIdx.y;
  void* const* tp = a.ptrs + (size_t)r * a.n_tensors;
  const size_t rec0 = (size_t)r * a.S * a.J + j;
  const int rows = min(SWR, a.M - SWR * j);   // the last block of a batch that is not a multiple of 16: its other rows stay zero

  // the chain's input, normalised by its producer's mean (the lazy rescale: reference einsum.py:387 on the step before)
  {
    double pv = 0.0;
    if (a.partIn) {
      const double* __restrict__ pr = a.partIn + (size_t)r * a.strideIn;
      pv = pr[min(lane, a.PIn - 1)];
      if (a.PIn > 64)
        for (int i = lane + 64; i < a.PIn; i += 64) pv += pr[i];
      pv = lane < a.PIn ? pv : 0.0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) pv += __shfl_xor(pv, o, 64);
    }
    const float nI = (float)pv;
    const float inv = (a.partIn && nI > (float)a.min_norm) ? 1.0f / (nI / (float)a.numelIn) : 1.0f;
    const float* __restrict__ Ein = (const float*)tp[a.idIn] + (int64_t)(SWR * j) * a.ldIn;
    for (int i = tid; i < SWR * D / 4; i += NT) {
      const int row = i / (D / 4), c4 = i - row * (D / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < rows) v = *reinterpret_cast<const float4*>(Ein + (int64_t)row * a.ldIn + 4 * c4);
      *reinterpret_cast<float4*>(&img[0][row * LD + 4 * c4]) = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
    }
  }

  // a lane's own offsets into a core (bytes): row 4 kg of a group of its wave's part of l, its four columns r of every p
  uint32_t voff[P];
#pragma unroll
  for (int p = 0; p < P; ++p)
    voff[p] = (uint32_t)(((int64_t)(LW * kh + 4 * kg) * a.ldWl + (int64_t)p * a.ldWp + 64 * w4 + 4 * i16) * 4);
  const int64_t stepW = a.ldWl * 4;           // next k-step of a group (bytes)

  sw_gptr Wcur = (sw_gptr)tp[a.site_ids[0]];
  sw_gptr Xcur = (sw_gptr)tp[a.site_ids[1]];
  f32x4 wq[SWQ][P];
  // the groups of a site are walked from a block-dependent start (CUs of one XCD then ask its L2 for different lines)
  const int rot = (j >> 3) & (NG - 1);        // (-3 % of the time against every block starting at group 0)
  sw_gptr wpf = Wcur + (int64_t)(16 * rot) * stepW;   // the group being requested (wave-uniform)
  auto wrequest = [&](f32x4 (&dst)[P], int t) {
#pragma unroll
    for (int p = 0; p < P; ++p) dst[p] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(wpf + (int64_t)t * stepW + voff[p]);
  };
#pragma unroll
  for (int u = 0; u < SWQ; ++u) wrequest(wq[u], u);

  f32x4 acc[P][4];
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[p][c][e] = 0.f;

  __syncthreads();                            // the input image is in place
#ifdef CTN_STAMPS
  unsigned long long st_loop = 0, st_epi = 0, st0 = __builtin_amdgcn_s_memtime();
  const unsigned long long st_begin = st0;
#endif
  float inv_s = 1.0f;
  int cur = 0;
  sw_gptr Wnext = Wcur, Xnext = Xcur;
  for (int s = 0; s < a.S; ++s) {
    const bool last = s + 1 == a.S;
    // next site's tensors (the last site re-requests its own first k-steps: in bounds, never used)
    const int sn = last ? s : s + 1;
    Wnext = (sw_gptr)tp[a.site_ids[2 * sn]];
    Xnext = (sw_gptr)tp[a.site_ids[2 * sn + 1]];
    float xr[P];                               // the inputs' weights of this site (rows beyond the batch: 0)
    {
      const int row = min(SWR * j + i16, a.M - 1);
      sw_gptr xp = Xcur + (int64_t)row * a.ldX * 4;
      if constexpr (P == 4) {
        const f32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(xp);
        xr[0] = v.x; xr[1] = v.y; xr[2] = v.z; xr[3] = v.w;
      } else {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 v = *reinterpret_cast<const __attribute__((address_space(1))) f32x2*>(xp);
        xr[0] = v.x; xr[1] = v.y;
      }
    }
    const float* erow = &img[cur][i16 * LD + LW * kh + 4 * kg];
    float4 ef = *reinterpret_cast<const float4*>(erow + 16 * rot);
#pragma unroll 1
    for (int G = 0; G < NG; ++G) {
      const int gn = (G + 1 + rot) & (NG - 1);         // the next group: its B operands, its cores' requests
      wpf = (G == NG - 1 ? Wnext : Wcur) + (int64_t)(16 * gn) * stepW;   // (after the last group: the next site's first)
      const float4 en = *reinterpret_cast<const float4*>(erow + 16 * gn);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float ev = t == 0 ? ef.x : t == 1 ? ef.y : t == 2 ? ef.z : ef.w;
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            // (LAB_NOTES R3.6, cycles per site of the slower wave of a SIMD at D = 256, P = 4: these MFMAs alone, on a queue
            // that is never refilled, 34 k - the ideal is 32.8 k; the stream alone, one FMA here, ~36 k; together 44 k, on
            // 128 CUs as on 256: a CU's own load path, not the L2, sets the pace)
            acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[t][p][c], ev, acc[p][c], 0, 0, 0);
          }
        wrequest(wq[t], t);
        if (t == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // the next group's B operands first
#pragma unroll
        for (int p = 0; p < P; ++p) {        // a request after every four MFMAs (all of them behind the k-step's MFMAs: +5 % time)
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
      ef = en;
    }
#ifdef CTN_STAMPS
    { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); st_loop += t1 - st0; st0 = t1; }
#endif
    // ---- the site's epilogue: E'[b][r] = sum_p x[b][p] C[b][(p, r)], this wave's part of the sum over l; a lane has
    // r = 64 w4 + 16 kg + 4 e + c.  The NL parts meet through LDS: every wave leaves its four 16-byte pieces e there and
    // finishes EK = 4 / NL of them - the sum over the parts in their order.  The state in LDS is the un-rescaled one: its
    // scale goes into the weights - P multiplications per site.  (ds_add_f32 into a zeroed image instead: 12 k cycles.)
    float xs[P];
#pragma unroll
    for (int p = 0; p < P; ++p) xs[p] = (SWR * j + i16 < a.M ? xr[p] : 0.f) * inv_s;
    const int nxt = cur ^ 1;
    float o[4][4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float v = xs[0] * acc[0][c][e];
#pragma unroll
        for (int p = 1; p < P; ++p) v = fmaf(xs[p], acc[p][c][e], v);
        o[e][c] = v;
      }
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[p][c][e] = 0.f;
    if constexpr (NL > 1) {
      float4* gv = reinterpret_cast<float4*>(xch) + (w * 4) * 64 + lane;
#pragma unroll
      for (int e = 0; e < 4; ++e) gv[e * 64] = make_float4(o[e][0], o[e][1], o[e][2], o[e][3]);
      // two barriers per site; only the LDS traffic is waited for - the cores requested ahead stay in flight
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      float asum = 0.f;
#pragma unroll
      for (int q = 0; q < EK; ++q) {
        const int e = EK * kh + q;             // (wave-uniform)
        float4 f;
        if constexpr (NL > 1) {
          f = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int part = 0; part < NL; ++part) {
            const float4 tk = (reinterpret_cast<const float4*>(xch) + ((part * NR + w4) * 4 + e) * 64)[lane];
            f = part == 0 ? tk : make_float4(f.x + tk.x, f.y + tk.y, f.z + tk.z, f.w + tk.w);
          }
        } else {
          f = make_float4(o[q][0], o[q][1], o[q][2], o[q][3]);
        }
        const int col = 64 * w4 + 16 * kg + 4 * e;
        if (last) {   // the last site's rows leave with this block's own scale (k_sweep_finish brings them to the common one)
          if (i16 < rows) {
            sw_gout og = (sw_gout)tp[a.idOut] + (int64_t)(SWR * j + i16) * a.ldOut + col;
            *reinterpret_cast<__attribute__((address_space(1))) f32x4*>(og) = f32x4{f.x, f.y, f.z, f.w};
          }
        } else {
          *reinterpret_cast<float4*>(&img[nxt][i16 * LD + col]) = f;
        }
        asum += (fabsf(f.x) + fabsf(f.y)) + (fabsf(f.z) + fabsf(f.w));
      }
      double part = (double)asum;
#pragma unroll
      for (int of = 32; of > 0; of >>= 1) part += __shfl_xor(part, of, 64);
      if (lane == 0) red[w] = part;
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < NWV; ++i) tot += red[i];
    const float sc = (!last && tot > 1e-30) ? (float)(tot / (double)(SWR * D)) : 1.0f;
    inv_s = 1.0f / sc;
    if (tid == 0) {
      a.rec_a[rec0 + (size_t)s * a.J] = tot;
      a.rec_s[rec0 + (size_t)s * a.J] = sc;
    }
    cur = nxt;
    Wcur = Wnext;
    Xcur = Xnext;
#ifdef CTN_STAMPS
    { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); st_epi += t1 - st0; st0 = t1; }
#endif
  }
#ifdef CTN_STAMPS
  if (a.dbg && (tid == 0 || tid == NT - 64)) {
    unsigned long long* d = a.dbg + ((size_t)r * a.J + j) * 8 + (tid ? 4 : 0);
    d[0] = st_begin; d[1] = st0; d[2] = st_loop; d[3] = st_epi;
  }
#endif
}

// The bookkeeping behind a sweep: three short launches.  (Double-precision log / exp are long dependent chains: a thread
// that walks the sites and takes one of them per site spends 70 us on a 100-site chain; every transcendental below is
// evaluated once, in parallel, and the walks over the sites only add and compare.)
//
// k_sweep_logs: la[s][j] = log a[s][j] (-inf for an all-zero block), ls[s][j] = log s[s][j].  grid (S, R), 256 threads.
__global__ __launch_bounds__(256) void k_sweep_logs(const double* __restrict__ rec_a, const float* __restrict__ rec_s, int S, int J,
                                                    double* __restrict__ la, double* __restrict__ ls) {
  const size_t base = ((size_t)blockIdx.y * S + blockIdx.x) * J;
  for (int j = threadIdx.x; j < J; j += 256) {
    const double av = rec_a[base + j];
    la[base + j] = av > 0.0 ? log(av) : -INFINITY;
    ls[base + j] = log((double)rec_s[base + j]);
  }
}

// k_sweep_z: Z_s = log of the whole tensor's mean |.| after site s, a log-sum-exp over the blocks of
// la[s][j] + sum_{i < s} ls[i][j] (see k_sweep_f32).  grid (S, R), 256 threads; fixed-order sums.
__global__ __launch_bounds__(256) void k_sweep_z(const double* __restrict__ la, const double* __restrict__ ls, int S, int J,
                                                 double numel, double* __restrict__ Z) {
  __shared__ double red[4];
  const int s = blockIdx.x, r = blockIdx.y;
  const double* pa = la + ((size_t)r * S + s) * J;
  const double* ps = ls + (size_t)r * S * J;
  double mx = -INFINITY;
  for (int j = threadIdx.x; j < J; j += 256) {
    double g = 0.0;
    for (int i = 0; i < s; ++i) g += ps[(size_t)i * J + j];
    mx = fmax(mx, pa[j] + g);
  }
#pragma unroll
  for (int of = 32; of > 0; of >>= 1) mx = fmax(mx, __shfl_xor(mx, of, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  __syncthreads();
  if (mx == -INFINITY) {                      // an all-zero tensor
    if (threadIdx.x == 0) Z[(size_t)r * S + s] = -INFINITY;
    return;
  }
  double sum = 0.0;
  for (int j = threadIdx.x; j < J; j += 256) {
    double g = 0.0;
    for (int i = 0; i < s; ++i) g += ps[(size_t)i * J + j];
    const double t = pa[j] + g;
    if (t > -INFINITY) sum += exp(t - mx);
  }
  const double tot = block_sum(sum, red);
  if (threadIdx.x == 0) Z[(size_t)r * S + s] = mx + log(tot) - log(numel);
}

// k_sweep_finish: the reference's rescale factors of a sweep's steps from the Z_s, and the last site's rows at the
// common scale.  grid (J, R), 256 threads.
struct SweepFinish {
  void* const* ptrs;
  int32_t n_tensors, idOut, S, J, R, D, M;   // D: bond dimension; M: rows (the last block may hold fewer than 16)
  int64_t ldOut;
  const double* Z;           // [R][S]
  const double* ls;          // [R][S][J] log of the scales the blocks applied
  const int64_t* part_off;   // [S]: the step's region in the partials buffer starts at part_off[s] * R doubles
  const int32_t* part_slots; // [S]: slots per replica of that region
  double* partials;
  double numel, min_norm;
};

constexpr int kSweepMaxSites = 1024;

__global__ __launch_bounds__(256) void k_sweep_finish(SweepFinish f) {
  __shared__ double lognorm[kSweepMaxSites];
  __shared__ double fac_log;
  const int j = blockIdx.x, r = blockIdx.y;
  const double* Z = f.Z + (size_t)r * f.S;
  if (threadIdx.x == 0) {
    // the reference's recurrence (einsum.py:97-106 over the chain's steps), in logs: norm_s = numel exp(Z_s) / R_{s-1};
    // rescaled iff norm_s > min_norm, then R_s = exp(Z_s)
    const double lnum = log(f.numel), lmin = f.min_norm > 0.0 ? log(f.min_norm) : -INFINITY;
    double logR = 0.0, logR_before_last = 0.0;
    for (int s = 0; s < f.S; ++s) {
      const double ln = Z[s] == -INFINITY ? -INFINITY : lnum + Z[s] - logR;
      if (s + 1 == f.S) logR_before_last = logR;
      lognorm[s] = ln;
      if (ln > lmin) logR = Z[s];
    }
    // this block's rows of the last site: V = W exp(g[j][S - 2]); the reference's stored tensor = V / R_{S-2}
    double g = 0.0;
    const double* ps = f.ls + (size_t)r * f.S * f.J + j;
    for (int i = 0; i + 1 < f.S; ++i) g += ps[(size_t)i * f.J];
    fac_log = g - logR_before_last;
  }
  __syncthreads();
  if (j == 0) {                               // what each step's own launch would have left: its abs-sum, in slot 0
    for (int s = threadIdx.x; s < f.S; s += 256) {
      double* dst = f.partials + (size_t)f.part_off[s] * f.R + (size_t)r * f.part_slots[s];
      dst[0] = lognorm[s] == -INFINITY ? 0.0 : exp(lognorm[s]);
      for (int i = 1; i < f.part_slots[s]; ++i) dst[i] = 0.0;
    }
  }
  const float fac = (float)exp(fac_log);
  float* out = (float*)f.ptrs[(size_t)r * f.n_tensors + f.idOut] + (int64_t)(SWR * j) * f.ldOut;
  const int rows = min(SWR, f.M - SWR * j), q4 = f.D / 4;
  for (int i = threadIdx.x; i < rows * q4; i += 256) {
    const int row = i / q4, c4 = i - row * q4;
    float4* p = reinterpret_cast<float4*>(out + (int64_t)row * f.ldOut) + c4;
    float4 v = *p;
    v.x *= fac; v.y *= fac; v.z *= fac; v.w *= fac;
    *p = v;
  }
}

}  // namespace ctn
