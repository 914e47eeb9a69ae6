// Development tool: does the chip hold a different clock on v_mfma_f32_32x32x2_f32 vs 16x16x4_f32?
// Bare MFMA loops on pseudo-random operands (optionally re-read from LDS every k-step, like a GEMM
// main loop), same 64 accumulator registers per wave; prints TFLOP/s and in-kernel clock.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o tools/bin/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rnd(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return ((int)(x & 0xffffff) - 0x800000) * (1.0f / 0x800000) * 0.0625f;
}

template <int SHAPE, bool LDS, int NV = 0>
__global__ __launch_bounds__(256) void k_probe(float* out, unsigned long long* stamps, int iters) {
  __shared__ float sm[8192];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 256) sm[i] = rnd(i * 2654435761u + blockIdx.x);
  __syncthreads();
  float ra[8], rb[8];
  for (int i = 0; i < 8; ++i) { ra[i] = rnd(tid * 977 + i * 13 + blockIdx.x * 7919); rb[i] = rnd(tid * 613 + i * 29 + 5); }
  f32x16 a32[4];
  f32x4 a16[16];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) a32[i][e] = 0.f;
  for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) a16[i][e] = 0.f;
  int vv[8] = {tid, tid + 1, tid + 2, tid + 3, tid + 4, tid + 5, tid + 6, tid + 7};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int q = 0; q < NV; ++q) asm volatile("v_add_u32 %0, %0, %1" : "+v"(vv[q & 7]) : "v"(tid));
      float x0, x1, y0, y1;
      if (LDS) {
        const int base = ((it * 8 + u) & 15) * 512;
        x0 = sm[base + lane]; x1 = sm[base + 64 + lane]; y0 = sm[base + 128 + lane]; y1 = sm[base + 192 + lane];
      } else {
        x0 = ra[u]; x1 = ra[(u + 3) & 7]; y0 = rb[u]; y1 = rb[(u + 5) & 7];
      }
      if (SHAPE == 32) {  // 2x2 blocking: 4 MFMAs of 4096 flops
        a32[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, a32[0], 0, 0, 0);
        a32[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, a32[1], 0, 0, 0);
        a32[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, a32[2], 0, 0, 0);
        a32[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, a32[3], 0, 0, 0);
      } else {            // same flops: 8 MFMAs of 2048 flops (2 x 4 blocking of 16x16 tiles, k = 4)
        a16[(u & 1) * 8 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, y0, a16[(u & 1) * 8 + 0], 0, 0, 0);
        a16[(u & 1) * 8 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, y1, a16[(u & 1) * 8 + 1], 0, 0, 0);
        a16[(u & 1) * 8 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, y0, a16[(u & 1) * 8 + 2], 0, 0, 0);
        a16[(u & 1) * 8 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, y1, a16[(u & 1) * 8 + 3], 0, 0, 0);
        a16[(u & 1) * 8 + 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(y0, x0, a16[(u & 1) * 8 + 4], 0, 0, 0);
        a16[(u & 1) * 8 + 5] = __builtin_amdgcn_mfma_f32_16x16x4f32(y0, x1, a16[(u & 1) * 8 + 5], 0, 0, 0);
        a16[(u & 1) * 8 + 6] = __builtin_amdgcn_mfma_f32_16x16x4f32(y1, x0, a16[(u & 1) * 8 + 6], 0, 0, 0);
        a16[(u & 1) * 8 + 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(y1, x1, a16[(u & 1) * 8 + 7], 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int q = 0; q < 8; ++q) s += (float)vv[q];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += a32[i][e];
  for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) s += a16[i][e];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int SHAPE, bool LDS, int NV = 0>
void run(int ncu, int wps) {
  const int iters = 12000, blocks = ncu * wps, threads = 256;
  float* out; unsigned long long* st;
  hipMalloc(&out, (size_t)blocks * threads * 4);
  hipMalloc(&st, (size_t)blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_probe<SHAPE, LDS, NV>), dim3(blocks), dim3(threads), 0, 0, out, st, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int w = 0; w < 4; ++w) hipLaunchKernelGGL((k_probe<SHAPE, LDS, NV>), dim3(blocks), dim3(threads), 0, 0, out, st, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 4;
  std::vector<unsigned long long> h(blocks * 2);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> clk;
  for (int b = 0; b < blocks; ++b) clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0);
  std::sort(clk.begin(), clk.end());
  const double flops = (double)blocks * 4 * iters * 8.0 * 4 * 4096.0;
  printf("shape %dx%d %s +%d VALU per 4 MFMA, waves/SIMD %d: %6.1f TFLOP/s  clock median %.0f MHz  kernel %.2f ms\n", SHAPE, SHAPE,
         LDS ? "LDS-fed " : "reg-fed ", NV, wps, flops / (ms * 1e-3) / 1e12, clk[clk.size() / 2], ms);
  hipFree(out); hipFree(st);
}

// `mfma_shape_probe power`: each shape for ~4 s back to back, so that the power controller settles; prints the
// rate and in-kernel clock of the last second (is one shape cheaper in energy per flop = faster at the cap?)
template <int SHAPE, bool LDS>
void run_settled(int ncu, int wps, double seconds) {
  const int iters = 12000, blocks = ncu * wps, threads = 256;
  float* out; unsigned long long* st;
  hipMalloc(&out, (size_t)blocks * threads * 4);
  hipMalloc(&st, (size_t)blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_probe<SHAPE, LDS, 0>), dim3(blocks), dim3(threads), 0, 0, out, st, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_probe<SHAPE, LDS, 0>), dim3(blocks), dim3(threads), 0, 0, out, st, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms1; hipEventElapsedTime(&ms1, e0, e1);
  const int n = (int)(seconds * 1e3 / ms1) + 1, tail = n / 4 + 1;
  for (int w = 0; w < n - tail; ++w) hipLaunchKernelGGL((k_probe<SHAPE, LDS, 0>), dim3(blocks), dim3(threads), 0, 0, out, st, iters);
  hipEventRecord(e0);
  for (int w = 0; w < tail; ++w) hipLaunchKernelGGL((k_probe<SHAPE, LDS, 0>), dim3(blocks), dim3(threads), 0, 0, out, st, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= tail;
  std::vector<unsigned long long> h(blocks * 2);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> clk;
  for (int b = 0; b < blocks; ++b) clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0);
  std::sort(clk.begin(), clk.end());
  const double flops = (double)blocks * 4 * iters * 8.0 * 4 * 4096.0;
  printf("settled %.0f s: shape %dx%d %s waves/SIMD %d: first launch %6.1f TFLOP/s, last quarter %6.1f TFLOP/s, clock median %.0f MHz\n",
         seconds, SHAPE, SHAPE, LDS ? "LDS-fed" : "reg-fed", wps, flops / (ms1 * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12,
         clk[clk.size() / 2]);
  fflush(stdout);
  hipFree(out); hipFree(st);
}

int main(int argc, char** argv) {
  int ncu = 256;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  if (argc > 1) {
    for (int rep = 0; rep < 2; ++rep) {
      run_settled<32, true>(ncu, 2, 4.0);
      run_settled<16, true>(ncu, 2, 4.0);
    }
    run_settled<32, false>(ncu, 2, 4.0);
    run_settled<16, false>(ncu, 2, 4.0);
    return 0;
  }
  for (int rep = 0; rep < 1; ++rep)
    for (int wps = 1; wps <= 2; ++wps) {
      run<32, false>(ncu, wps); run<16, false>(ncu, wps);
      run<32, true>(ncu, wps);  run<16, true>(ncu, wps);
      run<32, true, 1>(ncu, wps); run<32, true, 2>(ncu, wps); run<32, true, 4>(ncu, wps); run<32, true, 8>(ncu, wps);
    }
  return 0;
}
