// Development tool: how fast does a VALU-only wave run while ANOTHER wave on the same SIMD streams MFMAs?
// 8 waves per workgroup (2 per SIMD): waves 0-3 run an fp32 MFMA loop (or nothing), waves 4-7 run a chain-free
// v_add loop; prints cycles per VALU instruction of the VALU waves with and without the MFMA neighbours.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_vs_mfma_probe.hip -o tools/bin/valu_vs_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NOPS>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, unsigned long long* mcyc, int mfma_iters, int valu_iters, int prio, int flip) {
  const int tid = threadIdx.x, w = flip ? ((threadIdx.x >> 6) + 4) % 8 : (threadIdx.x >> 6);   // flip: the VALU waves are waves 0-3
  float s = 0.f;
  if (w < 4) {
    f32x16 a0, a1, a2, a3;
    for (int e = 0; e < 16; ++e) { a0[e] = 0; a1[e] = 0; a2[e] = 0; a3[e] = 0; }
    float x = tid * 1e-3f + 1.f, y = tid * 2e-3f + 0.5f;
#define YIELD() do { if (NOPS >= 1) asm volatile("s_nop 15"); if (NOPS >= 2) asm volatile("s_nop 15"); if (NOPS >= 3) asm volatile("s_nop 15"); } while (0)
    const unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); YIELD();
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0); YIELD();
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0); YIELD();
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0); YIELD();
      }
    }
    const unsigned long long m1 = __builtin_amdgcn_s_memtime();
    if ((tid & 63) == 0 && mfma_iters) mcyc[blockIdx.x * 4 + (w & 3)] = m1 - m0;
    for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
  } else {
    if (prio) __builtin_amdgcn_s_setprio(3);
    int v[8] = {tid, tid + 1, tid + 2, tid + 3, tid + 4, tid + 5, tid + 6, tid + 7};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[u & 7]) : "v"(tid));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int q = 0; q < 8; ++q) s += (float)v[q];
    if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (w - 4)] = t1 - t0;
  }
  out[blockIdx.x * 512 + tid] = s;
}

int main() {
  int ncu = 256;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  float* out; unsigned long long *cyc, *mcyc;
  hipMalloc(&out, (size_t)ncu * 512 * 4);
  hipMalloc(&cyc, (size_t)ncu * 4 * 8);
  hipMalloc(&mcyc, (size_t)ncu * 4 * 8);
  const int valu_iters = 2000;   // 64000 VALU instructions per wave
  auto run = [&](auto tag, int nops) {
    for (int mf : {0, 20000}) {
      hipLaunchKernelGGL(k<decltype(tag)::value>, dim3(ncu), dim3(512), 0, 0, out, cyc, mcyc, mf, valu_iters, 0, 0);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(ncu * 4), hm(ncu * 4);
      hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
      hipMemcpy(hm.data(), mcyc, hm.size() * 8, hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.end()); std::sort(hm.begin(), hm.end());
      printf("%d x s_nop 15 after each MFMA, MFMA neighbour %s: %.2f cycles per VALU instruction; %.1f cycles per MFMA\n", nops,
             mf ? "streaming" : "idle     ", (double)h[h.size() / 2] / (valu_iters * 32.0),
             mf ? (double)hm[hm.size() / 2] / (mf * 32.0) : 0.0);
    }
  };
  run(std::integral_constant<int, 0>{}, 0);
  run(std::integral_constant<int, 1>{}, 1);
  run(std::integral_constant<int, 2>{}, 2);
  run(std::integral_constant<int, 3>{}, 3);
  return 0;
}
