// Development probe: HBM store bandwidth by access shape (what the epilogues of the tile kernels do vs a plain fill).
// hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o tools/bin/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// mode 0: wave instruction = 1 KiB contiguous (fill); mode 1: 4 rows x 256 B (row stride = ld floats); mode 2: 8 rows x 128 B;
// mode 3: as 1 but nontemporal; rows of a tile are `ld` floats apart, tiles tile the matrix [rows][ld]
template <int MODE>
__global__ __launch_bounds__(256) void k_store(float* __restrict__ p, long rows, long ld, int tiles_per_wg) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float4 v = make_float4(1.f + lane, 2.f, 3.f, 4.f);
  for (int t = 0; t < tiles_per_wg; ++t) {
    const long tile = (long)blockIdx.x * tiles_per_wg + t;          // a tile = 128 rows x 64 floats (32 KB), 4 waves x 32 rows
    const long tiles_per_row = ld / 64;
    const long r0 = (tile / tiles_per_row) * 128 + w * 32, c0 = (tile % tiles_per_row) * 64;
    if (r0 >= rows) return;
    if (MODE == 0) {
      float* base = p + tile * (128 * 64) + w * (32 * 64);
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(base + i * 256 + lane * 4) = v;
    } else {
      constexpr int RPI = MODE == 2 ? 8 : 4;            // rows per wave instruction
      constexpr int VW = 64 / RPI;                      // float4 per row segment
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i) {
        float* dst = p + (r0 + i * RPI + lane / VW) * ld + c0 + (lane % VW) * 4;
        if (MODE == 3) {
          typedef float vf4 __attribute__((ext_vector_type(4)));
          vf4 q = {v.x, v.y, v.z, v.w};
          __builtin_nontemporal_store(q, reinterpret_cast<vf4*>(dst));
        } else {
          *reinterpret_cast<float4*>(dst) = v;
        }
      }
    }
  }
}

int main() {
  const long rows = 1 << 15, ld = 4096;                 // 512 MiB
  float* d;
  CK(hipMalloc(&d, rows * ld * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long tiles = rows / 128 * (ld / 64);
  for (int mode = 0; mode < 4; ++mode)
    for (int tpw : {1, 4}) {
      const dim3 g((unsigned)(tiles / tpw));
      float best = 1e9;
      for (int it = 0; it < 5; ++it) {
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(k_store<0>, g, dim3(256), 0, 0, d, rows, ld, tpw);
        if (mode == 1) hipLaunchKernelGGL(k_store<1>, g, dim3(256), 0, 0, d, rows, ld, tpw);
        if (mode == 2) hipLaunchKernelGGL(k_store<2>, g, dim3(256), 0, 0, d, rows, ld, tpw);
        if (mode == 3) hipLaunchKernelGGL(k_store<3>, g, dim3(256), 0, 0, d, rows, ld, tpw);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("mode %d tiles/wg %d: %.3f ms  %.2f TB/s\n", mode, tpw, best, rows * ld * 4 / best / 1e9);
    }
  return 0;
}
