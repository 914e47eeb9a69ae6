// Development probe: multi-stream capture patterns (fork by event, cross-stream waits, join by tail events) -
// which of them hipStreamEndCapture / hipGraphInstantiate of this ROCm accept.  hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_add(float* p, int i) { if (threadIdx.x == 0 && blockIdx.x == 0) p[i] += 1.f; }

int run(int n_side, int n_steps, unsigned ev_flags, int seed, bool wait_same_event_twice, int mode) {
  hipStream_t main_s;
  CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
  std::vector<hipStream_t> side(n_side);
  for (auto& s : side) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float* d;
  CK(hipMalloc(&d, 4096));
  CK(hipMemset(d, 0, 4096));
  std::vector<hipEvent_t> ev(n_steps);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, ev_flags));
  hipEvent_t fork;
  CK(hipEventCreateWithFlags(&fork, ev_flags));
  srand(seed);
  std::vector<int> lane(n_steps), tail(n_side + 1, -1);
  std::vector<char> joined(n_side + 1, 0);
  CK(hipStreamBeginCapture(main_s, hipStreamCaptureModeThreadLocal));
  CK(hipEventRecord(fork, main_s));
  for (int s = 0; s < n_steps; ++s) {
    const int ln = rand() % (n_side + 1);
    lane[s] = ln;
    hipStream_t st = ln == 0 ? main_s : side[ln - 1];
    if (ln != 0 && !joined[ln]) { CK(hipStreamWaitEvent(st, fork, 0)); joined[ln] = 1; }
    for (int t = 0; t < 2 && s > 0; ++t) {       // wait for up to two random earlier steps on other lanes
      const int dsp = rand() % s;
      // mode 1: a side lane never waits for another side lane; mode 2: only main waits for side lanes (pure fork-join);
      // mode 3: side lanes wait for side lanes, main never waits inside the loop
      const bool side_side = ln != 0 && lane[dsp] != 0;
      if (mode == 1 && side_side) continue;
      if (mode == 2 && ln != 0) continue;
      if (mode == 3 && ln == 0) continue;
      if (lane[dsp] != ln) {
        CK(hipStreamWaitEvent(st, ev[dsp], 0));
        if (wait_same_event_twice) CK(hipStreamWaitEvent(st, ev[dsp], 0));
      }
    }
    hipLaunchKernelGGL(k_add, dim3(1), dim3(64), 0, st, d, s);
    CK(hipEventRecord(ev[s], st));
    tail[ln] = s;
  }
  for (int ln = 1; ln <= n_side; ++ln)
    if (joined[ln] && tail[ln] >= 0) CK(hipStreamWaitEvent(main_s, ev[tail[ln]], 0));
  hipLaunchKernelGGL(k_add, dim3(1), dim3(64), 0, main_s, d, n_steps);
  hipGraph_t g = nullptr;
  printf("  end capture ...\n"); fflush(stdout);
  CK(hipStreamEndCapture(main_s, &g));
  hipGraphExec_t ge = nullptr;
  printf("  instantiate ...\n"); fflush(stdout);
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, main_s));
  CK(hipStreamSynchronize(main_s));
  std::vector<float> h(n_steps + 1);
  CK(hipMemcpy(h.data(), d, (n_steps + 1) * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (float v : h) bad += v != 3.f;
  printf("  ok: %d steps, %d side streams, flags %u, twice %d -> %d wrong\n", n_steps, n_side, ev_flags, (int)wait_same_event_twice, bad);
  return bad;
}

int main(int argc, char** argv) {
  const int n_side = argc > 1 ? atoi(argv[1]) : 2, n_steps = argc > 2 ? atoi(argv[2]) : 20;
  const unsigned flags = argc > 3 ? (unsigned)atoi(argv[3]) : hipEventDisableTiming;
  const bool twice = argc > 4 && atoi(argv[4]);
  const int mode = argc > 5 ? atoi(argv[5]) : 0;
  printf("probe side=%d steps=%d flags=%u\n", n_side, n_steps, flags); fflush(stdout);
  printf("mode %d\n", mode); return run(n_side, n_steps, flags, 1, twice, mode);
}
