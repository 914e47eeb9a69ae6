// plan_check.cpp - host-only driver of the planner (plan.cpp), for sanitizer runs on the CPU.
//
//   make -C contractn_amd/csrc asan      ->  ../lib/plan_check_asan   (-fsanitize=address,undefined)
//   plan_check_asan < plans.txt
//
// The planner is pure host code (label classification, layout choice, gather-table construction, workspace
// arena): exactly the kind of index arithmetic a sanitizer is good at, and it needs no GPU.  Input is a stream
// of plan descriptions (tests/test_planner_asan.py writes the golden fixtures' and the fuzz generator's):
//
//   plan <dtype 0|1> <n_inputs> <n_steps>
//   in <ndim> <dim>*ndim <label>*ndim          (n_inputs lines)
//   step <lhs> <rhs|-1> <out_ndim> <label>*out_ndim   (n_steps lines)
//
// For every plan one line goes to stdout: status, table entries, workspace bytes, flops, and a checksum over
// everything a kernel launch would read from the plan (tables, per-step shapes and modes) - every table entry
// is also checked against the extent of the tensor it indexes.
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

#include "plan.h"

using namespace ctn;

static uint64_t mix(uint64_t h, uint64_t v) { return (h ^ v) * 0x100000001b3ull; }

int main() {
  std::string tok;
  int n_plans = 0, n_fail = 0;
  while (std::cin >> tok) {
    if (tok != "plan") { fprintf(stderr, "expected 'plan', got '%s'\n", tok.c_str()); return 2; }
    ctn_plan_desc d{};
    std::cin >> d.dtype >> d.n_inputs >> d.n_steps;
    std::vector<int32_t> in_ndim, in_labels, lhs, rhs, ond, olab;
    std::vector<int64_t> in_dims;
    for (int i = 0; i < d.n_inputs; ++i) {
      int nd;
      std::cin >> tok >> nd;
      in_ndim.push_back(nd);
      for (int a = 0; a < nd; ++a) { int64_t x; std::cin >> x; in_dims.push_back(x); }
      for (int a = 0; a < nd; ++a) { int32_t x; std::cin >> x; in_labels.push_back(x); }
    }
    for (int s = 0; s < d.n_steps; ++s) {
      int l, r, nd;
      std::cin >> tok >> l >> r >> nd;
      lhs.push_back(l); rhs.push_back(r); ond.push_back(nd);
      for (int a = 0; a < nd; ++a) { int32_t x; std::cin >> x; olab.push_back(x); }
    }
    if (!std::cin) { fprintf(stderr, "truncated plan description\n"); return 2; }
    in_dims.push_back(0); in_labels.push_back(0); olab.push_back(0);   // keep .data() non-null for rank-0 tensors
    d.in_ndim = in_ndim.data(); d.in_dims = in_dims.data(); d.in_labels = in_labels.data(); d.in_strides = nullptr;
    d.step_lhs = lhs.data(); d.step_rhs = rhs.data(); d.step_out_ndim = ond.data(); d.step_out_labels = olab.data();
    d.stabilize = 1; d.min_norm = 1e-7;
    Plan P;
    std::string err;
    const int rc = build_plan(d, P, err);
    ++n_plans;
    if (rc != CTN_OK) { printf("plan %d rc=%d %s\n", n_plans, rc, err.c_str()); continue; }
    uint64_t h = 1469598103934665603ull;
    bool ok = true;
    for (int32_t v : P.tables) h = mix(h, (uint32_t)v);
    for (int64_t v : P.tables64) h = mix(h, (uint64_t)v);
    for (const Step& st : P.steps) {
      h = mix(h, (uint64_t)st.kernel * 131 + st.modeA * 17 + st.modeB * 5 + st.tileM + st.tileN);
      h = mix(h, (uint64_t)st.Bt); h = mix(h, (uint64_t)st.M); h = mix(h, (uint64_t)st.N); h = mix(h, (uint64_t)st.K);
      // every (batch, m, k) / (batch, k, n) / (batch, m, n) table combination must stay inside its tensor
      const bool tiled = st.kernel == CTN_KERNEL_MFMA_F32 || st.kernel == CTN_KERNEL_MFMA_F64 || st.kernel == CTN_KERNEL_DOT;
      if (!(tiled || st.chain_ok)) continue;
      const int32_t* T = P.tables.data();
      const int64_t* T8 = P.tables64.data();
      auto maxof = [&](int64_t off, int64_t n) { int64_t m = 0; for (int64_t i = 0; i < n; ++i) m = std::max<int64_t>(m, T[off + i]); return m; };
      auto maxof8 = [&](int64_t off, int64_t n) { int64_t m = 0; for (int64_t i = 0; i < n; ++i) m = std::max<int64_t>(m, T8[off + i]); return m; };
      const int64_t a_max = maxof8(st.t.obA, st.Bt) + maxof(st.t.omA, st.M) + maxof(st.t.okA, st.K);
      const int64_t c_max = maxof8(st.t.obC, st.Bt) + maxof(st.t.omC, st.M) + maxof(st.t.onC, st.N);
      if (a_max >= P.tensors[st.lhs].numel || c_max >= P.tensors[st.out].numel) ok = false;
      if (st.lhs2 >= 0 && !st.epw) {   // fused step: the second tensor of the A side
        const int64_t a2_max = maxof8(st.t.obA2, st.Bt) + maxof(st.t.omA2, st.M) + maxof(st.t.okA2, st.K);
        if (a2_max >= P.tensors[st.lhs2].numel) ok = false;
      }
      if (st.epw) {   // epilogue weights: row offset + the short label's values (unit-stride)
        if (maxof(st.t.omA2, st.M) + st.epw - 1 >= P.tensors[st.lhs2].numel || st.N % st.epw != 0) ok = false;
        if (P.tensors[st.out].numel * st.epw != st.Bt * st.M * st.N) ok = false;
      }
      h = mix(h, (uint64_t)st.epw);
      if (st.rhs >= 0) {
        const int64_t b_max = maxof8(st.t.obB, st.Bt) + maxof(st.t.onB, st.N) + maxof(st.t.okB, st.K);
        if (b_max >= P.tensors[st.rhs].numel) ok = false;
      }
    }
    if (!ok) ++n_fail;
    printf("plan %d rc=0 tables=%zu ws=%lld flops=%.17g sum=%016llx %s\n", n_plans, P.tables.size(),
           (long long)P.ws_bytes_per_replica, P.flops, (unsigned long long)h, ok ? "ok" : "TABLE-OUT-OF-RANGE");
  }
  return n_fail ? 1 : 0;
}
