// plan.cpp - contraction-list lowering (host only, no HIP).  See plan.h.
#include "plan.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <numeric>

namespace ctn {
namespace {

enum LabelClass { kBatch = 0, kM = 1, kN = 2, kK = 3 };

struct LabelInfo {
  int32_t label = 0;
  int64_t ext = 0;
  int64_t sA = 0, sB = 0, sC = 0;
  bool inA = false, inB = false, inC = false;
  int64_t sA2 = 0;      // fused steps: stride in the SECOND tensor of the A side (A = X (.) Y); sA is X's
  bool inX = false, inA2 = false;  // fused steps: label carried by X / by Y (inA = either)
  int firstPos = 0;  // for stable ordering
  int cls = kK;
};

std::string fmt(const char* f, long long a = 0, long long b = 0, long long c = 0) {
  char buf[256];
  snprintf(buf, sizeof buf, f, a, b, c);
  return buf;
}

int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// Row-major enumeration of a label group: table[i] = sum_j idx_j * stride_j
// Grid of a streaming (grid-stride) step that wants `want` workgroups of 256 threads: all of them while each can keep
// its own abs-sum partial; a step of moderate size (at most 8 rounds per thread) runs on kMaxPartials workgroups
// rather than pay the collapse launch (5 us next to a 10 us step of a batched-MPS site); beyond that the full grid,
// collapsed.
int stream_grid(int64_t want) {
  if (want <= kMaxPartials) return (int)std::max<int64_t>(want, 1);
  return want <= kStreamMaxBlocks ? kMaxPartials : kStreamMaxBlocks;
}

void build_table(const std::vector<const LabelInfo*>& group, int which /*0 A,1 B,2 C*/,
                 int64_t padded, std::vector<int32_t>& out) {
  int64_t n = 1;
  for (auto* l : group) n *= l->ext;
  out.assign(std::max<int64_t>(padded, n), 0);
  std::vector<int64_t> idx(group.size(), 0);
  for (int64_t i = 0; i < n; ++i) {
    int64_t off = 0;
    for (size_t j = 0; j < group.size(); ++j) {
      int64_t s = which == 0 ? group[j]->sA : which == 1 ? group[j]->sB : which == 2 ? group[j]->sC : group[j]->sA2;
      off += idx[j] * s;
    }
    out[i] = (int32_t)off;
    for (int j = (int)group.size() - 1; j >= 0; --j) {
      if (++idx[j] < group[j]->ext) break;
      idx[j] = 0;
    }
  }
}

// the same enumeration in 64 bits: batch and outer-group tables (Plan::tables64)
void build_table64(const std::vector<const LabelInfo*>& group, int which, int64_t padded, std::vector<int64_t>& out) {
  int64_t n = 1;
  for (auto* l : group) n *= l->ext;
  out.assign(std::max<int64_t>(padded, n), 0);
  std::vector<int64_t> idx(group.size(), 0);
  for (int64_t i = 0; i < n; ++i) {
    int64_t off = 0;
    for (size_t j = 0; j < group.size(); ++j) {
      int64_t s = which == 0 ? group[j]->sA : which == 1 ? group[j]->sB : which == 2 ? group[j]->sC : group[j]->sA2;
      off += idx[j] * s;
    }
    out[i] = off;
    for (int j = (int)group.size() - 1; j >= 0; --j) {
      if (++idx[j] < group[j]->ext) break;
      idx[j] = 0;
    }
  }
}

// largest offset a group's table can hold for operand `which`
int64_t span_of(const std::vector<LabelInfo*>& group, int which) {
  int64_t sp = 0;
  for (auto* l : group) {
    const int64_t s = which == 0 ? std::max(l->sA, l->sA2) : which == 1 ? l->sB : l->sC;
    sp += (l->ext - 1) * s;
  }
  return sp;
}

// simple first-fit allocator over one replica's workspace
struct Arena {
  struct Block { int64_t off, size; };
  std::vector<Block> free_list;
  int64_t top = 0;
  int64_t alloc(int64_t bytes) {
    bytes = round_up(std::max<int64_t>(bytes, 1), kAlign);
    for (size_t i = 0; i < free_list.size(); ++i) {
      if (free_list[i].size >= bytes) {
        int64_t off = free_list[i].off;
        free_list[i].off += bytes;
        free_list[i].size -= bytes;
        if (free_list[i].size == 0) free_list.erase(free_list.begin() + i);
        return off;
      }
    }
    int64_t off = top;
    top += bytes;
    return off;
  }
  void release(int64_t off, int64_t bytes) {
    bytes = round_up(std::max<int64_t>(bytes, 1), kAlign);
    free_list.push_back({off, bytes});
    std::sort(free_list.begin(), free_list.end(),
              [](const Block& a, const Block& b) { return a.off < b.off; });
    for (size_t i = 0; i + 1 < free_list.size();) {
      if (free_list[i].off + free_list[i].size == free_list[i + 1].off) {
        free_list[i].size += free_list[i + 1].size;
        free_list.erase(free_list.begin() + i + 1);
      } else {
        ++i;
      }
    }
  }
};

int64_t append64(std::vector<int64_t>& all, const std::vector<int64_t>& t) {
  while (all.size() % 2) all.push_back(0);   // 16-byte aligned
  int64_t off = (int64_t)all.size();
  all.insert(all.end(), t.begin(), t.end());
  return off;
}

int64_t append(std::vector<int32_t>& all, const std::vector<int32_t>& t) {
  // keep every table 16-byte aligned inside the device buffer
  while (all.size() % 4) all.push_back(0);
  int64_t off = (int64_t)all.size();
  all.insert(all.end(), t.begin(), t.end());
  return off;
}

}  // namespace

int build_plan(const ctn_plan_desc& d, Plan& P, std::string& err) {
  if (d.dtype != CTN_F32 && d.dtype != CTN_F64) { err = "dtype must be CTN_F32 or CTN_F64"; return CTN_INVALID_ARG; }
  if (d.n_inputs < 1 || !d.in_ndim || (!d.in_dims && !d.in_labels)) { err = "plan needs at least one input"; return CTN_INVALID_ARG; }
  if (d.n_steps < 1 || !d.step_lhs || !d.step_rhs || !d.step_out_ndim) { err = "plan needs at least one step"; return CTN_INVALID_ARG; }
  P.dtype = d.dtype;
  P.n_inputs = d.n_inputs;
  P.n_steps = d.n_steps;
  P.stabilize = (d.stabilize & 1) != 0;
  P.free_out_order = (d.stabilize & 2) != 0;
  P.min_norm = d.min_norm;
  const int64_t es = (int64_t)P.elem_size();
  const int64_t vec = 16 / es;  // elements per 16-byte vector load

  // ---- inputs ---------------------------------------------------------------
  std::map<int32_t, int64_t> label_ext;
  P.tensors.clear();
  P.tensors.reserve(d.n_inputs + d.n_steps);
  int64_t cur = 0;
  int64_t in_bytes = 0;
  P.input_offsets.clear();
  for (int i = 0; i < d.n_inputs; ++i) {
    Tensor t;
    t.is_input = true;
    int nd = d.in_ndim[i];
    if (nd < 0 || nd > 64) { err = fmt("input %lld has invalid ndim %lld", i, nd); return CTN_INVALID_ARG; }
    for (int a = 0; a < nd; ++a) {
      int64_t ext = d.in_dims[cur + a];
      int32_t lab = d.in_labels[cur + a];
      if (ext < 1) { err = fmt("input %lld axis %lld has extent %lld (must be >= 1)", i, a, ext); return CTN_SHAPE_MISMATCH; }
      auto it = label_ext.find(lab);
      if (it == label_ext.end()) label_ext[lab] = ext;
      else if (it->second != ext) {
        err = fmt("label %lld has extent %lld in one operand and %lld in another", lab, it->second, ext);
        return CTN_SHAPE_MISMATCH;
      }
      t.labels.push_back(lab);
      t.dims.push_back(ext);
      t.numel *= ext;
    }
    t.strides.assign(nd, 1);
    if (d.in_strides) {
      for (int a = 0; a < nd; ++a) t.strides[a] = d.in_strides[cur + a];
    } else {
      for (int a = nd - 2; a >= 0; --a) t.strides[a] = t.strides[a + 1] * t.dims[a + 1];
    }
    cur += nd;
    P.input_offsets.push_back(in_bytes);
    in_bytes += round_up(t.numel * es, kAlign);
    P.bytes_min += t.numel * es;
    P.tensors.push_back(std::move(t));
  }
  P.input_bytes_per_replica = in_bytes;

  // ---- fusion pre-pass -----------------------------------------------------------
  // An element-wise product (Khatri-Rao / Hadamard / broadcast: no label summed) whose result only feeds a GEMM
  // is formed on the fly as that GEMM's A operand (pattern A: CP decomposition, `ad,ac->acd` then `acd,ae->cde`);
  // a GEMM whose result is only re-weighted and summed by a small tensor (`bl,plr->bpr` then `bpr,bp->br`: an MPS
  // site applied to a batch of inputs, reference paper Fig. 1d) is the same triple product regrouped (pattern B:
  // (v (.) x) . A).  Either way the intermediate - 4 GiB for CP with r = n = 1024 - never exists.  fp32 only (the
  // KR loader lives in k_mfma_f32); CTN_FUSE=0 disables, 1 fuses whenever a pattern matches (tests), 2 likewise but
  // without pattern C below (so that pattern B can be tested on the networks C would take).
  // By default only intermediates of at least 2^33 elements (32 GiB in fp32; what a streaming step can still write) are fused away: the fused GEMM runs on the
  // register-staged kernel (a direct-to-LDS load cannot multiply) at 0.57 - 0.68 of the MFMA peak, where materialising
  // the product and feeding the large-tile kernel reaches 0.85 - 0.92 - CP, r = n = 1024: 22.5 vs 19.6 ms end to end;
  // CP-wide, r = 4096 (a 16 GiB product): 100.7 vs 66.6 ms (round 4; the threshold was 2^28 elements until then).  On a
  // 288 GB part workspace of that size is not what to save time against; beyond it fusing trades time for memory.
  // (A product of 2^31 elements and more is laid out for its consumer - the labels it sums NOT outermost, see below.)
  constexpr double kFuseMinNumel = 8589934592.0;
  struct Fuse { int x = -1, y = -1, w = -1; int epw = 0; int32_t pl = -1; };
  std::vector<Fuse> fuse(d.n_steps);
  std::vector<char> absorbed(d.n_steps, 0);
  {
    const char* fe = getenv("CTN_FUSE");
    const int fmode = fe ? atoi(fe) : -1;
    if (P.dtype == CTN_F32 && fmode != 0) {
      const int nt = d.n_inputs + d.n_steps;
      std::vector<std::vector<int32_t>> labs(nt);
      std::vector<double> numel(nt, 1.0);
      {
        int64_t c = 0;
        for (int i = 0; i < d.n_inputs; ++i) {
          for (int a = 0; a < d.in_ndim[i]; ++a) labs[i].push_back(d.in_labels[c + a]);
          c += d.in_ndim[i];
        }
        c = 0;
        for (int s = 0; s < d.n_steps; ++s) {
          for (int a = 0; a < d.step_out_ndim[s]; ++a) labs[d.n_inputs + s].push_back(d.step_out_labels[c + a]);
          c += d.step_out_ndim[s];
        }
        for (int i = 0; i < nt; ++i)
          for (int32_t l : labs[i]) { auto it = label_ext.find(l); numel[i] *= it == label_ext.end() ? 1.0 : (double)it->second; }
      }
      auto has = [&](int id, int32_t l) { return std::find(labs[id].begin(), labs[id].end(), l) != labs[id].end(); };
      auto distinct = [&](int id) {
        for (size_t a = 0; a < labs[id].size(); ++a)
          for (size_t b = a + 1; b < labs[id].size(); ++b) if (labs[id][a] == labs[id][b]) return false;
        return true;
      };
      auto ext = [&](int32_t l) { auto it = label_ext.find(l); return it == label_ext.end() ? (int64_t)1 : it->second; };
      // is the last axis of network input `id` unit-stride?  (NULL strides = C-contiguous)
      auto in_unit_last = [&](int id) {
        if (!d.in_strides) return true;
        int64_t c = 0;
        for (int i = 0; i < id; ++i) c += d.in_ndim[i];
        return d.in_ndim[id] > 0 && d.in_strides[c + d.in_ndim[id] - 1] == 1;
      };
      std::vector<int> consumer(nt, -1);
      for (int s = 0; s < d.n_steps; ++s) {
        if (d.step_lhs[s] >= 0 && d.step_lhs[s] < nt) consumer[d.step_lhs[s]] = s;
        if (d.step_rhs[s] >= 0 && d.step_rhs[s] < nt) consumer[d.step_rhs[s]] = s;
      }
      // is (x (.) y) . w -> out a GEMM the tile kernel takes?  M from x/y, N from w only, K summed
      auto gemm_ok = [&](int x, int y, int w, int out) {
        int64_t M = 1, N = 1, K = 1;
        std::vector<int32_t> all;
        for (int id : {x, y, w}) for (int32_t l : labs[id]) if (std::find(all.begin(), all.end(), l) == all.end()) all.push_back(l);
        for (int32_t l : all) {
          const bool inA = has(x, l) || has(y, l), inB = has(w, l), inC = has(out, l);
          if (!inC) K *= ext(l);
          else if (inA && !inB) M *= ext(l);
          else if (inB && !inA) N *= ext(l);
        }
        return M >= 64 && N >= 32 && K >= 8 && M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31);
      };
      for (int s1 = 0; s1 + 1 < d.n_steps; ++s1) {
        const int t1 = d.n_inputs + s1, p = d.step_lhs[s1], q = d.step_rhs[s1];
        const int s2 = consumer[t1];
        if (q < 0 || p < 0 || s2 < 0 || absorbed[s1] || fuse[s1].x >= 0 || fuse[s2].x >= 0) continue;
        const int other = d.step_lhs[s2] == t1 ? d.step_rhs[s2] : d.step_lhs[s2];
        if (other < 0 || !distinct(p) || !distinct(q) || !distinct(other) || !distinct(t1)) continue;
        const int out2 = d.n_inputs + s2;
        bool summed1 = false;          // does s1 sum anything?
        for (int id : {p, q}) for (int32_t l : labs[id]) if (!has(t1, l)) summed1 = true;
        if (!summed1) {
          // pattern A: t1 = p (.) q element-wise, consumed by a GEMM with `other`
          bool k2 = false;
          for (int id : {t1, other}) for (int32_t l : labs[id]) if (!has(out2, l)) k2 = true;
          const bool big = numel[t1] >= kFuseMinNumel && numel[t1] >= 8.0 * (numel[p] + numel[q]);
          if (k2 && gemm_ok(p, q, other, out2) && (fmode >= 1 || big)) {
            // X = the factor with the larger stride pattern first is irrelevant: keep (p, q)
            fuse[s2].x = p; fuse[s2].y = q; fuse[s2].w = other; absorbed[s1] = 1;
          }
        } else {
          // pattern B: t1 = p . q (a GEMM), then re-weighted / summed by `other` whose labels all live in t1
          bool sub = true;
          for (int32_t l : labs[other]) if (!has(t1, l)) sub = false;
          if (!sub) continue;
          // pattern C first: `other` is a network input that sums ONE short label (extent 2 or 4, its own unit-stride
          // axis) which the GEMM keeps as a column label, while everything else it carries are row labels of the
          // GEMM: the GEMM runs as it is, with that label innermost among its columns, and the re-weighting and the
          // short sum happen on the accumulator tile in the epilogue - X = E . A never reaches memory and the
          // streaming step is gone (batched MPS site, B = 4096, D = 256: 30.6 + 10.9 us -> see DESIGN.md)
          if (fmode != 2) {
            int32_t pl = -1;
            int nsum = 0;
            for (int32_t l : labs[t1]) if (!has(out2, l)) { ++nsum; pl = l; }
            bool okc = nsum == 1 && has(other, pl) && (ext(pl) == 2 || ext(pl) == 4) && other < d.n_inputs &&
                       !labs[other].empty() && labs[other].back() == pl && in_unit_last(other);
            int opA = -1, opB = -1;
            if (okc) {
              if (has(q, pl) && !has(p, pl)) { opA = p; opB = q; }
              else if (has(p, pl) && !has(q, pl)) { opA = q; opB = p; }
              else okc = false;
            }
            if (okc)
              for (int32_t l : labs[other])
                if (l != pl && !(has(opA, l) && !has(opB, l) && has(out2, l))) okc = false;
            if (okc) {
              // the GEMM itself: rows from opA only, columns from opB only (pl among them), something summed
              int64_t M = 1, N = 1, K = 1, Bt = 1;
              std::vector<int32_t> all;
              for (int id : {opA, opB}) for (int32_t l : labs[id]) if (std::find(all.begin(), all.end(), l) == all.end()) all.push_back(l);
              for (int32_t l : all) {
                const bool inA = has(opA, l), inB = has(opB, l), inC = has(t1, l);
                if (!inC) K *= ext(l);
                else if (inA && inB) Bt *= ext(l);
                else if (inA) M *= ext(l);
                else N *= ext(l);
              }
              okc = M >= 64 && N >= 32 && K >= 8 && M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31) &&
                    (double)Bt * (double)M * (double)N >= 65536.0;
            }
            if (okc) {
              fuse[s2].x = opA; fuse[s2].y = other; fuse[s2].w = opB; fuse[s2].epw = (int)ext(pl); fuse[s2].pl = pl;
              absorbed[s1] = 1;
              continue;
            }
          }
          const bool big = numel[t1] >= kFuseMinNumel && numel[t1] >= 4.0 * numel[out2] && numel[other] * 16.0 <= numel[t1];
          if (!(fmode >= 1 || big)) continue;
          // `other` joins the side that carries more of its kept labels
          int kp = 0, kq = 0;
          for (int32_t l : labs[other]) if (has(out2, l)) { kp += has(p, l); kq += has(q, l); }
          int x = kp >= kq ? p : q, w = kp >= kq ? q : p;
          if (!gemm_ok(x, other, w, out2)) { std::swap(x, w); if (!gemm_ok(x, other, w, out2)) continue; }
          fuse[s2].x = x; fuse[s2].y = other; fuse[s2].w = w; absorbed[s1] = 1;
        }
      }
    }
  }

  // ---- steps ----------------------------------------------------------------
  std::vector<int> consumed(d.n_inputs + d.n_steps, 0);
  Arena arena;
  P.steps.clear();
  P.tables.clear();
  int64_t out_cur = 0;
  for (int s = 0; s < d.n_steps; ++s) {
    Step st;
    const int out_id = d.n_inputs + s;
    int lhs = d.step_lhs[s], rhs = d.step_rhs[s];
    const bool last = s == d.n_steps - 1 && !P.free_out_order;   // (free order: the last result is laid out like any other)
    if (lhs < 0 || lhs >= out_id || rhs < -1 || rhs >= out_id || lhs == rhs) {
      err = fmt("step %lld references invalid operands (%lld, %lld)", s, lhs, rhs);
      return CTN_INVALID_ARG;
    }
    if (consumed[lhs] || (rhs >= 0 && consumed[rhs])) {
      err = fmt("step %lld consumes a tensor that was already contracted away", s);
      return CTN_INVALID_ARG;
    }
    consumed[lhs] = 1;
    if (rhs >= 0) consumed[rhs] = 1;

    const int ond = d.step_out_ndim[s];
    std::vector<int32_t> out_labels(d.step_out_labels + out_cur, d.step_out_labels + out_cur + ond);
    out_cur += ond;

    if (absorbed[s]) {
      // this step's product is formed on the fly inside the step that consumes it: no kernel, no buffer; its
      // operands stay alive until that step has read them
      Tensor vt;
      vt.producer = s;
      vt.labels = out_labels;
      for (int32_t lab : out_labels) {
        auto it = label_ext.find(lab);
        if (it == label_ext.end()) { err = fmt("step %lld: output label %lld is in neither operand", s, lab); return CTN_INVALID_ARG; }
        vt.dims.push_back(it->second);
        vt.numel *= it->second;
      }
      vt.strides.assign(vt.labels.size(), 0);
      st.lhs = lhs; st.rhs = rhs; st.out = out_id;
      st.kernel = CTN_KERNEL_FUSED;
      st.Bt = st.M = st.N = st.K = 0;
      st.blocks = 0; st.partials = 0; st.flops = 0; st.chain_ok = false;
      P.tensors.push_back(std::move(vt));
      P.steps.push_back(std::move(st));
      continue;
    }
    const int epw = fuse[s].epw;                          // pattern C: re-weighted short sum in the epilogue
    const bool fused = fuse[s].x >= 0 && !epw;            // patterns A / B: A = X (.) Y (Y = fuse[s].y), B = W
    if (fuse[s].x >= 0) { lhs = fuse[s].x; rhs = fuse[s].w; }

    // -- label census
    std::vector<LabelInfo> info;
    auto find = [&](int32_t lab) -> LabelInfo* {
      for (auto& l : info) if (l.label == lab) return &l;
      return nullptr;
    };
    auto scan = [&](const Tensor& T, int which /*0 A (X), 1 B, 2 second A-side tensor Y, 3 epilogue weights*/) {
      for (size_t a = 0; a < T.labels.size(); ++a) {
        LabelInfo* l = find(T.labels[a]);
        if (!l) {
          info.emplace_back();
          l = &info.back();
          l->label = T.labels[a];
          l->ext = T.dims[a];
          l->firstPos = (int)info.size();
        }
        (which == 0 ? l->sA : which == 1 ? l->sB : l->sA2) += T.strides[a];  // repeated label = diagonal: strides add
        if (which == 3) { l->inA2 = true; continue; }   // the epilogue's weight tensor: strides only, no class
        if (which == 1) l->inB = true; else l->inA = true;
        if (which == 0) l->inX = true;
        if (which == 2) l->inA2 = true;
      }
    };
    scan(P.tensors[lhs], 0);
    if (fused) scan(P.tensors[fuse[s].y], 2);
    if (rhs >= 0) scan(P.tensors[rhs], 1);
    if (epw) scan(P.tensors[fuse[s].y], 3);
    for (size_t a = 0; a < out_labels.size(); ++a) {
      LabelInfo* l = find(out_labels[a]);
      if (!l) { err = fmt("step %lld: output label %lld is in neither operand", s, out_labels[a]); return CTN_INVALID_ARG; }
      if (l->inC) { err = fmt("step %lld: output label %lld is repeated", s, out_labels[a]); return CTN_INVALID_ARG; }
      l->inC = true;
    }

    if (epw) find(fuse[s].pl)->inC = true;   // a column label of the GEMM (summed later, in the epilogue; sC stays 0)

    // -- operand swap: the output's unit-stride label should be a column (N) label
    bool swap = false;
    if (rhs >= 0 && !fused && !epw) {   // (the on-the-fly product can only be the A operand; pattern C fixed the sides)
      if (last) {
        if (!out_labels.empty()) {
          LabelInfo* l = find(out_labels.back());
          if (l->inA && !l->inB) swap = true;
        }
      } else {
        bool anyM = false, anyN = false;
        for (auto& l : info) if (l.inC) { if (l.inA && !l.inB) anyM = true; if (l.inB && !l.inA) anyN = true; }
        if (anyM && !anyN) swap = true;
        // A very wide operand against a small one over K = 256 (a boundary tensor of a 2D grid absorbing a site: 2^20 x 256
        // x 256): the small one goes LEFT - the engine then keeps it in registers while the wide one streams
        // (k_mfma_f32_ares), and reads it through its tables whatever its layout, where a right operand that is
        // unit-stride along neither n nor k sends the whole step to the 4-byte-gather kernel.
        bool small_left = false, keep_sides = false;
        if (P.dtype == CTN_F32 && anyM && anyN) {
          int64_t eM = 1, eN = 1, eK = 1;
          for (auto& l : info) {
            if (!l.inC) eK *= l.ext;
            else if (l.inA && !l.inB) eM *= l.ext;
            else if (l.inB && !l.inA) eN *= l.ext;
          }
          small_left = eK == 256 && eN == 256 && eM >= 32768;
        }
        // A result of 2^31 elements or more is laid out [batch][M][N]; the step that consumes it can address a
        // CONTRACTED group only through a 32-bit table, so the labels it sums must be the inner (column) ones: when
        // they all come from the left operand, the operands change sides.
        double out_n = 1.0;
        for (auto& l : info) if (l.inC) out_n *= (double)l.ext;
        if (out_n >= 2147483648.0 && anyM && anyN) {
          int s2 = -1;
          for (int q = s + 1; q < d.n_steps && s2 < 0; ++q) if (d.step_lhs[q] == out_id || d.step_rhs[q] == out_id) s2 = q;
          if (s2 >= 0) {
            int64_t o2 = 0;
            for (int q = 0; q < s2; ++q) o2 += d.step_out_ndim[q];
            auto kept = [&](int32_t lab) {
              for (int a_ = 0; a_ < d.step_out_ndim[s2]; ++a_) if (d.step_out_labels[o2 + a_] == lab) return true;
              return false;
            };
            bool sumM = false, sumN = false;
            for (auto& l : info) if (l.inC && !kept(l.label)) { if (l.inA && !l.inB) sumM = true; if (l.inB && !l.inA) sumN = true; }
            if (sumM && !sumN) swap = true;
            if (sumN && !sumM) keep_sides = true;
          }
        }
        if (small_left && !keep_sides) swap = true;
      }
    }
    if (swap) {
      std::swap(lhs, rhs);
      for (auto& l : info) { std::swap(l.sA, l.sB); std::swap(l.inA, l.inB); }
    }
    st.lhs = lhs; st.rhs = rhs; st.out = out_id; st.swapped = swap;
    st.lhs2 = fused || epw ? fuse[s].y : -1;
    st.epw = epw;

    // -- classify and order
    std::vector<LabelInfo*> G[4];
    for (auto& l : info) {
      l.cls = l.inC ? (l.inA && l.inB ? kBatch : (l.inA ? kM : kN)) : kK;
      G[l.cls].push_back(&l);
    }
    auto by = [](bool useA) {
      return [useA](const LabelInfo* x, const LabelInfo* y) {
        // labels absent from the ordering operand (summed out of the other one) go outermost
        int64_t sx = useA ? (x->inA ? std::max(x->sA, x->sA2) : INT64_MAX) : (x->inB ? x->sB : INT64_MAX);
        int64_t sy = useA ? (y->inA ? std::max(y->sA, y->sA2) : INT64_MAX) : (y->inB ? y->sB : INT64_MAX);
        if (sx != sy) return sx > sy;  // descending stride: last label is the most contiguous
        return x->firstPos < y->firstPos;
      };
    };
    std::stable_sort(G[kBatch].begin(), G[kBatch].end(), by(true));
    std::stable_sort(G[kM].begin(), G[kM].end(), by(true));
    std::stable_sort(G[kN].begin(), G[kN].end(), by(false));
    if (epw) {   // the label summed in the epilogue is the innermost column label: four adjacent columns = its values
      auto it = std::find(G[kN].begin(), G[kN].end(), find(fuse[s].pl));
      LabelInfo* pl = *it;
      G[kN].erase(it);
      G[kN].push_back(pl);
    }
    bool aUnitInK = false, bUnitInK = false;
    for (auto* l : G[kK]) { if (l->inA && l->sA == 1) aUnitInK = true; if (l->inB && l->sB == 1) bUnitInK = true; }
    // k order = memory order of the operand that is unit-stride along a contracted label; when both are, but along
    // DIFFERENT labels, the larger operand decides (the other one falls back to 4-byte gathers: cheap for a 64 x 64
    // boundary tensor, 1.4x the time of the whole step for the 2-million-element one)
    bool kOrderA = !(bUnitInK && !aUnitInK);
    if (aUnitInK && bUnitInK && rhs >= 0 && P.tensors[rhs].numel > P.tensors[lhs].numel) kOrderA = false;
    std::stable_sort(G[kK].begin(), G[kK].end(), by(kOrderA));

    // A result of 2^31 elements or more (an 8 x 8 grid at bond 16 sliced over two labels only: 2^32-element boundary
    // tensors): the row / column tables hold 32-bit offsets, so free labels have to become batch labels (64-bit offsets)
    // until one batch entry's matrix is below 2^31 elements.  Which ones is decided HERE, before the layout: the outer
    // labels of the LARGER free group go, and as batch labels they are the outermost of the result - the small group
    // stays whole (256 rows against 2^24 columns keep their 256 rows: large tiles, the resident-operand kernel; taken
    // from the result's own outermost label afterwards, as the span rule below would, they became 16 rows).
    if (!last && !fused && !epw && rhs >= 0 && !G[kK].empty()) {   // (element-wise products: their own rule below)
      auto ext_of = [](const std::vector<LabelInfo*>& g) { double n = 1.0; for (auto* l : g) n *= (double)l->ext; return n; };
      double out_n = 1.0;
      for (auto& l : info) if (l.inC) out_n *= (double)l.ext;
      if (out_n >= 2147483648.0) {
        // ... and of those, labels the consuming step KEEPS: what it sums has to stay inside one batch entry's matrix
        // (a contracted group is addressed through a 32-bit table)
        int s2 = -1;
        for (int q = s + 1; q < d.n_steps && s2 < 0; ++q) if (d.step_lhs[q] == out_id || d.step_rhs[q] == out_id) s2 = q;
        int64_t o2 = 0;
        for (int q = 0; q < s2; ++q) o2 += d.step_out_ndim[q];
        auto kept = [&](int32_t lab) {
          if (s2 < 0) return true;
          for (int a_ = 0; a_ < d.step_out_ndim[s2]; ++a_) if (d.step_out_labels[o2 + a_] == lab) return true;
          return false;
        };
        while (ext_of(G[kM]) * ext_of(G[kN]) >= 2147483648.0) {
          const int cls = ext_of(G[kN]) >= ext_of(G[kM]) ? kN : kM;
          if (G[cls].size() <= 1) break;
          auto it = std::find_if(G[cls].begin(), G[cls].end(), [&](LabelInfo* l) { return kept(l->label); });
          if (it == G[cls].end()) it = G[cls].begin();
          G[kBatch].push_back(*it);
          G[cls].erase(it);
        }
      }
    }

    // -- output tensor and its layout
    Tensor out;
    out.producer = s;
    if (last) {
      out.labels = out_labels;
    } else {
      for (int c : {kBatch, kM, kN}) for (auto* l : G[c]) if (!epw || l->label != fuse[s].pl) out.labels.push_back(l->label);
      // An element-wise product (no label summed: a Khatri-Rao product, always a streaming step, which writes any
      // label order) of 2^31 elements or more: the step that consumes it can address what it SUMS only through 32-bit
      // tables, and a label both operands share - the batch group, outermost by default - is exactly what a CP
      // contraction sums next (`ad,ac->acd` then `acd,ae->cde`, r = 4096: a spans all 2^32 elements).  Order the result
      // as (labels the consumer keeps)(labels it sums)(the unit-stride label): the kept outer labels then become batch
      // labels of the consumer (64-bit offsets) and its contracted group spans one small matrix.
      if (G[kK].empty() && out.labels.size() >= 3) {
        double out_n = 1.0;
        for (int32_t lab : out.labels) out_n *= (double)find(lab)->ext;
        int s2 = -1;
        for (int q = s + 1; q < d.n_steps && s2 < 0; ++q) if (d.step_lhs[q] == out_id || d.step_rhs[q] == out_id) s2 = q;
        if (out_n >= 2147483648.0 && s2 >= 0) {
          int64_t o2 = 0;
          for (int q = 0; q < s2; ++q) o2 += d.step_out_ndim[q];
          auto kept = [&](int32_t lab) {
            for (int a_ = 0; a_ < d.step_out_ndim[s2]; ++a_) if (d.step_out_labels[o2 + a_] == lab) return true;
            return false;
          };
          const int32_t unit = out.labels.back();
          if (kept(unit)) {
            std::vector<int32_t> order;
            for (size_t a_ = 0; a_ + 1 < out.labels.size(); ++a_) if (kept(out.labels[a_])) order.push_back(out.labels[a_]);
            for (size_t a_ = 0; a_ + 1 < out.labels.size(); ++a_) if (!kept(out.labels[a_])) order.push_back(out.labels[a_]);
            order.push_back(unit);
            out.labels = order;
          }
        }
      }
    }
    for (int32_t lab : out.labels) out.dims.push_back(find(lab)->ext);
    out.strides.assign(out.labels.size(), 1);
    for (int a = (int)out.labels.size() - 2; a >= 0; --a) out.strides[a] = out.strides[a + 1] * out.dims[a + 1];
    for (size_t a = 0; a < out.labels.size(); ++a) { out.numel *= out.dims[a]; find(out.labels[a])->sC = out.strides[a]; }

    // Epilogue-summed steps: with the summed label p innermost, four adjacent columns are the four p of ONE r - the B
    // operand is then gathered four bytes at a time and so is the output.  When the next column label u is unit-stride
    // in B (r of an MPS core), split it as u = 4 u_hi + u_lo and order the columns (..., u_hi, p, u_lo): a lane's
    // four columns are four consecutive u of one p (16-byte loads of B, 16-byte stores of C), and the sum over p
    // runs across 2 / 4 adjacent lanes.  Synthetic labels live in `synth`; `split_parent` is the label they replace.
    LabelInfo synth[2];
    const LabelInfo* split_parent = nullptr;
    if (epw && G[kN].size() >= 2) {
      LabelInfo* u = G[kN][G[kN].size() - 2];
      if (u->inB && u->sB == 1 && u->ext % vec == 0 && vec == 4) {
        synth[0] = *u; synth[0].ext = u->ext / 4; synth[0].sB = 4 * u->sB; synth[0].sC = 4 * u->sC;
        synth[1] = *u; synth[1].ext = 4;
        LabelInfo* pl = G[kN].back();
        G[kN].pop_back(); G[kN].pop_back();
        G[kN].push_back(&synth[0]); G[kN].push_back(pl); G[kN].push_back(&synth[1]);
        split_parent = u;
        st.epw_split = true;
      }
    }

    // Tensors of 2^31 elements and more: the m / n / k tables hold 32-bit offsets, the batch tables 64-bit ones, so
    // outer free labels (largest strides first) move into the batch group until what is left of the row and column
    // groups spans less than 2^31 elements of every tensor that carries it.  A free label of one operand is a perfectly
    // good batch label - the other operand simply has stride 0 along it - so the arithmetic is unchanged; only the
    // tiling sees smaller matrices.  Contracted labels cannot move: a K group spanning 2^31 elements is refused.
    {
      const int64_t lim = (1LL << 31) - 1;
      auto promote = [&](int cls, int wa, int wb) {
        while (!G[cls].empty() && (span_of(G[cls], wa) > lim || span_of(G[cls], wb) > lim)) {
          if (fused || epw) return false;                  // (their extra tables are not re-derived: refuse)
          G[kBatch].push_back(G[cls].front());
          G[cls].erase(G[cls].begin());
        }
        return true;
      };
      if (!promote(kM, 0, 2) || !promote(kN, 1, 2) || span_of(G[kK], 0) > lim || span_of(G[kK], 1) > lim) {
        err = fmt("step %lld: an index group spans 2^31 or more elements of one operand and cannot be split off as a batch", s);
        return CTN_UNSUPPORTED;
      }
    }

    auto extent = [](const std::vector<LabelInfo*>& g) { int64_t n = 1; for (auto* l : g) n *= l->ext; return n; };
    st.Bt = extent(G[kBatch]); st.M = extent(G[kM]); st.N = extent(G[kN]); st.K = extent(G[kK]);
    st.has_k = !G[kK].empty();
    if (st.Bt >= (1LL << 31) || st.M >= (1LL << 31) || st.N >= (1LL << 31) || st.K >= (1LL << 31)) {
      err = "index group with >= 2^31 entries is not supported"; return CTN_UNSUPPORTED;
    }

    // -- vector-load modes: 1 = unit stride on the free index, 2 = unit stride on k
    auto mode_of = [&](bool isA) -> int {
      auto stride = [&](const LabelInfo* l) { return isA ? l->sA : l->sB; };
      auto present = [&](const LabelInfo* l) { return isA ? l->inA : l->inB; };
      const auto& freeG = isA ? G[kM] : G[kN];
      auto others_ok = [&](const LabelInfo* unit) {
        for (auto& l : info) if (&l != unit && &l != split_parent && present(&l) && stride(&l) % vec != 0) return false;
        return true;
      };
      if (!freeG.empty()) {
        const LabelInfo* l = freeG.back();
        if (stride(l) == 1 && l->ext % vec == 0 && others_ok(l)) return 1;
      }
      if (!G[kK].empty()) {
        const LabelInfo* l = G[kK].back();
        if (present(l) && stride(l) == 1 && l->ext % vec == 0 && others_ok(l)) return 2;
      }
      return 0;
    };
    st.modeA = fused ? 3 : mode_of(true);   // 3 = element-wise product of two tensors formed while staging (KR loader)
    if (fused) {
      // 16-byte accesses where the factors allow it: along 4 consecutive rows (mode 4) or 4 consecutive k (mode 5) a
      // factor is contiguous (1), constant (2: it does not carry the innermost label there) or gathered (0)
      auto kind_along = [&](const LabelInfo* l, bool isX) -> int {
        const bool present = isX ? l->inX : l->inA2;
        const int64_t stride = isX ? l->sA : l->sA2;
        if (!present || stride == 0) return 2;
        if (stride != 1) return 0;
        for (auto& o : info)
          if (&o != l && (isX ? o.inX : o.inA2) && (isX ? o.sA : o.sA2) % vec != 0) return 0;
        return 1;
      };
      for (int dir = 1; dir <= 2 && st.modeA == 3; ++dir) {
        const auto& grp = dir == 1 ? G[kM] : G[kK];
        if (grp.empty() || grp.back()->ext % vec != 0) continue;
        const int kx = kind_along(grp.back(), true), ky = kind_along(grp.back(), false);
        if (kx == 1 || ky == 1) { st.modeA = 3 + dir; st.krX = kx; st.krY = ky; }
      }
    }
    st.modeB = rhs >= 0 ? mode_of(false) : 0;
    if (!G[kN].empty()) {
      const LabelInfo* u = G[kN].back();
      bool ok = u->sC == 1 && u->ext % vec == 0;
      for (auto& l : info) if (&l != u && &l != split_parent && l.inC && l.sC % vec != 0) ok = false;
      st.cvec = ok;
    }

    // -- kernel choice
    const int64_t outs = st.Bt * st.M * st.N;
    // tile kernels: both free extents >= 32, or one long (>= 128) and the other at least 8 wide - a
    // mostly-masked MFMA tile still beats the streaming kernels by an order of magnitude there
    // ... and so does a huge K against a few rows and columns (8 x 8 x 4,194,304: one mostly-masked 64 x 64 tile
    // per K split streams both operands once - 64 separate dot products re-read them 8 times each)
    // ... and a long side against only 4 - 7 rows or columns when K is long too (the step that closes an MPS overlap,
    // 256 x 4 x 256: its 1024 outputs are ONE workgroup of the streaming kernel walking K = 256 a dependent round trip
    // at a time - 170 us, a tenth of the whole 100-site contraction with one network in flight; a masked 16 x 16 tile per
    // workgroup with K split over its waves: a few us)
    const bool tileable = st.K >= 8 && ((st.M >= 32 && st.N >= 32) || (st.M >= 128 && st.N >= 8) ||
                                        (st.N >= 128 && st.M >= 8) ||
                                        (st.K >= 128 && ((st.M >= 128 && st.N >= 4) || (st.N >= 128 && st.M >= 4))) ||
                                        (st.K >= 32768 && st.M >= 4 && st.N >= 4 && st.M * st.N >= 32));
    if (P.dtype == CTN_F32 && tileable) {
      st.kernel = CTN_KERNEL_MFMA_F32;
      // 64-wide column tiles when they cover N with less padding (e.g. N = 64, 192, 320)
      st.tileN = ((st.N + 63) / 64) * 64 < ((st.N + kTileN - 1) / kTileN) * kTileN ? 64 : kTileN;
      // ... and 64-high row tiles likewise (M = 64 against a huge N: a PEPS boundary absorption seen from the
      // other side; with 128-row tiles half of every MFMA and of every A load is padding)
      st.tileM = ((st.M + 63) / 64) * 64 < ((st.M + kTileM - 1) / kTileM) * kTileM ? 64 : kTileM;
      if (st.Bt * ((st.M + 63) / 64) * ((st.N + 63) / 64) >= (1LL << 31)) { err = "step with 2^31 or more tiles"; return CTN_UNSUPPORTED; }
      st.blocks = (int)(st.Bt * ((st.M + st.tileM - 1) / st.tileM) * ((st.N + st.tileN - 1) / st.tileN));
      // 256 x 128 tiles fed by LDS-DMA (kernels_mfma_g.h): each operand unit-stride along its free index
      // or along k (16-byte requests either way; not the general gather), at least two 16-deep k-tiles,
      // and M, N such that 256-row
      // tiles pad at most 15 % more than 128-row ones (ragged edges are masked in the epilogue).
      // blocks / partial slots stay counted in 128 x 128 units.
      const int64_t pad128 = round_up(st.M, kTileM) * round_up(st.N, kTileN);
      const int64_t pad256 = round_up(st.M, 256) * round_up(st.N, kTileN);
      if (!fused && !epw && st.modeA >= 1 && st.modeB >= 1 && st.tileN == kTileN && st.cvec && st.K >= 32 &&
          st.M > kTileM && pad256 * 100 <= pad128 * 115 &&
          st.rhs >= 0 && span_of(G[kM], 0) + span_of(G[kK], 0) < (1LL << 30) &&
          span_of(G[kN], 1) + span_of(G[kK], 1) < (1LL << 30)) {  // 32-bit byte offsets inside one batch entry's matrices
        st.tileM = 256;
        st.blocks = (int)(st.Bt * ((st.M + kTileM - 1) / kTileM) * ((st.N + st.tileN - 1) / st.tileN));
      }
    } else if (kEnableMfmaF64 && P.dtype == CTN_F64 && tileable) {
      st.kernel = CTN_KERNEL_MFMA_F64;
      st.tileN = kTile64N;
      st.blocks = (int)(st.Bt * ((st.M + kTile64M - 1) / kTile64M) * ((st.N + kTile64N - 1) / kTile64N));
      // 128 x 128 tiles fed by LDS-DMA (kernels_mfma_g64.h): same rule as the fp32 large-tile kernel
      // (two 8-deep k-tiles at least; blocks / partial slots stay counted in 128 x 64 units)
      if (st.modeA >= 1 && st.modeB >= 1 && st.K >= 16 && st.N > kTile64N &&
          round_up(st.N, 128) * 100 <= round_up(st.N, kTile64N) * 115 && st.rhs >= 0 &&
          P.tensors[st.lhs].numel <= (1LL << 29) && P.tensors[st.rhs].numel <= (1LL << 29))  // 32-bit byte offsets
        st.tileN = 128;
    } else if (outs <= kWaveOutputs && st.K >= 512) {
      st.kernel = CTN_KERNEL_DOT;
      st.blocks = (int)outs;
    } else {
      // streaming decomposition: n = C's unit-stride label, the other C labels split into (hi, lo)
      std::vector<const LabelInfo*> cl;
      for (int32_t lab : out.labels) cl.push_back(find(lab));
      const LabelInfo* nl = nullptr;
      if (!cl.empty()) { nl = cl.back(); cl.pop_back(); }
      st.Nv = nl ? nl->ext : 1;
      st.sAn = nl && nl->inA ? nl->sA : 0;
      st.sBn = nl && nl->inB ? nl->sB : 0;
      std::vector<const LabelInfo*> lo, hi;
      int64_t lprod = 1;
      int64_t spanLo[3] = {0, 0, 0};     // 32-bit tables: what the inner group spans in A, B, C must stay below 2^31
      while (!cl.empty() && (lo.empty() || lprod * cl.back()->ext <= 65536)) {
        const LabelInfo* c_ = cl.back();
        const int64_t add[3] = {(c_->ext - 1) * c_->sA, (c_->ext - 1) * c_->sB, (c_->ext - 1) * c_->sC};
        if (spanLo[0] + add[0] >= (1LL << 31) || spanLo[1] + add[1] >= (1LL << 31) || spanLo[2] + add[2] >= (1LL << 31)) break;
        for (int q_ = 0; q_ < 3; ++q_) spanLo[q_] += add[q_];
        lprod *= c_->ext;
        lo.insert(lo.begin(), c_);
        cl.pop_back();
      }
      hi = cl;
      st.L = lprod;
      st.H = 1;
      for (auto* l : hi) st.H *= l->ext;
      if (nl && (std::max<int64_t>(st.sAn, st.sBn) >= (1LL << 31) || span_of(G[kK], 0) >= (1LL << 31) ||
                 span_of(G[kK], 1) >= (1LL << 31) || st.H * lprod * nl->ext >= (1LL << 33))) {
        err = fmt("step %lld: a streaming step this large (an inner / contracted group spanning 2^31 elements, or 2^33 outputs) is not supported", s);
        return CTN_UNSUPPORTED;
      }
      std::vector<int32_t> t6;
      std::vector<int64_t> t8;
      build_table64(hi, 0, st.H, t8); st.t.ohA = append64(P.tables64, t8);
      build_table64(hi, 1, st.H, t8); st.t.ohB = append64(P.tables64, t8);
      build_table64(hi, 2, st.H, t8); st.t.ohC = append64(P.tables64, t8);
      build_table(lo, 0, st.L, t6); st.t.olA = append(P.tables, t6);
      build_table(lo, 1, st.L, t6); st.t.olB = append(P.tables, t6);
      build_table(lo, 2, st.L, t6); st.t.olC = append(P.tables, t6);
      // 16-byte vectors along n: C always, an operand when it is unit-stride there (then everything of it
      // must be vector-aligned); a broadcast operand is one scalar, any other stride a gather of V scalars
      bool vok = nl && st.Nv % vec == 0;
      for (auto& l : info) {
        if (&l == nl) continue;
        if (l.inC && l.sC % vec != 0) vok = false;
        if (st.sAn == 1 && l.inA && l.sA % vec != 0) vok = false;
        if (st.sBn == 1 && l.inB && l.sB % vec != 0) vok = false;
      }
      st.vecw = vok ? (int)vec : 1;
      // a long sum over a unit-stride K of the larger operand: lanes along k instead
      const bool aBig = rhs < 0 || P.tensors[lhs].numel >= P.tensors[rhs].numel;
      const bool kContig = aBig ? st.modeA == 2 : st.modeB == 2;
      bool kUnit = kContig;
      if (!kUnit && !G[kK].empty()) {  // mode flags need vector alignment; unit stride alone is enough here
        const LabelInfo* kl = G[kK].back();
        kUnit = aBig ? (kl->inA && kl->sA == 1) : (kl->inB && kl->sB == 1);
      }
      // a SHORT unit-stride K of the left operand under a strided output index (`abk,k->ab`: a vector applied to the
      // innermost leg): one output per thread, the operand's K elements as 16-byte loads (k_stream_kvec) - with
      // vectors along n instead every element is its own 4-byte gather
      bool kv = rhs >= 0 && G[kK].size() == 1 && st.K >= (int64_t)vec && st.K <= 64 && st.K % vec == 0 && st.sAn != 1;
      if (kv) {
        const LabelInfo* kl = G[kK][0];
        if (!(kl->inA && kl->sA == 1 && kl->inB)) kv = false;
        for (auto& l : info) if (&l != kl && l.inA && l.sA % vec != 0) kv = false;
      }
      if (kv && outs >= (1LL << 31)) kv = false;
      if (kv) {
        st.kernel = CTN_KERNEL_ELEMENT;
        st.kvec = 1;
        st.vecw = 1;
        st.blocks = stream_grid((st.H * st.L * st.Nv + 255) / 256);
      } else if (st.K >= 256 && kUnit && outs < (1LL << 31)) {
        st.kernel = CTN_KERNEL_ROWDOT;
        // persistent-style grid: at most 16 workgroups per CU, waves stride over the outputs
        st.blocks = (int)std::min<int64_t>((outs + 3) / 4, kStreamMaxBlocks);
      } else {
        st.kernel = CTN_KERNEL_ELEMENT;
        const int64_t items = st.H * st.L * ((st.Nv + st.vecw - 1) / st.vecw);
        if (items >= (1LL << 31)) { err = fmt("step %lld: a streaming step with 2^31 or more work items is not supported", s); return CTN_UNSUPPORTED; }
        st.blocks = stream_grid((items + 255) / 256);
      }
    }
    st.chain_ok = outs <= kChainMaxOut && outs * st.K <= kChainMaxWork && !epw;
    st.collapse = st.blocks > kMaxPartials;
    st.partials = st.collapse ? 1 : st.blocks;
    // (x2: the launcher may halve the column tile of an under-filled fp32 MFMA launch)
    if (st.collapse) P.max_collapse_blocks = std::max<int64_t>(P.max_collapse_blocks, 2 * (int64_t)st.blocks);
    st.flops = (st.has_k ? 2.0 : 1.0) * (double)st.Bt * (double)st.M * (double)st.N * (double)st.K +
               (P.stabilize ? 3.0 * (double)out.numel : 0.0) +
               (fused ? (double)st.Bt * (double)st.M * (double)st.K : 0.0) +   // the multiplies of the fused product
               (epw ? 2.0 * (double)st.Bt * (double)st.M * (double)st.N : 0.0);  // the re-weighting and the short sum
    P.flops += st.flops;

    // -- gather-offset tables
    std::vector<const LabelInfo*> gb(G[kBatch].begin(), G[kBatch].end()), gm(G[kM].begin(), G[kM].end()),
        gn(G[kN].begin(), G[kN].end()), gk(G[kK].begin(), G[kK].end());
    const int64_t padM = round_up(st.M, std::max(kTileM, st.tileM)), padN = round_up(st.N, kTileN), padK = round_up(st.K, kPadK) + 2 * kPadK;  // kernels prefetch table entries two tiles ahead
    std::vector<int32_t> tb;
    const bool tiled = st.kernel == CTN_KERNEL_MFMA_F32 || st.kernel == CTN_KERNEL_MFMA_F64 || st.kernel == CTN_KERNEL_DOT;
    std::vector<int64_t> tb8;
    if (tiled || st.chain_ok) {  // (batch, m, n) tables: tile kernels and the chain walker
      build_table64(gb, 0, st.Bt, tb8); st.t.obA = append64(P.tables64, tb8);
      build_table64(gb, 1, st.Bt, tb8); st.t.obB = append64(P.tables64, tb8);
      build_table64(gb, 2, st.Bt, tb8); st.t.obC = append64(P.tables64, tb8);
      build_table(gm, 0, padM, tb);  st.t.omA = append(P.tables, tb);
      build_table(gm, 2, padM, tb);  st.t.omC = append(P.tables, tb);
      build_table(gn, 1, padN, tb);  st.t.onB = append(P.tables, tb);
      build_table(gn, 2, padN, tb);  st.t.onC = append(P.tables, tb);
    }
    build_table(gk, 0, padK, tb);  st.t.okA = append(P.tables, tb);
    build_table(gk, 1, padK, tb);  st.t.okB = append(P.tables, tb);
    if (fused) {   // the second tensor of the A side (a label it does not carry has stride 0)
      build_table64(gb, 3, st.Bt, tb8); st.t.obA2 = append64(P.tables64, tb8);
      build_table(gm, 3, padM, tb);  st.t.omA2 = append(P.tables, tb);
      build_table(gk, 3, padK, tb);  st.t.okA2 = append(P.tables, tb);
    }
    if (epw) {     // the epilogue's weights: offset of every row (their short label is unit-stride: + p)
      build_table(gm, 3, padM, tb);  st.t.omA2 = append(P.tables, tb);
    }

    // -- workspace: allocate the output, then release the consumed intermediates
    if (s != d.n_steps - 1) out.ws_offset = arena.alloc(out.numel * es);   // (the final result is the caller's buffer)
    for (int id : {lhs, rhs, st.lhs2}) {
      if (id >= d.n_inputs) arena.release(P.tensors[id].ws_offset, P.tensors[id].numel * es);
    }
    P.tensors.push_back(std::move(out));
    P.steps.push_back(std::move(st));
  }
  for (int id = 0; id < d.n_inputs + d.n_steps - 1; ++id) {
    if (!consumed[id]) { err = fmt("tensor %lld is never contracted: the path must reduce the network to one tensor", id); return CTN_INVALID_ARG; }
  }
  // latency-bound DAGs (e.g. 1000 dependent 3x3 products): no per-step launch at all
  P.chain = d.n_steps >= 4 && d.n_steps <= kChainMaxSteps;
  for (const Step& st : P.steps)
    if (!st.chain_ok) P.chain = false;
  if (P.chain)
    for (Step& st : P.steps) { st.partials = 1; st.collapse = false; }
  P.ws_bytes_per_replica = arena.top;
  P.bytes_min += P.output().numel * es;
  return CTN_OK;
}

}  // namespace ctn
