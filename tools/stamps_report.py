"""Summarise the per-tile cycle stamps of a `make STAMPS=1` build (CTN_DEBUG_STAMPS=<file>).
Slots per tile: 0 start, 1 first k-tile in LDS, 2 main loop done, 3 end, 4/5 epilogue halves (wave 0),
6 wave 3 epilogue done, 7 HW_ID | XCC_ID << 32."""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
hw = a[:, 7].copy()
a = a.astype(np.int64)
ok = (a[:, :7] > 0).all(axis=1)
a, hw = a[ok], hw[ok]
pro = a[:, 1] - a[:, 0]; main = a[:, 2] - a[:, 1]; epi = a[:, 3] - a[:, 2]; tot = a[:, 3] - a[:, 0]
span = a[:, 3].max() - a[:, 0].min()
med = lambda x: int(np.median(x))
print("tiles", len(a), "ticks: prologue med %d  mainloop med %d  epilogue med %d  total med %d ; kernel span %d"
      % (med(pro), med(main), med(epi), med(tot), span))
print("shares: pro %.1f%% main %.1f%% epi %.1f%%" % (100 * pro.sum() / tot.sum(), 100 * main.sum() / tot.sum(),
                                                   100 * epi.sum() / tot.sum()))
print("epilogue split (wave 0): half0 med %d  half1 med %d  wave3 done med %d  block_sum+tail med %d"
      % (med(a[:, 4] - a[:, 2]), med(a[:, 5] - a[:, 4]), med(a[:, 6] - a[:, 2]),
         med(a[:, 3] - np.maximum(a[:, 5], a[:, 6]))))
print("sum(tile lifetimes)/span = %.1f resident tiles on average" % (tot.sum() / span))
# co-residency per CU: HW_ID bits: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13 (gfx9 layout); XCC in the high word
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 0xF) << 4) | ((hw >> 32) << 8)
ids, cnt = np.unique(cu, return_counts=True)
print("distinct CU ids", len(ids), "tiles per CU min/med/max", cnt.min(), med(cnt), cnt.max())
conc = []
for c in ids[:64]:
    t = a[cu == c]
    ev = sorted([(x, 1) for x in t[:, 0]] + [(x, -1) for x in t[:, 3]])
    cur = 0; last = ev[0][0]; area = 0
    for x, d in ev:
        area += cur * (x - last); last = x; cur += d
    conc.append(area / max(1, ev[-1][0] - ev[0][0]))
print("mean resident workgroups per CU (first 64 CUs): %.2f" % np.mean(conc))
