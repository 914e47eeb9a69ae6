import sys, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch, bench
import config_bench as cb
tn, einstr, shapes, path = bench.build_network(100, 256, 4)
for R in [int(x) for x in sys.argv[1:]] or (2, 4, 6, 8, 12, 16, 32):
    bc, ops, out, wall, ms, by = cb.run_device(einstr, shapes, path, replicas=R, iters=10, scale=16.0)
    tiles = {}
    for t in bc.executor.step_tiles():
        tiles[str(t)] = tiles.get(str(t), 0) + 1
    print(json.dumps({"R": R, "ms": round(wall * 1e3, 3), "tflops": round(bc.plan.flops * R / wall / 1e12, 2), "tiles": tiles}), flush=True)
    del ops, out, bc
    torch.cuda.empty_cache()
