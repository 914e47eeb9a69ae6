#!/usr/bin/env python3
"""Headline benchmark: contractions/s on <phi|psi> of two 100-site MPS (bond 256, phys 4, fp32).

    python bench.py --gpus N --steps K --warmup W [--replicas R] [--sites 100 --bond 256 --phys 4]
    python bench.py --config peps --gpus N --steps K --warmup W [--rows 8 --cols 8 --bond 8 --slices 64]

Default (``--config mps``, the BASELINE metric): one "step" = one pass of the hot path (TN.contract's
pairwise loop, stabilised, split format) over one batch of R independent synthetic networks whose tensors
are already resident in HBM.  N>1: launched by torch.distributed.run, one rank per GPU; every rank
contracts its own R networks (weak scaling, no data-path collective: replicas are independent -
SURVEY.md 8e); timing is bracketed by a barrier + device synchronize and the MAX over ranks is reported.

``--config peps`` (BASELINE config 5): ONE closed rows x cols PEPS (d = 2, bond D) is
contracted per step, sharded over the ranks by index slicing: slice labels and path are chosen together
(`dist.sliced_plan`), every rank runs its share of the slices as replicas of one plan on its GPU and the
partial results in split format are joined by ONE all_gather + log-sum-exp combine (strong scaling).

Rank 0 prints ONE JSON line (contract in the task statement) including
  roofline     - dominant kernel: algorithmic flop (or bytes) / HIP-event duration vs the MFMA (or HBM) peak
  cpu_baseline - the NumPy oracle on the same network/path on this box's host cores (N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz
PEAK_F64_MFMA_TFLOPS = 78.6   # half of it (secondary measurement: --dtype f64)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["mps", "peps"], default="mps",
                    help="mps = the BASELINE metric (replicas, weak scaling); peps = one 2D grid sharded by "
                         "index slicing with a single all_gather join (strong scaling)")
    ap.add_argument("--rows", type=int, default=8, help="peps: grid rows")
    ap.add_argument("--cols", type=int, default=8, help="peps: grid columns")
    ap.add_argument("--slices", type=int, default=64, help="peps: at least this many slices (shared by all ranks)")
    ap.add_argument("--max-intermediate", type=int, default=None,
                    help="peps: slice until no intermediate has more elements than this")
    ap.add_argument("--workspace-gib", type=float, default=64.0, help="peps: workspace budget per GPU for slices in flight")
    ap.add_argument("--plain-slicing", action="store_true",
                    help="peps: every slice repeats the whole path (dist.SlicedContraction) instead of the staged form, in "
                         "which slice-independent parts of the tree are contracted once (dist.StagedSlicedContraction)")
    ap.add_argument("--cross-check", action="store_true",
                    help="peps: contract the network once more on a different slicing (other labels, another tree) and "
                         "report the distance between the two values")
    ap.add_argument("--single-gpu-reference", action="store_true",
                    help="peps: rank 0 also measures the best single-GPU form of the same network (unsliced where it fits, "
                         "and the sliced plan on one GPU) and the line reports the speed-up against it")
    ap.add_argument("--no-peps", dest="with_peps", action="store_false",
                    help="mps: skip the secondary PEPS strong-scaling measurement that follows the headline")
    ap.add_argument("--peps-bonds", type=int, nargs="*", default=[8, 16], help="mps: bond dimensions of that secondary")
    ap.add_argument("--peps-timeout", type=float, default=240.0, help="mps: watchdog for the secondary, seconds")
    ap.add_argument("--no-batched", dest="with_batched", action="store_false",
                    help="mps: skip the secondary batched-MPS (BASELINE config 3b) measurement")
    ap.add_argument("--batch", type=int, default=4096, help="mps: inputs per GPU of that secondary")
    ap.add_argument("--no-configs", dest="with_configs", action="store_false",
                    help="mps: skip the `configs` object (BASELINE configs 1, 2, 3b at B = 1024 and 4 (i)-(iv), one GPU)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--replicas", type=int, default=512, help="independent networks per step per GPU")
    ap.add_argument("--sites", type=int, default=100)
    ap.add_argument("--bond", type=int, default=None, help="bond dimension (default: 256 for mps, 8 for peps)")
    ap.add_argument("--phys", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="mps: skip the R = 1 / 8 / 64 networks-in-flight figures")
    ap.add_argument("--streams", type=int, default=1,
                    help="split the replicas into this many groups, each on its own HIP stream")
    ap.add_argument("--event-passes", type=int, default=3, help="timed passes bracketed by HIP events")
    ap.add_argument("--dump-steps", default=None, help="write per-step kernel info + mean ms to this JSON file")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline time budget")
    ap.add_argument("--path", choices=["zipper", "auto"], default="zipper",
                    help="zipper = the explicit path of the metric's config (SURVEY.md 8d); auto = what the library's "
                         "own path search returns for this network (exploration; the workload label says so)")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32",
                    help="f32 = the BASELINE metric; f64 = the same workload in double precision (secondary)")
    args = ap.parse_args()
    if args.bond is None:
        args.bond = 256 if args.config == "mps" else 8
    return args


def build_network(sites, bond, phys):
    """Compile the TN once (front-end + path), exactly as TN.make_contract_fun would."""
    from contractn_amd import TN
    from contractn_amd.paths import ssa_to_linear
    from tests import networks as nets

    # tiny placeholder tensors are enough to build the graph; shapes are what matter
    tn, ssa = nets.mps_overlap(TN, sites, bond, phys, dtype=np.float32, seed=3, scale=16.0)
    path = ssa_to_linear(ssa, 2 * sites)
    shapes = tuple(tuple(p.shape) for p in tn.params)
    return tn, tn.einsum_str, shapes, path


def init_ranks(args):
    """One process per GPU (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE); returns the rank context."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but the launcher started {world} ranks"
    # rehearsal knobs (one-GPU box): CTN_BENCH_BACKEND=gloo + CTN_BENCH_ONE_DEVICE=1 run all ranks on cuda:0
    backend = os.environ.get("CTN_BENCH_BACKEND", "nccl")
    if os.environ.get("CTN_BENCH_ONE_DEVICE"):
        local_rank = 0
    # CTN_JOIN_WORLD1=1 under a one-rank launcher (`torch.distributed.run --nproc-per-node 1`): the process group is
    # created although there is nobody to talk to, and the joins issue their collectives anyway (dist.join_alone) - on
    # a one-GPU box this is the only way the RCCL path runs on hardware (tests/test_gpu_dist.py)
    alone = world == 1 and os.environ.get("CTN_JOIN_WORLD1") == "1" and "RANK" in os.environ
    if world > 1 or alone:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    return world, rank, local_rank, backend, torch.device("cuda", local_rank)


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as a CHILD
    `python -m torch.distributed.run ... bench.py <same arguments>` (one process per GPU, rendezvous on 127.0.0.1),
    pass its output through and return its exit code.  Called before anything has touched the GPU - this process
    never does - and never replaces itself: the ranks are children, the parent only waits."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    import torch
    import torch.distributed as dist

    world, rank, local_rank, backend, dev = init_ranks(args)
    if args.config == "peps":
        result = run_peps(args, world, rank, local_rank, backend, dev)
    else:
        result = run_mps(args, world, rank, local_rank, backend, dev)
        if args.with_batched and args.dtype == "f32":
            batched_secondary(args, result, world, rank, backend, dev)
        if args.with_peps and args.dtype == "f32":
            peps_secondary(args, result, world, rank, local_rank, backend, dev)
        if args.with_configs and args.dtype == "f32" and world == 1:
            configs_secondary(args, result, world, rank, backend, dev)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def batched_secondary(args, result, world, rank, backend, dev, batch=None, out=None, replicas=1):
    """BASELINE config 3b next to the headline: `--batch` inputs per GPU through ONE 100-site MPS (D = 256, d = 4)
    hanging on a batch hyperedge - the reference paper's ML workload (README Fig. 1d).  Data-parallel over the
    batch (weak scaling, no collective on the data path: SURVEY.md 8e); value = inputs/s over all ranks.  On one
    GPU the first 64 outputs are checked against the oracle run on those 64 inputs alone (batch independence).
    `replicas` > 1: that many BATCHES (own inputs, the same cores) as replicas of one launch sequence - how batches too
    small to fill the chip by themselves are meant to be run (1024 inputs are 64 row blocks of the sweep: a quarter of
    the CUs).  Errors are recorded in the object; the headline line is printed regardless."""
    import torch
    import torch.distributed as dist

    RB = int(replicas)
    if out is None:
        out = result.setdefault("batched_mps", {}) if rank == 0 else {}
    try:
        from contractn_amd import TN
        from contractn_amd.einsum import BatchedContraction, accumulate_log_scale
        from contractn_amd.paths import ssa_to_linear
        from tests import networks as nets

        B, n_sites, bond, phys = batch or args.batch, args.sites, args.bond, args.phys

        class Shape:
            def __init__(self, shape):
                self.shape, self.ndim = tuple(shape), len(shape)

        tn = TN()
        hub = tn.add_copy_node(n_sites + 1)
        cores = [Shape((phys, bond) if i in (0, n_sites - 1) else (phys, bond, bond)) for i in range(n_sites)]
        nodes = nets.add_mps(tn, cores)
        for i, node in enumerate(nodes):
            inp = tn.add_input_node((B, phys), var_shape_axes=(0,))
            tn.connect_nodes(inp, node, 1, 0)
            tn.connect_nodes(hub, inp, i, 0)
        shapes = [c.shape for c in cores] + [(B, phys)] * n_sites
        path = ssa_to_linear(nets.batched_mps_path(n_sites), 2 * n_sites)
        bc = BatchedContraction(tn.einsum_str, shapes, np.float32, optimize=path, replicas=RB, device=dev.index)
        gen = torch.Generator(device=dev)
        gen.manual_seed(11 + rank)
        ops = [torch.randn(sh, generator=gen, device=dev) / 4.0 for sh in shapes]
        # further batches: their own inputs, the cores of the first
        more = [[torch.randn(sh, generator=gen, device=dev) / 4.0 for sh in shapes[n_sites:]] for _ in range(RB - 1)]
        sets = [ops] + [ops[:n_sites] + m_ for m_ in more]
        res = torch.zeros((RB,) + tuple(bc.plan.out_shape), device=dev)
        launch = bc.executor.make_enqueue([t.data_ptr() for set_ in sets for t in set_], [res[r_].data_ptr() for r_ in range(RB)])
        for _ in range(3):
            launch()
        bc.executor.synchronize()
        K = max(args.steps, 10)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            launch()
        bc.executor.synchronize()
        torch.cuda.synchronize()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        sec = float(dt.item()) / K
        if rank != 0:
            return
        infos = bc.plan.step_infos()
        launches = sum(i["kernel"] != 5 for i in infos)
        out.update({
            "workload": f"batched_mps_{n_sites}sites_D{bond}_d{phys}_B{B}_per_gpu" + (f"_x{RB}_batches_in_flight" if RB > 1 else ""),
            "value": round(world * RB * B / sec, 1),
            "unit": "inputs/s", "ms_per_pass": round(sec * 1e3, 4), "passes": K, "scaling": "weak",
            "achieved_tflops": round(world * RB * bc.plan.flops / sec / 1e12, 2),
            "frac_of_mfma_peak": round(RB * bc.plan.flops / sec / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "launches_per_pass": launches, "epilogue_summed_steps": sum(i["epilogue_sum"] > 0 for i in infos),
            "largest_intermediate_elements": max(i["out_numel"] for i in infos if i["kernel"] != 5),
        })
        # the per-site launch (GEMM with the physical-leg sum in its epilogue) on its own: HIP events around every
        # launch of two more passes (eager launches, outside the timed region above)
        bc.executor.set_timing(2)
        for _ in range(2):
            launch()
        bc.executor.synchronize()
        ms = bc.executor.step_ms().astype(np.float64)
        bc.executor.set_timing(0)
        dom = [i for i, x in enumerate(infos) if x["epilogue_sum"] > 0]
        tiles = bc.executor.step_tiles()
        sweep = [i for i, t in enumerate(tiles) if t[0] == 16 and t[1] >= 128]     # (16, bond x phys): k_sweep_f32
        if sweep:
            # the interior sites went out as ONE launch (k_sweep_f32) at the position of the last of them; its events
            # also bracket the three bookkeeping launches behind it (k_sweep_logs, k_sweep_z, k_sweep_finish)
            members = [i for i in dom if i == sweep[0] or (i < sweep[0] and tiles[i] == (1, 1))]
            us = float(ms[sweep[0]]) * 1e3
            fl = float(sum(infos[i]["flops"] for i in members)) * RB
            out["launches_per_pass"] = sum(i["kernel"] != 5 for k, i in enumerate(infos) if tiles[k] != (1, 1)) + 3
            out["sites_in_one_launch"] = len(members)
            out["dominant_kernel"] = {
                "kernel": "k_sweep_f32 (every interior site in one launch: 16 inputs per workgroup walk the chain, the cores "
                          "stream from L2 into MFMA operand registers) + k_sweep_logs, k_sweep_z, k_sweep_finish (the reference's rescale factors)",
                "bound": "mfma", "launches_per_pass": 1, "avg_launch_us": round(us, 2), "flop_per_launch": fl,
                "achieved": round(fl / (us * 1e-6) / 1e12, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(fl / (us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                "timing_mode": "HIP events on the executor's stream around the launch in 2 extra passes (eager launches)",
            }
        elif dom:
            us = float(np.mean(ms[dom])) * 1e3
            fl = float(np.mean([infos[i]["flops"] for i in dom])) * RB
            out["dominant_kernel"] = {
                "kernel": "k_mfma_f32<..., EPW> (one launch per site: GEMM + re-weighted sum over the physical leg in the epilogue)",
                "bound": "mfma", "launches_per_pass": len(dom), "avg_launch_us": round(us, 2), "flop_per_launch": fl,
                "achieved": round(fl / (us * 1e-6) / 1e12, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(fl / (us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                "timing_mode": "HIP events on the executor's stream around every launch of 2 extra passes (eager launches)",
            }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import cpu_ref

            _dev_log, resc = bc.executor.fetch()
            err = 0.0
            for r_ in sorted({0, RB - 1}):          # the first and the last batch in flight
                c = float(accumulate_log_scale(resc[r_], np.dtype(np.float32)))
                got = res[r_, :64].cpu().numpy().astype(np.float64) * np.exp(c)
                h_ops = [o[:64].cpu().numpy() if tuple(o.shape) == (B, phys) else o.cpu().numpy() for o in sets[r_]]
                rt, rc = cpu_ref.contract(tn.einsum_str, *h_ops, path=list(path), split_format=True)
                ref = np.asarray(rt, dtype=np.float64) * np.exp(float(rc))
                err = max(err, float(np.max(np.abs(got - ref)) / np.max(np.abs(ref))))
            out["parity_vs_oracle_first_64_inputs"] = {"ok": bool(err <= 1e-3), "max_rel_err": err, "tolerance": 1e-3,
                                                       "batches_checked": sorted({0, RB - 1})}
    except Exception as exc:  # noqa: BLE001 - recorded, the headline line must still be printed
        if rank == 0:
            out["error"] = repr(exc)
    finally:
        torch.cuda.empty_cache()


def configs_secondary(args, result, world, rank, backend, dev):
    """The other BASELINE configs in the driver's own line (round-3 verdict, item 2), one GPU, after the headline and
    outside its timed region; every leg on its own try / except so that the headline never depends on them:

    * cfg1 - README copy-tensor example (README.md:21-33): order-101 copy node + 100 vectors, fp64; device us per
      contraction (HIP events), ms per `contract_fun` call, the known answer [1, 0.99^100];
    * cfg2 - 1000 x (3 x 3) chain, split format (README.md:62-77): device us per step, the register's bits
      (0x1.12a72fbccf574p+10);
    * cfg3b_B1024 - the batched MPS at the smaller of the two batch sizes SURVEY.md 8d names;
    * cfg4_i .. cfg4_iv - CP through a copy node, Tucker with a dense hub, Tucker with a MATERIALISED delta hub (must
      equal 4 (i)), CP-wide r = 4096: TFLOP/s end to end, fraction of the fp32 MFMA peak, dominant kernel, an entry
      against the definition in float64.
    cfg3a is the headline itself (`value`, `latency`), cfg5 is `peps_strong_scaling`."""
    import torch

    if rank != 0:
        return
    cfgs = result.setdefault("configs", {"cfg3a": "the headline: `value`, `roofline`, `latency`",
                                         "cfg5": "`peps_strong_scaling` (8 x 8 PEPS, D = 8 and D = 16)"})
    from contractn_amd import TN
    from contractn_amd import einsum as E
    from contractn_amd.einsum import BatchedContraction

    def leg(name, fn):
        try:
            cfgs[name] = fn()
        except Exception as exc:  # noqa: BLE001 - recorded, the headline line must still be printed
            cfgs[name] = {"error": repr(exc)}
        finally:
            torch.cuda.empty_cache()

    def readme_example(which):
        tn = TN()
        if which == 1:
            hub = tn.add_copy_node(101)
            for i in range(100):
                tn.connect_nodes(hub, tn.add_dense_node(np.array([1, 0.99])), i, 0)
        else:
            prev = tn.add_dense_node(np.ones(3))
            for _ in range(1000):
                mat = tn.add_dense_node(np.ones((3, 3)))
                tn.connect_nodes(prev, mat, -1, 0)
                prev = mat
        fun = tn.make_contract_fun(split_format=True)
        params = tn.params
        t_hat, c = fun(params, ())
        calls = []
        for _ in range(20):
            t0 = time.perf_counter()
            fun(params, ())
            calls.append(time.perf_counter() - t0)
        bc = BatchedContraction(tn.einsum_str, [p.shape for p in params], np.float64, replicas=1, device=dev.index)
        sets = [list(params)]
        bc.run_host(sets)
        bc.executor.set_timing(5)
        for _ in range(5):
            bc.run_host(sets)
        dev_ms = float(bc.executor.step_ms().sum())
        bc.executor.close()
        out = {"steps": bc.plan.n_steps, "dtype": "f64", "call_ms_median": round(float(np.median(calls)) * 1e3, 4),
               "device_us_per_contraction": round(dev_ms * 1e3, 2), "device_us_per_step": round(dev_ms * 1e3 / bc.plan.n_steps, 3),
               "timing_mode": "HIP events around the launches of 5 passes (device); perf_counter around 20 contract_fun calls"}
        with np.errstate(over="ignore"):      # (config 2's register is ~1098: its plain value is inf by design)
            full = np.asarray(t_hat, dtype=np.float64) * np.exp(float(c))
        if which == 1:
            want = np.array([1.0, 0.99 ** 100])
            out["workload"] = "readme_copy_node_order101_100_vectors"
            out["known_answer"] = {"ok": bool(np.allclose(full, want, rtol=1e-12)), "got": [float(x) for x in full],
                                   "want": [float(x) for x in want], "source": "README.md:33"}
            out["reference_published_ms"] = 6.85
        else:
            out["workload"] = "chain_1000x3x3_split_format"
            out["known_answer"] = {"ok": bool(np.array_equal(np.asarray(t_hat), np.ones(3)) and float(c).hex() == "0x1.12a72fbccf574p+10"),
                                   "t_hat": [float(x) for x in np.asarray(t_hat)], "log_scale": float(c),
                                   "log_scale_hex": float(c).hex(), "want_hex": "0x1.12a72fbccf574p+10", "source": "README.md:73-76"}
        return out

    def cfg4(einstr, shapes, seed, scale, spot, hub=None):
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        ops = []
        for i, sh in enumerate(shapes):
            if hub == "delta" and i == 0:
                t = torch.zeros(sh, device=dev)
                idx = torch.arange(sh[0], device=dev)
                t[idx, idx, idx] = 1.0
            else:
                t = torch.randn(sh, generator=gen, device=dev) / scale
            ops.append(t)
        bc = BatchedContraction(einstr, shapes, np.float32, optimize="auto", replicas=1, device=dev.index)
        res = torch.empty(tuple(bc.plan.out_shape), device=dev)
        torch.cuda.synchronize(dev)
        launch = bc.executor.make_enqueue([t.data_ptr() for t in ops], [res.data_ptr()])
        for _ in range(3):          # (the third enqueue captures and instantiates the hipGraph: not in the timed region)
            launch()
        bc.executor.synchronize()
        K = 3
        t0 = time.perf_counter()
        for _ in range(K):
            launch()
        bc.executor.synchronize()
        sec = (time.perf_counter() - t0) / K
        bc.executor.set_timing(2)
        for _ in range(2):
            launch()
        bc.executor.synchronize()
        ms = bc.executor.step_ms().astype(np.float64)
        bc.executor.set_timing(0)
        infos, tiles = bc.plan.step_infos(), bc.executor.step_tiles()
        c = float(bc.fetch_log_scale()[0])
        dom = int(np.argmax(ms))
        key = (infos[dom]["kernel"], infos[dom]["mode_a"], infos[dom]["mode_b"], tiles[dom][0], tiles[dom][1])
        dom_flops = infos[dom]["flops"] + sum(i["flops"] for i in infos if i["kernel"] == 5)   # absorbed steps run inside it
        got, ref, peak_abs = spot(ops, res, c)
        out = {"workload": einstr, "shapes": [list(sh) for sh in shapes], "ms_per_contraction": round(sec * 1e3, 3),
               "flop_per_contraction": bc.plan.flops, "achieved_tflops": round(bc.plan.flops / sec / 1e12, 2),
               "frac_of_mfma_peak": round(bc.plan.flops / sec / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
               "steps": [{"kernel": kernel_label((i["kernel"], i["mode_a"], i["mode_b"], tl[0], tl[1])), "ms": round(float(m), 3)}
                         for i, tl, m in zip(infos, tiles, ms)],
               "dominant_kernel": {"kernel": kernel_label(key), "avg_launch_ms": round(float(ms[dom]), 3),
                                   "achieved": round(dom_flops / (ms[dom] * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                                   "frac": round(dom_flops / (ms[dom] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)},
               "spot_check_vs_definition_f64": {"ok": bool(abs(got - ref) <= 1e-3 * peak_abs), "got": got, "want": ref,
                                                "tolerance": "1e-3 of the largest |entry|"},
               "timing_mode": f"perf_counter around {K} enqueues + synchronize (hipGraph replay); per step: HIP events, 2 more passes"}
        bc.executor.close()
        return out

    def spot_cp(ops, res, c, e=(5, 700, 1023)):
        A, B, C = ops[-3:]
        ref = float((A[:, e[0]].double() * B[:, e[1]].double() * C[:, e[2]].double()).sum())
        scale = float(np.exp(c))
        return float(res[e]) * scale, ref, float(res.abs().max()) * scale

    def spot_tucker(ops, res, c, e=(5, 700, 1023)):
        H, A, B, C = ops
        v = torch.einsum("abc,c->ab", H.double(), C[:, e[2]].double())
        ref = float(A[:, e[0]].double() @ v @ B[:, e[1]].double())
        scale = float(np.exp(c))
        return float(res[e]) * scale, ref, float(res.abs().max()) * scale

    leg("cfg1", lambda: readme_example(1))
    leg("cfg2", lambda: readme_example(2))
    b1024 = {}
    batched_secondary(args, result, world, rank, backend, dev, batch=1024, out=b1024)
    cfgs["cfg3b_B1024"] = b1024
    b1024x4 = {}      # ... and what small batches are meant to do: several in flight as replicas of one launch sequence
    batched_secondary(args, result, world, rank, backend, dev, batch=1024, out=b1024x4, replicas=4)
    cfgs["cfg3b_B1024_x4_in_flight"] = b1024x4
    n = 1024
    leg("cfg4_i_cp_hyper", lambda: cfg4("ac,ad,ae->cde", [(n, n)] * 3, 5, 32.0, spot_cp))
    leg("cfg4_ii_tucker_dense_hub", lambda: cfg4("abc,ae,bf,cg->efg", [(n, n, n)] + [(n, n)] * 3, 7, 32.0, spot_tucker))
    leg("cfg4_iii_tucker_delta_hub", lambda: cfg4("abc,ae,bf,cg->efg", [(n, n, n)] + [(n, n)] * 3, 5, 32.0, spot_cp, hub="delta"))
    leg("cfg4_iv_cp_wide_r4096", lambda: cfg4("ac,ad,ae->cde", [(4096, n)] * 3, 9, 64.0, spot_cp))


def peps_secondary(args, result, world, rank, local_rank, backend, dev):
    """After the headline measurement, the SAME launch also contracts the 8 x 8 PEPS of BASELINE config 5 sharded
    over all its ranks (`--config peps` in short form: D = 8 with 64 slices, D = 16 with 256), so that a run at
    N = 1, 2, 4, 8 GPUs leaves the strong-scaling figures of the index-sliced path next to the weak-scaling
    headline.  It never loses the headline line: errors are recorded in the object, and a watchdog thread prints
    the line and ends the process - with a non-zero exit code - if a rank gets stuck in a collective."""
    import copy
    import threading

    import torch

    torch.cuda.empty_cache()
    out = result.setdefault("peps_strong_scaling", {}) if rank == 0 else {}

    def give_up():
        # a rank stuck in a kernel or a collective: the headline line (already measured) is still printed, but the
        # process ends NON-ZERO on every rank, so that the hang is investigated instead of being read as a clean run
        if rank == 0:
            out["error"] = (f"watchdog: secondary not finished after {args.peps_timeout:.0f}s; every rank exits "
                            f"non-zero (rank 0 with 4, the others with 3)")
            print(json.dumps(result), flush=True)
        os._exit(4 if rank == 0 else 3)

    dog = threading.Timer(args.peps_timeout, give_up)
    dog.daemon = True
    dog.start()
    try:
        for bond, slices, max_int, steps, warmup in ((8, 64, None, 20, 3), (16, 64, 2 ** 32, 1, 1)):
            if bond not in args.peps_bonds:
                continue
            a2 = copy.copy(args)
            a2.rows, a2.cols, a2.bond, a2.slices, a2.max_intermediate = 8, 8, bond, slices, max_int
            a2.steps, a2.warmup, a2.no_cpu_baseline, a2.dump_steps, a2.event_passes = steps, warmup, True, None, 1
            a2.single_gpu_reference = True
            a2.cross_check = bond == 16        # D = 16 has no CPU oracle: two different slicings must agree
            try:
                full = run_peps(a2, world, rank, local_rank, backend, dev)
            except Exception as exc:  # noqa: BLE001 - recorded, the headline line must still be printed
                full = {"error": repr(exc)}
            if rank == 0:
                keep = ("value", "unit", "ms_per_step", "steps", "warmup", "scaling", "achieved_tflops", "result", "error",
                        "strong_scaling", "cross_check")
                short = {k: full[k] for k in keep if k in full}
                if "config" in full:
                    short["config"] = {k: full["config"].get(k) for k in ("workload", "slices", "slices_per_gpu",
                                                                          "work_overhead_vs_unsliced",
                                                                          "flop_per_contraction_sliced",
                                                                          "largest_intermediate_elements",
                                                                          "hbm_free_gib_with_all_stage_buffers_allocated")}
                    r = full["roofline"]
                    short["dominant_kernel"] = {k: r[k] for k in ("kernel", "bound", "achieved", "unit", "frac",
                                                                  "share_of_device_time")}
                out[f"D{bond}"] = short
            torch.cuda.empty_cache()
        if rank == 0:
            out["quoted_size"] = peps_quoted_size()
    finally:
        dog.cancel()


def peps_quoted_size():
    """Which PEPS size the strong-scaling claim is quoted on, and the per-rank evidence one GPU allows (round-3 verdict,
    item 1): `tools/peps_rank_time.py` builds and times EVERY rank's share of the staged plan for world = 2, 4, 8 on one
    GPU; the committed summaries (profiles/r04_peps_D{8,16}_rank_time.jsonl) are attached with their source - they are NOT
    measured in this run.  D = 8 is 0.12 TFLOP per contraction: 3 ms on one GPU, and a rank's share at 8 ranks is ~0.9 ms
    of 7-us launches whatever its slice count (predicted 3.2 x); D = 16 (247 TFLOP unsliced, 256 slices over two labels in stages, 4.0 s
    on one GPU) is the size whose work is large enough to shard: quoted."""
    out = {"size": "D16", "why": ("8 x 8, D = 8 is 0.12 TFLOP: 3 ms on ONE GPU, a rank's share at 8 ranks is ~0.9 ms of "
                                  "latency-bound launches; D = 16 (256 slices over two labels, 4.0 s on one GPU) is quoted for strong scaling"),
           "predicted_from_one_gpu_rank_timing": {}}
    for bond in (8, 16):
        f = os.path.join(ROOT, "profiles", f"r04_peps_D{bond}_rank_time.jsonl")
        try:
            rows = [json.loads(ln) for ln in open(f) if ln.strip()]
        except (OSError, ValueError):
            continue
        out["predicted_from_one_gpu_rank_timing"][f"D{bond}"] = {
            "source": f"profiles/r04_peps_D{bond}_rank_time.jsonl (tools/peps_rank_time.py: every rank's share timed on one MI355X)",
            "by_world": {str(r_["world"]): {k: r_[k] for k in ("max_rank_ms", "min_rank_ms", "best_single_gpu_ms", "join_us_assumed",
                                                                 "predicted_speedup")}
                         for r_ in rows if r_.get("summary")}}
    return out


def run_mps(args, world, rank, local_rank, backend, dev):
    """The BASELINE metric: R independent 100-site MPS overlaps per GPU and step (weak scaling)."""
    import torch
    import torch.distributed as dist

    from contractn_amd.einsum import BatchedContraction

    R = args.replicas
    tn, einstr, shapes, path = build_network(args.sites, args.bond, args.phys)
    if args.path == "auto":
        from contractn_amd import paths as _paths

        _terms, _out, _sizes = _paths.parse_einsum_input(einstr, shapes)
        path = tuple(tuple(p) for p in _paths.find_path(_terms, _out, _sizes, "auto"))
    S = max(1, args.streams)
    assert R % S == 0, "--replicas must be a multiple of --streams"
    Rg = R // S
    f64 = args.dtype == "f64"
    np_dt, t_dt, esz = (np.float64, torch.float64, 8) if f64 else (np.float32, torch.float32, 4)
    peak = PEAK_F64_MFMA_TFLOPS if f64 else PEAK_F32_MFMA_TFLOPS
    groups = [BatchedContraction(einstr, shapes, np_dt, optimize=path, replicas=Rg, device=local_rank)
              for _ in range(S)]
    bc = groups[0]
    plan, ex = bc.plan, bc.executor
    infos = plan.step_infos()

    # synthetic inputs, resident in HBM: standard normal / 16, one seed per replica
    gen = torch.Generator(device=dev)
    numels = [int(np.prod(s)) for s in shapes]
    offs = np.concatenate([[0], np.cumsum([(n + 63) // 64 * 64 for n in numels])])
    flat = []
    in_ptrs = []
    for r in range(R):
        gen.manual_seed(3 + r + 1000 * rank)
        buf = torch.randn(int(offs[-1]), generator=gen, device=dev, dtype=t_dt) / 16.0
        flat.append(buf)
        base = buf.data_ptr()
        in_ptrs.extend(base + esz * int(offs[i]) for i in range(len(shapes)))
    out = torch.zeros(R, max(1, int(np.prod(plan.out_shape))), device=dev, dtype=t_dt)
    out_ptrs = [out[r].data_ptr() for r in range(R)]
    n_in = len(shapes)
    launchers = [g.executor.make_enqueue(in_ptrs[i * Rg * n_in:(i + 1) * Rg * n_in], out_ptrs[i * Rg:(i + 1) * Rg])
                 for i, g in enumerate(groups)]

    def launch():
        for fn in launchers:
            fn()

    def sync_all():
        for g in groups:
            g.executor.synchronize()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    # an executor captures its launch sequence as a hipGraph on its second enqueue: with fewer than two warm-up
    # steps that one-off (a few ms) would land in the timed region, so the missing enqueues are issued here
    for _ in range(max(0, 2 - args.warmup)):
        launch()
    for _ in range(args.warmup):
        launch()
    sync_all()

    # ---- timed region: exactly K steps --------------------------------------------------
    # HIP events around every step's kernels, on the executor's stream, for the first passes of the
    # timed region (sampling keeps the event overhead out of most of the measurement)
    ex.set_timing(min(args.steps, args.event_passes))
    sync_all()
    sampler = PowerSampler(dev) if rank == 0 else None   # sysfs reads on a host thread: clock and power WHILE timed
    t0 = time.perf_counter()
    for _ in range(args.steps):
        launch()
    sync_all()
    elapsed = time.perf_counter() - t0
    under_load = sampler.stop() if sampler else {}
    step_ms_last = ex.step_ms().astype(np.float64)  # per-step mean over the event-bracketed passes
    ex.set_timing(0)
    logs = np.concatenate([g.fetch_log_scale() for g in groups])
    t_hat = out[:, 0].cpu().numpy()

    if world > 1:
        tmax = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    total_contractions = R * args.steps * world
    value = total_contractions / elapsed
    flops_per = plan.flops
    tflops = value * flops_per / 1e12

    if args.dump_steps and rank == 0:
        with open(args.dump_steps, "w") as fh:
            json.dump([dict(info, ms=float(step_ms_last[s])) for s, info in enumerate(infos)], fh)

    # ---- roofline of the dominant kernel (per launch, HIP-event durations) ---------------
    by_kernel = {}
    tiles = ex.step_tiles()   # what the launcher actually ran (it may retile by replica count)
    pending, pending_bytes, zip_bytes = 0.0, 0.0, {}
    for s, info in enumerate(infos):
        if tiles[s] == (1, 1):      # the first step of a zipper pair: it runs inside the next step's launch (k_zip_f32)
            pending += info["flops"] * Rg
            pending_bytes += esz * Rg * info["batch"] * (info["m"] * info["k"] + info["k"] * info["n"])   # E and X once
            continue
        key = (info["kernel"], info["mode_a"], info["mode_b"], tiles[s][0], tiles[s][1])
        d = by_kernel.setdefault(key, {"ms": 0.0, "flops": 0.0, "launches": 0})
        d["ms"] += step_ms_last[s]
        d["flops"] += info["flops"] * Rg + pending
        d["launches"] += 1
        if pending:                 # ... Y once, E' once; T never reaches memory
            zip_bytes[key] = pending_bytes + esz * Rg * info["batch"] * (info["k"] * info["n"] + info["m"] * info["n"])
        pending, pending_bytes = 0.0, 0.0
    dom_key = max(by_kernel, key=lambda k: by_kernel[k]["ms"])
    dom = by_kernel[dom_key]
    from contractn_amd.engine import KERNEL_NAMES

    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
    mfma_ms = sum(d["ms"] for k, d in by_kernel.items() if k[0] in (2, 3))
    mfma_flops = sum(d["flops"] for k, d in by_kernel.items() if k[0] in (2, 3))
    workload = f"mps_overlap_{args.sites}sites_D{args.bond}_d{args.phys}_{args.path}_R{R}"
    traffic, traffic_source = pmc_traffic(workload, kernel_label(dom_key))
    # algorithmic bytes of one launch of the dominant kernel: every operand and the output once
    dom_info = next(info for s_, info in enumerate(infos)
                    if (info["kernel"], info["mode_a"], info["mode_b"], tiles[s_][0], tiles[s_][1]) == dom_key)
    alg_bytes = esz * Rg * dom_info["batch"] * (dom_info["m"] * dom_info["k"] + dom_info["k"] * dom_info["n"]
                                                + dom_info["m"] * dom_info["n"])
    if dom_key in zip_bytes:
        alg_bytes = zip_bytes[dom_key]
    roofline = {
        "bound": "mfma",
        "kernel": kernel_label(dom_key),
        "achieved": round(achieved, 3),
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4),
        "traffic": traffic,
        "traffic_source": traffic_source,
        "algorithmic_bytes_per_launch": alg_bytes,
        "timing_mode": (f"HIP events on the executor's stream around every launch of the first "
                        f"{min(args.steps, args.event_passes)} timed steps (eager launches; the other timed steps "
                        f"replay the same kernels as one hipGraph)"),
        "launches_per_step": dom["launches"],
        "avg_launch_us": round(dom["ms"] * 1e3 / max(dom["launches"], 1), 2),
        "flop_per_launch": dom["flops"] / max(dom["launches"], 1),
        "all_mfma_launches_tflops": round(mfma_flops / (mfma_ms * 1e-3) / 1e12, 3) if mfma_ms > 0 else None,
        "other_mfma_kernels": [
            {"kernel": kernel_label(k), "launches_per_step": d["launches"],
             "avg_launch_us": round(d["ms"] * 1e3 / max(d["launches"], 1), 2),
             "achieved": round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 3) if d["ms"] > 0 else None}
            for k, d in sorted(by_kernel.items(), key=lambda kv: -kv[1]["ms"]) if k[0] in (2, 3) and k != dom_key],
        "end_to_end_tflops": round(tflops / world, 3),
        "end_to_end_frac": round(tflops / world / peak, 4),
    }

    result = {
        "metric": (f"contractions/sec (MPS-{args.sites} overlap, bond={args.bond}, phys={args.phys}, "
                   f"{'fp64' if f64 else 'fp32'}, stabilised split format)"),
        "value": round(value, 2),
        "unit": "contractions/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic (standard normal / 16, on-device generator, seeds 3+replica)",
        "config": {
            "workload": workload,
            "replicas_per_gpu": R,
            "steps_per_contraction": plan.n_steps,
            "flop_per_contraction": flops_per,
            "bytes_min_per_contraction": plan.bytes_min,
            "parallelism": f"replicas x{world}",
        },
        "achieved_tflops": round(tflops, 3),
        "roofline": roofline,
        "device": dict(device_info(dev), **under_load),
    }
    # context, not the graded fraction: the timed region runs at the board's power cap, below the 2.4 GHz the
    # nominal peak assumes; the same kernel figure against the peak at the clock that was actually held
    if under_load.get("sclk_under_load_mhz"):
        at_clock = peak * under_load["sclk_under_load_mhz"] / 2400.0
        roofline["peak_at_measured_clock"] = round(at_clock, 1)
        roofline["frac_at_measured_clock"] = round(achieved / at_clock, 4)

    # ---- a few networks in flight (SURVEY.md 8d cfg 3a: R in {1, 8, 64}; H1: "report R = 1 latency and R >> 1
    # throughput"): the same network, path and tensors, R of the replicas above per launch sequence, rank 0 ----
    if rank == 0 and not args.no_latency:
        result["latency"] = few_in_flight(args, einstr, shapes, path, np_dt, in_ptrs, out_ptrs, n_in, local_rank, plan.flops, peak)

    # ---- CPU baseline: the oracle on the same network and path (rank 0, N=1 only) --------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(einstr, shapes, path, flat[0], offs, numels, args.cpu_seconds,
                                              float(t_hat[0]), float(logs[0]), f64)
        if R > 1:   # ... and the LAST replica of the launch against the oracle as well (one more CPU contraction)
            extra = cpu_baseline(einstr, shapes, path, flat[R - 1], offs, numels, 0.0, float(t_hat[R - 1]), float(logs[R - 1]), f64)
            pv = result["cpu_baseline"]["parity_vs_gpu"]
            pv["last_replica"] = extra["parity_vs_gpu"]
            pv["ok"] = bool(pv["ok"] and extra["parity_vs_gpu"]["ok"])
    return result if rank == 0 else None


def few_in_flight(args, einstr, shapes, path, np_dt, in_ptrs, out_ptrs, n_in, device, flops, peak):
    """ms per pass and TFLOP/s with R = 1, 8 and 64 networks in flight (three short timed loops after the headline,
    outside its timed region; hipGraph replay like the headline).  One network of this dependent chain cannot fill
    256 CUs: R = 1 is a latency figure, the headline's R is the throughput figure."""
    from contractn_amd.einsum import BatchedContraction

    out = {"unit": "ms per pass of R networks", "passes": None, "by_R": {}}
    try:
        for R in (1, 8, 64):
            if R > len(out_ptrs):
                break
            bc = BatchedContraction(einstr, shapes, np_dt, optimize=path, replicas=R, device=device)
            launch = bc.executor.make_enqueue(in_ptrs[:R * n_in], out_ptrs[:R])
            for _ in range(4):
                launch()
            bc.executor.synchronize()
            reps = 20 if R < 64 else 10
            t0 = time.perf_counter()
            for _ in range(reps):
                launch()
            bc.executor.synchronize()
            sec = (time.perf_counter() - t0) / reps
            out["by_R"][str(R)] = {"ms": round(sec * 1e3, 4), "contractions_per_s": round(R / sec, 1),
                                   "tflops": round(R * flops / sec / 1e12, 2),
                                   "frac_of_mfma_peak": round(R * flops / sec / 1e12 / peak, 4)}
            out["passes"] = reps
            bc.executor.close()
    except Exception as exc:  # noqa: BLE001 - recorded, the headline line must still be printed
        out["error"] = repr(exc)
    return out


PEAK_HBM_TBS = 8.0            # MI355X_MICROARCH.md: HBM3E peak (spec)


def peps_network(rows, cols, bond):
    """The closed rows x cols PEPS of BASELINE config 5: site legs (phys, up, left, down,
    right - the existing ones), d = 2, one vector per physical leg, built through the TN API; values
    standard normal / sqrt(D), fp32, seed 6 (the same tensors on every rank)."""
    from contractn_amd import TN
    from tests import networks as nets

    tn = nets.peps_closed(TN, rows, cols, 2, dtype=np.float32, seed=6)   # graph only: bond-2 placeholders
    shapes = [tuple(bond if (d == 2 and p.ndim > 1 and ax > 0) else d for ax, d in enumerate(p.shape))
              for p in tn.params]
    rng = np.random.default_rng(6)
    ops = [(rng.standard_normal(s) / bond ** 0.5).astype(np.float32) for s in shapes]
    return tn.einsum_str, shapes, ops


def run_peps(args, world, rank, local_rank, backend, dev):
    """BASELINE config 5: one PEPS contraction per step, slices sharded over the ranks, one all_gather join."""
    import torch
    import torch.distributed as dist

    from contractn_amd import dist as cdist
    from contractn_amd.engine import KERNEL_NAMES

    rows, cols, bond = args.rows, args.cols, args.bond
    einstr, shapes, ops = peps_network(rows, cols, bond)
    # slice labels + path: found once (rank 0; cached under contractn_amd/plans/), shared with every rank
    staged = not args.plain_slicing
    box = [None]
    t0 = time.perf_counter()
    if rank == 0:
        if staged:
            box[0] = cdist.staged_plan(einstr, shapes, min_slices=args.slices, max_intermediate=args.max_intermediate)
        else:
            box[0] = cdist.sliced_plan(einstr, shapes, min_slices=args.slices, max_intermediate=args.max_intermediate)
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    labels, path, rep = box[0]
    search_s = time.perf_counter() - t0
    make = cdist.StagedSlicedContraction if staged else cdist.SlicedContraction
    sc = make(einstr, ops, labels, optimize=path, rank=rank, world=world, device=local_rank,
              workspace_budget=int(args.workspace_gib * 2 ** 30))
    stages = sc.stage_list()          # [(BatchedContraction, evaluations on this rank, replicas per launch, launches)]
    hbm_free, hbm_total = torch.cuda.mem_get_info(dev)       # with every stage buffer and workspace of this rank allocated
    execs = [st[0].executor for st in stages]
    max_chunks = max([st[3] for st in stages], default=1)

    pg = dist.is_available() and dist.is_initialized()      # (also at world 1 under CTN_JOIN_WORLD1=1, see init_ranks)

    def sync_all():
        for x in execs:
            x.synchronize()
        torch.cuda.synchronize(dev)
        if pg:
            dist.barrier()

    # (an executor replays its launch sequence as a hipGraph from its third enqueue on: at least two eager ones first)
    for _ in range(max(0, 2 - args.warmup * max(max_chunks, 1))):
        sc.run()
    for _ in range(args.warmup):
        sc.run()
    sync_all()

    # ---- timed region: exactly K contractions, join included -------------------------------
    timed_passes = min(args.steps, args.event_passes)
    for x, st in zip(execs, stages):
        x.set_timing(timed_passes * st[3])
    sync_all()
    sampler = PowerSampler(dev) if rank == 0 else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t_hat, log_scale = sc.run()
    sync_all()
    elapsed = time.perf_counter() - t0
    under_load = sampler.stop() if sampler else {}
    # per step: mean over the recorded launches (--event-passes 0, counter-collection runs: none recorded)
    stage_ms = [x.step_ms().astype(np.float64) if timed_passes else np.zeros(st[0].plan.n_steps) for x, st in zip(execs, stages)]
    for x in execs:
        x.set_timing(0)
    # evaluations per stage over all ranks (a stage below no sliced label is evaluated by every rank: counted each time)
    evals_all = torch.tensor([float(st[1]) for st in stages], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
    if pg:
        tmax = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        dist.all_reduce(evals_all, op=dist.ReduceOp.SUM)
    evals_all = evals_all.cpu().numpy()
    # everything the accounting below needs, as plain data: the executors can then be freed (the cross-check and the
    # single-GPU references build their own)
    class StageData:
        pass

    stage_data = []
    for k_, (bc_, n_loc, R_, chunks_) in enumerate(stages):
        sd = StageData()
        sd.infos, sd.tiles = bc_.plan.step_infos(), bc_.executor.step_tiles()
        sd.step_strs = [c[2] for c in bc_.contract_list]
        sd.flops, sd.n_steps, sd.R, sd.launches, sd.evaluated = bc_.plan.flops, bc_.plan.n_steps, R_, chunks_, n_loc
        sd.dep = len(sc.stage_desc[k_]["dep"]) if staged else len(labels)
        stage_data.append(sd)
    sizes, n_total, n_mine, sc_R, sc_outer = sc.sizes, sc.n_total, len(sc.my_slices), sc.R, getattr(sc, "outer", 0)
    sc_unsliced = len(getattr(sc, "unsliced", ()))
    cpu_check = None
    if world == 1 and not args.no_cpu_baseline:
        cpu_check = peps_cpu_baseline(sc, einstr, ops, labels, path, args.cpu_seconds)
    del sc, execs, stages
    torch.cuda.empty_cache()
    cross = None
    if getattr(args, "cross_check", False):
        # the same network through a DIFFERENT set of sliced labels and another tree (the plain-slicing search's plan,
        # executed in stages as well), once, sharded over the same ranks: the two values must agree to 1e-3
        cross = peps_cross_check(args, einstr, shapes, ops, labels, rank, world, local_rank, backend, dev,
                                 float(t_hat), float(log_scale))
        torch.cuda.empty_cache()
    if rank != 0:
        return None

    value = args.steps / elapsed
    # work of one contraction over ALL ranks: every stage's plan x its evaluations (a staged plan evaluates a stage
    # once per joint value of the sliced labels it depends on; plain slicing: one stage, once per slice)
    n_eval_total = [int(x) for x in evals_all]
    flop_sliced = float(sum(sd.flops * n for sd, n in zip(stage_data, n_eval_total)))
    tflops = value * flop_sliced / 1e12

    # ---- roofline of the dominant kernel on rank 0 (per launch of R evaluations, HIP-event durations) ---------
    # algorithmic bytes of a step: both operands and the output once, from the step's own einsum string
    by_kernel, dump = {}, []
    for k_, (sd, ms_) in enumerate(zip(stage_data, stage_ms)):
        infos, tiles, R_, chunks_ = sd.infos, sd.tiles, sd.R, sd.launches
        for s_, (info, step_str) in enumerate(zip(infos, sd.step_strs)):
            lhs, out = step_str.split("->")
            nbytes = 4 * sum(int(np.prod([sizes[x] for x in set(t)])) if t else 1 for t in lhs.split(",") + [out])
            key = (info["kernel"], info["mode_a"], info["mode_b"], tiles[s_][0], tiles[s_][1])
            d = by_kernel.setdefault(key, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
            d["ms"] += ms_[s_] * chunks_               # per contraction (rank 0): the step's mean launch x its launches
            d["flops"] += info["flops"] * R_ * chunks_
            d["bytes"] += nbytes * R_ * chunks_
            d["launches"] += chunks_
            dump.append(dict(info, stage=k_, ms=float(ms_[s_]), bytes=nbytes, replicas=R_, launches=chunks_,
                             tile=list(tiles[s_]), einsum=step_str))
    if args.dump_steps:
        with open(args.dump_steps, "w") as fh:
            json.dump(dump, fh)
    dom_key = max(by_kernel, key=lambda k: by_kernel[k]["ms"])
    dom = by_kernel[dom_key]
    mfma = dom_key[0] in (2, 3)
    # an MFMA step below the ridge (157.3 TFLOP/s / 8 TB/s = 19.7 flop/B) is priced against HBM as well
    ai = dom["flops"] / max(dom["bytes"], 1.0)
    hbm_bound = (not mfma) or ai < PEAK_F32_MFMA_TFLOPS / PEAK_HBM_TBS
    sec = dom["ms"] * 1e-3
    ach_tf = dom["flops"] / sec / 1e12 if sec > 0 else 0.0
    ach_tb = dom["bytes"] / sec / 1e12 if sec > 0 else 0.0
    total_ms = sum(d["ms"] for d in by_kernel.values())
    roofline = {
        "bound": "hbm" if hbm_bound else "mfma",
        "kernel": kernel_label(dom_key),
        "achieved": round(ach_tb * 1e3 if hbm_bound else ach_tf, 3),
        "peak": PEAK_HBM_TBS * 1e3 if hbm_bound else PEAK_F32_MFMA_TFLOPS,
        "unit": "GB/s" if hbm_bound else "TFLOP/s",
        "frac": round(ach_tb / PEAK_HBM_TBS if hbm_bound else ach_tf / PEAK_F32_MFMA_TFLOPS, 4),
        "traffic": None,
        "traffic_source": None,
        "arithmetic_intensity_flop_per_byte": round(ai, 2),
        "launches_per_contraction": dom["launches"],
        "avg_launch_us": round(dom["ms"] * 1e3 / max(dom["launches"], 1), 2),
        "share_of_device_time": round(dom["ms"] / total_ms, 4) if total_ms > 0 else None,
        "algorithmic_bytes_per_launch": dom["bytes"] / max(dom["launches"], 1),
        "flop_per_launch": dom["flops"] / max(dom["launches"], 1),
        "timing_mode": (f"HIP events on the executor's stream around every launch of the first {timed_passes} timed "
                        f"steps (eager launches; later steps replay the same kernels as one hipGraph), rank 0"),
        "kernels": [
            {"kernel": kernel_label(k), "launches": d["launches"], "ms": round(d["ms"], 4),
             "tflops": round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 2) if d["ms"] > 0 else None,
             "tb_per_s": round(d["bytes"] / (d["ms"] * 1e-3) / 1e12, 3) if d["ms"] > 0 else None}
            for k, d in sorted(by_kernel.items(), key=lambda kv: -kv[1]["ms"])],
        "end_to_end_tflops": round(tflops, 3),
        "end_to_end_frac_of_mfma_peak_per_gpu": round(tflops / world / PEAK_F32_MFMA_TFLOPS, 4),
    }
    result = {
        "metric": (f"contractions/sec ({rows}x{cols} PEPS, bond={bond}, phys=2, closed, fp32, stabilised split "
                   f"format, index-sliced over the ranks, one all_gather join)"),
        "value": round(value, 4),
        "unit": "contractions/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (standard normal / sqrt(D), NumPy default_rng(6), identical on every rank)",
        "config": {
            "workload": f"peps_{rows}x{cols}_D{bond}_d2_sliced{n_total}",
            "slices": n_total,
            "slices_per_gpu": n_mine,
            "sliced_labels": len(labels),
            "slices_in_flight_per_launch": sc_R,
            "hbm_free_gib_with_all_stage_buffers_allocated": round(hbm_free / 2 ** 30, 1),
            "outer_labels_walked_on_the_host": sc_outer,
            "labels_the_root_stage_sums_itself": sc_unsliced,
            "execution": ("staged: a stage of the tree is evaluated once per joint value of the sliced labels below it "
                          "(dist.StagedSlicedContraction)" if staged else "plain: every slice repeats the whole path"),
            "stages": [{"depends_on_sliced_labels": sd.dep, "evaluations": n_eval_total[k], "evaluations_on_rank0": sd.evaluated,
                        "steps": sd.n_steps, "flop_per_evaluation": sd.flops} for k, sd in enumerate(stage_data)],
            "work_overhead_vs_unsliced": round(rep["work_overhead"], 3),
            "work_overhead_if_every_slice_repeated_everything": round(rep.get("plain_overhead", rep["work_overhead"]), 3),
            "largest_intermediate_elements": rep["largest_intermediate"],
            "unsliced_largest_intermediate_elements": rep["unsliced_largest_intermediate"],
            "steps_per_slice": sum(sd.n_steps for sd in stage_data),
            "flop_per_contraction_sliced": flop_sliced,
            "multiply_adds_unsliced_path": rep["unsliced_flops"],
            "slice_search_s": round(search_s, 2),
            "parallelism": f"index slicing x{world}, one all_gather of (T_hat, c) per contraction",
        },
        "achieved_tflops": round(tflops, 3),
        "collectives": {"process_group": bool(pg), "backend": (dist.get_backend() if pg else None), "world": world,
                        "join_all_gather_per_contraction": bool(world > 1 or cdist.join_alone())},
        "result": {"t_hat": float(t_hat), "log_scale": float(log_scale)},
        "roofline": roofline,
        "device": dict(device_info(dev), **under_load),
    }
    if cpu_check is not None:
        result["cpu_baseline"] = cpu_check
        if rep["unsliced_largest_intermediate"] <= 2 ** 28:
            result["unsliced_check"] = peps_unsliced_check(einstr, shapes, ops, float(t_hat), float(log_scale), local_rank)
    if cross is not None:
        result["cross_check"] = cross
    if getattr(args, "single_gpu_reference", False):
        result["strong_scaling"] = peps_single_gpu_reference(args, einstr, shapes, ops, labels, path, rep, local_rank,
                                                             world, elapsed / args.steps, float(t_hat), float(log_scale))
    return result


def peps_cross_check(args, einstr, shapes, ops, labels, rank, world, local_rank, backend, dev, t_hat, log_scale):
    """One more contraction of the same network on another slicing - the labels and tree of `dist.sliced_plan` (the
    plain-slicing search: other labels than the staged plan's) - sharded over the same ranks; rank 0 reports the value
    and its distance from the main run's."""
    import torch.distributed as dist

    from contractn_amd import dist as cdist

    box = [None]
    if rank == 0:
        # (under a 2^28-element cap at most: the plain-slicing search's shipped plan, whatever cap the main run had)
        cap2 = min(args.max_intermediate, 2 ** 28) if args.max_intermediate else None
        box[0] = cdist.sliced_plan(einstr, shapes, min_slices=args.slices, max_intermediate=cap2)
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    labels2, path2, rep2 = box[0]
    sc2 = cdist.StagedSlicedContraction(einstr, ops, labels2, optimize=path2, rank=rank, world=world, device=local_rank,
                                        workspace_budget=int(args.workspace_gib * 2 ** 30))
    t0 = time.perf_counter()
    t2, c2 = sc2.run()
    sec = time.perf_counter() - t0
    return {"sliced_labels": len(labels2), "slices": sc2.n_total, "labels_shared_with_the_main_plan": len(set(labels2) & set(labels)),
            "t_hat": float(t2), "log_scale": float(c2), "abs_log_diff": abs(float(c2) - log_scale),
            "ok": bool(float(t2) == t_hat and abs(float(c2) - log_scale) <= 1e-3), "tolerance_abs_log": 1e-3,
            "seconds": round(sec, 3)}


def peps_single_gpu_reference(args, einstr, shapes, ops, labels, path, rep, device, world, sec_per_step, t_hat, log_scale):
    """What ONE GPU does best on the same network, measured by rank 0 in the same run (the other ranks wait), so
    that the strong-scaling figure is quoted against the best single-GPU time and not only against the sliced
    plan run on one GPU: (a) the network UNSLICED on the library's own path where its largest intermediate fits
    (8 x 8, D = 8: 2^27 elements); (b) the sliced plan with all slices on one GPU (what N = 1 of this benchmark
    runs; for D = 16, whose unsliced peak is 2^36 elements, the only single-GPU form).  Both values are checked
    against the sharded result (1e-3)."""
    import torch

    from contractn_amd import dist as cdist
    from contractn_amd.einsum import BatchedContraction

    dev = torch.device("cuda", device)
    out = {"n_gpus": world, "ms_per_contraction": round(sec_per_step * 1e3, 4)}

    def agrees(t1, c1):
        return bool(abs(np.exp(c1 - log_scale) * (t1 * t_hat) - 1.0) <= 1e-3)

    if world == 1:
        sliced_ms = sec_per_step * 1e3
    else:
        make = cdist.SlicedContraction if args.plain_slicing else cdist.StagedSlicedContraction
        sc1 = make(einstr, ops, labels, optimize=path, rank=0, world=1, device=device,
                   workspace_budget=int(args.workspace_gib * 2 ** 30))
        reps = 1 if sec_per_step * world > 1.0 else max(3, args.steps)
        for _ in range(0 if reps == 1 else 3):
            sc1.run()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            t1, c1 = sc1.run()
        torch.cuda.synchronize(dev)
        sliced_ms = (time.perf_counter() - t0) / reps * 1e3
        out["sliced_single_gpu_agrees"] = agrees(float(t1), float(c1))
        del sc1
        torch.cuda.empty_cache()
    out["sliced_single_gpu_ms"] = round(sliced_ms, 4)
    best_ms, best = sliced_ms, "sliced plan, all slices on one GPU"
    if rep["unsliced_largest_intermediate"] <= 2 ** 28:
        bc = BatchedContraction(einstr, shapes, np.float32, optimize="auto", replicas=1, device=device)
        d_ops = [torch.as_tensor(o, device=dev) for o in ops]
        res = torch.zeros(1, device=dev)
        torch.cuda.synchronize(dev)
        launch = bc.executor.make_enqueue([t.data_ptr() for t in d_ops], [res.data_ptr()])
        for _ in range(3):
            launch()
        bc.executor.synchronize()
        reps = max(5, args.steps)
        t0 = time.perf_counter()
        for _ in range(reps):
            launch()
        bc.executor.synchronize()
        unsliced_ms = (time.perf_counter() - t0) / reps * 1e3
        c_u = float(bc.fetch_log_scale()[0])
        out["unsliced_single_gpu_ms"] = round(unsliced_ms, 4)
        out["unsliced_single_gpu_agrees"] = agrees(float(res.cpu()[0]), c_u)
        out["unsliced_flop"] = bc.plan.flops
        if unsliced_ms < best_ms:
            best_ms, best = unsliced_ms, "unsliced, the library's own path (`auto`)"
        bc.executor.close()
    out["best_single_gpu_ms"] = round(best_ms, 4)
    out["best_single_gpu"] = best
    out["speedup_vs_best_single_gpu"] = round(best_ms / (sec_per_step * 1e3), 3)
    out["speedup_vs_sliced_single_gpu"] = round(sliced_ms / (sec_per_step * 1e3), 3)
    return out


def peps_cpu_baseline(sc, einstr, ops, labels, path, budget_s):
    """The NumPy oracle on a bounded sample of the SAME slices (same sliced network, same path), timed on this
    box's host cores and compared slice by slice with what the GPU produced; the whole-contraction rate is
    extrapolated over all slices."""
    from contractn_amd import dist as cdist
    from oracle import cpu_ref

    try:
        from threadpoolctl import threadpool_info

        threads = max((p.get("num_threads", 1) for p in threadpool_info()), default=1)
    except Exception:
        threads = os.cpu_count()
    # per-evaluation values of the root stage (run() itself joins them on the device): evaluation q is the sum of the
    # slices sc.root_members[q] - one slice each unless the root took a sliced label back
    gpu_t, gpu_c = sc.slices_host()
    members = sc.root_members
    wanted, n_groups = set(), 0
    for grp in members:                     # whole evaluations only, within the slice budget
        if n_groups and len(wanted) + len(grp) > 64:
            break
        wanted.update(grp)
        n_groups += 1
    where = {v: q for q, v in enumerate(sc.my_slices)}
    ref, n, spent, clist = {}, 0, 0.0, None
    for vals, sliced_str, sl_ops in cdist.slice_network(einstr, ops, labels):
        q = where.get(tuple(vals))
        if q is None or q not in wanted:
            continue
        if clist is None:
            clist = cpu_ref.contraction_list(sliced_str, [o.shape for o in sl_ops], path)
        t0 = time.perf_counter()
        t_ref, c_ref, _ = cpu_ref.core_contract(sl_ops, clist)
        spent += time.perf_counter() - t0
        ref[q] = (t_ref, c_ref)
        n += 1
        if spent >= budget_s and n >= len(members[0]):
            break
    worst, signs_ok, compared = 0.0, True, 0
    for g_, grp in enumerate(members[:n_groups]):
        if not all(q in ref for q in grp):
            continue
        t_ref, c_ref = cdist.combine_split([ref[q] for q in grp]) if len(grp) > 1 else ref[grp[0]]
        signs_ok = signs_ok and float(gpu_t[g_]) == float(t_ref)
        worst = max(worst, abs(float(gpu_c[g_]) - float(c_ref)))
        compared += 1
    per_contraction = spent / n * sc.n_total
    return {
        "value": round(1.0 / per_contraction, 5),
        "unit": "contractions/s",
        "cores": int(threads),
        "host_cpus": os.cpu_count(),
        "kind": "port",
        "sample": (f"{n} of the {sc.n_total} slices (same sliced network and path, NumPy/OpenBLAS) in {spent:.1f}s; "
                   f"rate extrapolated to all slices"),
        "parity_vs_gpu": {"ok": bool(compared > 0 and signs_ok and worst <= 1e-3), "slices_compared": n,
                          "root_evaluations_compared": compared,
                          "max_abs_log_scale_diff": worst, "tolerance_abs_log": 1e-3, "signs_equal": bool(signs_ok)},
    }


def peps_unsliced_check(einstr, shapes, ops, t_sliced, c_sliced, device):
    """The same network contracted WITHOUT slicing on one GPU (the library's own path search): the sliced,
    joined value must agree to 1e-3."""
    from contractn_amd.einsum import BatchedContraction

    bc = BatchedContraction(einstr, shapes, np.float32, optimize="auto", replicas=1, device=device)
    outs, logs = bc.run_host([ops])
    t_u, c_u = float(outs[0]), float(logs[0])
    rel = abs(np.exp(c_sliced - c_u) * (t_sliced * t_u) - 1.0)
    return {"ok": bool(rel <= 1e-3), "rel_diff": float(rel), "unsliced": [t_u, c_u], "sliced": [t_sliced, c_sliced],
            "unsliced_flop": bc.plan.flops}


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of the dominant kernel from the committed counter passes (profiles/pmc_traffic.json,
    written by profiles/make_summary.py from separate `rocprofv3 --pmc` runs of this same command): NOT measured
    in this run, so it is attached only when that file was taken on the same workload and kernel, and always with
    its source.  Returns (bytes | None, source string | None)."""
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(tfile))
    except (OSError, ValueError):
        return None, None
    if d.get("workload") != workload or d.get("kernel_label") != kernel or d.get("hbm_bytes_per_launch") is None:
        return None, None
    if kernel.split(" ")[0].split("<")[0] not in str(d.get("kernel", "")):     # the counters of another kernel of that run
        return None, None
    return d["hbm_bytes_per_launch"], f"profiles/pmc_traffic.json@{d.get('git_sha', 'unknown')} ({d.get('round', '?')}: {d.get('note', '')})"


def sysfs_card_dir(dev):
    """/sys/class/drm/cardN/device of the HIP device (matched by PCI address; a box shows all its cards)."""
    import glob

    import torch

    try:
        p = torch.cuda.get_device_properties(dev)
        want = "%04x:%02x:%02x." % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        for d in sorted(glob.glob("/sys/class/drm/card*/device")):
            if os.path.basename(os.path.realpath(d)).startswith(want):
                return d
    except Exception:
        pass
    return None


class PowerSampler:
    """Mean shader clock and socket power during the timed region, from the driver's hwmon files (best effort;
    no GPU calls).  The headline run sits at the board's power cap, which is what sets its clock - the nominal
    peak in `roofline.peak` assumes 2.4 GHz."""

    def __init__(self, dev, period=0.05):
        import glob
        import threading

        self.files = {}
        card = sysfs_card_dir(dev)
        for key, pat in (("sclk_under_load_mhz", "freq1_input"), ("power_under_load_w", "power1_average"),
                         ("power_under_load_w", "power1_input")):
            hits = sorted(glob.glob(card + "/hwmon/hwmon*/" + pat)) if card else []
            if hits and key not in self.files:
                self.files[key] = hits[0]
        self.samples = {k: [] for k in self.files}
        self.period, self._stop = period, threading.Event()
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        while not self._stop.is_set():
            for k, f in self.files.items():
                try:
                    self.samples[k].append(float(open(f).read().strip()))
                except Exception:
                    pass
            self._stop.wait(self.period)

    def stop(self):
        self._stop.set()
        self.thread.join()
        out = {}
        for k, v in self.samples.items():
            v = v[len(v) // 4:]                      # drop the ramp at the start of the region
            if v:
                out[k] = round(sum(v) / len(v) / 1e6, 1)   # Hz -> MHz, microwatt -> W
        return out


def device_info(dev):
    """Name, CU count and clocks of the card the numbers were taken on (print them with
    every result).  ``sclk_now_mhz`` is the active level in the driver's sysfs table, best effort."""
    import torch

    props = torch.cuda.get_device_properties(dev)
    info = {"name": props.name, "arch": getattr(props, "gcnArchName", None),
            "compute_units": props.multi_processor_count,
            "max_clock_mhz": round(getattr(props, "clock_rate", 0) / 1e3, 1) or None,
            "hbm_gib": round(props.total_memory / 2**30, 1)}
    try:
        card = sysfs_card_dir(dev)
        for f in ([card + "/pp_dpm_sclk"] if card else []):
            levels = open(f).read().splitlines()
            mhz = lambda ln: int("".join(ch for ch in ln.split(":")[1] if ch.isdigit()))  # noqa: E731
            active = [ln for ln in levels if ln.strip().endswith("*")]
            if active:
                info["sclk_now_mhz"] = mhz(active[0])
                info["max_clock_mhz"] = info["max_clock_mhz"] or max(mhz(ln) for ln in levels if ":" in ln)
                break
        import glob

        for f in (glob.glob(card + "/hwmon/hwmon*/power1_cap") if card else []):
            info["power_cap_w"] = round(float(open(f).read().strip()) / 1e6, 1)
    except Exception:
        pass
    return info


def kernel_label(key):
    """rocprof-visible kernel behind a (kernel kind, modeA, modeB, tile rows, tile columns) group."""
    from contractn_amd.engine import KERNEL_NAMES

    kind, ma, mb, tm, tn = key
    if kind == 2 and tm == 512 and tn == 128:
        return "k_zip64_f32 (two zipper GEMM steps per launch, 64 x 256 outputs per workgroup: the form for 64 ... 127 networks in flight)"
    if kind == 2 and tm == 512:
        return "k_zip_f32 (two zipper GEMM steps per launch: T = E.psi stays in registers, E' = T.phi; 128 x 256 outputs per workgroup)"
    if kind == 2 and tn == 256 and tm in (32, 64):
        return (f"k_zip_lat<Q,{tm}> (two zipper GEMM steps per launch, latency form: 16 values of u x {tm} of m1 "
                f"per workgroup, partial results as {256 // tm} slabs that the next pair adds up while loading)")
    if kind == 2 and tm == 256 and tn > 256:
        return (f"k_mfma_f32_ares<{mb}> (the 256 x 256 left operand resident in registers, {tn // 128} column tiles of 128 per "
                f"workgroup, only the right operand streams)")
    if kind == 2 and tm == 256:
        return f"k_mfma_f32_g<{'8' if tn == 256 else '4'},2,asm,{ma},{mb}> ({tm}x{tn} tiles, LDS-DMA ring)"
    if kind == 3 and tn == 128:
        return "k_mfma_f64_g (128x128 tiles, LDS-DMA ring)"
    if kind in (2, 3) and tm:
        return f"k_{KERNEL_NAMES[kind]}<modeA={ma},modeB={mb}> ({tm}x{tn} tiles)"
    return f"k_{KERNEL_NAMES[kind]}<modeA={ma},modeB={mb}>"


def cpu_baseline(einstr, shapes, path, flat0, offs, numels, budget_s, gpu_t, gpu_c, f64=False):
    """Time the NumPy oracle (oracle/cpu_ref.py, kind 'port') on replica 0's tensors."""
    from oracle import cpu_ref

    try:
        from threadpoolctl import threadpool_info

        threads = max((p.get("num_threads", 1) for p in threadpool_info()), default=1)
    except Exception:
        threads = os.cpu_count()
    host = flat0.cpu().numpy()
    ops = [host[int(offs[i]): int(offs[i]) + numels[i]].reshape(shapes[i]) for i in range(len(shapes))]
    clist = cpu_ref.contraction_list(einstr, shapes, path)
    t_ref, c_ref, _ = cpu_ref.core_contract(ops, clist)  # warm-up + parity sample
    # the CPU's best, not its default: the zipper's GEMMs are small (256 x 256 x 1024), and all 128 OpenBLAS threads
    # of a GPU box oversubscribe them - one contraction per candidate thread count, then the budget on the best
    sweep, limit = {}, None
    if budget_s > 0:
        try:
            from threadpoolctl import threadpool_limits

            for nt in (8, 16, 32, 64, 128):
                if nt > max(threads, 8):
                    break
                with threadpool_limits(limits=nt, user_api="blas"):
                    t0 = time.perf_counter()
                    cpu_ref.core_contract(ops, clist)
                    sweep[nt] = round(1.0 / (time.perf_counter() - t0), 3)
            if sweep:
                threads = max(sweep, key=sweep.get)
                limit = threadpool_limits(limits=threads, user_api="blas")
        except Exception:   # noqa: BLE001 - no threadpoolctl: the library's default thread count, as before
            sweep, limit = {}, None
    n, t0 = 0, time.perf_counter()
    while True:
        cpu_ref.core_contract(ops, clist)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 50:
            break
    if limit is not None:
        limit.restore_original_limits()
    # a scalar in split format is (+-1, log|value|): the sign must match exactly and the value's relative error is
    # |dc| itself (north_star: 1e-3 for fp32, 1e-6 for fp64; asserted an order tighter)
    tol = 1e-7 if f64 else 1e-4
    ok = gpu_t == float(t_ref) and abs(gpu_c - float(c_ref)) <= tol
    return {
        "value": round(n / dt, 3),
        "unit": "contractions/s",
        "cores": int(threads),
        "host_cpus": os.cpu_count(),
        "kind": "port",
        "sample": f"{n} full contractions of replica 0 (same 100-site network, same zipper path, "
                  f"NumPy/OpenBLAS) in {dt:.1f}s on the best of the BLAS thread counts tried",
        "contractions_per_s_by_blas_threads": sweep,
        "parity_vs_gpu": {"ok": bool(ok), "tolerance_abs_log": tol, "gpu": [gpu_t, gpu_c],
                          "cpu": [float(t_ref), float(c_ref)]},
    }


if __name__ == "__main__":
    main()
