#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the BLUEST sample-allocation hot path on MI355X.

metric   : Phi-assemblies/s.  One assembly = for ONE output, build Phi(m) over all K_tot groups, solve for
           V = e0^T Phi^-1 e0 and evaluate grad V for every group.
workload : BASELINE.json configs[3]: n=20 models, groups up to size 5 (K_tot=21699), n_out=8 outputs,
           synthetic Wishart covariances (bluest_amd/synth.py).  A "step" evaluates one allocation vector m for
           all 8 outputs (what one SPG iteration / one MOSAP.variance_GH call does): 2 kernel launches
           (Phi chunks -> fused fold + solve + gradient tiles), inputs resident in HBM, results left in HBM.
N > 1    : one process per GPU.  Three ways to use more than one GPU:
           --shard candidates (default): the unit of work is the evaluation of one allocation vector; independent vectors
             (line-search trial points, integer-projection candidates, budget / tolerance sweeps) are the partition of the
             path that has no exchange at all, so rank r evaluates ITS OWN allocation vector for all outputs in every
             step.  Per-GPU work fixed => "scaling": "weak", value = N * n_out * steps / time; NO data-path collective.
           --shard outputs: ONE allocation vector, rank r assembles outputs r*n_out/N.. (the outputs are independent
             sample-allocation problems, bluest/mosap.py:39); no data-path collective; total work fixed ("strong"), and
             latency-bound at the headline size: one output costs 13 us, eight cost 15 us.
           --shard groups: ONE allocation vector, the group set is sharded, each step all-reduces the partial Phi records
             (n_out*(n^2+2n+1) f64) over RCCL, every rank solves redundantly and evaluates the gradient of its shard
             (bluest_amd/dist.py); "strong"; for K_tot in the 10^5..10^6 range.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_MODELS, KMAX, N_OUT = 20, 5, 8
HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md


def build_outputs(prob):
    """plan description (bluest_amd.plan.Plan) of every output of the synthetic problem"""
    groups = prob["groups"]
    sizes = [len(g) for g in groups]
    return [{"K": prob["kmax"], "sizes": sizes, "groups": groups, "C": prob["C"][o], "mapping": None} for o in range(prob["n_out"])]


def sap_wallclock(prob):
    """second half of BASELINE.json's metric: setup_solver-style wall-clock from the covariances to the continuous optimum
    m* (MOSAP construction: group pseudo-inverses + HBM layouts, then solver="spg" on the GPU), third (warm) repetition"""
    import torch
    from bluest_amd.mosap import MOSAP
    groups, n_out, kmax = prob["groups"], prob["n_out"], prob["kmax"]
    import gc
    res = None
    mos = None
    for rep in range(3):                # the LAST repetition is reported: the first is cold, and the construction right after
        # the first solve of a process sometimes stalls ~80 ms inside one HIP call while the runtime tears down that solve's graphs
        mos = None                      # release the previous plan (hipFree of ~45 MB) outside the timed region
        gc.collect()                    # as timeit does: no cyclic-GC pause (30-70 ms in a process with torch loaded) inside
        torch.zeros(1, device="cuda").cpu()   # a small synchronous copy: HIP finishes tearing down the PREVIOUS solve's graphs
        gc.disable()                         # inside the next blocking copy (~80 ms, at random), which is not this repetition's work                    # a 0.25 s measurement
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
                    prob["costs"], [prob["costs"]] * n_out, verbose=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        gc.enable()
        res = {"setup_s": t1 - t0, "solve_s": t2 - t1, "total_s": t2 - t0, "spg_iterations": int(mos.solver_info["it"]),
               "objective_evaluations": int(mos.solver_info["count"]), "max_variance": float(max(mos.variances(m))),
               "budget": float(prob["budget"]), "solver": "spg (scaled metric, device-resident loop), continuous relaxation"}
    return res


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/*_pmc_traffic.json, written
    by tools/pmc_traffic.py with the guide's gfx950 FETCH_SIZE correction); None if that profile is absent"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        return d["kernels"][kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(prob, seconds=14.0):
    """the reference CPU path on ONE host core, output 0 only (bounded sample), same inputs:
    R  (kind "reference", primary when oracle/_ref travelled): bluest/misc.py:479-495 as executed -- numpy psi@m (BLAS dgemv,
       1 thread), numpy pinv, and the reference's own compiled gradK_c (oracle/_ref, built from cmisc.cpp in the build
       container);
    B1 (kind "port"): the same evaluation in plain C (oracle/bluest_oracle.c: dense psi GEMV + Jacobi pinv + gradK loops);
    B2: the sparse objectiveK_c-style loop (cmisc.cpp:25-40) + the same solve and gradK."""
    from oracle import oracle as orc
    orc.build()
    sap = orc.OracleSAP(prob["C"][0], prob["kmax"], prob["groups"], prob["costs"])
    m = prob["m"][0]
    res = {}

    def rate(fn, budget):
        fn()
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < budget:
            fn()
            reps += 1
        return reps / (time.perf_counter() - t0)

    res["B1"] = rate(lambda: sap.c_variance_GH(m, dense=True), seconds / 3)
    res["B2"] = rate(lambda: sap.c_variance_GH(m, dense=False), seconds / 3)
    kind, value = "port", res["B1"]
    if orc.ref_native() is not None:
        try:
            from threadpoolctl import threadpool_limits
            with threadpool_limits(limits=1):
                res["R"] = rate(lambda: sap.variance_GH_as_executed(m), seconds / 3)
            kind, value = "reference", res["R"]
        except Exception as err:
            sys.stderr.write("cpu_baseline: reference leg unavailable (%s)\n" % err)
    return {"value": value, "unit": "Phi-assemblies/s", "cores": 1, "kind": kind,
            "sample": "output 0 of the n=%d,k_max=%d workload, variance+gradient evaluations repeated for ~%.0f s per leg on one core; "
                      "reference leg = numpy psi@m (1 BLAS thread) + numpy pinv + the reference's compiled gradK_c (bluest/misc.py:479-495)"
                      % (prob["n"], prob["kmax"], seconds / 3),
            "as_executed_reference_value": res.get("R"), "B1_plain_C_dense_psi_value": res["B1"], "B2_plain_C_sparse_value": res["B2"],
            "host_cores_available": os.cpu_count()}


def rocprof_avg_us(kernel):
    """average dispatch duration of `kernel` in the committed rocprofv3 --kernel-trace --stats summary (profiles/), or None"""
    import csv
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    for f in sorted(glob.glob(os.path.join(here, "profiles", "*_kernel_stats.csv")), reverse=True):
        try:
            for row in csv.DictReader(open(f)):
                if kernel + "<" in row["Name"] or kernel + "(" in row["Name"]:
                    return float(row["AverageNs"]) * 1e-3
        except Exception:
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sap", action="store_true", help="skip the SAP wall-clock leg (covariances -> continuous optimum)")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying a hipGraph")
    ap.add_argument("--graph-steps", type=int, default=40, help="steps captured per hipGraph")
    ap.add_argument("--shard", choices=["auto", "candidates", "outputs", "groups"], default="auto")
    ap.add_argument("--n", type=int, default=N_MODELS)
    ap.add_argument("--kmax", type=int, default=KMAX)
    ap.add_argument("--n-out", type=int, default=N_OUT)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from bluest_amd import synth
    from bluest_amd.plan import Plan

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs the torch.distributed.run launcher with --nproc-per-node %d" % (args.gpus, args.gpus))
    # BLUEST_BENCH_BACKEND=gloo + BLUEST_BENCH_SHARE_GPU=1: rehearsal of the N>1 path with all ranks on one GPU (RCCL refuses
    # two ranks per device); the driver's real runs use nccl = RCCL, one rank per GPU.
    backend = os.environ.get("BLUEST_BENCH_BACKEND", "nccl")
    if os.environ.get("BLUEST_BENCH_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    prob = synth.problem(args.n, args.kmax, args.n_out)
    L = prob["K_tot"]
    shard = args.shard
    if shard == "auto":
        shard = "candidates"
    if shard == "outputs" and args.n_out % world:
        raise SystemExit("--shard outputs needs n_out divisible by the number of GPUs")
    my_outputs = list(range(args.n_out))
    if world == 1 or shard == "candidates":
        plan = Plan(args.n, L, build_outputs(prob), max_candidates=1, device=dev)
        sharded = None
    elif shard == "outputs":
        per = args.n_out // world
        my_outputs = list(range(rank * per, (rank + 1) * per))
        plan = Plan(args.n, L, [build_outputs(prob)[o] for o in my_outputs], max_candidates=1, device=dev)
        sharded = None
    else:
        from bluest_amd.dist import ShardedPlan
        sharded = ShardedPlan(args.n, [len(g) for g in prob["groups"]], build_outputs(prob), max_candidates=1, device=dev)
        plan = sharded.plan
    n_out = plan.n_out

    # a small ring of different allocation vectors so that consecutive steps do not repeat the same input
    rng = np.random.RandomState(2024 + (rank if shard == "candidates" else 0))   # candidates: every rank its own vectors
    ring = [torch.from_numpy(prob["m"][0] if rank == 0 or shard != "candidates" else 10.0 * rng.rand(L)).to(dev)] + [torch.from_numpy(10.0 * rng.rand(L)).to(dev) for _ in range(3)]
    var = torch.empty((1, n_out), dtype=torch.float64, device=dev)
    grad = torch.empty((1, plan.grad_len), dtype=torch.float64, device=dev)
    status = torch.empty((1, n_out), dtype=torch.int32, device=dev)
    rec = torch.empty((1, n_out, plan.reclen), dtype=torch.float64, device=dev)

    def step(i):
        m = ring[i % len(ring)]
        if sharded is None:
            plan.eval(m, out=(var, grad, status))
        else:
            v2, g2, st = sharded.eval(m, rec=rec)      # phi -> all-reduce(SUM) over RCCL -> solve -> grad of the shard
            var.copy_(v2)

    # ---- optional hipGraph of one ring cycle (single GPU; RCCL is left eager) -----------------------
    use_graph = (sharded is None) and not args.no_graph
    cycle = max(len(ring), args.graph_steps - args.graph_steps % len(ring))
    graph = None
    if use_graph:
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for i in range(cycle):
                    step(i)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            # thread_local: a process-group watchdog thread may touch the runtime while we capture
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                for i in range(cycle):
                    step(i)
        except Exception as err:   # keep the benchmark alive: eager launches measure the same work
            sys.stderr.write("hipGraph capture failed (%s); falling back to eager launches\n" % err)
            graph = None
            torch.cuda.synchronize()

    def run(nsteps):
        if graph is not None:
            for _ in range(nsteps // cycle):
                graph.replay()
            for i in range(nsteps % cycle):
                step(i)
        else:
            for i in range(nsteps):
                step(i)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    assert bool((status == 0).all()) or sharded is not None
    assert bool(torch.isfinite(var).all())
    weak = world > 1 and shard == "candidates"
    if world > 1 and not weak:
        # self-check outside the timed region: the sharded evaluation equals the unsharded one
        step(0)
        full = Plan(args.n, L, build_outputs(prob), max_candidates=1, device=dev)
        v_full, g_full, _ = full.eval(ring[0])
        v_mine = v_full[0, my_outputs] if sharded is None else v_full[0]
        assert float((var[0] / v_mine - 1).abs().max()) < 1e-10, "sharded evaluation disagrees with the single-GPU one"
        del full

    # ---- per-kernel durations with HIP events on the launch stream (single GPU) -----------------------
    roofline = None
    kern = {}
    if rank == 0:
        ab = synth.algorithmic_bytes(args.n, args.kmax)
        if world == 1:
            from bluest_amd.plan import _stream
            lib, h = plan.lib, plan._h
            m = ring[0]
            var2, vv, st2 = plan.solve(plan.phi(m, out=rec))
            R = 50

            def timed(fn):
                """average duration of one launch: hipGraph of R back-to-back launches on the launch stream, HIP events
                around the replay (this is what rocprofv3 --kernel-trace reports per dispatch: end-to-end on a busy queue)"""
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    fn()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    for _ in range(R):
                        fn()
                g.replay()
                torch.cuda.synchronize()
                ts = []
                for _ in range(20):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); g.replay(); e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e-3 / R)
                return float(np.median(ts))

            t_chunks = timed(lambda: lib.bluest_plan_phi_chunks(h, m.data_ptr(), 1, L, _stream()))
            t_nograd = timed(lambda: plan.eval(m, want_grad=False, out=(var, None, status)))
            t_grad = timed(lambda: plan.grad(vv, st2, out=grad))
            t_step = timed(lambda: plan.eval(m, out=(var, grad, status)))
            kern = {"method": "hipGraph of %d back-to-back launches, HIP events around the replay, median of 20" % R,
                    "k_phi_chunks_us": t_chunks * 1e6, "step_us": t_step * 1e6,
                    "k_solve_grad_us(step - chunks; fused solve+gradient)": (t_step - t_chunks) * 1e6,
                    "separate_path": {"k_phi_chunks+k_solve_from_chunks_us": t_nograd * 1e6,
                                      "k_solve_from_chunks_us(by difference)": (t_nograd - t_chunks) * 1e6,
                                      "k_grad_tiles_us": t_grad * 1e6}}
            # dominant kernel = the longer of the two streaming passes
            t_sg = t_step - t_chunks
            if t_sg >= t_chunks:
                kname, tk, abytes, lbytes = "k_solve_grad", max(t_sg, 1e-9), ab["grad"] * n_out, plan.grad_bytes
            else:
                kname = "k_phi_chunks_shared" if n_out >= 2 else "k_phi_chunks"
                tk, abytes, lbytes = max(t_chunks, 1e-9), ab["phi"] * n_out, plan.phi_bytes
            achieved = abytes / tk
            roofline = {"bound": "hbm", "kernel": kname, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK, "traffic": pmc_traffic(kname),
                        "traffic_source": "profiles/*_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950-corrected)",
                        "algorithmic_bytes_per_launch": abytes, "layout_bytes_per_launch": lbytes, "avg_launch_us": tk * 1e6,
                        "avg_launch_us_note": "HIP events around a chain of launches: one launch = kernel + the ~2.5 us dispatch gap to "
                                              "its successor; rocprofv3 --kernel-trace (begin to end of the dispatch alone) is in rocprof_avg_us",
                        "rocprof_avg_us": rocprof_avg_us(kname),
                        "step": {"algorithmic_bytes": ab["eval"] * n_out, "achieved_GBps": ab["eval"] * n_out * args.steps / elapsed / 1e9,
                                 "frac": ab["eval"] * n_out * args.steps / elapsed / HBM_PEAK}}

    if rank == 0:
        out = {
            "metric": "Phi-assemblies/s", "value": args.steps * args.n_out * (world if weak else 1) / elapsed, "unit": "assemblies/s (one output: Phi(m) over all groups -> V, grad V)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if (weak or world == 1) else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "collective_backend": None if world == 1 else backend,
            "config": {"workload": "n=%d models, groups up to size %d (K_tot=%d), n_out=%d, Wishart covariances; one step = V and grad V of one allocation for all outputs"
                                   % (args.n, args.kmax, L, args.n_out),
                       "n_models": args.n, "k_max": args.kmax, "K_tot": L, "n_out": args.n_out, "batch": 1,
                       "parallelism": "single GPU" if world == 1 else (
                           "candidate axis over %d GPUs: every GPU evaluates its own allocation vector for all %d outputs per step, no data-path collective"
                           % (world, args.n_out) if weak else
                           "outputs sharded over %d GPUs (%d per GPU), no data-path collective" % (world, len(my_outputs)) if sharded is None
                           else "group set sharded over %d GPUs, all-reduce of the Phi records per step" % world),
                       "launch": "hipGraph replay" if graph is not None else "eager"},
            "kernels_us": kern,
        }
        if roofline is not None:
            out["roofline"] = roofline
        if world == 1 and not args.no_sap:
            out["sap_wallclock"] = sap_wallclock(prob)
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(prob)
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
