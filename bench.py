#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the BLUEST sample-allocation hot path on MI355X.

metric   : Phi-assemblies/s.  One assembly = for ONE output, build Phi(m) over all K_tot groups, solve for
           V = e0^T Phi^-1 e0 and evaluate grad V for every group.
N = 1    : BASELINE.json configs[3] (the configuration the metric is quoted on): n=20 models, groups up to size 5
           (K_tot=21699), n_out=8 outputs, synthetic Wishart covariances (bluest_amd/synth.py).  A "step" evaluates ONE allocation
           vector m for all 8 outputs (what one SPG iteration / one MOSAP.variance_GH call does), inputs resident in HBM,
           results left in HBM.  `value` is batch 1; `batched` reports the same workload with 4 / 16 allocation vectors per launch.
N > 1    : BASELINE.json configs[4]: n=25, groups up to size 6 (K_tot=245505), single output, the GROUP SET sharded over
           the N GPUs (bluest_amd/dist.py): every step each rank assembles the partial Phi of its groups, ONE all-reduce(SUM) of
           the Phi record (n^2+2n+1 f64) over RCCL/xGMI, every rank solves redundantly and evaluates the gradient of its shard.
           Total work fixed => "scaling": "strong".  The line also carries the same configuration on ONE GPU
           (`single_gpu_value`) and N independent replicas of it (`replica_value`, no collective).
           --shard candidates|outputs keep the headline configuration and split it by allocation vector / by output.
timing   : W warm-up steps, then `repeats` x K steps between barrier + synchronize; `repeats` is chosen so that the timed region
           lasts >= 50 ms (K as passed; ms_per_step = elapsed / (repeats*K)).

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

HEADLINE = (20, 5, 8)      # n, k_max, n_out  -- BASELINE.json configs[3]
SHARDED = (25, 6, 1)       # BASELINE.json configs[4]
HBM_PEAK = 8.0e12          # B/s, /opt/skills/guides/MI355X_MICROARCH.md
MIN_TIMED_S = 0.05


def build_outputs(prob):
    """plan description (bluest_amd.plan.Plan) of every output of the synthetic problem"""
    groups = prob["groups"]
    sizes = [len(g) for g in groups]
    return [{"K": prob["kmax"], "sizes": sizes, "groups": groups, "C": prob["C"][o], "mapping": None} for o in range(prob["n_out"])]


def sap_wallclock(prob, reps=4):
    """second half of BASELINE.json's metric: wall-clock from the covariances to the continuous optimum m* (MOSAP construction:
    group pseudo-inverses + HBM layouts, then solver="spg" on the GPU).  Nothing is hidden: repetition 0 is the cold one (first
    launches of every kernel), the others are warm; releasing a problem (its plans go back to the library's block cache) plus an
    explicit full collection is timed as `release_s` of the repetition that created it, the garbage collector stays on."""
    import gc
    import torch
    from bluest_amd.mosap import MOSAP
    groups, n_out, kmax = prob["groups"], prob["n_out"], prob["kmax"]
    rows = []
    for rep in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
                    prob["costs"], [prob["costs"]] * n_out, verbose=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        row = {"setup_s": t1 - t0, "solve_s": t2 - t1, "total_s": t2 - t0, "spg_iterations": int(mos.solver_info["it"]),
               "objective_evaluations": int(mos.solver_info["count"]), "max_variance": float(max(mos.variances(m))),
               "setup_phases_ms": {k: round(v, 3) for k, v in mos.setup_phases.items()},
               "method": mos.solver_info.get("method", "spg"), "certified_gap": float(mos.solver_info.get("certified_gap", float("nan"))),
               "full_evaluations": int(mos.solver_info.get("fevals", 0)), "rounds": int(mos.solver_info.get("rounds", 0))}
        # everything the solve does (full-problem and working-set steps, restricted plans, pricing, host checks) per trial point
        row["solve_us_per_evaluation"] = row["solve_s"] / max(row["objective_evaluations"], 1) * 1e6
        t3 = time.perf_counter()
        mos = None
        gc.collect()
        torch.cuda.synchronize()
        row["release_s"] = time.perf_counter() - t3
        rows.append(row)
    # max_model_samples (bluest/sap.py:222-240): the three most sampled models capped at half of what the free optimum gives them
    capped = None
    try:
        mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
                    prob["costs"], [prob["costs"]] * n_out, verbose=False)
        m_free = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
        usage = np.array([float(mos.ES[i] @ m_free) for i in range(prob["n"])])
        caps = np.full(prob["n"], np.inf)
        for i in np.argsort(-usage)[:3]:
            caps[i] = max(1.0, np.floor(0.5 * usage[i]))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m_cap = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True, max_model_samples=caps)
        torch.cuda.synchronize()
        capped = {"capped_solve_s": time.perf_counter() - t0, "capped_models": [int(i) for i in np.flatnonzero(np.isfinite(caps))],
                  "caps": [float(caps[i]) for i in np.flatnonzero(np.isfinite(caps))], "method": mos.solver_info.get("method", "spg"),
                  "certified_gap": float(mos.solver_info.get("certified_gap", float("nan"))),
                  "max_variance": float(max(mos.variances(m_cap))), "max_variance_free": float(max(mos.variances(m_free))),
                  "cap_usage": [float(mos.ES[i] @ m_cap / caps[i]) for i in np.flatnonzero(np.isfinite(caps))],
                  "caps_method": mos.solver_info.get("caps", "first-order loop over the capped set (bluest_amd/capped.py)"),
                  "note": "includes the unconstrained solve that finds the caps violated (MOSAP.solve does both); caps_method is the "
                          "path enforce_sample_caps took: free solves under shifted costs first, rows of the master's KKT system second, "
                          "the first-order loop third"}
        mos = None
    except Exception as err:      # the headline line must not depend on this leg
        capped = {"error": repr(err)[:300]}
    warm = rows[1:]
    med = sorted(warm, key=lambda r: r["total_s"])[len(warm) // 2]
    return {"cold_s": rows[0]["total_s"], "max_model_samples": capped, "cold": rows[0], "warm_total_s": med["total_s"], "warm": med,
            "warm_all_total_s": [r["total_s"] for r in warm], "release_s_all": [r["release_s"] for r in rows],
            "budget": float(prob["budget"]),
            "solver": "solver=\"spg\" with the second-order finish: multiplicative phase on all groups, column generation with a "
                      "single-workgroup Newton master, certified duality gap; continuous relaxation",
            "note": "total_s = set-up + solve; cold = first repetition of the process; warm = median of the following %d; release_s = "
                    "dropping the problem afterwards + a full Python collection; spg_iterations = Newton iterations of the masters, "
                    "objective_evaluations = evaluations on all groups + inside the masters" % len(warm)}


def pmc_traffic(kernel, cfg):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of THIS configuration (profiles/*_pmc_traffic.json for
    the headline workload, *_pmc_traffic_n<n>_k<k>_o<o>.json for the others; written by tools/pmc_traffic.py with the guide's gfx950
    FETCH_SIZE correction); None when no such profile is committed"""
    import glob
    suffix = "" if tuple(cfg) == HEADLINE else "_n%d_k%d_o%d" % tuple(cfg)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic%s.json" % suffix)), reverse=True):
        try:
            return json.load(open(f))["kernels"][kernel]["hbm_bytes_per_launch"]
        except Exception:
            continue
    return None


def rocprof_avg_us(kernel, cfg):
    """average dispatch duration of `kernel` in the newest committed rocprofv3 --kernel-trace --stats summary for THIS
    configuration: profiles/<tag>_kernel_stats.csv for the headline workload, profiles/<tag>_kernel_stats_n<n>_k<k>_o<o>.csv for
    the others; None when no such profile is committed"""
    import csv
    import glob
    suffix = "" if tuple(cfg) == HEADLINE else "_n%d_k%d_o%d" % tuple(cfg)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernel_stats%s.csv" % suffix)), reverse=True):
        try:
            for row in csv.DictReader(open(f)):
                if kernel + "<" in row["Name"] or kernel + "(" in row["Name"]:
                    return float(row["AverageNs"]) * 1e-3
        except Exception:
            continue
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
    orc.select_fast_math()          # the reference's compiler flags for the timed legs (the checker uses the strict build)
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
    orc.select_strict()            # the checker's build again for whatever follows in this process
    return {"value": value, "unit": "Phi-assemblies/s", "cores": 1, "kind": kind,
            "sample": "output 0 of the n=%d,k_max=%d workload, variance+gradient evaluations repeated for ~%.0f s per leg on one core; "
                      "reference leg = numpy psi@m (1 BLAS thread) + numpy pinv + the reference's compiled gradK_c (bluest/misc.py:479-495)"
                      % (prob["n"], prob["kmax"], seconds / 3),
            "as_executed_reference_value": res.get("R"), "B1_plain_C_dense_psi_value": res["B1"], "B2_plain_C_sparse_value": res["B2"],
            "host_cores_available": os.cpu_count()}


def sap_cpu_baseline(prob, gpu_max_variance, solve_seconds=10.0, setup_cap_s=14.0):
    """the SAP wall-clock half of the metric on ONE host core, as the reference executes it:
    set-up  = per output the constructor of bluest/sap.py:53-129 -- K_tot calls of numpy.linalg.pinv (:69-79), the indicator
              vectors (:89-94), the dense psi (:129) -- through the oracle's restatement (oracle.OracleSAP), every output as
              bluest/mosap.py:36-47 does; when one output predicts more than setup_cap_s for all of them, only the first ones are
              built and the figure is scaled to n_out (the sample is stated);
    solve   = the reference's spg() (bluest/spg.py:39-132) on x = cost*m/B over the simplex with the reference's evaluation as the
              callback (misc.py:479-495: max over the outputs, gradient of the largest), stopped after solve_seconds; the best
              objective it reached is reported next to the GPU solver's optimum.  A reported baseline, not a target."""
    from oracle import oracle as orc
    orc.build()
    from threadpoolctl import threadpool_limits
    n_out, kmax, groups, w, B = prob["n_out"], prob["kmax"], prob["groups"], prob["costs"], prob["budget"]
    with threadpool_limits(limits=1):
        saps, t_setup = [], []
        for o in range(n_out):
            t0 = time.perf_counter()
            saps.append(orc.OracleSAP(prob["C"][o], kmax, [g.copy() for g in groups], w))
            t_setup.append(time.perf_counter() - t0)
            if sum(t_setup) / len(t_setup) * n_out > setup_cap_s and sum(t_setup) > 0.5 * setup_cap_s:
                break
        built = len(saps)
        setup_s = sum(t_setup) / built * n_out
        # time-boxed reference SPG on the outputs that were built (all of them unless the cap cut the set-up short)
        use_ref = orc.ref_native() is not None
        if use_ref:
            orc.select_fast_math()
        scale = B / w
        state = {"best": np.inf, "evals": 0, "t0": time.perf_counter()}

        class _TimeUp(Exception):
            pass

        def both(x):
            if time.perf_counter() - state["t0"] > solve_seconds:
                raise _TimeUp()
            m = scale * x
            vals, grads = [], []
            for sp in saps:
                # model 0 not sampled by any group with |m| > 1e-6: the reference's variance() asserts (bluest/misc.py:470) and its
                # solve() gives up (sap.py:209-213); variance_GH would silently return another model's variance (misc.py:490)
                if not (np.abs(m[np.asarray(sp.e) > 0]) > 1.0e-6).any():
                    vals.append(np.inf); grads.append(None)
                    continue
                try:
                    v, g, _ = sp.variance_GH_as_executed(m) if use_ref else sp.variance_GH(m, nohess=True)
                except AssertionError:
                    v, g = np.inf, None
                vals.append(v); grads.append(g)
            state["evals"] += 1
            o = int(np.argmax(vals))
            if np.isfinite(vals[o]):
                state["best"] = min(state["best"], float(vals[o]))
            f0 = state.setdefault("f0", float(vals[o]))           # objective normalised by its value at the start (as the GPU loop does)
            return float(vals[o]) / f0, (None if grads[o] is None else grads[o] * scale / f0)
        cache = {}

        def feval(x):
            f, g = both(x)
            cache["x"], cache["g"] = x.copy(), g
            return f

        def geval(x):
            if cache.get("x") is not None and np.array_equal(cache["x"], x) and cache["g"] is not None:
                return cache["g"]
            return both(x)[1]
        it = None
        try:
            res = orc.spg(feval, geval, orc.simplex_projection, np.full(len(w), 1.0 / len(w)), eps=1.0e-10, maxit=10 ** 6, lmbda_max=1.0e3)
            it = int(res["it"])
        except _TimeUp:
            pass
        orc.select_strict()
    return {"cores": 1, "kind": "reference" if use_ref else "port",
            "setup_s": setup_s, "setup_outputs_built": built, "setup_s_per_output": [round(t, 4) for t in t_setup],
            "setup_sample": "bluest/sap.py:53-129 per output (K_tot numpy pinv + indicator vectors + dense psi), %d of %d outputs built%s"
                            % (built, n_out, "" if built == n_out else ", scaled to all outputs"),
            "solve_time_box_s": solve_seconds, "solve_evaluations": state["evals"], "solve_spg_iterations_completed": it,
            "solve_outputs": built, "max_variance_reached": state["best"], "gpu_max_variance": gpu_max_variance,
            "excess_over_gpu_optimum": (state["best"] / gpu_max_variance - 1.0) if (built == n_out and np.isfinite(state["best"])) else None,
            "solve_sample": "bluest/spg.py:39-132 from the uniform allocation, callbacks = max_o V_o and the gradient of the largest output "
                            "(bluest/misc.py:479-495 as executed), stopped after the time box; best objective seen"}


class Stepper(object):
    """W warm-up steps, then repeats x K timed steps; a hipGraph of `cycle` consecutive steps is replayed when that is possible
    (cycle <= K, so replays really happen inside the timed region) and everything else is launched eagerly -- the line says which"""

    def __init__(self, torch, step, ring_len, steps, graph_steps, allow_graph, barrier, agree=None):
        self.torch, self.step, self.barrier = torch, step, barrier
        self.agree = agree if agree is not None else (lambda v: v)      # N > 1: max over the ranks, so all of them loop alike
        self.steps = steps
        self.cycle = 0
        self.graph = None
        cycle = min(graph_steps, steps)
        cycle -= cycle % ring_len
        if allow_graph and cycle >= ring_len:
            try:
                from bluest_amd._lib import capture_guard
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    for i in range(cycle):
                        step(i)
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with capture_guard():
                    # thread_local: a process-group watchdog thread may touch the runtime while we capture
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        for i in range(cycle):
                            step(i)
                self.graph, self.cycle = g, cycle
            except Exception as err:   # keep the benchmark alive: eager launches measure the same work
                sys.stderr.write("hipGraph capture failed (%s); falling back to eager launches\n" % err)
                self.graph = None
                torch.cuda.synchronize()

    def run(self, nsteps):
        done = 0
        if self.graph is not None:
            for _ in range(nsteps // self.cycle):
                self.graph.replay()
            done = nsteps - nsteps % self.cycle
        for i in range(done, nsteps):
            self.step(i)

    def launch_label(self):
        if self.graph is None:
            return "eager"
        replayed = self.steps - self.steps % self.cycle
        return "hipGraph replay of %d-step graphs (%d of %d steps), rest eager" % (self.cycle, replayed, self.steps)

    def timed(self, warmup):
        """returns (seconds per step, repeats): the K-step block is repeated until the timed region is >= MIN_TIMED_S"""
        self.run(warmup)
        self.barrier()
        t0 = time.perf_counter()
        self.run(self.steps)
        self.barrier()
        pilot = time.perf_counter() - t0
        repeats = int(self.agree(max(1, int(np.ceil(MIN_TIMED_S / max(pilot, 1e-9))))))
        while True:
            self.barrier()
            t0 = time.perf_counter()
            for _ in range(repeats):
                self.run(self.steps)
            self.barrier()
            elapsed = time.perf_counter() - t0
            # the pilot block carries the synchronisation latency, so short blocks under-estimate the repeats: go again.  Every
            # decision is taken on a value all ranks agree on (a different repeat count per rank would unbalance the collectives)
            slowest = float(self.agree(elapsed))
            if slowest >= MIN_TIMED_S or repeats >= 10 ** 6:
                return elapsed / (repeats * self.steps), repeats
            repeats = int(np.ceil(repeats * 1.25 * MIN_TIMED_S / max(slowest, 1e-9)))


def chain_time(torch, fn, R=50, reps=20):
    """average duration of one launch sequence fn(): hipGraph of R back-to-back calls on the launch stream, HIP events around
    the replay (one "launch" = the kernels plus the dispatch gap to the successor), median of `reps` replays"""
    from bluest_amd._lib import capture_guard
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with capture_guard():
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(R):
                fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3 / R)
    return float(np.median(ts))


def eager_time(torch, fn, R=50, reps=7):
    """average duration of one call of fn() launched eagerly R times between two HIP events on the current stream (collectives
    of RCCL are not captured into graphs here), median of `reps`"""
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(R):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3 / R)
    return float(np.median(ts))


def launch_ranks(n, argv):
    """run this script as n ranks under torch.distributed.run (rendezvous on 127.0.0.1, a free port), stdout/stderr passed
    through; returns the launcher's exit code"""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL and the peer-write mailboxes need it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    # rank 0's JSON line goes to stdout, whatever else the ranks print there (gloo's connection banner, library notices) to stderr:
    # the caller reads ONE line
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in child.stdout:
        (sys.stdout if line.startswith('{"metric"') else sys.stderr).write(line)
        sys.stdout.flush()
    return child.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sap", action="store_true", help="skip the SAP wall-clock leg (covariances -> continuous optimum)")
    ap.add_argument("--no-batched", action="store_true", help="skip the batched-throughput leg")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying a hipGraph")
    ap.add_argument("--graph-steps", type=int, default=40, help="steps captured per hipGraph (capped at --steps)")
    ap.add_argument("--shard", choices=["auto", "candidates", "outputs", "groups"], default="auto")
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--kmax", type=int, default=None)
    ap.add_argument("--n-out", type=int, default=None)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly (`python bench.py --gpus N`): this process becomes the launcher -- it starts the N ranks as a CHILD
        # (torch.distributed.run, one process per GPU), relays rank 0's JSON line and exits with the child's code.  Nothing here has
        # touched the GPU (torch is not even imported yet) and nothing is exec'ed.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from bluest_amd import synth
    from bluest_amd.plan import Plan

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks (--nproc-per-node must equal --gpus)" % (args.gpus, world))
    # BLUEST_BENCH_BACKEND=gloo + BLUEST_BENCH_SHARE_GPU=1: rehearsal of the N>1 path with all ranks on one GPU (RCCL refuses
    # two ranks per device); the driver's real runs use nccl = RCCL, one rank per GPU.
    backend = os.environ.get("BLUEST_BENCH_BACKEND", "nccl")
    if os.environ.get("BLUEST_BENCH_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks_seen = 1
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        ones = torch.ones(1, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        ranks_seen = int(ones[0])

    shard = args.shard
    if shard == "auto":
        shard = "groups" if world > 1 else "candidates"
    base = SHARDED if (world > 1 and shard == "groups") else HEADLINE
    n = args.n if args.n is not None else base[0]
    kmax = args.kmax if args.kmax is not None else base[1]
    n_out_all = args.n_out if args.n_out is not None else base[2]
    prob = synth.problem(n, kmax, n_out_all)
    L = prob["K_tot"]
    if shard == "outputs" and n_out_all % world:
        raise SystemExit("--shard outputs needs n_out divisible by the number of GPUs")
    BATCHES = (4, 16)
    max_cand = max(BATCHES) if (world == 1 and not args.no_batched) else 1
    my_outputs = list(range(n_out_all))
    sharded = None
    if world == 1 or shard == "candidates":
        plan = Plan(n, L, build_outputs(prob), max_candidates=max_cand, device=dev)
    elif shard == "outputs":
        per = n_out_all // world
        my_outputs = list(range(rank * per, (rank + 1) * per))
        plan = Plan(n, L, [build_outputs(prob)[o] for o in my_outputs], max_candidates=1, device=dev)
    else:
        from bluest_amd.dist import ShardedPlan
        sharded = ShardedPlan(n, [len(g) for g in prob["groups"]], build_outputs(prob), max_candidates=1 if args.no_batched else max(BATCHES),
                              device=dev)
        plan = sharded.plan
    n_out = plan.n_out

    # a small ring of different allocation vectors so that consecutive steps do not repeat the same input
    weak = world > 1 and shard == "candidates"
    rng = np.random.RandomState(2024 + (rank if weak else 0))   # candidates: every rank its own vectors
    first = prob["m"][0] if (rank == 0 or not weak) else 10.0 * rng.rand(L)
    ring = [torch.from_numpy(v).to(dev) for v in [first] + [10.0 * rng.rand(L) for _ in range(3)]]
    var = torch.empty((1, n_out), dtype=torch.float64, device=dev)
    grad = torch.empty((1, plan.grad_len), dtype=torch.float64, device=dev)
    status = torch.empty((1, n_out), dtype=torch.int32, device=dev)
    rec = torch.empty((1, n_out, plan.reclen), dtype=torch.float64, device=dev)

    def step(i):
        m = ring[i % len(ring)]
        if sharded is None:
            plan.eval(m, out=(var, grad, status))
        else:
            sharded.eval(m, rec=rec, out=(var, grad, status))   # phi -> all-reduce(SUM) over RCCL -> solve -> grad of the shard

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # RCCL launches are left eager (a collective inside a captured graph is a different code path of the library); the peer-write
    # exchange is a plain kernel and is captured with the rest of the step
    graphable = (sharded is None) or (sharded.exchange is not None)
    def agree(v):
        """max over the ranks (host value); identity on one GPU"""
        if world == 1:
            return v
        t = torch.tensor([float(v)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    stepper = Stepper(torch, step, len(ring), args.steps, args.graph_steps, graphable and not args.no_graph, barrier, agree)
    sec_per_step, repeats = stepper.timed(args.warmup)
    if world > 1:
        t = torch.tensor([sec_per_step], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sec_per_step = float(t[0])
    assert bool((status == 0).all())
    assert bool(torch.isfinite(var).all())

    extra = {}
    if world > 1 and not weak:
        # outside the timed region: (1) the sharded evaluation equals the unsharded one; (2) the same configuration on ONE GPU
        # and as N independent replicas (every rank the whole problem, no collective)
        step(0)
        torch.cuda.synchronize()
        full = Plan(n, L, build_outputs(prob), max_candidates=1, device=dev)
        v_full, g_full, st_full = full.eval(ring[0])
        v_mine = v_full[0, my_outputs] if sharded is None else v_full[0]
        assert float((var[0] / v_mine - 1).abs().max()) < 1e-10, "sharded evaluation disagrees with the single-GPU one"
        fv = torch.empty_like(v_full); fg = torch.empty_like(g_full); fs = torch.empty_like(st_full)
        full_step = lambda i: full.eval(ring[i % len(ring)], out=(fv, fg, fs))      # noqa: E731
        rep_stepper = Stepper(torch, full_step, len(ring), args.steps, args.graph_steps, not args.no_graph, barrier, agree)
        rep_sec, _ = rep_stepper.timed(args.warmup)
        t = torch.tensor([rep_sec], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        extra["replica_value"] = world * n_out_all / float(t[0])
        extra["replica_note"] = "%d independent replicas of the same configuration (each GPU the whole group set, no collective); max over ranks" % world
        # one GPU alone, the others idle at the barrier
        solo = torch.zeros(1, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if rank == 0:
            solo_stepper = Stepper(torch, full_step, len(ring), args.steps, args.graph_steps, not args.no_graph, lambda: torch.cuda.synchronize())
            solo[0] = solo_stepper.timed(args.warmup)[0]
        dist.all_reduce(solo)
        extra["single_gpu_value"] = n_out_all / float(solo[0])
        extra["single_gpu_ms_per_step"] = float(solo[0]) * 1e3
        del full
        if sharded is not None and not args.no_batched:
            # ... and the same GPU alone on the batched step (the reference the batched sharded step is to be compared with)
            sb = torch.zeros(len(BATCHES), dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            if rank == 0:
                fullb = Plan(n, L, build_outputs(prob), max_candidates=max(BATCHES), device=dev)
                for bi, nc in enumerate(BATCHES):
                    rngb = np.random.RandomState(77 + nc)
                    rings = [torch.from_numpy(10.0 * rngb.rand(nc, L)).to(dev) for _ in range(2)]
                    bv = torch.empty((nc, n_out), dtype=torch.float64, device=dev)
                    bgr = torch.empty((nc, fullb.grad_len), dtype=torch.float64, device=dev)
                    bs = torch.empty((nc, n_out), dtype=torch.int32, device=dev)
                    bstep = lambda i, rings=rings, bv=bv, bgr=bgr, bs=bs: fullb.eval(rings[i % 2], out=(bv, bgr, bs))   # noqa: E731
                    bst = Stepper(torch, bstep, 2, max(40, args.steps // nc), args.graph_steps, not args.no_graph, lambda: torch.cuda.synchronize())
                    sb[bi] = bst.timed(max(4, args.warmup // nc))[0]
                del fullb
            dist.all_reduce(sb)
            extra["single_gpu_batched"] = {"n_cand=%d" % nc: {"value": nc * n_out_all / float(sb[bi]), "ms_per_step": float(sb[bi]) * 1e3}
                                           for bi, nc in enumerate(BATCHES)}
        if sharded is not None and not args.no_sap:
            # the solve of the same configuration over the sharded plan (collective, so every rank runs it) -- outside the timed region
            try:
                from bluest_amd.dist import sharded_spg
                rows = []
                for rep in range(2):
                    barrier()
                    t0 = time.perf_counter()
                    m_sh, info = sharded_spg(sharded, prob["costs"], budget=prob["budget"])
                    barrier()
                    rows.append(time.perf_counter() - t0)
                vs, _, _ = sharded.eval(torch.from_numpy(m_sh).to(dev), want_grad=False)
                extra["sharded_solve"] = {"cold_s": rows[0], "warm_s": rows[1], "newton_iterations": int(info["it"]),
                                          "full_evaluations": int(info.get("fevals", 0)), "rounds": int(info.get("rounds", 0)),
                                          "us_per_full_evaluation_all_in": rows[1] / max(int(info.get("fevals", 1)), 1) * 1e6,
                                          "certified_gap": float(info.get("certified_gap", float("nan"))),
                                          "max_variance": float(vs.max()), "method": info.get("method", "spg"),
                                          "loop": "second-order finish over the sharded plan (colgen.colgen_solve): sharded multiplicative "
                                                  "phase (one record exchange per evaluation, nothing of length K_tot exchanged), "
                                                  "collective column generation, redundant bit-identical masters",
                                          "note": "set-up not included (the sharded plan is the one timed above); compare sap_wallclock of "
                                                  "the single-GPU run of this configuration: profiles/*_bench_n25_k6_o1.json"}
            except Exception as err:      # the headline line must not depend on this leg
                extra["sharded_solve"] = {"error": repr(err)[:300]}

    # ---- per-kernel durations with HIP events on the launch stream -----------------------------------
    roofline = None
    kern = {}
    batched = None
    ab = synth.algorithmic_bytes(n, kmax)
    if world > 1 and sharded is not None:
        # every rank times its own launches (collective: the exchange needs all ranks); rank 0 reports the MAX over ranks
        from bluest_amd.plan import _stream
        lib, h = plan.lib, plan._h
        m0 = ring[0]
        timer = chain_time if graphable and not args.no_graph else eager_time
        t_chunks = timer(torch, lambda: lib.bluest_plan_phi_chunks(h, m0.data_ptr(), 1, L, _stream()))
        t_phi = timer(torch, lambda: plan.phi(m0, out=rec))
        t_xchg = timer(torch, lambda: sharded.reduce_records(rec))
        plan.phi(m0, out=rec); sharded.reduce_records(rec)
        t_sg = timer(torch, lambda: plan.solve_grad(rec, out=(var, grad, status)))
        tt = torch.tensor([t_chunks, t_phi, t_xchg, t_sg], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_chunks, t_phi, t_xchg, t_sg = (float(v) for v in tt)
        rec_bytes = int(n_out * plan.reclen * 8)
        kern = {"method": ("hipGraph of 50 back-to-back launches" if timer is chain_time else "50 eager launches") + ", HIP events, median; max over ranks",
                "k_phi_chunks_us(shard)": t_chunks * 1e6, "phi_pass_us(chunks + fold to record)": t_phi * 1e6,
                "exchange_us": t_xchg * 1e6, "k_solve_grad_us(redundant solve + shard gradient)": t_sg * 1e6,
                "exchange": {"algorithm": sharded.exchange_name, "bytes_per_rank_per_step": rec_bytes,
                             "doubles": int(n_out * plan.reclen)}}
        if not args.no_batched:
            # the batched sharded step (sharding pays once the candidate axis is batched across the exchange):
            # nc allocation vectors per step, the partial records of all of them in ONE exchange of nc * n_out * reclen doubles,
            # then nc redundant solves and the shard's gradients.  Collective: every rank runs it.
            batched = {}
            for nc in BATCHES:
                rngb = np.random.RandomState(77 + nc)            # the same vectors on every rank
                rings = [torch.from_numpy(10.0 * rngb.rand(nc, L)).to(dev) for _ in range(2)]
                bv = torch.empty((nc, n_out), dtype=torch.float64, device=dev)
                bgr = torch.empty((nc, plan.grad_len), dtype=torch.float64, device=dev)
                bs = torch.empty((nc, n_out), dtype=torch.int32, device=dev)
                brec = torch.empty((nc, n_out, plan.reclen), dtype=torch.float64, device=dev)
                bstep = lambda i, rings=rings, bv=bv, bgr=bgr, bs=bs, brec=brec: sharded.eval(rings[i % 2], rec=brec, out=(bv, bgr, bs))   # noqa: E731
                bst = Stepper(torch, bstep, 2, max(40, args.steps // nc), args.graph_steps, graphable and not args.no_graph, barrier, agree)
                sec, rpt = bst.timed(max(4, args.warmup // nc))
                sec = float(agree(sec))
                assert bool((bs == 0).all())
                t_bx = timer(torch, lambda brec=brec: sharded.reduce_records(brec))
                t_bx = float(agree(t_bx))
                batched["n_cand=%d" % nc] = {"value": nc * n_out_all / sec, "ms_per_step": sec * 1e3, "steps": bst.steps, "repeats": rpt,
                                             "launch": bst.launch_label(), "exchange_us": t_bx * 1e6,
                                             "exchange_bytes_per_rank_per_step": int(nc * n_out * plan.reclen * 8),
                                             "exchanges_per_step": 1}
            batched["note"] = ("group set sharded as in the headline step, nc allocation vectors per step with ONE exchange of the nc "
                               "records; value = assemblies/s of the whole job (max over ranks)")
        # roofline of the shard's dominant streaming kernel: algorithmic bytes of the shard (1/world of the pass: shards are
        # balanced by sum k^2) over its launch time
        if t_sg >= t_chunks:
            kname, tk, abytes = "k_solve_grad", max(t_sg, 1e-9), ab["grad"] * n_out / world
        else:
            kname, tk, abytes = "k_phi_chunks", max(t_chunks, 1e-9), ab["phi"] * n_out / world
        roofline = {"bound": "hbm", "kernel": kname + " (one shard)", "achieved": abytes / tk / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": abytes / tk / HBM_PEAK, "traffic": None, "algorithmic_bytes_per_launch": abytes, "avg_launch_us": tk * 1e6,
                    "clocks": "this run, per GPU; one launch = kernel + dispatch gap to its successor",
                    "step": {"algorithmic_bytes_all_gpus": ab["eval"] * n_out, "achieved_GBps_all_gpus": ab["eval"] * n_out / sec_per_step / 1e9,
                             "frac_of_aggregate_peak": ab["eval"] * n_out / sec_per_step / (HBM_PEAK * world)}}
    if rank == 0 and world == 1:
        from bluest_amd.plan import _stream
        lib, h = plan.lib, plan._h
        m = ring[0]
        var2, vv, st2 = plan.solve(plan.phi(m, out=rec))
        t_chunks = chain_time(torch, lambda: lib.bluest_plan_phi_chunks(h, m.data_ptr(), 1, L, _stream()))
        t_nograd = chain_time(torch, lambda: plan.eval(m, want_grad=False, out=(var, None, status)))
        t_grad = chain_time(torch, lambda: plan.grad(vv, st2, out=grad))
        t_step = chain_time(torch, lambda: plan.eval(m, out=(var, grad, status)))
        kern = {"method": "hipGraph of 50 back-to-back launches, HIP events around the replay, median of 20",
                "k_phi_chunks_us": t_chunks * 1e6, "step_us": t_step * 1e6,
                "k_solve_grad_us(step - chunks; fused solve+gradient)": (t_step - t_chunks) * 1e6,
                "separate_path": {"k_phi_chunks+k_solve_from_chunks_us": t_nograd * 1e6,
                                  "k_solve_from_chunks_us(by difference)": (t_nograd - t_chunks) * 1e6,
                                  "k_grad_tiles_us": t_grad * 1e6}}
        # dominant kernel = the longer of the two launches of a step
        t_sg = t_step - t_chunks
        if plan.matfree:
            # matrix-free plan (csrc/matfree.hip): the step is k_phi_matfree + k_mf_reduce + k_solve_grad_mf and reads no stored
            # inverse; the stored kernels timed above exist in the plan (batches of vectors use them) but are not what a step runs
            t_phi = chain_time(torch, lambda: plan.phi(m, out=rec))
            t_sg = t_step - t_phi
            kern.update({"evaluation": "matrix-free", "phi_matfree_to_record_us(k_phi_matfree + k_mf_reduce)": t_phi * 1e6,
                         "k_solve_grad_mf_us(step - Phi pass)": t_sg * 1e6, "bytes_moved_per_step(layout)": plan.matfree_bytes})
            t_chunks = t_phi
        if plan.matfree and t_sg < t_chunks:
            kname, tk, abytes, lbytes = "k_phi_matfree", max(t_chunks, 1e-9), ab["phi"] * n_out, plan.matfree_bytes // 2
        elif plan.matfree:
            kname, tk, abytes, lbytes = "k_solve_grad_mf", max(t_sg, 1e-9), ab["grad"] * n_out, plan.matfree_bytes // 2
        elif t_sg >= t_chunks:
            kname, tk, abytes, lbytes = "k_solve_grad", max(t_sg, 1e-9), ab["grad"] * n_out, plan.grad_bytes
        else:
            kname = "k_phi_chunks_shared" if n_out >= 2 else "k_phi_chunks"
            tk, abytes, lbytes = max(t_chunks, 1e-9), ab["phi"] * n_out, plan.phi_bytes
        rp = rocprof_avg_us(kname, (n, kmax, n_out_all))
        roofline = {"bound": "hbm", "kernel": kname, "achieved": abytes / tk / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": abytes / tk / HBM_PEAK, "headline_clock": "frac = frac_chain (measured live in this run)",
                    "frac_chain": abytes / tk / HBM_PEAK,
                    "frac_rocprof": None if rp is None else abytes / (rp * 1e-6) / HBM_PEAK,
                    "traffic": pmc_traffic(kname, (n, kmax, n_out_all)),
                    "traffic_source": "profiles/*_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950-corrected)",
                    "algorithmic_bytes_per_launch": abytes, "layout_bytes_per_launch": lbytes, "avg_launch_us": tk * 1e6,
                    "clocks": "chain: HIP events around a graph of dependent launches, one launch = kernel + dispatch gap to its "
                              "successor (this run); rocprof: begin-to-end of the dispatch alone from the committed "
                              "rocprofv3 --kernel-trace summary under profiles/",
                    "rocprof_avg_us": rp,
                    "note": None if not plan.matfree else "matrix-free evaluation: `achieved` is the prescribed figure -- the reference layout's "
                            "ALGORITHMIC bytes over the launch time -- but those bytes are not moved (the group inverses are recomputed in "
                            "registers), so it can exceed the HBM peak; layout_bytes_per_launch / traffic are what the kernel reads and writes",
                    "step": {"algorithmic_bytes": ab["eval"] * n_out, "achieved_GBps": ab["eval"] * n_out / sec_per_step / 1e9,
                             "frac": ab["eval"] * n_out / sec_per_step / HBM_PEAK}}
        if not args.no_batched:
            # batched throughput: nc DIFFERENT allocation vectors per launch (line-search trial points, integer candidates);
            # the inverse-covariance streams are read once per launch whatever nc is
            batched = {}
            for nc in BATCHES:
                rngb = np.random.RandomState(77 + nc)
                rings = [torch.from_numpy(10.0 * rngb.rand(nc, L)).to(dev) for _ in range(2)]
                bv = torch.empty((nc, n_out), dtype=torch.float64, device=dev)
                bg = torch.empty((nc, plan.grad_len), dtype=torch.float64, device=dev)
                bs = torch.empty((nc, n_out), dtype=torch.int32, device=dev)
                bstep = lambda i, rings=rings, bv=bv, bg=bg, bs=bs: plan.eval(rings[i % 2], out=(bv, bg, bs))   # noqa: E731
                bst = Stepper(torch, bstep, 2, max(40, args.steps // nc), args.graph_steps, not args.no_graph, barrier)
                sec, rpt = bst.timed(max(4, args.warmup // nc))
                assert bool((bs == 0).all())
                # bytes really moved per sequence: the two layouts once per batch + per vector its allocation (gathered), the chunk
                # partials (written, then folded by every solve workgroup's output) and its gradient
                moved = plan.phi_bytes + plan.grad_bytes + nc * (L * 8 + plan.grad_len * 8 + 2 * (plan.phi_bytes // (256 * 12)) * 16)
                batched["n_cand=%d" % nc] = {"value": nc * n_out / sec, "ms_per_launch_sequence": sec * 1e3,
                                             "steps": bst.steps, "repeats": rpt, "launch": bst.launch_label(),
                                             "moved_bytes_per_sequence": int(moved), "moved_GBps": moved / sec / 1e9,
                                             "frac_of_hbm_peak_moved": moved / sec / HBM_PEAK,
                                             "algorithmic_GBps": nc * ab["eval"] * n_out / sec / 1e9}
            batched["note"] = ("same workload, nc allocation vectors evaluated per launch sequence; the roofline figure is in MOVED "
                               "bytes (the inverse-covariance streams are read once per batch); algorithmic_GBps counts every "
                               "evaluation in the reference layout and is a throughput label, not a bandwidth")

    if rank == 0:
        if world == 1:
            par = "single GPU"
        elif weak:
            par = "candidate axis over %d GPUs: every GPU evaluates its own allocation vector for all %d outputs per step, no data-path collective" % (world, n_out_all)
        elif sharded is None:
            par = "outputs sharded over %d GPUs (%d per GPU), no data-path collective" % (world, len(my_outputs))
        else:
            par = "group set sharded over %d GPUs (contiguous shards balanced by sum k^2), one all-reduce(SUM) of the Phi record (%d f64) per step" % (world, n_out * plan.reclen)
        out = {
            "metric": "Phi-assemblies/s", "value": n_out_all * (world if weak else 1) / sec_per_step,
            "unit": "assemblies/s (one output: Phi(m) over all groups -> V, grad V)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "repeats": repeats, "ms_per_step": sec_per_step * 1e3,
            "timed_region_s": sec_per_step * repeats * args.steps,
            "higher_is_better": True, "scaling": "weak" if (weak or world == 1) else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "collective_backend": None if (world == 1 or weak or sharded is None) else ("rccl" if backend == "nccl" else backend),
            "ranks_seen": ranks_seen,
            "exchange": None if sharded is None else sharded.exchange_name,
            "config": {"workload": "n=%d models, groups up to size %d (K_tot=%d), n_out=%d, Wishart covariances; one step = V and grad V of one allocation for all outputs"
                                   % (n, kmax, L, n_out_all),
                       "n_models": n, "k_max": kmax, "K_tot": L, "n_out": n_out_all, "batch": 1, "parallelism": par,
                       "evaluation": "matrix-free (group inverses recomputed in registers, csrc/matfree.hip)" if plan.matfree else "stored group inverses",
                       "launch": stepper.launch_label()},
            "kernels_us": kern,
        }
        out.update(extra)
        if roofline is not None:
            out["roofline"] = roofline
        if batched is not None:
            out["batched"] = batched
        if world == 1 and not args.no_sap:
            out["sap_wallclock"] = sap_wallclock(prob)
            if not args.no_cpu_baseline:
                try:
                    out["sap_wallclock"]["cpu_baseline"] = sap_cpu_baseline(prob, out["sap_wallclock"]["warm"]["max_variance"])
                    cb_s = out["sap_wallclock"]["cpu_baseline"]
                    out["sap_wallclock"]["setup_speedup_vs_cpu_baseline"] = cb_s["setup_s"] / out["sap_wallclock"]["warm"]["setup_s"]
                except Exception as err:      # the headline line must not depend on this leg
                    out["sap_wallclock"]["cpu_baseline"] = {"error": repr(err)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(prob)
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
