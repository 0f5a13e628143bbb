"""
BLUEProblem -- the user-facing entry of the sample-allocation path: the signatures and return values
of `setup_solver()` / `solve()` of bluest/blue_models.py:448-576, on top of bluest_amd.mosap.MOSAP.

In scope here: given covariances and model costs -> model groups (cliques of the coupling graph up to size K, or the user's
groups), their union over the outputs, group costs, one MOSAP on the GPU, `solver="spg"`, the reference's return dictionaries;
`solve()` then samples the selected groups through the user's `sampler` / `evaluate` and forms the BLUE estimators.
Out of scope (refused with BLUESTError): estimating covariances or costs by sampling, the SPD
projection of incomplete covariances, saving / loading model graphs, MLMC / MFMC / plain-MC drivers.  An MPI communicator passed
as `comm` is honoured the way the reference uses it (optimiser and estimators on rank 0 + bcast, samples split over the ranks).

Conventions kept from the reference (bluest/blue_models.py:43-56, :166-179): in a user covariance an infinite entry means "never
couple these two models", a zero entry means "uncorrelated" (such pairs are not coupled either when `remove_uncorrelated`, the
default); `get_covariance()` returns NaN where two models are not coupled.  Models that cannot be reached from model 0 through
couplings are left out of every group (:312-322).
"""
from itertools import combinations

import numpy as np

from .mosap import MOSAP
from .host import in_host_section
from .sap import BLUESTError

default_params = {"verbose": True, "comm": None, "remove_uncorrelated": True, "optimization_solver": "spg", "sample_batch_size": 1}


class _SerialComm(object):
    """what `get_comm()` returns when no MPI communicator was passed (the reference defaults to mpi4py's COMM_WORLD,
    bluest/blue_models.py:22; mpi4py is not a dependency of this build): one rank, collectives are identities"""

    def Get_rank(self): return 0
    def Get_size(self): return 1
    def bcast(self, obj, root=0): return obj
    def allreduce(self, obj, op=None): return obj
    def barrier(self): return None


class _Coupling(object):
    """coupling structure of ONE output: the covariance and which pairs of models may share a group"""

    def __init__(self, C, remove_uncorrelated):
        C = np.array(C, dtype=np.float64)
        M = C.shape[0]
        if C.shape != (M, M):
            raise ValueError("covariance must be square")
        if np.isnan(C).any():
            raise BLUESTError("unknown (NaN) covariance entries would have to be estimated by sampling and projected to SPD "
                              "(bluest/blue_models.py:326-433): outside this GPU build -- pass complete covariances")
        never = np.isinf(C)
        self.cov = np.where(never, 0.0, C)
        self.linked = ~never
        if remove_uncorrelated:
            self.linked &= (self.cov != 0.0)
        np.fill_diagonal(self.linked, True)
        self.linked &= self.linked.T
        # models reachable from model 0
        seen, frontier = {0}, [0]
        while frontier:
            nxt = []
            for i in frontier:
                for j in np.flatnonzero(self.linked[i]).tolist():
                    if j not in seen:
                        seen.add(j)
                        nxt.append(j)
            frontier = nxt
        self.component = sorted(seen)

    def covariance(self):
        out = self.cov.copy()
        out[~self.linked] = np.nan
        return out

    def is_clique(self, group):
        g = np.asarray(group, dtype=np.int64)
        return bool(self.linked[np.ix_(g, g)].all()) and all(int(i) in set(self.component) for i in g)

    def cliques(self, K):
        """all cliques of up to K models inside model 0's component, per size, each sorted, lexicographic order"""
        nodes = self.component
        sub = self.linked[np.ix_(nodes, nodes)]
        if sub.all():
            return [np.array(list(combinations(nodes, k)), dtype=np.int64).reshape(-1, k) for k in range(1, K + 1)]
        out = [[(i,) for i in nodes]]
        for k in range(2, K + 1):                                # extend every (k-1)-clique by a larger, fully linked model
            bigger = []
            for c in out[-1]:
                cand = self.linked[list(c)].all(axis=0)
                bigger += [c + (j,) for j in nodes if j > c[-1] and cand[j]]
            out.append(bigger)
        return [np.array(level, dtype=np.int64).reshape(-1, k + 1) for k, level in enumerate(out)]


class BLUEProblem(object):
    def __init__(self, M, C=None, costs=None, mlmc_variances=None, datafile=None, n_outputs=1, **params):
        self.M, self.n_outputs = int(M), int(n_outputs)
        self.params = dict(default_params, **params)
        self.default_params = default_params
        comm = self.params["comm"]
        if comm is None:
            comm = _SerialComm()
        self.mpiRank = comm.Get_rank()
        self.mpiSize = comm.Get_size()
        self.comm = comm
        self.warning = self.mpiRank == 0
        self.verbose = bool(self.params["verbose"]) and self.warning
        self.MOSAP = None
        self.MOSAP_output = None
        if datafile is not None:
            raise BLUESTError("model-graph files (bluest/blue_models.py:265-299) are outside this GPU build: pass C and costs")
        if C is None or costs is None:
            raise BLUESTError("covariances and costs must be given: estimating them by sampling (bluest/blue_models.py:326-346, "
                              ":435-441) is outside this GPU build")
        Cs = list(C) if isinstance(C, (list, tuple)) else [C]
        if len(Cs) != self.n_outputs or any(np.shape(c) != (self.M, self.M) for c in Cs):
            raise ValueError("need one %d x %d covariance per output" % (self.M, self.M))
        self._costs = np.array(costs, dtype=np.float64)
        if self._costs.shape != (self.M,):
            raise ValueError("costs must have one entry per model")
        self._coupling = [_Coupling(c, self.params["remove_uncorrelated"]) for c in Cs]
        self.SG = [cp.component for cp in self._coupling]
        for n, cp in enumerate(self._coupling):
            if len(cp.component) < self.M and self.warning:
                print("WARNING! Model graph %d is not connected. Connected graph size: %d" % (n, len(cp.component)))
        self.check_costs(warning=True)
        if self.verbose: print("\nBLUE estimator ready.\n")

    # ---- supplied by the user (bluest/blue_models.py:105-119) -----------------------------------------------------------------
    def evaluate(self, ls, samples, N=1):
        raise NotImplementedError("subclass BLUEProblem and implement evaluate(ls, samples)")

    def sampler(self, ls, N=1):
        raise NotImplementedError("subclass BLUEProblem and implement sampler(ls)")

    def get_models_inner_products(self):
        return [lambda a, b: a * b for n in range(self.n_outputs)]

    def get_comm(self):
        return self.comm

    # ---- data access ----------------------------------------------------------------------------------------------------------
    def get_costs(self):
        return self._costs.copy()

    def get_group_costs(self, groups):
        """cost of a group = sum of the costs of its models (bluest/blue_models.py:137-140); one entry per group, size-major"""
        parts = [self._costs[np.asarray(gk, dtype=np.int64).reshape(len(gk), -1)].sum(axis=1) for gk in groups if len(gk)]
        return np.concatenate(parts) if parts else np.zeros(0)

    def check_costs(self, warning=True):
        dearer = np.flatnonzero(self._costs > self._costs[0]).tolist()
        if dearer:
            if not warning:
                raise ValueError("Model zero is not the most expensive model. Consider removing the more expensive models %s" % dearer)
            if self.warning:
                print("WARNING! Model zero is not the most expensive model. The more expensive models are: %s" % dearer)
        return dearer

    def get_covariance(self, n=0):
        return self._coupling[n].covariance()

    def get_covariances(self):
        return [cp.covariance() for cp in self._coupling]

    def get_correlation(self, n=0):
        C = self.get_covariance(n)
        sd = np.sqrt(np.diag(C))
        return C / np.outer(sd, sd)

    def get_correlations(self):
        return [self.get_correlation(n) for n in range(self.n_outputs)]

    # ---- groups ---------------------------------------------------------------------------------------------------------------
    def _groups_per_output(self, K, multi_groups):
        """per output: list over sizes of (L_k, k) int arrays, empty trailing sizes dropped; and the largest size per output"""
        per_output = []
        for n, cp in enumerate(self._coupling):
            if multi_groups is None:
                levels = cp.cliques(min(int(K), self.M))
            else:
                kept = sorted({tuple(sorted(int(i) for i in g)) for g in multi_groups[n] if len(g)})
                kept = [g for g in kept if cp.is_clique(g)]
                if not kept:
                    raise ValueError("no admissible model group for output %d" % n)
                kmax = max(len(g) for g in kept)
                levels = [np.array([g for g in kept if len(g) == k], dtype=np.int64).reshape(-1, k) for k in range(1, kmax + 1)]
            while levels and len(levels[-1]) == 0:
                levels.pop()
            per_output.append(levels)
        return per_output, [len(levels) for levels in per_output]

    @staticmethod
    def _union(per_output, K):
        """the global group list: union over the outputs per size, sorted (bluest/blue_models.py:491-501, hashed)"""
        first = per_output[0]
        if all(len(lv) == len(first) and all(np.array_equal(a, b) for a, b in zip(lv, first)) for lv in per_output[1:]):
            return [lv.copy() for lv in first] + [np.zeros((0, k), dtype=np.int64) for k in range(len(first) + 1, K + 1)]
        out = []
        for k in range(1, K + 1):
            rows = {tuple(r) for lv in per_output if len(lv) >= k for r in lv[k - 1].tolist()}
            out.append(np.array(sorted(rows), dtype=np.int64).reshape(-1, k))
        return out

    # ---- the path -------------------------------------------------------------------------------------------------------------
    @in_host_section
    def setup_solver(self, K=4, budget=None, eps=None, groups=None, multi_groups=None, solver=None, continuous_relaxation=False,
                     max_model_samples=None, optimization_solver_params=None):
        """bluest/blue_models.py:448-538: returns {"models", "samples", "errors", "total_cost"}"""
        if budget is None and eps is None:
            raise ValueError("Need to specify either budget or RMSE tolerance")
        if budget is not None:
            eps = None                                                  # the budget wins (bluest/blue_models.py:450)
        elif np.isscalar(eps):
            eps = [eps] * self.n_outputs
        solver = self.params["optimization_solver"] if solver is None else solver
        if multi_groups is not None and len(multi_groups) != self.n_outputs:
            raise ValueError("multi_groups must be a list of groupings of the same length as the number of outputs.")
        if multi_groups is None and groups is not None:
            multi_groups = [groups] * self.n_outputs
        per_output, Ks = self._groups_per_output(K, multi_groups)
        K = max(Ks)
        union = self._union(per_output, K)
        costs = self.get_group_costs(union)
        multi_costs = [self.get_group_costs(levels) for levels in per_output]
        C = self.get_covariances()

        if self.verbose: print("Computing optimal sample allocation...")
        result = None
        if self.mpiRank == 0:                                           # the optimiser runs on one rank (:508)
            self.MOSAP = MOSAP(C, K, Ks, union, per_output, costs, multi_costs, verbose=self.verbose)
            self.MOSAP.solve(eps=eps, budget=budget, solver=solver, continuous_relaxation=continuous_relaxation,
                             max_model_samples=max_model_samples, solver_params=optimization_solver_params)
            if self.MOSAP.samples is not None:
                Vs = self.MOSAP.variances(self.MOSAP.samples)
                result = {"budget": budget, "eps": eps, "samples": self.MOSAP.samples, "flattened_groups": self.MOSAP.flattened_groups,
                          "variances": Vs, "cost": self.MOSAP.tot_cost}
                if self.verbose:
                    cost_MC = max(C[n][0, 0] / Vs[n] for n in range(self.n_outputs)) * costs[0]
                    print("\nBLUE cost: ", result["cost"], "MC cost: ", cost_MC, "Savings: ", cost_MC / result["cost"])
        result = self.comm.bcast(result, root=0)                        # :526
        self.MOSAP_output = result
        if result is None:
            raise BLUESTError("MOSAP solution failed!")

        chosen = np.flatnonzero(result["samples"] > 0)
        data = {"models": [result["flattened_groups"][i] for i in chosen], "samples": result["samples"][chosen].copy(),
                "errors": np.sqrt(result["variances"]), "total_cost": result["cost"]}
        if self.verbose:
            print("\nModel groups selected: %s\n" % data["models"])
            print("BLUE estimator setup. Max error: ", np.sqrt(max(result["variances"])), " Cost: ", result["cost"], "\n")
        return data

    def _group_sums(self, ls, N):
        """sum over N joint samples of the models `ls`, per output: [n_outputs][len(ls)] (what bluest/blue_fn.py returns first).
        Host Python around the user's model -- sampling is not part of the accelerated path.  With an MPI communicator of
        several ranks the N samples are split as the reference splits them (blue_fn.py:107-111: N // size each, the first
        N % size ranks one more) and the sums are all-reduced (:178-182), so every rank returns the sums of all N."""
        comm = self.get_comm()
        size, rank = comm.Get_size(), comm.Get_rank()
        mine = int(N) // size + (1 if rank < int(N) % size else 0)
        sums = [[0 for _ in ls] for _ in range(self.n_outputs)]
        for _ in range(mine):
            while True:
                values = self.evaluate(ls, self.sampler(ls))
                if all(np.all(np.isfinite(v)) for out in values for v in out):
                    break                                               # non-finite model output: draw again (blue_fn.py:118-129)
            for n in range(self.n_outputs):
                for i in range(len(ls)):
                    sums[n][i] = sums[n][i] + values[n][i]
        if size > 1:
            for n in range(self.n_outputs):
                for i in range(len(ls)):
                    sums[n][i] = comm.allreduce(sums[n][i])            # default op of mpi4py's allreduce is SUM
        return sums

    def solve(self, K=4, budget=None, eps=None, groups=None, multi_groups=None, solver=None, verbose=True, continuous_relaxation=False,
              max_model_samples=None, optimization_solver_params=None):
        """bluest/blue_models.py:540-576: returns (estimates, their standard errors, total cost)"""
        have = self.MOSAP_output
        changed = have is not None and ((budget is not None and budget != have["budget"]) or
                                        (eps is not None and np.any(np.asarray(eps) != np.asarray(have["eps"]))))
        if have is None or changed:
            self.setup_solver(K=K, budget=budget, eps=eps, groups=groups, multi_groups=multi_groups, solver=solver,
                              continuous_relaxation=continuous_relaxation, max_model_samples=max_model_samples,
                              optimization_solver_params=optimization_solver_params)
        if self.verbose and verbose: print("\nSampling BLUE...\n")
        out = self.MOSAP_output
        sums = [[] for _ in range(self.n_outputs)]
        for ls, count in zip(out["flattened_groups"], out["samples"]):
            got = self._group_sums(ls, count) if count > 0 else [[0] * len(ls)] * self.n_outputs
            for n in range(self.n_outputs):
                sums[n].append(got[n])
        if self.mpiRank == 0:                                           # only rank 0 owns a MOSAP (:565-571)
            mus, Vs = self.MOSAP.compute_BLUE_estimators(sums, out["samples"])
        else:
            mus, Vs = None, None
        mus = self.comm.bcast(mus, root=0)
        Vs = self.comm.bcast(Vs, root=0)
        return mus, np.sqrt(Vs), out["cost"]

    # ---- refused ----------------------------------------------------------------------------------------------------------------
    def _out_of_scope(self, *a, **k):
        raise BLUESTError("outside this GPU build (SURVEY.md section 2): only setup_solver() / solve() with given covariances")

    setup_mlmc = solve_mlmc = setup_mfmc = solve_mfmc = setup_mc = solve_mc = _out_of_scope
    save_graph_data = load_graph_data = estimate_missing_covariances = project_covariances = estimate_costs = _out_of_scope
    complexity_test = variance_test = _out_of_scope
