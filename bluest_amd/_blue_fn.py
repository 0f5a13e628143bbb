"""
blue_fn -- sample a group of coupled models and return the sums the BLUE estimator is made of (role of bluest/blue_fn.py:36-211,
exported by the reference's package next to SAP / MOSAP / BLUEProblem, bluest/__init__.py:7-10).

Host Python around the user's model: nothing here is on the accelerated path (SURVEY.md section 2 lists sampling as out of scope),
so this is the plain accumulation -- sums of the outputs and of their pairwise inner products, cost = evaluation time unless the
problem states its own -- with the reference's argument names.  The sample files (`filename`, `outputs_to_save`) and the MLMC
difference sums (`compute_mlmc_differences`) belong to the reference's covariance-estimation / MLMC drivers, which this build
refuses (BLUEProblem._out_of_scope), and are refused here too.
"""
import time
from inspect import signature

import numpy as np


class _Solo(object):
    """the communicator of a run without MPI"""

    def Get_rank(self): return 0
    def Get_size(self): return 1
    def allreduce(self, v, op=None): return v


def _all_finite(values):
    return all(bool(np.all(np.isfinite(v))) for out in values for v in out)


def blue_fn(ls, N, problem, sampler=None, inners=None, comm=None, N1=1, No=1, verbose=True, compute_mlmc_differences=False,
            filename=None, outputs_to_save=None):
    """(sumse, sumsc, cost): sumse[n][i] = sum over the N paths of output n of model ls[i], sumsc[n][i, j] = sum of the inner
    products of outputs i and j, cost = N * problem.cost if the problem states one, else the seconds spent in problem.evaluate.
    problem.evaluate(ls, samples) -> Ps[n][i]; sampler(ls) or sampler(ls, count) -> one input per model (default: the same
    standard normal draw for every model); with an MPI communicator the paths are split over its ranks and the sums reduced."""
    if compute_mlmc_differences or filename is not None or outputs_to_save is not None:
        from .sap import BLUESTError
        raise BLUESTError("blue_fn: sample files and MLMC difference sums are outside this GPU build (SURVEY.md section 2)")
    comm = comm if comm is not None else _Solo()
    rank, size = comm.Get_rank(), comm.Get_size()
    n_models = len(ls)
    if inners is None:
        inners = [lambda a, b: a * b] * No
    if sampler is None:
        rng = np.random.RandomState(1 + rank)

        def sampler(ls, count=1):
            draw = rng.randn(count)
            return [draw] * n_models
    batched = len(signature(sampler).parameters) > 1
    chunk = int(N1) if batched else 1
    todo = int(N) // size + (1 if rank < int(N) % size else 0)
    sumse = [[0] * n_models for _ in range(No)]
    sumsc = [np.zeros((n_models, n_models)) for _ in range(No)]
    spent = 0.0
    while todo > 0:
        count = min(chunk, todo)
        while True:                                        # a non-finite model output is drawn again
            inputs = sampler(ls, count) if batched else sampler(ls)
            t0 = time.time()
            Ps = problem.evaluate(ls, inputs)
            spent += time.time() - t0
            if _all_finite(Ps):
                break
            if verbose:
                print("blue_fn: non-finite model output for models %s, drawing again" % (ls,), flush=True)
        paths = (lambda v: [v]) if chunk == 1 else (lambda v: [v[q] for q in range(count)])
        for n in range(No):
            rows = [paths(Ps[n][i]) for i in range(n_models)]
            for i in range(n_models):
                for v in rows[i]:
                    sumse[n][i] = sumse[n][i] + v
            for i in range(n_models):
                for j in range(n_models):
                    sumsc[n][j, i] += np.sum([inners[n](a, b) for a, b in zip(rows[i], rows[j])])
        todo -= count
    cost = N * problem.cost if hasattr(problem, "cost") else comm.allreduce(spent)
    for n in range(No):
        sumsc[n] = comm.allreduce(sumsc[n])
        sumse[n] = [comm.allreduce(v) for v in sumse[n]]
    return sumse, sumsc, cost
