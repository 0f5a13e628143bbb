"""
blue_fn -- sample a group of coupled models and accumulate sums / sums of products.  Same call signature and return
shapes as bluest/blue_fn.py:36-227; the cost there is the user's model, so this is plain host Python (SURVEY.md
section 2 row 13: outside the accelerated path).  Differences: samples are not split over MPI ranks unless an
mpi4py-like `comm` with Get_rank/Get_size/allreduce is passed, and the optional per-sample file dump is not provided.
"""
from inspect import signature
from time import time

from numpy import array, isfinite, ndarray, zeros
from numpy.random import RandomState


class SerialComm(object):
    """stand-in for mpi4py.MPI.COMM_WORLD on one process"""

    def Get_size(self): return 1
    def Get_rank(self): return 0
    def bcast(self, x, root=0): return x
    def barrier(self): pass
    def allreduce(self, x, op=None): return x


def is_output_finite(Ps):
    """bluest/blue_fn.py:15-29"""
    for n in range(len(Ps)):
        for i in range(len(Ps[0])):
            check = isfinite(Ps[n][i])
            if isinstance(check, ndarray):
                check = check.all()
            else:
                try: check = all(check)
                except TypeError: pass
            if not check:
                return False, i, n
    return True, None, None


def blue_fn(ls, N, problem, sampler=None, inners=None, comm=None, N1=1, No=1, verbose=True, compute_mlmc_differences=False,
            filename=None, outputs_to_save=None):
    """returns (sumse, sumsc, cost[, sumsd1, sumsd2]) with sumse[n][i] = sum P_n,i and sumsc[n][i,j] = sum <P_n,i, P_n,j>"""
    if filename is not None:
        raise NotImplementedError("per-sample file dumps (samplefile=...) are not part of this build")
    L = len(ls)
    if comm is None: comm = SerialComm()
    mpiRank, mpiSize = comm.Get_rank(), comm.Get_size()
    if inners is None: inners = [lambda a, b: a * b for n in range(No)]
    if sampler is None:
        RNG = RandomState(1 + mpiRank)

        def sampler(ls, N=1):
            sample = RNG.randn(N)
            return [sample for i in range(L)]

    sumse = [[0 for i in range(L)] for n in range(No)]
    sumsc = [zeros((L, L)) for n in range(No)]
    sumsd1 = [[[0 for j in range(L)] for i in range(L)] for n in range(No)]
    sumsd2 = [[[0 for j in range(L)] for i in range(L)] for n in range(No)]
    NN = [N // mpiSize + (1 if i < N % mpiSize else 0) for i in range(mpiSize)]
    nobatch = len(signature(sampler).parameters) == 1
    if nobatch: N1 = 1
    cpu_cost = 0.0
    for it in range(1, NN[mpiRank] + 1, N1):
        N2 = min(N1, NN[mpiRank] - it + 1)
        ok = False
        while not ok:                                   # resample on inf/NaN (blue_fn.py:118-129)
            samples = sampler(ls) if nobatch else sampler(ls, N2)
            start = time()
            Ps = problem.evaluate(ls, samples)
            end = time()
            ok, model_n, output_n = is_output_finite(Ps)
            if not ok:
                print("Warning! Problem evaluation returned inf or NaN value for model %d and output %d. Resampling..." % (model_n, output_n), flush=True)
        cpu_cost += end - start
        for n in range(No):
            reps = [None] if N1 == 1 else range(N2)
            for r in reps:
                P = Ps[n] if r is None else [Ps[n][i][r] for i in range(L)]
                for i in range(L):
                    sumse[n][i] += P[i]
                sumsc[n] += array([[inners[n](P[i], P[j]) for i in range(L)] for j in range(L)])
                if compute_mlmc_differences:
                    for i in range(L):
                        for j in range(i + 1, L):
                            sumsd1[n][i][j] += P[i] - P[j]
                            sumsd2[n][i][j] += inners[n](P[i] - P[j], P[i] - P[j])
    cost = N * problem.cost if hasattr(problem, 'cost') else comm.allreduce(cpu_cost)
    for n in range(No):
        sumsc[n] = comm.allreduce(sumsc[n])
        for i in range(L):
            sumse[n][i] = comm.allreduce(sumse[n][i])
            if compute_mlmc_differences:
                for j in range(i + 1, L):
                    sumsd1[n][i][j] = comm.allreduce(sumsd1[n][i][j])
                    sumsd2[n][i][j] = comm.allreduce(sumsd2[n][i][j])
    if compute_mlmc_differences:
        return (sumse, sumsc, cost, sumsd1, sumsd2)
    return (sumse, sumsc, cost)
