"""
Nonmonotone spectral projected gradient method, host-driven: the algorithm, the argument names, the result dict and the
stopping rules of bluest/spg.py:3-132 (same arithmetic in the same order, so a run reproduces the reference's iterates --
tests/test_gpu_parity.py::test_spg_trajectory), organised around a small state object so that x, g, d may be numpy arrays
OR torch tensors living in HBM: all vector work then stays on the GPU and only the handful of scalars the control flow needs
cross PCIe.

In the reference `spg` serves the covariance projection only (bluest/blue_models.py:396); here it is also the host-driven
variant of the sample-allocation solver behind `solver="spg"`, with `proj` = simplex projection.
The default solver path is the device-resident loop in spg_device.py; this file is the layer it is tested against.
"""
import numpy as np

from .host import in_host_section

# constants of the line search (bluest/spg.py:5-7)
STEP_SHRINK_BELOW = 0.1      # below this step the search simply halves
STEP_KEEP_FRACTION = 0.9     # an interpolated step above this fraction of the old one is replaced by half of it
ARMIJO = 10 ** -4

REASONS = {0: "SPG: Optimal solution found.\n",
           1: "WARNING! SPG: Maximum number of iterations reached.\n",
           2: "WARNING! SPG: Maximum number of functional evaluations reached.\n"}


def _scalar(v):
    """python float of a 0-dim numpy / torch value"""
    return float(v)


def _dot(a, b):
    return _scalar(a @ b)


def linesearch(feval, x, f, g, d, Hlength, last_fval, max_fevals, count, gdotd=None):
    """Nonmonotone Armijo search along d with safeguarded quadratic interpolation (bluest/spg.py:3-37): accept as soon as the
    trial value is below max(history) + 1e-4 * alpha * g.d.  `gdotd` may come from a projection kernel that already reduced g.d.
    Returns (evaluation count, value, point, 0 = accepted | 2 = out of evaluations)."""
    slope = _dot(g, d) if gdotd is None else gdotd
    ceiling = max(last_fval)

    def too_high(value, step):
        return value > ceiling + ARMIJO * step * slope

    alpha = 1.0
    while True:
        trial = x + alpha * d
        value = feval(trial)
        count += 1
        if not too_high(value, alpha) or count >= max_fevals:
            break
        if alpha <= STEP_SHRINK_BELOW:
            alpha *= 0.5
            continue
        guess = -0.5 * (alpha ** 2) * slope / (value - f - alpha * slope)      # minimiser of the interpolating parabola
        alpha = 0.5 * alpha if (guess < STEP_SHRINK_BELOW or guess > STEP_KEEP_FRACTION * alpha) else guess
    return count, value, trial, (0 if value <= ceiling + ARMIJO * alpha * slope else 2)     # (a NaN value ends as 2)


class _Run(object):
    """iterate, gradient and bookkeeping of one spg() call"""

    def __init__(self, feval, geval, proj, proj_step, metric_dot, x, Hlength, limits):
        self.feval, self.geval, self.proj, self.proj_step, self.metric_dot = feval, geval, proj, proj_step, metric_dot
        self.lmbda_min, self.lmbda_max = limits
        self.history = -np.inf * np.ones((Hlength,))
        self.x = proj(x)
        self.f = feval(self.x)
        self.g = geval(self.x)
        self.count, self.it = 1, 0
        self.history[0] = self.f
        self.gpmax = self.stationarity()
        self.lmbda = self.clamp(1.0 / self.gpmax) if self.gpmax > 1.0e-15 else 0.0

    def clamp(self, step):
        return min(self.lmbda_max, max(self.lmbda_min, step))

    def direction(self, lmbda):
        """d = P(x - lmbda g) - x, with g.d and max|d| when a fused projection kernel supplies them"""
        if self.proj_step is not None:
            return self.proj_step(self.x, self.g, lmbda)
        return self.proj(self.x - lmbda * self.g) - self.x, None, None

    def stationarity(self):
        """sup-norm of the projected gradient P(x - g) - x: the stopping measure"""
        gp, _, sup = self.direction(1.0)
        return _scalar(abs(gp).max()) if sup is None else sup

    def advance(self, xnew, fnew):
        """accept xnew: history, Barzilai-Borwein step from s = xnew - x, y = g(xnew) - g(x)"""
        self.f = fnew
        self.history[self.it % len(self.history)] = fnew
        gnew = self.geval(xnew)
        s, y = xnew - self.x, gnew - self.g
        sdots = _dot(s, s) if self.metric_dot is None else self.metric_dot(s, self.x)
        sdoty = _dot(s, y)
        self.x, self.g = xnew, gnew
        self.gpmax = self.stationarity()
        self.lmbda = self.lmbda_max if sdoty <= 0 else self.clamp(sdots / sdoty)

    def result(self, info):
        return {"x": self.x, "f": self.f, "gpmax": self.gpmax, "it": self.it, "count": self.count, "solver_info": info}


@in_host_section
def spg(feval, geval, proj, x, eps=1.0e-4, maxit=200, max_fevals=10 ** 5, verbose=True, lmbda_min=10. ** -30,
        lmbda_max=10. ** 30, Hlength=10, proj_step=None, callback=None, metric_dot=None):
    """bluest/spg.py:39-132: minimise feval over the set proj projects onto.

    feval(x) -> float; geval(x) -> vector; proj(x) -> vector.
    Optional `proj_step(x, g, lmbda) -> (d, gdotd, dmax)` fuses d = proj(x - lmbda*g) - x with the reductions g.d and max|d|
    (one kernel on the GPU); without it the three are formed as in the reference.
    Optional `metric_dot(s, x) -> float` replaces s.s in the Barzilai-Borwein step by s^T D(x)^-1 s when `proj_step` works in a
    variable diagonal metric D(x) (scaled SPG); x is the point the step s started from.
    Returns {"x", "f", "gpmax", "it", "count", "solver_info"}; solver_info 0 = converged (gpmax <= eps), 1 = maxit, 2 = max_fevals.
    """
    def say(text):
        if verbose:
            print(text)

    say("\nSPECTRAL PROJECTED GRADIENT METHOD.\n")
    say("Problem size:\t%d\n" % len(x))
    say(" ITER\t      F\t\t   GPINFNORM\n")
    run = _Run(feval, geval, proj, proj_step, metric_dot, x, Hlength, (lmbda_min, lmbda_max))
    while run.gpmax > eps and run.it < maxit and run.count < max_fevals:
        say(" %d\t %e\t %e" % (run.it, run.f, run.gpmax))
        run.it += 1
        d, gdotd, _ = run.direction(run.lmbda)
        run.count, fnew, xnew, ls_info = linesearch(feval, run.x, run.f, run.g, d, Hlength, run.history, max_fevals, run.count,
                                                    gdotd=gdotd)
        if ls_info == 2:
            say(REASONS[2])
            return run.result(2)
        run.advance(xnew, fnew)
        if callback is not None:
            callback(run.it, run.f, run.gpmax, run.lmbda)
    say(" %d\t %e\t %e" % (run.it, run.f, run.gpmax))
    say("\n")
    say("Number of iterations               : %d\n" % run.it)
    say("Number of functional evaluations   : %d\n" % run.count)
    say("Objective function value           : %e\n" % run.f)
    say("Sup-norm of the projected gradient : %e\n" % run.gpmax)
    info = 0 if run.gpmax <= eps else (1 if run.it >= maxit else 2)
    say(REASONS[info])
    return run.result(info)
