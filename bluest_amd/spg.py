"""
Nonmonotone spectral projected gradient driver: same algorithm, argument names, return dict and stopping rules
as bluest/spg.py:3-132, written so that x, g, d may be numpy arrays OR torch tensors living in HBM (all vector
work then stays on the GPU; only the handful of scalars the control flow needs cross PCIe).

In the reference `spg` serves the covariance projection only (bluest/blue_models.py:396); here it is also the
sample-allocation solver behind `solver="spg"`, with `proj` = simplex projection.
"""
import numpy as np


def _f(x):
    """python float of a 0-dim numpy/torch value"""
    return float(x)


def _absmax(v):
    return _f(abs(v).max())


def _dot(a, b):
    return _f(a @ b)


def linesearch(feval, x, f, g, d, Hlength, last_fval, max_fevals, count, gdotd=None):
    """bluest/spg.py:3-37 (safeguarded quadratic interpolation, sigma in [0.1,0.9], gamma=1e-4, history max).
    `gdotd` may be supplied by a projection kernel that already reduced g.d."""
    sigma_min = 0.1
    sigma_max = 0.9
    gamma = 10 ** -4

    fmax = max(last_fval)
    if gdotd is None:
        gdotd = _dot(g, d)

    alpha = 1.0
    xnew = x + alpha * d
    fnew = feval(xnew)
    count += 1

    while fnew > fmax + gamma * alpha * gdotd and count < max_fevals:
        if alpha <= sigma_min:
            alpha *= 0.5
        else:
            alpha_t = -0.5 * (alpha ** 2) * gdotd / (fnew - f - alpha * gdotd)
            if alpha_t < sigma_min or alpha_t > sigma_max * alpha:
                alpha_t = 0.5 * alpha
            alpha = alpha_t
        xnew = x + alpha * d
        fnew = feval(xnew)
        count += 1

    linesearch_info = 0 if fnew <= fmax + gamma * alpha * gdotd else 2
    return count, fnew, xnew, linesearch_info


def spg(feval, geval, proj, x, eps=1.0e-4, maxit=200, max_fevals=10 ** 5, verbose=True, lmbda_min=10. ** -30,
        lmbda_max=10. ** 30, Hlength=10, proj_step=None, callback=None, metric_dot=None):
    """bluest/spg.py:39-132.

    feval(x) -> float; geval(x) -> vector; proj(x) -> vector.
    Optional `proj_step(x, g, lmbda) -> (d, gdotd, dmax)` fuses d = proj(x - lmbda*g) - x with the reductions
    g.d and max|d| (one kernel on the GPU); without it the three are formed as in the reference.
    Optional `metric_dot(s, x) -> float` replaces s.s in the Barzilai-Borwein step by s^T D(x)^-1 s when `proj_step`
    works in a variable diagonal metric D(x) (scaled SPG); x is the point the step s started from.
    """
    n = len(x)
    if verbose:
        print("\nSPECTRAL PROJECTED GRADIENT METHOD.\n")
        print("Problem size:\t%d\n" % n)
        print(" ITER\t      F\t\t   GPINFNORM\n")

    it = 0
    count = 0
    last_fval = -np.inf * np.ones((Hlength,))

    x = proj(x)
    f = feval(x)
    g = geval(x)
    count += 1
    last_fval[0] = f

    def projected_step(x, g, lmbda):
        if proj_step is not None:
            return proj_step(x, g, lmbda)
        d = proj(x - lmbda * g) - x
        return d, None, None

    gp, _, gpmax = projected_step(x, g, 1.0)
    if gpmax is None:
        gpmax = _absmax(gp)
    if gpmax > 1.0e-15:
        lmbda = min(lmbda_max, max(lmbda_min, 1.0 / gpmax))
    else:
        lmbda = 0.0

    while gpmax > eps and it < maxit and count < max_fevals:
        if verbose:
            print(" %d\t %e\t %e" % (it, f, gpmax))
        it += 1

        d, gdotd, _ = projected_step(x, g, lmbda)
        count, fnew, xnew, linesearch_info = linesearch(feval, x, f, g, d, Hlength, last_fval, max_fevals, count, gdotd=gdotd)

        if linesearch_info == 2:
            if verbose:
                print("WARNING! SPG: Maximum of functional evaluations reached.\n")
            return {"x": x, "f": f, "gpmax": gpmax, "it": it, "count": count, "solver_info": 2}

        f = fnew
        last_fval[it % Hlength] = f
        gnew = geval(xnew)

        s = xnew - x
        y = gnew - g
        sdots = _dot(s, s) if metric_dot is None else metric_dot(s, x)
        sdoty = _dot(s, y)

        x = xnew
        g = gnew

        gp, _, gpmax = projected_step(x, g, 1.0)
        if gpmax is None:
            gpmax = _absmax(gp)

        if sdoty <= 0:
            lmbda = lmbda_max
        else:
            lmbda = min(lmbda_max, max(lmbda_min, sdots / sdoty))
        if callback is not None:
            callback(it, f, gpmax, lmbda)

    if verbose:
        print(" %d\t %e\t %e" % (it, f, gpmax))
        print("\n")
        print("Number of iterations               : %d\n" % it)
        print("Number of functional evaluations   : %d\n" % count)
        print("Objective function value           : %e\n" % f)
        print("Sup-norm of the projected gradient : %e\n" % gpmax)

    if gpmax <= eps:
        solver_info = 0
        if verbose:
            print("SPG: Optimal solution found.\n")
    elif it >= maxit:
        solver_info = 1
        if verbose:
            print("WARNING! SPG: Maximum number of iterations reached.\n")
    else:
        solver_info = 2
        if verbose:
            print("WARNING! SPG: Maximum number of functional evaluations reached.\n")
    return {"x": x, "f": f, "gpmax": gpmax, "it": it, "count": count, "solver_info": solver_info}
