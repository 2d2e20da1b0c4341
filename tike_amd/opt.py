"""Generic optimisation helpers (reference src/tike/opt.py).

Work on torch tensors (device) or NumPy arrays alike.
"""
import logging
import warnings

import numpy as np
import torch

logger = logging.getLogger(__name__)


def is_converged(algorithm_options):
    """True if the cost slope over the window is non-negative (opt.py:21-45)."""
    window = algorithm_options.convergence_window
    if (window >= 2 and len(algorithm_options.costs) >= window
            and len(algorithm_options.costs) % window // 2 == 0):
        m = np.array(algorithm_options.costs[-window:])
        m = np.mean(np.reshape(m, (len(m), -1)), axis=1)
        p = np.polyfit(x=range(window), y=m, deg=1, full=False, cov=False)
        if p[0] >= 0:
            return True
    return False


def momentum(g, v, m, vdecay=None, mdecay=0.9):
    """m = mdecay*m + (1-mdecay)*g (opt.py:67-82)."""
    m = 0 if m is None else m
    m = mdecay * m + (1 - mdecay) * g
    return m, None, m


def adam(g, v=None, m=None, vdecay=0.999, mdecay=0.9, eps=1e-8):
    """ADAM direction (opt.py:165-213)."""
    is_t = isinstance(g, torch.Tensor)
    zeros = torch.zeros_like if is_t else np.zeros_like
    sqrt = torch.sqrt if is_t else np.sqrt
    v = zeros(g.real) if v is None else v
    m = zeros(g) if m is None else m
    m = mdecay * m + (1 - mdecay) * g
    v = vdecay * v + (1 - vdecay) * (g * g.conj()).real
    m_ = m / (1 - mdecay)
    v_ = sqrt(v / (1 - vdecay))
    return m_ / (v_ + eps), v, m


def fit_line_least_squares(y, x):
    """(slope, intercept) of the least-squares line (opt.py:383-400)."""
    x = np.asarray(x, dtype=float)
    y = np.asarray(y, dtype=float)
    assert len(x) == len(y)
    count = len(x)
    sum_x, sum_y = np.sum(x), np.sum(y)
    slope = (count * np.sum(x * y) - sum_x * sum_y) / (count * np.sum(x * x) -
                                                       sum_x * sum_x)
    return slope, (sum_y - slope * sum_x) / count


def line_search(f, x, d, update_multi, step_length=1, step_shrink=0.5,
                cost=None):
    """Backtracking line search (opt.py:216-278)."""
    assert 0 < step_shrink < 1
    fx = f(x) if cost is None else cost
    while True:
        xsd = update_multi(x, step_length, d)
        fxsd = f(xsd)
        if fxsd <= fx:
            break
        step_length *= step_shrink
        if step_length < 1e-32:
            warnings.warn("Line search failed for conjugate gradient.")
            step_length, fxsd, xsd = 0, fx, x
            break
    return step_length, fxsd, xsd


def direction_dy(xp, grad1, grad0=None, dir_=None):
    """Dai-Yuan search direction (opt.py:281-301); lists of one array per
    device as in the reference."""
    if dir_ is None:
        return [-grad1[0]]
    g1 = grad1[0]
    norm2 = (g1 * g1.conj()).real.sum()
    den = (dir_[0].conj() * (g1 - grad0[0])).sum() + 1e-32
    return [-g1 + dir_[0] * norm2 / den]


def update_single(x, step_length, d):
    return x + step_length * d


def dir_single(x):
    return x


def conjugate_gradient(array_module, x, cost_function, grad,
                       direction_dy=direction_dy, dir_multi=dir_single,
                       update_multi=update_single, num_iter=1, step_length=1,
                       num_search=None, cost=None):
    """Nonlinear conjugate gradient with backtracking (opt.py:312-380)."""
    num_search = num_iter if num_search is None else num_search
    grad0 = dir_ = None
    for i in range(num_iter):
        grad1 = grad(x)
        if i == 0:
            dir_ = direction_dy(array_module, grad1)
        else:
            dir_ = direction_dy(array_module, grad1, grad0, dir_)
        grad0 = grad1
        dir_list = dir_multi(dir_)
        if i < num_search:
            step_length, cost, x = line_search(
                f=cost_function, x=x, d=dir_list, update_multi=update_multi,
                step_length=step_length, cost=cost)
        else:
            x = update_multi(x, step_length, dir_list)
    if num_search < num_iter:
        cost = cost_function(x)
    return x, cost
