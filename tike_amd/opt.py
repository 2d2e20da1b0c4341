"""Generic optimisation helpers (reference src/tike/opt.py).

Work on torch tensors (device) or NumPy arrays alike.
"""
import logging
import warnings

import numpy as np
import torch

logger = logging.getLogger(__name__)


def _recent_cost_slope(history, window):
    """Slope of the least-squares line through the last `window` epoch costs
    (an epoch's minibatch costs are averaged first)."""
    per_epoch = np.asarray(history[-window:], dtype=float).reshape(
        window, -1).mean(axis=1)
    return np.polyfit(np.arange(window), per_epoch, 1)[0]


def is_converged(algorithm_options):
    """Has the cost stopped falling?  Every half window the slope of the
    recent costs is examined; a slope that is not negative means converged
    (the criterion of the reference's opt.is_converged, opt.py:21-45)."""
    window = algorithm_options.convergence_window
    epochs = len(algorithm_options.costs)
    if window < 2 or epochs < window or epochs % window >= 2:
        return False
    flat = _recent_cost_slope(algorithm_options.costs, window) >= 0
    if flat:
        logger.info("cost has not decreased over the last %d epochs", window)
    return bool(flat)


def batch_indicies(n, m=1, use_random=True):
    """The indices 0 .. n-1 in m groups of (nearly) equal size, shuffled with
    `tike_amd.random.randomizer_np` unless use_random is False (opt.py:47-56)."""
    if not 0 < m <= n:
        raise AssertionError((m, n))
    from . import random as trandom
    pool = trandom.randomizer_np.permutation(n) if use_random else np.arange(n)
    return np.array_split(pool, m)


def get_batch(x, b, n):
    """Rows of x that belong to batch n of the index lists b."""
    return x[b[n]]


def put_batch(y, x, b, n):
    """Store y as the rows of x that belong to batch n of b."""
    x[b[n]] = y


def momentum(g, v, m, vdecay=None, mdecay=0.9):
    """Exponential moving average of the search direction: returns
    (direction, None, average) like `adam` (opt.py:67-82)."""
    average = (1 - mdecay) * g if m is None else mdecay * m + (1 - mdecay) * g
    return average, None, average


def adam(g, v=None, m=None, vdecay=0.999, mdecay=0.9, eps=1e-8):
    """Adaptive-moment direction (Kingma & Ba, arXiv:1412.6980) in the variant
    the reference uses (opt.py:165-213): both moving averages are divided by
    their single-step weights (1 - decay), not by 1 - decay**t.  Returns
    (direction, second moment, first moment)."""
    root = torch.sqrt if isinstance(g, torch.Tensor) else np.sqrt
    power = (g * g.conj()).real
    first = (1 - mdecay) * g
    second = (1 - vdecay) * power
    if m is not None:
        first = mdecay * m + first
    if v is not None:
        second = vdecay * v + second
    direction = (first / (1 - mdecay)) / (root(second / (1 - vdecay)) + eps)
    return direction, second, first


def fit_line_least_squares(y, x):
    """(slope, intercept) of the least-squares line y = slope * x + intercept
    (opt.py:383-400), from the centred moments."""
    x = np.asarray(x, dtype=float)
    y = np.asarray(y, dtype=float)
    if len(x) != len(y) or not len(x):
        raise AssertionError("x and y must be equally long and not empty")
    dx = x - x.mean()
    slope = np.dot(dx, y - y.mean()) / np.dot(dx, dx)
    return slope, y.mean() - slope * x.mean()


def _shrinking_steps(first, factor, floor=1e-32):
    """first, first * factor, first * factor**2, ... while >= floor (the first
    one always)."""
    step = first
    while True:
        yield step
        step *= factor
        if step < floor:
            return


def line_search(f, x, d, update_multi, step_length=1, step_shrink=0.5,
                cost=None):
    """Backtracking line search (opt.py:216-278): the first of the step
    lengths step_length * step_shrink**k (>= 1e-32) whose cost does not exceed
    f(x).  Returns (step, cost at the accepted point, accepted point); when no
    step is accepted, (0, f(x), x) with a warning."""
    assert 0 < step_shrink < 1
    start_cost = f(x) if cost is None else cost
    for step in _shrinking_steps(step_length, step_shrink):
        candidate = update_multi(x, step, d)
        candidate_cost = f(candidate)
        if candidate_cost <= start_cost:
            return step, candidate_cost, candidate
    warnings.warn("Line search failed for conjugate gradient.")
    return 0, start_cost, x


def direction_dy(xp, grad1, grad0=None, dir_=None):
    """Dai-Yuan conjugate direction -g1 + beta * d with
    beta = |g1|^2 / <d, g1 - g0> (opt.py:281-301).  Arguments and result are
    one-element lists (one array per device in the reference)."""
    g1 = grad1[0]
    if dir_ is None:
        return [-g1]
    d = dir_[0]
    beta = (g1 * g1.conj()).real.sum() / ((d.conj() *
                                           (g1 - grad0[0])).sum() + 1e-32)
    return [d * beta - g1]


def update_single(x, step_length, d):
    return x + step_length * d


def dir_single(x):
    return x


class _ConjugateDirections:
    """The part of nonlinear CG that remembers: turns each new gradient into
    the next search direction with the given rule (Dai-Yuan by default)."""

    def __init__(self, array_module, rule, spread):
        self.xp, self.rule, self.spread = array_module, rule, spread
        self.previous = ()  # (gradient, direction) of the last iteration

    def __call__(self, gradient):
        direction = self.rule(self.xp, gradient, *self.previous)
        self.previous = (gradient, direction)
        return self.spread(direction)


def conjugate_gradient(array_module, x, cost_function, grad,
                       direction_dy=direction_dy, dir_multi=dir_single,
                       update_multi=update_single, num_iter=1, step_length=1,
                       num_search=None, cost=None):
    """Nonlinear conjugate gradient (opt.py:312-380): `num_iter` Dai-Yuan
    directions; the first `num_search` of them (default: all) are followed by
    a backtracking line search that starts from the step length the previous
    search accepted, the others reuse that step length.  Returns (x, cost)."""
    searches = num_iter if num_search is None else min(num_search, num_iter)
    blind = num_iter - searches
    next_move = _ConjugateDirections(array_module, direction_dy, dir_multi)
    while searches > 0:
        searches -= 1
        found = line_search(cost_function, x, next_move(grad(x)),
                            update_multi, step_length, cost=cost)
        step_length, cost, x = found
    if blind == 0:
        return x, cost
    while blind > 0:  # the rest reuse the last accepted step length
        blind -= 1
        x = update_multi(x, step_length, next_move(grad(x)))
    return x, cost_function(x)
