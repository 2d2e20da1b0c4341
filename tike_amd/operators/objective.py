"""Cost functions and their far-plane gradients
(reference operators/cupy/objective.py:11-124).  Mean instead of sum so that
costs of mini-batches of different sizes are comparable."""
import numpy as np
import torch

from .. import _arrays as A
from .._lib import check, lib

_MODELS = {"gaussian": 0, "poisson": 1}


def _each(data, intensity, model):
    kind = intensity
    intensity = A.to_device(intensity, np.float32)
    data = A.to_device(data, np.float32).expand_as(intensity).contiguous()
    lead = intensity.shape[:-2] if intensity.ndim >= 2 else ()
    npix = int(np.prod(intensity.shape[len(lead):]))
    n = int(np.prod(lead)) if lead else 1
    costs = torch.empty(n, dtype=torch.float32, device=intensity.device)
    check(
        lib.tike_cost_each_pattern(A.ptr(data), A.ptr(intensity), A.ptr(costs),
                                   n, npix, _MODELS[model], A.stream_ptr()),
        f"{model}_each_pattern")
    return A.like_input(costs.reshape(tuple(lead)), kind)


def _grad(data, farplane, intensity, model):
    kind = farplane
    farplane = A.to_device(farplane, np.complex64)
    intensity = A.to_device(intensity, np.float32)
    data = A.to_device(data, np.float32).expand_as(intensity).contiguous()
    n = intensity.shape[0]
    npix = intensity.shape[-2] * intensity.shape[-1]
    S = farplane.numel() // (n * npix)
    out = torch.empty_like(farplane)
    check(
        lib.tike_objective_grad(A.ptr(data), A.ptr(farplane), A.ptr(intensity),
                                A.ptr(out), n, S, npix, _MODELS[model],
                                A.stream_ptr()), f"{model}_grad")
    return A.like_input(out, kind)


def gaussian_each_pattern(data, intensity):
    """mean((sqrt(I) - sqrt(d))^2) per pattern (objective.py:47-66)."""
    return _each(data, intensity, "gaussian")


def gaussian(data, intensity):
    """objective.py:18-28."""
    c = gaussian_each_pattern(data, intensity)
    return c.mean()


def gaussian_grad(data, farplane, intensity):
    """farplane * (1 - sqrt(d) / (sqrt(I) + 1e-9)) (objective.py:31-44)."""
    return _grad(data, farplane, intensity, "gaussian")


def poisson_each_pattern(data, intensity):
    """mean(I - d log(I + 1e-9)) per pattern (objective.py:112-124)."""
    return _each(data, intensity, "poisson")


def poisson(data, intensity):
    """objective.py:77-94."""
    c = poisson_each_pattern(data, intensity)
    return c.mean()


def poisson_grad(data, farplane, intensity):
    """farplane * (1 - d / (I + 1e-9)) (objective.py:97-109)."""
    return _grad(data, farplane, intensity, "poisson")
