"""Random complex arrays (reference src/tike/random.py:10-26)."""
import numpy as np

from . import precision

randomizer_np = np.random.default_rng()


def numpy_complex(*shape):
    """Complex random array with parts uniform in [-0.5, 0.5)."""
    return (randomizer_np.random(size=(*shape, 2), dtype=precision.floating) -
            0.5).view(precision.cfloating)[..., 0]
