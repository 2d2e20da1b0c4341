"""Helpers for ptychographic deep learning (mirror of ``tike.ptycho.learn``)."""
from ..operators import Patch
from .position import check_allowed_positions


def extract_patches(psi, scan, patch_width):
    """Patches of `psi` (..., WIDE, HIGH) at the scan positions (..., POSI, 2),
    bilinearly interpolated by the Patch operator's HIP kernel
    (learn.py:10-39).  Returns a NumPy array (..., POSI, width, width)."""
    check_allowed_positions(scan, psi, (patch_width, patch_width))
    with Patch() as operator:
        patches = operator.fwd(images=operator.asarray(psi),
                               positions=operator.asarray(scan),
                               patch_width=patch_width)
        return operator.asnumpy(patches)
