"""Patch extraction / scatter-add (reference operators/cupy/patch.py:59-188).

Device work: ``tike_patch_fwd`` / ``tike_patch_adj`` (csrc/patch.hip), the HIP
replacement of the reference's CUDA ``fwd_patch`` / ``adj_patch``.
"""
import numpy as np
import torch

from .. import _arrays as A
from .._lib import check, lib
from .operator import Operator


class Patch(Operator):
    """Extract (zero-padded) patches from images at provided positions.

    images (..., H, W) complex64; positions (..., N, 2) float32 (y, x) of the
    minimum corner; patches (..., N * nrepeat, width+, width+) complex64 or
    (..., K, width+, width+) with (N * nrepeat) % K == 0 for the adjoint.
    """

    def fwd(self, images, positions, patches=None, patch_width=0, height=0,
            width=0, nrepeat=1):
        kind = images
        patch_width = patches.shape[-1] if patch_width == 0 else patch_width
        images = A.to_device(images, np.complex64)
        positions = A.to_device(positions, np.float32)
        lead = tuple(positions.shape[:-2])
        if patches is None:
            patches_t = torch.zeros(
                (*lead, positions.shape[-2] * nrepeat, patch_width,
                 patch_width), dtype=torch.complex64, device=images.device)
        else:
            patches_t = A.to_device(patches, np.complex64)
        assert patch_width <= patches_t.shape[-1]
        assert tuple(images.shape[:-2]) == lead
        assert tuple(patches_t.shape[:-3]) == lead, (positions.shape,
                                                     patches_t.shape)
        assert positions.shape[-2] * nrepeat == patches_t.shape[-3]
        assert positions.shape[-1] == 2, positions.shape
        nimage = int(np.prod(lead)) if lead else 1
        check(
            lib.tike_patch_fwd(A.ptr(images), A.ptr(patches_t),
                               A.ptr(positions), nimage, images.shape[-2],
                               images.shape[-1], positions.shape[-2], nrepeat,
                               patch_width, patches_t.shape[-1],
                               A.stream_ptr()), "Patch.fwd")
        if patches is not None and A.is_device(patches) and \
                patches_t.data_ptr() != patches.data_ptr():
            patches.copy_(patches_t)
            return patches
        return A.like_input(patches_t, kind)

    def adj(self, positions, patches, images=None, patch_width=0, height=0,
            width=0, nrepeat=1):
        kind = patches
        patches = A.to_device(patches, np.complex64)
        positions = A.to_device(positions, np.float32)
        patch_width = patches.shape[-1] if patch_width == 0 else patch_width
        assert patch_width <= patches.shape[-1]
        lead = tuple(positions.shape[:-2])
        if images is None:
            images_t = torch.zeros((*lead, height, width),
                                   dtype=torch.complex64,
                                   device=patches.device)
        else:
            images_t = A.to_device(images, np.complex64)
        height, width = images_t.shape[-2:]
        assert tuple(images_t.shape[:-2]) == lead
        N = positions.shape[-2]
        assert positions.shape[-1] == 2
        assert tuple(patches.shape[:-3]) == lead
        K = patches.shape[-3]
        assert (N * nrepeat) % K == 0 and K >= nrepeat
        assert patches.shape[-1] == patches.shape[-2]
        nimage = int(np.prod(lead)) if lead else 1
        check(
            lib.tike_patch_adj(A.ptr(images_t), A.ptr(patches),
                               A.ptr(positions), nimage, height, width, N,
                               nrepeat, patch_width, patches.shape[-1], K,
                               A.stream_ptr()), "Patch.adj")
        if images is not None and A.is_device(images) and \
                images_t.data_ptr() != images.data_ptr():
            images.copy_(images_t)
            return images
        return A.like_input(images_t, kind)
