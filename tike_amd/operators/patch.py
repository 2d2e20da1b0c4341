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

    @staticmethod
    def _geometry(positions, patches, image_stack, patch_width):
        """Shape rules common to `fwd` and `adj` (the reference's asserts,
        patch.py:96-101,143-157): returns (stack dims, number of images, N)."""
        stack = tuple(positions.shape[:-2])
        assert positions.shape[-1] == 2, positions.shape
        assert patch_width <= patches.shape[-1]
        assert tuple(image_stack) == stack
        assert tuple(patches.shape[:-3]) == stack, (positions.shape,
                                                    patches.shape)
        return stack, int(np.prod(stack, dtype=np.int64)), positions.shape[-2]

    @staticmethod
    def _result(device_array, given, kind):
        """The caller's own array when one was given (filled in place, as the
        reference does), else a new array of the caller's kind."""
        if given is not None and A.is_device(given):
            if device_array.data_ptr() != given.data_ptr():
                given.copy_(device_array)
            return given
        return A.like_input(device_array, kind)

    def fwd(self, images, positions, patches=None, patch_width=0, height=0,
            width=0, nrepeat=1):
        """patches[..., n * nrepeat + r] = the window of `images` at
        positions[..., n] (bilinear), written into the centre of the (possibly
        wider) patch array; allocates zeros when `patches` is None."""
        kind = images
        window = patches.shape[-1] if patch_width == 0 else patch_width
        images = A.to_device(images, np.complex64)
        positions = A.to_device(positions, np.float32)
        if patches is None:
            out = torch.zeros((*positions.shape[:-2],
                               positions.shape[-2] * nrepeat, window, window),
                              dtype=torch.complex64, device=images.device)
        else:
            out = A.to_device(patches, np.complex64)
        _, nimage, count = self._geometry(positions, out, images.shape[:-2],
                                          window)
        assert count * nrepeat == out.shape[-3]
        check(
            lib.tike_patch_fwd(A.ptr(images), A.ptr(out), A.ptr(positions),
                               nimage, images.shape[-2], images.shape[-1],
                               count, nrepeat, window, out.shape[-1],
                               A.stream_ptr()), "Patch.fwd")
        return self._result(out, patches, kind)

    def adj(self, positions, patches, images=None, patch_width=0, height=0,
            width=0, nrepeat=1):
        """images += the patches scattered back with the same bilinear
        weights; K = patches.shape[-3] patches are cycled over the
        N * nrepeat windows (K = 1: one patch added everywhere); allocates a
        zero (height, width) image stack when `images` is None."""
        kind = patches
        patches = A.to_device(patches, np.complex64)
        positions = A.to_device(positions, np.float32)
        window = patches.shape[-1] if patch_width == 0 else patch_width
        if images is None:
            out = torch.zeros((*positions.shape[:-2], height, width),
                              dtype=torch.complex64, device=patches.device)
        else:
            out = A.to_device(images, np.complex64)
        _, nimage, count = self._geometry(positions, patches, out.shape[:-2],
                                          window)
        given = patches.shape[-3]
        assert (count * nrepeat) % given == 0 and given >= nrepeat
        assert patches.shape[-1] == patches.shape[-2]
        check(
            lib.tike_patch_adj(A.ptr(out), A.ptr(patches), A.ptr(positions),
                               nimage, out.shape[-2], out.shape[-1], count,
                               nrepeat, window, patches.shape[-1], given,
                               A.stream_ptr()), "Patch.adj")
        return self._result(out, images, kind)
