"""tike_amd -- MI355X-native ptychography hot path behind tike's Operator API.

Drop-in for the hot path of AdvancedPhotonSource/tike: ``tike_amd.operators``
mirrors ``tike.operators`` (Ptycho / Propagation / Convolution / Patch and the
objective functions) and ``tike_amd.ptycho`` mirrors ``tike.ptycho``
(``reconstruct``, ``Reconstruction``, ``simulate``, the option dataclasses and
the ``lstsq_grad`` / ``cgrad`` solvers).  All device arithmetic runs in
hand-written HIP kernels for gfx950 (``tike_amd/csrc``) reached through a
C ABI (``include/tike_amd.h``); PyTorch only provides device memory, streams
and ``torch.distributed`` (RCCL).
"""
__version__ = "0.1.0"
