"""Inter-GPU communication over RCCL (mirror of ``tike.communicators``)."""
from .comm import Comm

__all__ = ["Comm"]
