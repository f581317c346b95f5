"""ceno_amd — MI355X-native GKR / sumcheck prover core for the Ceno zkVM.

Layout: csrc/ (HIP kernels + C ABI, libceno_hip.so), host/ (C++ host layer mirroring the
reference's prover interface, libceno_prover.so), api.py / prover.py (ctypes bindings),
dist.py (hypercube sharding over torch.distributed / RCCL).
There is no CPU fallback: the bindings raise if the HIP libraries are missing.
"""
from .api import CenoHipError, Device, Mle, Sumcheck  # noqa: F401

__all__ = ["CenoHipError", "Device", "Mle", "Sumcheck"]
