"""geobipy_amd -- MI355X (gfx950) implementation of GeoBIPy's per-sounding hot path.

Scope: the 1-D layered-earth FDEM and TDEM forward solves, the FDEM Jacobian and the
Gaussian data misfit / log-likelihood evaluated at every rjMCMC proposal, behind the reference's
``FdemSystem / TdemSystem / Model / FdemDataPoint / TdemDataPoint`` interface, plus batched device-resident
entry points (``FdemBatch``, ``TdemBatch``) and the rjMCMC step around them (``Inference1D`` with the reference's random
streams, ``DeviceChains`` resident on the GPU).
All arithmetic runs in hand-written HIP kernels (geobipy_amd/csrc); importing the classes works on
a CPU-only machine, evaluating anything needs the built library and a GPU -- there is no fallback.
"""
from .system import CircularLoop, FdemSystem
from .model import Model, RectilinearMesh1D
from .datapoint import FdemDataPoint
from .batch import FdemBatch
from .tdem import TdemBatch, TdemDataPoint, TdemDeviceChains, TdemSystem, TempestDataPoint
from .inference import BatchedInference, Inference1D
from .rjmcmc_gpu import DeviceChains
from . import rjmcmc, survey, synthetic

__all__ = ["CircularLoop", "FdemSystem", "Model", "RectilinearMesh1D", "FdemDataPoint", "FdemBatch", "TdemSystem", "TdemDataPoint", "TempestDataPoint", "TdemBatch", "TdemDeviceChains",
           "Inference1D", "BatchedInference", "DeviceChains", "rjmcmc", "survey", "synthetic"]
