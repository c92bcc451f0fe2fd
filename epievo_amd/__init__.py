"""epievo_amd -- MI355X (gfx950) implementation of epievo's MCEM inner loop: the per-site
Metropolis-Hastings end-conditioned CTMC path sampler, behind a C ABI
(include/epievo_mi355x.h) with a SingleSiteSampler-shaped host interface.

  epievo_amd.sampler   ctypes binding of the HIP library + SingleSiteSampler mirror
  epievo_amd.host      ctypes binding of the C++ host library (model, M-step, formats,
                       synthetic-input simulator)
  epievo_amd._build    in-tree builds (hipcc --offload-arch=gfx950, g++)
"""
__all__ = ["sampler", "host"]
