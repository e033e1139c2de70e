"""stoch_gpmp_amd -- the StochGPMP inner loop as hand-written HIP kernels for MI355X (gfx950).

Drop-in for the `StochGPMP.optimize()` / `CostComposite.eval()` path of anindex/stoch_gpmp:
the module layout and class names mirror `stoch_gpmp.*`; all arithmetic on trajectory batches
runs in libsgpmp.so (include/sgpmp.h).  There is no CPU or torch fallback.
"""
__version__ = "0.1.0"
