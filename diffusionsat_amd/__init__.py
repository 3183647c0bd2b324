"""MI355X-native implementation of DiffusionSAT's denoising-sampling hot path.

The compute path is the HIP extension ``libdsat.so`` (C-ABI in ``include/dsat.h``);
this package is the thin Python host side that keeps the reference's
``DiffusionSampler(model_path, dimacs_filename).samples(n)`` /
``QuerySAT.diffusion_step`` surface.  See DESIGN.md.
"""
__version__ = "0.1.0"
