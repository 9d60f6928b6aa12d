"""MI355X-native uncertainty-guided diffusion sampling engine (hot path only).

Python host side mirroring the reference's duck-typed API
(``init_model`` / ``generate_samples`` / ``get_uncertainty_scheduler`` /
scheduler ``.step()``) on top of a C-ABI HIP library (``include/dua_hip.h``).
There is no CPU fallback: every compute entry point raises if the HIP library
or a GPU tensor is missing.
"""
__version__ = "0.1.0"
