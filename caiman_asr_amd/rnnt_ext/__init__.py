"""Mirror of the reference's `rnnt_ext` package (training/lib/src/rnnt_ext), backed by
the gfx950 C-ABI library instead of the CUDA pybind11 extensions."""
