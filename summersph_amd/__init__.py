"""summersph_amd -- MI355X-native SPH core behind SUMMERSPH's simulate() path.

The product is the C-ABI HIP library (include/summersph.h, summersph_amd/csrc) and the thin
Fortran host over it (summersph_amd/host).  This Python package is the harness around them:
ctypes binding for tests and bench, seeded IC generators, the text format, and the
one-process-per-GPU launcher used for multi-GPU runs.
"""
__all__ = ["ic", "txtio"]
