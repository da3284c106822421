"""bayestyper_amd — MI355X (gfx950) implementation of BayesTyper's k-mer matching + Gibbs genotyping path.

The product is the C-ABI shared library ``libbtgpu.so`` (include/btgpu.h, sources in ``csrc/``) and the C++
host layer in ``host/`` that mirrors the reference's class interface.  This Python package is only a thin
ctypes binding used by the tests and ``bench.py``; it contains no compute and no CPU fallback: importing
:mod:`bayestyper_amd.lib` fails loudly when the HIP library has not been built.
"""
__all__ = ["lib"]
