"""MI355X-native (gfx950) RNN-T hot path behind the reference's operator boundary.

Layout
  csrc/      hand-written HIP kernels + the C-ABI (include/caiman_rnnt.h)
  _lib.py    builds / loads lib/libcaiman_rnnt.so through ctypes (raw pointers only)
  rnnt_ext/  mirror of the reference's `rnnt_ext` package (training/lib/src/rnnt_ext):
             same module paths, function names, argument order and error behaviour
  rnnt/      mirror of the model-level interface (training/caiman_asr_train/rnnt)
  ...
There is NO CPU fallback: every op raises if the HIP library is missing.
"""
__version__ = "0.1.0"
