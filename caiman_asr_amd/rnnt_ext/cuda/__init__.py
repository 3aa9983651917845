"""Same module path as the reference's compiled extensions (`rnnt_ext.cuda.*`,
training/lib/setup.py:10-29).  "cuda" is kept for drop-in imports; the device code is HIP."""
