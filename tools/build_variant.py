"""Measurement build: the library with ONE translation unit recompiled under extra flags (-D switches of an experiment),
linked with the current objects of the others, as caiman_asr_amd/lib/variants/libcaiman_<NAME>.so.  Run it through
CAIMAN_LIB_OVERRIDE=<that path> (caiman_asr_amd/_lib.py) to A/B compile-time variants on one box.
python tools/build_variant.py NAME file.hip [-DX=1 ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from caiman_asr_amd import _lib  # noqa: E402

name, unit, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
_lib.build()
vdir = os.path.join(_lib.LIB_DIR, "variants")
os.makedirs(vdir, exist_ok=True)
objs = []
for src in _lib.sources():
    obj = os.path.join(_lib.LIB_DIR, "obj", os.path.basename(src) + ".o")
    if os.path.basename(src) == unit:
        obj = os.path.join(vdir, f"{unit}.{name}.o")
        subprocess.check_call([_lib.HIPCC, *_lib.HIP_FLAGS, *_lib.FILE_FLAGS.get(unit, []), *extra, "-c", src, "-o", obj])
    objs.append(obj)
out = os.path.join(vdir, f"libcaiman_{name}.so")
subprocess.check_call([_lib.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
print(out)
