#!/usr/bin/env python3
"""Build an EXPERIMENTAL copy of libpsfm_hip.so with extra -D flags for one TU (measurement only; the product library is
particle-sfm_amd/lib/libpsfm_hip.so, built by particle-sfm_amd/build.py).

    python scripts/build_variant.py NAME psfm_persist.hip -DPP_SKEW=1 ...

writes particle-sfm_amd/lib/variants/libpsfm_hip_NAME.so (git-ignored; travels to the GPU box); select it with
PSFM_HIP_LIB=<path>.  The other objects come from particle-sfm_amd/build/ (run build.py first)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import build as psfm_build


def main():
    name, tu, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    psfm_build.build()
    vdir = os.path.join(psfm_build.LIBDIR, "variants")
    os.makedirs(vdir, exist_ok=True)
    obj = os.path.join(psfm_build.OBJ, "variant_%s_%s" % (name, tu.replace(".hip", ".o")))
    subprocess.check_call([psfm_build.HIPCC] + psfm_build.FLAGS + flags + ["-c", os.path.join(psfm_build.CSRC, tu), "-o", obj])
    objs = [obj if s == tu else os.path.join(psfm_build.OBJ, s.replace(".hip", ".o")) for s in psfm_build.SOURCES]
    out = os.path.join(vdir, "libpsfm_hip_%s.so" % name)
    subprocess.check_call([psfm_build.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    print(out)


if __name__ == "__main__":
    main()
