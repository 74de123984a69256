"""Build libpsfm_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python particle-sfm_amd/build.py [--force]

One object per translation unit under csrc/ (rebuilt when the source or a header is newer), linked
into particle-sfm_amd/lib/libpsfm_hip.so.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libpsfm_hip.so")
SOURCES = ["psfm_api.hip", "psfm_track.hip", "psfm_persist.hip", "psfm_finalize.hip", "psfm_sort.hip", "psfm_solver.hip", "psfm_window.hip", "psfm_matches.hip", "psfm_shard.hip", "psfm_batch.hip", "psfm_ingest.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result",
         # counts are aggregated by hand (ballot -> LDS -> one atomic per block); the compiler's own per-wave
         # atomic aggregation would add a readfirstlane + wait after every atomic and serialise them
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]


def source_hash():
    """sha256 (first 16 hex digits) over the sources the library is built from (csrc/*, include/psfm.h, the compile flags): what
    profiles/*.json stamp their PMC figures with and bench.py compares against, so that a replayed figure says which kernels it was
    measured on.  (The .so itself is not hashed: it is git-ignored and rebuilt by the driver.)"""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files.append(os.path.join(os.path.dirname(HERE), "include", "psfm.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    extra = os.environ.get("PSFM_EXTRA_FLAGS", "").split()   # e.g. -DPSFM_TIMELINE (debug builds only)
    if extra:
        FLAGS.extend(extra)
        force = True
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "psfm.h"))
    jobs, objs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append([HIPCC] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
        if verbose and r.stdout.strip():
            print(r.stdout)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _newer(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    if "--source-hash" in sys.argv:
        print(source_hash())
    else:
        print(build(force="--force" in sys.argv, verbose=True))
