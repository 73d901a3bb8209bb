#!/usr/bin/env python3
"""Build libmi355kzg.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the
repository snapshot (it is git-ignored but not gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmi355kzg.so")
SOURCES = ["api.hip", "api_verify.hip", "api_cells.hip", "api_multi.hip", "api_exchange.hip", "msm_kernels.hip", "verify_kernels.hip", "cells_kernels.hip", "fk20_kernels.hip", "recover_kernels.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def _deps(src, seen=None):
    """the source and every header it includes, transitively (quoted includes only): a host-side header edit then recompiles the api*.hip
    files and not the kernel files, which take minutes"""
    import re
    seen = set() if seen is None else seen
    src = os.path.normpath(src)
    if src in seen or not os.path.exists(src):
        return seen
    seen.add(src)
    with open(src) as fh:
        for inc in re.findall(r'^\s*#include\s+"([^"]+)"', fh.read(), flags=re.M):
            _deps(os.path.join(os.path.dirname(src), inc), seen)
    return seen


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    m = os.path.getmtime(target)
    return any(os.path.getmtime(d) > m for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(os.path.join(CSRC, "constants29.h")):
        subprocess.check_call([sys.executable, os.path.join(CSRC, "gen_constants.py")])
    # generated, not tracked (600 KB of digit strings): the FK20 transform stage's fixed scalars and network programs
    if _stale(os.path.join(CSRC, "fk20_toom.h"), [os.path.join(CSRC, "gen_fk20_toom.py")]):
        subprocess.check_call([sys.executable, os.path.join(CSRC, "gen_fk20_toom.py")], stdout=subprocess.DEVNULL)
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, sorted(_deps(src))):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            for cmd, rc in zip(jobs, ex.map(lambda c: subprocess.run(c, capture_output=not verbose, text=True), jobs)):
                if rc.returncode != 0:
                    raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), (rc.stderr or "")[-4000:]))
    if force or jobs or _stale(OUT, objs):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"])      # api_exchange.hip dlopens librccl
    return OUT


def _main():
    try:
        print(build(force="--force" in sys.argv, verbose=True))
    except Exception as e:                      # the LAST line of the output must say so (people pipe this through tail)
        print(str(e)[-3000:])
        print("BUILD FAILED")
        sys.exit(1)


if __name__ == "__main__":
    _main()
