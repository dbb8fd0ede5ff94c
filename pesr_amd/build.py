"""Build libpesr_hip.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs both in the CPU-only build container and on
the GPU box.  The .so lands next to this file (pesr_amd/libpesr_hip.so) so that it travels with
the repository snapshot; it is git-ignored.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpesr_hip.so")
STAMP = LIB + ".stamp"
ARCH = "gfx950"
# -amdgpu-mfma-vgpr-form: MFMA accumulators stay in VGPRs.  Left to its heuristic hipcc parks the accumulators of the smaller
# kernels (the 256-thread direct-conv configurations above all) in AGPRs and moves them out and back around every loop
# iteration: v_accvgpr_read / _write for each accumulator register, each write waiting for the MFMAs it follows.
HIPCC_FLAGS = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-fvisibility=hidden", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
               "-Wno-unused-result"]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest() -> str:
    h = hashlib.sha256()
    h.update(" ".join(HIPCC_FLAGS).encode())          # a change of the compile command (flags, arch) rebuilds too
    files = _sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    files.append(os.path.join(HERE, "..", "include", "pesr_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())      # names, not absolute paths: the tree is moved to the GPU box as it is
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def is_current() -> bool:
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as fh:
        return fh.read().strip() == _digest()


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every .hip under csrc/ into one shared object. Returns the library path."""
    if not force and is_current():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objs = []
    build_dir = os.path.join(HERE, "build")
    os.makedirs(build_dir, exist_ok=True)
    procs = []
    for src in _sources():
        obj = os.path.join(build_dir, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [hipcc] + HIPCC_FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, "-shared", f"--offload-arch={ARCH}", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as fh:
        fh.write(_digest())
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
