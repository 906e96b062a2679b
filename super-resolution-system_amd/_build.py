"""Builds csrc/ into libsrhip.so (gfx950) next to this file.  In-tree build so the .so travels
with the repo snapshot to the GPU box; hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libsrhip.so")
SOURCES = [os.path.join(CSRC, "sr_engine.hip"), os.path.join(CSRC, "sr_host.cpp")]
HEADERS = [os.path.join(CSRC, "sr_internal.h"), os.path.join(_ROOT, "include", "sr_hip.h")]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def build_native(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           # fp-contract off: the fp32 expressions must round exactly like oracle/sr_oracle.c (bit-exact parity).
           # no SLP: hipcc's v_pk_*_f32 packing costs more register shuffling than it saves here (measured).
           "-ffp-contract=off", "-fno-slp-vectorize", "-fvisibility=hidden", "-DSR_BUILD",
           "-I", os.path.join(_ROOT, "include"), "-I", CSRC, "-o", LIB] + SOURCES
    cmd = [c for c in cmd if c] + os.environ.get("SR_HIPCC_EXTRA", "").split()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
