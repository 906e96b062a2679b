"""Builds csrc/ into libsrhip.so (gfx950) next to this file.  In-tree build so the .so travels
with the repo snapshot to the GPU box; hipcc cross-compiles without a GPU.

Every source is compiled to its own object (in parallel, only when it or a header changed) and the
objects are linked into the shared library; a digest of the sources is compiled in (sr_source_digest)
so a stale binary can be told from a current one without trusting file times."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libsrhip.so")
OBJ_DIR = os.path.join(_HERE, "build")
SOURCES = [os.path.join(CSRC, n) for n in ("sr_engine.hip", "sr_lpips.hip", "sr_adjust.hip", "sr_encode.cpp",
                                           "sr_comm.cpp", "sr_host.cpp")]
HEADERS = [os.path.join(CSRC, "sr_internal.h"), os.path.join(CSRC, "sr_ctx.h"),
           os.path.join(_ROOT, "include", "sr_hip.h")]


def source_digest() -> str:
    """sha1 over the sources and headers the library is built from (first 16 hex digits)."""
    h = hashlib.sha1()
    for p in SOURCES + HEADERS:
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode())
            h.update(f.read())
    return h.hexdigest()[:16]


DIGEST_FILE = os.path.join(_HERE, "libsrhip.digest")


def _stale() -> bool:
    """The library is current iff the digest recorded beside it equals the digest of the sources as they are now
    (file times are not trusted: the tree is copied to the GPU box)."""
    if not os.path.exists(LIB) or not os.path.exists(DIGEST_FILE):
        return True
    with open(DIGEST_FILE) as f:
        return f.read().strip() != source_digest()


def _flags():
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
            # fp-contract off: the fp32 expressions must round exactly like oracle/sr_oracle.c (bit-exact parity).
            # no SLP: hipcc's v_pk_*_f32 packing costs more register shuffling than it saves here (measured).
            "-ffp-contract=off", "-fno-slp-vectorize", "-fvisibility=hidden", "-DSR_BUILD",
            "-I", os.path.join(_ROOT, "include"), "-I", CSRC] + os.environ.get("SR_HIPCC_EXTRA", "").split()


def build_native(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    digest = source_digest()
    hdr_t = max(os.path.getmtime(p) for p in HEADERS)
    jobs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(obj)
        is_host = src.endswith("sr_host.cpp")          # carries the digest: always rebuilt (a second of gcc-class work)
        if not force and not is_host and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
            continue
        cmd = [hipcc] + _flags() + (["-x", "hip"] if src.endswith(".hip") else []) + \
              ([f'-DSR_SOURCE_DIGEST="{digest}"'] if is_host else []) + ["-c", src, "-o", obj]
        jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(4, max(len(jobs), 1))) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", "-o", LIB] + objs + ["-lpthread", "-lz", "-ldl"])
    with open(DIGEST_FILE, "w") as f:
        f.write(digest + "\n")
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
