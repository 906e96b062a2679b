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
HEADERS = [os.path.join(CSRC, "sr_internal.h"), os.path.join(CSRC, "sr_ctx.h"), os.path.join(CSRC, "sr_march.inc"), os.path.join(CSRC, "sr_down2.inc"),
           os.path.join(_ROOT, "include", "sr_hip.h")]


def _flags():
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
            # fp-contract off: the fp32 expressions must round exactly like oracle/sr_oracle.c (bit-exact parity).
            # no SLP: hipcc's automatic v_pk_*_f32 packing costs more register shuffling than it saves here (measured);
            # where packed fp32 pays (the fused gather) the code spells the pairs out itself.
            "-ffp-contract=off", "-fno-slp-vectorize", "-fvisibility=hidden", "-DSR_BUILD",
            "-I", os.path.join(_ROOT, "include"), "-I", CSRC] + os.environ.get("SR_HIPCC_EXTRA", "").split()


def _flag_digest_input() -> bytes:
    # include paths are the same tree at another mount point on the GPU box: only what changes the code generated counts
    return " ".join(f for f in _flags() if not f.startswith("/") and f != "-I").encode()


def sources_present() -> bool:
    return all(os.path.exists(p) for p in SOURCES + HEADERS)


def source_digest() -> str:
    """sha1 over the sources, the headers AND the compile flags the library is built with (first 16 hex digits): an
    SR_HIPCC_EXTRA experiment can not pass for the default build."""
    h = hashlib.sha1()
    h.update(_flag_digest_input())
    for p in SOURCES + HEADERS:
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode())
            h.update(f.read())
    return h.hexdigest()[:16]


def _object_digest(src: str) -> str:
    """What one object depends on: its source, every header, the flags."""
    h = hashlib.sha1()
    h.update(_flag_digest_input())
    for p in [src] + HEADERS:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


DIGEST_FILE = os.path.join(_HERE, "libsrhip.digest")
LOCK_FILE = os.path.join(_HERE, ".build.lock")


def _stale() -> bool:
    """The library is current iff the digest recorded beside it equals the digest of the sources (and flags) as they are
    now (file times are not trusted: the tree is copied to the GPU box).  A tree shipped WITHOUT csrc/ uses the library it
    came with."""
    if not sources_present():
        if os.path.exists(LIB):
            return False
        raise FileNotFoundError(f"neither {LIB} nor the sources under {CSRC} are present")
    if not os.path.exists(LIB) or not os.path.exists(DIGEST_FILE):
        return True
    with open(DIGEST_FILE) as f:
        return f.read().strip() != source_digest()


def build_native(force: bool = False, verbose: bool = False) -> str:
    """Compiles what changed and links libsrhip.so.  Safe under concurrency (every rank of ``bench.py --gpus N`` and the
    spawn tests call load() at once): one builder at a time under a lock file, objects keyed by a content digest, the
    library linked to a temporary name and moved into place atomically."""
    if not force and not _stale():
        return LIB
    import fcntl
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(LOCK_FILE, "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():                 # another process built it while this one waited
            return LIB
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        digest = source_digest()
        jobs, objs = [], []
        for src in SOURCES:
            stem = os.path.splitext(os.path.basename(src))[0]
            obj, tag = os.path.join(OBJ_DIR, stem + ".o"), os.path.join(OBJ_DIR, stem + ".digest")
            objs.append(obj)
            is_host = src.endswith("sr_host.cpp")      # carries the library digest: always rebuilt (a second of work)
            od = _object_digest(src)
            if not force and not is_host and os.path.exists(obj) and os.path.exists(tag) and open(tag).read().strip() == od:
                continue
            cmd = [hipcc] + _flags() + (["-x", "hip"] if src.endswith(".hip") else []) + \
                  ([f'-DSR_SOURCE_DIGEST="{digest}"'] if is_host else []) + ["-c", src, "-o", obj]
            jobs.append((cmd, tag, od))

        def run(job):
            cmd, tag, od = job
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            if tag and os.path.exists(tag):
                os.remove(tag)                         # an interrupted compile must not leave a matching tag behind
            subprocess.check_call(cmd)
            if tag:
                with open(tag, "w") as f:
                    f.write(od + "\n")

        with ThreadPoolExecutor(max_workers=min(4, max(len(jobs), 1))) as ex:
            list(ex.map(run, jobs))
        tmp = LIB + f".tmp.{os.getpid()}"
        run(([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", "-o", tmp] + objs + ["-lpthread", "-lz", "-ldl"],
             None, None))
        os.replace(tmp, LIB)
        with open(DIGEST_FILE + ".tmp", "w") as f:
            f.write(digest + "\n")
        os.replace(DIGEST_FILE + ".tmp", DIGEST_FILE)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
