"""AddressSanitizer + UBSan over the host side of the C ABI: csrc/sr_host.cpp and csrc/sr_encode.cpp are
compiled with g++ -fsanitize=address,undefined -fno-sanitize-recover=all together with tools/sanitize/host_driver.cpp, which
calls every host-only entry point with exact-size buffers over randomised geometries.  (GPU sanitizers are not available
on the pool; the kernels' indexing is covered by the poisoned-buffer tests of the GPU suite.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_host_abi_under_sanitizers(tmp_path, san):
    """address,undefined: memory and arithmetic errors; thread: data races of the multi-threaded writers (strips / chunks /
    MCU rows are claimed from an atomic counter and written to disjoint buffers)."""
    csrc = os.path.join(ROOT, "super-resolution-system_amd", "csrc")
    exe = str(tmp_path / "host_driver")
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=" + san, "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-DSR_BUILD", "-I", os.path.join(ROOT, "include"), "-I", csrc, os.path.join(ROOT, "tools", "sanitize", "host_driver.cpp"),
           os.path.join(csrc, "sr_host.cpp"), os.path.join(csrc, "sr_encode.cpp"), "-lz", "-lpthread", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=600)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "sanitizer-driver-ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
