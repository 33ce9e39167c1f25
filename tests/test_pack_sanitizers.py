"""The host-side weight packer (~1 000 lines of index arithmetic on a thread pool, audiosourcesep_amd/csrc/glowk_pack.h) built
WITHOUT HIP under AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer, run on every instantiated level shape
and checked against a scalar reading of the image layouts (tests/pack_sanitize_main.cpp).  GPU sanitizers are not available on
this pool; the packer is the host code they would have covered."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "pack_sanitize_main.cpp")


def build_and_run(tmp_path, name, flags, env=None):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / name)
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-pthread"] + flags + [SRC, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    r = subprocess.run([exe, "4"], capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0 and "PACK_SANITIZE_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-6000:])


def test_packer_under_address_and_ub_sanitizers(tmp_path):
    build_and_run(tmp_path, "pack_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"],
                  {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})


def test_packer_thread_pool_under_thread_sanitizer(tmp_path):
    build_and_run(tmp_path, "pack_tsan", ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=1"})
