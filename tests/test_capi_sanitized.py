"""The C-ABI layer (dvi_ekf_amd/csrc/eskf_capi.cpp: handle lifetime, argument validation, workspace growth, schedule walking,
stream fork / join, the rules of stream capture) built HOST-ONLY with -fsanitize=address,undefined against a stand-in HIP
runtime whose "kernel launches" touch every buffer over the extent the real kernels touch it (tests/hipstub/), and driven
through the whole C surface.  CPU only: GPU sanitizers are not available on the pool, and this is the part of the library that
is plain host C++."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipstub")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_c_abi_layer_under_address_and_undefined_behaviour_sanitizers():
    subprocess.check_call(["make", "-C", HERE, "-s", "-B"], timeout=600)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([os.path.join(HERE, "_build", "capi_driver")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert "0 failed expectations" in p.stdout and "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr
