"""The C++ adapters must compile against the reference's own headers (only checkable where
/root/reference exists; the GPU box does not have it)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference headers not present")
def test_adapters_compile_against_reference_headers():
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-Wno-pragma-once-outside-header",
           "-I" + os.path.join(ROOT, "include"), "-I/root/reference/src", "-I/root/reference/includes",
           "-x", "c++", os.path.join(ROOT, "fba_pomdp_amd", "csrc", "host", "adapters.hpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
