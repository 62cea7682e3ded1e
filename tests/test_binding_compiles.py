"""INTEGRATION.md §3a is not prose only: oracle/binding_check.cpp (the replay of the device's report events into AlnRes / AlnSinkWrap::report)
compiles against the reference's own headers with the reference's flags (oracle/Makefile.ref).  Needs /root/reference."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/root/reference/hi_aligner.h"), reason="needs the reference tree")
def test_reference_side_binding_compiles(tmp_path):
    out = str(tmp_path)
    r = subprocess.run(["make", "-f", os.path.join(ROOT, "oracle", "Makefile.ref"), "OUT=" + out, os.path.join(out, "obj", "binding_check.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert os.path.getsize(os.path.join(out, "obj", "binding_check.o")) > 1000
