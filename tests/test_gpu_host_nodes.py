"""Runs the C++ host-graph DSP node tests (tests/host/test_nodes_gpu.cpp) on the GPU box:
the reference's own source -> node -> CheckNode tests, through the C++ Node runtime + C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_dsp_nodes_in_graphs():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "comms_rs_amd", "host"), "-s"], timeout=600)
    out = subprocess.run([os.path.join(ROOT, "comms_rs_amd", "lib", "test_nodes_gpu")], capture_output=True,
                         text=True, timeout=300, cwd=ROOT)   # (reads tests/golden/reference_kats.json)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all passed" in out.stdout
