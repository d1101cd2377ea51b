"""The C ABI used from plain C: tests/c_abi_consumer.c links libic_hip.so and the HIP runtime only
(built by __graft_entry__.build()), runs a search + histogram + centroid pass and checks it
against its own float64 brute force."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_plain_c_consumer():
    exe = os.path.join(HERE, "c_abi_consumer")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=100)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "0 problems" in out.stdout
