"""The C++ facade (tracker::FeatureDetector / contrastFunctor / ContrastBatch with the
reference's names) built with the host compiler and run on the GPU box."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CPP = os.path.join(HERE, "cpp")


def test_facade_compiles_against_the_abi(ebo, orc):
    """CPU: the facade headers + test program compile and link against libebo_hip.so."""
    ebo.lib()
    subprocess.check_call(["make", "-s", "-C", CPP, "facade_test"])
    assert os.path.exists(os.path.join(CPP, "facade_test"))


@pytest.mark.gpu
def test_facade_against_oracle_on_gpu(ebo, orc):
    ebo.lib()
    subprocess.check_call(["make", "-s", "-C", CPP, "facade_test"])
    out = subprocess.run([os.path.join(CPP, "facade_test")], capture_output=True, text=True, timeout=600)
    print(out.stdout[-3000:], out.stderr[-2000:])
    assert out.returncode == 0, out.stdout[-3000:]
    assert "all passed" in out.stdout
