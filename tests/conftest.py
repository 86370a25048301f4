import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_port():
    """Our C restatement of cmatrices.c (test infrastructure)."""
    from oracle import binding
    if not os.path.exists(binding.PORT_SO):
        binding.build()
    return binding.port()


@pytest.fixture(scope="session")
def oracle_ref():
    """The reference's own cmatrices.c compiled unmodified (present when built in the dev container)."""
    from oracle import binding
    if not binding.have_ref():
        if os.path.isdir("/root/reference/radiomics/src"):
            binding.build()
        if not binding.have_ref():
            pytest.skip("oracle/_ref/libcmatrices_ref.so not available")
    return binding.ref()


@pytest.fixture(scope="session")
def checker(oracle_port):
    """The CPU checker the GPU parity tests compare against: the reference's OWN cmatrices.c (oracle/_ref, built in the
    dev container and shipped to the GPU box with the repo snapshot) when present, else our C restatement (which
    tests/test_oracle.py pins bit-for-bit to the reference)."""
    from oracle import binding
    return binding.ref() if binding.have_ref() else oracle_port
