import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """CPU-side libraries must exist before any test (the GPU box uses the prebuilt files)."""
    import oracle
    oracle.build()
    lib = os.path.join(ROOT, "nano-vllm-rs_amd", "libnvllm_attn.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()


class _Policy:
    """Pins launch decisions through nvllm_debug_set_policy (the library reads the NVLLM_*
    environment only once, so a test cannot steer it with setenv); everything is restored to the
    environment's values when the test ends."""

    def setenv(self, name, value):
        import nano_vllm_rs_amd as nv
        v = {"a": 1, "b": 0}.get(value, value) if isinstance(value, str) else value
        rc = nv.capi.lib().nvllm_debug_set_policy(name.encode(), int(v))
        assert rc == 0, nv.capi.last_error()


@pytest.fixture
def policy():
    import nano_vllm_rs_amd as nv
    yield _Policy()
    nv.capi.lib().nvllm_debug_reset_policy()
