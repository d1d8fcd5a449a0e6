import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    """The HIP device handle of the product library.  GPU tests only: no fallback, fails loudly."""
    from libacm_amd import capi
    d = capi.Device(0)
    # GPU tests test the GPU: acm_read() of the drop-in API goes to the device whatever the stream's length (by default streams below
    # 128 Msamples are synthesised on the host while no device is open: include/acm_hip.h, acmhip_set_host_synth_limit)
    L = capi.lib()
    prev = L.acmhip_host_synth_limit()
    L.acmhip_set_host_synth_limit(0)
    yield d
    L.acmhip_set_host_synth_limit(prev)
    d.close()
