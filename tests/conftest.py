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
def gpu_ctx_factory():
    """Factory for product contexts; skips nothing: on a GPU box a missing library is a failure."""
    import hessgpu_amd

    made = []

    # HESS_TEST_DEV_BUILD=1 (tools/robustness.sh): every context of the suite comes from the developer build, which reads
    # the schedule switches the matrix sets (HESS_NO_TOP_FUSION, HESS_CHAIN_FROM, ...) from the environment
    dev_default = os.environ.get("HESS_TEST_DEV_BUILD") == "1"

    def make(**overrides):
        overrides.setdefault("dev_switches", dev_default)
        c = hessgpu_amd.HessContext(0, **overrides)
        made.append(c)
        return c

    yield make
    for c in made:
        c.close()
