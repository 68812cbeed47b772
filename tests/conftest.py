import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


TESTS = os.path.join(ROOT, "tests")
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)   # tests/experiments imports the helpers of tests/

# The kernels that lost their A/B runs live in scripts/experiments and are not in the product library; their parity tests
# (tests/experiments) are collected only on request, against a library built by scripts/build_experiments.sh.
collect_ignore_glob = [] if os.environ.get("PD_TEST_EXPERIMENTS") else ["experiments/*"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experiments: kernels outside the product library (PD_TEST_EXPERIMENTS=1, scripts/build_experiments.sh)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, so a plain `pytest tests` works anywhere."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
