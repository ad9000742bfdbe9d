import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver's GPU tier)")
    config.addinivalue_line("markers", "slow: GPU test that also runs minutes of CPU oracle work (`-m 'gpu and not slow'` skips it)")
    config.addinivalue_line("markers", "perf: wall-clock assertions (host enqueue cost, graph vs eager time); NOT part of `-m gpu` -- "
                                       "run with `-m perf` on a quiet box")


# Cheap, wide parity first; the heaviest full-width runs last -- with `-x` one fault must not hide the operator suite
# (round 4: an abort in the graph test, then tenth in line, cost the driver's run 190 tests).
_ORDER = ["test_host_cpu", "test_oracle_golden", "test_ref_interop_cpu", "test_parallel_gloo",
          "test_gpu_ops", "test_gpu_golden", "test_gpu_parity_full", "test_gpu_parity_targets", "test_gpu_tools",
          "test_gpu_world", "test_gpu_graph", "test_gpu_stress", "test_gpu_perf"]


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no GPU is visible, so a plain `pytest tests/`
    on the CPU container stays green; `-m gpu` on the GPU box runs them for real."""
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(name) if name in _ORDER else len(_ORDER)
    items.sort(key=rank)               # stable: the order inside a file is kept
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords or "perf" in item.keywords:
            item.add_marker(skip)
