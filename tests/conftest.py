import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def _usable_cpus() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                   # containers: every host cpu is visible, the cgroup quota is what counts
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


@pytest.fixture(scope="session", autouse=True)
def _blas_threads_match_the_cpu_quota():
    """The fp64 oracle runs on numpy's BLAS; on the GPU box 256 host cpus are visible but 16 are granted — 256 BLAS
    threads make the oracle 10x slower there."""
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=_usable_cpus()):
            yield
    except ImportError:
        yield


@pytest.fixture
def tuning():
    """Set development knobs of the library for one test (`aid_set_tuning`); every knob touched goes back to "heuristic"
    afterwards.  The library reads AID_* environment variables only once, when it is loaded — never on the launch path."""
    from aid_amd import ops
    touched = []

    def set_(name, value):
        touched.append(name)
        ops.set_tuning(name, value)

    yield set_
    for name in touched:
        ops.set_tuning(name, -1)
