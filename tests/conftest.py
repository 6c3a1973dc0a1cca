"""pytest configuration: `gpu` marker, import paths, and on-demand builds of the test artefacts.

`-m "not gpu"` runs everywhere (CPU container): oracle vs golden vectors / reference-on-host, host
logic, C-ABI symbol checks.  `-m gpu` needs a real MI355X and calls through the C-ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    try:  # a CPU-quota'd container on a many-core host: keep torch's intra-op pool inside the quota (hostcpu.py)
        from detectron_pytorch_amd import hostcpu

        hostcpu.respect_cpu_quota()
    except Exception:  # noqa: BLE001
        pass


def _gpu_available():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle (builds oracle/liboracle.so with gcc on first use)."""
    import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def ref_mod():
    """The reference's own sources built for the host (oracle/_ref); skip when unavailable."""
    import subprocess

    from oracle import ref

    if not ref.available() and os.path.isdir("/root/reference/lib"):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "oracle", "build_ref.py")])
    if not ref.available():
        pytest.skip("oracle/_ref not built and /root/reference absent")
    return ref


@pytest.fixture(scope="session")
def hip_lib_path():
    """Path of libmi_detectron_ops.so, building it with hipcc when missing/stale (cross-compiles on CPU)."""
    from detectron_pytorch_amd import build as b

    return b.build(verbose=False)


def load_golden(name):
    import numpy as np

    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)
