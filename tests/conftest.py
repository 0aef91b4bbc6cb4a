import os
import sys
from pathlib import Path

import pytest

# development knobs / fault-injection hooks of the library are only read when this is set (csrc/common.hpp: dev_env)
os.environ.setdefault("MHIP_DEVELOPER", "1")

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _gpu_present():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU (/dev/kfd missing)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
