import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# puts 3d-wsis_amd/ and 3d-wsis_amd/model/ on sys.path (drop-in import names)
importlib.import_module("3d-wsis_amd")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experimental: exercises a retired design of DESIGN.md section 8; needs the "
                                       "EXPERIMENTAL build of libwsis_hip.so (make -C 3d-wsis_amd/csrc EXPERIMENTAL=1), "
                                       "skipped on the default build")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def _experimental_build():
    try:
        import wsis_native
        return wsis_native.experimental()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if not _experimental_build():
        skip_x = pytest.mark.skip(reason="default build of libwsis_hip.so: the retired designs are in the EXPERIMENTAL "
                                         "build (make -C 3d-wsis_amd/csrc EXPERIMENTAL=1)")
        for item in items:
            if "experimental" in item.keywords:
                item.add_marker(skip_x)
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
