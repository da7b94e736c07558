import os
import sys

# (before anything initialises the GPU: see mdmm/__init__.py)
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import helpers  # noqa: E402,F401  (puts the repo root and the package dir on sys.path)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
