import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'no_quiet: the test creates pending device work on purpose; no device synchronise is added around it')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(autouse=True)
def _device_quiet_between_gpu_tests(request):
    """every GPU test starts on an idle device: collect the previous test's garbage and wait for all of its work (round 5: a training graph built
    while the previous test's work was still pending replayed with garbage gradients; traced to memset / memcpy graph nodes, DESIGN.md section 3 -- mapping_challenge_amd.unet_models._quiesce)"""
    yield
    if 'gpu' in request.keywords and 'no_quiet' not in request.keywords:
        import gc
        import torch
        if torch.cuda.is_available():
            gc.collect()
            torch.cuda.synchronize()
