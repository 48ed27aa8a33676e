import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def rip_lib():
    """The built C-ABI library; building is part of the test environment, never a CPU fallback."""
    from raw_image_pipeline_amd import build as B
    B.build()
    from raw_image_pipeline_amd import load_library
    return load_library()


@pytest.fixture()
def host_pipe(rip_lib):
    """Parameter-only handle (RIP_DEVICE_NONE): host logic without a GPU."""
    from raw_image_pipeline_amd import RawImagePipeline
    return RawImagePipeline(False, "", "", "", device=-1)


@pytest.fixture()
def gpu_pipe(rip_lib):
    import torch
    assert torch.cuda.is_available(), "gpu-marked test without a GPU"
    from raw_image_pipeline_amd import RawImagePipeline
    p = RawImagePipeline(False, "", "", "", device=0)
    # a neutral starting point: every optional stage off
    p.set_white_balance(False)
    p.set_undistortion(False)
    return p
