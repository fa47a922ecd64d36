import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def synthetic_sds():
    from tts_amd import spec
    return spec.synthetic_state_dicts(0)


@pytest.fixture(scope="session")
def oracle_model(synthetic_sds):
    import warnings
    warnings.filterwarnings("ignore")
    from tts_amd import spec
    from oracle import kokoro_ref
    return kokoro_ref.build(spec.DEFAULT_CONFIG, synthetic_sds)


@pytest.fixture(scope="session")
def engine(synthetic_sds):
    from tts_amd.engine import Engine
    e = Engine(synthetic_sds)
    yield e
    e.close()
