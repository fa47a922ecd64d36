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
    """Seeded random weights of the exact architecture -- or, with K82_REAL_PTH=<kokoro-v1_0.pth>, the real
    checkpoint (tools/check_real_weights.py): the same parity suite then runs on real weights."""
    from tts_amd import spec
    real = os.environ.get("K82_REAL_PTH")
    if real:
        from tts_amd import weights
        return weights.load_pth(real)                    # torch.load(weights_only=True)
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
