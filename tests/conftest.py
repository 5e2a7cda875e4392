import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import pytest  # noqa: E402


@pytest.fixture(autouse=True)
def _reload_knobs_after_test():
    """The library reads its CGH_* environment knobs once (cgh_init) or on cgh_config_reload(); tests that set one call
    gpu_common.setknob, and once monkeypatch has restored the environment the knobs are read again here -- by
    cgh_config_reset(), which also forgets what a test set through the cgh_config_* setters (those outlive a reload)."""
    yield
    try:
        import conjugate_gradient_amd as cga
        if cga.binding._inited is not None:
            cga.lib().cgh_config_reset()
    except Exception:
        pass
