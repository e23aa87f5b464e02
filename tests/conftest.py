import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def map_cache():
    """Synthetic maps of thousands of keyframes take ~10 s to build: the full-size tests of several modules share them (read-only: every
    solve works on a copy)."""
    from textslam_amd import synth
    cache = {}

    def get(**kw):
        key = tuple(sorted(kw.items()))
        if key not in cache:
            text = kw.pop("n_text", 0)
            if text:
                cache[key] = synth.make_problem(n_text=text, max_targets=8, text_targets=5, frozen_frac=0.0, rot_deg=0.2, trans_m=0.01, **kw)
            else:
                cache[key] = synth.config_global(**kw)
        return cache[key]
    return get
