import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gold_dir():
    return GOLD


_scene_cache = {}


@pytest.fixture(scope="session")
def host_scene():
    """Factory: flattened scenes from the C++ host surface, cached for the session."""
    import raytracing_amd as ra

    def get(name, param=0):
        key = (name, param)
        if key not in _scene_cache:
            if name == "mrspheres":       # the model's geometry buffer is not in the reference repository: a synthetic one is written on first use
                import mrspheres_bin
                mrspheres_bin.ensure(ra.default_assets_dir())
            _scene_cache[key] = ra.HostScene(name, param)
        return _scene_cache[key]
    return get


_oracle_cache = {}


@pytest.fixture(scope="session")
def oracle_scene(host_scene):
    import oracle_lib as ol

    def get(name, param=0):
        key = (name, param)
        if key not in _oracle_cache:
            _oracle_cache[key] = ol.OracleScene(host_scene(name, param).desc_ptr)
        return _oracle_cache[key]
    return get


@pytest.fixture(scope="session")
def gpu_ctx():
    import raytracing_amd as ra
    ctx = ra.Context(0)
    yield ctx
    ctx.close()


_dev_cache = {}


@pytest.fixture(scope="session")
def device_scene(host_scene, gpu_ctx):
    import raytracing_amd as ra

    def get(name, param=0):
        key = (name, param)
        if key not in _dev_cache:
            _dev_cache[key] = ra.DeviceScene(gpu_ctx, host_scene(name, param).desc_ptr)
        return _dev_cache[key]
    yield get
    for s in _dev_cache.values():
        s.close()
    _dev_cache.clear()
