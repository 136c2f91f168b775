"""pytest configuration: the `gpu` marker, import paths and shared fixtures.

`-m "not gpu"`: oracle known-answer tests, golden regression, host logic, C-ABI
symbol check (no compute).  `-m gpu`: parity of the HIP path against the oracle,
always through the C ABI (spvo/capi.py is a ctypes shim, nothing more).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """PyTorch ships its own HIP runtime; libspvo.so links the system one.  Both work in one process when torch
    initialises the device FIRST (the order bench.py uses); the reverse order leaves torch without a device
    ("No HIP GPUs are available").  Some GPU tests hand torch device buffers to the library, so fix the order here."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:   # no torch / no GPU: the CPU suite does not need it
        pass
    # a fresh checkout has no built libraries (they are git-ignored): build them once (hipcc cross-compiles without a GPU)
    pkg = os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd")
    if not (os.path.exists(os.path.join(pkg, "libspvo.so")) and os.path.exists(os.path.join(pkg, "libspvo_host.so"))):
        import shutil
        import subprocess
        if shutil.which("hipcc") and shutil.which("make"):
            subprocess.run(["make", "-C", pkg, "all"], check=False)


@pytest.fixture
def tuning():
    """Diagnostic switches of the library for one test: tuning(name=value, ...) sets them (spvo_set_tuning), everything is forgotten
    again when the test ends.  The library reads no environment variable for these."""
    from spvo import capi

    def set_(**kw):
        capi.clear_tuning()
        for k, v in kw.items():
            capi.set_tuning(k, v)
    yield set_
    capi.clear_tuning()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def sample_images():
    from PIL import Image
    d = os.path.join(GOLDEN, "images")
    return [np.asarray(Image.open(os.path.join(d, f))) for f in sorted(os.listdir(d)) if f.endswith(".png")]


@pytest.fixture(scope="session")
def vgg_plan():
    from spvo import weights
    return weights.vgg_plan(seed=0)


@pytest.fixture(scope="session")
def vgg_weights_path(tmp_path_factory, vgg_plan):
    from spvo import weights
    p = str(tmp_path_factory.mktemp("w") / "superpoint_pretrained_2_360_1176_FP32.spvw")
    weights.save(vgg_plan, p)
    return p


@pytest.fixture(scope="session")
def squeeze_weights_path():
    return os.path.join(GOLDEN, "sp_squeeze.spvw")


@pytest.fixture(scope="session")
def squeeze_plan(squeeze_weights_path):
    from spvo import weights
    return weights.load(squeeze_weights_path)


@pytest.fixture(scope="session")
def kitti_P():
    from spvo import synth
    return synth.projection_matrices()


def make_ctx(weights_path=None, **kw):
    from spvo import capi
    ctx = capi.Context(**kw)
    if weights_path:
        ctx.load_weights(weights_path)
    return ctx


@pytest.fixture(scope="module")
def ctx_vgg(vgg_weights_path):
    c = make_ctx(vgg_weights_path)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_squeeze(squeeze_weights_path):
    c = make_ctx(squeeze_weights_path)
    yield c
    c.close()
