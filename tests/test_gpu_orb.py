"""GPU parity of the classic front end's ORB (csrc/orb.hip.h, spvo_orb_detect) against the compiled CPU restatement
(oracle/cpu/orb_cpu.inc through oracle/cpu_backend.py): integer stages and descriptors bit for bit, through the C ABI."""
import os

import numpy as np
import pytest

import oracle  # noqa: F401
from oracle import cpu_backend
from tests.conftest import make_ctx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cpu():
    c = cpu_backend.CpuBackend(net_height=64, net_width=96)
    yield c
    c.close()


def test_orb_tables_are_the_restatement_s(cpu):
    ctx = make_ctx()
    gp, gt = ctx.orb_tables()
    cp, ct = cpu.orb_tables()
    ctx.close()
    assert np.array_equal(gp, cp) and np.array_equal(gt, ct)
    assert np.abs(gp).max() <= 13 and np.all(gp == np.round(gp)) and abs(float(gt.sum()) - 1.0) < 1e-6


@pytest.mark.parametrize("case", ["kitti", "kitti_view", "small", "noise", "flat"])
def test_orb_keypoints_and_descriptors_are_bit_exact(cpu, sample_images, case):
    """Pyramid, FAST score, suppression, per-level selection and order, direction, steered BRIEF: every keypoint field that is
    integer-valued and all 256 descriptor bits equal the CPU restatement's; the reported angle agrees to 1e-5 rad."""
    rng = np.random.RandomState(1)
    if case == "kitti":
        img = sample_images[0]
    elif case == "kitti_view":
        img = sample_images[1][3:370, 5:1200]                         # a strided view: rows are not contiguous
    elif case == "small":
        img = sample_images[2][100:260, 300:620].copy()               # the upper pyramid levels fall below the border size
    elif case == "noise":
        img = rng.randint(0, 256, (200, 320)).astype(np.uint8)        # corners everywhere: ties in the response, the quota binds on every level
    else:
        img = np.full((120, 160), 77, np.uint8)                       # no corner at all
    ctx = make_ctx()
    g = ctx.orb(img)
    ctx.close()
    r = cpu.orb(img)
    assert len(g["xy"]) == len(r["xy"])
    if case == "flat":
        assert len(g["xy"]) == 0
        return
    assert len(g["xy"]) > (100 if case == "small" else 500)
    assert np.array_equal(g["octave"], r["octave"]) and np.array_equal(g["response"], r["response"])
    assert np.array_equal(g["xy"], r["xy"])
    assert np.array_equal(g["desc"], r["desc"])
    d = np.abs(g["angle"] - r["angle"])
    assert np.minimum(d, 2 * np.pi - d).max() <= 1e-5


def test_orb_buffers_follow_the_image_shape(cpu):
    """One context, images of different aspect ratios one after the other: a later image with FEWER pixels but longer rows needs
    longer resize tables and other pyramid offsets than the first one (the buffers are sized by what each image needs, not by
    rows x cols).  Every image still equals the CPU restatement bit for bit."""
    rng = np.random.RandomState(7)
    ctx = make_ctx()
    for shape in ((400, 400), (100, 1500), (700, 120), (400, 400)):
        img = rng.randint(0, 256, shape).astype(np.uint8)
        g, r = ctx.orb(img), cpu.orb(img)
        assert len(g["xy"]) == len(r["xy"]) and len(g["xy"]) > 100, shape
        assert np.array_equal(g["xy"], r["xy"]) and np.array_equal(g["octave"], r["octave"]) and np.array_equal(g["desc"], r["desc"]), shape
    ctx.close()
