"""CPU checks of the drop-in boundary: the C-ABI library loads and exports exactly
the symbols include/spvo.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from spvo import capi, weights
from tests.conftest import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "spvo.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spvo_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(capi.LIB_PATH), "build first: python __graft_entry__.py"
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name


def test_struct_layouts_match_header():
    # spvo_obs must stay 28 bytes (3+2 floats, 2 int32): the kernels read it as-is
    assert capi.OBS_DTYPE.itemsize == 28
    assert ctypes.sizeof(capi.Config) == 9 * 4


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(capi.SpvoError) as e:
        capi.Context()
    assert "no CPU path" in str(e.value) or "HIP" in str(e.value)


def test_bad_config_rejected_before_touching_the_device():
    lib = capi.load()
    cfg = capi.Config()
    lib.spvo_default_config(ctypes.byref(cfg))
    assert (cfg.net_height, cfg.net_width, cfg.max_keypoints, cfg.dist_thresh) == (360, 1176, 1000, 4)
    cfg.net_width = 1241                                 # not a multiple of 8 (feature_detection.hpp:296)
    h = ctypes.c_void_p()
    assert lib.spvo_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"multiples of 8" in lib.spvo_last_error(None)
    cfg.net_width, cfg.max_batch = 1176, 3               # nn.cpp:489-491 "Wrong batch size"
    assert lib.spvo_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"Wrong batch size" in lib.spvo_last_error(None)


def test_weight_file_round_trip(tmp_path, vgg_plan):
    p = str(tmp_path / "w.spvw")
    weights.save(vgg_plan, p)
    q = weights.load(p)
    assert q.n_params() == vgg_plan.n_params() == 1300865
    assert q.tensors == vgg_plan.tensors and len(q.ops) == len(vgg_plan.ops)
    for a, b in zip(q.ops, vgg_plan.ops):
        assert (a.type, a.inp, a.out, a.out_c_off, a.cin, a.cout, a.ksize, a.flags) == \
               (b.type, b.inp, b.out, b.out_c_off, b.cin, b.cout, b.ksize, b.flags)
        if b.weight is not None:
            assert np.array_equal(a.weight, b.weight) and np.array_equal(a.bias, b.bias)
    assert weights.engine_name("superpoint_pretrained", 2, 360, 1176, "FP32") == "superpoint_pretrained_2_360_1176_FP32.spvw"


def test_squeeze_fixture_is_the_reference_graph(squeeze_plan):
    # 844 353 parameters = the reference's sp_squeeze ONNX
    assert squeeze_plan.n_params() == 844353
    assert squeeze_plan.tensors[squeeze_plan.det_tensor] == (65, 3)
    assert squeeze_plan.tensors[squeeze_plan.desc_tensor] == (256, 3)


@pytest.mark.parametrize("name,n_params,n_dw", [("mbv1", 752779, 7), ("mbv2", 945035, 7)])
def test_mobilenet_fixtures_are_the_reference_graphs(name, n_params, n_dw):
    """sp_mbv1 / sp_mbv2 (config 3's graphs): every Relu / BatchNormalization / Add / MaxPool of the ONNX
    graph is folded into the producing convolution, so the plan holds convolutions and the L2 tail only."""
    from spvo import weights
    plan = weights.load(os.path.join(GOLDEN, f"sp_{name}.spvw"))
    n = sum(op.weight.size + op.bias.size + (op.bn.size - 1 if op.bn is not None else 0)
            for op in plan.ops if op.weight is not None)
    assert n == n_params
    kinds = [op.type for op in plan.ops]
    assert kinds.count(weights.OP_DWCONV) == n_dw and kinds.count(weights.OP_MAXPOOL) == 0
    assert kinds[-1] == weights.OP_L2NORM
    assert plan.tensors[plan.det_tensor] == (65, 3) and plan.tensors[plan.desc_tensor] == (256, 3)
    if name == "mbv1":
        assert all((op.flags & weights.FLAG_BN) for op in plan.ops[1:16:2])       # each pointwise layer
    else:
        res = [op for op in plan.ops if op.flags & weights.FLAG_ADD]
        assert len(res) == 6 and all(plan.tensors[op.residual][0] == op.cout for op in res)


@pytest.mark.parametrize("name", ["mbv1", "mbv2"])
def test_oracle_runs_mobilenet_graphs(name):
    """The oracle's executor against a direct numpy evaluation of one fused MobileNet epilogue."""
    import torch
    from oracle import net
    from spvo import weights
    plan = weights.load(os.path.join(GOLDEN, f"sp_{name}.spvw"))
    rng = np.random.RandomState(0)
    x = rng.rand(1, 1, 16, 24).astype(np.float32)
    det, desc, vals = net.forward(plan, x, return_all=True)
    assert det.shape == (1, 65, 2, 3) and desc.shape == (1, 256, 2, 3)
    assert np.allclose(np.linalg.norm(desc, axis=1), 1.0, atol=1e-5)
    op = next(o for o in plan.ops if o.flags & (weights.FLAG_BN | weights.FLAG_ADD) and o.cin > 1)
    xin = vals[op.inp].astype(np.float64)
    y = np.einsum("oc,bchw->bohw", op.weight[:, :, 0, 0].astype(np.float64), xin) + op.bias[None, :, None, None]
    if op.flags & weights.FLAG_RELU:
        y = np.maximum(y, 0)
    if op.flags & weights.FLAG_BN:
        c = op.cout
        g, b, m, v = (op.bn[i * c:(i + 1) * c].astype(np.float64)[None, :, None, None] for i in range(4))
        y = np.maximum((y - m) / np.sqrt(v + float(op.bn[4 * c])) * g + b, 0)
    if op.flags & weights.FLAG_ADD:
        y = np.maximum(y + vals[op.residual], 0)
    if op.flags & weights.FLAG_POOL:
        y = y.reshape(1, op.cout, y.shape[2] // 2, 2, y.shape[3] // 2, 2).max(axis=(3, 5))
    assert np.abs(vals[op.out] - y).max() < 1e-4


def test_fp16_engine_file_and_oracle(tmp_path):
    """The precision travels in the engine file (weights stay canonical fp32) and the oracle's FP16 restatement rounds
    what an FP16 engine stores: weights and intermediate activations, not the bindings."""
    import copy
    from oracle import net
    from spvo import weights
    plan = weights.vgg_plan(seed=0)
    p16 = copy.copy(plan)
    p16.precision = "FP16"
    path = str(tmp_path / weights.engine_name("superpoint_pretrained", 2, 32, 48, "FP16"))
    weights.save(p16, path)
    back = weights.load(path)
    assert back.precision == "FP16" and weights.load(os.path.join(GOLDEN, "sp_squeeze.spvw")).precision == "FP32"
    assert all(np.array_equal(a.weight, b.weight) for a, b in zip(plan.ops, back.ops) if a.weight is not None)
    x = np.random.RandomState(1).rand(1, 1, 32, 48).astype(np.float32)
    det32, desc32 = net.forward(plan, x)
    det16, desc16, vals = net.forward(back, x, return_all=True)
    mid = vals[3]                                     # an intermediate activation: exactly representable in fp16
    assert np.array_equal(mid, mid.astype(np.float16).astype(np.float32))
    assert not np.array_equal(det16, det16.astype(np.float16).astype(np.float32))     # outputs stay fp32
    assert 0 < np.abs(det16 - det32).max() < 2e-2 * np.abs(det32).max()


HOST = os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd", "host")


@pytest.mark.parametrize("opencv", [False, True])
def test_host_library_and_a_node_like_caller_compile_against_both_type_sets(opencv):
    """host/*.cpp (the FeatureFrontEnd mirror) and tests/boundary_node_caller.cpp (the call sequence of
    visual_odometry_node.cpp:150-262, 316, 330-403: both constructors, addStereoImagePair, matchDescriptors,
    visualize*, solveStereoOdometry(tf2::Transform&), clearLagecyData, preprocessImageImpl, verbose_, the public
    deques) are ONE source for two builds: the stand-in types of this image and -- with SPVO_USE_OPENCV -- OpenCV /
    tf2 shaped headers (tests/mock_ros: methods, not fields).  The second build is what a ROS workspace compiles."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    flags = ["-DSPVO_USE_OPENCV", "-I" + os.path.join(ROOT, "tests", "mock_ros")] if opencv else []
    for src in ("feature_detection.cpp", "vo_io.cpp", "harness_capi.cpp", os.path.join(ROOT, "tests", "boundary_node_caller.cpp")):
        path = src if os.path.isabs(src) else os.path.join(HOST, src)
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + HOST] + flags + [path],
                           capture_output=True, text=True)
        assert r.returncode == 0, (src, r.stderr[-2000:])


def test_classic_front_end_is_declared_and_refuses_loudly_without_a_device():
    """ClassicFeatureFrontEnd (hpp:184-235; constructed at node.cpp:353-360) is part of the interface.  Its ORB + ORB
    configuration runs on the GPU (spvo_orb_detect); on a machine without one addStereoImagePair logs and returns
    (nn.cpp:53-55 convention) and leaves the deques empty -- there is no CPU path in the product."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by tests/test_gpu_host.py")
    lib = ctypes.CDLL(os.path.join(os.path.dirname(capi.LIB_PATH), "libspvo_host.so"))
    lib.spvo_host_classic_probe.restype = ctypes.c_int
    lib.spvo_host_classic_probe.argtypes = [ctypes.c_char_p, ctypes.c_int]
    buf = ctypes.create_string_buffer(512)
    rc = lib.spvo_host_classic_probe(buf, 512)
    assert rc == 0                                        # nothing pushed
    assert b"no HIP device" in buf.value or b"no CPU path" in buf.value
