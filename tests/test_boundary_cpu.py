"""CPU checks of the drop-in boundary: the C-ABI library loads and exports exactly
the symbols include/spvo.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from spvo import capi, weights

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
