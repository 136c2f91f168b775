"""Freezes the DIRECT evaluation of the reference's ONNX graphs (oracle/onnx_direct.py: own protobuf reader, literal
operator semantics, nothing of the product's plan packer involved) on a seeded 64x96 input.  Runs only where
/root/reference exists; the fixtures (inputs + outputs, ~0.4 MB each) travel, the ONNX files do not.

    python tests/golden/make_onnx_direct_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import onnx_direct  # noqa: E402

MODELS = "/root/reference/src/odml_visual_odometry/models"


def main():
    for name in ("sp_squeeze", "sp_mbv1", "sp_mbv2"):
        x = np.random.RandomState(7).rand(1, 1, 64, 96).astype(np.float32)
        out = onnx_direct.run(os.path.join(MODELS, name + "_b1.onnx"), x)
        det, desc = out["output_det"], out["output_desc"]
        assert det.shape == (1, 65, 8, 12) and desc.shape == (1, 256, 8, 12)
        np.savez_compressed(os.path.join(HERE, f"onnx_direct_{name}_64x96.npz"), x=x, det=det, desc=desc)
        print(name, "det range", float(det.min()), float(det.max()))


if __name__ == "__main__":
    main()
