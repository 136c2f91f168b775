"""Oracle for the detector/descriptor front end.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  numpy restatement, line by
line, of

  * FeatureFrontEnd::preprocessImageImpl            base.cpp:68-121
  * SuperPointFeatureFrontEnd::preprocessImage      nn.cpp:139-161
  * postprocessDetectionAndDescription              nn.cpp:264-364
  * processOneHeatmap                               nn.cpp:188-262
  * bilinearInterpolationDesc                       nn.cpp:366-431

("nn.cpp" = src/odml_visual_odometry/src/feature_detection_neural_network.cpp,
 "base.cpp" = src/odml_visual_odometry/src/feature_detection_base.cpp.)

Third-party semantics restated from their published behaviour (OpenCV 4.5.4 is
not installed here, so these are "parity unpinned"):
  cv::resize(INTER_LINEAR) on CV_8UC1: pixel centres src = (dst+0.5)*scale-0.5,
  11-bit fixed-point coefficients (INTER_RESIZE_COEF_BITS = 11), horizontal pass
  to int, vertical pass ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.
"""
from __future__ import annotations

import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


# ----------------------------------------------------------------------------
# cv::resize, 8-bit, INTER_LINEAR
# ----------------------------------------------------------------------------
def _linear_coeffs(dst: int, src: int):
    """Index + 2 fixed-point taps per destination sample (OpenCV resize.cpp)."""
    scale = float(src) / float(dst)           # double, = 1 / inv_scale
    idx = np.zeros(dst, np.int32)
    a0 = np.zeros(dst, np.int32)
    a1 = np.zeros(dst, np.int32)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if s < 0:
            s, f = 0, np.float32(0.0)
        if s >= src - 1:
            s, f = src - 1, np.float32(0.0)
        idx[d] = s
        # saturate_cast<short>(float) rounds to nearest-even (cvRound)
        a0[d] = int(np.rint(np.float32(np.float32(1.0) - f) * np.float32(COEF_SCALE)))
        a1[d] = int(np.rint(f * np.float32(COEF_SCALE)))
    return idx, a0, a1


def resize_linear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    assert img.dtype == np.uint8 and img.ndim == 2
    h, w = img.shape
    if (h, w) == (out_h, out_w):
        return img.copy()                     # cv::resize copies when sizes match
    xi, xa0, xa1 = _linear_coeffs(out_w, w)
    yi, yb0, yb1 = _linear_coeffs(out_h, h)
    src = img.astype(np.int32)
    xi1 = np.minimum(xi + 1, w - 1)
    # horizontal pass: int rows, scaled by 2^11
    hor = src[:, xi] * xa0[None, :] + src[:, xi1] * xa1[None, :]
    yi1 = np.minimum(yi + 1, h - 1)
    s0 = hor[yi] >> 4
    s1 = hor[yi1] >> 4
    out = (((yb0[:, None] * s0) >> 16) + ((yb1[:, None] * s1) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


# ----------------------------------------------------------------------------
# preprocessImageImpl  (base.cpp:68-121)
# ----------------------------------------------------------------------------
def crop_geometry(rows: int, cols: int, net_h: int, net_w: int):
    """Returns (row_off, col_off, crop_rows, crop_cols, scale_f32) exactly as base.cpp:75-119."""
    real = np.float32(cols) / np.float32(rows)
    expected = np.float32(net_w) / np.float32(net_h)
    crop_rows, crop_cols, row_off, col_off = rows, cols, 0, 0
    if expected > real:
        crop_rows = int(np.float32(cols) / expected)          # base.cpp:86 (int / float -> trunc)
        row_off = (rows - crop_rows) // 2
    elif expected < real:
        crop_cols = int(np.float32(rows) * expected)          # base.cpp:102
        col_off = (cols - crop_cols) // 2
    scale = np.float32(net_w) / np.float32(crop_cols)         # base.cpp:118-119
    return row_off, col_off, crop_rows, crop_cols, scale


def preprocess(img: np.ndarray, P: np.ndarray, net_h: int, net_w: int, bug_compat: bool = True):
    """img: u8 [rows, cols]; P: f64 [3,4].  Returns (resized u8 [net_h, net_w], P')."""
    rows, cols = img.shape
    row_off, col_off, crop_rows, crop_cols, scale = crop_geometry(rows, cols, net_h, net_w)
    P = np.array(P, dtype=np.float64).reshape(3, 4).copy()
    if bug_compat:
        # base.cpp:95,111: `projection_matrix.at<float>(r, 2) -= offset` on a CV_64F
        # matrix (node.cpp:91) writes the low half of P[r][1]; cx, cy are untouched.
        Pf = P.view(np.float32)               # little endian: [3, 8]
        if crop_rows != rows:
            Pf[1, 2] -= np.float32(row_off)
        elif crop_cols != cols:
            Pf[0, 2] -= np.float32(col_off)
    else:
        if crop_rows != rows:
            P[1, 2] -= float(np.float32(row_off))
        elif crop_cols != cols:
            P[0, 2] -= float(np.float32(col_off))
    cropped = img[row_off:row_off + crop_rows, col_off:col_off + crop_cols]
    resized = resize_linear_u8(np.ascontiguousarray(cropped), net_h, net_w)
    P[0:2, :] *= float(scale)                 # base.cpp:120
    return resized, P


def to_network_input(resized: np.ndarray) -> np.ndarray:
    """nn.cpp:159: img.convertTo(CV_32FC1, 1/255)  (u8 * f32 scale, computed in f32... cvt uses f32 math)."""
    return resized.astype(np.float32) * np.float32(1.0 / 255.0)


# ----------------------------------------------------------------------------
# detector half of postprocessDetectionAndDescription  (nn.cpp:266-326)
# ----------------------------------------------------------------------------
def heatmap(det: np.ndarray) -> np.ndarray:
    """det: f32 [65, Hc, Wc] -> heat f32 [8*Hc, 8*Wc]."""
    assert det.dtype == np.float32 and det.shape[0] == 65
    _, hc, wc = det.shape
    dense = np.exp(det)                                       # nn.cpp:271, no max-subtraction
    s = dense.sum(axis=0, dtype=np.float32) + np.float32(0.00001)   # nn.cpp:274-283
    dense = dense / s[None]
    nodust = dense[:64]                                       # nn.cpp:289-295
    # [64,Hc,Wc] -> [Hc,Wc,8,8] -> [Hc,8,Wc,8] -> [H,W]        nn.cpp:298-326
    hm = nodust.transpose(1, 2, 0).reshape(hc, wc, 8, 8).transpose(0, 2, 1, 3)
    return np.ascontiguousarray(hm.reshape(hc * 8, wc * 8))


# ----------------------------------------------------------------------------
# processOneHeatmap  (nn.cpp:188-262)
# ----------------------------------------------------------------------------
def rank_candidates(heat: np.ndarray, conf_thresh: float):
    """Candidates above threshold in the reference's visiting order.

    nn.cpp:202-213 walks an Eigen column-major SparseMatrix (outer = column,
    inner = row ascending) and nn.cpp:214-217 std::sort()s by confidence
    descending.  std::sort is unstable, so ties have no defined order in the
    reference; the total order pinned here (and in the HIP path) is
    (confidence desc, column-major index x*H + y asc).
    """
    H, W = heat.shape
    ys, xs = np.nonzero(heat > np.float32(conf_thresh))       # strict '>'  nn.cpp:203
    conf = heat[ys, xs]
    cm = xs.astype(np.int64) * H + ys
    order = np.lexsort((cm, -conf.astype(np.float64)))
    return xs[order].astype(np.int32), ys[order].astype(np.int32), conf[order]


def nms(heat: np.ndarray, conf_thresh: float = 0.015, dist_thresh: int = 4, border: int = 4,
        max_keypoints: int = 1000) -> np.ndarray:
    """Returns int32 [n, 2] (x, y) in emission order.  nn.cpp:219-258."""
    H, W = heat.shape
    xs, ys, _ = rank_candidates(heat, conf_thresh)
    suppressed = np.zeros((H, W), np.bool_)
    out = []
    for x, y in zip(xs.tolist(), ys.tolist()):
        if not suppressed[y, x]:
            if border <= y and y + border < H and border <= x and x + border < W:
                out.append((x, y))
            suppressed[max(0, y - dist_thresh):y + dist_thresh + 1,
                       max(0, x - dist_thresh):x + dist_thresh + 1] = True
        if len(out) >= max_keypoints:                         # nn.cpp:256-257
            break
    return np.asarray(out, np.int32).reshape(-1, 2)


# ----------------------------------------------------------------------------
# bilinearInterpolationDesc  (nn.cpp:366-431)
# ----------------------------------------------------------------------------
def sample_descriptors(desc_nchw: np.ndarray, xy: np.ndarray, H: int, W: int) -> np.ndarray:
    """desc_nchw f32 [256, Hc, Wc]; xy int [n,2] -> f32 [n,256], each row unit norm."""
    C, hc, wc = desc_nchw.shape
    nhwc = np.ascontiguousarray(desc_nchw.transpose(1, 2, 0))  # nn.cpp:339-342
    f32 = np.float32
    out = np.zeros((len(xy), C), f32)
    for i, (col, row) in enumerate(np.asarray(xy).tolist()):
        row8 = f32(row) / f32(H - 1) * f32(hc - 1)            # nn.cpp:377-382 (align_corners=True)
        col8 = f32(col) / f32(W - 1) * f32(wc - 1)
        r0 = int(np.floor(row8))
        c0 = int(np.floor(col8))
        rr = f32(1.0) - (row8 - f32(r0))                      # nn.cpp:391-392
        cr = f32(1.0) - (col8 - f32(c0))
        r1 = min(r0 + 1, hc - 1)                              # weight is 0 whenever this clamps
        c1 = min(c0 + 1, wc - 1)
        tl, tr, bl, br = nhwc[r0, c0], nhwc[r0, c1], nhwc[r1, c0], nhwc[r1, c1]
        v = (tl * rr * cr + tr * rr * (f32(1.0) - cr) + bl * (f32(1.0) - rr) * cr
             + br * (f32(1.0) - rr) * (f32(1.0) - cr))        # nn.cpp:423-427
        out[i] = v / np.sqrt(np.sum(v * v, dtype=f32))        # nn.cpp:428 normalize()
    return out


def detect(plan, img_u8: np.ndarray, P: np.ndarray, net_h: int, net_w: int, conf_thresh=0.015,
           dist_thresh=4, border=4, max_keypoints=1000, bug_compat=True):
    """One image through rows B-H.  Returns dict with every intermediate."""
    from . import net
    resized, P2 = preprocess(img_u8, P, net_h, net_w, bug_compat)
    x = to_network_input(resized)[None, None]
    det, desc = net.forward(plan, x)
    heat = heatmap(det[0])
    xy = nms(heat, conf_thresh, dist_thresh, border, max_keypoints)
    d = sample_descriptors(desc[0], xy, net_h, net_w)
    return dict(resized=resized, P=P2, det=det[0], desc=desc[0], heat=heat, xy=xy, descriptors=d)
