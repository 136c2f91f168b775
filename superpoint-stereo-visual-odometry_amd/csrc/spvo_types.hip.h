// spvo_types.hip.h -- the plain structs that cross translation units: kernel-argument records the context keeps between
// calls.  The kernel headers include this file and define no such type themselves, so that a translation unit that only
// needs the context (csrc/spvo_internal.hip.h) does not pull in -- and re-define -- another unit's kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spvo {

// ---- K8-K10 (post.hip.h): the NMS state map is padded (NMS_PAD rows/columns of ST_NONE on every side, row pitch a multiple
// of 4) so that a candidate's whole window is read with aligned 32-bit loads and no clipping.
constexpr int NMS_PAD = 8;           // >= largest supported dist_thresh
constexpr int NMS_MAX_LAUNCH = 16;   // round launches per host batch
constexpr int NMS_COUNTER_INTS = 8 + NMS_MAX_LAUNCH;

__host__ __device__ inline int nms_state_pitch(int W) { return ((W + 2 * NMS_PAD + 3) / 4) * 4; }

struct NmsBuffers {   // per image
  uint8_t *state;     // [(H + 2*NMS_PAD)][pitch]
  int *cand;          // [H*W] row-major pixel index of each candidate
  int *counters;      // [0] n_cand, [1] n_survivors, [2] n_out, [3] overflow, [8 + l] undecided after launch l
  unsigned long long *surv_key;  // [surv_cap]
  int *rank;          // [surv_cap], zero between uses
  int *out_xy;        // [max_kp][2]
};
struct NmsPair { NmsBuffers b[2]; };   // blockIdx.y / blockIdx.z selects the image

// ---- K15 (odometry.hip.h)
struct RansacWork {      // device scratch
  int *counts;           // [iterations]  (-1 = invalid hypothesis)
  double *poses;         // [iterations][7]  q(xyzw), t
  double *result;        // [8]: rvec(3), tvec(3), ok, n_inliers
  int *inliers;          // [n]
};

// ---- K16 (odometry.hip.h)
struct ObsDev {   // mirrors spvo_obs (include/spvo.h)
  float X[3];
  float uv[2];
  int32_t cam;
  int32_t inverse;
};

struct RefineOut {   // device, doubles: q(4) t(3) iterations converged usable initial_cost final_cost
  double v[12];
};

// ---- ORB (orb.hip.h)
struct OrbKeypoint { float x, y, angle, response; int32_t octave; };   // mirrors spvo_orb_keypoint (include/spvo.h)

}  // namespace spvo
