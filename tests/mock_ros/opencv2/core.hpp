// mock of <opencv2/core.hpp>: declarations only (see README.md)
#pragma once
#include <cstddef>
#include <vector>
#define CV_CN_SHIFT 3
#define CV_DEPTH_MAX (1 << CV_CN_SHIFT)
#define CV_MAT_DEPTH_MASK (CV_DEPTH_MAX - 1)
#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) (((depth) & CV_MAT_DEPTH_MASK) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)
typedef unsigned char uchar;
namespace cv {
template <typename T> class Ptr {
public:
  Ptr();
  template <typename U> Ptr(const Ptr<U> &o);
  T *operator->() const;
  T *get() const;
  explicit operator bool() const;
};
template <typename T> class Point_ {
public:
  Point_();
  Point_(T x, T y);
  T x, y;
};
typedef Point_<float> Point2f;
template <typename T> class Size_ {
public:
  Size_();
  Size_(T w, T h);
  T width, height;
};
typedef Size_<int> Size;
struct MatStep {
  MatStep();
  operator size_t() const;
  size_t *p;
};
class Mat {
public:
  Mat();
  Mat(int rows, int cols, int type);
  Mat(const Mat &m);
  ~Mat();
  Mat &operator=(const Mat &m);
  void create(int rows, int cols, int type);
  Mat clone() const;
  void release();
  bool empty() const;
  int type() const;
  int depth() const;
  int channels() const;
  size_t elemSize() const;
  Mat rowRange(int startrow, int endrow) const;
  Mat colRange(int startcol, int endcol) const;
  template <typename T> T &at(int row, int col);
  template <typename T> const T &at(int row, int col) const;
  template <typename T> T *ptr(int i0 = 0);
  template <typename T> const T *ptr(int i0 = 0) const;
  int flags, dims, rows, cols;
  uchar *data;
  MatStep step;
};
class _InputArray {
public:
  _InputArray();
  _InputArray(const Mat &m);
};
class _OutputArray : public _InputArray {
public:
  _OutputArray();
  _OutputArray(Mat &m);
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
InputArray noArray();
class KeyPoint {
public:
  KeyPoint();
  KeyPoint(Point2f pt, float size, float angle = -1, float response = 0, int octave = 0, int class_id = -1);
  Point2f pt;
  float size, angle, response;
  int octave, class_id;
};
class DMatch {
public:
  DMatch();
  DMatch(int queryIdx, int trainIdx, float distance);
  int queryIdx, trainIdx, imgIdx;
  float distance;
};
enum NormTypes { NORM_INF = 1, NORM_L1 = 2, NORM_L2 = 4, NORM_L2SQR = 5, NORM_HAMMING = 6, NORM_HAMMING2 = 7 };
}  // namespace cv
