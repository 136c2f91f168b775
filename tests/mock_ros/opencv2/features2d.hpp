// mock of <opencv2/features2d.hpp>: declarations only (see README.md)
#pragma once
#include "core.hpp"
namespace cv {
class Feature2D {
public:
  virtual ~Feature2D();
  virtual void detect(InputArray image, std::vector<KeyPoint> &keypoints, InputArray mask = noArray());
  virtual void compute(InputArray image, std::vector<KeyPoint> &keypoints, OutputArray descriptors);
};
typedef Feature2D FeatureDetector;
typedef Feature2D DescriptorExtractor;
class ORB : public Feature2D {
public:
  enum ScoreType { HARRIS_SCORE = 0, FAST_SCORE = 1 };
  static Ptr<ORB> create(int nfeatures = 500, float scaleFactor = 1.2f, int nlevels = 8, int edgeThreshold = 31, int firstLevel = 0, int WTA_K = 2,
                         ORB::ScoreType scoreType = ORB::HARRIS_SCORE, int patchSize = 31, int fastThreshold = 20);
};
class BRISK : public Feature2D {
public:
  static Ptr<BRISK> create(int thresh = 30, int octaves = 3, float patternScale = 1.0f);
};
class AKAZE : public Feature2D {
public:
  static Ptr<AKAZE> create();
};
class SIFT : public Feature2D {
public:
  static Ptr<SIFT> create();
};
class FastFeatureDetector : public Feature2D {
public:
  static Ptr<FastFeatureDetector> create(int threshold = 10, bool nonmaxSuppression = true);
};
class GFTTDetector : public Feature2D {
public:
  static Ptr<GFTTDetector> create(int maxCorners = 1000, double qualityLevel = 0.01, double minDistance = 1, int blockSize = 3,
                                  bool useHarrisDetector = false, double k = 0.04);
};
class DescriptorMatcher {
public:
  virtual ~DescriptorMatcher();
  void match(InputArray queryDescriptors, InputArray trainDescriptors, std::vector<DMatch> &matches, InputArray mask = noArray()) const;
  void knnMatch(InputArray queryDescriptors, InputArray trainDescriptors, std::vector<std::vector<DMatch>> &matches, int k,
                InputArray mask = noArray(), bool compactResult = false) const;
};
class BFMatcher : public DescriptorMatcher {
public:
  static Ptr<BFMatcher> create(int normType = NORM_L2, bool crossCheck = false);
};
}  // namespace cv
