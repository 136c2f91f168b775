// mock of <tf2/LinearMath/Transform.h> (+ Quaternion.h, Vector3.h): declarations only (see README.md)
#pragma once
typedef double tf2Scalar;
namespace tf2 {
class Vector3 {
public:
  Vector3();
  Vector3(const tf2Scalar &x, const tf2Scalar &y, const tf2Scalar &z);
  const tf2Scalar &x() const;
  const tf2Scalar &y() const;
  const tf2Scalar &z() const;
  tf2Scalar length() const;
};
class Quaternion {
public:
  Quaternion();
  Quaternion(const tf2Scalar &x, const tf2Scalar &y, const tf2Scalar &z, const tf2Scalar &w);
  const tf2Scalar &x() const;
  const tf2Scalar &y() const;
  const tf2Scalar &z() const;
  const tf2Scalar &w() const;
};
class Transform {
public:
  Transform();
  Transform(const Quaternion &q, const Vector3 &c);
  void setRotation(const Quaternion &q);
  void setOrigin(const Vector3 &origin);
  Quaternion getRotation() const;
  const Vector3 &getOrigin() const;
  Transform inverse() const;
  Transform operator*(const Transform &t) const;
  void setIdentity();
};
}  // namespace tf2
