// TEST INFRASTRUCTURE: the few more Eigen types oracle/ref_recipe/dump_fixtures.cpp touches, as stand-ins (see registration.h in this
// directory): enough for `g++ -fsyntax-only`, nothing is computed with them.
#pragma once
#include <pcl/registration/registration.h>
namespace Eigen {
template <typename T, int R, int C>
struct Matrix {
  T m[R * C];
  T* data() { return m; }
  const T* data() const { return m; }
  T& operator()(int i, int j) { return m[j * R + i]; }
  const T& operator()(int i, int j) const { return m[j * R + i]; }
  T& operator()(int i) { return m[i]; }
  const T& operator()(int i) const { return m[i]; }
  void setZero() { for (T& v : m) v = T(0); }
  static Matrix UnitX() { return Matrix(); }
  static Matrix UnitY() { return Matrix(); }
  static Matrix UnitZ() { return Matrix(); }
};
using Matrix3d = Matrix<double, 3, 3>;
using Vector3i = Matrix<int, 3, 1>;
using Vector3f = Matrix<float, 3, 1>;
struct Affine3f { Matrix4f matrix() const { return Matrix4f::Identity(); } };
template <typename T> struct AngleAxis { AngleAxis(T, const Vector3f&) {} };
template <typename T, int D> struct Translation { Translation(T, T, T) {} };
template <typename T> Affine3f operator*(const Translation<T, 3>&, const AngleAxis<T>&) { return Affine3f(); }
template <typename T> Affine3f operator*(const Affine3f&, const AngleAxis<T>&) { return Affine3f(); }
}  // namespace Eigen
