// mmx_device_d.hpp -- double-precision vector / quaternion / loss primitives shared by the double instantiation (mmx_f64.hip)
// and the mixed-precision one-launch solve (mmx_fused.hip, kMix): Eigen's operation order, like their float twins in
// mmx_device.hpp.
#pragma once

#include "mmx_device.hpp"

namespace mmx {

struct D3 {
  double x, y, z;
};
struct DQ {
  double x, y, z, w;
};
__device__ __forceinline__ D3 operator+(D3 a, D3 b) {
  return D3{a.x + b.x, a.y + b.y, a.z + b.z};
}
__device__ __forceinline__ D3 operator-(D3 a, D3 b) {
  return D3{a.x - b.x, a.y - b.y, a.z - b.z};
}
__device__ __forceinline__ D3 operator*(double s, D3 a) {
  return D3{s * a.x, s * a.y, s * a.z};
}
__device__ __forceinline__ D3 dcross(D3 a, D3 b) {
  return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ double ddot(D3 a, D3 b) {
  return a.x * b.x + a.y * b.y + a.z * b.z;
}
__device__ __forceinline__ DQ dqmul(DQ a, DQ b) { // Eigen quaternion product
  return DQ{
      a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
      a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
      a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x,
      a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
__device__ __forceinline__ D3 dqrot(DQ q, D3 v) { // Eigen _transformVector
  const D3 qv{q.x, q.y, q.z};
  D3 uv = dcross(qv, v);
  uv = uv + uv;
  return v + q.w * uv + dcross(qv, uv);
}
__device__ __forceinline__ D3 dqmatCol(DQ q, int c) { // column c of Eigen toRotationMatrix
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  if (c == 0) {
    return D3{1.0 - (tyy + tzz), txy + twz, txz - twy};
  }
  if (c == 1) {
    return D3{txy - twz, 1.0 - (txx + tzz), tyz + twx};
  }
  return D3{txz + twy, tyz - twx, 1.0 - (txx + tyy)};
}
__device__ __forceinline__ DQ dqnormalized(DQ q) {
  const double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  if (n2 > 0.0) {
    const double n = sqrt(n2);
    return DQ{q.x / n, q.y / n, q.z / n, q.w / n};
  }
  return q;
}
__device__ __forceinline__ double dlossValue(const LossDev& l, double s) { // generalized_loss.cpp:104-140
  const double ic = 1.0 / (double(l.c) * double(l.c)), q = s * ic; // (GeneralizedLossT<double>: 1 / c^2 in double)
  switch (l.type) {
    case 0:
      return q;
    case 1:
      return sqrt(q + 1.0) - 1.0;
    case 2:
      return log(0.5 * q + 1.0);
    case 3:
      return 1.0 - exp(-0.5 * q);
    default: {
      const double a = double(l.alpha);
      return (pow(q / fabs(a - 2.0) + 1.0, 0.5 * a) - 1.0) * fabs(a - 2.0) / a;
    }
  }
}
__device__ __forceinline__ double dlossDeriv(const LossDev& l, double s) {
  const double ic = 1.0 / (double(l.c) * double(l.c)), q = s * ic; // (GeneralizedLossT<double>: 1 / c^2 in double)
  switch (l.type) {
    case 0:
      return ic;
    case 1:
      return 0.5 * ic / sqrt(q + 1.0);
    case 2:
      return ic / (ic * s + 2.0);
    case 3:
      return 0.5 * ic * exp(-0.5 * q);
    default: {
      const double a = double(l.alpha);
      return 0.5 * ic * pow(q / fabs(a - 2.0) + 1.0, 0.5 * a - 1.0);
    }
  }
}

} // namespace mmx
