// Small device helpers shared by the SMPL sub-mesh kernels (smpl.hip, smpl_tile.hip). Internal, gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

namespace empose {

struct Rod {
  float ux, uy, uz, ang, dx, dy, dz, s, c;   // (ux, uy, uz) = ang * d(ang)/d(r)
};

// Axis-angle -> rotation, R = I + sin(a) K + (1 - cos a) K^2 with K = hat(r / a).  The un-vendored BodyModel's guard of
// the angle at r = 0 is not pinned, so both published conventions are kept (EMPOSE_RODRIGUES_*):
//   smplx  a = ||r + 1e-8||                 (smplx lbs.batch_rodrigues)
//   so3    a = sqrt(max(||r||^2, 1e-4))     (reference helpers/so3.py:116-121; below the clamp a is constant)
__device__ __forceinline__ void rodrigues(float rx, float ry, float rz, int conv, Rod& q, float (&R)[9]) {
  if (conv == 0) {
    q.ux = rx + 1e-8f; q.uy = ry + 1e-8f; q.uz = rz + 1e-8f;
    q.ang = sqrtf(q.ux * q.ux + q.uy * q.uy + q.uz * q.uz);
  } else {
    const float n2 = rx * rx + ry * ry + rz * rz;
    const bool clamped = n2 < 1e-4f;
    q.ux = clamped ? 0.f : rx; q.uy = clamped ? 0.f : ry; q.uz = clamped ? 0.f : rz;
    q.ang = sqrtf(fmaxf(n2, 1e-4f));
  }
  q.dx = rx / q.ang; q.dy = ry / q.ang; q.dz = rz / q.ang;
  sincosf(q.ang, &q.s, &q.c);
  const float oc = 1.f - q.c;
  // K = [[0,-dz,dy],[dz,0,-dx],[-dy,dx,0]];  R = I + s K + (1-c) K K
  R[0] = 1.f + oc * (-q.dz * q.dz - q.dy * q.dy);
  R[1] = -q.s * q.dz + oc * (q.dx * q.dy);
  R[2] = q.s * q.dy + oc * (q.dx * q.dz);
  R[3] = q.s * q.dz + oc * (q.dx * q.dy);
  R[4] = 1.f + oc * (-q.dz * q.dz - q.dx * q.dx);
  R[5] = -q.s * q.dx + oc * (q.dy * q.dz);
  R[6] = -q.s * q.dy + oc * (q.dx * q.dz);
  R[7] = q.s * q.dx + oc * (q.dy * q.dz);
  R[8] = 1.f + oc * (-q.dy * q.dy - q.dx * q.dx);
}


// sin and cos of a non-negative angle of ordinary size (axis-angle magnitudes: a few radians): Cody-Waite reduction by
// pi/2 in three pieces and the single-precision minimax polynomials on [-pi/4, pi/4] (errors below 1 ulp there).  The
// library's sincosf carries its large-argument reduction along (~150 dependent instructions); a wave of the
// frame-per-lane kernel has nothing else to issue while it waits for them.  Valid for |x| < ~1e4.
__device__ __forceinline__ void sincos_small(float x, float* s, float* c) {
  const float kf = rintf(x * 0.63661977236758134f);
  const int k = (int)kf;
  float r = fmaf(-kf, 1.5707962513e+00f, x);
  r = fmaf(-kf, 7.5497894159e-08f, r);
  r = fmaf(-kf, 5.3903029534e-15f, r);
  const float z = r * r;
  const float ps = r + r * z * fmaf(z, fmaf(z, fmaf(z, 2.718311493989822e-6f, -1.9839334836096632e-4f), 8.3333293858894632e-3f), -1.6666654611e-1f);
  const float pc = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f), fmaf(-0.5f, z, 1.f));
  const float sv = (k & 1) ? pc : ps, cv = (k & 1) ? ps : pc;
  *s = (k & 2) ? -sv : sv;
  *c = ((k + 1) & 2) ? -cv : cv;
}

// rodrigues() with sincos_small (the frame-per-lane kernels; same conventions, same outputs to ~1 ulp)
__device__ __forceinline__ void rodrigues_fast(float rx, float ry, float rz, int conv, Rod& q, float (&R)[9]) {
  if (conv == 0) {
    q.ux = rx + 1e-8f; q.uy = ry + 1e-8f; q.uz = rz + 1e-8f;
    q.ang = sqrtf(q.ux * q.ux + q.uy * q.uy + q.uz * q.uz);
  } else {
    const float n2 = rx * rx + ry * ry + rz * rz;
    const bool clamped = n2 < 1e-4f;
    q.ux = clamped ? 0.f : rx; q.uy = clamped ? 0.f : ry; q.uz = clamped ? 0.f : rz;
    q.ang = sqrtf(fmaxf(n2, 1e-4f));
  }
  const float inv = 1.f / q.ang;
  q.dx = rx * inv; q.dy = ry * inv; q.dz = rz * inv;
  sincos_small(q.ang, &q.s, &q.c);
  const float oc = 1.f - q.c;
  R[0] = 1.f + oc * (-q.dz * q.dz - q.dy * q.dy);
  R[1] = -q.s * q.dz + oc * (q.dx * q.dy);
  R[2] = q.s * q.dy + oc * (q.dx * q.dz);
  R[3] = q.s * q.dz + oc * (q.dx * q.dy);
  R[4] = 1.f + oc * (-q.dz * q.dz - q.dx * q.dx);
  R[5] = -q.s * q.dx + oc * (q.dy * q.dz);
  R[6] = -q.s * q.dy + oc * (q.dx * q.dz);
  R[7] = q.s * q.dx + oc * (q.dy * q.dz);
  R[8] = 1.f + oc * (-q.dy * q.dy - q.dx * q.dx);
}

__device__ __forceinline__ void cross3(const float* a, const float* b, float* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ float norm3(const float* a) { return sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
// y = x/|x|  ->  dx = (dy - y (y.dy)) / |x|
__device__ __forceinline__ void unit_bwd(const float* dy, const float* y, float inv_n, float* dx) {
  const float d = dy[0] * y[0] + dy[1] * y[1] + dy[2] * y[2];
  dx[0] = (dy[0] - y[0] * d) * inv_n;
  dx[1] = (dy[1] - y[1] * d) * inv_n;
  dx[2] = (dy[2] - y[2] * d) * inv_n;
}

}  // namespace empose
