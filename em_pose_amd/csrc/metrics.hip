// Evaluation metrics on the device (reference empose/eval/metrics.py:18-66,110-162 and
// helpers/utils.py:165-199): per frame the Euclidean joint distances, the distances after a similarity Procrustes
// alignment of the prediction onto the ground truth (the reference runs a NumPy SVD per frame in a Python loop), and
// the geodesic angle between predicted and ground-truth GLOBAL joint orientations.
// One thread per frame, float64 arithmetic (the accumulated rows feed means / standard deviations on the host).
#include "kernels.h"

namespace empose {

constexpr int MJ = 22;

__device__ inline void jacobi_eig3(double B[9], double V[9]) {
  // cyclic Jacobi on a symmetric 3x3; on return B is diagonal (eigenvalues) and the columns of V are eigenvectors
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(B[1]) + fabs(B[2]) + fabs(B[5]);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = B[p * 3 + q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (B[q * 3 + q] - B[p * 3 + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {  // B <- B J
          const double bkp = B[k * 3 + p], bkq = B[k * 3 + q];
          B[k * 3 + p] = c * bkp - s * bkq;
          B[k * 3 + q] = s * bkp + c * bkq;
        }
        for (int k = 0; k < 3; ++k) {  // B <- J^T B
          const double bpk = B[p * 3 + k], bqk = B[q * 3 + k];
          B[p * 3 + k] = c * bpk - s * bqk;
          B[q * 3 + k] = s * bpk + c * bqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
          V[k * 3 + p] = c * vkp - s * vkq;
          V[k * 3 + q] = s * vkp + c * vkq;
        }
      }
  }
}

__device__ inline double det3(const double* M) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// reference helpers/so3.py:86-128 (clamped-angle Rodrigues), in float64
__device__ inline void exp_map(const float* r, double* R) {
  const double x = r[0], y = r[1], z = r[2];
  const double n2 = x * x + y * y + z * z;
  const double a = sqrt(n2 > 1e-4 ? n2 : 1e-4);
  const double f1 = sin(a) / a, f2 = (1.0 - cos(a)) / (a * a);
  const double K[9] = {0, -z, y, z, 0, -x, -y, x, 0};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double kk = 0;
      for (int k = 0; k < 3; ++k) kk += K[i * 3 + k] * K[k * 3 + j];
      R[i * 3 + j] = f1 * K[i * 3 + j] + f2 * kk + (i == j ? 1.0 : 0.0);
    }
}

__global__ void metrics_rows_kernel(MetricsArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.T) return;
  const float* X = a.joints_gt + (size_t)t * MJ * 3;
  const float* Y = a.joints_hat + (size_t)t * MJ * 3;
  double* row = a.rows + (size_t)t * 65;

  // ---- Euclidean distances and Procrustes (align Y onto X; metrics.py:18-66 with optimal scale)
  double muX[3] = {0, 0, 0}, muY[3] = {0, 0, 0};
  for (int j = 0; j < MJ; ++j)
    for (int c = 0; c < 3; ++c) { muX[c] += X[j * 3 + c]; muY[c] += Y[j * 3 + c]; }
  for (int c = 0; c < 3; ++c) { muX[c] /= MJ; muY[c] /= MJ; }
  double ssX = 0, ssY = 0, A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < MJ; ++j) {
    double dx[3], dy[3];
    for (int c = 0; c < 3; ++c) { dx[c] = X[j * 3 + c] - muX[c]; dy[c] = Y[j * 3 + c] - muY[c]; }
    for (int c = 0; c < 3; ++c) { ssX += dx[c] * dx[c]; ssY += dy[c] * dy[c]; }
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k) A[i * 3 + k] += dx[i] * dy[k];  // X0^T Y0 (un-normalised)
    const double ex = (double)X[j * 3] - Y[j * 3], ey = (double)X[j * 3 + 1] - Y[j * 3 + 1],
                 ez = (double)X[j * 3 + 2] - Y[j * 3 + 2];
    row[j] = sqrt(ex * ex + ey * ey + ez * ez);
  }
  const double normX = sqrt(ssX), normY = sqrt(ssY);
  for (int i = 0; i < 9; ++i) A[i] /= (normX * normY);
  // SVD of A via the eigen-decomposition of A^T A:  A = U S V^T
  double B[9], V[9];
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) {
      double s = 0;
      for (int m = 0; m < 3; ++m) s += A[m * 3 + i] * A[m * 3 + k];
      B[i * 3 + k] = s;
    }
  jacobi_eig3(B, V);
  int ord[3] = {0, 1, 2};
  double lam[3] = {B[0], B[4], B[8]};
  for (int i = 0; i < 2; ++i)
    for (int k = i + 1; k < 3; ++k)
      if (lam[ord[k]] > lam[ord[i]]) { const int tmp = ord[i]; ord[i] = ord[k]; ord[k] = tmp; }
  double Vs[9], U[9], S[3];
  for (int c = 0; c < 3; ++c) {
    S[c] = sqrt(lam[ord[c]] > 0 ? lam[ord[c]] : 0.0);
    for (int r = 0; r < 3; ++r) Vs[r * 3 + c] = V[r * 3 + ord[c]];
  }
  for (int c = 0; c < 2; ++c) {
    double n = 0;
    for (int r = 0; r < 3; ++r) {
      double s = 0;
      for (int m = 0; m < 3; ++m) s += A[r * 3 + m] * Vs[m * 3 + c];
      U[r * 3 + c] = s;
      n += s * s;
    }
    n = sqrt(n);
    for (int r = 0; r < 3; ++r) U[r * 3 + c] = n > 0 ? U[r * 3 + c] / n : (r == c ? 1.0 : 0.0);
  }
  {  // third left vector: A v3 / s3 when well defined, else completes a right-handed/any orthonormal basis
    double u3[3], n = 0;
    for (int r = 0; r < 3; ++r) {
      double s = 0;
      for (int m = 0; m < 3; ++m) s += A[r * 3 + m] * Vs[m * 3 + 2];
      u3[r] = s;
      n += s * s;
    }
    n = sqrt(n);
    if (n > 1e-12 * (S[0] > 0 ? S[0] : 1.0)) {
      for (int r = 0; r < 3; ++r) U[r * 3 + 2] = u3[r] / n;
    } else {
      U[2] = U[3] * U[7] - U[6] * U[4];
      U[5] = U[6] * U[1] - U[0] * U[7];
      U[8] = U[0] * U[4] - U[3] * U[1];
    }
  }
  double Tm[9];
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) {
      double s = 0;
      for (int m = 0; m < 3; ++m) s += Vs[i * 3 + m] * U[k * 3 + m];
      Tm[i * 3 + k] = s;
    }
  if (det3(Tm) < 0) {  // make it a rotation (metrics.py:46-50)
    for (int r = 0; r < 3; ++r) Vs[r * 3 + 2] = -Vs[r * 3 + 2];
    S[2] = -S[2];
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k) {
        double s = 0;
        for (int m = 0; m < 3; ++m) s += Vs[i * 3 + m] * U[k * 3 + m];
        Tm[i * 3 + k] = s;
      }
  }
  const double trace = S[0] + S[1] + S[2];
  for (int j = 0; j < MJ; ++j) {
    double y0[3], z[3];
    for (int c = 0; c < 3; ++c) y0[c] = (Y[j * 3 + c] - muY[c]) / normY;
    for (int c = 0; c < 3; ++c)
      z[c] = normX * trace * (y0[0] * Tm[0 * 3 + c] + y0[1] * Tm[1 * 3 + c] + y0[2] * Tm[2 * 3 + c]) + muX[c];
    const double ex = X[j * 3] - z[0], ey = X[j * 3 + 1] - z[1], ez = X[j * 3 + 2] - z[2];
    row[22 + j] = sqrt(ex * ex + ey * ey + ez * ez);
  }

  // ---- global joint-angle error (root = identity; metrics.py:229-237, utils.py:165-199)
  if (!a.pose_gt) {
    for (int j = 0; j < MJ - 1; ++j) row[44 + j] = 0.0;
    return;
  }
  double Gg[MJ * 9], Gh[MJ * 9];
  for (int i = 0; i < 9; ++i) Gg[i] = Gh[i] = (i % 4 == 0) ? 1.0 : 0.0;
  const float* pg = a.pose_gt + (size_t)t * 63;
  const float* ph = a.pose_hat + (size_t)t * 63;
  for (int j = 1; j < MJ; ++j) {
    double Lg[9], Lh[9];
    exp_map(pg + (j - 1) * 3, Lg);
    exp_map(ph + (j - 1) * 3, Lh);
    const int p = a.parents[j];
    double tr = 0;
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k) {
        double sg = 0, sh = 0;
        for (int m = 0; m < 3; ++m) {
          sg += Gg[p * 9 + i * 3 + m] * Lg[m * 3 + k];
          sh += Gh[p * 9 + i * 3 + m] * Lh[m * 3 + k];
        }
        Gg[j * 9 + i * 3 + k] = sg;
        Gh[j * 9 + i * 3 + k] = sh;
      }
    for (int i = 0; i < 9; ++i) tr += Gg[j * 9 + i] * Gh[j * 9 + i];
    double cosv = (tr - 1.0) * 0.5;
    cosv = cosv > 1.0 ? 1.0 : (cosv < -1.0 ? -1.0 : cosv);
    row[44 + j - 1] = acos(cosv) * 57.29577951308232;
  }
}

hipError_t launch_metrics_rows(const MetricsArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(metrics_rows_kernel, dim3((a.T + 63) / 64), dim3(64), 0, stream, a);
  return hipGetLastError();
}

}  // namespace empose
