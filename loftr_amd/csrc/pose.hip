// Relative pose from matches: five-point essential matrix inside RANSAC + cheirality (host code, double precision).
//
// Replaces: estimate_pose (src/utils/metrics.py:72-98) = cv2.findEssentialMat(kpts0, kpts1, I, threshold, prob, RANSAC)
// followed by cv2.recoverPose(E, kpts0, kpts1, I, 1e9, mask) on intrinsics-normalised key points -- the pose step of the
// reference's test_step (compute_pose_errors, metrics.py:101-136).  OpenCV is a CPU library and is not in this image, so
// this is a restatement of the published algorithms, NOT of OpenCV's source:
//   * D. Nister, "An efficient solution to the five-point relative pose problem", PAMI 2004: the 4-dimensional null
//     space of the epipolar constraints, the ten cubic constraints (det E = 0, 2 E E^T E - tr(E E^T) E = 0) eliminated
//     by Gauss-Jordan to a 3x3 polynomial matrix in z whose determinant is a tenth-degree polynomial;
//   * RANSAC with the Sampson distance (squared, against threshold^2), adaptive iteration count from the confidence,
//     at most 1000 iterations -- the parameters cv2.findEssentialMat documents;
//   * the four (R, t) decompositions of E disambiguated by triangulating the inliers (positive depth in both views,
//     depth < 1e9), as cv2.recoverPose documents.
// PARITY UNPINNED: the random sampling sequence (and therefore the selected hypothesis on noisy data) cannot match
// OpenCV's; the tests check the solver on exact data and the recovered pose on synthetic scenes with known ground truth.
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <complex>
#include <vector>
#include "../../include/loftr_hip.h"

namespace {

// ---- small dense linear algebra -----------------------------------------------------------------------------
// symmetric eigen-decomposition by cyclic Jacobi: a (n x n, destroyed) -> eigenvalues w, eigenvectors in columns of v
void jacobi_eig(double* a, int n, double* w, double* v) {
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) v[i * n + j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) off += a[i * n + j] * a[i * n + j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[p * n + q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (a[q * n + q] - a[p * n + p]) / (2 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
        const double c = 1 / sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = a[k * n + p], akq = a[k * n + q];
          a[k * n + p] = c * akp - s * akq; a[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = a[p * n + k], aqk = a[q * n + k];
          a[p * n + k] = c * apk - s * aqk; a[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = v[k * n + p], vkq = v[k * n + q];
          v[k * n + p] = c * vkp - s * vkq; v[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < n; ++i) w[i] = a[i * n + i];
}

void mat3_mul(const double* a, const double* b, double* c) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}
double det3(const double* m) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}

// E = U diag(s) V^T with U, V proper or improper orthogonal (columns), via the eigen-decomposition of E^T E
void svd3(const double* E, double* U, double* s, double* V) {
  double ete[9], w[3], v[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) ete[i * 3 + j] = E[i] * E[j] + E[3 + i] * E[3 + j] + E[6 + i] * E[6 + j];
  jacobi_eig(ete, 3, w, v);
  int o[3] = {0, 1, 2};
  for (int i = 0; i < 3; ++i) for (int j = i + 1; j < 3; ++j) if (w[o[j]] > w[o[i]]) { int t = o[i]; o[i] = o[j]; o[j] = t; }
  for (int k = 0; k < 3; ++k) {
    s[k] = sqrt(w[o[k]] > 0 ? w[o[k]] : 0);
    for (int i = 0; i < 3; ++i) V[i * 3 + k] = v[i * 3 + o[k]];
  }
  double u[3][3];
  for (int k = 0; k < 2; ++k) {
    for (int i = 0; i < 3; ++i) u[k][i] = E[i * 3] * V[k] + E[i * 3 + 1] * V[3 + k] + E[i * 3 + 2] * V[6 + k];
    double nrm = sqrt(u[k][0] * u[k][0] + u[k][1] * u[k][1] + u[k][2] * u[k][2]);
    if (nrm < 1e-300) nrm = 1;
    for (int i = 0; i < 3; ++i) u[k][i] /= nrm;
  }
  // re-orthogonalise the second against the first, third = cross product
  double d = u[0][0] * u[1][0] + u[0][1] * u[1][1] + u[0][2] * u[1][2];
  for (int i = 0; i < 3; ++i) u[1][i] -= d * u[0][i];
  double nrm = sqrt(u[1][0] * u[1][0] + u[1][1] * u[1][1] + u[1][2] * u[1][2]);
  if (nrm < 1e-300) nrm = 1;
  for (int i = 0; i < 3; ++i) u[1][i] /= nrm;
  cross3(u[0], u[1], u[2]);
  for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) U[i * 3 + k] = u[k][i];
}

// ---- polynomials in (x, y, z) up to degree 3, in the column order of the elimination ---------------------------
// x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy | xz^2 xz x yz^2 yz y z^3 z^2 z 1
const int MONO[20][3] = {{3,0,0},{0,3,0},{2,1,0},{1,2,0},{2,0,1},{2,0,0},{0,2,1},{0,2,0},{1,1,1},{1,1,0},
                         {1,0,2},{1,0,1},{1,0,0},{0,1,2},{0,1,1},{0,1,0},{0,0,3},{0,0,2},{0,0,1},{0,0,0}};
int mono_index(int a, int b, int c) {
  for (int i = 0; i < 20; ++i) if (MONO[i][0] == a && MONO[i][1] == b && MONO[i][2] == c) return i;
  return -1;
}
struct Poly { double c[20]; Poly() { memset(c, 0, sizeof(c)); } };
Poly operator+(const Poly& a, const Poly& b) { Poly r; for (int i = 0; i < 20; ++i) r.c[i] = a.c[i] + b.c[i]; return r; }
Poly operator-(const Poly& a, const Poly& b) { Poly r; for (int i = 0; i < 20; ++i) r.c[i] = a.c[i] - b.c[i]; return r; }
Poly operator*(const Poly& a, double s) { Poly r; for (int i = 0; i < 20; ++i) r.c[i] = a.c[i] * s; return r; }
Poly operator*(const Poly& a, const Poly& b) {            // the product must stay within degree 3 (callers guarantee it)
  static int table[20][20];
  static bool init = false;
  if (!init) {
    for (int i = 0; i < 20; ++i) for (int j = 0; j < 20; ++j) {
      const int x = MONO[i][0] + MONO[j][0], y = MONO[i][1] + MONO[j][1], z = MONO[i][2] + MONO[j][2];
      table[i][j] = x + y + z <= 3 ? mono_index(x, y, z) : -1;
    }
    init = true;
  }
  Poly r;
  for (int i = 0; i < 20; ++i) if (a.c[i] != 0)
    for (int j = 0; j < 20; ++j) if (b.c[j] != 0 && table[i][j] >= 0) r.c[table[i][j]] += a.c[i] * b.c[j];
  return r;
}

// polynomials in z
typedef std::vector<double> P1;
P1 p1_mul(const P1& a, const P1& b) { P1 r(a.size() + b.size() - 1, 0.0); for (size_t i = 0; i < a.size(); ++i) for (size_t j = 0; j < b.size(); ++j) r[i + j] += a[i] * b[j]; return r; }
P1 p1_sub(const P1& a, const P1& b) { P1 r(a.size() > b.size() ? a.size() : b.size(), 0.0); for (size_t i = 0; i < a.size(); ++i) r[i] += a[i]; for (size_t i = 0; i < b.size(); ++i) r[i] -= b[i]; return r; }
P1 p1_add(const P1& a, const P1& b) { P1 r(a.size() > b.size() ? a.size() : b.size(), 0.0); for (size_t i = 0; i < a.size(); ++i) r[i] += a[i]; for (size_t i = 0; i < b.size(); ++i) r[i] += b[i]; return r; }
double p1_eval(const P1& a, double z) { double r = 0; for (size_t i = a.size(); i-- > 0;) r = r * z + a[i]; return r; }

// real roots of a polynomial (ascending coefficients) by Aberth-Ehrlich iteration + Newton polishing on the real axis
void real_roots(const P1& pin, std::vector<double>& roots) {
  P1 p = pin;
  while (p.size() > 1 && fabs(p.back()) < 1e-14 * fabs(p[0] + 1e-300) && fabs(p.back()) < 1e-300) p.pop_back();
  double scale = 0;
  for (double c : p) scale = fabs(c) > scale ? fabs(c) : scale;
  if (scale == 0) return;
  while (p.size() > 1 && fabs(p.back()) < 1e-13 * scale) p.pop_back();        // numerically lower degree
  const int n = (int)p.size() - 1;
  if (n < 1) return;
  typedef std::complex<double> cd;
  double radius = 0;
  for (int i = 0; i < n; ++i) { const double r = fabs(p[i] / p[n]); radius = r > radius ? r : radius; }
  radius = 1 + radius;                                                         // Cauchy bound
  std::vector<cd> z(n);
  for (int i = 0; i < n; ++i) z[i] = std::polar(radius * (0.3 + 0.7 * (i + 1) / n), 2 * M_PI * i / n + 0.4);
  for (int it = 0; it < 200; ++it) {
    double change = 0;
    for (int i = 0; i < n; ++i) {
      cd f = p[n], df = 0;
      for (int k = n - 1; k >= 0; --k) { df = df * z[i] + f; f = f * z[i] + p[k]; }
      if (std::abs(f) < 1e-300) continue;
      const cd ratio = f / (std::abs(df) > 1e-300 ? df : cd(1e-300, 0));
      cd sum = 0;
      for (int j = 0; j < n; ++j) if (j != i) { const cd d = z[i] - z[j]; sum += 1.0 / (std::abs(d) > 1e-300 ? d : cd(1e-300, 0)); }
      const cd step = ratio / (1.0 - ratio * sum);
      z[i] -= step;
      change = std::abs(step) > change ? std::abs(step) : change;
    }
    if (change < 1e-14 * radius) break;
  }
  for (int i = 0; i < n; ++i) {
    if (fabs(z[i].imag()) > 1e-6 * (1 + fabs(z[i].real()))) continue;
    double x = z[i].real();
    for (int it = 0; it < 8; ++it) {                                            // polish
      double f = p[n], df = 0;
      for (int k = n - 1; k >= 0; --k) { df = df * x + f; f = f * x + p[k]; }
      if (fabs(df) < 1e-300) break;
      x -= f / df;
    }
    bool dup = false;
    for (double r : roots) if (fabs(r - x) < 1e-9 * (1 + fabs(x))) dup = true;
    if (!dup) roots.push_back(x);
  }
}

// ---- five-point solver: n >= 5 normalised correspondences -> up to 10 essential matrices (row-major 3x3) --------
int five_point(const double* q0, const double* q1, const int* idx, int n, double (*Es)[9]) {
  // epipolar constraints q1^T E q0 = 0  ->  A e = 0,  e = row-major E
  double ata[81];
  memset(ata, 0, sizeof(ata));
  for (int k = 0; k < n; ++k) {
    const int i = idx ? idx[k] : k;
    const double x0 = q0[2 * i], y0 = q0[2 * i + 1], x1 = q1[2 * i], y1 = q1[2 * i + 1];
    const double r[9] = {x1 * x0, x1 * y0, x1, y1 * x0, y1 * y0, y1, x0, y0, 1.0};
    for (int a = 0; a < 9; ++a) for (int b = 0; b < 9; ++b) ata[a * 9 + b] += r[a] * r[b];
  }
  double w[9], v[81];
  jacobi_eig(ata, 9, w, v);
  int order[9];
  for (int i = 0; i < 9; ++i) order[i] = i;
  for (int i = 0; i < 9; ++i) for (int j = i + 1; j < 9; ++j) if (w[order[j]] < w[order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
  double N[4][9];                                           // null-space basis X, Y, Z, W
  for (int b = 0; b < 4; ++b) for (int i = 0; i < 9; ++i) N[b][i] = v[i * 9 + order[b]];
  // E(x, y, z) = x X + y Y + z Z + W as polynomials
  Poly E[9];
  const int ix = mono_index(1, 0, 0), iy = mono_index(0, 1, 0), iz = mono_index(0, 0, 1), i1 = mono_index(0, 0, 0);
  for (int i = 0; i < 9; ++i) { E[i].c[ix] = N[0][i]; E[i].c[iy] = N[1][i]; E[i].c[iz] = N[2][i]; E[i].c[i1] = N[3][i]; }
  Poly eq[10];
  // det E = 0
  eq[0] = E[0] * (E[4] * E[8] - E[5] * E[7]) - E[1] * (E[3] * E[8] - E[5] * E[6]) + E[2] * (E[3] * E[7] - E[4] * E[6]);
  // 2 E E^T E - tr(E E^T) E = 0
  Poly EEt[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
    EEt[i * 3 + j] = E[i * 3] * E[j * 3] + E[i * 3 + 1] * E[j * 3 + 1] + E[i * 3 + 2] * E[j * 3 + 2];
  const Poly tr = EEt[0] + EEt[4] + EEt[8];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    const Poly t = EEt[i * 3] * E[j] + EEt[i * 3 + 1] * E[3 + j] + EEt[i * 3 + 2] * E[6 + j];
    eq[1 + i * 3 + j] = t * 2.0 - tr * E[i * 3 + j];
  }
  // Gauss-Jordan on the first ten columns (partial pivoting)
  double M[10][20];
  for (int r = 0; r < 10; ++r) for (int c = 0; c < 20; ++c) M[r][c] = eq[r].c[c];
  for (int c = 0; c < 10; ++c) {
    int piv = c;
    for (int r = c + 1; r < 10; ++r) if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
    if (fabs(M[piv][c]) < 1e-14) return 0;                   // degenerate sample
    if (piv != c) for (int k = 0; k < 20; ++k) { const double t = M[c][k]; M[c][k] = M[piv][k]; M[piv][k] = t; }
    const double inv = 1.0 / M[c][c];
    for (int k = 0; k < 20; ++k) M[c][k] *= inv;
    for (int r = 0; r < 10; ++r) if (r != c) {
      const double f = M[r][c];
      if (f != 0) for (int k = 0; k < 20; ++k) M[r][k] -= f * M[c][k];
    }
  }
  // rows e..j (4..9): <x^2z>, <x^2>, <y^2z>, <y^2>, <xyz>, <xy>;  k = e - z f, l = g - z h, m = i - z j are
  // x * p3(z) + y * q3(z) + r4(z): the 3x3 polynomial matrix B(z)
  P1 B[3][3];
  for (int t = 0; t < 3; ++t) {
    const double* a = M[4 + 2 * t];                         // the row that carries the extra z
    const double* b = M[5 + 2 * t];
    B[t][0] = P1{a[12], a[11] - b[12], a[10] - b[11], -b[10]};                     // x: 1, z, z^2, z^3
    B[t][1] = P1{a[15], a[14] - b[15], a[13] - b[14], -b[13]};                     // y
    B[t][2] = P1{a[19], a[18] - b[19], a[17] - b[18], a[16] - b[17], -b[16]};      // 1: up to z^4
  }
  const P1 det = p1_add(p1_sub(p1_mul(B[0][0], p1_sub(p1_mul(B[1][1], B[2][2]), p1_mul(B[1][2], B[2][1]))),
                               p1_mul(B[0][1], p1_sub(p1_mul(B[1][0], B[2][2]), p1_mul(B[1][2], B[2][0])))),
                        p1_mul(B[0][2], p1_sub(p1_mul(B[1][0], B[2][1]), p1_mul(B[1][1], B[2][0]))));
  std::vector<double> zs;
  real_roots(det, zs);
  int ns = 0;
  for (double z : zs) {
    if (ns >= 10) break;
    double b[3][3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) b[r][c] = p1_eval(B[r][c], z);
    // (x, y, 1) spans the null space of B(z): cross product of the two best-conditioned rows
    double best[3] = {0, 0, 0}, bestn = -1;
    for (int r0 = 0; r0 < 3; ++r0) for (int r1 = r0 + 1; r1 < 3; ++r1) {
      double c[3];
      cross3(b[r0], b[r1], c);
      const double nn = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
      if (nn > bestn && fabs(c[2]) > 1e-12 * sqrt(nn + 1e-300)) { bestn = nn; best[0] = c[0]; best[1] = c[1]; best[2] = c[2]; }
    }
    if (bestn <= 0) continue;
    const double x = best[0] / best[2], y = best[1] / best[2];
    double nrm = 0;
    for (int i = 0; i < 9; ++i) { Es[ns][i] = x * N[0][i] + y * N[1][i] + z * N[2][i] + N[3][i]; nrm += Es[ns][i] * Es[ns][i]; }
    nrm = sqrt(nrm);
    if (!(nrm > 1e-300)) continue;
    for (int i = 0; i < 9; ++i) Es[ns][i] /= nrm;
    ++ns;
  }
  return ns;
}

// squared Sampson distance of every correspondence; returns the number below thr2
long score(const double* E, const double* q0, const double* q1, long n, double thr2, uint8_t* mask) {
  long cnt = 0;
  for (long i = 0; i < n; ++i) {
    const double x0 = q0[2 * i], y0 = q0[2 * i + 1], x1 = q1[2 * i], y1 = q1[2 * i + 1];
    const double l0 = E[0] * x0 + E[1] * y0 + E[2], l1 = E[3] * x0 + E[4] * y0 + E[5], l2 = E[6] * x0 + E[7] * y0 + E[8];   // E q0
    const double m0 = E[0] * x1 + E[3] * y1 + E[6], m1 = E[1] * x1 + E[4] * y1 + E[7];                                     // E^T q1
    const double r = x1 * l0 + y1 * l1 + l2;
    const double den = l0 * l0 + l1 * l1 + m0 * m0 + m1 * m1;
    const bool in = den > 0 && r * r < thr2 * den;
    if (mask) mask[i] = in;
    cnt += in;
  }
  return cnt;
}

struct Rng {                                                 // xorshift64*
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) { if (!s) s = 1; }
  uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1Dull; }
  long below(long n) { return (long)(next() % (uint64_t)n); }
};

// cheirality: number of inliers in front of both cameras for (R, t); optionally marks them
long cheirality(const double* R, const double* t, const double* q0, const double* q1, long n, const uint8_t* in, double dist,
                uint8_t* good) {
  long cnt = 0;
  for (long i = 0; i < n; ++i) {
    if (good) good[i] = 0;
    if (!in[i]) continue;
    // P0 = [I | 0], P1 = [R | t]; linear triangulation: A X = 0 with the four rows below, X via the smallest eigenvector
    const double x0 = q0[2 * i], y0 = q0[2 * i + 1], x1 = q1[2 * i], y1 = q1[2 * i + 1];
    double A[4][4] = {{-1, 0, x0, 0}, {0, -1, y0, 0},
                      {x1 * R[6] - R[0], x1 * R[7] - R[1], x1 * R[8] - R[2], x1 * t[2] - t[0]},
                      {y1 * R[6] - R[3], y1 * R[7] - R[4], y1 * R[8] - R[5], y1 * t[2] - t[1]}};
    double ata[16], w[4], v[16];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) { double s = 0; for (int k = 0; k < 4; ++k) s += A[k][a] * A[k][b]; ata[a * 4 + b] = s; }
    jacobi_eig(ata, 4, w, v);
    int m = 0;
    for (int k = 1; k < 4; ++k) if (w[k] < w[m]) m = k;
    double X[4] = {v[m], v[4 + m], v[8 + m], v[12 + m]};
    if (fabs(X[3]) < 1e-300) continue;
    for (int k = 0; k < 3; ++k) X[k] /= X[3];
    const double z0 = X[2];
    const double z1 = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
    const bool ok = z0 > 0 && z0 < dist && z1 > 0 && z1 < dist;
    if (good) good[i] = ok;
    cnt += ok;
  }
  return cnt;
}

}  // namespace

extern "C" int loftr_five_point(const double* q0, const double* q1, int n, double* E_out, int* n_solutions) {
  if (!q0 || !q1 || !E_out || !n_solutions || n < 5) return LOFTR_ERR_BAD_ARG;
  double Es[10][9];
  const int ns = five_point(q0, q1, nullptr, n, Es);
  memcpy(E_out, Es, sizeof(double) * 9 * ns);
  *n_solutions = ns;
  return LOFTR_OK;
}

extern "C" int loftr_estimate_pose(const float* kpts0, const float* kpts1, long M, const float* K0, const float* K1,
                                   float thresh_px, float conf, unsigned seed, float* R_out, float* t_out,
                                   uint8_t* inliers_out, long* n_inliers) {
  if (!kpts0 || !kpts1 || !K0 || !K1 || !R_out || !t_out || !inliers_out || !n_inliers || M < 0) return LOFTR_ERR_BAD_ARG;
  *n_inliers = -1;                                           // "None" of the reference: too few points / no model
  if (M < 5) return LOFTR_OK;
  std::vector<double> q0(2 * M), q1(2 * M);
  for (long i = 0; i < M; ++i) {                             // (kpts - [cx, cy]) / [fx, fy]        metrics.py:76-77
    q0[2 * i] = ((double)kpts0[2 * i] - K0[2]) / K0[0]; q0[2 * i + 1] = ((double)kpts0[2 * i + 1] - K0[5]) / K0[4];
    q1[2 * i] = ((double)kpts1[2 * i] - K1[2]) / K1[0]; q1[2 * i + 1] = ((double)kpts1[2 * i + 1] - K1[5]) / K1[4];
  }
  // ransac_thr = thresh / mean([K0[0,0], K1[1,1], K0[0,0], K1[1,1]])                             metrics.py:80
  const double thr = (double)thresh_px / (((double)K0[0] + K1[4] + K0[0] + K1[4]) / 4.0);
  const double thr2 = thr * thr;
  Rng rng(seed);
  double bestE[9] = {0};
  long best = 0;
  int max_iters = 1000, iters = max_iters;
  for (int it = 0; it < iters; ++it) {
    int idx[5];
    for (int k = 0; k < 5;) {
      const int c = (int)rng.below(M);
      bool dup = false;
      for (int j = 0; j < k; ++j) dup = dup || idx[j] == c;
      if (!dup) idx[k++] = c;
    }
    double Es[10][9];
    const int ns = five_point(q0.data(), q1.data(), idx, 5, Es);
    for (int s = 0; s < ns; ++s) {
      const long cnt = score(Es[s], q0.data(), q1.data(), M, thr2, nullptr);
      if (cnt > best) {
        best = cnt;
        memcpy(bestE, Es[s], sizeof(bestE));
        const double w = (double)cnt / (double)M;            // adaptive iteration count from the confidence
        const double p_all = pow(w, 5.0);
        if (p_all > 1 - 1e-12) iters = it + 1;
        else if (p_all > 1e-12) {
          const double need = log(1.0 - (double)conf) / log(1.0 - p_all);
          if (need < iters) iters = need < it + 1 ? it + 1 : (int)ceil(need);
        }
      }
    }
  }
  if (best < 5) return LOFTR_OK;
  std::vector<uint8_t> in(M), good(M), bestgood(M);
  score(bestE, q0.data(), q1.data(), M, thr2, in.data());
  // E = U diag(1,1,0) V^T -> R in {U W V^T, U W^T V^T}, t = +-u3
  double U[9], s[3], V[9];
  svd3(bestE, U, s, V);
  if (det3(U) < 0) for (int i = 0; i < 9; ++i) U[i] = -U[i];
  if (det3(V) < 0) for (int i = 0; i < 9; ++i) V[i] = -V[i];
  const double Wm[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1}, Wt[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
  double Vt[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Vt[i * 3 + j] = V[j * 3 + i];
  double R1[9], R2[9], tmp[9];
  mat3_mul(U, Wm, tmp); mat3_mul(tmp, Vt, R1);
  mat3_mul(U, Wt, tmp); mat3_mul(tmp, Vt, R2);
  const double tp[3] = {U[2], U[5], U[8]}, tn[3] = {-U[2], -U[5], -U[8]};
  const double* Rs[4] = {R1, R2, R1, R2};
  const double* ts[4] = {tp, tp, tn, tn};
  long bestc = -1; int bi = 0;
  for (int c = 0; c < 4; ++c) {
    const long cnt = cheirality(Rs[c], ts[c], q0.data(), q1.data(), M, in.data(), 1e9, good.data());
    if (cnt > bestc) { bestc = cnt; bi = c; bestgood = good; }
  }
  if (bestc <= 0) return LOFTR_OK;                            // recoverPose found no point in front of both cameras
  for (int i = 0; i < 9; ++i) R_out[i] = (float)Rs[bi][i];
  for (int i = 0; i < 3; ++i) t_out[i] = (float)ts[bi][i];
  memcpy(inliers_out, bestgood.data(), (size_t)M);
  *n_inliers = bestc;
  return LOFTR_OK;
}
