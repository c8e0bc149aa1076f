// eig3.hpp — FastEigen3x3 (Homework1/YuF_KIT-第1章作业/my_pybind11/src/mylib.cpp:9-189; the reference took it from open3d):
// unit eigenvector of the SMALLEST eigenvalue of a symmetric 3x3, closed form — scale by the (signed) largest coefficient,
// trigonometric roots of the characteristic polynomial, eigenvectors from row cross products.  One implementation for the
// host entry point pcr_fast_eigen3x3 (ground fit, N2) and the device (per-point normals, N1).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>

#define EIG3_HD __host__ __device__ inline

namespace pcr {
namespace eig3 {

struct V3 {
    double a, b, c;
};
EIG3_HD V3 cross(const V3& u, const V3& v) { return V3{ u.b * v.c - u.c * v.b, u.c * v.a - u.a * v.c, u.a * v.b - u.b * v.a }; }
EIG3_HD double dot(const V3& u, const V3& v) { return u.a * v.a + u.b * v.b + u.c * v.c; }
EIG3_HD V3 scaled(double s, const V3& u) { return V3{ u.a * s, u.b * s, u.c * s }; }
EIG3_HD V3 divided(const V3& u, double s) { return V3{ u.a / s, u.b / s, u.c / s }; }
EIG3_HD V3 minus(const V3& u, const V3& v) { return V3{ u.a - v.a, u.b - v.b, u.c - v.c }; }

// kernel vector of (M - lambda I): the longest of the pairwise row cross products (mylib.cpp:9-38)
EIG3_HD V3 null_vector(const double M[9], double lambda)
{
    const V3 r0{ M[0] - lambda, M[1], M[2] }, r1{ M[1], M[4] - lambda, M[5] }, r2{ M[2], M[5], M[8] - lambda };
    const V3 x01 = cross(r0, r1), x02 = cross(r0, r2), x12 = cross(r1, r2);
    const double l0 = dot(x01, x01), l1 = dot(x02, x02), l2 = dot(x12, x12);
    double longest = l0;
    int which = 0;
    if (l1 > longest) { longest = l1; which = 1; }
    if (l2 > longest) which = 2;
    return which == 0 ? divided(x01, sqrt(l0)) : which == 1 ? divided(x02, sqrt(l1)) : divided(x12, sqrt(l2));
}

// eigenvector of `lambda` inside the plane orthogonal to the known eigenvector w (mylib.cpp:40-102)
EIG3_HD V3 in_plane_vector(const double M[9], const V3& w, double lambda)
{
    V3 U;
    if (fabs(w.a) > fabs(w.b)) {
        const double inv = 1 / sqrt(w.a * w.a + w.c * w.c);
        U = V3{ -w.c * inv, 0, w.a * inv };
    } else {
        const double inv = 1 / sqrt(w.b * w.b + w.c * w.c);
        U = V3{ 0, w.c * inv, -w.b * inv };
    }
    const V3 V = cross(w, U);
    const V3 MU{ M[0] * U.a + M[1] * U.b + M[2] * U.c, M[1] * U.a + M[4] * U.b + M[5] * U.c, M[2] * U.a + M[5] * U.b + M[8] * U.c };
    const V3 MV{ M[0] * V.a + M[1] * V.b + M[2] * V.c, M[1] * V.a + M[4] * V.b + M[5] * V.c, M[2] * V.a + M[5] * V.b + M[8] * V.c };
    double g00 = U.a * MU.a + U.b * MU.b + U.c * MU.c - lambda;
    double g01 = U.a * MV.a + U.b * MV.b + U.c * MV.c;
    double g11 = V.a * MV.a + V.b * MV.b + V.c * MV.c - lambda;
    const double n00 = fabs(g00), n01 = fabs(g01), n11 = fabs(g11);
    if (n00 >= n11) {
        if (!(fmax(n00, n01) > 0)) return U;
        if (n00 >= n01) { g01 /= g00; g00 = 1 / sqrt(1 + g01 * g01); g01 *= g00; }
        else { g00 /= g01; g01 = 1 / sqrt(1 + g00 * g00); g00 *= g01; }
        return minus(scaled(g01, U), scaled(g00, V));
    }
    if (!(fmax(n11, n01) > 0)) return U;
    if (n11 >= n01) { g01 /= g11; g11 = 1 / sqrt(1 + g01 * g01); g01 *= g11; }
    else { g11 /= g01; g01 = 1 / sqrt(1 + g11 * g11); g11 *= g01; }
    return minus(scaled(g11, U), scaled(g01, V));
}

// mylib.cpp:105-189.  (0,0,0) when the signed maximum coefficient is 0 (:112-115).
EIG3_HD void smallest_eigenvector(const double A[9], double normal[3])
{
    double M[9];
    double big = A[0];                                        // maxCoeff(): the signed maximum
    for (int k = 1; k < 9; k++) big = A[k] > big ? A[k] : big;
    V3 out{ 0, 0, 0 };
    if (big == 0) { normal[0] = normal[1] = normal[2] = 0; return; }
    for (int k = 0; k < 9; k++) M[k] = A[k] / big;
    const double off2 = M[1] * M[1] + M[2] * M[2] + M[5] * M[5];
    if (off2 > 0) {
        const double mean = (M[0] + M[4] + M[8]) / 3;
        const double d0 = M[0] - mean, d1 = M[4] - mean, d2 = M[8] - mean;
        const double spread = sqrt((d0 * d0 + d1 * d1 + d2 * d2 + off2 * 2) / 6);
        const double k00 = d1 * d2 - M[5] * M[5], k01 = M[1] * d2 - M[5] * M[2], k02 = M[1] * M[5] - d1 * M[2];
        const double det = (d0 * k00 - M[1] * k01 + M[2] * k02) / (spread * spread * spread);
        const double half = fmin(fmax(det * 0.5, -1.0), 1.0);
        const double phi = acos(half) / (double)3;
        const double two_thirds_pi = 2.09439510239319549;
        const double hi = cos(phi) * 2, lo = cos(phi + two_thirds_pi) * 2, mid = -(lo + hi);
        const double w0 = mean + spread * lo, w1 = mean + spread * mid, w2 = mean + spread * hi;
        if (half >= 0) {                                      // the largest root is the well-separated one: start there
            const V3 e2 = null_vector(M, w2);
            if (w2 < w0 && w2 < w1) out = e2;
            else {
                const V3 e1 = in_plane_vector(M, e2, w1);
                out = (w1 < w0 && w1 < w2) ? e1 : cross(e1, e2);
            }
        } else {
            const V3 e0 = null_vector(M, w0);
            if (w0 < w1 && w0 < w2) out = e0;
            else {
                const V3 e1 = in_plane_vector(M, e0, w1);
                out = (w1 < w0 && w1 < w2) ? e1 : cross(e0, e1);
            }
        }
    } else {                                                  // already diagonal (:177-187)
        const double a0 = M[0] * big, a1 = M[4] * big, a2 = M[8] * big;
        if (a0 < a1 && a0 < a2) out = V3{ 1, 0, 0 };
        else if (a1 < a0 && a1 < a2) out = V3{ 0, 1, 0 };
        else out = V3{ 0, 0, 1 };
    }
    normal[0] = out.a; normal[1] = out.b; normal[2] = out.c;
}

}  // namespace eig3
}  // namespace pcr
