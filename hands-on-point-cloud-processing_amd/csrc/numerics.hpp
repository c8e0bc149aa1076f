// numerics.hpp — 3x3 numerics of the Kabsch step (A7 solve) and pose composition, shared by the host path
// (numerics.cpp) and the device-resident ICP update kernel (kabsch.hip): ONE implementation, compiled for both,
// so that the synchronous and the pipelined ICP loops produce the same pose.
// Follows Homework9/hw9/src/registration.cpp:979-1002; Eigen's JacobiSVD<Matrix3f> is replaced by an f64
// one-sided Jacobi SVD with singular values sorted descending (the order Eigen delivers, which matters for
// the reference's det<0 branch, :990-996).  Every operation is a plain IEEE f64/f32 op (no FMA contraction:
// the library is built with -ffp-contract=off), sqrt and division are correctly rounded on both sides.
#pragma once

#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

#define PCR_HD __host__ __device__ inline

namespace pcr {
namespace num {

struct M3 {
    double a[3][3];
};

// (Every index below is a compile-time constant — the pair loop is a template, the descending order is a network of conditional
// column swaps, the rank-1 pivot is picked with selects — so that on the device the two 3x3 work matrices live in registers: with
// run-time column indices they sat in scratch memory, a dependent memory round trip per access inside one thread's serial chain.
// The arithmetic — operations, operands and order — is that of the loops it replaces; host and device share it.)
template <int P, int Q>
PCR_HD double col_dot(const M3& m)
{
    return m.a[0][P] * m.a[0][Q] + m.a[1][P] * m.a[1][Q] + m.a[2][P] * m.a[2][Q];
}

template <int P, int Q>
PCR_HD void rotate_cols(M3& m, double c, double s)
{
    for (int r = 0; r < 3; r++) {
        const double mp = m.a[r][P], mq = m.a[r][Q];
        m.a[r][P] = c * mp - s * mq;
        m.a[r][Q] = s * mp + c * mq;
    }
}

PCR_HD double dabs(double v) { return v < 0 ? -v : v; }

// one Jacobi rotation of the column pair (P, Q); false when the pair is already orthogonal to working precision.
// (Measured and dropped: selecting the rotated columns instead of branching, so that the convergence test's square root runs beside
// the rotation's own chain — the converged pairs of the last sweep then pay the whole chain: update kernel 8.5 -> 10.9 us; only the
// first division hoisted above the branch: no difference.)
template <int P, int Q>
PCR_HD bool jacobi_pair(M3& W, M3& R)
{
    const double eps = 2.220446049250313e-16;   // DBL_EPSILON
    const double alpha = col_dot<P, P>(W), beta = col_dot<Q, Q>(W), gamma = col_dot<P, Q>(W);
    if (gamma == 0.0 || dabs(gamma) <= eps * sqrt(alpha * beta)) return false;
    const double zeta = (beta - alpha) / (2.0 * gamma);
    const double tn = (zeta >= 0 ? 1.0 : -1.0) / (dabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + tn * tn), s = c * tn;
    rotate_cols<P, Q>(W, c, s);
    rotate_cols<P, Q>(R, c, s);
    return true;
}

// columns A and B of (sv, W, R) change places when the later one carries the larger singular value
template <int A, int B>
PCR_HD void order_cols(double sv[3], M3& W, M3& R)
{
    const bool sw = sv[B] > sv[A];
    const double sa = sv[A], sb = sv[B];
    sv[A] = sw ? sb : sa; sv[B] = sw ? sa : sb;
    for (int r = 0; r < 3; r++) {
        const double wa = W.a[r][A], wb = W.a[r][B], ra = R.a[r][A], rb = R.a[r][B];
        W.a[r][A] = sw ? wb : wa; W.a[r][B] = sw ? wa : wb;
        R.a[r][A] = sw ? rb : ra; R.a[r][B] = sw ? ra : rb;
    }
}

// A row-major; A = U diag(S) V^T, S descending
PCR_HD void svd3(const double A[9], double U[9], double S[3], double V[9])
{
    const double eps = 2.220446049250313e-16;   // DBL_EPSILON
    M3 W, R;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { W.a[r][c] = A[3 * r + c]; R.a[r][c] = r == c ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; sweep++) {
        const bool r01 = jacobi_pair<0, 1>(W, R);
        const bool r02 = jacobi_pair<0, 2>(W, R);
        const bool r12 = jacobi_pair<1, 2>(W, R);
        if (!(r01 || r02 || r12)) break;
    }
    double sv[3];
    sv[0] = sqrt(col_dot<0, 0>(W)); sv[1] = sqrt(col_dot<1, 1>(W)); sv[2] = sqrt(col_dot<2, 2>(W));
    order_cols<0, 1>(sv, W, R);          // the exchange network (0,1) (0,2) (1,2) on '>' — ties keep their column order
    order_cols<0, 2>(sv, W, R);
    order_cols<1, 2>(sv, W, R);
    const double smax = sv[0];
    bool have[3] = { false, false, false };
    for (int c = 0; c < 3; c++) {
        S[c] = sv[c];
        for (int r = 0; r < 3; r++) V[3 * r + c] = R.a[r][c];
        if (sv[c] > 0.0 && sv[c] > smax * eps * 8.0) {
            for (int r = 0; r < 3; r++) U[3 * r + c] = W.a[r][c] / sv[c];
            have[c] = true;
        } else {
            for (int r = 0; r < 3; r++) U[3 * r + c] = 0.0;
        }
    }
    if (!have[0]) {   // A == 0
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) U[3 * r + c] = r == c ? 1.0 : 0.0;
        return;
    }
    if (!have[1]) {   // rank 1: any unit vector orthogonal to u0
        const double u0[3] = { U[0], U[3], U[6] };
        int k = 0;
        double uk = u0[0];
        if (dabs(u0[1]) < dabs(uk)) { k = 1; uk = u0[1]; }
        if (dabs(u0[2]) < dabs(uk)) { k = 2; uk = u0[2]; }
        double v[3] = { -uk * u0[0], -uk * u0[1], -uk * u0[2] };
        for (int i = 0; i < 3; i++) v[i] = i == k ? v[i] + 1.0 : v[i];
        const double nv = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        U[1] = v[0] / nv; U[4] = v[1] / nv; U[7] = v[2] / nv;
    }
    if (!have[2]) {   // u2 = u0 x u1
        U[2] = U[3] * U[7] - U[6] * U[4];
        U[5] = U[6] * U[1] - U[0] * U[7];
        U[8] = U[0] * U[4] - U[3] * U[1];
    }
}

PCR_HD void mul3(const double A[9], const double B[9], double C[9])
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}

// The two Kabsch blocks of the reference differ in ONE statement order (both kept as written):
//   ICPpoint2point (registration.cpp:979-998):  R = U V^T; if det R < 0: R = V B U^T;  t = qbar - R pbar      (t from the final R)
//   RANSAC         (registration.cpp:373-392):  R = U V^T; t = qbar - R pbar;  if det R < 0: R = V B U^T      (t keeps the FIRST R)
// T_FIRST selects the second.  Returns 0, or -1 when no pair was kept.
template <bool T_FIRST>
PCR_HD int kabsch_solve_as(const double sums[16], float R[9], float t[3])
{
    const double M = sums[15];
    if (!(M > 0.0)) return -1;
    double pbar[3], qbar[3], H[9];
    for (int c = 0; c < 3; c++) { pbar[c] = sums[c] / M; qbar[c] = sums[3 + c] / M; }        // :979-980 / :373-374
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) H[3 * r + c] = sums[6 + 3 * r + c] - M * qbar[r] * pbar[c];   // :982-985 / :375-379
    double U[9], S[3], V[9], Vt[9], Ut[9], Rd[9];
    svd3(H, U, S, V);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { Vt[3 * r + c] = V[3 * c + r]; Ut[3 * r + c] = U[3 * c + r]; }
    mul3(U, Vt, Rd);                                                                            // :988 / :382
    if (T_FIRST) {                                                                              // :383
        for (int r = 0; r < 3; r++) {
            const double Rp = ((double)(float)Rd[3 * r] * pbar[0] + (double)(float)Rd[3 * r + 1] * pbar[1]) + (double)(float)Rd[3 * r + 2] * pbar[2];
            t[r] = (float)(qbar[r] - Rp);
        }
    }
    const double det = Rd[0] * (Rd[4] * Rd[8] - Rd[5] * Rd[7]) - Rd[1] * (Rd[3] * Rd[8] - Rd[5] * Rd[6])
                     + Rd[2] * (Rd[3] * Rd[7] - Rd[4] * Rd[6]);
    if (det < 0) {                                                                              // :990-996 / :386-392
        double VB[9];
        for (int r = 0; r < 3; r++) { VB[3 * r] = V[3 * r]; VB[3 * r + 1] = V[3 * r + 1]; VB[3 * r + 2] = V[3 * r + 2] * det; }
        mul3(VB, Ut, Rd);   // V * B * U^T, as the reference writes it
    }
    for (int k = 0; k < 9; k++) R[k] = (float)Rd[k];
    if (!T_FIRST) {
        for (int r = 0; r < 3; r++) {                                                           // :998
            const double Rp = ((double)R[3 * r] * pbar[0] + (double)R[3 * r + 1] * pbar[1]) + (double)R[3 * r + 2] * pbar[2];
            t[r] = (float)(qbar[r] - Rp);
        }
    }
    return 0;
}

PCR_HD int kabsch_solve(const double sums[16], float R[9], float t[3]) { return kabsch_solve_as<false>(sums, R, t); }
PCR_HD int kabsch_solve_ransac(const double sums[16], float R[9], float t[3]) { return kabsch_solve_as<true>(sums, R, t); }

// ---- exact, order-independent accumulation of the Kabsch moments (kabsch.hip) -------------------------------------------
// Every term of the 16 sums is exactly representable in f64 (an f32 coordinate, or the product of two), but an f64 running
// sum rounds, so its bits depend on the order of the additions: on the launch geometry, on the order the queries are visited
// in, on how many GPUs share the work.  Instead each term is cut into 40-bit integer LIMBS on one fixed-point grid per ICP
// (unit 2^(e-80) for coordinates, 2^(2e-120) for products, where 2^e bounds every coordinate of a kept pair): limbs are
// integers below 2^40 held in doubles, integer sums below 2^53 are exact in any order, and carries are propagated between
// reduction levels (thread -> workgroup -> launch -> ranks).  The sums every rank / kernel / query order ends up with are
// the same bits; bits of a term below the unit (coordinates 2^36 times smaller than the scene) are cut off per term, which
// is deterministic too.
//   layout of a NORMALISED row (KB_NL doubles): coordinate sum c (0..5: sum p, sum q) -> 3c + {limb 0, limb 1, carry};
//   product k (0..8, row-major q_r p_c) -> 18 + 4k + {limb 0, limb 1, limb 2, carry}; count -> 54
constexpr int KB_W = 40;
constexpr int KB_NL = 55;
constexpr double KB_2W = 1099511627776.0;              // 2^40
constexpr double KB_2mW = 1.0 / 1099511627776.0;       // 2^-40

PCR_HD double trunc_f64(double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return ::trunc(v);
#else
    return __builtin_trunc(v);
#endif
}

// carry propagation of one sum: n real limbs + the carry limb behind them; afterwards |limb| < 2^40 (signs may differ)
PCR_HD void limbs_normalize(double* L, int n)
{
    for (int j = 0; j < n; j++) {
        const double c = trunc_f64(L[j] * KB_2mW);
        L[j] = L[j] - c * KB_2W;          // exact: both are integers below 2^53
        L[j + 1] = L[j + 1] + c;
    }
}

PCR_HD void limbs_normalize_row(double* row)
{
    for (int c = 0; c < 6; c++) limbs_normalize(row + 3 * c, 2);
    for (int k = 0; k < 9; k++) limbs_normalize(row + 18 + 4 * k, 3);
}

// value of n limbs (+ carry) * 2^unit_exp, evaluated top-down with an error-free two-sum: the same bits on host and device
PCR_HD double limbs_value(const double* L, int n, int unit_exp)
{
    double s = 0.0, c = 0.0, w = 1.0;
    double scale[5];
    for (int j = 0; j <= n; j++) { scale[j] = w; w = w * KB_2W; }
    for (int j = n; j >= 0; j--) {
        const double t = L[j] * scale[j];     // exact: an integer below 2^53 times a power of two
        const double u = s + t;
        const double bb = u - s;
        const double e1 = (s - (u - bb)) + (t - bb);
        s = u;
        c = c + e1;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    return ::ldexp(s + c, unit_exp);
#else
    return __builtin_ldexp(s + c, unit_exp);
#endif
}

// normalised row -> the 16 moments of kabsch_solve (sum p, sum q, sum q p^T row-major, count)
PCR_HD void limbs_to_sums(const double* row, int e, double sums[16])
{
    for (int c = 0; c < 6; c++) sums[c] = limbs_value(row + 3 * c, 2, e - 2 * KB_W);
    for (int k = 0; k < 9; k++) sums[6 + k] = limbs_value(row + 18 + 4 * k, 3, 2 * e - 3 * KB_W);
    sums[15] = row[54];
}

// out = A * B, 4x4 row-major f32, sequential k, unfused (registration.cpp:1002); out may alias A or B
PCR_HD void mat4_mul_f32(const float A[16], const float B[16], float out[16])
{
    float tmp[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            float acc = A[4 * r] * B[c];
            acc = acc + A[4 * r + 1] * B[4 + c];
            acc = acc + A[4 * r + 2] * B[8 + c];
            acc = acc + A[4 * r + 3] * B[12 + c];
            tmp[4 * r + c] = acc;
        }
    for (int k = 0; k < 16; k++) out[k] = tmp[k];
}

}  // namespace num
}  // namespace pcr
