/*
 * pcr_oracle.c — CPU ORACLE (test infrastructure, NOT the product).  See pcr_oracle.h.
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off matters: one fused multiply-add changes d2 in the last bit and flips
 * near-tie winners.
 */
#include "pcr_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ A1 */
/* nanoflann.hpp:403-406: result = 0; for d in 0..2: diff = a[d] - pt(b,d); result += diff*diff */
float orc_d2_f32(float qx, float qy, float qz, float tx, float ty, float tz)
{
    float result = 0.0f;
    float d0 = qx - tx; result += d0 * d0;
    float d1 = qy - ty; result += d1 * d1;
    float d2 = qz - tz; result += d2 * d2;
    return result;
}

/* ------------------------------------------------------------------ A6 (search) */
void orc_nn1_f32(const float* tx, const float* ty, const float* tz, size_t nt,
                 const float* sx, const float* sy, const float* sz, size_t ns,
                 uint32_t* idx, float* d2)
{
    for (size_t i = 0; i < ns; i++) {
        /* nanoflann.hpp:163 worst = FLT_MAX; :1360 accept iff dist < worst; :184 strict > on insert.
         * Ascending j + strict < gives "min d2, lowest index on ties". */
        float best = FLT_MAX;
        uint32_t bi = UINT32_MAX;
        const float qx = sx[i], qy = sy[i], qz = sz[i];
        for (size_t j = 0; j < nt; j++) {
            float d = orc_d2_f32(qx, qy, qz, tx[j], ty[j], tz[j]);
            if (d < best) { best = d; bi = (uint32_t)j; }
        }
        idx[i] = bi;
        d2[i] = (bi == UINT32_MAX) ? INFINITY : best;
    }
}

/* The same search with the queries split over `threads` host threads (contiguous blocks; every query is independent, so
 * the result is identical to orc_nn1_f32).  Used by the full-size (120 k x 120 k) checks when oracle/_ref is absent. */
typedef struct {
    const float *tx, *ty, *tz, *sx, *sy, *sz;
    size_t nt, begin, end;
    uint32_t* idx;
    float* d2;
} orc_nn1_job;

static void* orc_nn1_worker(void* arg)
{
    const orc_nn1_job* j = (const orc_nn1_job*)arg;
    orc_nn1_f32(j->tx, j->ty, j->tz, j->nt, j->sx + j->begin, j->sy + j->begin, j->sz + j->begin, j->end - j->begin,
                j->idx + j->begin, j->d2 + j->begin);
    return NULL;
}

void orc_nn1_f32_mt(const float* tx, const float* ty, const float* tz, size_t nt,
                    const float* sx, const float* sy, const float* sz, size_t ns,
                    uint32_t* idx, float* d2, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if ((size_t)threads > ns) threads = ns ? (int)ns : 1;
    pthread_t th[256];
    orc_nn1_job job[256];
    int live[256];
    for (int t = 0; t < threads; t++) {
        orc_nn1_job jb = { tx, ty, tz, sx, sy, sz, nt, ns * (size_t)t / (size_t)threads, ns * (size_t)(t + 1) / (size_t)threads, idx, d2 };
        job[t] = jb;
        live[t] = 0;
        if (t + 1 < threads) live[t] = pthread_create(&th[t], NULL, orc_nn1_worker, &job[t]) == 0;
        if (!live[t]) orc_nn1_worker(&job[t]);      /* the last block, or no thread available: run it here */
    }
    for (int t = 0; t < threads; t++)
        if (live[t]) pthread_join(th[t], NULL);
}

void orc_nn1_tiecount_f32(const float* tx, const float* ty, const float* tz, size_t nt,
                          const float* sx, const float* sy, const float* sz, size_t ns,
                          uint32_t* tie_count)
{
    for (size_t i = 0; i < ns; i++) {
        float best = FLT_MAX;
        uint32_t cnt = 0;
        for (size_t j = 0; j < nt; j++) {
            float d = orc_d2_f32(sx[i], sy[i], sz[i], tx[j], ty[j], tz[j]);
            if (d < best) { best = d; cnt = 1; }
            else if (d == best) cnt++;
        }
        tie_count[i] = cnt;
    }
}

/* ------------------------------------------------------------------ A2 */
double orc_dist_f64(const double* t, const double* q, int dim)
{
    /* kdtree.hpp:341-346: diff = 0; diff += pow(db[idx][i] - query[i], 2); diff = sqrt(diff).
     * glibc pow(x, 2) == x*x exactly (correctly rounded square). */
    double diff = 0.0;
    for (int i = 0; i < dim; i++) {
        double e = t[i] - q[i];
        diff += e * e;
    }
    return sqrt(diff);
}

/* canonical (distance, index) insertion: ascending distance, ties by ascending index */
static void canon_insert(double* dist, int32_t* idx, int k, int* count, double d, int32_t i)
{
    int n = *count;
    if (n == k) {
        /* full: reject unless strictly better than the last in canonical order */
        if (!(d < dist[k - 1] || (d == dist[k - 1] && i < idx[k - 1]))) return;
        n = k - 1;
    }
    int pos = n;
    while (pos > 0 && (dist[pos - 1] > d || (dist[pos - 1] == d && idx[pos - 1] > i))) {
        dist[pos] = dist[pos - 1];
        idx[pos] = idx[pos - 1];
        pos--;
    }
    dist[pos] = d;
    idx[pos] = i;
    *count = n + 1;
}

void orc_knn_f64(const double* db, size_t n, int dim, const double* q, size_t m, int k,
                 int32_t* idx, double* dist)
{
    for (size_t qi = 0; qi < m; qi++) {
        double* dd = dist + qi * (size_t)k;
        int32_t* ii = idx + qi * (size_t)k;
        int count = 0;
        for (int s = 0; s < k; s++) { dd[s] = 1e10; ii[s] = 0; }   /* resultSet.hpp:35-42 */
        for (size_t j = 0; j < n; j++) {
            double d = orc_dist_f64(db + j * (size_t)dim, q + qi * (size_t)dim, dim);
            if (d > 1e10) continue;                                   /* resultSet.hpp:69 with worst=1e10 */
            canon_insert(dd, ii, k, &count, d, (int32_t)j);
        }
        for (int s = count; s < k; s++) { dd[s] = 1e10; ii[s] = 0; }
    }
}

/* ------------------------------------------------------------------ A11 */
void orc_radius_f64(const double* db, size_t n, int dim, const double* q, size_t m, double r,
                    int64_t* row_ptr, int32_t* idx, double* dist)
{
    int64_t w = 0;
    for (size_t qi = 0; qi < m; qi++) {
        row_ptr[qi] = w;
        for (size_t j = 0; j < n; j++) {
            double d = orc_dist_f64(db + j * (size_t)dim, q + qi * (size_t)dim, dim);
            if (d <= r) {                                             /* resultSet.hpp:133 */
                if (idx) idx[w] = (int32_t)j;
                if (dist) dist[w] = d;
                w++;
            }
        }
    }
    row_ptr[m] = w;
}

/* counts only, on host threads: counts[qi] = |{ j : dist(db_j, q_qi) <= r }| — the same comparison, pair by pair, as orc_radius_f64
 * (the full-size test compares EVERY row count of the 120 000-point scan: 1.44e10 pairs) */
typedef struct { const double* db; size_t n; int dim; const double* q; size_t b, e; double r; int64_t* counts; } orc_rcount_job;
static void* orc_rcount_worker(void* p)
{
    const orc_rcount_job* jb = (const orc_rcount_job*)p;
    for (size_t qi = jb->b; qi < jb->e; qi++) {
        int64_t c = 0;
        for (size_t j = 0; j < jb->n; j++)
            c += orc_dist_f64(jb->db + j * (size_t)jb->dim, jb->q + qi * (size_t)jb->dim, jb->dim) <= jb->r;   /* resultSet.hpp:133 */
        jb->counts[qi] = c;
    }
    return NULL;
}

void orc_radius_count_f64_mt(const double* db, size_t n, int dim, const double* q, size_t m, double r, int64_t* counts, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if ((size_t)threads > m) threads = m ? (int)m : 1;
    pthread_t th[256];
    orc_rcount_job job[256];
    int live[256];
    for (int t = 0; t < threads; t++) {
        orc_rcount_job jb = { db, n, dim, q, m * (size_t)t / (size_t)threads, m * (size_t)(t + 1) / (size_t)threads, r, counts };
        job[t] = jb;
        live[t] = 0;
        if (t + 1 < threads) live[t] = pthread_create(&th[t], NULL, orc_rcount_worker, &job[t]) == 0;
        if (!live[t]) orc_rcount_worker(&job[t]);
    }
    for (int t = 0; t < threads; t++)
        if (live[t]) pthread_join(th[t], NULL);
}

void orc_radius_f32(const float* db, size_t n, int dim, const float* q, size_t m, float r,
                    int64_t* row_ptr, int32_t* idx, float* dist)
{
    /* Homework7/hw7/src/kdtree.cpp: same loop with ElemType float (include/kdtree.hpp:13) */
    int64_t w = 0;
    for (size_t qi = 0; qi < m; qi++) {
        row_ptr[qi] = w;
        for (size_t j = 0; j < n; j++) {
            float diff = 0.0f;
            for (int c = 0; c < dim; c++) {
                float e = db[j * (size_t)dim + c] - q[qi * (size_t)dim + c];
                diff = (float)((double)diff + (double)e * (double)e);   /* `diff += pow(e, 2)`: double pow, float += */
            }
            diff = sqrtf(diff);
            if (diff <= r) {
                if (idx) idx[w] = (int32_t)j;
                if (dist) dist[w] = diff;
                w++;
            }
        }
    }
    row_ptr[m] = w;
}

/* ------------------------------------------------------------------ A4 */
void orc_hw2_knn_add(double* dist, int* index, int capacity, int* count, double* worst,
                     double d, int i)
{
    /* resultSet.hpp:65-91 */
    if (d > *worst) return;
    if (*count < capacity) (*count)++;
    int pos = *count - 1;
    while (pos > 0) {
        if (dist[pos - 1] > d) {
            dist[pos] = dist[pos - 1];
            index[pos] = index[pos - 1];
            pos--;
        } else break;
    }
    dist[pos] = d;
    index[pos] = i;
    *worst = dist[capacity - 1];
}

/* ------------------------------------------------------------------ A3 */
void orc_nano_knn_add(float* dists, size_t* indices, size_t capacity, size_t* count,
                      float d, size_t index)
{
    /* leaf gate nanoflann.hpp:1360 (worst re-read here; nanoflann caches it per leaf :1354,
     * which only delays the rejection of candidates the insertion loop would drop anyway) */
    if (!(d < dists[capacity - 1])) return;
    size_t i;
    for (i = *count; i > 0; --i) {                                   /* :182-192 */
        if (dists[i - 1] > d) {
            if (i < capacity) { dists[i] = dists[i - 1]; indices[i] = indices[i - 1]; }
        } else break;
    }
    if (i < capacity) { dists[i] = d; indices[i] = index; }
    if (*count < capacity) (*count)++;
}

/* ------------------------------------------------------------------ A8 */
void orc_transform_f32(float* x, float* y, float* z, size_t n, const float R[9], const float t[3])
{
    /* registration.cpp:173 `point = R * point + t` — Eigen evaluates the 3x3 * 3x1 product as a
     * lazy coefficient-wise inner product, left to right, then adds t. */
    for (size_t i = 0; i < n; i++) {
        float px = x[i], py = y[i], pz = z[i];
        float nx = ((R[0] * px + R[1] * py) + R[2] * pz) + t[0];
        float ny = ((R[3] * px + R[4] * py) + R[5] * pz) + t[1];
        float nz = ((R[6] * px + R[7] * py) + R[8] * pz) + t[2];
        x[i] = nx; y[i] = ny; z[i] = nz;
    }
}

/* ------------------------------------------------------------------ A7 */
int64_t orc_kabsch_accumulate(const float* sx, const float* sy, const float* sz, size_t ns,
                              const float* tx, const float* ty, const float* tz,
                              const uint32_t* idx, const float* d2, float max_corr,
                              double sums[16])
{
    for (int k = 0; k < 16; k++) sums[k] = 0.0;
    int64_t last = -1;
    for (size_t i = 0; i < ns; i++) {
        if (!(d2[i] < max_corr)) continue;                            /* registration.cpp:936 */
        uint32_t j = idx[i];
        double p[3] = { sx[i], sy[i], sz[i] };
        double q[3] = { tx[j], ty[j], tz[j] };
        for (int c = 0; c < 3; c++) { sums[c] += p[c]; sums[3 + c] += q[c]; }
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) sums[6 + 3 * r + c] += q[r] * p[c];
        sums[15] += 1.0;
        last = (int64_t)i;
    }
    return last;
}

static void mat3_mul(const double A[9], const double B[9], double C[9])
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}

static double det3(const double M[9])
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6])
         + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

void orc_svd3(const double A[9], double U[9], double S[3], double V[9])
{
    /* One-sided (Hestenes) Jacobi: rotate column pairs of W = A*V until mutually orthogonal. */
    double W[9], Vm[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    memcpy(W, A, sizeof W);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < 2; p++) {
            for (int q = p + 1; q < 3; q++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int r = 0; r < 3; r++) {
                    alpha += W[3 * r + p] * W[3 * r + p];
                    beta += W[3 * r + q] * W[3 * r + q];
                    gamma += W[3 * r + p] * W[3 * r + q];
                }
                if (gamma == 0.0) continue;
                double lim = sqrt(alpha * beta);
                if (fabs(gamma) <= DBL_EPSILON * lim) continue;
                off = fmax(off, fabs(gamma) / (lim > 0 ? lim : 1.0));
                double zeta = (beta - alpha) / (2.0 * gamma);
                double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
                for (int r = 0; r < 3; r++) {
                    double wp = W[3 * r + p], wq = W[3 * r + q];
                    W[3 * r + p] = c * wp - s * wq;
                    W[3 * r + q] = s * wp + c * wq;
                    double vp = Vm[3 * r + p], vq = Vm[3 * r + q];
                    Vm[3 * r + p] = c * vp - s * vq;
                    Vm[3 * r + q] = s * vp + c * vq;
                }
            }
        }
        if (off == 0.0) break;
    }
    double sv[3];
    for (int c = 0; c < 3; c++)
        sv[c] = sqrt(W[c] * W[c] + W[3 + c] * W[3 + c] + W[6 + c] * W[6 + c]);
    /* sort descending (Eigen::JacobiSVD convention) */
    int ord[3] = { 0, 1, 2 };
    for (int a = 0; a < 2; a++)
        for (int b = a + 1; b < 3; b++)
            if (sv[ord[b]] > sv[ord[a]]) { int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    double smax = sv[ord[0]];
    for (int c = 0; c < 3; c++) {
        int o = ord[c];
        S[c] = sv[o];
        for (int r = 0; r < 3; r++) V[3 * r + c] = Vm[3 * r + o];
        if (sv[o] > 0.0 && sv[o] > smax * DBL_EPSILON * 8.0) {
            for (int r = 0; r < 3; r++) U[3 * r + c] = W[3 * r + o] / sv[o];
        } else {
            for (int r = 0; r < 3; r++) U[3 * r + c] = 0.0;   /* completed below */
        }
    }
    /* complete U for (numerically) rank-deficient A: keep it orthonormal */
    int defined[3];
    for (int c = 0; c < 3; c++)
        defined[c] = (U[c] != 0.0 || U[3 + c] != 0.0 || U[6 + c] != 0.0);
    if (!defined[0]) {             /* A == 0 */
        double I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
        memcpy(U, I, sizeof I);
        return;
    }
    if (!defined[1]) {
        /* any unit vector orthogonal to u0 */
        double u0[3] = { U[0], U[3], U[6] };
        int k = 0;
        if (fabs(u0[1]) < fabs(u0[k])) k = 1;
        if (fabs(u0[2]) < fabs(u0[k])) k = 2;
        double e[3] = { 0, 0, 0 };
        e[k] = 1.0;
        double dot = u0[k];
        double v[3] = { e[0] - dot * u0[0], e[1] - dot * u0[1], e[2] - dot * u0[2] };
        double nv = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        U[1] = v[0] / nv; U[4] = v[1] / nv; U[7] = v[2] / nv;
        defined[1] = 1;
    }
    if (!defined[2]) {
        /* u2 = u0 x u1 */
        U[2] = U[3] * U[7] - U[6] * U[4];
        U[5] = U[6] * U[1] - U[0] * U[7];
        U[8] = U[0] * U[4] - U[3] * U[1];
    }
}

int orc_kabsch_solve(const double sums[16], float R[9], float t[3])
{
    double M = sums[15];
    if (!(M > 0.0)) return -1;
    double pbar[3], qbar[3];
    for (int c = 0; c < 3; c++) { pbar[c] = sums[c] / M; qbar[c] = sums[3 + c] / M; }   /* :979-980 */
    double H[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            H[3 * r + c] = sums[6 + 3 * r + c] - M * qbar[r] * pbar[c];                   /* :982-985 */
    double U[9], S[3], V[9], Vt[9], Rd[9];
    orc_svd3(H, U, S, V);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Vt[3 * r + c] = V[3 * c + r];
    mat3_mul(U, Vt, Rd);                                                                 /* :988 */
    double det = det3(Rd);
    if (det < 0) {                                                                       /* :990-996 */
        double Ut[9], VB[9];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Ut[3 * r + c] = U[3 * c + r];
        for (int r = 0; r < 3; r++) {
            VB[3 * r] = V[3 * r]; VB[3 * r + 1] = V[3 * r + 1]; VB[3 * r + 2] = V[3 * r + 2] * det;
        }
        mat3_mul(VB, Ut, Rd);                                                            /* V*B*U^T (sic) */
    }
    for (int k = 0; k < 9; k++) R[k] = (float)Rd[k];
    for (int r = 0; r < 3; r++) {                                                        /* :998 */
        double Rp = ((double)R[3 * r] * pbar[0] + (double)R[3 * r + 1] * pbar[1]) + (double)R[3 * r + 2] * pbar[2];
        t[r] = (float)(qbar[r] - Rp);
    }
    return 0;
}

/* The Kabsch block of Registration::RANSAC, registration.cpp:372-392, statement by statement.  It is NOT the ICP block: here
 *   R_ = U * V^T;  t_ = target_center - R_ * source_center;      (:382-383)
 * come first, and the reflection repair `if (det R_ < 0) R_ = V * B * U^T` (:386-392) replaces R_ only — t_ is never
 * recomputed, so a reflected quad carries the translation of the un-repaired rotation into its consensus count (:404). */
int orc_kabsch_solve_ransac(const double sums[16], float R[9], float t[3])
{
    const double M = sums[15];
    if (!(M > 0.0)) return -1;
    double sc[3], tc[3], H[9];
    for (int c = 0; c < 3; c++) { sc[c] = sums[c] / M; tc[c] = sums[3 + c] / M; }        /* :373-374 rowwise().mean() */
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) H[3 * r + c] = sums[6 + 3 * r + c] - M * tc[r] * sc[c];  /* :375-379 target_c * source_c^T */
    double U[9], S[3], V[9], Vt[9], R0[9];
    orc_svd3(H, U, S, V);                                                                /* :379-381 */
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Vt[3 * r + c] = V[3 * c + r];
    mat3_mul(U, Vt, R0);                                                                 /* :382 */
    float Rf[9];
    for (int k = 0; k < 9; k++) Rf[k] = (float)R0[k];
    for (int r = 0; r < 3; r++) {                                                        /* :383 — with the UN-repaired R_ */
        const double Rs = ((double)Rf[3 * r] * sc[0] + (double)Rf[3 * r + 1] * sc[1]) + (double)Rf[3 * r + 2] * sc[2];
        t[r] = (float)(tc[r] - Rs);
    }
    const double det = det3(R0);                                                         /* :386 */
    if (det < 0) {                                                                       /* :387-392 */
        double Ut[9], VB[9], R1[9];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Ut[3 * r + c] = U[3 * c + r];
        for (int r = 0; r < 3; r++) { VB[3 * r] = V[3 * r]; VB[3 * r + 1] = V[3 * r + 1]; VB[3 * r + 2] = V[3 * r + 2] * det; }
        mat3_mul(VB, Ut, R1);
        for (int k = 0; k < 9; k++) Rf[k] = (float)R1[k];
    }
    memcpy(R, Rf, sizeof Rf);
    return 0;
}

void orc_mat4_mul_f32(const float A[16], const float B[16], float out[16])
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
    memcpy(out, tmp, sizeof tmp);
}

/* ------------------------------------------------------------------ A9 */
void orc_icp_p2p_f32(const float* sx, const float* sy, const float* sz, size_t ns,
                     const float* tx, const float* ty, const float* tz, size_t nt,
                     const float init_T[16], const orc_icp_params* prm,
                     float out_T[16], orc_icp_stats* stats,
                     float* per_iter_T, uint64_t* per_iter_pairs)
{
    float* px = (float*)malloc(sizeof(float) * (ns ? ns : 1));
    float* py = (float*)malloc(sizeof(float) * (ns ? ns : 1));
    float* pz = (float*)malloc(sizeof(float) * (ns ? ns : 1));
    uint32_t* idx = (uint32_t*)malloc(sizeof(uint32_t) * (ns ? ns : 1));
    float* d2 = (float*)malloc(sizeof(float) * (ns ? ns : 1));
    memcpy(px, sx, sizeof(float) * ns);
    memcpy(py, sy, sizeof(float) * ns);
    memcpy(pz, sz, sizeof(float) * ns);

    float R0[9] = { init_T[0], init_T[1], init_T[2], init_T[4], init_T[5], init_T[6],
                    init_T[8], init_T[9], init_T[10] };
    float t0[3] = { init_T[3], init_T[7], init_T[11] };
    orc_transform_f32(px, py, pz, ns, R0, t0);                        /* :872-874 */

    float T_total[16] = { R0[0], R0[1], R0[2], t0[0], R0[3], R0[4], R0[5], t0[1],
                          R0[6], R0[7], R0[8], t0[2], 0, 0, 0, 1 };   /* :910-913 */
    orc_icp_stats st;
    memset(&st, 0, sizeof st);
    float last_loss = 0.0f;                                           /* :915 */
    uint64_t unchanged = 0;                                           /* :916 */
    for (uint64_t iter = 0; iter < prm->max_iter; iter++) {           /* :917 */
        orc_nn1_f32(tx, ty, tz, nt, px, py, pz, ns, idx, d2);         /* :925-934 */
        double sums[16];
        int64_t last = orc_kabsch_accumulate(px, py, pz, ns, tx, ty, tz, idx, d2,
                                             prm->max_corr, sums);   /* :936-940, :964-985 */
        float loss = 0.0f;
        if (last >= 0) loss = d2[last] * d2[last];                    /* :939 (overwritten, not summed) */
        st.last_pairs = (uint64_t)sums[15];
        st.last_loss = loss;
        if (per_iter_pairs) per_iter_pairs[iter] = st.last_pairs;
        if (fabsf(last_loss - loss) < prm->eps) unchanged++;          /* :948-951 (never reset) */
        if (unchanged > 15) { st.converged = 1; break; }              /* :954-958 */
        last_loss = loss;                                             /* :961 */
        float Rd[9], td[3];
        if (orc_kabsch_solve(sums, Rd, td) != 0) { st.empty_pairs = 1; break; }
        float T_delta[16] = { Rd[0], Rd[1], Rd[2], td[0], Rd[3], Rd[4], Rd[5], td[1],
                              Rd[6], Rd[7], Rd[8], td[2], 0, 0, 0, 1 };
        if (per_iter_T) memcpy(per_iter_T + 16 * iter, T_delta, sizeof T_delta);
        orc_mat4_mul_f32(T_delta, T_total, T_total);                  /* :1000-1002 */
        orc_transform_f32(px, py, pz, ns, Rd, td);                    /* :1003 */
        st.iters_run++;
    }
    memcpy(out_T, T_total, sizeof T_total);                           /* :1008-1009 */
    if (stats) *stats = st;
    free(px); free(py); free(pz); free(idx); free(d2);
}

/* ------------------------------------------------------------------ A10 */
static inline double plane_dist(double x, double y, double z, const double* p)
{
    /* np.c_[X, 1].dot(params): row . params, k = 0..3 in order, f64 */
    return fabs(((x * p[0] + y * p[1]) + z * p[2]) + 1.0 * p[3]);
}

void orc_plane_count_f32pts(const float* x, const float* y, const float* z, size_t n,
                            const double* planes4, size_t n_planes, double thr, int64_t* counts)
{
    for (size_t h = 0; h < n_planes; h++) {
        int64_t c = 0;
        for (size_t i = 0; i < n; i++)
            c += plane_dist((double)x[i], (double)y[i], (double)z[i], planes4 + 4 * h) < thr;
        counts[h] = c;
    }
}

void orc_plane_count_f64pts(const double* x, const double* y, const double* z, size_t n,
                            const double* planes4, size_t n_planes, double thr, int64_t* counts)
{
    for (size_t h = 0; h < n_planes; h++) {
        int64_t c = 0;
        for (size_t i = 0; i < n; i++)
            c += plane_dist(x[i], y[i], z[i], planes4 + 4 * h) < thr;
        counts[h] = c;
    }
}

void orc_plane_mask_f32pts(const float* x, const float* y, const float* z, size_t n,
                           const double plane4[4], double thr, uint8_t* mask)
{
    for (size_t i = 0; i < n; i++)
        mask[i] = plane_dist((double)x[i], (double)y[i], (double)z[i], plane4) < thr;
}

void orc_plane_from_3pts(const double p[9], double params[4])
{
    /* ground_detection_ransac.py:158-169 */
    double v1[3] = { p[3] - p[0], p[4] - p[1], p[5] - p[2] };
    double v2[3] = { p[6] - p[0], p[7] - p[1], p[8] - p[2] };
    double a = (v1[1] * v2[2]) - (v1[2] * v2[1]);
    double b = (v1[2] * v2[0]) - (v1[0] * v2[2]);
    double c = (v1[0] * v2[1]) - (v1[1] * v2[0]);
    double d = -(a * p[0] + b * p[1] + c * p[2]);
    double n = sqrt(a * a + b * b + c * c);
    params[0] = a / n; params[1] = b / n; params[2] = c / n; params[3] = d / n;
}

/* ------------------------------------------------------------------ N3 */
typedef struct { long long h; size_t i; } orc_hi;

static int orc_hi_cmp(const void* a, const void* b)
{
    const orc_hi* p = (const orc_hi*)a; const orc_hi* q = (const orc_hi*)b;
    if (p->h != q->h) return p->h < q->h ? -1 : 1;
    return p->i < q->i ? -1 : (p->i > q->i ? 1 : 0);      /* list.sort is stable: ascending index inside a voxel */
}

size_t orc_voxel_filter_f32(const float* x, const float* y, const float* z, size_t n, double leaf_size,
                            float* ox, float* oy, float* oz)
{
    if (n == 0) return 0;
    float x_max = x[0], y_max = y[0], x_min = x[0], y_min = y[0], z_min = z[0];      /* voxel_filter.py:22-26 */
    for (size_t i = 1; i < n; i++) {
        if (x[i] > x_max) x_max = x[i];
        if (y[i] > y_max) y_max = y[i];
        if (x[i] < x_min) x_min = x[i];
        if (y[i] < y_min) y_min = y[i];
        if (z[i] < z_min) z_min = z[i];
    }
    const long long Dx = (long long)ceil((double)(float)(x_max - x_min) / leaf_size);   /* :28 */
    const long long Dy = (long long)ceil((double)(float)(y_max - y_min) / leaf_size);   /* :29 */
    orc_hi* hl = (orc_hi*)malloc(sizeof(orc_hi) * n);
    for (size_t i = 0; i < n; i++) {                                                     /* :32-36 */
        const long long hx = (long long)floor((double)(float)(x[i] - x_min) / leaf_size);
        const long long hy = (long long)floor((double)(float)(y[i] - y_min) / leaf_size);
        const long long hz = (long long)floor((double)(float)(z[i] - z_min) / leaf_size);
        hl[i].h = hx + hy * Dx + hz * Dx * Dy;
        hl[i].i = i;
    }
    qsort(hl, n, sizeof(orc_hi), orc_hi_cmp);                                            /* :37 */
    size_t out = 0, start = 0;
    for (size_t k = 1; k <= n; k++) {
        if (k < n && hl[k].h == hl[start].h) continue;
        if (k == n) break;                            /* :41-50: the last group is never flushed */
        float sx = 0.0f, sy = 0.0f, sz = 0.0f;
        for (size_t t = start; t < k; t++) { sx += x[hl[t].i]; sy += y[hl[t].i]; sz += z[hl[t].i]; }
        const float cnt = (float)(k - start);
        ox[out] = sx / cnt; oy[out] = sy / cnt; oz[out] = sz / cnt;
        out++;
        start = k;
    }
    free(hl);
    return out;
}

/* ------------------------------------------------------------------ N1 */
static float hw7_dist_f32(float tx, float ty, float tz, float qx, float qy, float qz)
{
    /* Homework7/hw7/src/kdtree.cpp:310-315: `ElemType diff = 0; diff += pow(db - query, 2);` — pow() is the double
     * overload, the += rounds the double sum back to float */
    float s = 0.0f;
    s = (float)((double)s + (double)(tx - qx) * (double)(tx - qx));
    s = (float)((double)s + (double)(ty - qy) * (double)(ty - qy));
    s = (float)((double)s + (double)(tz - qz) * (double)(tz - qz));
    return sqrtf(s);
}

/* eigenvalues of a symmetric 3x3 (row-major, f64) by cyclic Jacobi, ascending */
static void sym_eig3(const double A[9], double w[3])
{
    double a[3][3] = { { A[0], A[1], A[2] }, { A[3], A[4], A[5] }, { A[6], A[7], A[8] } };
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        if (off == 0.0) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (a[p][q] == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 3; k++) { double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - sn * akq; a[k][q] = sn * akp + c * akq; }
                for (int k = 0; k < 3; k++) { double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - sn * aqk; a[q][k] = sn * apk + c * aqk; }
            }
    }
    w[0] = a[0][0]; w[1] = a[1][1]; w[2] = a[2][2];
    for (int i = 0; i < 2; i++) for (int j = i + 1; j < 3; j++) if (w[j] < w[i]) { double t = w[i]; w[i] = w[j]; w[j] = t; }
}

void orc_iss_f32(const float* x, const float* y, const float* z, size_t n, float local_radius, float non_max_radius,
                 float gamma21, float gamma32, int min_neighbors, int weighted, uint8_t* is_key, float* lambda3_out)
{
    uint32_t* cnt = (uint32_t*)calloc(n ? n : 1, sizeof(uint32_t));
    float* l3 = (float*)malloc(sizeof(float) * (n ? n : 1));
    for (size_t i = 0; i < n; i++)                                                 /* iss_detector.cpp:47-57 */
        for (size_t j = 0; j < n; j++)
            cnt[i] += hw7_dist_f32(x[j], y[j], z[j], x[i], y[i], z[i]) <= local_radius;
    for (size_t i = 0; i < n; i++) {                                               /* :69-83 */
        l3[i] = -1.0f;
        if (cnt[i] < 3) continue;
        double cov[9] = { 0 }, wsum = 0.0;
        for (size_t j = 0; j < n; j++) {
            if (!(hw7_dist_f32(x[j], y[j], z[j], x[i], y[i], z[i]) <= local_radius)) continue;
            const double w = weighted ? (double)(1.0f / (float)cnt[j]) : 1.0;       /* :130 weight_nn = 1.f / size */
            const double d[3] = { (double)(float)(x[j] - x[i]), (double)(float)(y[j] - y[i]), (double)(float)(z[j] - z[i]) };
            for (int r = 0; r < 3; r++) for (int c = r; c < 3; c++) cov[3 * r + c] += (w * d[r]) * d[c];   /* upper triangle */
            wsum += w;
        }
        if (weighted) for (int k = 0; k < 9; k++) cov[k] /= wsum;                  /* :137 */
        cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];                         /* symmetric by construction */
        double w3[3];
        sym_eig3(cov, w3);                                                          /* :148-151, ascending */
        const float lambda1 = (float)w3[2], lambda2 = (float)w3[1], lambda3 = (float)w3[0];
        if (lambda2 / lambda1 < gamma21 && lambda3 / lambda2 < gamma32 && lambda3 > 0) l3[i] = lambda3;   /* :79 */
    }
    for (size_t i = 0; i < n; i++) {                                               /* :86-105 */
        is_key[i] = 0;
        if (l3[i] == -1.0f) continue;
        size_t m = 0;
        int is_max = 1;
        for (size_t j = 0; j < n; j++) {
            if (!(hw7_dist_f32(x[j], y[j], z[j], x[i], y[i], z[i]) <= non_max_radius)) continue;
            m++;
            if (l3[i] < l3[j]) is_max = 0;
        }
        if ((int)m < min_neighbors) continue;
        is_key[i] = (uint8_t)is_max;
    }
    if (lambda3_out) memcpy(lambda3_out, l3, sizeof(float) * n);
    free(cnt); free(l3);
}

/* ------------------------------------------------------------------ N4 */
float orc_d2_dim_f32(const float* a, const float* b, int dim)
{
    float result = 0.0f;
    int d = 0;
    for (; d + 3 < dim; d += 4) {                                     /* nanoflann.hpp:390-400 */
        const float d0 = a[d] - b[d], d1 = a[d + 1] - b[d + 1], d2 = a[d + 2] - b[d + 2], d3 = a[d + 3] - b[d + 3];
        result += ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
    }
    for (; d < dim; d++) {                                            /* :402-405 */
        const float d0 = a[d] - b[d];
        result += d0 * d0;
    }
    return result;
}

void orc_nn1_dim_f32(const float* db, size_t n, const float* q, size_t m, int dim, uint32_t* idx, float* d2)
{
    for (size_t i = 0; i < m; i++) {
        float best = FLT_MAX;                                         /* nanoflann.hpp:163 */
        uint32_t bi = UINT32_MAX;
        for (size_t j = 0; j < n; j++) {
            const float d = orc_d2_dim_f32(q + i * (size_t)dim, db + j * (size_t)dim, dim);
            if (d < best) { best = d; bi = (uint32_t)j; }             /* strict: lowest index among equals */
        }
        idx[i] = bi;
        d2[i] = bi == UINT32_MAX ? INFINITY : best;
    }
}

typedef struct { uint32_t s, t; float d; size_t pos; } match_rec;
static int match_cmp(const void* a, const void* b)
{
    const match_rec* x = (const match_rec*)a; const match_rec* y = (const match_rec*)b;
    if (x->d < y->d) return -1;
    if (y->d < x->d) return 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos ? 1 : 0);
}

size_t orc_match_union_f32(const float* desc_src, size_t n_src, const float* desc_tgt, size_t n_tgt, int dim,
                           float rejection_rate, uint32_t* pairs, float* dist)
{
    const size_t total = n_src + n_tgt;
    match_rec* rec = (match_rec*)malloc(sizeof(match_rec) * (total ? total : 1));
    uint32_t* idx = (uint32_t*)malloc(sizeof(uint32_t) * (total ? total : 1));
    float* d2 = (float*)malloc(sizeof(float) * (total ? total : 1));
    orc_nn1_dim_f32(desc_src, n_src, desc_tgt, n_tgt, dim, idx, d2);               /* :561-577 */
    for (size_t i = 0; i < n_tgt; i++) { rec[i].s = idx[i]; rec[i].t = (uint32_t)i; rec[i].d = d2[i]; rec[i].pos = i; }
    orc_nn1_dim_f32(desc_tgt, n_tgt, desc_src, n_src, dim, idx, d2);               /* :579-595 */
    for (size_t i = 0; i < n_src; i++) { rec[n_tgt + i].s = (uint32_t)i; rec[n_tgt + i].t = idx[i]; rec[n_tgt + i].d = d2[i]; rec[n_tgt + i].pos = n_tgt + i; }
    qsort(rec, total, sizeof(match_rec), match_cmp);                               /* :598-603 */
    const float keep_f = floorf((1 - rejection_rate) * (float)total);              /* :605, float arithmetic */
    size_t keep = keep_f > 0 ? (size_t)keep_f : 0;
    if (keep > total) keep = total;
    for (size_t i = 0; i < keep; i++) { pairs[2 * i] = rec[i].s; pairs[2 * i + 1] = rec[i].t; dist[i] = rec[i].d; }
    free(rec); free(idx); free(d2);
    return keep;
}

size_t orc_match_inter_f32(const float* desc_src, size_t n_src, const float* desc_tgt, size_t n_tgt, int dim,
                           float rejection_rate, uint32_t* pairs, float* dist)
{
    uint32_t* t2s = (uint32_t*)malloc(sizeof(uint32_t) * (n_tgt ? n_tgt : 1));
    uint32_t* s2t = (uint32_t*)malloc(sizeof(uint32_t) * (n_src ? n_src : 1));
    float* dt = (float*)malloc(sizeof(float) * (n_tgt ? n_tgt : 1));
    float* ds = (float*)malloc(sizeof(float) * (n_src ? n_src : 1));
    match_rec* rec = (match_rec*)malloc(sizeof(match_rec) * (n_src ? n_src : 1));
    orc_nn1_dim_f32(desc_src, n_src, desc_tgt, n_tgt, dim, t2s, dt);               /* :455-473 */
    orc_nn1_dim_f32(desc_tgt, n_tgt, desc_src, n_src, dim, s2t, ds);               /* :475-494 */
    size_t m = 0;
    for (size_t s = 0; s < n_src; s++) {                                           /* :497-508 */
        const uint32_t t = s2t[s];
        if (t == UINT32_MAX) continue;
        if (t2s[t] == (uint32_t)s) { rec[m].s = (uint32_t)s; rec[m].t = t; rec[m].d = ds[s]; rec[m].pos = m; m++; }
    }
    qsort(rec, m, sizeof(match_rec), match_cmp);                                   /* :519-521 */
    const float keep_f = floorf((1 - rejection_rate) * (float)m);                  /* :523 */
    size_t keep = keep_f > 0 ? (size_t)keep_f : 0;
    if (keep > m) keep = m;
    for (size_t i = 0; i < keep; i++) { pairs[2 * i] = rec[i].s; pairs[2 * i + 1] = rec[i].t; dist[i] = rec[i].d; }
    free(t2s); free(s2t); free(dt); free(ds); free(rec);
    return keep;
}

int orc_ransac_hypothesis(const float* src_xyz, const float* tgt_xyz, const uint32_t* pairs, const uint32_t quad[4],
                          float R[9], float t[3])
{
    double sums[16];
    for (int k = 0; k < 16; k++) sums[k] = 0.0;
    for (int k = 0; k < 4; k++) {                                                  /* :354-366 */
        const float* ps = src_xyz + 3 * (size_t)pairs[2 * (size_t)quad[k]];
        const float* qt = tgt_xyz + 3 * (size_t)pairs[2 * (size_t)quad[k] + 1];
        const double p[3] = { ps[0], ps[1], ps[2] }, q[3] = { qt[0], qt[1], qt[2] };
        for (int c = 0; c < 3; c++) { sums[c] += p[c]; sums[3 + c] += q[c]; }
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) sums[6 + 3 * r + c] += q[r] * p[c];
        sums[15] += 1.0;
    }
    return orc_kabsch_solve_ransac(sums, R, t);                                    /* :372-392 */
}

uint32_t orc_consensus_count_f32(const float* src_xyz, const float* tgt_xyz, const uint32_t* pairs, size_t n_pairs,
                                 const float R[9], const float t[3], float thr)
{
    uint32_t c = 0;
    for (size_t i = 0; i < n_pairs; i++) {                                         /* :395-421 */
        const float* s = src_xyz + 3 * (size_t)pairs[2 * i];
        const float* q = tgt_xyz + 3 * (size_t)pairs[2 * i + 1];
        float e[3];
        for (int r = 0; r < 3; r++) e[r] = q[r] - (((R[3 * r] * s[0] + R[3 * r + 1] * s[1]) + R[3 * r + 2] * s[2]) + t[r]);
        const float distance = sqrtf((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
        c += distance <= thr;
    }
    return c;
}

int64_t orc_ransac_global_f32(const float* src_xyz, const float* tgt_xyz, const uint32_t* pairs, size_t n_pairs,
                              const uint32_t* quads, size_t n_hyp, float thr, float R[9], float t[3],
                              uint32_t* best_count, uint32_t* counts)
{
    uint32_t max_consensus = 0;
    int64_t winner = -1;
    for (size_t h = 0; h < n_hyp; h++) {
        float Rh[9], th[3];
        uint32_t c = 0;
        if (orc_ransac_hypothesis(src_xyz, tgt_xyz, pairs, quads + 4 * h, Rh, th) == 0)
            c = orc_consensus_count_f32(src_xyz, tgt_xyz, pairs, n_pairs, Rh, th, thr);
        if (counts) counts[h] = c;
        if (c > max_consensus) {                                                   /* :423-428 */
            max_consensus = c;
            winner = (int64_t)h;
            memcpy(R, Rh, sizeof Rh);
            memcpy(t, th, sizeof th);
        }
    }
    if (best_count) *best_count = max_consensus;
    return winner;
}

/* ------------------------------------------------------------------ N2 */
typedef struct { double v[3]; } vec3d;
static vec3d v3(double a, double b, double c) { vec3d r = { { a, b, c } }; return r; }
static vec3d v3_cross(vec3d a, vec3d b)
{
    return v3(a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2], a.v[0] * b.v[1] - a.v[1] * b.v[0]);
}
static double v3_dot(vec3d a, vec3d b) { return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2]; }
static vec3d v3_scale(vec3d a, double s) { return v3(a.v[0] * s, a.v[1] * s, a.v[2] * s); }
static vec3d v3_div(vec3d a, double s) { return v3(a.v[0] / s, a.v[1] / s, a.v[2] / s); }
static vec3d v3_sub(vec3d a, vec3d b) { return v3(a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]); }

/* mylib.cpp:9-38: null vector of (A - eval I) as the largest of the three row cross products */
static vec3d fe_evec0(const double A[9], double eval)
{
    const vec3d r0 = v3(A[0] - eval, A[1], A[2]), r1 = v3(A[1], A[4] - eval, A[5]), r2 = v3(A[2], A[5], A[8] - eval);
    const vec3d c01 = v3_cross(r0, r1), c02 = v3_cross(r0, r2), c12 = v3_cross(r1, r2);
    const double d0 = v3_dot(c01, c01), d1 = v3_dot(c02, c02), d2 = v3_dot(c12, c12);
    double dmax = d0;
    int imax = 0;
    if (d1 > dmax) { dmax = d1; imax = 1; }
    if (d2 > dmax) imax = 2;
    if (imax == 0) return v3_div(c01, sqrt(d0));
    if (imax == 1) return v3_div(c02, sqrt(d1));
    return v3_div(c12, sqrt(d2));
}

/* mylib.cpp:40-102: second eigenvector inside the plane orthogonal to evec0 (2x2 problem in the basis U, V) */
static vec3d fe_evec1(const double A[9], vec3d e0, double eval1)
{
    vec3d U;
    if (fabs(e0.v[0]) > fabs(e0.v[1])) {
        const double inv = 1 / sqrt(e0.v[0] * e0.v[0] + e0.v[2] * e0.v[2]);
        U = v3(-e0.v[2] * inv, 0, e0.v[0] * inv);
    } else {
        const double inv = 1 / sqrt(e0.v[1] * e0.v[1] + e0.v[2] * e0.v[2]);
        U = v3(0, e0.v[2] * inv, -e0.v[1] * inv);
    }
    const vec3d V = v3_cross(e0, U);
    const vec3d AU = v3(A[0] * U.v[0] + A[1] * U.v[1] + A[2] * U.v[2], A[1] * U.v[0] + A[4] * U.v[1] + A[5] * U.v[2],
                        A[2] * U.v[0] + A[5] * U.v[1] + A[8] * U.v[2]);
    const vec3d AV = v3(A[0] * V.v[0] + A[1] * V.v[1] + A[2] * V.v[2], A[1] * V.v[0] + A[4] * V.v[1] + A[5] * V.v[2],
                        A[2] * V.v[0] + A[5] * V.v[1] + A[8] * V.v[2]);
    double m00 = U.v[0] * AU.v[0] + U.v[1] * AU.v[1] + U.v[2] * AU.v[2] - eval1;
    double m01 = U.v[0] * AV.v[0] + U.v[1] * AV.v[1] + U.v[2] * AV.v[2];
    double m11 = V.v[0] * AV.v[0] + V.v[1] * AV.v[1] + V.v[2] * AV.v[2] - eval1;
    const double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
    if (a00 >= a11) {
        if (!(fmax(a00, a01) > 0)) return U;
        if (a00 >= a01) { m01 /= m00; m00 = 1 / sqrt(1 + m01 * m01); m01 *= m00; }
        else { m00 /= m01; m01 = 1 / sqrt(1 + m00 * m00); m00 *= m01; }
        return v3_sub(v3_scale(U, m01), v3_scale(V, m00));
    }
    if (!(fmax(a11, a01) > 0)) return U;
    if (a11 >= a01) { m01 /= m11; m11 = 1 / sqrt(1 + m01 * m01); m01 *= m11; }
    else { m11 /= m01; m01 = 1 / sqrt(1 + m11 * m11); m11 *= m01; }
    return v3_sub(v3_scale(U, m11), v3_scale(V, m01));
}

void orc_fast_eigen3x3(const double Ain[9], double normal[3])
{
    double A[9];
    memcpy(A, Ain, sizeof A);
    double max_coeff = A[0];                                             /* mylib.cpp:112 maxCoeff(): signed maximum */
    for (int k = 1; k < 9; k++) if (A[k] > max_coeff) max_coeff = A[k];
    vec3d out = v3(0, 0, 0);
    if (max_coeff == 0) { memcpy(normal, out.v, sizeof out.v); return; }
    for (int k = 0; k < 9; k++) A[k] /= max_coeff;                        /* :116 */
    const double norm = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    if (norm > 0) {
        const double q = (A[0] + A[4] + A[8]) / 3;
        const double b00 = A[0] - q, b11 = A[4] - q, b22 = A[8] - q;
        const double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2) / 6);
        const double c00 = b11 * b22 - A[5] * A[5], c01 = A[1] * b22 - A[5] * A[2], c02 = A[1] * A[5] - b11 * A[2];
        const double det = (b00 * c00 - A[1] * c01 + A[2] * c02) / (p * p * p);
        double half_det = det * 0.5;
        half_det = fmin(fmax(half_det, -1.0), 1.0);
        const double angle = acos(half_det) / (double)3;
        const double two_thirds_pi = 2.09439510239319549;
        const double beta2 = cos(angle) * 2, beta0 = cos(angle + two_thirds_pi) * 2, beta1 = -(beta0 + beta2);
        const double e0 = q + p * beta0, e1 = q + p * beta1, e2 = q + p * beta2;
        if (half_det >= 0) {                                              /* :152-163 */
            const vec3d v2 = fe_evec0(A, e2);
            if (e2 < e0 && e2 < e1) out = v2;
            else {
                const vec3d v1 = fe_evec1(A, v2, e1);
                out = (e1 < e0 && e1 < e2) ? v1 : v3_cross(v1, v2);
            }
        } else {                                                          /* :164-176 */
            const vec3d v0 = fe_evec0(A, e0);
            if (e0 < e1 && e0 < e2) out = v0;
            else {
                const vec3d v1 = fe_evec1(A, v0, e1);
                out = (e1 < e0 && e1 < e2) ? v1 : v3_cross(v0, v1);
            }
        }
    } else {                                                              /* :177-187 diagonal matrix (scaled back) */
        const double a0 = A[0] * max_coeff, a1 = A[4] * max_coeff, a2 = A[8] * max_coeff;
        if (a0 < a1 && a0 < a2) out = v3(1, 0, 0);
        else if (a1 < a0 && a1 < a2) out = v3(0, 1, 0);
        else out = v3(0, 0, 1);
    }
    memcpy(normal, out.v, sizeof out.v);
}

static int cmp_f32(const void* a, const void* b)
{
    const float x = *(const float*)a, y = *(const float*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

size_t orc_ground_seeds_f64(const float* x, const float* y, const float* z, size_t n, size_t lpr_size,
                            double threshold_seeds, uint8_t* seed_mask, double* upper_bound)
{
    (void)x; (void)y;
    const double z_high = -1.73 + 0.5;                                    /* :47 */
    float* cand = (float*)malloc(sizeof(float) * (n ? n : 1));
    size_t m = 0;
    for (size_t i = 0; i < n; i++) if ((double)z[i] < z_high) cand[m++] = z[i];     /* :50-52 */
    qsort(cand, m, sizeof(float), cmp_f32);
    const size_t k = lpr_size > m ? m : lpr_size;                         /* :54-60 */
    double sum = 0.0;
    for (size_t i = 0; i < k; i++) sum += (double)cand[i];                /* np.mean(axis=0), z column (:62) */
    const double lpr_z = sum / (double)k;                                 /* 0/0 = NaN without candidates */
    const double ub = lpr_z + threshold_seeds;                            /* :65 */
    size_t count = 0;
    for (size_t i = 0; i < n; i++) {
        seed_mask[i] = ((double)z[i] < z_high) && ((double)z[i] < ub);    /* :67-68 */
        count += seed_mask[i];
    }
    if (upper_bound) *upper_bound = ub;
    free(cand);
    return count;
}

size_t orc_estimate_plane_f64(const float* x, const float* y, const float* z, size_t n, const uint8_t* mask, double params[4])
{
    double s[3] = { 0, 0, 0 };
    size_t m = 0;
    for (size_t i = 0; i < n; i++) if (mask[i]) { s[0] += x[i]; s[1] += y[i]; s[2] += z[i]; m++; }
    const double c[3] = { s[0] / (double)m, s[1] / (double)m, s[2] / (double)m };   /* :75 */
    double xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
    for (size_t i = 0; i < n; i++) {
        if (!mask[i]) continue;
        const double dx = x[i] - c[0], dy = y[i] - c[1], dz = z[i] - c[2];           /* :76 */
        xx += dx * dx; xy += dx * dy; xz += dx * dz; yy += dy * dy; yz += dy * dz; zz += dz * dz;   /* :77 */
    }
    const double XTX[9] = { xx, xy, xz, xy, yy, yz, xz, yz, zz };
    double nrm[3];
    orc_fast_eigen3x3(XTX, nrm);                                          /* :83 */
    params[0] = nrm[0]; params[1] = nrm[1]; params[2] = nrm[2];
    params[3] = -(nrm[0] * c[0] + nrm[1] * c[1] + nrm[2] * c[2]);          /* :84 */
    return m;
}

size_t orc_ground_detection_f64(const float* x, const float* y, const float* z, size_t n, int max_iter, size_t lpr_size,
                                double threshold_dist, double params[4], uint8_t* ground_mask)
{
    uint8_t* seeds = (uint8_t*)malloc(n ? n : 1);
    size_t count = orc_ground_seeds_f64(x, y, z, n, lpr_size, threshold_dist, seeds, NULL);   /* :90, threshold_seeds = threshold_dist */
    for (size_t i = 0; i < n; i++) ground_mask[i] = 0;
    for (int it = 0; it < max_iter; it++) {
        if (count == 0) { free(seeds); return (size_t)-1; }
        orc_estimate_plane_f64(x, y, z, n, seeds, params);                 /* :94 */
        count = 0;
        for (size_t i = 0; i < n; i++) {
            seeds[i] = fabs(plane_dist((double)x[i], (double)y[i], (double)z[i], params)) < threshold_dist;   /* :96-97 */
            count += seeds[i];
        }
        memcpy(ground_mask, seeds, n);
    }
    free(seeds);
    return count;
}

/* ------------------------------------------------------------------ N1, second consumer */
void orc_knn_sq_f32pts(const float* x, const float* y, const float* z, size_t n, const float* qx, const float* qy,
                       const float* qz, size_t m, int k, double cap_s, int32_t* idx, double* s_out, uint32_t* found)
{
    for (size_t i = 0; i < m; i++) {
        int32_t* bi = idx + i * (size_t)k;
        double* bs = s_out + i * (size_t)k;
        int cnt = 0;
        for (int c = 0; c < k; c++) { bi[c] = -1; bs[c] = DBL_MAX; }
        for (size_t j = 0; j < n; j++) {
            const double dx = (double)x[j] - (double)qx[i], dy = (double)y[j] - (double)qy[i], dz = (double)z[j] - (double)qz[i];
            const double s = (dx * dx + dy * dy) + dz * dz;
            if (!(s < cap_s)) continue;
            if (cnt == k && !(s < bs[k - 1])) continue;            /* ascending j: an equal s never displaces a lower index */
            int pos = cnt < k ? cnt : k - 1;
            while (pos > 0 && bs[pos - 1] > s) { bs[pos] = bs[pos - 1]; bi[pos] = bi[pos - 1]; pos--; }
            bs[pos] = s; bi[pos] = (int32_t)j;
            if (cnt < k) cnt++;
        }
        if (found) found[i] = (uint32_t)cnt;
    }
}

void orc_normals_knn_f64(const float* x, const float* y, const float* z, size_t n, int k, double radius, double* normals)
{
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)k);
    double* ss = (double*)malloc(sizeof(double) * (size_t)k);
    const double cap_s = radius < 0 ? INFINITY : radius * radius;
    for (size_t i = 0; i < n; i++) {
        uint32_t cnt = 0;
        orc_knn_sq_f32pts(x, y, z, n, x + i, y + i, z + i, 1, k, cap_s, idx, ss, &cnt);
        double* out = normals + 3 * i;
        out[0] = out[1] = out[2] = 0.0;
        if (cnt < 3) continue;                                       /* pca_normal.py:97 */
        double s[3] = { 0, 0, 0 };
        for (uint32_t c = 0; c < cnt; c++) { s[0] += x[idx[c]]; s[1] += y[idx[c]]; s[2] += z[idx[c]]; }
        const double ctr[3] = { s[0] / (double)cnt, s[1] / (double)cnt, s[2] / (double)cnt };   /* :20 */
        double xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
        for (uint32_t c = 0; c < cnt; c++) {
            const double dx = x[idx[c]] - ctr[0], dy = y[idx[c]] - ctr[1], dz = z[idx[c]] - ctr[2];
            xx += dx * dx; xy += dx * dy; xz += dx * dz; yy += dy * dy; yz += dy * dz; zz += dz * dz;   /* :22 */
        }
        const double XTX[9] = { xx, xy, xz, xy, yy, yz, xz, yz, zz };
        orc_fast_eigen3x3(XTX, out);                                 /* PCA_faster :39-45 */
    }
    free(idx); free(ss);
}

/* ------------------------------------------------------------------ ICPpoint2plane */
int orc_solve6(const double M[36], const double v[6], double x[6])
{
    double a[6][7];
    for (int r = 0; r < 6; r++) { for (int c = 0; c < 6; c++) a[r][c] = M[6 * r + c]; a[r][6] = v[r]; }
    for (int col = 0; col < 6; col++) {
        int piv = col;
        for (int r = col + 1; r < 6; r++) if (fabs(a[r][col]) > fabs(a[piv][col])) piv = r;
        if (!(fabs(a[piv][col]) > 1e-300)) return -1;
        if (piv != col) for (int c = 0; c < 7; c++) { const double t = a[col][c]; a[col][c] = a[piv][c]; a[piv][c] = t; }
        for (int r = col + 1; r < 6; r++) {
            const double f = a[r][col] / a[col][col];
            for (int c = col; c < 7; c++) a[r][c] -= f * a[col][c];
        }
    }
    for (int r = 5; r >= 0; r--) {
        double s = a[r][6];
        for (int c = r + 1; c < 6; c++) s -= a[r][c] * x[c];
        x[r] = s / a[r][r];
    }
    for (int r = 0; r < 6; r++) if (!(fabs(x[r]) <= DBL_MAX)) return -1;
    return 0;
}

void orc_icp_p2plane_f32(const float* sx, const float* sy, const float* sz, size_t ns,
                         const float* tx, const float* ty, const float* tz, size_t nt,
                         const float* tnx, const float* tny, const float* tnz,
                         const float init_T[16], const orc_icp_params* prm, float out_T[16], orc_icp_stats* stats)
{
    float* px = (float*)malloc(sizeof(float) * (ns ? ns : 1));
    float* py = (float*)malloc(sizeof(float) * (ns ? ns : 1));
    float* pz = (float*)malloc(sizeof(float) * (ns ? ns : 1));
    uint32_t* idx = (uint32_t*)malloc(sizeof(uint32_t) * (ns ? ns : 1));
    float* d2 = (float*)malloc(sizeof(float) * (ns ? ns : 1));
    memcpy(px, sx, sizeof(float) * ns); memcpy(py, sy, sizeof(float) * ns); memcpy(pz, sz, sizeof(float) * ns);
    const float R0[9] = { init_T[0], init_T[1], init_T[2], init_T[4], init_T[5], init_T[6], init_T[8], init_T[9], init_T[10] };
    const float t0[3] = { init_T[3], init_T[7], init_T[11] };
    orc_transform_f32(px, py, pz, ns, R0, t0);                                      /* :722 */
    float T_total[16] = { R0[0], R0[1], R0[2], t0[0], R0[3], R0[4], R0[5], t0[1], R0[6], R0[7], R0[8], t0[2], 0, 0, 0, 1 };   /* :759-760 */
    orc_icp_stats st;
    memset(&st, 0, sizeof st);
    float last_loss = 0.0f;
    uint64_t unchanged = 0;
    for (uint64_t iter = 0; iter < prm->max_iter; iter++) {
        orc_nn1_f32(tx, ty, tz, nt, px, py, pz, ns, idx, d2);                        /* :771-781 */
        double M[36], v[6], btb = 0.0, cnt = 0.0;
        memset(M, 0, sizeof M); memset(v, 0, sizeof v);
        for (size_t i = 0; i < ns; i++) {
            if (!(d2[i] < prm->max_corr)) continue;                                  /* :778 */
            const uint32_t j = idx[i];
            const float p[3] = { px[i], py[i], pz[i] }, q[3] = { tx[j], ty[j], tz[j] }, n[3] = { tnx[j], tny[j], tnz[j] };
            const float A[6] = { n[2] * p[1] - n[1] * p[2], n[0] * p[2] - n[2] * p[0], n[1] * p[0] - n[0] * p[1], n[0], n[1], n[2] };   /* :807-812 */
            const float b = n[0] * q[0] + n[1] * q[1] + n[2] * q[2] - n[0] * p[0] - n[1] * p[1] - n[2] * p[2];                         /* :814 */
            for (int r = 0; r < 6; r++) {
                for (int c = r; c < 6; c++) M[6 * r + c] += (double)A[r] * (double)A[c];
                v[r] += (double)A[r] * (double)b;
            }
            btb += (double)b * (double)b;
            cnt += 1.0;
        }
        for (int r = 0; r < 6; r++) for (int c = 0; c < r; c++) M[6 * r + c] = M[6 * c + r];
        st.last_pairs = (uint64_t)cnt;
        double x64[6];
        if (cnt == 0.0 || orc_solve6(M, v, x64) != 0) { st.empty_pairs = 1; break; }   /* :818 (the reference: NaN) */
        float x[6];
        for (int k = 0; k < 6; k++) x[k] = (float)x64[k];
        double xMx = 0.0, xv = 0.0;
        for (int r = 0; r < 6; r++) { for (int c = 0; c < 6; c++) xMx += (double)x[r] * M[6 * r + c] * (double)x[c]; xv += (double)x[r] * v[r]; }
        const float loss = (float)(xMx - 2.0 * xv + btb);                            /* :820 */
        st.last_loss = loss;
        if (fabsf(last_loss - loss) < prm->eps) unchanged++;                         /* :828-831 */
        if (unchanged > 15) { st.converged = 1; break; }                             /* :834-838 */
        last_loss = loss;
        const float Rd[9] = { 1, -x[2], x[1], x[2], 1, -x[0], -x[1], x[0], 1 };      /* :843 */
        const float td[3] = { x[3], x[4], x[5] };
        const float T_delta[16] = { Rd[0], Rd[1], Rd[2], td[0], Rd[3], Rd[4], Rd[5], td[1], Rd[6], Rd[7], Rd[8], td[2], 0, 0, 0, 1 };
        orc_mat4_mul_f32(T_delta, T_total, T_total);                                 /* :849 */
        orc_transform_f32(px, py, pz, ns, Rd, td);                                   /* :851 */
        st.iters_run++;
    }
    memcpy(out_T, T_total, sizeof T_total);
    if (stats) *stats = st;
    free(px); free(py); free(pz); free(idx); free(d2);
}
