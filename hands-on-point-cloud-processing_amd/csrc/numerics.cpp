// numerics.cpp — host entry points of the shared 3x3 numerics (numerics.hpp).
#include "pcr_internal.hpp"
#include "numerics.hpp"

namespace pcr {

void svd3(const double A[9], double U[9], double S[3], double V[9]) { num::svd3(A, U, S, V); }

int kabsch_solve(const double sums[16], float R[9], float t[3])
{
    return num::kabsch_solve(sums, R, t) == 0 ? PCR_OK : PCR_ERR_EMPTY;
}

void mat4_mul_f32(const float A[16], const float B[16], float out[16]) { num::mat4_mul_f32(A, B, out); }

}  // namespace pcr
