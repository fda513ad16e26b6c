// Host build of the P3P core for debugging: g++ -O2 -ffp-contract=off -shared -fPIC p3p_host.cpp -o libp3p_host.so
#include "../../pose_estimation_amd/csrc/p3p_core.hpp"
extern "C" int solve_p3p(const double* X9, const double* f9, double* Rt48) {
    double X[3][3], f[3][3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { X[r][c] = X9[3 * r + c]; f[r][c] = f9[3 * r + c]; }
    return p3p::solve(X, f, reinterpret_cast<double (*)[12]>(Rt48));
}
