// Host build of the 5-point core for debugging: g++ -O2 -ffp-contract=off -shared -fPIC fivepoint_host.cpp -o libfp_host.so
#include "../../pose_estimation_amd/csrc/fivepoint_core.hpp"
extern "C" int solve5(const double* p, float* eo) {
    double pts[5][4];
    for (int r = 0; r < 5; ++r) for (int c = 0; c < 4; ++c) pts[r][c] = p[4 * r + c];
    for (int k = 0; k < 90; ++k) eo[k] = 0.f;
    return ph5::solve_fivepoint(pts, eo);
}
