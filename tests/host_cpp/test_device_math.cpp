// TEST: device math helpers of the kernels, compiled for the host against the fiber emulator's headers (tests/emu) and swept against
// double-precision libm.  fm_phase is the discriminator's atan2 (vfo_kernels.h); the audio parity bar (1e-5 RMS) needs it to stay
// within a few 1e-7 rad of atan2f everywhere, including the axes, the octant seams and tiny / huge magnitudes.
#include <hip/hip_runtime.h>
#include <sdrpp_gfx950.h>
#include "vfo_kernels.h"
#include <cmath>
#include <cstdio>
#include <random>

int main() {
    double worst = 0.0, worst_vs_f = 0.0;
    float wy = 0, wx = 0;
    auto check = [&](float y, float x) {
        const float got = sdrpp_k::fm_phase(y, x);
        const double ref = std::atan2((double)y, (double)x);
        double e = std::fabs((double)got - ref);
        if (e > 3.14159265358979) { e = std::fabs(e - 2.0 * 3.14159265358979323846); }  // -pi against +pi on the negative real axis
        if (e > worst) { worst = e; wy = y; wx = x; }
        const double ef = std::fabs((double)got - (double)std::atan2(y, x));
        if (ef < 3.0 && ef > worst_vs_f) { worst_vs_f = ef; }
    };
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    for (int i = 0; i < 2000000; i++) { check(u(rng), u(rng)); }
    for (int i = 0; i < 200000; i++) {  // magnitudes from 1e-30 to 1e30
        const float s = std::pow(10.0f, 30.0f * u(rng));
        check(u(rng) * s, u(rng) * s);
    }
    for (int k = 0; k < 360000; k++) {  // a fine sweep of the circle: octant seams, axes
        const double a = (double)k * (3.14159265358979323846 / 180000.0);
        check((float)std::sin(a), (float)std::cos(a));
    }
    const float axes[][2] = { { 0.0f, 1.0f }, { 1.0f, 0.0f }, { 0.0f, -1.0f }, { -1.0f, 0.0f }, { 1.0f, 1.0f }, { -1.0f, 1.0f }, { 1.0f, -1.0f }, { -1.0f, -1.0f }, { -0.0f, -1.0f } };
    for (auto& p : axes) { check(p[0], p[1]); }
    int fail = 0;
    if (sdrpp_k::fm_phase(0.0f, 0.0f) != 0.0f) { printf("fm_phase(0, 0) = %g, atan2f gives 0\n", sdrpp_k::fm_phase(0.0f, 0.0f)); fail = 1; }
    printf("fm_phase: max |error| against double atan2 %.3g rad (at y = %g, x = %g); max distance from libm atan2f %.3g rad\n", worst, wy, wx, worst_vs_f);
    if (worst > 3.5e-7) { fail = 1; }
    // normalize_phase (math/normalize_phase.h): result in (-pi, pi] for differences of two phases
    for (int i = 0; i < 100000; i++) {
        const float d = 6.2831853f * u(rng);
        const float n = sdrpp_k::normalize_phase(d);
        if (!(n > -3.1415927f - 1e-6f && n <= 3.1415927f + 1e-6f) || std::fabs(std::remainder((double)d - (double)n, 2.0 * 3.14159265358979)) > 1e-6) {
            printf("normalize_phase(%g) = %g\n", d, n);
            fail = 1;
            break;
        }
    }
    return fail;
}
