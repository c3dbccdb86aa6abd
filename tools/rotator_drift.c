// Measurement (round 6, VERDICT r5 "next" #3): can the closed-form NCO be CALIBRATED to the reference's float rotator?
//
// The reference translates with VOLK's rotator (core/src/dsp/channel/frequency_xlator.h:43-50): phase *= phaseDelta in float, renormalised every
// 512 samples and at the end of a call with a remainder (restated in oracle/shim/volk/volk.h:196-234, which is what this program steps through).
// Its phase departs from the exact phase n * arg(phaseDelta).  If that departure were a straight line, a host could run the recursion for 2^18 steps
// when a VFO is tuned (it does not depend on the samples), fit the slope and fold it into the closed-form NCO's increment.  This program steps the
// recursion for BASELINE cfg 4's 42 USB channels (61.44 MS/s, 307 200-sample blocks, VFO centre = carrier + bandwidth / 2) and prints, per channel,
// the raw departure and what is left of it once the best straight line through the first 2^18 steps is taken out.
//
//   gcc -O2 -ffp-contract=off -o rotator_drift tools/rotator_drift.c -lm && ./rotator_drift [block [samples]]
//
// Test infrastructure / measurement only: nothing in the product uses it.  Output of the committed run: profiles/r06_rotator_drift.md
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char** argv) {
    const double sr = 61.44e6, spacing = 400e3, bw = 2800.0;
    const int block = argc > 1 ? atoi(argv[1]) : 307200;
    const long long N = argc > 2 ? atoll(argv[2]) : 20000000LL;
    const long long NFIT = 1LL << 18;
    double worst_raw[4] = { 0, 0, 0, 0 }, worst_cal[4] = { 0, 0, 0, 0 };
    printf("| k | VFO centre (Hz) | fitted slope (rad/sample) | raw departure at 2e5 / 1e6 / 1e7 / %lld samples (rad) | after the fit: 1e6 / 1e7 / %lld |\n|---|---|---|---|---|\n", N, N);
    for (int k = 2; k < 128; k += 3) {
        const double centre = (k - 63.5) * spacing + bw / 2.0;  // sdrplusplus_amd/workloads.py: vfo_plan(4)
        const double w = 2.0 * M_PI * (-centre / sr);           // frequency_xlator.h:17,28: phaseDelta for -offset
        const float dr = (float)cos(w), di = (float)sin(w);
        const double theta = atan2((double)di, (double)dr);      // the closed form uses arg() of the FLOAT pair
        float pr = 1.0f, pi = 0.0f;
        double sxx = 0, sxy = 0, slope = 0, maxres[4] = { 0, 0, 0, 0 }, maxraw[4] = { 0, 0, 0, 0 };
        long long n = 0;
        int fitted = 0;
        while (n < N) {
            const long long m = (n + block > N) ? N - n : block;
            long long kk = 0;
            for (long long i = 0; i < m / 512; i++) {
                for (int j = 0; j < 512; j++, kk++) {
                    const long long t = n + kk;
                    if ((t & 63) == 0) {
                        double e = atan2((double)pi, (double)pr) - fmod(theta * (double)t, 2 * M_PI);
                        e -= 2 * M_PI * rint(e / (2 * M_PI));
                        if (t < NFIT) { sxx += (double)t * (double)t; sxy += (double)t * e; }
                        else if (!fitted) { slope = sxy / sxx; fitted = 1; }
                        const int b = t <= 200000 ? 0 : (t <= 1000000 ? 1 : (t <= 10000000 ? 2 : 3));
                        if (fitted && fabs(e - slope * (double)t) > maxres[b]) { maxres[b] = fabs(e - slope * (double)t); }
                        if (fabs(e) > maxraw[b]) { maxraw[b] = fabs(e); }
                    }
                    const float nr = (pr * dr) - (pi * di), ni = (pr * di) + (pi * dr);
                    pr = nr;
                    pi = ni;
                }
                const float h = hypotf(pr, pi);
                pr = pr / h;
                pi = pi / h;
            }
            const long long rem = m % 512;
            for (long long i = 0; i < rem; i++, kk++) {
                const float nr = (pr * dr) - (pi * di), ni = (pr * di) + (pi * dr);
                pr = nr;
                pi = ni;
            }
            if (rem) {
                const float h = hypotf(pr, pi);
                pr = pr / h;
                pi = pi / h;
            }
            n += m;
        }
        printf("| %d | %+.0f | %+.3e | %.1e / %.1e / %.1e / %.1e | %.1e / %.1e / %.1e |\n", k, centre, slope, maxraw[0], maxraw[1], maxraw[2], maxraw[3], maxres[1], maxres[2], maxres[3]);
        for (int b = 0; b < 4; b++) {
            if (maxraw[b] > worst_raw[b]) { worst_raw[b] = maxraw[b]; }
            if (maxres[b] > worst_cal[b]) { worst_cal[b] = maxres[b]; }
        }
    }
    printf("\nworst channel, raw: %.1e (2e5) %.1e (1e6) %.1e (1e7) %.1e (end); after the fit: %.1e (1e6) %.1e (1e7) %.1e (end)\n", worst_raw[0], worst_raw[1], worst_raw[2], worst_raw[3], worst_cal[1], worst_cal[2],
           worst_cal[3]);
    return 0;
}
