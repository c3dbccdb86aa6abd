// Host-side (CPU, double precision) design maths of the SDR++ hot path, restated for callers that do not link SDR++'s
// headers.  These functions produce the CONSTANTS the kernels consume (taps, windows, NCO increments, twiddle tables);
// they never touch sample data.  Each cites the reference header it follows (paths relative to /root/reference).
#include "../../include/sdrpp_gpu.h"
#include "host_design.h"

#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

namespace {
constexpr double kPi = 3.14159265358979323846;  // DB_M_PI, core/src/dsp/math/constants.h:3

// core/src/dsp/window/cosine.h:7-15 — generalised cosine-sum window, alternating signs
double cosineWindow(double n, double N, const double* a, int terms) {
    double acc = 0.0, sign = 1.0;
    for (int i = 0; i < terms; i++) {
        acc += sign * a[i] * std::cos((double)i * 2.0 * kPi * n / N);
        sign = -sign;
    }
    return acc;
}
double nuttall(double n, double N) {  // window/nuttall.h:5-8
    static const double a[] = { 0.355768, 0.487396, 0.144232, 0.012604 };
    return cosineWindow(n, N, a, 4);
}
double blackman(double n, double N) {  // window/blackman.h:5-8
    static const double a[] = { 0.42, 0.5, 0.08 };
    return cosineWindow(n, N, a, 3);
}
double sinc(double x) { return (x == 0.0) ? 1.0 : std::sin(x) / x; }           // math/sinc.h:5-7
double hzToRads(double f, double sr) { return 2.0 * kPi * (f / sr); }            // math/hz_to_rads.h:6-8
int estimateTapCount(double tw, double sr) { return (int)(3.8 * sr / tw); }      // taps/estimate_tap_count.h:5

// taps/windowed_sinc.h:9-29 specialised to the two windows the hot path uses
int windowedSinc(int count, double omega, bool alternate, float* taps, int max) {
    const double half = (double)count / 2.0;
    const double corr = omega / kPi;  // norm = 1.0
    for (int i = 0; i < count && i < max; i++) {
        const double t = (double)i - half + 0.5;
        double w = nuttall(t - half, (double)count);
        if (alternate) {  // taps/high_pass.h:11-13: nuttall * (-1)^round(n)
            w = w * ((((int)std::round(t - half)) % 2) ? -1.0f : 1.0f);
        }
        taps[i] = (float)(sinc(t * omega) * w * corr);
    }
    return count;
}
}  // namespace

extern "C" {

int sdrpp_design_low_pass(double cutoff, double trans_width, double sample_rate, int odd, float* taps, int max) {
    int count = estimateTapCount(trans_width, sample_rate);
    if (odd && !(count % 2)) { count++; }
    return windowedSinc(count, hzToRads(cutoff, sample_rate), false, taps, max);
}

int sdrpp_design_high_pass(double cutoff, double trans_width, double sample_rate, int odd, float* taps, int max) {
    int count = estimateTapCount(trans_width, sample_rate);
    if (odd && !(count % 2)) { count++; }
    return windowedSinc(count, hzToRads((sample_rate / 2.0) - cutoff, sample_rate), true, taps, max);
}

int sdrpp_design_fft_window(int kind, int nz, float* window) {
    if (kind < 0 || kind > 2 || nz <= 0 || !window) { return SDRPP_ERR_INVALID; }
    for (int i = 0; i < nz; i++) {
        const float flip = (i % 2) ? -1.0f : 1.0f;  // fftshift folded into the window, iq_frontend.cpp:284-290
        if (kind == 0) { window[i] = 1.0f * flip; }
        else if (kind == 1) { window[i] = blackman(i, nz) * flip; }
        else { window[i] = nuttall(i, nz) * flip; }
    }
    return SDRPP_OK;
}

void sdrpp_design_reshape_params(double sample_rate, int fft_size, double fft_rate, int* skip, int* nz) {
    const int interval = (int)std::round(sample_rate / fft_rate);  // iq_frontend.h:60
    *nz = interval < fft_size ? interval : fft_size;
    *skip = interval - *nz;
}

void sdrpp_design_phase_delta(double offset_hz, double sample_rate, float* re, float* im) {
    const double w = hzToRads(offset_hz, sample_rate);
    *re = (float)std::cos(w);
    *im = (float)std::sin(w);
}

float sdrpp_design_deemphasis_alpha(double tau, double sample_rate) {
    const float dt = 1.0f / sample_rate;  // deephasis.h:91-92: float dt = 1.0f / _samplerate; alpha = dt / (_tau + dt);
    return (float)(dt / (tau + dt));
}

int sdrpp_design_resampler(double in_sr, double out_sr, int max_ratio, int* mode, int* predec_ratio, int* interp, int* decim,
                           float* taps, int max) {
    // rational_resampler.h:120-165.  Note the reference clamps the POWER against the maximum RATIO (line 122); kept.
    int power = (int)std::floor(std::log2(in_sr / out_sr));
    if (power > max_ratio) { power = max_ratio; }
    int ratio = (power >= 0 && power < 31) ? (1 << power) : max_ratio;
    if (ratio > max_ratio) { ratio = max_ratio; }
    const bool useDecim = (in_sr > out_sr && power > 0);
    double intSr = in_sr;
    *predec_ratio = 1;
    if (useDecim) {
        intSr = in_sr / (double)ratio;
        *predec_ratio = ratio;
    }
    const int iSr = (int)std::round(intSr), oSr = (int)std::round(out_sr);
    const int g = std::gcd(iSr, oSr);
    *interp = oSr / g;
    *decim = iSr / g;
    if (*interp == *decim) {
        *mode = useDecim ? 1 : 3;
        return 0;
    }
    const double tapSr = intSr * (double)(*interp);
    const double tapBw = (in_sr < out_sr ? in_sr : out_sr) / 2.0;
    const double tapTw = tapBw * 0.1;
    const int n = sdrpp_design_low_pass(tapBw, tapTw, tapSr, 0, taps, max);
    for (int i = 0; i < n && i < max; i++) { taps[i] *= (float)(*interp); }  // line 159
    *mode = useDecim ? 0 : 2;
    return n;
}

void sdrpp_design_waterfall_view(double view_offset, double view_bandwidth, double whole_bandwidth, int raw_fft_size,
                                 int* draw_data_start, int* draw_data_size) {
    const double offsetRatio = view_offset / (whole_bandwidth / 2.0);                        // waterfall.cpp:892
    const int size = (view_bandwidth / whole_bandwidth) * raw_fft_size;                       // :893
    const int start = (((double)raw_fft_size / 2.0) * (offsetRatio + 1)) - (size / 2);        // :894
    *draw_data_start = start;
    *draw_data_size = size;
}

}  // extern "C"

namespace sdrpp_host {

// FFT twiddle tw(e, L) = exp(-2*pi*i*e/L) as float2.  Only the first octant is evaluated with libm (double, rounded
// once); the rest of the circle follows from exact sign/swap symmetries, so tw(e + L/4) == -j * tw(e) bit for bit —
// the kernels rely on that to derive half of their twiddles by a register swap.
void twiddle(int e, int L, float* re, float* im) {
    int circle = L, idx = ((e % L) + L) % L;
    if (circle < 8) { idx *= 8 / circle; circle = 8; }
    const int quarter = circle / 4;
    const int quadrant = idx / quarter, rem = idx % quarter;
    float c, s;
    if (rem <= circle / 8) {
        const double a = 2.0 * kPi * ((double)rem / (double)circle);
        c = (float)std::cos(a);
        s = (float)std::sin(a);
    }
    else {
        const double a = 2.0 * kPi * ((double)(quarter - rem) / (double)circle);
        c = (float)std::sin(a);
        s = (float)std::cos(a);
    }
    float cf, sf;
    if (quadrant == 0) { cf = c; sf = s; }
    else if (quadrant == 1) { cf = -s; sf = c; }
    else if (quadrant == 2) { cf = -c; sf = -s; }
    else { cf = s; sf = -c; }
    *re = cf;
    *im = -sf;
}

// doZoom's float32 running index (waterfall.cpp:74-89) evaluated once per view change: for each output pixel the first
// bin and the number of bins it covers.  The kernel then only does the max-reduction.
void zoomTable(int offset, int width, int inSize, int outSize, std::vector<int32_t>& start, std::vector<int32_t>& count) {
    if (offset < 0) { offset = 0; }
    if (width > 524288) { width = 524288; }
    start.assign((size_t)outSize, 0);
    count.assign((size_t)outSize, 0);
    const float factor = (float)width / (float)outSize;
    const float sFactor = std::ceil(factor);
    float id = (float)offset;
    for (int i = 0; i < outSize; i++) {
        const int sId = (int)id;
        const float uFactor = (sId + sFactor > inSize) ? sFactor - ((sId + sFactor) - inSize) : sFactor;
        int n = 0;
        for (int j = 0; j < uFactor; j++) { n++; }  // same float comparison as the reference loop
        start[(size_t)i] = sId;
        count[(size_t)i] = n;
        id += factor;
    }
}

// Effective NCO increment of the reference rotator in turns per sample: the recursion phase *= phaseDelta advances by
// arg(phaseDelta) each sample, where phaseDelta is the float pair the reference stores (frequency_xlator.h:17).
double turnsPerSample(float re, float im) { return std::atan2((double)im, (double)re) / (2.0 * kPi); }

}  // namespace sdrpp_host
