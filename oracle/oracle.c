/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under sdrplusplus_amd/ may include, link or call this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Plain-C restatement of the SDR++ streaming-DSP hot path (reference = /root/reference, citations are relative to
 * it).  Every function follows the reference's arithmetic operation by operation (same fp32/fp64 types, same
 * summation order, same state carried between blocks) so that it can be compared BIT-EXACTLY with oracle/_ref
 * (the reference's own headers compiled against oracle/shim).
 *
 * PINNING STATUS: the reference ships no tests, golden vectors or fixtures for this path, so
 * parity is pinned against outputs of the reference itself run in the build container (oracle/_ref, and the
 * fixtures under tests/golden/ generated from it by tests/golden/make_golden.py).  The reference's third-party
 * arithmetic — libvolk (unpinned, distro 2.5.x) and libfftw3f (unpinned, FFTW_ESTIMATE) — is NOT under
 * /root/reference and not installed; it is restated here from VOLK's published generic kernels and, for the FFT,
 * as a fully specified fp32 algorithm ("any exact DFT" is all the reference promises).  For those two pieces
 * parity is therefore "unpinned by the reference" and bounded instead by float64 recomputation in tests/.
 *
 * Compile: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  fmaf() is used only where
 * the FFT / log2 specifications below say so; the compiler must not contract anything else.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define DB_M_PI 3.14159265358979323846 /* dsp/math/constants.h:3 */
#define FL_M_PI 3.1415926535f          /* dsp/math/constants.h:4 */

/* ===================================================================================================================
 * 1. Specified fp32 maths shared (operation for operation) with the HIP kernels
 * =================================================================================================================== */

/* log2 of a positive finite float.  Specification:
 *   x subnormal  -> x *= 2^23, e -= 23 (exact)
 *   x = m * 2^e, m in [1,2);  if m >= 1.41421354f { m *= 0.5 (exact); e += 1 }
 *   f = m - 1 (exact);  P = Horner with fmaf over C9..C0;  result = fmaf(f, P, (float)e)
 * Special cases: x == 0 -> -inf; x < 0 or NaN -> NaN; x == +inf -> +inf. */
static const float LOG2_C[10] = {
    1.44269502f, -0.721347332f, 0.480898529f, -0.360696554f, 0.288552552f,
    -0.239608124f, 0.204857647f, -0.191388384f, 0.184760332f, -0.107497893f
};

float sdrpp_oracle_log2f(float x) {
    union { float f; uint32_t u; } v;
    v.f = x;
    if (x != x) { return x; }
    if (x < 0.0f) { return NAN; }
    if (x == 0.0f) { return -INFINITY; }
    if (v.u == 0x7f800000u) { return INFINITY; }
    int e = 0;
    if (v.u < 0x00800000u) { /* subnormal */
        v.f = x * 8388608.0f;
        e = -23;
    }
    e += (int)(v.u >> 23) - 127;
    v.u = (v.u & 0x007fffffu) | 0x3f800000u;
    float m = v.f;
    if (m >= 1.41421354f) {
        m = m * 0.5f;
        e += 1;
    }
    const float f = m - 1.0f;
    float p = LOG2_C[9];
    for (int i = 8; i >= 0; i--) { p = fmaf(p, f, LOG2_C[i]); }
    return fmaf(f, p, (float)e);
}

/* VOLK volk_common.h log2f_non_ieee: an infinite log2 is replaced by copysignf(127.0f, result). */
float sdrpp_oracle_log2f_non_ieee(float x) {
    const float r = sdrpp_oracle_log2f(x);
    return isinf(r) ? copysignf(127.0f, r) : r;
}

/* Twiddle tw(e, L) = e^{-2 pi i e / L} as two floats.  Only angles in [0, pi/4] are evaluated (double libm, rounded
 * once to float); every other entry follows by the exact symmetries of sine and cosine, so the table is exactly
 * symmetric: tw(0) = (1, -0), tw(L/4) = (-0, -1), tw(e + L/4) = -j * tw(e) bit for bit. */
void sdrpp_oracle_twiddle(int e, int L, float* re, float* im) {
    int Lv = L, ev = ((e % L) + L) % L;
    if (Lv < 8) { ev *= 8 / Lv; Lv = 8; }
    const int quarter = Lv / 4, qd = ev / quarter, r = ev % quarter;
    float c, s;
    if (r <= Lv / 8) {
        const double a = 2.0 * DB_M_PI * ((double)r / (double)Lv);
        c = (float)cos(a);
        s = (float)sin(a);
    }
    else {
        const double a = 2.0 * DB_M_PI * ((double)(quarter - r) / (double)Lv);
        c = (float)sin(a);
        s = (float)cos(a);
    }
    float cv, sv; /* cos and sin of the full angle */
    switch (qd) {
    case 0: cv = c; sv = s; break;
    case 1: cv = -s; sv = c; break;
    case 2: cv = -c; sv = -s; break;
    default: cv = s; sv = -c; break;
    }
    *re = cv;
    *im = -sv;
}

#define ORC_FFT_SINGLE_PASS_MAX 4096

static unsigned bitrev(unsigned v, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; i++) {
        r = (r << 1) | (v & 1u);
        v >>= 1;
    }
    return r;
}

/* Sub-FFT, L = 2^lg <= 4096: classic radix-2 decimation in time on the bit-reversed sequence.  Every butterfly
 * (u, v, w = tw(k, M)), including the ones with w = 1 and w = -j, is
 *     X0.re = fmaf(-w.im, v.im, fmaf(w.re, v.re, u.re))
 *     X0.im = fmaf( w.im, v.re, fmaf(w.re, v.im, u.im))
 *     X1    = fmaf(2, u, -X0)                                   (per component)
 * i.e. six fused multiply-adds.  Because tw(0) and tw(M/4) are exact, X0 degenerates to u + v resp.
 * (u.re + v.im, u.im - v.re) for them, bit for bit (up to the sign of a zero, which never reaches the dB output).
 * tw(k, M) is read as tw(k * L/M, L) - the same float because the symmetric table only depends on k/M. */
static void sub_fft(int lg, const float* in, int in_stride /* complex elements */, float* y, const float* tw /* L/2 pairs */) {
    const int L = 1 << lg;
    for (int p = 0; p < L; p++) {
        const int n = (int)bitrev((unsigned)p, lg);
        y[2 * p] = in[2 * (size_t)n * in_stride];
        y[2 * p + 1] = in[2 * (size_t)n * in_stride + 1];
    }
    for (int s = 1; s <= lg; s++) {
        const int M = 1 << s, H = M >> 1, tstep = L / M;
        for (int b = 0; b < L; b += M) {
            for (int k = 0; k < H; k++) {
                float* U = &y[2 * (b + k)];
                float* V = &y[2 * (b + k + H)];
                const float ur = U[0], ui = U[1], vr = V[0], vi = V[1];
                const float wr = tw[2 * k * tstep], wi = tw[2 * k * tstep + 1];
                const float x0r = fmaf(-wi, vi, fmaf(wr, vr, ur));
                const float x0i = fmaf(wi, vr, fmaf(wr, vi, ui));
                U[0] = x0r; U[1] = x0i;
                V[0] = fmaf(2.0f, ur, -x0r);
                V[1] = fmaf(2.0f, ui, -x0i);
            }
        }
    }
}

/* Twiddle tables are cached per size (test-infrastructure convenience; the VALUES are those of sdrpp_oracle_twiddle). */
static float* tw_half_cache[21];
static float* tw_full_cache[21];
static const float* get_tw_half(int lg) {
    if (!tw_half_cache[lg]) {
        const int L = 1 << lg;
        float* t = (float*)malloc(sizeof(float) * 2 * (size_t)(L / 2 > 0 ? L / 2 : 1));
        for (int e = 0; e < L / 2; e++) { sdrpp_oracle_twiddle(e, L, &t[2 * e], &t[2 * e + 1]); }
        tw_half_cache[lg] = t;
    }
    return tw_half_cache[lg];
}
static const float* get_tw_full(int lg) {
    if (!tw_full_cache[lg]) {
        const int L = 1 << lg;
        float* t = (float*)malloc(sizeof(float) * 2 * (size_t)L);
        for (int e = 0; e < L; e++) { sdrpp_oracle_twiddle(e, L, &t[2 * e], &t[2 * e + 1]); }
        tw_full_cache[lg] = t;
    }
    return tw_full_cache[lg];
}

/* Forward unnormalised DFT, sign -1, n = 2^m, 1 <= m <= 20 (what fftwf_plan_dft_1d(FFTW_FORWARD) computes,
 * iq_frontend.cpp:62,255).  n <= 4096: one sub-FFT.  n > 4096: four-step, N1 = 2^floor(m/2) up to n = 65536 and N1 = n / 4096 above
 * (the row transform is then the largest single sub-FFT), N2 = n / N1:
 *   A[k1][n2] = subFFT_N1 over n1 of x[N2*n1 + n2]
 *   B[k1][n2] = A[k1][n2] * tw(n2*k1, n):  re = fmaf(a.re, w.re, -(a.im*w.im)),  im = fmaf(a.re, w.im, a.im*w.re)
 *   X[k1 + N1*k2] = subFFT_N2 over n2 of B[k1][n2]
 * Not thread-safe on first use of a size (table cache); tests and the baseline warm it from one thread. */
void sdrpp_oracle_fft(int n, const float* in, float* out) {
    int m = 0;
    while ((1 << m) < n) { m++; }
    if ((1 << m) != n || n < 2 || m > 20) {
        fprintf(stderr, "sdrpp_oracle_fft: unsupported size %d\n", n);
        abort();
    }
    if (n <= ORC_FFT_SINGLE_PASS_MAX) {
        const float* tw = get_tw_half(m);
        float* y = (float*)malloc(sizeof(float) * 2 * (size_t)n);
        sub_fft(m, in, 1, y, tw);
        memcpy(out, y, sizeof(float) * 2 * (size_t)n);
        free(y);
        return;
    }
    const int lg1 = (m <= 16) ? m / 2 : m - 12, lg2 = m - lg1;
    const int N1 = 1 << lg1, N2 = 1 << lg2;
    const float* tw1 = get_tw_half(lg1);
    const float* tw2 = get_tw_half(lg2);
    const float* twN = get_tw_full(m);
    float* B = (float*)malloc(sizeof(float) * 2 * (size_t)n); /* [k1][n2] */
    float* col = (float*)malloc(sizeof(float) * 2 * (size_t)(N1 > N2 ? N1 : N2));
    for (int n2 = 0; n2 < N2; n2++) {
        sub_fft(lg1, in + 2 * (size_t)n2, N2, col, tw1);
        for (int k1 = 0; k1 < N1; k1++) {
            const float ar = col[2 * k1], ai = col[2 * k1 + 1];
            float* dst = &B[2 * ((size_t)k1 * N2 + n2)];
            const size_t e = (size_t)n2 * k1;
            const float wr = twN[2 * e], wi = twN[2 * e + 1];
            const float p = ai * wi;
            const float q = ai * wr;
            dst[0] = fmaf(ar, wr, -p);
            dst[1] = fmaf(ar, wi, q);
        }
    }
    for (int k1 = 0; k1 < N1; k1++) {
        sub_fft(lg2, &B[2 * (size_t)k1 * N2], 1, col, tw2);
        for (int k2 = 0; k2 < N2; k2++) {
            out[2 * ((size_t)k1 + (size_t)N1 * k2)] = col[2 * k2];
            out[2 * ((size_t)k1 + (size_t)N1 * k2) + 1] = col[2 * k2 + 1];
        }
    }
    free(col);
    free(B);
}

/* ===================================================================================================================
 * 2. Host-side (double precision) design maths
 * =================================================================================================================== */

/* dsp/window/cosine.h:7-15 */
static double win_cosine(double n, double N, const double* coefs, int coefCount) {
    double win = 0.0;
    double sign = 1.0;
    for (int i = 0; i < coefCount; i++) {
        win += sign * coefs[i] * cos((double)i * 2.0 * DB_M_PI * n / N);
        sign = -sign;
    }
    return win;
}
/* dsp/window/nuttall.h:5-8 */
double orc_nuttall(double n, double N) {
    const double coefs[] = { 0.355768, 0.487396, 0.144232, 0.012604 };
    return win_cosine(n, N, coefs, 4);
}
/* dsp/window/blackman.h:5-8 */
double orc_blackman(double n, double N) {
    const double coefs[] = { 0.42, 0.5, 0.08 };
    return win_cosine(n, N, coefs, 3);
}
/* dsp/math/sinc.h:5-7 */
static double m_sinc(double x) { return (x == 0.0) ? 1.0 : (sin(x) / x); }
/* dsp/math/hz_to_rads.h:6-8 */
static double hz_to_rads(double freq, double samplerate) { return 2.0 * DB_M_PI * (freq / samplerate); }

/* dsp/taps/estimate_tap_count.h:5 (double -> int truncation) */
int orc_estimate_tap_count(double transWidth, double samplerate) { return (int)(3.8 * samplerate / transWidth); }

/* dsp/taps/windowed_sinc.h:9-29 with window = nuttall (low_pass.h:10) or nuttall * (-1)^round(n) (high_pass.h:11-13).
 * Returns the tap count; writes min(count, max) taps. */
static int windowed_sinc(int count, double omega, int highpass, float* taps, int max) {
    const double half = (double)count / 2.0;
    const double corr = 1.0 * omega / DB_M_PI;
    for (int i = 0; i < count && i < max; i++) {
        const double t = (double)i - half + 0.5;
        double w = orc_nuttall(t - half, count);
        if (highpass) { w = w * ((((int)round(t - half)) % 2) ? -1.0f : 1.0f); }
        taps[i] = (float)(m_sinc(t * omega) * w * corr);
    }
    return count;
}
/* dsp/taps/low_pass.h:7-11 */
int orc_low_pass(double cutoff, double transWidth, double sampleRate, int oddTapCount, float* taps, int max) {
    int count = orc_estimate_tap_count(transWidth, sampleRate);
    if (oddTapCount && !(count % 2)) { count++; }
    return windowed_sinc(count, hz_to_rads(cutoff, sampleRate), 0, taps, max);
}
/* dsp/taps/high_pass.h:7-15 */
int orc_high_pass(double cutoff, double transWidth, double sampleRate, int oddTapCount, float* taps, int max) {
    int count = orc_estimate_tap_count(transWidth, sampleRate);
    if (oddTapCount && !(count % 2)) { count++; }
    return windowed_sinc(count, hz_to_rads((sampleRate / 2.0) - cutoff, sampleRate), 1, taps, max);
}

/* signal_path/iq_frontend.h:59-63 */
void orc_gen_reshape_params(double sampleRate, int size, double rate, int* skip, int* nzSampCount) {
    const int fftInterval = (int)round(sampleRate / rate);
    *nzSampCount = fftInterval < size ? fftInterval : size;
    *skip = fftInterval - *nzSampCount;
}

/* signal_path/iq_frontend.cpp:280-291 — kind: 0 RECTANGULAR, 1 BLACKMAN, 2 NUTTALL (iq_frontend.h:18-22).
 * The (-1)^i factor moves DC to bin N/2 (fftshift folded into the window). */
void orc_fft_window(int kind, int nz, float* w) {
    for (int i = 0; i < nz; i++) {
        const float sgn = (i % 2) ? -1.0f : 1.0f;
        if (kind == 0) { w[i] = 1.0f * sgn; }
        else if (kind == 1) { w[i] = orc_blackman(i, nz) * sgn; } /* double * float -> double, stored float */
        else { w[i] = orc_nuttall(i, nz) * sgn; }
    }
}

/* ===================================================================================================================
 * 3. FFT -> log-power -> waterfall line
 * =================================================================================================================== */

/* VOLK volk_32fc_s32f_power_spectrum_32f generic, as called at iq_frontend.cpp:262 with norm = fftSize. */
void orc_power_spectrum(const float* X, float norm, int n, float* out) {
    const float inv = 1.0f / norm;
    for (int i = 0; i < n; i++) {
        const float re = X[2 * i] * inv;
        const float im = X[2 * i + 1] * inv;
        const float p = (re * re) + (im * im);
        out[i] = 3.01029995663981209120f * sdrpp_oracle_log2f_non_ieee(p);
    }
}

/* Reshaper framing (dsp/buffer/reshaper.h:101-128, ring_buffer.h:66-110) + IQFrontEnd::handler
 * (iq_frontend.cpp:248-267): frame k = samples [k*(nz+skip), k*(nz+skip)+nz) counted from stream start; each frame
 * is windowed into fftIn[0:nz], fftIn[nz:N] stays zero (iq_frontend.cpp:301), transformed and converted to dB. */
typedef struct {
    int N, nz, skip;
    float* window;   /* nz */
    float* frame;    /* nz complex being assembled */
    int have;        /* samples of the current frame already collected */
    long long toskip; /* samples still to drop before the next frame starts */
    float* fftIn;
    float* fftOut;
} orc_spectrum;

orc_spectrum* orc_spectrum_create(int fftSize, int nz, int skip, const float* window) {
    orc_spectrum* s = (orc_spectrum*)calloc(1, sizeof(orc_spectrum));
    s->N = fftSize; s->nz = nz; s->skip = skip;
    s->window = (float*)malloc(sizeof(float) * (size_t)nz);
    memcpy(s->window, window, sizeof(float) * (size_t)nz);
    s->frame = (float*)calloc((size_t)nz * 2, sizeof(float));
    s->fftIn = (float*)calloc((size_t)fftSize * 2, sizeof(float));
    s->fftOut = (float*)calloc((size_t)fftSize * 2, sizeof(float));
    return s;
}
void orc_spectrum_destroy(orc_spectrum* s) {
    if (!s) { return; }
    free(s->window); free(s->frame); free(s->fftIn); free(s->fftOut); free(s);
}
/* Returns the number of dB lines written to `lines` (each N floats), at most maxLines (excess frames are still
 * consumed but not stored — callers size maxLines generously). */
int orc_spectrum_push(orc_spectrum* s, const float* iq, int count, float* lines, int maxLines) {
    int produced = 0;
    int i = 0;
    while (i < count) {
        if (s->toskip > 0) {
            long long d = s->toskip < (long long)(count - i) ? s->toskip : (long long)(count - i);
            s->toskip -= d;
            i += (int)d;
            continue;
        }
        int take = s->nz - s->have;
        if (take > count - i) { take = count - i; }
        memcpy(&s->frame[2 * s->have], &iq[2 * (size_t)i], sizeof(float) * 2 * (size_t)take);
        s->have += take;
        i += take;
        if (s->have == s->nz) {
            /* volk_32fc_32f_multiply_32fc (iq_frontend.cpp:252) */
            for (int k = 0; k < s->nz; k++) {
                s->fftIn[2 * k] = s->frame[2 * k] * s->window[k];
                s->fftIn[2 * k + 1] = s->frame[2 * k + 1] * s->window[k];
            }
            sdrpp_oracle_fft(s->N, s->fftIn, s->fftOut);
            if (produced < maxLines) { orc_power_spectrum(s->fftOut, (float)s->N, s->N, &lines[(size_t)produced * s->N]); }
            produced++;
            s->have = 0;
            s->toskip = s->skip;
        }
    }
    return produced;
}

/* gui/widgets/waterfall.cpp:65-90 — max-decimation of a raw line to outSize pixels; float32 running index. */
void orc_do_zoom(int offset, int width, int inSize, int outSize, const float* in, float* out) {
    if (offset < 0) { offset = 0; }
    if (width > 524288) { width = 524288; }
    float factor = (float)width / (float)outSize;
    float sFactor = ceilf(factor);
    float uFactor;
    float id = offset;
    float maxVal;
    int sId;
    for (int i = 0; i < outSize; i++) {
        maxVal = -INFINITY;
        sId = (int)id;
        uFactor = (sId + sFactor > inSize) ? sFactor - ((sId + sFactor) - inSize) : sFactor;
        for (int j = 0; j < uFactor; j++) {
            if (in[sId + j] > maxVal) { maxVal = in[sId + j]; }
        }
        out[i] = maxVal;
        id += factor;
    }
}

/* gui/widgets/waterfall.cpp:891-894 */
void orc_waterfall_view(double viewOffset, double viewBandwidth, double wholeBandwidth, int rawFFTSize, int* drawDataStart, int* drawDataSize) {
    double offsetRatio = viewOffset / (wholeBandwidth / 2.0);
    int size = (viewBandwidth / wholeBandwidth) * rawFFTSize;
    int start = (((double)rawFFTSize / 2.0) * (offsetRatio + 1)) - (size / 2);
    *drawDataStart = start;
    *drawDataSize = size;
}

/* gui/widgets/waterfall.cpp:899-905 — palette index: clamp, normalise, truncate to WATERFALL_RESOLUTION-1 levels. */
#define WATERFALL_RESOLUTION 1000000 /* gui/widgets/waterfall.h:11 */
void orc_palette_index(const float* zoomed, int n, float waterfallMin, float waterfallMax, int32_t* idx) {
    float dataRange = waterfallMax - waterfallMin;
    for (int j = 0; j < n; j++) {
        float v = zoomed[j];
        v = (v < waterfallMin) ? waterfallMin : ((waterfallMax < v) ? waterfallMax : v); /* std::clamp */
        float pixel = (v - waterfallMin) / dataRange;
        idx[j] = (int)(pixel * (WATERFALL_RESOLUTION - 1));
    }
}

/* ===================================================================================================================
 * 4. Channeliser building blocks (all streaming, state carried across calls)
 * =================================================================================================================== */

/* ---- FrequencyXlator (dsp/channel/frequency_xlator.h) + VOLK rotator2 generic ---------------------------------- */
/* `ideal` (TEST SWITCH, never the pinned configuration): the float recursion is replaced by a float64 NCO at exactly
 * arg(phaseDelta_f32) per sample — same float phaseDelta, same float complex product, everything behind the xlator unchanged —
 * so a comparison against it isolates what the recursion's own rounding (drift, amplitude saw-tooth) contributes. */
typedef struct { float pr, pi, dr, di; int ideal; double theta, acc; } orc_xlator;

static void xlator_set_offset(orc_xlator* x, double offset, double samplerate) {
    const double o = hz_to_rads(offset, samplerate);
    x->dr = (float)cos(o); /* frequency_xlator.h:17,28: lv_cmake(cos(offset), sin(offset)) */
    x->di = (float)sin(o);
    x->theta = atan2((double)x->di, (double)x->dr) / (2.0 * DB_M_PI); /* turns per sample of the STORED float phaseDelta */
}
static void xlator_init(orc_xlator* x, double offset, double samplerate) {
    x->pr = 1.0f; x->pi = 0.0f;
    x->ideal = 0; x->acc = 0.0;
    xlator_set_offset(x, offset, samplerate);
}
static void rot_norm(float* pr, float* pi) {
    const float h = hypotf(*pr, *pi);
    *pr = *pr / h;
    *pi = *pi / h;
}
static void xlator_process(orc_xlator* x, int count, const float* in, float* out) {
    if (x->ideal) {
        for (int k = 0; k < count; k++) {
            double ph = x->acc + x->theta * (double)k;
            ph -= floor(ph);
            const float pr = (float)cos(2.0 * DB_M_PI * ph), pi = (float)sin(2.0 * DB_M_PI * ph);
            const float xr = in[2 * k], xi = in[2 * k + 1];
            out[2 * k] = (xr * pr) - (xi * pi);
            out[2 * k + 1] = (xr * pi) + (xi * pr);
        }
        x->acc += x->theta * (double)count;
        x->acc -= floor(x->acc);
        return;
    }
    float pr = x->pr, pi = x->pi;
    const float dr = x->dr, di = x->di;
    int k = 0, i;
    for (i = 0; i < count / 512; i++) {
        for (int j = 0; j < 512; j++, k++) {
            const float xr = in[2 * k], xi = in[2 * k + 1];
            out[2 * k] = (xr * pr) - (xi * pi);
            out[2 * k + 1] = (xr * pi) + (xi * pr);
            const float nr = (pr * dr) - (pi * di);
            const float ni = (pr * di) + (pi * dr);
            pr = nr; pi = ni;
        }
        rot_norm(&pr, &pi);
    }
    for (i = 0; i < count % 512; i++, k++) {
        const float xr = in[2 * k], xi = in[2 * k + 1];
        out[2 * k] = (xr * pr) - (xi * pi);
        out[2 * k + 1] = (xr * pi) + (xi * pr);
        const float nr = (pr * dr) - (pi * di);
        const float ni = (pr * di) + (pi * dr);
        pr = nr; pi = ni;
    }
    if (i) { rot_norm(&pr, &pi); }
    x->pr = pr; x->pi = pi;
}

/* ---- FIR with optional decimation (dsp/filter/fir.h:62-83, decimating_fir.h:45-68); `width` = 2 for
 * complex_t / stereo_t data (volk_32fc_32f_dot_prod_32fc), 1 for float data (volk_32f_x2_dot_prod_32f). ---------- */
typedef struct {
    int ntaps, decim, width, offset;
    float* taps;
    float* buf;  /* (ntaps-1 + cap) * width floats; history first */
    int cap;
} orc_fir;

static void fir_init(orc_fir* f, const float* taps, int ntaps, int decim, int width) {
    memset(f, 0, sizeof(*f));
    f->ntaps = ntaps; f->decim = decim; f->width = width; f->offset = 0;
    f->taps = (float*)malloc(sizeof(float) * (size_t)ntaps);
    memcpy(f->taps, taps, sizeof(float) * (size_t)ntaps);
    f->cap = 0; f->buf = NULL;
}
static void fir_free(orc_fir* f) { free(f->taps); free(f->buf); f->taps = NULL; f->buf = NULL; }
static void fir_reserve(orc_fir* f, int count) {
    if (count <= f->cap && f->buf) { return; }
    const size_t hist = (size_t)(f->ntaps - 1) * f->width;
    float* nb = (float*)calloc(hist + (size_t)count * f->width + 16, sizeof(float));
    if (f->buf) { memcpy(nb, f->buf, hist * sizeof(float)); }
    free(f->buf);
    f->buf = nb;
    f->cap = count;
}
/* FIR::setTaps (dsp/filter/fir.h:31-52): the delay line survives a change of the tap count — fewer taps keep the NEWEST ntaps - 1 samples,
 * more taps put zeros in FRONT of the old ones. */
static void fir_set_taps(orc_fir* f, const float* taps, int ntaps) {
    const int old = f->ntaps, w = f->width;
    if (f->buf) {
        const size_t nh = (size_t)(ntaps - 1) * w;
        float* nb = (float*)calloc(nh + (size_t)f->cap * w + 16, sizeof(float));
        if (ntaps < old) { memcpy(nb, &f->buf[(size_t)(old - ntaps) * w], nh * sizeof(float)); }
        else { memcpy(&nb[(size_t)(ntaps - old) * w], f->buf, (size_t)(old - 1) * w * sizeof(float)); }
        free(f->buf);
        f->buf = nb;
    }
    free(f->taps);
    f->taps = (float*)malloc(sizeof(float) * (size_t)ntaps);
    memcpy(f->taps, taps, sizeof(float) * (size_t)ntaps);
    f->ntaps = ntaps;
}
static void fir_reset(orc_fir* f) {
    if (f->buf) { memset(f->buf, 0, sizeof(float) * (size_t)(f->ntaps - 1) * f->width); }
    f->offset = 0;
}
static int fir_process(orc_fir* f, int count, const float* in, float* out) {
    fir_reserve(f, count);
    const int w = f->width, nt = f->ntaps;
    memcpy(&f->buf[(size_t)(nt - 1) * w], in, sizeof(float) * (size_t)count * w);
    int outCount = 0;
    for (; f->offset < count; f->offset += f->decim) {
        const float* a = &f->buf[(size_t)f->offset * w];
        if (w == 2) {
            float re = 0.0f, im = 0.0f;
            for (int k = 0; k < nt; k++) {
                re += a[2 * k] * f->taps[k];
                im += a[2 * k + 1] * f->taps[k];
            }
            out[2 * outCount] = re;
            out[2 * outCount + 1] = im;
        }
        else {
            float acc = 0.0f;
            for (int k = 0; k < nt; k++) { acc += a[k] * f->taps[k]; }
            out[outCount] = acc;
        }
        outCount++;
    }
    f->offset -= count;
    memmove(f->buf, &f->buf[(size_t)count * w], sizeof(float) * (size_t)(nt - 1) * w);
    return outCount;
}

/* ---- decimation plan tables (numbers extracted from dsp/multirate/decim/plans.h by tools/extract_decim_plans.cpp) -- */
#define ORC_MAX_STAGES 4
typedef struct { int decim, ntaps; float* taps; } orc_stage;
typedef struct { int ratio, nstages; orc_stage stages[ORC_MAX_STAGES]; } orc_plan;
typedef struct { int nplans; orc_plan* plans; } orc_plans;

orc_plans* orc_plans_load(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { return NULL; }
    char magic[4];
    uint32_t ver, n;
    if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "SDPL", 4) || fread(&ver, 4, 1, f) != 1 || fread(&n, 4, 1, f) != 1) { fclose(f); return NULL; }
    orc_plans* p = (orc_plans*)calloc(1, sizeof(orc_plans));
    p->nplans = (int)n;
    p->plans = (orc_plan*)calloc(n, sizeof(orc_plan));
    for (uint32_t i = 0; i < n; i++) {
        uint32_t ratio, ns;
        if (fread(&ratio, 4, 1, f) != 1 || fread(&ns, 4, 1, f) != 1 || ns > ORC_MAX_STAGES) { fclose(f); return NULL; }
        p->plans[i].ratio = (int)ratio;
        p->plans[i].nstages = (int)ns;
        for (uint32_t s = 0; s < ns; s++) {
            uint32_t d, nt;
            if (fread(&d, 4, 1, f) != 1 || fread(&nt, 4, 1, f) != 1) { fclose(f); return NULL; }
            p->plans[i].stages[s].decim = (int)d;
            p->plans[i].stages[s].ntaps = (int)nt;
            p->plans[i].stages[s].taps = (float*)malloc(sizeof(float) * nt);
            if (fread(p->plans[i].stages[s].taps, 4, nt, f) != nt) { fclose(f); return NULL; }
        }
    }
    fclose(f);
    return p;
}
void orc_plans_free(orc_plans* p) {
    if (!p) { return; }
    for (int i = 0; i < p->nplans; i++) {
        for (int s = 0; s < p->plans[i].nstages; s++) { free(p->plans[i].stages[s].taps); }
    }
    free(p->plans);
    free(p);
}
int orc_plans_max_ratio(const orc_plans* p) { return 1 << p->nplans; } /* power_decimator.h:29-31 */

/* ---- PowerDecimator (dsp/multirate/power_decimator.h:51-67, 93-111) ---------------------------------------------- */
typedef struct { int ratio, nstages; orc_fir firs[ORC_MAX_STAGES]; } orc_powdec;

static void powdec_init(orc_powdec* d, const orc_plans* plans, int ratio, int width) {
    memset(d, 0, sizeof(*d));
    d->ratio = ratio;
    if (ratio > 1) {
        int planId = (int)log2(ratio) - 1; /* power_decimator.h:100 */
        const orc_plan* pl = &plans->plans[planId];
        d->nstages = pl->nstages;
        for (int i = 0; i < pl->nstages; i++) { fir_init(&d->firs[i], pl->stages[i].taps, pl->stages[i].ntaps, pl->stages[i].decim, width); }
    }
}
static void powdec_free(orc_powdec* d) {
    for (int i = 0; i < d->nstages; i++) { fir_free(&d->firs[i]); }
    d->nstages = 0;
}
static int powdec_process(orc_powdec* d, int count, const float* in, float* out, int width) {
    if (d->ratio == 1) {
        memmove(out, in, sizeof(float) * (size_t)count * width);
        return count;
    }
    const float* data = in;
    for (int i = 0; i < d->nstages; i++) {
        count = fir_process(&d->firs[i], count, data, out);
        data = out;
    }
    return count;
}

/* ---- PolyphaseResampler (dsp/multirate/polyphase_resampler.h:69-99, polyphase_bank.h:15-47) ----------------------- */
typedef struct {
    int interp, decim, tapsPerPhase, width;
    float** phases;
    int phase, offset;
    float* buf;
    int cap;
} orc_poly;

static void poly_init(orc_poly* r, int interp, int decim, const float* taps, int ntaps, int width) {
    memset(r, 0, sizeof(*r));
    r->interp = interp; r->decim = decim; r->width = width;
    r->tapsPerPhase = (ntaps + interp - 1) / interp;
    r->phases = (float**)calloc((size_t)interp, sizeof(float*));
    for (int i = 0; i < interp; i++) { r->phases[i] = (float*)calloc((size_t)r->tapsPerPhase, sizeof(float)); }
    const int tot = interp * r->tapsPerPhase;
    for (int i = 0; i < tot; i++) { r->phases[(interp - 1) - (i % interp)][i / interp] = (i < ntaps) ? taps[i] : 0; }
}
static void poly_free(orc_poly* r) {
    if (r->phases) {
        for (int i = 0; i < r->interp; i++) { free(r->phases[i]); }
        free(r->phases);
    }
    free(r->buf);
    memset(r, 0, sizeof(*r));
}
static void poly_reserve(orc_poly* r, int count) {
    if (count <= r->cap && r->buf) { return; }
    const size_t hist = (size_t)(r->tapsPerPhase - 1) * r->width;
    float* nb = (float*)calloc(hist + (size_t)count * r->width + 16, sizeof(float));
    if (r->buf) { memcpy(nb, r->buf, hist * sizeof(float)); }
    free(r->buf);
    r->buf = nb;
    r->cap = count;
}
static int poly_process(orc_poly* r, int count, const float* in, float* out) {
    poly_reserve(r, count);
    const int w = r->width, tpp = r->tapsPerPhase;
    /* `in` may alias `out` (rational_resampler.h:85: resamp.process(count, out, out)): copy first, like the reference */
    memcpy(&r->buf[(size_t)(tpp - 1) * w], in, sizeof(float) * (size_t)count * w);
    int outCount = 0;
    while (r->offset < count) {
        const float* a = &r->buf[(size_t)r->offset * w];
        const float* t = r->phases[r->phase];
        if (w == 2) {
            float re = 0.0f, im = 0.0f;
            for (int k = 0; k < tpp; k++) {
                re += a[2 * k] * t[k];
                im += a[2 * k + 1] * t[k];
            }
            out[2 * outCount] = re;
            out[2 * outCount + 1] = im;
        }
        else {
            float acc = 0.0f;
            for (int k = 0; k < tpp; k++) { acc += a[k] * t[k]; }
            out[outCount] = acc;
        }
        outCount++;
        r->phase += r->decim;
        r->offset += r->phase / r->interp;
        r->phase = r->phase % r->interp;
    }
    r->offset -= count;
    memmove(r->buf, &r->buf[(size_t)count * w], sizeof(float) * (size_t)(tpp - 1) * w);
    return outCount;
}

/* ---- RationalResampler (dsp/multirate/rational_resampler.h:80-165) ------------------------------------------------- */
enum { RR_BOTH, RR_DECIM_ONLY, RR_RESAMP_ONLY, RR_NONE };
typedef struct {
    int mode, width;
    int predecRatio, interp, decim, ntaps;
    orc_powdec dec;
    orc_poly res;
    float* rtaps;
} orc_rresamp;

static long long gcd_ll(long long a, long long b) {
    a = a < 0 ? -a : a; b = b < 0 ? -b : b;
    while (b) { long long t = a % b; a = b; b = t; }
    return a;
}
static void rresamp_init(orc_rresamp* r, const orc_plans* plans, double inSR, double outSR, int width) {
    memset(r, 0, sizeof(*r));
    r->width = width;
    const int maxRatio = orc_plans_max_ratio(plans);
    /* rational_resampler.h:122-123 (note: the power is clamped against the max RATIO, as in the reference) */
    int predecPower = (int)floor(log2(inSR / outSR));
    if (predecPower > maxRatio) { predecPower = maxRatio; }
    int predecRatio = (predecPower >= 0 && predecPower < 31) ? (1 << predecPower) : maxRatio;
    if (predecRatio > maxRatio) { predecRatio = maxRatio; }
    double intSR = inSR;
    const int useDecim = (inSR > outSR && predecPower > 0);
    r->predecRatio = 1;
    if (useDecim) {
        intSR = inSR / (double)predecRatio;
        r->predecRatio = predecRatio;
        powdec_init(&r->dec, plans, predecRatio, width);
    }
    const int IntSR = (int)round(intSR);
    const int OutSR = (int)round(outSR);
    const int g = (int)gcd_ll(IntSR, OutSR);
    const int interp = OutSR / g;
    const int decim = IntSR / g;
    r->interp = interp; r->decim = decim;
    if (interp == decim) {
        r->mode = useDecim ? RR_DECIM_ONLY : RR_NONE;
        return;
    }
    const double tapSamplerate = intSR * (double)interp;
    const double tapBandwidth = (inSR < outSR ? inSR : outSR) / 2.0;
    const double tapTransWidth = tapBandwidth * 0.1;
    int n = orc_estimate_tap_count(tapTransWidth, tapSamplerate);
    r->rtaps = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    r->ntaps = orc_low_pass(tapBandwidth, tapTransWidth, tapSamplerate, 0, r->rtaps, n);
    for (int i = 0; i < r->ntaps; i++) { r->rtaps[i] *= (float)interp; } /* rational_resampler.h:159 */
    poly_init(&r->res, interp, decim, r->rtaps, r->ntaps, width);
    r->mode = useDecim ? RR_BOTH : RR_RESAMP_ONLY;
}
static void rresamp_free(orc_rresamp* r) {
    powdec_free(&r->dec);
    poly_free(&r->res);
    free(r->rtaps);
    r->rtaps = NULL;
}
static int rresamp_process(orc_rresamp* r, int count, const float* in, float* out) {
    switch (r->mode) {
    case RR_BOTH:
        count = powdec_process(&r->dec, count, in, out, r->width);
        return poly_process(&r->res, count, out, out);
    case RR_DECIM_ONLY:
        return powdec_process(&r->dec, count, in, out, r->width);
    case RR_RESAMP_ONLY:
        return poly_process(&r->res, count, in, out);
    default:
        memmove(out, in, sizeof(float) * (size_t)count * r->width);
        return count;
    }
}

/* Stand-alone handle so tests / the AF chain ("next" row) can drive a RationalResampler<stereo_t|complex_t|float>. */
typedef struct { orc_rresamp r; float* tmp; int cap; } orc_resampler;
orc_resampler* orc_resampler_create(const orc_plans* plans, double inSR, double outSR, int width) {
    orc_resampler* h = (orc_resampler*)calloc(1, sizeof(orc_resampler));
    rresamp_init(&h->r, plans, inSR, outSR, width);
    return h;
}
void orc_resampler_destroy(orc_resampler* h) { if (h) { rresamp_free(&h->r); free(h->tmp); free(h); } }
int orc_resampler_process(orc_resampler* h, int count, const float* in, float* out) {
    /* output may be larger than input when interp > decim: caller sizes `out` accordingly */
    return rresamp_process(&h->r, count, in, out);
}
void orc_resampler_info(const orc_resampler* h, int* mode, int* predec, int* interp, int* decim, int* ntaps, int* tapsPerPhase) {
    *mode = h->r.mode; *predec = h->r.predecRatio; *interp = h->r.interp; *decim = h->r.decim; *ntaps = h->r.ntaps;
    *tapsPerPhase = h->r.res.tapsPerPhase;
}

/* ---- RxVFO (dsp/channel/rx_vfo.h:19-36, 89-100, 117-121) --------------------------------------------------------- */
typedef struct {
    orc_xlator xl;
    orc_rresamp rs;
    orc_fir chan;
    int filterNeeded, chanTaps;
    double inSR, outSR, bandwidth, offset;
} orc_rxvfo;

orc_rxvfo* orc_rxvfo_create(const orc_plans* plans, double inSR, double outSR, double bandwidth, double offset) {
    orc_rxvfo* v = (orc_rxvfo*)calloc(1, sizeof(orc_rxvfo));
    v->inSR = inSR; v->outSR = outSR; v->bandwidth = bandwidth; v->offset = offset;
    v->filterNeeded = (bandwidth != outSR);
    xlator_init(&v->xl, -offset, inSR); /* rx_vfo.h:27 */
    rresamp_init(&v->rs, plans, inSR, outSR, 2);
    {   /* generateTaps, rx_vfo.h:117-121 — built even when the filter is bypassed */
        const double filterWidth = bandwidth / 2.0;
        int n = orc_estimate_tap_count(filterWidth * 0.1, outSR);
        float* t = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
        n = orc_low_pass(filterWidth, filterWidth * 0.1, outSR, 0, t, n);
        fir_init(&v->chan, t, n, 1, 2);
        v->chanTaps = n;
        free(t);
    }
    return v;
}
void orc_rxvfo_destroy(orc_rxvfo* v) {
    if (!v) { return; }
    rresamp_free(&v->rs);
    fir_free(&v->chan);
    free(v);
}
void orc_rxvfo_set_offset(orc_rxvfo* v, double offset) { /* rx_vfo.h:72-77: only phaseDelta changes */
    v->offset = offset;
    xlator_set_offset(&v->xl, -offset, v->inSR);
}
void orc_rxvfo_set_bandwidth(orc_rxvfo* v, double bandwidth) { /* rx_vfo.h:60-70: taps regenerated only when the filter is needed; FIR::setTaps keeps the delay line */
    v->bandwidth = bandwidth;
    v->filterNeeded = (bandwidth != v->outSR);
    if (!v->filterNeeded) { return; }
    const double filterWidth = bandwidth / 2.0; /* generateTaps, rx_vfo.h:117-121 */
    int n = orc_estimate_tap_count(filterWidth * 0.1, v->outSR);
    float* t = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    n = orc_low_pass(filterWidth, filterWidth * 0.1, v->outSR, 0, t, n);
    fir_set_taps(&v->chan, t, n);
    v->chanTaps = n;
    free(t);
}
/* `out` must hold `count` complex samples (it is used as the in-place work buffer, like the reference). */
int orc_rxvfo_process(orc_rxvfo* v, int count, const float* in, float* out) {
    xlator_process(&v->xl, count, in, out);
    if (!v->filterNeeded) { return rresamp_process(&v->rs, count, out, out); }
    count = rresamp_process(&v->rs, count, out, out);
    /* FIR::process with decimation 1: memcpy into its own buffer first, so in-place is safe */
    v->chan.offset = 0;
    fir_process(&v->chan, count, out, out);
    v->chan.offset = 0;
    return count;
}
void orc_rxvfo_info(const orc_rxvfo* v, int* mode, int* predec, int* interp, int* decim, int* rtaps, int* tapsPerPhase, int* chanTaps, int* filterNeeded) {
    *mode = v->rs.mode; *predec = v->rs.predecRatio; *interp = v->rs.interp; *decim = v->rs.decim; *rtaps = v->rs.ntaps;
    *tapsPerPhase = v->rs.res.tapsPerPhase; *chanTaps = v->chanTaps; *filterNeeded = v->filterNeeded;
}
void orc_rxvfo_phase_delta(const orc_rxvfo* v, float* dr, float* di) { *dr = v->xl.dr; *di = v->xl.di; }
/* test switch, see orc_xlator */
void orc_rxvfo_set_ideal_nco(orc_rxvfo* v, int on) { v->xl.ideal = on; }

/* ===================================================================================================================
 * 5. Demodulators
 * =================================================================================================================== */

/* dsp/math/normalize_phase.h:6-9 (float constant, single wrap) */
static float normalize_phase(float diff) {
    if (diff > FL_M_PI) { diff -= 2.0f * FL_M_PI; }
    else if (diff <= -FL_M_PI) { diff += 2.0f * FL_M_PI; }
    return diff;
}

/* dsp/demod/quadrature.h:39-46 */
typedef struct { float invDeviation, phase; } orc_quad;
static void quad_init(orc_quad* q, double deviation, double samplerate) {
    q->invDeviation = (float)(1.0 / hz_to_rads(deviation, samplerate));
    q->phase = 0.0f;
}
static void quad_process(orc_quad* q, int count, const float* in, float* out) {
    for (int i = 0; i < count; i++) {
        float cphase = atan2f(in[2 * i + 1], in[2 * i]); /* types.h:57-59 */
        out[i] = normalize_phase(cphase - q->phase) * q->invDeviation;
        q->phase = cphase;
    }
}

/* dsp/loop/agc.h:15-27, 70-109 (T = float) */
typedef struct { float setPoint, attack, invAttack, decay, invDecay, maxGain, maxOutputAmp, initGain, amp; } orc_agc;
static void agc_init(orc_agc* a, double setPoint, double attack, double decay, double maxGain, double maxOutputAmp, double initGain) {
    a->setPoint = setPoint;
    a->attack = attack;
    a->invAttack = 1.0f - a->attack;
    a->decay = decay;
    a->invDecay = 1.0f - a->decay;
    a->maxGain = maxGain;
    a->maxOutputAmp = maxOutputAmp;
    a->initGain = initGain;
    a->amp = a->setPoint / a->initGain;
}
static void agc_process_f(orc_agc* a, int count, const float* in, float* out) {
    for (int i = 0; i < count; i++) {
        float inAmp, gain;
        inAmp = fabsf(in[i]);
        if (inAmp != 0.0f) {
            a->amp = (inAmp > a->amp) ? ((a->amp * a->invAttack) + (inAmp * a->attack)) : ((a->amp * a->invDecay) + (inAmp * a->decay));
            { const float g_ = a->setPoint / a->amp; gain = (a->maxGain < g_) ? a->maxGain : g_; } /* std::min<float> */
        }
        else {
            gain = 1.0f;
        }
        if (inAmp * gain > a->maxOutputAmp) {
            float maxAmp = 0;
            for (int j = i; j < count; j++) {
                inAmp = fabsf(in[j]);
                if (inAmp > maxAmp) { maxAmp = inAmp; }
            }
            a->amp = maxAmp;
            { const float g_ = a->setPoint / a->amp; gain = (a->maxGain < g_) ? a->maxGain : g_; } /* std::min<float> */
        }
        out[i] = in[i] * gain;
    }
}
/* T = complex_t (carrier AGC, am.h:103-106) */
static void agc_process_c(orc_agc* a, int count, const float* in, float* out) {
    for (int i = 0; i < count; i++) {
        float inAmp, gain;
        inAmp = sqrt((in[2 * i] * in[2 * i]) + (in[2 * i + 1] * in[2 * i + 1])); /* types.h:81-83: double sqrt of a float */
        if (inAmp != 0.0f) {
            a->amp = (inAmp > a->amp) ? ((a->amp * a->invAttack) + (inAmp * a->attack)) : ((a->amp * a->invDecay) + (inAmp * a->decay));
            { const float g_ = a->setPoint / a->amp; gain = (a->maxGain < g_) ? a->maxGain : g_; } /* std::min<float> */
        }
        else {
            gain = 1.0f;
        }
        if (inAmp * gain > a->maxOutputAmp) {
            float maxAmp = 0;
            for (int j = i; j < count; j++) {
                inAmp = sqrt((in[2 * j] * in[2 * j]) + (in[2 * j + 1] * in[2 * j + 1]));
                if (inAmp > maxAmp) { maxAmp = inAmp; }
            }
            a->amp = maxAmp;
            { const float g_ = a->setPoint / a->amp; gain = (a->maxGain < g_) ? a->maxGain : g_; } /* std::min<float> */
        }
        out[2 * i] = in[2 * i] * gain;
        out[2 * i + 1] = in[2 * i + 1] * gain;
    }
}

/* dsp/correction/dc_blocker.h:54-60 (T = float) */
typedef struct { float rate, offset; } orc_dcblock;
static void dcblock_process(orc_dcblock* d, int count, const float* in, float* out) {
    for (int i = 0; i < count; i++) {
        out[i] = in[i] - d->offset;
        d->offset += out[i] * d->rate;
    }
}

/* Demodulator modes, mirroring decoder_modules/radio/src/demodulators/{wfm,nfm,am,usb,lsb,dsb}.h defaults. */
enum { ORC_WFM = 0, ORC_NFM = 1, ORC_AM = 2, ORC_USB = 3, ORC_LSB = 4, ORC_DSB = 5 };

typedef struct {
    int mode;
    double bandwidth, samplerate;
    orc_quad quad;
    orc_fir audio;      /* WFM: alFir (broadcast_fm.h:49,206-208); NFM: fir (fm.h:148-149); AM: lpf (am.h:34,116-118) */
    int audioTaps, lowPass;
    orc_agc agc, carrierAgc;
    int carrierMode;
    orc_dcblock dc;
    orc_xlator xl;      /* SSB second translation (ssb.h:24,78,106-117) */
    float* tmp;
    float* tmp2;
    int cap;
} orc_demod;

static void demod_reserve(orc_demod* d, int count) {
    if (count <= d->cap) { return; }
    free(d->tmp); free(d->tmp2);
    d->tmp = (float*)malloc(sizeof(float) * 2 * (size_t)count);
    d->tmp2 = (float*)malloc(sizeof(float) * 2 * (size_t)count);
    d->cap = count;
}

/* agcAttack / agcDecay are the radio module's UI values (default 50 / 5, am.h:98-99, usb.h:92-93); they are divided by
 * the IF rate exactly as the radio module does (am.h:34, usb.h:34).  lowPass applies to WFM/NFM (default on). */
orc_demod* orc_demod_create(int mode, double bandwidth, double ifSamplerate, int lowPass, double agcAttack, double agcDecay, int carrierAgc) {
    orc_demod* d = (orc_demod*)calloc(1, sizeof(orc_demod));
    d->mode = mode; d->bandwidth = bandwidth; d->samplerate = ifSamplerate; d->lowPass = lowPass;
    float one = 1.0f;
    if (mode == ORC_WFM) {
        /* wfm.h:78: demod.init(input, bandwidth / 2.0f, getIFSampleRate(), _stereo, _lowPass, _rds); mono path */
        quad_init(&d->quad, bandwidth / 2.0f, ifSamplerate);
        int n = orc_estimate_tap_count(4000.0, ifSamplerate);
        float* t = (float*)malloc(sizeof(float) * (size_t)n);
        n = orc_low_pass(15000.0, 4000.0, ifSamplerate, 0, t, n); /* broadcast_fm.h:49 */
        fir_init(&d->audio, t, n, 1, 1);
        d->audioTaps = n;
        free(t);
    }
    else if (mode == ORC_NFM) {
        /* nfm.h:29 -> fm.h:27-45: deviation bandwidth/2; LPF lowPass(bw/2, (bw/2)*0.1, sr) or a single unit tap */
        quad_init(&d->quad, bandwidth / 2.0, ifSamplerate);
        if (lowPass) {
            int n = orc_estimate_tap_count((bandwidth / 2.0) * 0.1, ifSamplerate);
            float* t = (float*)malloc(sizeof(float) * (size_t)n);
            n = orc_low_pass(bandwidth / 2.0, (bandwidth / 2.0) * 0.1, ifSamplerate, 0, t, n);
            fir_init(&d->audio, t, n, 1, 1);
            d->audioTaps = n;
            free(t);
        }
        else {
            fir_init(&d->audio, &one, 1, 1, 1);
            d->audioTaps = 1;
        }
    }
    else if (mode == ORC_AM) {
        /* am.h:34 (radio) -> demod/am.h:26-45 */
        const double att = agcAttack / ifSamplerate, dec = agcDecay / ifSamplerate;
        agc_init(&d->carrierAgc, 1.0, att, dec, 10e6, 10.0, INFINITY);
        agc_init(&d->agc, 1.0, att, dec, 10e6, 10.0, INFINITY);
        d->carrierMode = carrierAgc;
        d->dc.rate = (float)(100.0 / ifSamplerate);
        d->dc.offset = 0.0f;
        int n = orc_estimate_tap_count((bandwidth / 2.0) * 0.1, ifSamplerate);
        float* t = (float*)malloc(sizeof(float) * (size_t)n);
        n = orc_low_pass(bandwidth / 2.0, (bandwidth / 2.0) * 0.1, ifSamplerate, 0, t, n);
        fir_init(&d->audio, t, n, 1, 1);
        d->audioTaps = n;
        free(t);
    }
    else {
        /* usb.h:34 / lsb.h / dsb.h -> ssb.h:22-33, 106-117 */
        double tr = 0.0;
        if (mode == ORC_USB) { tr = bandwidth / 2.0; }
        else if (mode == ORC_LSB) { tr = -bandwidth / 2.0; }
        xlator_init(&d->xl, tr, ifSamplerate);
        agc_init(&d->agc, 1.0, agcAttack / ifSamplerate, agcDecay / ifSamplerate, 10e6, 10.0, INFINITY);
    }
    return d;
}
void orc_demod_destroy(orc_demod* d) {
    if (!d) { return; }
    if (d->audio.taps) { fir_free(&d->audio); }
    free(d->tmp); free(d->tmp2); free(d);
}
int orc_demod_audio_taps(const orc_demod* d) { return d->audioTaps; }
void orc_demod_set_ideal_nco(orc_demod* d, int on) { d->xl.ideal = on; } /* SSB second translation; test switch, see orc_xlator */

/* in: count complex IF samples; out: count stereo_t samples (l, r interleaved). */
int orc_demod_process(orc_demod* d, int count, const float* in, float* out) {
    demod_reserve(d, count);
    float* m = d->tmp;
    if (d->mode == ORC_WFM || d->mode == ORC_NFM) {
        /* broadcast_fm.h:146,205-211 (mono branch) / fm.h:88-94 */
        quad_process(&d->quad, count, in, m);
        if (d->lowPass) {
            d->audio.offset = 0;
            fir_process(&d->audio, count, m, m);
            d->audio.offset = 0;
        }
    }
    else if (d->mode == ORC_AM) {
        /* demod/am.h:101-131 (T = stereo_t) */
        const float* src = in;
        if (d->carrierMode) {
            agc_process_c(&d->carrierAgc, count, in, d->tmp2);
            src = d->tmp2;
        }
        for (int i = 0; i < count; i++) {
            const float re = src[2 * i], im = src[2 * i + 1];
            m[i] = sqrtf((re * re) + (im * im)); /* volk_32fc_magnitude_32f */
        }
        dcblock_process(&d->dc, count, m, m);
        if (!d->carrierMode) { agc_process_f(&d->agc, count, m, m); }
        d->audio.offset = 0;
        fir_process(&d->audio, count, m, m);
        d->audio.offset = 0;
    }
    else {
        /* ssb.h:77-92 */
        xlator_process(&d->xl, count, in, d->tmp2);
        for (int i = 0; i < count; i++) { m[i] = d->tmp2[2 * i]; } /* ComplexToReal */
        agc_process_f(&d->agc, count, m, m);
    }
    for (int i = count - 1; i >= 0; i--) { /* MonoToStereo / LRToStereo(x, x) */
        const float v = m[i];
        out[2 * i] = v;
        out[2 * i + 1] = v;
    }
    return count;
}

/* ===================================================================================================================
 * 6. Neighbouring rows ("next" in SURVEY.md §8f): AF-chain de-emphasis and file_source sample conversion
 * =================================================================================================================== */

/* dsp/filter/deephasis.h:58-77, 90-93 (T = stereo_t) */
typedef struct { float alpha, lastL, lastR; } orc_deemp;
orc_deemp* orc_deemp_create(double tau, double samplerate) {
    orc_deemp* d = (orc_deemp*)calloc(1, sizeof(orc_deemp));
    float dt = 1.0f / samplerate;
    d->alpha = dt / (tau + dt);
    return d;
}
void orc_deemp_destroy(orc_deemp* d) { free(d); }
void orc_deemp_process(orc_deemp* d, int count, const float* in, float* out) {
    if (count <= 0) { return; }
    const float alpha = d->alpha;
    out[0] = (alpha * in[0]) + ((1 - alpha) * d->lastL);
    out[1] = (alpha * in[1]) + ((1 - alpha) * d->lastR);
    for (int i = 1; i < count; i++) {
        out[2 * i] = (alpha * in[2 * i]) + ((1 - alpha) * out[2 * (i - 1)]);
        out[2 * i + 1] = (alpha * in[2 * i + 1]) + ((1 - alpha) * out[2 * (i - 1) + 1]);
    }
    d->lastL = out[2 * (count - 1)];
    d->lastR = out[2 * (count - 1) + 1];
}

/* Sink-side sample packing.
 *   dsp/compression/sample_stream_compressor.h:30-62 (SDR++ server wire format): [u16 0][u16 pcmType][f32 scaler][data];
 *     F32: scaler 0, raw copy; I8 / I16: scaler = value at volk_32f_index_max_32u (the largest VALUE, not magnitude), samples
 *     converted with scale 128 / scaler resp. 32768 / scaler.
 *   utils/wav.cpp:166 (recorder, int16 WAV): volk_32f_s32f_convert_16i(buf, samples, 32767.0f, n).
 * VOLK generic conversions: r = x * scalar, clamped to the integer range, rintf, cast. */
void orc_convert_16i(const float* in, float scalar, int n, int16_t* out) {
    for (int i = 0; i < n; i++) {
        float r = in[i] * scalar;
        if (r > 32767.0f) { r = 32767.0f; }
        else if (r < -32768.0f) { r = -32768.0f; }
        out[i] = (int16_t)rintf(r);
    }
}
void orc_convert_8i(const float* in, float scalar, int n, int8_t* out) {
    for (int i = 0; i < n; i++) {
        float r = in[i] * scalar;
        if (r > 127.0f) { r = 127.0f; }
        else if (r < -128.0f) { r = -128.0f; }
        out[i] = (int8_t)rintf(r);
    }
}
/* pcmType: 0 I8, 1 I16, 2 F32 (dsp/compression/pcm_type.h); in: count complex samples; returns bytes written */
int orc_compress(int count, int pcmType, const float* in, uint8_t* out) {
    uint16_t* compressionType = (uint16_t*)out;
    uint16_t* sampleType = (uint16_t*)&out[2];
    float* scaler = (float*)&out[4];
    void* dataBuf = &out[8];
    *compressionType = 0;
    *sampleType = (uint16_t)pcmType;
    if (pcmType == 2) {
        *scaler = 0;
        memcpy(dataBuf, in, (size_t)count * 8);
        return 8 + count * 8;
    }
    float maxVal = in[0];
    for (int i = 1; i < count * 2; i++) {
        if (in[i] > maxVal) { maxVal = in[i]; }
    }
    *scaler = maxVal;
    if (pcmType == 0) {
        orc_convert_8i(in, 128.0f / maxVal, count * 2, (int8_t*)dataBuf);
        return 8 + count * 2;
    }
    orc_convert_16i(in, 32768.0f / maxVal, count * 2, (int16_t*)dataBuf);
    return 8 + count * 4;
}

/* WaterFall display state around the raw-line ring (gui/widgets/waterfall.cpp, waterfallVisible == true):
 *   getFFTBuffer  :875-886   currentFFTLine--, wrap, fftLines = min(fftLines + 1, waterfallHeight); the handler writes the raw line there
 *   pushFFT       :888-941   doZoom of that line -> latestFFT; palette index row; FFT smoothing (:913-920, three separately rounded
 *                            VOLK passes); FFT hold (:935-939, starts at index 1)
 *   updateWaterfallFb :600-631  re-zoom of every stored line, newest first, rows beyond fftLines opaque black (returned as -1)
 *   setFFTSmoothing :1166-1188 (buffer starts as a copy of latestFFT), setFFTSmoothingSpeed :1190-1194, setFFTHold :1153-1160 (-1000) */
typedef struct {
    int height, N, dataWidth;
    float* raw;       /* [height][N] */
    int currentFFTLine, fftLines;
    float* latest;    /* [dataWidth] */
    float* smoothing; /* NULL = off */
    float* hold;
    int holdOn;
    float alpha, beta, holdSpeed;
} orc_wf;
orc_wf* orc_wf_create(int height, int N, int dataWidth) {
    orc_wf* w = (orc_wf*)calloc(1, sizeof(orc_wf));
    w->height = height;
    w->N = N;
    w->dataWidth = dataWidth;
    w->raw = (float*)calloc((size_t)height * N, sizeof(float));
    w->latest = (float*)calloc((size_t)dataWidth, sizeof(float));
    w->hold = (float*)calloc((size_t)dataWidth, sizeof(float));
    return w;
}
void orc_wf_destroy(orc_wf* w) {
    if (!w) { return; }
    free(w->raw);
    free(w->latest);
    free(w->smoothing);
    free(w->hold);
    free(w);
}
void orc_wf_set_smoothing(orc_wf* w, int enabled, float speed) {
    free(w->smoothing);
    w->smoothing = NULL;
    if (enabled) {
        w->smoothing = (float*)malloc(sizeof(float) * (size_t)w->dataWidth);
        memcpy(w->smoothing, w->latest, sizeof(float) * (size_t)w->dataWidth);
    }
    w->alpha = speed;
    w->beta = 1.0f - speed;
}
void orc_wf_set_hold(orc_wf* w, int enabled, float speed) {
    w->holdOn = enabled;
    if (enabled) {
        for (int i = 0; i < w->dataWidth; i++) { w->hold[i] = -1000.0; }
    }
    w->holdSpeed = speed;
}
/* one finished raw line: index row of the new top line goes to idx[dataWidth] */
void orc_wf_push(orc_wf* w, const float* line, int drawDataStart, int drawDataSize, float wmin, float wmax, int32_t* idx) {
    w->currentFFTLine--;
    w->fftLines++;
    w->currentFFTLine = ((w->currentFFTLine + w->height) % w->height);
    if (w->fftLines > w->height) { w->fftLines = w->height; }
    float* slot = &w->raw[(size_t)w->currentFFTLine * w->N];
    memcpy(slot, line, sizeof(float) * (size_t)w->N);
    orc_do_zoom(drawDataStart, drawDataSize, w->N, w->dataWidth, slot, w->latest);
    orc_palette_index(w->latest, w->dataWidth, wmin, wmax, idx);
    if (w->smoothing) {
        for (int i = 0; i < w->dataWidth; i++) { w->latest[i] = w->latest[i] * w->alpha; }
        for (int i = 0; i < w->dataWidth; i++) { w->smoothing[i] = w->smoothing[i] * w->beta; }
        for (int i = 0; i < w->dataWidth; i++) { w->smoothing[i] = w->smoothing[i] + w->latest[i]; }
        memcpy(w->latest, w->smoothing, sizeof(float) * (size_t)w->dataWidth);
    }
    if (w->holdOn) {
        for (int i = 1; i < w->dataWidth; i++) {
            const float a = w->latest[i], b = w->hold[i] - w->holdSpeed;
            w->hold[i] = (a < b) ? b : a; /* std::max<float>(a, b) */
        }
    }
}
/* calculateVFOSignalInfo (waterfall.cpp:558-598) on the newest stored line: strength = max over the VFO's bins, snr = max - mean of the
 * two side bands (one bandwidth wide in total), accumulated in double in bin order.  Returns 0 when no line is stored. */
int orc_wf_signal_info(const orc_wf* w, double centerOffset, double bandwidth, double wholeBandwidth, float* strength, float* snr) {
    if (w->fftLines <= 0) { return 0; }
    const float* fftLine = &w->raw[(size_t)w->currentFFTLine * w->N];
    const int rawFFTSize = w->N;
    double vfoMinSizeFreq = centerOffset - bandwidth;
    double vfoMinFreq = centerOffset - (bandwidth / 2.0);
    double vfoMaxFreq = centerOffset + (bandwidth / 2.0);
    double vfoMaxSizeFreq = centerOffset + bandwidth;
    int off[4];
    const double f[4] = { vfoMinSizeFreq, vfoMinFreq, vfoMaxFreq, vfoMaxSizeFreq };
    for (int i = 0; i < 4; i++) {
        int v = (int)(((f[i] / (wholeBandwidth / 2.0)) * (double)(rawFFTSize / 2)) + (rawFFTSize / 2));
        off[i] = v < 0 ? 0 : (v > rawFFTSize ? rawFFTSize : v); /* std::clamp<int>(.., 0, rawFFTSize) */
    }
    double avg = 0;
    float max = -INFINITY;
    int avgCount = 0;
    for (int i = off[0]; i < off[1]; i++) { avg += fftLine[i]; avgCount++; }
    for (int i = off[2] + 1; i < off[3]; i++) { avg += fftLine[i]; avgCount++; }
    avg /= (double)(avgCount);
    for (int i = off[1]; i <= off[2]; i++) {
        if (fftLine[i] > max) { max = fftLine[i]; }
    }
    *strength = max;
    *snr = max - avg;
    return 1;
}
void orc_wf_latest(const orc_wf* w, float* latest, float* hold) {
    if (latest) { memcpy(latest, w->latest, sizeof(float) * (size_t)w->dataWidth); }
    if (hold) { memcpy(hold, w->hold, sizeof(float) * (size_t)w->dataWidth); }
}
/* fb[height][dataWidth]: palette indices, -1 where the reference writes opaque black; returns the number of stored lines */
int orc_wf_raster(const orc_wf* w, int drawDataStart, int drawDataSize, float wmin, float wmax, int32_t* fb) {
    float* tmp = (float*)malloc(sizeof(float) * (size_t)w->dataWidth);
    const int count = (w->fftLines < w->height) ? w->fftLines : w->height;
    for (int i = 0; i < count; i++) {
        orc_do_zoom(drawDataStart, drawDataSize, w->N, w->dataWidth, &w->raw[(size_t)((i + w->currentFFTLine) % w->height) * w->N], tmp);
        orc_palette_index(tmp, w->dataWidth, wmin, wmax, &fb[(size_t)i * w->dataWidth]);
    }
    for (int i = count; i < w->height; i++) {
        for (int j = 0; j < w->dataWidth; j++) { fb[(size_t)i * w->dataWidth + j] = -1; }
    }
    free(tmp);
    return count;
}

/* IQFrontEnd pre-processing chain (core/src/signal_path/iq_frontend.cpp:32-39): PowerDecimator<complex_t> (enabled when the ratio
 * is > 1) -> DCBlocker<complex_t> (dsp/correction/dc_blocker.h:54-60, rate genDCBlockRate(effectiveSr) = 50 / effectiveSr,
 * iq_frontend.h:55-57) -> Conjugate (dsp/math/conjugate.h:12-15, volk_32fc_conjugate_32fc).  SURVEY.md 8f row 2. */
typedef struct { orc_powdec dec; int dc_on, conj_on; float rate, offRe, offIm; } orc_preproc;
orc_preproc* orc_preproc_create(const orc_plans* plans, int ratio, int dcBlocking, double dcRate, int conjugate) {
    orc_preproc* p = (orc_preproc*)calloc(1, sizeof(orc_preproc));
    powdec_init(&p->dec, plans, ratio, 2);
    p->dc_on = dcBlocking;
    p->rate = (float)dcRate; /* DCBlocker::init(in, double rate) stores a float _rate (dc_blocker.h:13-27) */
    p->conj_on = conjugate;
    return p;
}
void orc_preproc_destroy(orc_preproc* p) { if (p) { powdec_free(&p->dec); free(p); } }
/* out must hold `count` complex samples; returns the number produced */
int orc_preproc_process(orc_preproc* p, int count, const float* in, float* out) {
    int n = powdec_process(&p->dec, count, in, out, 2);
    if (p->dc_on) {
        for (int i = 0; i < n; i++) { /* out[i] = in[i] - offset; offset += out[i] * _rate (complex_t operators, types.h) */
            const float re = out[2 * i] - p->offRe, im = out[2 * i + 1] - p->offIm;
            out[2 * i] = re;
            out[2 * i + 1] = im;
            p->offRe += re * p->rate;
            p->offIm += im * p->rate;
        }
    }
    if (p->conj_on) {
        for (int i = 0; i < n; i++) { out[2 * i + 1] = -out[2 * i + 1]; }
    }
    return n;
}

/* source_modules/file_source/src/main.cpp:162: volk_16i_s32f_convert_32f(out, in, 32768.0f, 2*count) */
void orc_int16_to_float(const int16_t* in, float* out, int n) {
    const float iScalar = 1.0f / 32768.0f;
    for (int i = 0; i < n; i++) { out[i] = ((float)in[i]) * iScalar; }
}

/* stand-alone streaming FIR / decimating FIR / rotator handles for unit tests */
orc_fir* orc_fir_create(const float* taps, int ntaps, int decim, int width) {
    orc_fir* f = (orc_fir*)malloc(sizeof(orc_fir));
    fir_init(f, taps, ntaps, decim, width);
    return f;
}
void orc_fir_destroy(orc_fir* f) { if (f) { fir_free(f); free(f); } }
int orc_fir_process(orc_fir* f, int count, const float* in, float* out) { return fir_process(f, count, in, out); }
void orc_fir_reset(orc_fir* f) { fir_reset(f); }

orc_xlator* orc_xlator_create(double offset, double samplerate) {
    orc_xlator* x = (orc_xlator*)malloc(sizeof(orc_xlator));
    xlator_init(x, offset, samplerate);
    return x;
}
void orc_xlator_destroy(orc_xlator* x) { free(x); }
void orc_xlator_set_ideal(orc_xlator* x, int on) { x->ideal = on; }
void orc_xlator_process(orc_xlator* x, int count, const float* in, float* out) { xlator_process(x, count, in, out); }
void orc_xlator_state(const orc_xlator* x, float* pr, float* pi, float* dr, float* di) { *pr = x->pr; *pi = x->pi; *dr = x->dr; *di = x->di; }
