// ORACLE / TEST INFRASTRUCTURE ONLY — never linked into the product library.
//
// Restatement of the libvolk kernels the SDR++ hot path calls.  libvolk is NOT vendored in the
// reference and is not installed here (core/CMakeLists.txt:124 `pkg_check_modules(VOLK REQUIRED volk)`,
// unpinned; docker_builds/ubuntu_jammy ships libvolk2-dev 2.5.x).  Each function below restates the
// *generic* (scalar, protokernel "_generic") semantics published in VOLK's kernel headers
// (kernels/volk/volk_<name>.h), which is the behaviour every SIMD protokernel is tested against by
// VOLK's own QA within tolerance.  With this header on the include path the reference's own DSP headers
// (-I/root/reference/core/src) compile unmodified; see oracle/ref_api.cpp.
//
// Two build flavours:
//   default              : strict sequential fp32 sums  -> bit-exact ground truth for oracle/oracle.c
//   -DSDRPP_SHIM_SIMD    : dot products as vectorised reductions (lane-partial sums, what VOLK's AVX/AVX-512
//                          protokernels do), compiled -O3 -march=native -fopenmp-simd; used only for CPU-baseline TIMING.
#pragma once
#include <complex>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstring>

#define VOLK_VERSION 030100  // >= 3.1: frequency_xlator.h:44 takes the rotator2 branch

typedef std::complex<float> lv_32fc_t;
#define lv_cmake(r, i) lv_32fc_t((float)(r), (float)(i))
#define lv_creal(x) ((x).real())
#define lv_cimag(x) ((x).imag())

static inline size_t volk_get_alignment() { return 64; }
static inline void* volk_malloc(size_t size, size_t alignment) {
    void* p = NULL;
    if (size == 0) { size = alignment; }
    if (posix_memalign(&p, alignment, size) != 0) { return NULL; }
    return p;
}
static inline void volk_free(void* p) { free(p); }

// ---- element-wise kernels ------------------------------------------------------------------------------------------
// volk_32fc_32f_multiply_32fc_generic: c[i] = a[i] * b[i] (complex * real)
static inline void volk_32fc_32f_multiply_32fc(lv_32fc_t* c, const lv_32fc_t* a, const float* b, unsigned int n) {
    const float* ap = (const float*)a;
    float* cp = (float*)c;
    for (unsigned int i = 0; i < n; i++) {
        cp[2 * i] = ap[2 * i] * b[i];
        cp[2 * i + 1] = ap[2 * i + 1] * b[i];
    }
}

// volk_32fc_x2_multiply_32fc_generic: naive complex product
static inline void volk_32fc_x2_multiply_32fc(lv_32fc_t* c, const lv_32fc_t* a, const lv_32fc_t* b, unsigned int n) {
    const float* ap = (const float*)a;
    const float* bp = (const float*)b;
    float* cp = (float*)c;
    for (unsigned int i = 0; i < n; i++) {
        float ar = ap[2 * i], ai = ap[2 * i + 1], br = bp[2 * i], bi = bp[2 * i + 1];
        cp[2 * i] = (ar * br) - (ai * bi);
        cp[2 * i + 1] = (ar * bi) + (ai * br);
    }
}

static inline void volk_32fc_conjugate_32fc(lv_32fc_t* c, const lv_32fc_t* a, unsigned int n) {
    const float* ap = (const float*)a;
    float* cp = (float*)c;
    for (unsigned int i = 0; i < n; i++) {
        cp[2 * i] = ap[2 * i];
        cp[2 * i + 1] = -ap[2 * i + 1];
    }
}

// volk_32fc_magnitude_32f_generic: sqrtf(re*re + im*im)
static inline void volk_32fc_magnitude_32f(float* out, const lv_32fc_t* in, unsigned int n) {
    const float* ip = (const float*)in;
    for (unsigned int i = 0; i < n; i++) {
        const float re = ip[2 * i], im = ip[2 * i + 1];
        out[i] = sqrtf((re * re) + (im * im));
    }
}

static inline void volk_32fc_deinterleave_real_32f(float* out, const lv_32fc_t* in, unsigned int n) {
    const float* ip = (const float*)in;
    for (unsigned int i = 0; i < n; i++) { out[i] = ip[2 * i]; }
}

static inline void volk_32f_x2_interleave_32fc(lv_32fc_t* out, const float* a, const float* b, unsigned int n) {
    float* op = (float*)out;
    for (unsigned int i = 0; i < n; i++) {
        const float x = a[i], y = b[i];  // a, b and out may alias (in-place demod buffers)
        op[2 * i] = x;
        op[2 * i + 1] = y;
    }
}

static inline void volk_32f_s32f_multiply_32f(float* c, const float* a, const float s, unsigned int n) {
    for (unsigned int i = 0; i < n; i++) { c[i] = a[i] * s; }
}
static inline void volk_32f_x2_multiply_32f(float* c, const float* a, const float* b, unsigned int n) {
    for (unsigned int i = 0; i < n; i++) { c[i] = a[i] * b[i]; }
}
static inline void volk_32f_x2_add_32f(float* c, const float* a, const float* b, unsigned int n) {
    for (unsigned int i = 0; i < n; i++) { c[i] = a[i] + b[i]; }
}
static inline void volk_32f_x2_subtract_32f(float* c, const float* a, const float* b, unsigned int n) {
    for (unsigned int i = 0; i < n; i++) { c[i] = a[i] - b[i]; }
}
static inline void volk_32f_accumulator_s32f(float* result, const float* in, unsigned int n) {
    float acc = 0.0f;
    for (unsigned int i = 0; i < n; i++) { acc += in[i]; }
    *result = acc;
}
static inline void volk_32f_index_max_32u(uint32_t* target, const float* src, uint32_t n) {
    float max = src[0];
    uint32_t index = 0;
    for (uint32_t i = 1; i < n; i++) {
        if (src[i] > max) { index = i; max = src[i]; }
    }
    *target = index;
}

// converts (generic: multiply by the reciprocal of the scalar)
static inline void volk_16i_s32f_convert_32f(float* out, const int16_t* in, const float scalar, unsigned int n) {
    const float iScalar = 1.0f / scalar;
    for (unsigned int i = 0; i < n; i++) { out[i] = ((float)in[i]) * iScalar; }
}
static inline void volk_8i_s32f_convert_32f(float* out, const int8_t* in, const float scalar, unsigned int n) {
    const float iScalar = 1.0f / scalar;
    for (unsigned int i = 0; i < n; i++) { out[i] = ((float)in[i]) * iScalar; }
}
static inline void volk_32f_s32f_convert_16i(int16_t* out, const float* in, const float scalar, unsigned int n) {
    for (unsigned int i = 0; i < n; i++) {
        float r = in[i] * scalar;
        if (r > 32767.0f) { r = 32767.0f; }
        else if (r < -32768.0f) { r = -32768.0f; }
        out[i] = (int16_t)rintf(r);
    }
}
static inline void volk_32f_s32f_convert_8i(int8_t* out, const float* in, const float scalar, unsigned int n) {
    for (unsigned int i = 0; i < n; i++) {
        float r = in[i] * scalar;
        if (r > 127.0f) { r = 127.0f; }
        else if (r < -128.0f) { r = -128.0f; }
        out[i] = (int8_t)rintf(r);
    }
}

// ---- dot products (one call per FIR output sample) -------------------------------------------------------------------
#ifndef SDRPP_SHIM_SIMD
// volk_32fc_32f_dot_prod_32fc_generic: sequential fp32 accumulation of re and im
static inline void volk_32fc_32f_dot_prod_32fc(lv_32fc_t* result, const lv_32fc_t* input, const float* taps, unsigned int n) {
    const float* ap = (const float*)input;
    float re = 0.0f, im = 0.0f;
    for (unsigned int i = 0; i < n; i++) {
        re += ap[2 * i] * taps[i];
        im += ap[2 * i + 1] * taps[i];
    }
    *result = lv_32fc_t(re, im);
}
// volk_32f_x2_dot_prod_32f_generic
static inline void volk_32f_x2_dot_prod_32f(float* result, const float* input, const float* taps, unsigned int n) {
    float acc = 0.0f;
    for (unsigned int i = 0; i < n; i++) { acc += input[i] * taps[i]; }
    *result = acc;
}
#else
// Vectorised reductions (what VOLK's a_avx/u_avx512f protokernels do: lane-partial sums, horizontal add at the end).
// Timing flavour only; compiled with -O3 -march=native -fopenmp-simd.
static inline void volk_32fc_32f_dot_prod_32fc(lv_32fc_t* result, const lv_32fc_t* input, const float* taps, unsigned int n) {
    const float* ap = (const float*)input;
    float re = 0.0f, im = 0.0f;
#pragma omp simd reduction(+ : re, im)
    for (unsigned int i = 0; i < n; i++) {
        re += ap[2 * i] * taps[i];
        im += ap[2 * i + 1] * taps[i];
    }
    *result = lv_32fc_t(re, im);
}
static inline void volk_32f_x2_dot_prod_32f(float* result, const float* input, const float* taps, unsigned int n) {
    float acc = 0.0f;
#pragma omp simd reduction(+ : acc)
    for (unsigned int i = 0; i < n; i++) { acc += input[i] * taps[i]; }
    *result = acc;
}
#endif

// volk_32fc_x2_dot_prod_32fc_generic (complex taps; only the stereo-pilot filter uses it — off the hot path)
static inline void volk_32fc_x2_dot_prod_32fc(lv_32fc_t* result, const lv_32fc_t* input, const lv_32fc_t* taps, unsigned int n) {
    const float* ap = (const float*)input;
    const float* bp = (const float*)taps;
    float re = 0.0f, im = 0.0f;
    for (unsigned int i = 0; i < n; i++) {
        re += (ap[2 * i] * bp[2 * i]) - (ap[2 * i + 1] * bp[2 * i + 1]);
        im += (ap[2 * i] * bp[2 * i + 1]) + (ap[2 * i + 1] * bp[2 * i]);
    }
    *result = lv_32fc_t(re, im);
}

// ---- rotator ---------------------------------------------------------------------------------------------------------
// volk_32fc_s32fc_x2_rotator2_32fc_generic (VOLK >= 2.4 behaviour): out = in * phase; phase *= inc; the phase is
// renormalised every ROTATOR_RELOAD = 512 samples and once more at the end of every call that had a remainder.
#define SDRPP_SHIM_ROTATOR_RELOAD 512
static inline void sdrpp_shim_rot_norm(float* pr, float* pi) {
    const float h = hypotf(*pr, *pi);
    *pr = *pr / h;
    *pi = *pi / h;
}
static inline void volk_32fc_s32fc_x2_rotator2_32fc(lv_32fc_t* out, const lv_32fc_t* in, const lv_32fc_t* phase_inc, lv_32fc_t* phase, unsigned int n) {
    const float* ip = (const float*)in;
    float* op = (float*)out;
    float pr = phase->real(), pi = phase->imag();
    const float dr = phase_inc->real(), di = phase_inc->imag();
    unsigned int i = 0, k = 0;
    for (i = 0; i < n / SDRPP_SHIM_ROTATOR_RELOAD; ++i) {
        for (int j = 0; j < SDRPP_SHIM_ROTATOR_RELOAD; ++j, ++k) {
            const float xr = ip[2 * k], xi = ip[2 * k + 1];
            op[2 * k] = (xr * pr) - (xi * pi);
            op[2 * k + 1] = (xr * pi) + (xi * pr);
            const float nr = (pr * dr) - (pi * di);
            const float ni = (pr * di) + (pi * dr);
            pr = nr;
            pi = ni;
        }
        sdrpp_shim_rot_norm(&pr, &pi);
    }
    for (i = 0; i < n % SDRPP_SHIM_ROTATOR_RELOAD; ++i, ++k) {
        const float xr = ip[2 * k], xi = ip[2 * k + 1];
        op[2 * k] = (xr * pr) - (xi * pi);
        op[2 * k + 1] = (xr * pi) + (xi * pr);
        const float nr = (pr * dr) - (pi * di);
        const float ni = (pr * di) + (pi * dr);
        pr = nr;
        pi = ni;
    }
    if (i) { sdrpp_shim_rot_norm(&pr, &pi); }
    *phase = lv_32fc_t(pr, pi);
}
static inline void volk_32fc_s32fc_x2_rotator_32fc(lv_32fc_t* out, const lv_32fc_t* in, const lv_32fc_t phase_inc, lv_32fc_t* phase, unsigned int n) {
    volk_32fc_s32fc_x2_rotator2_32fc(out, in, &phase_inc, phase, n);
}

// ---- power spectrum ----------------------------------------------------------------------------------------------------
// volk_32fc_s32f_power_spectrum_32f_generic:
//   re' = re * (1/norm); im' = im * (1/norm); out = 3.01029995663981209120f * log2f_non_ieee(re'*re' + im'*im')
// where log2f_non_ieee(x) = log2f(x), with an infinite result replaced by copysignf(127.0f, result)
// (volk_common.h).  The log2 itself is delegated to sdrpp_oracle_log2f (oracle/oracle.c) — a fully specified fp32
// polynomial shared, operation for operation, with the HIP kernel so that waterfall palette indices can be compared
// bit-exactly; tests bound its distance from libm's log2f.
extern "C" float sdrpp_oracle_log2f_non_ieee(float x);
static inline void volk_32fc_s32f_power_spectrum_32f(float* out, const lv_32fc_t* in, const float norm, unsigned int n) {
    const float* ip = (const float*)in;
    const float inv = 1.0f / norm;
    for (unsigned int i = 0; i < n; i++) {
        const float re = ip[2 * i] * inv;
        const float im = ip[2 * i + 1] * inv;
        const float p = (re * re) + (im * im);
        out[i] = 3.01029995663981209120f * sdrpp_oracle_log2f_non_ieee(p);
    }
}
