// ORACLE / TEST INFRASTRUCTURE ONLY.
// Minimal stand-in for the four libfftw3f entry points IQFrontEnd uses (core/src/signal_path/iq_frontend.cpp:
// 12-14, 60-62, 255, 294-298).  libfftw3f is not vendored in the reference and not installed here; FFTW_ESTIMATE
// plans are machine dependent anyway, so "the" reference FFT is only defined as an exact unnormalised forward DFT.
// fftwf_execute forwards to sdrpp_oracle_fft (oracle/oracle.c), the fully specified fp32 FFT that the HIP kernels
// reproduce operation for operation.
#pragma once
#include <stdlib.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef float fftwf_complex[2];
struct sdrpp_shim_fftw_plan { int n; fftwf_complex* in; fftwf_complex* out; };
typedef struct sdrpp_shim_fftw_plan* fftwf_plan;
#define FFTW_FORWARD (-1)
#define FFTW_ESTIMATE (1U << 6)
void sdrpp_oracle_fft(int n, const float* in_interleaved, float* out_interleaved);
static inline void* fftwf_malloc(size_t n) { void* p = NULL; if (posix_memalign(&p, 64, n ? n : 64)) { return NULL; } return p; }
static inline void fftwf_free(void* p) { free(p); }
static inline fftwf_plan fftwf_plan_dft_1d(int n, fftwf_complex* in, fftwf_complex* out, int sign, unsigned flags) {
    (void)sign; (void)flags;
    fftwf_plan p = (fftwf_plan)malloc(sizeof(struct sdrpp_shim_fftw_plan));
    p->n = n; p->in = in; p->out = out;
    return p;
}
static inline void fftwf_execute(const fftwf_plan p) { sdrpp_oracle_fft(p->n, (const float*)p->in, (float*)p->out); }
static inline void fftwf_destroy_plan(fftwf_plan p) { free(p); }
#ifdef __cplusplus
}
#endif
