// TEST INFRASTRUCTURE — pins oracle.c's restatement of the waterfall widget's arithmetic to the reference's own text.
//
// core/src/gui/widgets/waterfall.cpp cannot be compiled here (ImGui + OpenGL), but the functions on the hot path are plain C++ over a
// handful of members.  oracle/Makefile cuts them out of the reference file WHERE IT LIES, function by function (sed pattern ranges),
// into oracle/_ref/waterfall_extract.inc — doZoom (:65-90), WaterFall::calculateVFOSignalInfo (:558-598), updateWaterfallFb
// (:600-631), getFFTBuffer (:875-886), pushFFT (:888-942), setFFTHold / setFFTHoldSpeed / setFFTSmoothing / setFFTSmoothingSpeed
// (:1153-1194) — and this file supplies the 40 lines around them: a class with the members those bodies touch (same names and types as
// gui/widgets/waterfall.h) and a C API.  Nothing of the reference is copied into the repository; _ref/ is git-ignored.
// The palette is set to the identity (waterfallPallet[i] = i), so the frame buffer holds palette INDICES — what the device returns.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <volk/volk.h>

#define WATERFALL_RESOLUTION 1000000  // gui/widgets/waterfall.h:11

namespace ImGui {
    struct WaterfallVFO {
        double centerOffset = 0.0, bandwidth = 0.0;
    };
    class WaterFall {
    public:
        bool calculateVFOSignalInfo(float* fftLine, WaterfallVFO* _vfo, float& strength, float& snr);
        void updateWaterfallFb();
        float* getFFTBuffer();
        void pushFFT();
        void setFFTHold(bool hold);
        void setFFTHoldSpeed(float speed);
        void setFFTSmoothing(bool enabled);
        void setFFTSmoothingSpeed(float speed);

        bool waterfallVisible = true, waterfallUpdate = false;
        float* rawFFTs = NULL;
        double viewOffset = 0.0, wholeBandwidth = 1.0, viewBandwidth = 1.0;
        int dataWidth = 0, waterfallHeight = 0, fftLines = 0, currentFFTLine = 0, rawFFTSize = 0;
        float waterfallMax = 0.0f, waterfallMin = -70.0f;
        uint32_t* waterfallFb = NULL;
        uint32_t* waterfallPallet = NULL;  // [WATERFALL_RESOLUTION]
        std::recursive_mutex buf_mtx, latestFFTMtx;
        std::mutex smoothingBufMtx;
        float* latestFFT = NULL;
        float* latestFFTHold = NULL;
        float* smoothingBuf = NULL;
        bool fftSmoothing = false, fftHold = false, snrSmoothing = false;
        float fftSmoothingAlpha = 0.0f, fftSmoothingBeta = 1.0f, fftHoldSpeed = 0.0f, snrSmoothingAlpha = 0.0f, snrSmoothingBeta = 1.0f, selectedVFOSNR = 0.0f;
        std::string selectedVFO = "";
        std::map<std::string, WaterfallVFO*> vfos;
    };
}

#include "_ref/waterfall_extract.inc"

extern "C" {
void ref_do_zoom(int offset, int width, int inSize, int outSize, float* in, float* out) { doZoom(offset, width, inSize, outSize, in, out); }

void* ref_wf_create(int height, int N, int dataWidth) {
    ImGui::WaterFall* w = new ImGui::WaterFall;
    w->waterfallHeight = height;
    w->rawFFTSize = N;
    w->dataWidth = dataWidth;
    w->rawFFTs = new float[(size_t)height * N]();
    w->waterfallFb = new uint32_t[(size_t)height * dataWidth]();
    w->waterfallPallet = new uint32_t[WATERFALL_RESOLUTION];
    for (uint32_t i = 0; i < WATERFALL_RESOLUTION; i++) { w->waterfallPallet[i] = i; }
    w->latestFFT = new float[dataWidth]();       // WaterFall::onResize allocates all three with the data width
    w->latestFFTHold = new float[dataWidth]();
    return w;
}
void ref_wf_destroy(void* h) {
    ImGui::WaterFall* w = (ImGui::WaterFall*)h;
    delete[] w->rawFFTs;
    delete[] w->waterfallFb;
    delete[] w->waterfallPallet;
    delete[] w->latestFFT;
    delete[] w->latestFFTHold;
    delete[] w->smoothingBuf;
    delete w;
}
// view as the widget keeps it: setViewOffset / setViewBandwidth / setBandwidth, waterfallMin / Max
void ref_wf_set_view(void* h, double viewOffset, double viewBandwidth, double wholeBandwidth, float wmin, float wmax) {
    ImGui::WaterFall* w = (ImGui::WaterFall*)h;
    w->viewOffset = viewOffset;
    w->viewBandwidth = viewBandwidth;
    w->wholeBandwidth = wholeBandwidth;
    w->waterfallMin = wmin;
    w->waterfallMax = wmax;
}
void ref_wf_set_smoothing(void* h, int enabled, float speed) {
    ImGui::WaterFall* w = (ImGui::WaterFall*)h;
    w->setFFTSmoothing(enabled != 0);
    w->setFFTSmoothingSpeed(speed);
}
void ref_wf_set_hold(void* h, int enabled, float speed) {
    ImGui::WaterFall* w = (ImGui::WaterFall*)h;
    w->setFFTHold(enabled != 0);
    w->setFFTHoldSpeed(speed);
}
// IQFrontEnd::handler's use of the pair (iq_frontend.cpp:258-266): acquire -> write the dB line -> release; idx = the new top row
void ref_wf_push(void* h, const float* line, int32_t* idx) {
    ImGui::WaterFall* w = (ImGui::WaterFall*)h;
    float* dst = w->getFFTBuffer();
    memcpy(dst, line, sizeof(float) * (size_t)w->rawFFTSize);
    w->pushFFT();
    for (int j = 0; j < w->dataWidth; j++) { idx[j] = (int32_t)w->waterfallFb[j]; }
}
void ref_wf_latest(void* h, float* latest, float* hold) {
    ImGui::WaterFall* w = (ImGui::WaterFall*)h;
    memcpy(latest, w->latestFFT, sizeof(float) * (size_t)w->dataWidth);
    memcpy(hold, w->latestFFTHold, sizeof(float) * (size_t)w->dataWidth);
}
int ref_wf_signal_info(void* h, double centerOffset, double bandwidth, float* strength, float* snr) {
    ImGui::WaterFall* w = (ImGui::WaterFall*)h;
    ImGui::WaterfallVFO v;
    v.centerOffset = centerOffset;
    v.bandwidth = bandwidth;
    return w->calculateVFOSignalInfo(w->fftLines > 0 ? &w->rawFFTs[(size_t)w->currentFFTLine * w->rawFFTSize] : NULL, &v, *strength, *snr) ? 1 : 0;
}
// full re-raster: fb[height][dataWidth], opaque black ((uint32_t)255 << 24) reported as -1; returns min(waterfallHeight, fftLines)
int ref_wf_raster(void* h, int32_t* fb) {
    ImGui::WaterFall* w = (ImGui::WaterFall*)h;
    w->updateWaterfallFb();
    const int count = std::min<int>(w->waterfallHeight, w->fftLines);
    for (size_t i = 0; i < (size_t)w->waterfallHeight * w->dataWidth; i++) {
        fb[i] = (i < (size_t)count * w->dataWidth) ? (int32_t)w->waterfallFb[i] : ((w->waterfallFb[i] == ((uint32_t)255 << 24)) ? -1 : -2);
    }
    return count;
}
}
