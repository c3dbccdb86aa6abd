// demod::Demodulator-shaped adaptor for the radio module (decoder_modules/radio/src/demod.h:34-61).
//
// In the reference every demodulator is a chain of dsp blocks with their own threads behind the VFO's output stream
// (radio_module.h:419-563: selectedDemod->init(name, &config, ifChain.out, bw, audioSR); afChain.setInput(selectedDemod->getOutput())).
// On the GPU the demodulator is fused behind its channeliser inside the VFO bank, so the object the radio module holds does no
// signal processing of its own: it finds the sdrpp_gpu::RxVFO whose `out` stream it was handed as `input`, attaches the requested
// demodulator to it (RxVFO::attachDemod) and returns that VFO's `audio` stream from getOutput().  The interface is taken from
// whatever header declares it — `Base` is the radio module's demod::Demodulator inside an SDR++ tree — so this file restates
// nothing but the per-mode constants of demodulators/{wfm,nfm,am,usb,lsb,dsb}.h.
//
//   #include "demod.h"                                   // radio module: declares demod::Demodulator
//   #include <sdrpp_gpu_radio.h>
//   using GpuWFM = sdrpp_gpu::FusedDemodulator<demod::Demodulator, sdrpp_gpu::Demod::WFM>;
//   ... demod = new GpuWFM(&sigpath::iqFrontEnd);  demod->init(name, &config, vfo->output, bw, audioSR);
//
// Not covered (a fused VFO cannot have CPU blocks between channeliser and demodulator): the radio module's IF chain (noise
// blanker, FM IF noise reduction, squelch; radio_module.h:84-96) must stay disabled — init() throws if `input` is not a VFO output.
// showMenu() draws nothing (GUI is out of scope); the options it would toggle are setLowPass / setAGC* / setCarrierAgc below.
#pragma once
#include <stdexcept>
#include <string>

#include "sdrpp_gpu_blocks.h"

class ConfigManager;  // core/src/config.h (only ever passed through)

namespace sdrpp_gpu {

// Per-mode constants of the radio module's analog demodulators (decoder_modules/radio/src/demodulators/*.h, "INFO" blocks:
// wfm.h:267-282, nfm.h:55-70, am.h:75-90, usb.h:69-84, lsb.h:68-83, dsb.h:68-83).  vfoReference: ImGui::WaterfallVFO::REF_LOWER = 0,
// REF_CENTER = 1, REF_UPPER = 2 (gui/widgets/waterfall.h:26-30); deemphasis mode: DEEMP_MODE_50US = 1, DEEMP_MODE_NONE = 3
// (demod.h:8-14).
struct DemodInfo {
    const char* name;
    double ifSampleRate, defaultBandwidth, minBandwidth, maxBandwidth, defaultSnapInterval;
    int vfoReference;
    bool deempAllowed;
    int defaultDeemphasisMode;
    bool fmIfnrAllowed, nbAllowed;
};
inline const DemodInfo& demodInfo(Demod m) {
    static const DemodInfo wfm{ "WFM", 250000.0, 150000.0, 50000.0, 250000.0, 100000.0, 1, true, 1, true, false };
    static const DemodInfo nfm{ "FM", 50000.0, 12500.0, 1000.0, 50000.0, 2500.0, 1, true, 3, true, false };
    static const DemodInfo am{ "AM", 15000.0, 10000.0, 1000.0, 15000.0, 1000.0, 1, false, 3, false, false };
    static const DemodInfo usb{ "USB", 24000.0, 2800.0, 500.0, 12000.0, 100.0, 0, false, 3, false, true };
    static const DemodInfo lsb{ "LSB", 24000.0, 2800.0, 500.0, 12000.0, 100.0, 2, false, 3, false, true };
    static const DemodInfo dsb{ "DSB", 24000.0, 4600.0, 1000.0, 12000.0, 100.0, 1, false, 3, false, true };
    switch (m) {
    case Demod::WFM: return wfm;
    case Demod::NFM: return nfm;
    case Demod::AM: return am;
    case Demod::USB: return usb;
    case Demod::LSB: return lsb;
    default: return dsb;
    }
}

template <class Base, Demod MODE>
class FusedDemodulator : public Base {
public:
    explicit FusedDemodulator(IQFrontEnd* frontEnd) : fe(frontEnd) {}
    ~FusedDemodulator() override {}

    void init(std::string name, ConfigManager* config, dsp::stream<dsp::complex_t>* input, double bandwidth, double audioSR) override {
        (void)config;  // the options live in the radio module's config; the setters below receive them
        (void)audioSR;
        _name = name;
        _bandwidth = bandwidth;
        bind(input);
    }
    // The VFO bank's worker runs the demodulator; starting / stopping it on its own has no counterpart (the front end's start / stop
    // covers it).  Kept so that the radio module's call sequence (demod.h:38-39, radio_module.h:455-473) stays valid.
    void start() override {}
    void stop() override {}
    void showMenu() override {}
    void setBandwidth(double bandwidth) override {
        _bandwidth = bandwidth;
        if (vfo) { vfo->setDemodBandwidth(bandwidth); }
    }
    void setInput(dsp::stream<dsp::complex_t>* input) override { bind(input); }
    void AFSampRateChanged(double newSR) override { (void)newSR; }
    const char* getName() override { return demodInfo(MODE).name; }
    double getIFSampleRate() override { return demodInfo(MODE).ifSampleRate; }
    double getAFSampleRate() override { return getIFSampleRate(); }
    double getDefaultBandwidth() override { return demodInfo(MODE).defaultBandwidth; }
    double getMinBandwidth() override { return demodInfo(MODE).minBandwidth; }
    double getMaxBandwidth() override { return demodInfo(MODE).maxBandwidth; }
    bool getBandwidthLocked() override { return false; }
    double getDefaultSnapInterval() override { return demodInfo(MODE).defaultSnapInterval; }
    int getVFOReference() override { return demodInfo(MODE).vfoReference; }
    bool getDeempAllowed() override { return demodInfo(MODE).deempAllowed; }
    bool getPostProcEnabled() override { return true; }
    int getDefaultDeemphasisMode() override { return demodInfo(MODE).defaultDeemphasisMode; }
    bool getFMIFNRAllowed() override { return demodInfo(MODE).fmIfnrAllowed; }
    bool getNBAllowed() override { return demodInfo(MODE).nbAllowed; }
    bool getHighPassAllowed() override { return true; }
    bool getSquelchAllowed() override { return true; }
    dsp::stream<dsp::stereo_t>* getOutput() override { return vfo ? &vfo->audio : nullptr; }

    // what the reference's showMenu() toggles (wfm.h:104-111 / nfm.h:37-43 "Low Pass"; am.h:40-63, usb.h:40-57 AGC attack / decay, carrier AGC)
    void setLowPass(bool enabled) { _lowPass = enabled; apply(); }
    void setAGCAttack(double attack) { _agcAttack = attack; apply(); }
    void setAGCDecay(double decay) { _agcDecay = decay; apply(); }
    void setCarrierAgc(bool enabled) { _carrierAgc = enabled; apply(); }
    RxVFO* channel() { return vfo; }

private:
    void bind(dsp::stream<dsp::complex_t>* input) {
        RxVFO* v = fe ? fe->vfoOfStream(input) : nullptr;
        if (!v) {
            throw std::runtime_error("[sdrpp_gpu::FusedDemodulator] the input is not the output stream of a VFO of this front end "
                                     "(the radio module's IF chain blocks must stay disabled: the demodulator is fused behind the channeliser)");
        }
        if (vfo && vfo != v) { vfo->attachDemod(Demod::RAW); }  // the previous channel goes back to delivering its IF
        vfo = v;
        apply();
    }
    void apply() {
        if (!vfo) { return; }
        vfo->demodBandwidth = _bandwidth;
        vfo->attachDemod(MODE, _lowPass, _agcAttack, _agcDecay, _carrierAgc);
    }

    IQFrontEnd* fe;
    RxVFO* vfo = nullptr;
    std::string _name;
    double _bandwidth = 0.0;
    bool _lowPass = true, _carrierAgc = false;
    double _agcAttack = 50.0, _agcDecay = 5.0;  // am.h:98-99, usb.h:92-93
};

}  // namespace sdrpp_gpu
