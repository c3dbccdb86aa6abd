// One process, several GPUs: independent wideband IQ streams, stream i on device i — the C++ counterpart of
// bench.py's one-rank-per-GPU layout.  Every stream gets its own sdrpp_gpu::IQFrontEnd (own context, own worker thread, own HIP streams
// on its device); nothing is exchanged between streams on the data path.  The only meeting point is the display side: the finished dB
// lines of all streams arrive through ONE handler, tagged with the stream index (in the multi-process layout that is the RCCL gather
// on rank 0; inside one process the lines are copied to the host by each worker anyway, so the gather is the shared handler).
#pragma once
#include <functional>
#include <memory>
#include <mutex>
#include <vector>

#include "sdrpp_gpu_blocks.h"

namespace sdrpp_gpu {

class StreamBank {
public:
    // called from the streams' worker threads, serialised: one finished dB line (fftSize floats, DC-centred) of stream `index`
    using LineHandler = std::function<void(int index, const float* line, int fftSize)>;

    // streams.size() front ends; stream i runs on device (firstDevice + i) % sdrpp_device_count().  Parameters as IQFrontEnd::init.
    void init(const std::vector<dsp::stream<dsp::complex_t>*>& streams, double sampleRate, bool buffering, int decimRatio, bool dcBlocking, int fftSize,
              double fftRate, IQFrontEnd::FFTWindow fftWindow, LineHandler onLine, const DecimPlans* plans = nullptr, int firstDevice = 0) {
        const int ndev = sdrpp_device_count();
        if (ndev <= 0) { throw std::runtime_error("[sdrpp_gpu::StreamBank] no device"); }
        _onLine = std::move(onLine);
        _fftSize = fftSize;
        for (size_t i = 0; i < streams.size(); i++) {
            slots.emplace_back(new Slot);
            Slot& s = *slots.back();
            s.bank = this;
            s.index = (int)i;
            s.line.assign((size_t)fftSize, 0.0f);
            s.fe.init(streams[i], sampleRate, buffering, decimRatio, dcBlocking, fftSize, fftRate, fftWindow, &StreamBank::acquire, &StreamBank::release, &s,
                      (firstDevice + (int)i) % ndev, plans);
        }
    }
    int size() const { return (int)slots.size(); }
    int fftSize() const { return _fftSize; }
    int deviceOf(int i) const { return slots[(size_t)i]->fe.device(); }
    IQFrontEnd& operator[](int i) { return slots[(size_t)i]->fe; }
    void setFFTSize(int size) {
        _fftSize = size;
        for (auto& s : slots) {
            s->line.assign((size_t)size, 0.0f);
            s->fe.setFFTSize(size);
        }
    }
    void start() { for (auto& s : slots) { s->fe.start(); } }
    void stop() { for (auto& s : slots) { s->fe.stop(); } }

private:
    struct Slot {
        StreamBank* bank = nullptr;
        int index = 0;
        std::vector<float> line;
        IQFrontEnd fe;
    };
    static float* acquire(void* c) { return ((Slot*)c)->line.data(); }
    static void release(void* c) {
        Slot* s = (Slot*)c;
        std::lock_guard<std::mutex> lck(s->bank->lineMtx);
        if (s->bank->_onLine) { s->bank->_onLine(s->index, s->line.data(), (int)s->line.size()); }
    }
    std::vector<std::unique_ptr<Slot>> slots;
    LineHandler _onLine;
    std::mutex lineMtx;
    int _fftSize = 0;
};

}  // namespace sdrpp_gpu
