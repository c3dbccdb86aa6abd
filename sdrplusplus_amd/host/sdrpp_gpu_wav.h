// File-source ingest for the GPU front end: the on-disk format and the block cadence of SDR++'s file_source module
// (source_modules/file_source/src/wavreader.h:14-79, main.cpp:154-181), feeding either a dsp::stream<complex_t> exactly like the module
// does, or the front end directly with the samples still in their 16-bit form — the conversion x / 32768 (volk_16i_s32f_convert_32f,
// main.cpp:162) then runs on the device (sdrpp_push_int16) and the bus carries 4 instead of 8 bytes per sample.
//
//   WavReader   44-byte canonical header ("RIFF" <size> "WAVE" "fmt " 16 <type> <channels> <rate> <bytes/s> <block align> <bits> "data" <size>),
//               validity = the two magic words only (wavreader.h:18-21: nothing else is checked), payload read in caller-sized pieces with
//               a wrap to byte 44 when the file ends inside a piece (wavreader.h:41-51: the file loops for ever).
//   WavSource   FileSourceModule's data side: block size min(int(sampleRate / 200.0f), 1 000 000) complex samples, int16 or float32
//               payload ("Float32 Mode"), worker until stopped, rewind on stop (main.cpp:99-109).
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "sdrpp_gpu_blocks.h"

namespace sdrpp_gpu {

class WavReader {
public:
    static constexpr long kHeaderBytes = 44;

    explicit WavReader(const std::string& path) {
        fp = fopen(path.c_str(), "rb");
        if (!fp) { return; }
        unsigned char h[kHeaderBytes];
        if (fread(h, 1, sizeof(h), fp) != sizeof(h)) { memset(h, 0, sizeof(h)); }
        valid = !memcmp(h, "RIFF", 4) && !memcmp(h + 8, "WAVE", 4);
        auto u16 = [&](int o) { return (uint16_t)(h[o] | (h[o + 1] << 8)); };
        auto u32 = [&](int o) { return (uint32_t)h[o] | ((uint32_t)h[o + 1] << 8) | ((uint32_t)h[o + 2] << 16) | ((uint32_t)h[o + 3] << 24); };
        sampleType = u16(20);
        channels = u16(22);
        rate = u32(24);
        bits = u16(34);
        dataBytes = u32(40);
    }
    ~WavReader() { close(); }
    WavReader(const WavReader&) = delete;
    WavReader& operator=(const WavReader&) = delete;

    bool isValid() const { return valid; }
    uint16_t getBitDepth() const { return bits; }
    uint16_t getChannelCount() const { return channels; }
    uint32_t getSampleRate() const { return rate; }
    uint16_t getSampleType() const { return sampleType; }
    uint32_t getDataSize() const { return dataBytes; }
    size_t getBytesRead() const { return consumed; }

    // `size` payload bytes into `data`; a file that ends inside the request continues at its first payload byte (one wrap per call, as in
    // the reference: a request longer than the whole payload is not topped up a second time)
    void readSamples(void* data, size_t size) {
        char* dst = (char*)data;
        size_t got = fp ? fread(dst, 1, size, fp) : 0;
        if (got < size && fp) {
            clearerr(fp);
            fseek(fp, kHeaderBytes, SEEK_SET);
            const size_t more = fread(dst + got, 1, size - got, fp);
            if (got + more < size) { memset(dst + got + more, 0, size - got - more); }  // (the reference leaves this part of its buffer as it was)
        }
        consumed += size;
    }
    void rewind() {
        if (fp) {
            clearerr(fp);
            fseek(fp, kHeaderBytes, SEEK_SET);
        }
    }
    void close() {
        if (fp) { fclose(fp); }
        fp = nullptr;
    }

private:
    FILE* fp = nullptr;
    bool valid = false;
    uint16_t sampleType = 0, channels = 0, bits = 0;
    uint32_t rate = 0, dataBytes = 0;
    size_t consumed = 0;
};

class WavSource {
public:
    dsp::stream<dsp::complex_t> stream;  // FileSourceModule::stream: what start() feeds

    ~WavSource() {
        stop();
        delete reader;
    }

    // menuHandler's file selection (main.cpp:120-140): a file with the magic words and a non-zero sample rate
    bool open(const std::string& path) {
        stop();
        delete reader;
        reader = new WavReader(path);
        if (!reader->isValid() || reader->getSampleRate() == 0) {
            delete reader;
            reader = nullptr;
            return false;
        }
        return true;
    }
    bool isOpen() const { return reader != nullptr; }
    double getSampleRate() const { return reader ? (double)reader->getSampleRate() : 0.0; }
    // main.cpp:156-157: float arithmetic, then the stream's capacity
    int blockSize() const {
        const double sampleRate = std::max(reader ? reader->getSampleRate() : 0u, (uint32_t)1);
        return std::min((int)(sampleRate / 200.0f), (int)STREAM_BUFFER_SIZE);
    }
    void setFloat32Mode(bool enabled) { float32Mode = enabled; }  // the module's "Float32 Mode" check box: payload = complex float32
    // 0 (default): read for ever, like the module; n > 0: hand over n blocks, then the worker ends (tests, file conversion)
    void setBlockLimit(long long n) { blockLimit = n; }
    long long blocksDone() const { return blocks.load(); }
    bool finished() const { return done.load(); }

    // FileSourceModule::start (main.cpp:86-96): blocks of blockSize() samples on `stream`
    void start() { launch(nullptr); }
    // The same cadence straight into a front end whose own worker is not running (IQFrontEnd::ingestInt16 / ingestFloat): int16 payloads
    // reach the device as they are
    void start(IQFrontEnd* fe) { launch(fe); }
    // FileSourceModule::stop (main.cpp:98-109)
    void stop() {
        if (!running) { return; }
        quit = true;
        stream.stopWriter();
        if (worker.joinable()) { worker.join(); }
        stream.clearWriteStop();
        running = false;
        if (reader) { reader->rewind(); }
    }

private:
    void launch(IQFrontEnd* fe) {
        if (running || !reader) { return; }
        quit = false;
        done = false;
        blocks = 0;
        running = true;
        worker = std::thread([this, fe]() {
            const int n = blockSize();
            std::vector<int16_t> i16(float32Mode ? 0 : (size_t)n * 2);
            std::vector<dsp::complex_t> f32((float32Mode && fe) ? (size_t)n : 0);
            while (!quit && (blockLimit <= 0 || blocks < blockLimit)) {
                if (float32Mode) {  // floatWorker, main.cpp:169-181
                    dsp::complex_t* dst = fe ? f32.data() : stream.writeBuf;
                    reader->readSamples(dst, (size_t)n * sizeof(dsp::complex_t));
                    if (fe ? fe->ingestFloat(dst, n) < 0 : !stream.swap(n)) { break; }
                }
                else {  // worker, main.cpp:154-167
                    reader->readSamples(i16.data(), (size_t)n * 2 * sizeof(int16_t));
                    if (fe) {
                        if (fe->ingestInt16(i16.data(), n) < 0) { break; }
                    }
                    else {
                        float* out = (float*)stream.writeBuf;
                        for (int i = 0; i < 2 * n; i++) { out[i] = (float)i16[(size_t)i] / 32768.0f; }  // volk_16i_s32f_convert_32f generic
                        if (!stream.swap(n)) { break; }
                    }
                }
                blocks++;
            }
            done = true;
        });
    }

    WavReader* reader = nullptr;
    bool float32Mode = false, running = false;
    std::atomic<bool> quit{ false }, done{ false };
    std::atomic<long long> blocks{ 0 };
    long long blockLimit = 0;
    std::thread worker;
};

}  // namespace sdrpp_gpu
