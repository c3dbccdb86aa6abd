// TEST INFRASTRUCTURE (oracle): the reference's OWN WAV reader compiled where it lies
// (/root/reference/source_modules/file_source/src/wavreader.h, included through -I) plus the few lines of FileSourceModule::worker /
// floatWorker around it (main.cpp:154-181: block size, int16 -> float conversion).  Pins sdrplusplus_amd/host/sdrpp_gpu_wav.h
// (tests/test_wav_source.py).  Nothing in the product path links or loads this file.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <wavreader.h>

extern "C" {

// header fields as the reference reader reports them; returns isValid()
int ref_wav_info(const char* path, unsigned* sampleRate, unsigned* bitDepth, unsigned* channels) {
    WavReader r(path);
    *sampleRate = r.getSampleRate();
    *bitDepth = r.getBitDepth();
    *channels = r.getChannelCount();
    const int ok = r.isValid() ? 1 : 0;
    r.close();
    return ok;
}

// file_source's block size for this file (main.cpp:156-157; STREAM_BUFFER_SIZE = 1 000 000, dsp/stream.h:9)
int ref_wav_block_size(const char* path) {
    WavReader r(path);
    double sampleRate = std::max(r.getSampleRate(), (uint32_t)1);
    const int blockSize = std::min((int)(sampleRate / 200.0f), 1000000);
    r.close();
    return blockSize;
}

// `nblocks` blocks exactly as FileSourceModule::worker produces them: readSamples(blockSize * 2 int16) -> x / 32768 (the generic
// volk_16i_s32f_convert_32f) -> out[nblocks][blockSize][2]; float32Mode: readSamples straight into the floats (floatWorker)
int ref_wav_blocks(const char* path, int float32Mode, int nblocks, float* out) {
    WavReader r(path);
    if (!r.isValid()) { return -1; }
    double sampleRate = std::max(r.getSampleRate(), (uint32_t)1);
    const int blockSize = std::min((int)(sampleRate / 200.0f), 1000000);
    int16_t* inBuf = new int16_t[(size_t)blockSize * 2];
    for (int b = 0; b < nblocks; b++) {
        float* dst = out + (size_t)b * blockSize * 2;
        if (float32Mode) { r.readSamples(dst, (size_t)blockSize * 2 * sizeof(float)); }
        else {
            r.readSamples(inBuf, (size_t)blockSize * 2 * sizeof(int16_t));
            for (int i = 0; i < blockSize * 2; i++) { dst[i] = ((float)inBuf[i]) / 32768.0f; }
        }
    }
    delete[] inBuf;
    r.close();
    return blockSize;
}
}
