// The one exchange step of the path in a one-process, several-GPU host: finished waterfall lines of every stream gathered
// on the display GPU with RCCL over xGMI.  Stream i lives on device i with its own context (sdrpp_gpu::StreamBank / the C-ABI); nothing
// else crosses devices.  The gather is a grouped ncclSend / ncclRecv (every rank sends its block to the root, the root receives one
// block per rank: what ncclGather does, spelled with the point-to-point calls so that it also runs on RCCL builds without the
// extension) — lines are a few hundred KB per refresh against ~153 GB/s per xGMI link, so a direct gather is right, not a ring.
// Needs <rccl/rccl.h> + the HIP runtime (link -lrccl -lamdhip64); kept out of sdrpp_gpu_multi.h so that hosts without RCCL build.
#pragma once
#include <climits>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "sdrpp_gpu_multi.h"

namespace sdrpp_gpu {

class LineGather {
public:
    ~LineGather() { destroy(); }

    // devices[i] = HIP device of stream i (all different: one RCCL rank per device); `floatsPerStream` = capacity of one stream's block
    void init(const std::vector<int>& devices, size_t floatsPerStream, int rootIndex = 0) {
        destroy();
        devs = devices;
        cap = floatsPerStream;
        root = rootIndex;
        const int n = (int)devs.size();
        if (n <= 0 || root < 0 || root >= n) { throw std::runtime_error("[sdrpp_gpu::LineGather] bad device list"); }
        comms.assign((size_t)n, nullptr);
        streams.assign((size_t)n, nullptr);
        check(ncclCommInitAll(comms.data(), n, devs.data()), "ncclCommInitAll");
        for (int i = 0; i < n; i++) {
            hip(hipSetDevice(devs[(size_t)i]), "hipSetDevice");
            hip(hipStreamCreateWithFlags(&streams[(size_t)i], hipStreamNonBlocking), "hipStreamCreate");
        }
        hip(hipSetDevice(devs[(size_t)root]), "hipSetDevice");
        hip(hipMalloc((void**)&gathered, (size_t)n * cap * sizeof(float)), "hipMalloc");
        ready = true;
    }
    int ranks() const { return (int)devs.size(); }
    const char* backend() const { return "rccl"; }

    // send[i]: `count` floats in device memory of devices[i] (e.g. the zoomed lines sdrpp_fft_device_buffers points at, after the
    // context's stream has been synchronised).  Returns the root's [ranks][count] buffer (device memory of the root), complete.
    const float* gather(const std::vector<const float*>& send, size_t count) {
        if (!ready || send.size() != devs.size() || count > cap) { throw std::runtime_error("[sdrpp_gpu::LineGather] bad gather call"); }
        const int n = (int)devs.size();
        check(ncclGroupStart(), "ncclGroupStart");
        for (int i = 0; i < n; i++) {
            check(ncclSend(send[(size_t)i], count, ncclFloat, root, comms[(size_t)i], streams[(size_t)i]), "ncclSend");
            if (i == root) {
                for (int r = 0; r < n; r++) { check(ncclRecv(gathered + (size_t)r * count, count, ncclFloat, r, comms[(size_t)i], streams[(size_t)i]), "ncclRecv"); }
            }
        }
        check(ncclGroupEnd(), "ncclGroupEnd");
        for (int i = 0; i < n; i++) {
            hip(hipSetDevice(devs[(size_t)i]), "hipSetDevice");
            hip(hipStreamSynchronize(streams[(size_t)i]), "hipStreamSynchronize");
        }
        return gathered;
    }
    // the gathered block to host memory ([ranks][count] floats)
    void toHost(float* dst, size_t count) {
        hip(hipSetDevice(devs[(size_t)root]), "hipSetDevice");
        hip(hipMemcpy(dst, gathered, devs.size() * count * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy");
    }

    void destroy() {
        if (!ready) { return; }
        for (size_t i = 0; i < comms.size(); i++) {
            if (comms[i]) { ncclCommDestroy(comms[i]); }
            if (streams[i]) {
                (void)hipSetDevice(devs[i]);
                (void)hipStreamDestroy(streams[i]);
            }
        }
        if (gathered) {
            (void)hipSetDevice(devs[(size_t)root]);
            (void)hipFree(gathered);
        }
        gathered = nullptr;
        comms.clear();
        streams.clear();
        ready = false;
    }

private:
    static void check(ncclResult_t r, const char* what) {
        if (r != ncclSuccess) { throw std::runtime_error(std::string("[sdrpp_gpu::LineGather] ") + what + ": " + ncclGetErrorString(r)); }
    }
    static void hip(hipError_t e, const char* what) {
        if (e != hipSuccess) { throw std::runtime_error(std::string("[sdrpp_gpu::LineGather] ") + what + ": " + hipGetErrorString(e)); }
    }
    std::vector<int> devs;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    float* gathered = nullptr;
    size_t cap = 0;
    int root = 0;
    bool ready = false;
};

// StreamBank + LineGather: the newest waterfall line of every stream of a bank gathered on the display GPU.  Every front end keeps its newest
// raw dB line in memory of ITS device (IQFrontEnd::keepDeviceLine); a gather packs the lines of the streams that share a device into that
// device's block (device-to-device, on the device), then ONE grouped ncclSend / ncclRecv brings the blocks to the root device and one copy takes
// them to the host — the lines of the other GPUs never travel through host memory.  One RCCL rank per DISTINCT device of the bank (two streams
// on a one-GPU box: one rank, the collective still runs).
// The lines are COPIED out of the front ends (IQFrontEnd::copyLatestLineDevice: under the lock their workers flip / re-allocate the buffers
// with), in any mode of the front end (bypass, buffered, pipelined); a gather checks the bank's FFT size and every line's length and rebuilds
// its blocks and communicator when the size has changed since init (StreamBank::setFFTSize) instead of reading past buffers of the new size.
class BankLineGather {
public:
    ~BankLineGather() { destroy(); }
    void init(StreamBank& bank_, int rootStream = 0) {
        destroy();
        bank = &bank_;
        root = rootStream;
        n = bank->fftSize();
        const int ns = bank->size();
        if (ns <= 0 || rootStream < 0 || rootStream >= ns) { throw std::runtime_error("[sdrpp_gpu::BankLineGather] bad bank"); }
        rankOf.assign((size_t)ns, 0);
        slotOf.assign((size_t)ns, 0);
        for (int i = 0; i < ns; i++) {
            const int d = bank->deviceOf(i);
            size_t r = 0;
            while (r < devs.size() && devs[r] != d) { r++; }
            if (r == devs.size()) {
                devs.push_back(d);
                perRank.push_back(0);
            }
            rankOf[(size_t)i] = (int)r;
            slotOf[(size_t)i] = perRank[r]++;
            (*bank)[i].keepDeviceLine(true);
        }
        maxPer = 0;
        for (int c : perRank) { maxPer = c > maxPer ? c : maxPer; }
        blocks.assign(devs.size(), nullptr);
        for (size_t r = 0; r < devs.size(); r++) {
            hip(hipSetDevice(devs[r]), "hipSetDevice");
            hip(hipMalloc((void**)&blocks[r], (size_t)maxPer * (size_t)n * sizeof(float)), "hipMalloc");
            hip(hipMemset(blocks[r], 0, (size_t)maxPer * (size_t)n * sizeof(float)), "hipMemset");
        }
        g.init(devs, (size_t)maxPer * (size_t)n, rankOf[(size_t)rootStream]);
        ready = true;
    }
    int ranks() const { return g.ranks(); }
    // out: [streams][fftSize] floats, the newest line of every stream (zeros for a stream that has none yet).  Returns the streams that had one.
    int gather(std::vector<float>& out) {
        if (!ready) { throw std::runtime_error("[sdrpp_gpu::BankLineGather] not initialised"); }
        if (bank->fftSize() != n) { init(*bank, root); }  // the bank's FFT size changed since init: blocks and communicator for the new size
        const int ns = bank->size();
        int have = 0;
        for (int i = 0; i < ns; i++) {
            float* slot = blocks[(size_t)rankOf[(size_t)i]] + (size_t)slotOf[(size_t)i] * (size_t)n;
            const int got = (*bank)[i].copyLatestLineDevice(slot, n);  // a copy under the front end's lock, never a pointer into its buffers
            if (got == n) { have++; }
            else {
                // no line yet, or a line of another length (this stream's worker has not produced one at the bank's new size yet, or a size
                // change is on its way through the bank): the stream reports "no line" this time — zeros, as before its first line
                hip(hipSetDevice(devs[(size_t)rankOf[(size_t)i]]), "hipSetDevice");
                hip(hipMemset(slot, 0, (size_t)n * sizeof(float)), "hipMemset");
                if (got == INT_MIN) { throw std::runtime_error("[sdrpp_gpu::BankLineGather] device copy of a stream's line failed"); }
            }
        }
        std::vector<const float*> send(blocks.begin(), blocks.end());
        g.gather(send, (size_t)maxPer * (size_t)n);
        staging.resize(devs.size() * (size_t)maxPer * (size_t)n);
        g.toHost(staging.data(), (size_t)maxPer * (size_t)n);
        out.assign((size_t)ns * (size_t)n, 0.0f);
        for (int i = 0; i < ns; i++) {
            const float* src = &staging[((size_t)rankOf[(size_t)i] * (size_t)maxPer + (size_t)slotOf[(size_t)i]) * (size_t)n];
            std::copy(src, src + n, out.begin() + (size_t)i * (size_t)n);
        }
        return have;
    }
    void destroy() {
        if (!ready) { return; }
        for (size_t r = 0; r < blocks.size(); r++) {
            if (blocks[r]) {
                (void)hipSetDevice(devs[r]);
                (void)hipFree(blocks[r]);
            }
        }
        blocks.clear();
        devs.clear();
        perRank.clear();
        g.destroy();
        ready = false;
    }

private:
    static void hip(hipError_t e, const char* what) {
        if (e != hipSuccess) { throw std::runtime_error(std::string("[sdrpp_gpu::BankLineGather] ") + what + ": " + hipGetErrorString(e)); }
    }
    StreamBank* bank = nullptr;
    LineGather g;
    std::vector<int> devs, perRank, rankOf, slotOf;
    std::vector<float*> blocks;
    std::vector<float> staging;
    int n = 0, maxPer = 0, root = 0;
    bool ready = false;
};

}  // namespace sdrpp_gpu
