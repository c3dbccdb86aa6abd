// TEST DOUBLE of the contract of SDR++'s dsp::block (core/src/dsp/block.h:17-131) for building tests/host_cpp without the SDR++ tree:
// one worker thread per block running `while (run() >= 0)`, start/stop under ctrlMtx, tempStop/tempStart nesting around
// reconfiguration, registered input/output streams are told to stop so that a blocked worker returns.
#pragma once
#include <algorithm>
#include <cassert>
#include <mutex>
#include <thread>
#include <vector>
#include "stream.h"
namespace dsp {
    class generic_block {
    public:
        virtual ~generic_block() {}
        virtual void start() {}
        virtual void stop() {}
        virtual int run() { return -1; }
    };

    class block : public generic_block {
    public:
        ~block() override {
            if (!_block_init) { return; }
            stop();
            _block_init = false;
        }
        void start() override {
            assert(_block_init);
            std::lock_guard<std::recursive_mutex> lck(ctrlMtx);
            if (running) { return; }
            running = true;
            doStart();
        }
        void stop() override {
            assert(_block_init);
            std::lock_guard<std::recursive_mutex> lck(ctrlMtx);
            if (!running) { return; }
            doStop();
            running = false;
        }
        void tempStart() {
            if (!tempStopDepth || --tempStopDepth) { return; }
            if (tempStopped) {
                doStart();
                tempStopped = false;
            }
        }
        void tempStop() {
            if (tempStopDepth++) { return; }
            if (running && !tempStopped) {
                doStop();
                tempStopped = true;
            }
        }
        int run() override = 0;

    protected:
        void workerLoop() {
            while (run() >= 0) {}
        }
        virtual void doStart() { workerThread = std::thread(&block::workerLoop, this); }
        virtual void doStop() {
            for (auto& in : inputs) { in->stopReader(); }
            for (auto& out : outputs) { out->stopWriter(); }
            if (workerThread.joinable()) { workerThread.join(); }
            for (auto& in : inputs) { in->clearReadStop(); }
            for (auto& out : outputs) { out->clearWriteStop(); }
        }
        void registerInput(untyped_stream* s) { inputs.push_back(s); }
        void unregisterInput(untyped_stream* s) { inputs.erase(std::remove(inputs.begin(), inputs.end(), s), inputs.end()); }
        void registerOutput(untyped_stream* s) { outputs.push_back(s); }
        void unregisterOutput(untyped_stream* s) { outputs.erase(std::remove(outputs.begin(), outputs.end(), s), outputs.end()); }

        bool _block_init = false;
        std::recursive_mutex ctrlMtx;
        std::vector<untyped_stream*> inputs, outputs;
        bool running = false, tempStopped = false;
        int tempStopDepth = 0;
        std::thread workerThread;
    };
}
