// TEST DOUBLE of the contract of SDR++'s dsp::stream<T> (core/src/dsp/stream.h:24-141) for building tests/host_cpp without the SDR++
// tree: double-buffered hand-off — the producer fills writeBuf and calls swap(n) (blocks until the consumer flushed the previous
// block; false when stopped); the consumer calls read() (-1 when stopped), uses readBuf[0..n) and calls flush(); the two buffer
// pointers are exchanged on every swap; stopWriter/stopReader unblock either side.
#pragma once
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <utility>
#include "types.h"
#define STREAM_BUFFER_SIZE 1000000
namespace dsp {
    class untyped_stream {
    public:
        virtual ~untyped_stream() {}
        virtual bool swap(int) { return false; }
        virtual int read() { return -1; }
        virtual void flush() {}
        virtual void stopWriter() {}
        virtual void clearWriteStop() {}
        virtual void stopReader() {}
        virtual void clearReadStop() {}
    };

    template <class T>
    class stream : public untyped_stream {
    public:
        stream() {
            writeBuf = (T*)aligned_alloc(64, sizeof(T) * STREAM_BUFFER_SIZE);
            readBuf = (T*)aligned_alloc(64, sizeof(T) * STREAM_BUFFER_SIZE);
        }
        ~stream() override {
            free(writeBuf);
            free(readBuf);
        }
        bool swap(int size) override {
            {
                std::unique_lock<std::mutex> lck(swapMtx);
                swapCV.wait(lck, [this] { return canSwap || writerStop; });
                if (writerStop) { return false; }
                dataSize = size;
                std::swap(writeBuf, readBuf);
                canSwap = false;
            }
            {
                std::lock_guard<std::mutex> lck(rdyMtx);
                dataReady = true;
            }
            rdyCV.notify_all();
            return true;
        }
        int read() override {
            std::unique_lock<std::mutex> lck(rdyMtx);
            rdyCV.wait(lck, [this] { return dataReady || readerStop; });
            return readerStop ? -1 : dataSize;
        }
        void flush() override {
            {
                std::lock_guard<std::mutex> lck(rdyMtx);
                dataReady = false;
            }
            {
                std::lock_guard<std::mutex> lck(swapMtx);
                canSwap = true;
            }
            swapCV.notify_all();
        }
        void stopWriter() override {
            {
                std::lock_guard<std::mutex> lck(swapMtx);
                writerStop = true;
            }
            swapCV.notify_all();
        }
        void clearWriteStop() override { writerStop = false; }
        void stopReader() override {
            {
                std::lock_guard<std::mutex> lck(rdyMtx);
                readerStop = true;
            }
            rdyCV.notify_all();
        }
        void clearReadStop() override { readerStop = false; }
        T* writeBuf;
        T* readBuf;

    private:
        std::mutex swapMtx, rdyMtx;
        std::condition_variable swapCV, rdyCV;
        bool canSwap = true, dataReady = false, readerStop = false, writerStop = false;
        int dataSize = 0;
    };
}
