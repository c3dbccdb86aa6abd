// ORACLE stub for <core.h>: IQFrontEnd only calls core::setInputSampleRate (iq_frontend.cpp:128).
#pragma once
namespace core {
    inline double lastInputSampleRate = 0.0;
    inline void setInputSampleRate(double sr) { lastInputSampleRate = sr; }
}
