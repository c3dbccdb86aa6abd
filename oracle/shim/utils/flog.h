// ORACLE stub for <utils/flog.h>: only flog::error(const char*) is used by iq_frontend.cpp.
#pragma once
#include <cstdio>
namespace flog {
    template <typename... Args> inline void error(const char* fmt, Args...) { fprintf(stderr, "[flog::error] %s\n", fmt); }
    template <typename... Args> inline void warn(const char* fmt, Args...) { fprintf(stderr, "[flog::warn] %s\n", fmt); }
    template <typename... Args> inline void info(const char* fmt, Args...) { (void)fmt; }
}
