// TEST DOUBLE of the two sample types of SDR++'s core/src/dsp/types.h that cross the hot-path boundary (layout only: interleaved
// float32 pairs).  Lets tests/host_cpp build the host mirror on a machine without the SDR++ tree; real integrations compile
// sdrplusplus_amd/host/*.h against SDR++'s own headers.
#pragma once
namespace dsp {
    struct complex_t { float re, im; };
    struct stereo_t { float l, r; };
}
