// Internal host helpers shared by the C-ABI implementation (not part of the public ABI).
#pragma once
#include <cstdint>
#include <vector>
namespace sdrpp_host {
void twiddle(int e, int L, float* re, float* im);
void zoomTable(int offset, int width, int inSize, int outSize, std::vector<int32_t>& start, std::vector<int32_t>& count);
double turnsPerSample(float re, float im);
}
