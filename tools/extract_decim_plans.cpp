// Dumps the reference's power-of-two decimation plans (numeric FIR coefficient tables) to a
// binary fixture so that tests / bench / the standalone host mirror can build the same
// PowerDecimator cascades the reference builds (core/src/dsp/multirate/decim/plans.h,
// power_decimator.h:93-111).  Only numbers are extracted; no reference source is copied.
//
// Build + run (needs /root/reference, i.e. only in the build container):
//   g++ -std=c++17 -I/root/reference/core/src tools/extract_decim_plans.cpp -o /tmp/extract_plans
//   /tmp/extract_plans sdrplusplus_amd/data/decim_plans.bin
//
// File format (little endian):
//   char[4] "SDPL"; u32 version=1; u32 nplans;
//   per plan: u32 ratio; u32 nstages; per stage: u32 decimation; u32 ntaps; f32 taps[ntaps]
#include <cstdio>
#include <cstdint>
#include <dsp/multirate/decim/plans.h>

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s out.bin\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "wb");
    if (!f) { perror("fopen"); return 1; }
    using namespace dsp::multirate::decim;
    uint32_t ver = 1, n = plans_len;
    fwrite("SDPL", 1, 4, f);
    fwrite(&ver, 4, 1, f);
    fwrite(&n, 4, 1, f);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t ratio = 2u << i;  // plans[log2(ratio) - 1], power_decimator.h:100
        uint32_t ns = plans[i].stageCount;
        fwrite(&ratio, 4, 1, f);
        fwrite(&ns, 4, 1, f);
        uint32_t prod = 1;
        for (uint32_t s = 0; s < ns; s++) {
            uint32_t d = plans[i].stages[s].decimation, nt = plans[i].stages[s].tapcount;
            prod *= d;
            fwrite(&d, 4, 1, f);
            fwrite(&nt, 4, 1, f);
            fwrite(plans[i].stages[s].taps, 4, nt, f);
            printf("ratio %5u stage %u: decim %3u taps %3u\n", ratio, s, d, nt);
        }
        if (prod != ratio) { fprintf(stderr, "plan %u: product %u != ratio %u\n", i, prod, ratio); return 1; }
    }
    fclose(f);
    return 0;
}
