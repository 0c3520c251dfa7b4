// exr_asan_main.cpp — host-only harness for the EXR reader under AddressSanitizer / UBSan (tests/test_exr.py builds it
// from csrc/exr_io.cpp + csrc/exr_codecs.cpp with g++ -fsanitize=address,undefined; no GPU, no HIP).
// usage: exr_asan_main file...   prints "<rc> <w>x<h>" per file; any memory error aborts with the sanitizer's report.
#include <cstdio>
#include <cstdlib>

#include "../../include/insploray_hip.h"

extern "C" void ir_free(void* p) { std::free(p); }  // lives in insploray_hip.hip, which is not linked here

int main(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
        float* rgb = nullptr;
        uint32_t w = 0, h = 0;
        const int rc = ir_exr_load(argv[i], &rgb, &w, &h);
        double sum = 0.0;
        if (rc == IR_OK)
            for (size_t k = 0; k < (size_t)w * h * 3; ++k) sum += rgb[k];  // touch every sample
        std::printf("%d %ux%u %s\n", rc, w, h, rc == IR_OK ? (sum == sum ? "ok" : "nan") : ir_exr_last_error());
        ir_free(rgb);
    }
    return 0;
}
