// basal_internal.h -- declarations shared by the translation units of libbasal_amd.so (not part of the C ABI).
#pragma once
#include <string>
namespace basal {
void set_error(const std::string &s);  // text returned by basal_last_error()
uint32_t kmer_cutoff_index(uint32_t total_kmers, float ratio);  // refbase.cpp:363, single precision
}

#if defined(__HIPCC__)
#include <cstdlib>
namespace basal {
// BASAL_POISON=<byte>: every device allocation of the library is filled with that byte. A debugging aid -- no result may depend on what a fresh
// allocation happens to hold (tests/test_gpu_multi.py runs the small-batch pipelines under several values).
inline hipError_t poison_malloc(void **p, size_t n) {
    const hipError_t e = ::hipMalloc(p, n);
    static const char *pz = getenv("BASAL_POISON");
    if (e == hipSuccess && pz && n) {  // (the pipelines' streams do not wait for the null stream: the fill must have ended before anybody uses the block)
        const hipError_t m = ::hipMemset(*p, (int)(strtol(pz, nullptr, 0) & 0xff), n);
        return m != hipSuccess ? m : ::hipDeviceSynchronize();
    }
    return e;
}
}  // namespace basal
#define hipMalloc(p, n) basal::poison_malloc((void **)(p), (n))
#endif
