// basal_internal.h -- declarations shared by the translation units of libbasal_amd.so (not part of the C ABI).
#pragma once
#include <string>
namespace basal {
void set_error(const std::string &s);  // text returned by basal_last_error()
uint32_t kmer_cutoff_index(uint32_t total_kmers, float ratio);  // refbase.cpp:363, single precision
}
