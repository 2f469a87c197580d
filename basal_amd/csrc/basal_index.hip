// basal_index.hip -- the seed index built on the GPU (RefSeq::CreateIndex, refbase.cpp:261-439).
//
// The reference counts k-mers, allocates 16-byte headers and fills per-k-mer lists with two
// threads (forward / reverse strand). Here the same table -- per 3-letter k-mer: forward entries
// ascending, then reverse-complement entries ascending, as global coordinates -- is produced as
//   1. one (key = 2*kmer + strand, value = global coordinate) pair per indexed position, emitted in
//      block order (which is ascending coordinate order within a strand) while the keys are
//      counted with atomics;
//   2. an exclusive scan of the counts = CSR offsets;
//   3. a STABLE radix sort of the pairs by key, so equal keys keep ascending coordinates.
// It is HBM-bound streaming work (8 B written + read per pass and pair), not a GEMM.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#include "basal_bits.h"
#include "basal_core_priv.h"
#include "basal_internal.h"

using namespace basal;

namespace {

#define HIP_TRYI(x)                                                         \
    do {                                                                    \
        hipError_t e_ = (x);                                                \
        if (e_ != hipSuccess) {                                             \
            set_error(std::string(#x) + ": " + hipGetErrorString(e_));      \
            return BASAL_EDEVICE;                                           \
        }                                                                   \
    } while (0)

struct BlockDesc {  // one unmasked block of one strand (RefSeq::_blocks) prepared for the device
    unsigned long long first;  // index of its first indexed position in the global enumeration
    unsigned long long word_base;  // first 64-bit word of its contig inside xref
    uint32_t i0, strand, anchor, pad;
};

// one thread per indexed position: hash the k-mer, count the key, emit the pair
__global__ __launch_bounds__(256) void emit_pairs(const uint64_t *__restrict__ xf, const uint64_t *__restrict__ xr, const BlockDesc *__restrict__ blk,
                                                  uint32_t nblk, unsigned long long npos, uint32_t K, uint32_t I, uint32_t *__restrict__ keys,
                                                  uint32_t *__restrict__ vals, uint32_t *__restrict__ cnt) {
    for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < npos; t += (unsigned long long)gridDim.x * blockDim.x) {
        uint32_t lo = 0, hi = nblk;  // last block with first <= t
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (blk[mid].first <= t) lo = mid;
            else hi = mid;
        }
        const BlockDesc b = blk[lo];
        uint32_t pos = b.i0 + (uint32_t)(t - b.first) * I;
        const uint64_t *w = (b.strand ? xr : xf) + b.word_base + (pos >> 5);
        uint32_t a = (pos & 31) * 2;
        uint64_t v = a ? (w[0] << a) | (w[1] >> (64 - a)) : w[0];  // s_MakeSeed_1, refbase.cpp:254-255
        uint32_t key = XT((uint32_t)(v >> (64 - 2 * K))) * 2 + b.strand;
        keys[t] = key;
        vals[t] = b.anchor + pos;  // hit2int, refbase.cpp:485-487
        atomicAdd(&cnt[key], 1u);
    }
}

__global__ __launch_bounds__(256) void split_counts(const uint32_t *__restrict__ off2, const uint32_t *__restrict__ cnt2, uint32_t total_kmers,
                                                    uint32_t nlocs, uint32_t *__restrict__ koff, uint32_t *__restrict__ knfwd) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < total_kmers) {
        koff[k] = off2[2 * k];
        knfwd[k] = cnt2[2 * k];
    } else if (k == total_kmers) koff[k] = nlocs;
}

// Flank words: for every index entry the 32 reference bases that follow its seed and the 32 that
// precede it, copied from the packed strand it lies on. The align kernel tests a candidate's flank
// against the read with one XOR/AND/popcount on the COALESCED location stream; the test can only
// under-count mismatches, so a candidate it rejects is one CountMismatch (align.h:118-131) rejects
// too, and only the survivors pay the random reference gather. One thread per k-mer.
// 32 bases of packed strand x starting at base p (MSB first)
__device__ __forceinline__ uint64_t flank_word(const uint64_t *__restrict__ x, uint32_t p) {
    const uint32_t a = (p & 31) * 2;
    const uint64_t *w = x + (p >> 5);
    return a ? (w[0] << a) | (w[1] >> (64 - a)) : w[0];
}
// far != 0 (cores whose index keeps long lists, basal_core::heavy): four more words per entry, the 64 bases beyond each near flank --
// [e+K+32, e+K+64) at fa[2 * stride + i], [e-64, e-32) at [3], [e+K+64, e+K+96) at [4], [e-96, e-64) at [5] -- and the entry's own 16 bases
// [e, e+16) in seedw[i] (top bits first): with those a long list's candidates are scored from the coalesced stream alone (basal_core.hip,
// heavy_mode). The reference keeps 400 margin words either side.
__device__ __forceinline__ void store_flanks(const uint64_t *__restrict__ x, uint32_t g, uint32_t K, unsigned long long i, unsigned long long stride, int far,
                                             uint64_t *__restrict__ fa, uint32_t *__restrict__ seedw) {
    if (!far) {
        fa[i] = flank_word(x, g + K);
        fa[stride + i] = flank_word(x, g - 32);
        return;
    }
    // (as bit planes, basal_bits.h split_planes: the HEAVY kernels compare a window in four instructions that way)
    fa[i] = split_planes(flank_word(x, g + K));
    fa[stride + i] = split_planes(flank_word(x, g - 32));
    fa[2 * stride + i] = split_planes(flank_word(x, g + K + 32));
    fa[3 * stride + i] = split_planes(flank_word(x, g - 64));
    fa[4 * stride + i] = split_planes(flank_word(x, g + K + 64));
    fa[5 * stride + i] = split_planes(flank_word(x, g - 96));
    seedw[i] = split_planes16((uint32_t)(flank_word(x, g) >> 32));
}
__global__ __launch_bounds__(256) void fill_flanks(const uint64_t *__restrict__ xf, const uint64_t *__restrict__ xr, const uint32_t *__restrict__ koff,
                                                   const uint32_t *__restrict__ knfwd, const uint32_t *__restrict__ locs, uint32_t total_kmers, uint32_t K,
                                                   unsigned long long stride, int far, uint64_t *__restrict__ fa, uint32_t *__restrict__ seedw) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total_kmers) return;
    uint32_t b = koff[k], e = koff[k + 1], nf = knfwd[k];
    for (uint32_t i = b; i < e; i++) store_flanks((i - b) >= nf ? xr : xf, locs[i], K, i, stride, far, fa, seedw);
}

// The same words, one thread per index ENTRY: right after the GPU build's sort the entry's strand is bit 0 of its sorted key,
// so no per-k-mer list walk is needed (coalesced location reads and flank stores; lists of 10^4 entries cost what 10^4 short ones do).
__global__ __launch_bounds__(256) void fill_flanks_sorted(const uint64_t *__restrict__ xf, const uint64_t *__restrict__ xr, const uint32_t *__restrict__ keys,
                                                          const uint32_t *__restrict__ locs, unsigned long long nlocs, uint32_t K, unsigned long long stride,
                                                          int far, uint64_t *__restrict__ fa, uint32_t *__restrict__ seedw) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < nlocs; i += (unsigned long long)gridDim.x * blockDim.x)
        store_flanks((keys[i] & 1u) ? xr : xf, locs[i], K, i, stride, far, fa, seedw);
}

// ---- GAP cores (-g > 0): the flanks as BIT PLANES, 64 bases each side ----------------------------------------------------------------------
// The gap search (GapAlign, align.cpp:348-410) accepts a candidate when a left part at the candidate's own start and a right part shifted by
// up to g bases cover the read with few mismatches; a lower bound of that needs the candidate's reference bases at every shift. Kept as the
// two bit planes of the base codes (u64 = high-bit plane << 32 | low-bit plane, bit i = base i of the window, LSB first), a shift of the
// window is a shift of two words, and the exact mismatch bitmap against the read is three bit-selects (v_bfi) of four per-read "mismatches
// letter X here" masks. Four words per entry: bases [e+K, e+K+32), [e-32, e), [e+K+32, e+K+64), [e-64, e-32) of the entry's strand.
__device__ __forceinline__ uint32_t even_bits(uint64_t x) {  // bit 2j of x -> bit j
    x &= 0x5555555555555555ULL;
    x = (x | (x >> 1)) & 0x3333333333333333ULL;
    x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0fULL;
    x = (x | (x >> 4)) & 0x00ff00ff00ff00ffULL;
    x = (x | (x >> 8)) & 0x0000ffff0000ffffULL;
    return (uint32_t)(x | (x >> 16));
}
__device__ __forceinline__ uint64_t plane_word(const uint64_t *__restrict__ x, uint32_t p) {
    const uint32_t a = (p & 31) * 2;
    const uint64_t *w = x + (p >> 5);
    const uint64_t v = a ? (w[0] << a) | (w[1] >> (64 - a)) : w[0];  // base j at bits 63-2j (high), 62-2j (low)
    const uint64_t rv = __brevll(v);                                  // base j: high bit at 2j, low bit at 2j+1
    return ((uint64_t)even_bits(rv) << 32) | even_bits(rv >> 1);
}
__device__ __forceinline__ void store_planes(const uint64_t *__restrict__ x, uint32_t g, uint32_t K, unsigned long long i, unsigned long long stride,
                                             uint64_t *__restrict__ pl) {
    pl[i] = plane_word(x, g + K);
    pl[stride + i] = plane_word(x, g - 32);
    pl[2 * stride + i] = plane_word(x, g + K + 32);
    pl[3 * stride + i] = plane_word(x, g - 64);
}
__global__ __launch_bounds__(256) void fill_planes(const uint64_t *__restrict__ xf, const uint64_t *__restrict__ xr, const uint32_t *__restrict__ koff,
                                                   const uint32_t *__restrict__ knfwd, const uint32_t *__restrict__ locs, uint32_t total_kmers, uint32_t K,
                                                   unsigned long long stride, uint64_t *__restrict__ pl) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total_kmers) return;
    uint32_t b = koff[k], e = koff[k + 1], nf = knfwd[k];
    for (uint32_t i = b; i < e; i++) store_planes((i - b) >= nf ? xr : xf, locs[i], K, i, stride, pl);
}
__global__ __launch_bounds__(256) void fill_planes_sorted(const uint64_t *__restrict__ xf, const uint64_t *__restrict__ xr, const uint32_t *__restrict__ keys,
                                                          const uint32_t *__restrict__ locs, unsigned long long nlocs, uint32_t K, unsigned long long stride,
                                                          uint64_t *__restrict__ pl) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < nlocs; i += (unsigned long long)gridDim.x * blockDim.x)
        store_planes((keys[i] & 1u) ? xr : xf, locs[i], K, i, stride, pl);
}

}  // namespace

// d_sorted_keys: the GPU build's sorted (2 * kmer + strand) keys, entry for entry with d_locs; nullptr after set_index (host-built arrays)
int basal_build_flanks(basal_core *c, const uint32_t *d_sorted_keys) {
    // one allocation, the "before" words right behind the "after" words: the kernel picks one by adding nlocs + 64 to the index
    hipFree(c->d_flank_a);
    c->d_flank_a = c->d_flank_b = nullptr;
    const unsigned long long stride = c->nlocs + 64;
    c->heavy = false;
    if (c->p.gap > 0) {  // the GAP kernels read bit planes, four words per entry (see fill_planes); the others never look at them
        HIP_TRYI(hipMalloc(&c->d_flank_a, 4 * stride * 8));
        c->d_flank_b = c->d_flank_a + stride;
        if (d_sorted_keys && c->nlocs) {
            unsigned long long want = (c->nlocs + 255) / 256;
            uint32_t grid = (uint32_t)std::min<unsigned long long>(want, (unsigned long long)c->prop.multiProcessorCount * 64);
            hipLaunchKernelGGL(fill_planes_sorted, dim3(grid), dim3(256), 0, 0, c->d_xref[0], c->d_xref[1], d_sorted_keys, c->d_locs,
                               (unsigned long long)c->nlocs, c->p.seed_size, stride, c->d_flank_a);
        } else
            hipLaunchKernelGGL(fill_planes, dim3((c->total_kmers + 255) / 256), dim3(256), 0, 0, c->d_xref[0], c->d_xref[1], c->d_koff, c->d_knfwd, c->d_locs,
                               c->total_kmers, c->p.seed_size, stride, c->d_flank_a);
        HIP_TRYI(hipGetLastError());
        HIP_TRYI(hipDeviceSynchronize());
        // (the same rule as below picks the HEAVY GAP kernels: the survivors' stage on bit planes, hits booked in bulk; the flank layout is the GAP kernels')
        const char *e = getenv("BASAL_HEAVY");
        c->heavy = e ? atoi(e) != 0 : c->max_kmer_num >= 32768;
        return BASAL_OK;
    }
    // Long lists are the rule where the over-represented-k-mer cut-off is high (a repeat-rich genome: 107 091 on the hg38-like stand-in,
    // 11 507 on the uniform one): such a core keeps two more words per entry and runs the kernel instantiation that streams long lists
    // through a three-window test (basal_core.hip, HEAVY). BASAL_HEAVY=0/1 overrides the rule (the tests run every fixture both ways).
    {
        const char *e = getenv("BASAL_HEAVY");
        c->heavy = e ? atoi(e) != 0 : c->max_kmer_num >= 32768;
    }
    const int far = c->heavy ? 1 : 0;
    HIP_TRYI(hipMalloc(&c->d_flank_a, (far ? 6 : 2) * stride * 8));
    c->d_flank_b = c->d_flank_a + stride;
    hipFree(c->d_seedw);
    c->d_seedw = nullptr;
    if (far) HIP_TRYI(hipMalloc(&c->d_seedw, stride * 4));
    if (d_sorted_keys && c->nlocs) {
        unsigned long long want = (c->nlocs + 255) / 256;
        uint32_t grid = (uint32_t)std::min<unsigned long long>(want, (unsigned long long)c->prop.multiProcessorCount * 64);
        hipLaunchKernelGGL(fill_flanks_sorted, dim3(grid), dim3(256), 0, 0, c->d_xref[0], c->d_xref[1], d_sorted_keys, c->d_locs, (unsigned long long)c->nlocs,
                           c->p.seed_size, stride, far, c->d_flank_a, c->d_seedw);
    } else
        hipLaunchKernelGGL(fill_flanks, dim3((c->total_kmers + 255) / 256), dim3(256), 0, 0, c->d_xref[0], c->d_xref[1], c->d_koff, c->d_knfwd, c->d_locs,
                           c->total_kmers, c->p.seed_size, stride, far, c->d_flank_a, c->d_seedw);
    HIP_TRYI(hipGetLastError());
    HIP_TRYI(hipDeviceSynchronize());
    return BASAL_OK;
}

extern "C" int basal_core_build_index(basal_core_t *c, const uint32_t *blocks, uint64_t nblocks, uint32_t *max_kmer_num_out) {
    if (!c || (!blocks && nblocks)) { set_error("build_index: null argument"); return BASAL_EINVAL; }
    if (!c->have_ref) { set_error("build_index: stage the reference first (basal_core_set_reference)"); return BASAL_ESTATE; }
    if (int rc = basal_core_placement_commit(c)) return rc;  // (a second placement kept aside would go stale)
    HIP_TRYI(hipSetDevice(c->device));
    const uint32_t K = c->p.seed_size, I = c->p.index_interval, total = c->total_kmers;
    // contig word bases and anchors come from the staged reference
    std::vector<uint32_t> anchor(c->ncontig + 1);
    HIP_TRYI(hipMemcpy(anchor.data(), c->d_anchor, (c->ncontig + 1) * 4, hipMemcpyDeviceToHost));
    std::vector<BlockDesc> bd;
    unsigned long long npos = 0;
    for (uint64_t b = 0; b < nblocks; b++) {
        uint32_t id = blocks[3 * b], beg = blocks[3 * b + 1], end = blocks[3 * b + 2];
        if ((id >> 1) >= c->ncontig || end < beg) { set_error("build_index: bad block (contig not staged, or end < begin)"); return BASAL_EINVAL; }
        if (end < K) continue;
        uint32_t i0 = (beg / I) * I, i2 = ((end - K) / I) * I;  // refbase.cpp:310-316
        if (i2 < i0) continue;
        BlockDesc d;
        d.first = npos;
        d.word_base = anchor[id >> 1] / 32;
        d.i0 = i0;
        d.strand = id & 1;
        d.anchor = anchor[id >> 1];
        d.pad = 0;
        bd.push_back(d);
        npos += (i2 - i0) / I + 1;
    }
    if (npos >= 0xFFFFFFFFull) { set_error("build_index: more than 2^32-1 index entries"); return BASAL_EINVAL; }
    hipFree(c->d_koff); hipFree(c->d_knfwd); hipFree(c->d_locs);
    c->d_koff = c->d_knfwd = c->d_locs = nullptr;
    c->have_index = false;
    uint32_t *d_keys = nullptr, *d_keys2 = nullptr, *d_vals = nullptr, *d_vals2 = nullptr, *d_cnt = nullptr, *d_off2 = nullptr;
    BlockDesc *d_blk = nullptr;
    void *d_tmp = nullptr;
    auto cleanup = [&]() { hipFree(d_keys); hipFree(d_keys2); hipFree(d_vals); hipFree(d_vals2); hipFree(d_cnt); hipFree(d_off2); hipFree(d_blk); hipFree(d_tmp); };
#define TRYC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(std::string(#x) + ": " + hipGetErrorString(e_)); cleanup(); return BASAL_EDEVICE; } } while (0)
    const size_t nkeys = (size_t)total * 2;
    TRYC(hipMalloc(&d_cnt, nkeys * 4));
    TRYC(hipMalloc(&d_off2, nkeys * 4));
    TRYC(hipMemset(d_cnt, 0, nkeys * 4));
    TRYC(hipMalloc(&c->d_koff, ((size_t)total + 1) * 4));
    TRYC(hipMalloc(&c->d_knfwd, (size_t)total * 4));
    TRYC(hipMalloc(&d_vals2, (npos + 64) * 4));  // becomes locs
    if (npos) {
        TRYC(hipMalloc(&d_keys, npos * 4));
        TRYC(hipMalloc(&d_keys2, npos * 4));
        TRYC(hipMalloc(&d_vals, npos * 4));
        TRYC(hipMalloc(&d_blk, bd.size() * sizeof(BlockDesc)));
        TRYC(hipMemcpy(d_blk, bd.data(), bd.size() * sizeof(BlockDesc), hipMemcpyHostToDevice));
        unsigned long long want = (npos + 255) / 256;
        uint32_t grid = (uint32_t)std::min<unsigned long long>(want, (unsigned long long)c->prop.multiProcessorCount * 32);
        hipLaunchKernelGGL(emit_pairs, dim3(grid), dim3(256), 0, 0, c->d_xref[0], c->d_xref[1], d_blk, (uint32_t)bd.size(), npos, K, I, d_keys, d_vals, d_cnt);
        TRYC(hipGetLastError());
    }
    size_t tmp_bytes = 0, tmp2 = 0;
    // item counts go in as size_t: an hg38-sized genome at -I 2 has 3.1 G entries, more than an int holds
    TRYC(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_cnt, d_off2, nkeys));
    if (npos) {
        int end_bit = 1;
        while ((1ull << end_bit) < nkeys) end_bit++;
        TRYC(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp2, d_keys, d_keys2, d_vals, d_vals2, (size_t)npos, 0, end_bit));
        tmp_bytes = std::max(tmp_bytes, tmp2);
    }
    TRYC(hipMalloc(&d_tmp, tmp_bytes + 256));
    TRYC(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_cnt, d_off2, nkeys));
    if (npos) {
        int end_bit = 1;
        while ((1ull << end_bit) < nkeys) end_bit++;
        size_t tb = tmp2;
        TRYC(hipcub::DeviceRadixSort::SortPairs(d_tmp, tb, d_keys, d_keys2, d_vals, d_vals2, (size_t)npos, 0, end_bit));
    }
    hipLaunchKernelGGL(split_counts, dim3((total + 1 + 255) / 256), dim3(256), 0, 0, d_off2, d_cnt, total, (uint32_t)npos, c->d_koff, c->d_knfwd);
    TRYC(hipGetLastError());
    TRYC(hipDeviceSynchronize());
    c->d_locs = d_vals2;
    d_vals2 = nullptr;
    c->nlocs = npos;
    // over-represented k-mer cut-off (refbase.cpp:362-363) from the per-k-mer totals
    {
        std::vector<uint32_t> off((size_t)total + 1);
        TRYC(hipMemcpy(off.data(), c->d_koff, ((size_t)total + 1) * 4, hipMemcpyDeviceToHost));
        uint32_t idx = kmer_cutoff_index(total, c->p.max_kmer_ratio);
        uint32_t mk;
        if (idx >= total - 1) mk = off[total] - off[total - 1];
        else {
            uint32_t mx = 0;
            for (uint32_t k = 0; k + 1 < total; k++) mx = std::max(mx, off[k + 1] - off[k]);
            std::vector<uint64_t> hist((size_t)mx + 1, 0);
            for (uint32_t k = 0; k + 1 < total; k++) hist[off[k + 1] - off[k]]++;
            uint64_t run = 0;
            uint32_t v = 0;
            for (; v <= mx; v++) {
                run += hist[v];
                if (run > idx) break;
            }
            mk = v;
        }
        c->max_kmer_num = mk;
    }
    // the flank words, while the sorted keys (strand bit per entry) are still there; the other build buffers go first (HBM head-room)
    hipFree(d_keys); hipFree(d_vals); hipFree(d_tmp); hipFree(d_cnt); hipFree(d_off2);
    d_keys = d_vals = d_cnt = d_off2 = nullptr; d_tmp = nullptr;
    if (int rc = basal_build_flanks(c, d_keys2)) { cleanup(); return rc; }
    cleanup();
#undef TRYC
    c->have_index = true;
    if (max_kmer_num_out) *max_kmer_num_out = c->max_kmer_num;
    return BASAL_OK;
}
