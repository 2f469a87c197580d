// basal_core_priv.h -- the core object shared by the HIP translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <mutex>
#include <vector>

#include "../../include/basal_core.h"

// the device-side staging of one basal_core_align_batch call in flight
struct CoreLane {
    uint8_t *d_bases = nullptr; size_t cap_bases = 0;
    basal_read *d_reads = nullptr; size_t cap_reads = 0;
    basal_stale *d_stales = nullptr; size_t cap_stales = 0;
    basal_result *d_results = nullptr;
    basal_hit *d_stream = nullptr; size_t cap_stream = 0;
    unsigned long long *d_used = nullptr;
    hipStream_t stream = nullptr;
    unsigned int *d_counter = nullptr;  // lanes 1..: their own queue head + guard ledger and per-wave hit logs (lane 0 uses the core's)
    basal_hit *d_scratch = nullptr;
    bool busy = false;
};
constexpr int kMaxLanes = 4;

struct basal_core {
    basal_params p;
    int device = 0;
    hipDeviceProp_t prop;
    // reference + index in HBM
    uint64_t *d_xref[2] = {nullptr, nullptr};
    ulonglong2 *d_xpl[2] = {nullptr, nullptr};  // GAP cores: the strands as bit planes, 16 bytes per 64 bases (the HEAVY GAP kernels' gap search reads them)
    uint64_t nwords = 0;
    uint32_t *d_anchor = nullptr, *d_size = nullptr, *d_rcoff = nullptr;
    uint32_t ncontig = 0;
    uint32_t *d_koff = nullptr, *d_knfwd = nullptr, *d_locs = nullptr;
    uint64_t *d_flank_a = nullptr, *d_flank_b = nullptr;  // per index entry: the 32 reference bases after / before the seed
    uint32_t *d_seedw = nullptr;  // heavy cores: per index entry its own 16 bases
    uint64_t nlocs = 0;
    uint32_t total_kmers = 0, max_kmer_num = 0;
    bool have_ref = false, have_index = false;
    void *alt[10] = {};      // a second placement of the long-lived buffers kept aside (basal_core_placement_fork / _swap / _commit)
    bool alt_valid = false;
    bool heavy = false;  // the index keeps long lists (high cut-off): four flank words per entry, HEAVY kernel instantiation (set by basal_build_flanks)
    uint8_t *d_tables = nullptr;
    // work buffers
    basal_hit *d_scratch = nullptr;
    uint32_t scratch_per_wave = 0;
    unsigned int *d_counter = nullptr;  // [0] work queue head, [1..24] guard ledger
    uint32_t grid = 0, last_grid = 0;
    int nwt = 0;
    // staging for the host-buffer entry points: lane 0 (also the paired-end entry point's); more lanes appear when several host threads call
    // basal_core_align_batch at once (the reference runs one SingleAlign per worker thread, main.cpp:60-92): each call takes a free lane
    CoreLane lane0;
    std::vector<CoreLane *> more_lanes;  // lazily created, at most kMaxLanes - 1
    std::mutex lane_m;
    std::condition_variable lane_cv;
    hipStream_t last_stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;  // around the align launch; around the pairing kernel
    bool timing = false, timed = false, pair_timed = false;
    // paired-end pairing on the device (basal_pe.hip)
    basal_pe_pair *d_pe_pairs = nullptr; size_t cap_pe_pairs = 0;
    basal_pe_rec *d_pe_recs = nullptr; size_t cap_pe_recs = 0;
    basal_hit *d_pe_work = nullptr; size_t cap_pe_work = 0;
    unsigned long long *d_pe_misc = nullptr;  // [0] records used, then nine 32-bit statistics
    // contig names for the device-side SAM writer (basal_core_set_contig_names)
    char *d_names = nullptr;
    uint32_t *d_name_off = nullptr;
    uint32_t n_names = 0;
};

// what a pipeline slot adds to an align launch (basal_pipe.hip)
struct basal_align_extra {
    const uint32_t *order = nullptr;  // device: read numbers to align (nullptr: 0..n-1)
    const uint32_t *n_ptr = nullptr;  // device: how many of them (nullptr: n)
    uint32_t ghost_base = 0xFFFFFFF0u;
    unsigned int *counter = nullptr;  // device: [0] queue head, [1..24] guard ledger of this slot
    basal_hit *scratch = nullptr;     // device: per-wave hit logs of this slot
};
int basal_launch_align(basal_core *c, const void *d_bases, uint64_t nbases_dev, const void *d_reads, uint32_t n, const void *d_stales, uint32_t nstale,
                       uint32_t max_len, int stream_mode, void *d_results, void *d_stream, uint64_t stream_cap, void *d_stream_used, hipStream_t s,
                       const basal_align_extra *ex);
int basal_launch_align_carry(basal_core *c, const void *d_bases, uint64_t nbases_dev, const void *d_reads, uint32_t n, const void *d_stales, uint32_t nstale,
                             uint32_t max_len, int stream_mode, void *d_results, void *d_stream, uint64_t stream_cap, void *d_stream_used, const uint8_t carry[2][2],
                             hipStream_t s, const basal_align_extra *ex);
int basal_fill_async(void *p, size_t bytes, uint32_t value, hipStream_t s);  // bytes: a multiple of 4; a kernel, not a memset (see the definition)
int basal_pe_enqueue(basal_core *c, const void *d_reads, const void *d_results, const void *d_stream, void *d_work, uint32_t npairs, void *d_pairs, void *d_recs,
                     uint64_t recs_cap, void *d_recs_used, void *d_stats, hipStream_t s);
int basal_validate_batch(const basal_params &P, const basal_read *reads, uint32_t n, uint64_t nbases, const basal_stale *stales, uint32_t nstale, const char *who,
                         uint32_t *max_len_out);  // the descriptor checks every host-buffer entry point makes (BASAL_EINVAL + message)
int basal_ensure_launch_geometry(basal_core *c);  // sizes c->grid (largest grid any instantiation uses) and the core's own scratch
int basal_report_guard(const unsigned int *guard);  // BASAL_OK, or BASAL_EDEVICE + message if the kernel's bounds ledger is not clean

int basal_build_flanks(basal_core *c, const uint32_t *d_sorted_keys);
