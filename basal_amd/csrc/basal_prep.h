// basal_prep.h -- device-side read preparation and SAM assembly (shared by basal_prep.hip and basal_pipe.hip; not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/basal_core.h"
#include "basal_core_priv.h"

namespace basal {

// the Param fields the prep / format kernels read, by value in the kernel arguments
struct PrepConst {
    uint32_t K, I, max_readlen, min_read_size, max_ns, trim_qual, zero_qual, default_qual, n_adapter, max_snp_num, gap, chains;
    uint32_t out_unmap, out_ref, report_repeat_hits;
    uint8_t adapter[10][16];  // TrimAdapter compares at most the first 15 bases (align.cpp:418-435)
    uint8_t adapter_len[10];  // min(strlen, 15)
    char useful_nt[8];
};

// what FilterReads (align.cpp:548-563) leaves of one read, for the SAM writer
struct ReadAux {
    uint16_t seq_len;   // bases printed (after adapter / quality trimming)
    uint16_t qual_len;  // qualities printed
    uint8_t qc_failed;  // FilterReads returned 1
    uint8_t qual_fill;  // the quality string was replaced by seq_len default characters (align.cpp:54-57)
    uint8_t cls;        // read-length class of the align kernel (0: <= 128, 1: <= 256, 2: <= 480), 3 = not aligned
    uint8_t pad;
};

// What a SingleAlign object carries from read to read (align.cpp:475-480, align.h:73,90), kept on the device across batches:
// per slot (0: SE reads and mate 1, 1: mate 2) the stack of earlier reads whose seed slots a later, shorter read can still see
// (most recent first, seed-slot counts strictly increasing), and the last read that defined the start offset.
constexpr int kStackMax = 480;
struct CarryRead {
    uint16_t npos, len;
    uint8_t readset, valid, pad[2];
    uint8_t seq[BASAL_MAXREADLEN];
};
struct CarryState {
    uint32_t depth[2];
    uint32_t next_index;  // the global read number the next batch starts at (text form: only the device has counted the reads)
    uint32_t seq;         // the number of the batch this state is the start of: carry_update checks that batch b really starts from the state batch b - 1 left
                          // (a hand-over between streams or GPUs that ran out of order would otherwise show only as a different read number -- a different
                          // rotation start and pick for reads with several equally good hits -- in one batch)
    CarryRead ghost[2];             // last defining read per slot (valid = 0: none yet)
    CarryRead stack[2][kStackMax];  // [slot][0] = most recent
};

// per-batch device counters
struct BatchCounters {
    unsigned long long n_aligned, n_unique, n_multiple, n_filtered;
    unsigned long long out_bytes;    // SAM bytes of the batch
    unsigned long long stream_used;  // hit-stream records used by the align kernels (-r 2)
    uint32_t n_reads;                // reads in the batch (device-parsed text: known here first)
    uint32_t n_lines;
    uint32_t irregular;              // bit 0: text form, the text is not 4 (2) regular lines per record; bit 1: a hit stream overflowed; bit 2: the carry state this batch
                                     // started from was not the one the batch before it left (CarryState::seq)
    uint32_t n_stale;
    uint32_t cls_n[3];               // reads per length class
    uint32_t pair_err;               // paired-end: a pair whose names differ from the first character on (FixPairReadName exits there) ...
    uint32_t pair_err_at;            // ... and the first such pair's number in the batch
    uint32_t pe[9];                  // paired-end: aligned / unique / multiple for pairs, mate 1, mate 2 (pairs.cpp's counters)
    unsigned long long pe_recs_used; // paired-end: records the pairing kernel wanted to write
};

// everything one batch slot owns on the device
struct SlotDev {
    uint8_t *text = nullptr;        // the batch's byte blob (+ 2 x 512 bytes behind it for the ghost reads)
    uint64_t text_cap = 0;
    basal_rawread *raw = nullptr;   // [max_reads]
    basal_read *desc = nullptr;     // [max_reads + 2] (the last two: ghost reads)
    ReadAux *aux = nullptr;         // [max_reads]
    basal_stale *stales = nullptr;  // [max_reads], indexed by read number
    uint16_t *npos = nullptr;       // [2][max_reads] seed-slot count of aligned reads per slot, else 0
    uint16_t *bmax1 = nullptr, *bmax2 = nullptr;  // maxima of npos over 64 / 4096 reads, per slot
    int32_t *defidx = nullptr;      // [2][max_reads] read number if it defines its slot's start offset, else -1; then its exclusive max-scan
    uint32_t *order = nullptr;      // [3][max_reads] read numbers per length class
    basal_result *results = nullptr;
    basal_hit *stream = nullptr;
    uint64_t stream_cap = 0;
    uint32_t *nl = nullptr;         // newline positions (text form)
    uint32_t *blk_cnt = nullptr;    // newline count per 4 KB block, then its exclusive scan
    unsigned long long *out_off = nullptr;  // [max_reads + 1] SAM byte offset of every read
    uint8_t *out = nullptr;         // SAM text
    uint64_t out_cap = 0;
    basal_pe_pair *pe_pairs = nullptr;  // paired-end pipes: [max_reads / 2] per pair, where its records start
    basal_pe_rec *pe_recs = nullptr;    // the records to print (pairing kernel, basal_pe.hip)
    uint64_t pe_recs_cap = 0;
    basal_hit *pe_work = nullptr;       // the pairing kernel's sorted copies of the mates' logs: as many records as the hit stream
    uint64_t pe_work_cap = 0;
    BatchCounters *cnt = nullptr;
    unsigned int *counter = nullptr;  // [3][32]: align queue head + guard ledger per class launch
    basal_hit *scratch = nullptr;     // per-wave hit logs
    void *cub_tmp = nullptr;
    size_t cub_tmp_bytes = 0;
};

struct PrepShared {  // per pipe, shared by the slots
    // ring of carry states: batch b reads [b % ncarry], writes [(b + 1) % ncarry]; ncarry = batches in flight (depth x GPUs of the pipe) + 1, so the state a batch started
    // from survives until every batch that was in flight with it has been collected (a batch can be re-submitted from it)
    CarryState *carry[129] = {nullptr};
    uint32_t ncarry = 0;
    const char *names = nullptr;                // contig names blob
    const uint32_t *name_off = nullptr;         // [ncontig + 1]
};

int prep_make_const(const basal_params &p, PrepConst &k);
// queue: text -> raw table (text form only)
// first_index 0xFFFFFFFF: continue from the carry state's next_index
// pair_n != 0: the text is mate 1's pair_n records (pair_split bytes) followed by mate 2's pair_n records; the table comes out interleaved
int prep_enqueue_index_text(basal_core *c, SlotDev &s, const PrepShared &sh, uint32_t batch_no, uint64_t nbytes, int format, uint32_t first_index, uint32_t read_end,
                            uint32_t readset, uint32_t max_reads, uint32_t pair_split, uint32_t pair_n, hipStream_t st);
// queue: raw table -> descriptors, QC, stale table, length-class lists, carry state of the next batch
int prep_enqueue_filter(basal_core *c, const PrepConst &k, SlotDev &s, const PrepShared &sh, uint32_t batch_no, uint32_t max_reads, bool n_on_device,
                        uint32_t n_host, hipStream_t st);
// queue: results -> SAM text in s.out
int prep_enqueue_format(basal_core *c, const PrepConst &k, SlotDev &s, const PrepShared &sh, uint32_t max_reads, hipStream_t st);
// paired-end pipes: FixPairReadName + the every-mode mark (behind the filter), and the pairing kernel's records -> SAM text in s.out
int prep_enqueue_pair_fix(basal_core *c, SlotDev &s, uint32_t npairs, hipStream_t st);
int prep_enqueue_format_pe(basal_core *c, const PrepConst &k, SlotDev &s, const PrepShared &sh, uint32_t npairs, uint32_t max_reads, hipStream_t st);
size_t prep_cub_tmp_bytes(uint32_t max_reads, uint64_t max_bytes);

}  // namespace basal
