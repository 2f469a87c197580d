/*
 * basal_core.h -- C ABI of the MI355X-native BASAL alignment core (libbasal_amd.so).
 *
 * This is the drop-in boundary for the seed-and-extend hot path of JiejunShi/BASAL.  The
 * reference has no plugin API; the seam is the SingleAlign object that its host drives
 * (align.h:29-116), used from main.cpp:60-84 (t_SingleAlign: ImportBatchReads -> Do_Batch ->
 * _str_align) and from pairs.cpp:132-202 (PairAlign drives two SingleAligns mode by mode).
 * Each entry point below names the reference interface it replaces.  Plain pointers and
 * sizes only; no C++/HIP/torch types cross this boundary.  All functions return 0 on success
 * or a negative BASAL_E* code (the reference exit(1)s instead); basal_last_error() gives text.
 *
 * Two groups:
 *   basal_core_*  the GPU core: reference + seed index resident in HBM, batches of reads in,
 *                 fixed-size hit records out.  Replaces SingleAlign::{RunAlign, ConvertBina(r)ySeq,
 *                 ReorderSeed, SnpAlign, GapAlign, AddHit, int2hit} and the hit selection of
 *                 StringAlign (align.cpp:79-612).
 *   basal_host_*  host-side helpers that produce exactly what the core consumes and turn
 *                 what it emits into SAM: Param::SetAlign (param.cpp:163-263), RefSeq::
 *                 Run_ConvertBinseq/CreateIndex (refbase.cpp:186-439), FilterReads
 *                 (align.cpp:548-563), s_OutHit (align.cpp:616-669).  The `basal` CLI is built
 *                 from these; a maintainer of the reference would call basal_core_* from
 *                 SingleAlign::Do_Batch instead.
 */
#ifndef BASAL_CORE_H
#define BASAL_CORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BASAL_OK 0
#define BASAL_EINVAL (-1)   /* bad argument / unsupported parameter combination */
#define BASAL_ENOMEM (-2)   /* host or device allocation failed */
#define BASAL_EDEVICE (-3)  /* HIP runtime error, or no gfx950 device */
#define BASAL_ESTATE (-4)   /* call order (e.g. align before set_reference) */
#define BASAL_EOVERFLOW (-5)/* hit stream capacity too small; results for flagged reads incomplete */
#define BASAL_EIO (-6)

#define BASAL_MAXSNPS 15    /* param.h:18 */
#define BASAL_MAXGAPS 3     /* param.h:19 */
#define BASAL_MAXHITS 1000  /* makefile:4 */
#define BASAL_MAXREADLEN 480 /* (FIXELEMENT-1)*SEGLEN, param.cpp:58 */
#define BASAL_REF_MARGIN 400 /* refbase.h:16, in 64-bit words */

/* Every Param field the hot path reads (param.h:44-148) plus the 2-bit code tables that
 * Param::SetAlign derives from -M (param.cpp:163-263). Fill with basal_host_params_*. */
typedef struct basal_params {
    uint32_t seed_size;          /* -s, 10..16 */
    uint32_t index_interval;     /* -I, 1..16 */
    uint32_t max_snp_num;        /* -v as stored by main.cpp:324-338 (>=100 means percent+100) */
    uint32_t max_num_hits;       /* -w, <=1000 */
    uint32_t chains;             /* -n */
    uint32_t randseed;           /* -S; 0 is not reproducible in the reference, here it is the hash RNG with seed 0 */
    uint32_t gap;                /* -g, <=3 */
    uint32_t gap_edge;           /* 6 */
    uint32_t n_mis;              /* hidden -N */
    uint32_t new_rule;           /* 1: CountMismatch_new/MismatchPattern*_new (align.cpp:300) */
    uint32_t report_repeat_hits; /* -r */
    uint32_t pairend;
    uint32_t max_ns, min_read_size, max_readlen, trim_qual_threshold, n_adapter;
    uint32_t out_ref, out_unmap, min_insert, max_insert;
    float max_kmer_ratio;        /* -k */
    uint8_t zero_qual, default_qual, pad0, pad1;
    char adapter[10][128];
    char refnt, readnts[5];
    uint8_t readnt_cnt, pad2;
    uint8_t alphabet[256], rev_alphabet[256], reg_alphabet[256];
    uint8_t alphabet_mread[256], rev_alphabet_mread[256];
    char useful_nt[12];
} basal_params;

/* One stored alignment: gHit (param.h:35-42) plus where AddHit (align.h:329-347) put it. */
typedef struct basal_hit {
    uint32_t loc;      /* 0-based on the forward strand of the contig */
    uint32_t chr;      /* 2*contig + ref_strand, 18 bits like the reference's bit-field */
    int8_t gap_size;   /* >0 deletion from the read (CIGAR D), <0 insertion (CIGAR I) */
    uint8_t strand;    /* (ref_strand<<1)|read_chain */
    uint16_t gap_pos;  /* 9 bits */
    uint8_t level;     /* mismatch level w the hit was stored at (xhits[chain][w]) */
    uint8_t chain;     /* read chain: 0 as sequenced, 1 reverse complement */
    uint8_t mode;      /* SnpAlign mode (seed segment rank) that produced it; PE truncation */
    uint8_t pad;
} basal_hit;

/* Per-read summary: what StringAlign (align.cpp:583-612) needs. 32 bytes. */
typedef struct basal_result {
    basal_hit best;       /* the hit printed for a unique read or for -r 1 (myrand pick); valid if best_level!=0xFF */
    uint16_t n_hit;       /* _cur_n_hit[best_level]  */
    uint16_t n_chit;      /* _cur_n_chit[best_level] */
    uint8_t best_level;   /* lowest non-empty level, 0xFF if the read has no hit */
    uint8_t start_off[2]; /* xseed_start_offset[chain] after this read (align.cpp:475-480 carry) */
    uint8_t status;       /* BASAL_READ_* */
    uint32_t stream_first;/* first record of this read in the hit stream */
    uint32_t stream_n;    /* number of records of this read in the hit stream */
} basal_result;

#define BASAL_READ_OK 0
#define BASAL_READ_SKIPPED 1   /* len==0 descriptor (QC-failed on the host) */
#define BASAL_READ_OVERFLOW 2  /* hit stream full: stream_n is what was needed, nothing written */

/* One read as FilterReads (align.cpp:548-563) leaves it. 16 bytes. */
typedef struct basal_read {
    uint32_t seq_off;   /* offset of the first base in the batch's base buffer */
    uint32_t index;     /* ReadInf.index: global read number, feeds myrand */
    uint16_t len;       /* 0: skip */
    uint8_t readset;    /* 0 SE, 1 mate 1, 2 mate 2; | BASAL_READ_ALLMODES for a mate aligned by PairAlign::RunAlign */
    uint8_t max_snp;    /* read_max_snp_num */
    uint32_t stale_idx; /* BASAL_STALE_NONE, or this read's entry in the batch's basal_stale table */
} basal_read;
#define BASAL_READ_ALLMODES 0x80 /* PE (pairs.cpp:164-174): run every SnpAlign mode, no per-read early stop; the host
                                    replays GetPairs mode by mode over the mode-tagged hit log (BASAL_STREAM_ALL) */
#define BASAL_STALE_NONE 0xFFFFFFFFu
#define BASAL_STALE_CARRY 0xFFFFFFFEu

/* State a read inherits from EARLIER reads of the same SingleAlign object. The reference keeps
 * xseed_start_offset and xseed_array as members that a read with (len-I+1)%k==0 does not rewrite
 * (align.cpp:475-480): such a read reuses the previous start offset and, through it, may index
 * seed slots beyond its own length that still hold an earlier, longer read's seeds. Reproducing
 * `-p 1` output bit for bit needs both; basal_host_stale_* computes this table on the host. */
typedef struct basal_stale {
    uint32_t src;             /* read number in this batch whose start offset is inherited, or BASAL_STALE_CARRY */
    uint32_t overlay[2][15];  /* [chain][j]: xseed_array[chain][npos+j] as left by earlier reads: 3-letter seed
                                 hash, bit 31 = window held a non-ACGT base; npos = len-k+1 of this read */
} basal_stale;

/* what goes into the hit stream */
#define BASAL_STREAM_NONE 0 /* -r 0/1 SE: basal_result.best is enough */
#define BASAL_STREAM_BEST 1 /* -r 2 SE: all hits of the best level, chain 0 then chain 1, insertion order */
#define BASAL_STREAM_ALL 2  /* PE: every stored hit, insertion order, tagged with level/chain/mode */

typedef struct basal_core basal_core_t;

/* ---- GPU core ---- */
/* device: HIP device ordinal. Replaces `SingleAlign a;` (main.cpp:61). */
int basal_core_create(const basal_params *p, int device, basal_core_t **out);
void basal_core_destroy(basal_core_t *c);

/* Stage the 2-bit reference (RefSeq::xref[2], ref_anchor, title; refbase.cpp:186-252) and the
 * seed index (RefSeq::index2, refbase.cpp:261-439, flattened: kmer_off[3^k+1] CSR offsets into
 * locs, kmer_nfwd[3^k] = number of forward-strand entries of each k-mer) into HBM, once.
 * Host pointers; the core copies. */
int basal_core_set_reference(basal_core_t *c, const uint64_t *xref_fwd, const uint64_t *xref_rc, uint64_t nwords,
                             const uint32_t *ref_anchor, const uint32_t *contig_size, const uint32_t *rc_offset,
                             uint32_t ncontig);
int basal_core_set_index(basal_core_t *c, const uint32_t *kmer_off, const uint32_t *kmer_nfwd, const uint32_t *locs,
                         uint64_t nlocs, uint32_t max_kmer_num);
/* Build the seed index on the GPU from the staged reference (same result as set_index with the
 * host-built arrays; refbase.cpp:261-439). blocks = RefSeq::_blocks (id,begin,end triples). */
int basal_core_build_index(basal_core_t *c, const uint32_t *blocks, uint64_t nblocks, uint32_t *max_kmer_num_out);
/* Copy the device index back (tests; handing the index to a CPU baseline). Any pointer may be NULL. */
int basal_core_get_index(basal_core_t *c, uint32_t *kmer_off, uint32_t *kmer_nfwd, uint32_t *locs, uint64_t *nlocs,
                         uint32_t *max_kmer_num);

/* Align one batch; host buffers in, host buffers out. Replaces SingleAlign::Do_Batch's RunAlign
 * loop (align.cpp:565-580). carry[slot][chain]: xseed_start_offset inherited from the previous
 * batch (slot 0: SE reads and mate 1, slot 1: mate 2); updated on return.
 * stream may be NULL when stream_mode==BASAL_STREAM_NONE. */
int basal_core_align_batch(basal_core_t *c, const uint8_t *bases, uint64_t nbases, const basal_read *reads, uint32_t n,
                           const basal_stale *stales, uint32_t nstale, int stream_mode, basal_result *results,
                           basal_hit *stream, uint64_t stream_cap, uint64_t *stream_used, uint8_t carry[2][2]);

/* Same, but every buffer is already resident in HBM (device pointers) and the work is queued on
 * `stream` (a hipStream_t passed as void*, NULL = the null stream) without synchronising.
 * d_stream_used: device uint64 counter, must be zero on entry. max_len: an upper bound of the read
 * lengths in the batch (<=480); it selects the kernel instantiation (reads longer than it are skipped).  This is the entry point the
 * benchmark times and the one a multi-GPU driver uses before its RCCL gather of results. */
int basal_core_align_batch_device(basal_core_t *c, const void *d_bases, const void *d_reads, uint32_t n,
                                  const void *d_stales, uint32_t nstale, int stream_mode, void *d_results,
                                  void *d_stream, uint64_t stream_cap, void *d_stream_used, const uint8_t carry[2][2],
                                  uint32_t max_len, void *hip_stream);

/* Wait for the work queued by the last align_batch_device call and report what the kernel's bounds
 * ledger recorded: BASAL_OK, or BASAL_EDEVICE with the kind of violation in basal_last_error()
 * (an index outside its array is never dereferenced on the device; it is counted instead). */
int basal_core_sync_check(basal_core_t *c);

/* Kernel-time instrumentation: milliseconds of the last align_batch* kernel measured with HIP
 * events on the stream it ran on (0 if timing is off). */
int basal_core_set_timing(basal_core_t *c, int on);
float basal_core_last_kernel_ms(basal_core_t *c);
float basal_core_last_pair_ms(basal_core_t *c); /* the pairing kernel of the last basal_core_align_pairs_batch */
/* launch geometry actually used (blocks, threads per block, waves) for reporting */
int basal_core_launch_info(basal_core_t *c, uint32_t *blocks, uint32_t *threads, uint32_t *lds_bytes);

const char *basal_last_error(void);

/* ---- device-side read preparation, SAM assembly and the batch pipeline ----
 * The reference's host does, per 50 000-read batch and single-threaded: parse the input (ReadClass::LoadBatchReads,
 * reads.cpp:42-110), QC-filter every read (SingleAlign::FilterReads, align.cpp:548-563 with TrimAdapter 418-435,
 * TrimLowQual 51-76, CountNs 40-47), align, and print SAM (StringAlign/s_OutHit, align.cpp:583-669). basal_pipe_* runs
 * all of that on the GPU: raw FASTQ/FASTA text (or a table of raw reads over a byte blob) goes in, finished SAM text
 * comes out, several batches in flight (H2D of batch k+1 and D2H of batch k-1 overlap the kernels of batch k).
 * State the reference carries from read to read inside one SingleAlign object (basal_stale) is tracked on the device
 * across batches, so the text equals the reference's `-p 1` output whatever the batch size. */

/* Contig names for the RNAME column (RefSeq::title[].name). names: ncontig NUL-terminated strings. */
int basal_core_set_contig_names(basal_core_t *c, const char *const *names, uint32_t ncontig);

/* One raw read as the input decoder leaves it (ReadInf, reads.h:17-23; before FilterReads): where its name, bases and
 * qualities lie in the batch's byte blob. 24 bytes. */
typedef struct basal_rawread {
    uint32_t name_off, seq_off, qual_off; /* byte offsets into the blob */
    uint16_t name_len, seq_len, qual_len; /* qual_len 0 with seq_len > 0: no qualities (FASTA reads) */
    uint8_t readset, pad;                 /* 0 SE, 1 mate 1, 2 mate 2 */
    uint32_t index;                       /* ReadInf.index: global read number */
} basal_rawread;

typedef struct basal_pipe basal_pipe_t;
typedef struct basal_pipe_opts {
    uint32_t depth;          /* batches in flight (2..8; 0 = 3) */
    uint32_t max_reads;      /* reads per batch (0 = 4 Mi) */
    uint64_t max_bytes;      /* input bytes per batch, < 4 GiB (0 = 512 MiB) */
    uint32_t output;         /* BASAL_PIPE_OUT_* */
    uint32_t flags;          /* BASAL_PIPE_PAIRS */
} basal_pipe_opts;
#define BASAL_PIPE_PAIRS 1u      /* paired-end (PairAlign::Do_Batch, pairs.cpp:179-202): submit_records takes mate pairs interleaved (a0, b0, a1, b1, ...;
                                  * readset 1 / 2), FilterReads x 2, FixPairReadName, both mates' alignment, the pairing rounds and the text of
                                  * s_OutHitPair / s_OutHitUnpair all run on the device; output is the pairs' SAM text, stats->pe the nine counters */
#define BASAL_PIPE_OUT_SAM 0     /* SAM text (s_OutHit), reads in input order */
#define BASAL_PIPE_OUT_RESULTS 1 /* basal_result[n] (+ nothing else): what a multi-GPU gather moves */
#define BASAL_FMT_FASTQ 0
#define BASAL_FMT_FASTA 1

typedef struct basal_batch_stats { /* main.cpp:606-612 */
    uint64_t n_reads, n_aligned, n_unique, n_multiple, n_filtered;
    float ms_h2d, ms_prep, ms_align, ms_format, ms_d2h; /* HIP-event times of this batch's stages on its stream */
    uint32_t pe[9];  /* BASAL_PIPE_PAIRS: aligned / unique / multiple for pairs, mate 1, mate 2 (main.cpp:114-117) */
    uint32_t pad;
} basal_batch_stats;

int basal_pipe_create(basal_core_t *c, const basal_pipe_opts *o, basal_pipe_t **out);
/* The same pipe over several GPUs (one staged core each; the same parameters, reference and index on all of them): whole batches fan out
 * over the GPUs like the batches of the reference's worker threads (main.cpp:60-92, 154-166), `depth` in flight per GPU, each batch on one
 * GPU from its text to its SAM bytes; the state a SingleAlign carries from read to read (align.cpp:475-480) follows the batch numbers from
 * GPU to GPU, so the output is the one-GPU output. Every other basal_pipe_* call works on it unchanged. */
int basal_pipe_create_multi(basal_core_t *const *cores, int ncores, const basal_pipe_opts *o, basal_pipe_t **out);
void basal_pipe_destroy(basal_pipe_t *p);
/* The next free batch slot's page-locked input buffers (blocks while all `depth` slots are in flight and uncollected).
 * blob: max_bytes bytes; raw: max_reads entries (used by submit_records only; may be ignored). */
int basal_pipe_acquire(basal_pipe_t *p, uint8_t **blob, basal_rawread **raw);
/* Submit the acquired slot. text form: `nbytes` of FASTQ (4 lines per record) or FASTA-reads (2 lines per record) text,
 * whole records only, ending in a newline; the device finds the lines; reads are numbered first_index, first_index+1, ...
 * (0xFFFFFFFF: continue where the batch before stopped -- only the device has counted its reads) with the given readset.
 * Returns BASAL_EINVAL-class errors at once; a text the device finds irregular (blank lines, wrapped sequences, white space
 * inside a sequence line: anything where iostream token parsing and line parsing differ) is reported by collect as BASAL_EIO
 * for that batch -- re-submit it through submit_records after parsing it on the host. */
int basal_pipe_submit_text(basal_pipe_t *p, uint64_t nbytes, int format, uint32_t first_index, uint32_t readset);
int basal_pipe_submit_records(basal_pipe_t *p, uint64_t nblob, uint32_t n);
/* BASAL_PIPE_PAIRS pipes, text form: the blob holds mate 1's `npairs` records (`split` bytes of FASTQ / FASTA-reads text) followed by mate
 * 2's `npairs` records (the caller only has to count lines to cut the two files alike); pair i gets read number first_index + i. A text the
 * device finds irregular -- or whose halves do not hold npairs records each -- is refused like submit_text's (collect: BASAL_EIO). */
int basal_pipe_submit_text_pairs(basal_pipe_t *p, uint64_t nbytes, uint64_t split, uint32_t npairs, int format, uint32_t first_index);
/* Already QC-filtered reads (the basal_core_align_batch input) in the acquired slot's blob: bases at blob[0..nbases), the n
 * descriptors written to (basal_read *)raw by the caller. Output must be BASAL_PIPE_OUT_RESULTS. */
int basal_pipe_submit_prepared(basal_pipe_t *p, uint64_t nbases, uint32_t n, uint32_t max_len);
/* Wait for the OLDEST submitted batch. out/nbytes: its output in page-locked host memory (SAM text, or basal_result[n]),
 * valid until basal_pipe_release or the next collect; stats may be NULL. Returns BASAL_ESTATE when nothing is in flight.
 * BASAL_EIO: the batch's text is irregular (see submit_text); the pipe is now stopped, see basal_pipe_rewind: re-submit
 * this batch (as records) and what followed it.
 * One thread may acquire/submit while another collects/releases. */
int basal_pipe_collect(basal_pipe_t *p, const void **out, uint64_t *nbytes, basal_batch_stats *stats);
/* Done with the output collect handed out: its slot can take a new batch (the next collect does this by itself). */
int basal_pipe_release(basal_pipe_t *p);
/* Give an acquired slot back without submitting it. */
int basal_pipe_cancel(basal_pipe_t *p);
/* After collect returned BASAL_EIO the pipe is stopped: acquire / submit / collect return BASAL_ESTATE (a blocked acquire wakes
 * up) until the caller -- with its submitting thread quiet -- calls basal_pipe_rewind, which drops everything in flight and
 * continues with the refused batch's number and the device state that batch started from. basal_pipe_stop stops the pipe the
 * same way without a refused batch (continuing from the oldest batch in flight). */
int basal_pipe_stop(basal_pipe_t *p);
int basal_pipe_rewind(basal_pipe_t *p);
/* -B / -E (ReadClass::InitIndex, reads.cpp:22-40): the number the next read gets, and the number at which loading stops
 * (text form: records from read_end on are ignored). Only with nothing in flight. */
int basal_pipe_set_read_range(basal_pipe_t *p, uint32_t next_index, uint32_t read_end);

/* ---- paired-end pairing on the device ----
 * PairAlign::RunAlign's pairing rounds (SortHits4PE + GetPairs, pairs.cpp:29-177) and the choices of StringAlignPair / StringAlignUnpair
 * (pairs.cpp:204-305) run on the GPU over the mates' mode-tagged hit logs; what comes back is, per read pair, the list of records to
 * print, in order.  The host turns them into text (basal_host_format_pe_records = s_OutHitPair / s_OutHitUnpair, pairs.cpp:307-485). */
#define BASAL_PE_PAIR 0   /* one pair: both mates' lines (s_OutHitPair) */
#define BASAL_PE_UNPAIR 1 /* one line of one mate (s_OutHitUnpair) */
typedef struct basal_pe_rec {
    uint8_t kind;     /* BASAL_PE_* */
    uint8_t side;     /* UNPAIR: 0 = the line is mate 1's, 1 = mate 2's */
    uint8_t chain_a;  /* PAIR: pairhit.chain; UNPAIR: chain of the printed mate's hit */
    uint8_t chain_b;  /* UNPAIR: chain of the other mate's hit */
    int32_t ma;       /* PAIR: number of pairs at the best level (the n of s_OutHitPair); UNPAIR: hits of the printed mate (-1 failed QC, 0 none) */
    uint32_t na;      /* mismatch level of mate 1's hit (PAIR) / of the printed mate's hit (UNPAIR) */
    int32_t mb;       /* PAIR: mismatch level of mate 2's hit; UNPAIR: hits of the other mate as StringAlignUnpair passes them (<= 0: none to point at) */
    uint32_t insert;  /* PAIR: insert size */
    basal_hit ha, hb; /* PAIR: mate 1's and mate 2's hit; UNPAIR: the printed mate's hit and the other mate's */
} basal_pe_rec;
typedef struct basal_pe_pair {
    uint32_t first, n; /* this pair's records in the record array */
    uint32_t status;   /* BASAL_READ_OVERFLOW: the hit stream or the record array was too small */
} basal_pe_pair;
/* Align the 2 * npairs mates of a batch (interleaved a0, b0, a1, b1, ...; mates that both passed QC carry BASAL_READ_ALLMODES, as for
 * basal_host_format_pe) and pair them, all on the device; pairs_out[npairs], recs_out[recs_cap]. stats[9] is incremented: aligned /
 * unique / multiple for pairs, mate 1, mate 2. BASAL_EOVERFLOW: *recs_used says how many records are needed. */
int basal_core_align_pairs_batch(basal_core_t *c, const uint8_t *bases, uint64_t nbases, const basal_read *reads, uint32_t npairs, const basal_stale *stales,
                                 uint32_t nstale, basal_pe_pair *pairs_out, basal_pe_rec *recs_out, uint64_t recs_cap, uint64_t *recs_used, uint32_t stats[9],
                                 uint8_t carry[2][2]);

/* Instrumentation (no reference counterpart): one line per align-kernel instantiation in text[cap], "nwt newrule gap heavy pe assumed fit slack" --
 * the blocks per CU its launch bounds ask for, the blocks the runtime says fit on this GPU, the LDS bytes a block could still grow by. Returns the
 * number of lines. */
int basal_core_occupancy_report(char *text, size_t cap);

/* Placement (no reference counterpart): on a repeat-rich index the align kernel's time depends on where hipMalloc put the core's buffers in HBM -- steady
 * for a given placement, up to 10 % apart between placements. A host with a representative batch can time a launch on two
 * placements and keep the better one (bench.py does, in its set-up). Results do not depend on any of this; no launch of the core may be in flight.
 *   basal_core_placement_fork   every long-lived buffer (flank words, locations, seed words, k-mer tables, reference, hit logs) is copied into freshly
 *                               allocated memory; the core now runs on the copies, the originals are kept aside. Returns the number of buffers copied,
 *                               0 when HBM has no room for a second set (nothing changed), or a negative error.
 *   basal_core_placement_swap   back onto the set kept aside (which the current one then becomes).
 *   basal_core_placement_commit frees the set kept aside.
 *   basal_core_move_buffers     experiment hook: one class moved (allocation, copy, old one freed): 0 locations, 1 flank words, 2 seed words, 3 k-mer
 *                               tables, 4 reference, 5 hit logs; returns the buffers moved. */
int basal_core_placement_fork(basal_core_t *c);
int basal_core_placement_swap(basal_core_t *c);
int basal_core_placement_commit(basal_core_t *c);
int basal_core_move_buffers(basal_core_t *c, int which);

/* ---- several GPUs of one node: reads sharded by read number, hit records gathered with RCCL ----
 * The reference fans batches out to host threads (main.cpp:60-92); this fans the reads of a batch out to GPUs. Every GPU holds the whole
 * reference + index, aligns a contiguous range of the batch's reads, and ONE ncclGather per batch moves the 32-byte records to GPU 0. */
/* [begin, end) of the reads rank `rank` of `world` takes of a batch of n (contiguous, sizes differ by at most one). No GPU needed. */
void basal_shard_range(uint64_t n, uint32_t rank, uint32_t world, uint64_t *begin, uint64_t *end);
typedef struct basal_multi basal_multi_t;
typedef struct basal_ref basal_ref_t; /* (host-side reference object, see basal_host_ref_* below) */
int basal_multi_create(const basal_params *p, const int *devices, int ndev, basal_multi_t **out); /* one basal_core_t per device + the RCCL communicators */
void basal_multi_destroy(basal_multi_t *m);
int basal_multi_ndev(const basal_multi_t *m);
basal_core_t *basal_multi_core(basal_multi_t *m, int rank);
/* Instrumentation: the bytes the last basal_multi_align_batch copied to GPU `rank` (its shard's bases, descriptors and stale entries). */
uint64_t basal_multi_last_h2d_bytes(const basal_multi_t *m, int rank);
int basal_multi_upload(basal_multi_t *m, const basal_ref_t *r, int build_index_on_gpu, uint32_t *max_kmer_num); /* basal_host_ref_upload on every GPU */
/* basal_core_align_batch over all GPUs: same arguments, same results. */
int basal_multi_align_batch(basal_multi_t *m, const uint8_t *bases, uint64_t nbases, const basal_read *reads, uint32_t n, const basal_stale *stales,
                            uint32_t nstale, int stream_mode, basal_result *results, basal_hit *stream, uint64_t stream_cap, uint64_t *stream_used,
                            uint8_t carry[2][2]);

/* ---- host helpers ---- */
void basal_host_params_defaults(basal_params *p);                    /* Param::Param, param.cpp:7-68 */
int basal_host_params_set_seed_size(basal_params *p, int n);         /* Param::SetSeedSize, param.cpp:108-115 */
int basal_host_params_set_align(basal_params *p, const char *rule);  /* Param::SetAlign, param.cpp:163-263 */
void basal_host_params_set_v(basal_params *p, double v);             /* main.cpp:324-338 */

typedef struct basal_ref basal_ref_t; /* RefSeq without the index arrays' ownership rules */
/* RefSeq::Run_ConvertBinseq (refbase.cpp:186-252) from a FASTA file (plain or .gz) or memory. */
int basal_host_ref_load(const basal_params *p, const char *fasta_path, basal_ref_t **out);
int basal_host_ref_load_mem(const basal_params *p, const char *buf, size_t len, basal_ref_t **out);
void basal_host_ref_free(basal_ref_t *r);
/* views into the loaded reference (owned by r) */
uint32_t basal_host_ref_ncontig(const basal_ref_t *r);
const char *basal_host_ref_name(const basal_ref_t *r, uint32_t contig);
const uint32_t *basal_host_ref_sizes(const basal_ref_t *r);
const uint32_t *basal_host_ref_rc_offsets(const basal_ref_t *r);
const uint32_t *basal_host_ref_anchors(const basal_ref_t *r);
uint64_t basal_host_ref_nwords(const basal_ref_t *r);
const uint64_t *basal_host_ref_words(const basal_ref_t *r, int strand);
uint64_t basal_host_ref_nblocks(const basal_ref_t *r);
const uint32_t *basal_host_ref_blocks(const basal_ref_t *r); /* (id,begin,end) triples, sorted */
/* RefSeq::CreateIndex on the CPU (refbase.cpp:261-439): fills the flattened index inside r. */
int basal_host_ref_build_index(basal_ref_t *r, const basal_params *p, int threads);
uint32_t basal_host_ref_total_kmers(const basal_ref_t *r);
const uint32_t *basal_host_ref_kmer_off(const basal_ref_t *r);
const uint32_t *basal_host_ref_kmer_nfwd(const basal_ref_t *r);
const uint32_t *basal_host_ref_locs(const basal_ref_t *r);
uint64_t basal_host_ref_nlocs(const basal_ref_t *r);
uint32_t basal_host_ref_max_kmer_num(const basal_ref_t *r);
/* convenience: set_reference + (set_index | build_index) */
int basal_host_ref_upload(const basal_ref_t *r, basal_core_t *c, int build_index_on_gpu, uint32_t *max_kmer_num);

/* Tracks, per SingleAlign object (slot 0: SE reads and mate 1; slot 1: mate 2), what later reads can
 * inherit (see basal_stale). Call begin_batch at the start of every batch, then visit every read of
 * the batch in input order (QC-failed ones too, with qc_failed=1). visit returns 1 and fills *out
 * when the read needs a basal_stale entry (its descriptor's stale_idx must then point at it), else 0. */
typedef struct basal_stale_tracker basal_stale_tracker_t;
basal_stale_tracker_t *basal_host_stale_new(const basal_params *p);
void basal_host_stale_free(basal_stale_tracker_t *t);
void basal_host_stale_begin_batch(basal_stale_tracker_t *t);
int basal_host_stale_visit(basal_stale_tracker_t *t, const char *seq, uint32_t len, uint32_t readset, int qc_failed,
                           uint32_t read_number_in_batch, basal_stale *out);

/* FilterReads (align.cpp:548-563): trims seq/qual in place (NUL-terminated, caller-owned,
 * qual buffer at least as long as seq), returns 1 if the read fails QC, else 0 with
 * *read_max_snp_num set. */
int basal_host_filter_read(const basal_params *p, char *seq, char *qual, uint32_t *read_max_snp_num);

/* SAM text for one SE read: StringAlign + s_OutHit (align.cpp:583-669). status: 0 aligned
 * normally, 1 QC-failed. Appends to a caller buffer; returns bytes written or <0 (BASAL_EOVERFLOW
 * if cap is too small; nothing is written then). */
int64_t basal_host_format_se(const basal_params *p, const basal_ref_t *r, const char *name, const char *seq,
                             const char *qual, uint32_t readset, int qc_failed, const basal_result *res,
                             const basal_hit *stream, char *out, size_t cap);
/* One read pair -> SAM: PairAlign::RunAlign's pairing rounds (SortHits4PE + GetPairs, pairs.cpp:29-177) replayed over
 * the two mates' mode-tagged hit logs, then StringAlignPair / StringAlignUnpair / s_OutHitPair / s_OutHitUnpair
 * (pairs.cpp:204-485). Mates that both passed QC must have been aligned with BASAL_READ_ALLMODES and
 * BASAL_STREAM_ALL; if one mate failed QC the other is aligned like an SE read (pairs.cpp:191-196).
 * Names must already be fixed (FixPairReadName, pairs.cpp:487-507: basal_host_fix_pair_names). stats[9] is
 * incremented: aligned/unique/multiple for pairs, mate 1, mate 2. */
typedef struct basal_mate {
    const char *name, *seq, *qual; /* as FilterReads left them */
    uint32_t readset, index, max_snp;
    int qc_failed;
    const basal_result *res;
} basal_mate;
int64_t basal_host_format_pe(const basal_params *p, const basal_ref_t *r, const basal_mate *a, const basal_mate *b,
                             const basal_hit *stream, char *out, size_t cap, uint32_t stats[9]);
/* The text of one pair from the device's records (basal_core_align_pairs_batch): only name / seq / qual / readset of the mates are read. */
int64_t basal_host_format_pe_records(const basal_params *p, const basal_ref_t *r, const basal_mate *a, const basal_mate *b, const basal_pe_rec *recs,
                                     uint32_t n, char *out, size_t cap);
/* FixPairReadName: truncates both names in place to their common prefix up to its last digit; -1 if they share nothing */
int basal_host_fix_pair_names(char *name_a, char *name_b);

int64_t basal_host_sam_header(const basal_ref_t *r, const char *cmdline, char *out, size_t cap); /* main.cpp:586-597 */

#ifdef __cplusplus
}
#endif
#endif
