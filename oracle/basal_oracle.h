/*
 * basal_oracle.h -- CPU ORACLE for the BASAL seed-and-extend hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference algorithm
 * (JiejunShi/BASAL, files cited per function as file:line under /root/reference).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as
 * the checker.  The product path (basal_amd/csrc, include/basal_core.h) never links, loads
 * or calls anything in oracle/.
 *
 * Parity status: PINNED.  The reference ships no tests of its own, so the
 * oracle is pinned against outputs of the reference itself: oracle/Makefile.ref builds the
 * unmodified reference from /root/reference into oracle/_ref/basal, tools/make_golden.py runs
 * it on seeded synthetic inputs, and tests/test_oracle_golden.py requires this restatement to
 * reproduce every committed golden SAM under tests/golden/ byte for byte (minus the @PG line).
 */
#ifndef BASAL_ORACLE_H
#define BASAL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_SEGLEN 32      /* param.h:4 */
#define ORC_FIXELEMENT 16  /* param.h:17 */
#define ORC_MAXSNPS 15     /* param.h:18 */
#define ORC_MAXGAPS 3      /* param.h:19 */
#define ORC_MAXHITS 1000   /* makefile:4 -DMAXHITS=1000 */
#define ORC_REF_MARGIN 400 /* refbase.h:16 */
#define ORC_BINSEQPAD 2    /* refbase.h:17 */
#define ORC_FIXSIZE (ORC_SEGLEN * ORC_FIXELEMENT)

/* ---- bit primitives (param.h:104-142, utilities.cpp:38-48) ---- */
uint32_t orc_XT(uint32_t tt);
uint64_t orc_XT64(uint64_t tt);
uint64_t orc_XC64(uint64_t tt);
uint32_t orc_XM64(uint64_t tt);
uint64_t orc_M2_judge(uint64_t tt);
uint32_t orc_myrand(int i, uint32_t randseed);

/* ---- parameters (param.h:44-148, param.cpp:7-115,163-263, main.cpp:272-364) ---- */
typedef struct orc_param {
    uint32_t seed_size, seed_bits, seed_bits_lz, index_interval;
    uint32_t max_snp_num, max_num_hits, chains, randseed, gap, gap_edge;
    uint32_t max_ns, min_read_size, n_adapter, trim_qual_threshold;
    uint32_t out_ref, out_unmap, report_repeat_hits, sam_header, max_readlen;
    uint32_t pairend, min_insert, max_insert, N_mis, read_start, read_end, num_procs;
    uint8_t zero_qual, default_qual;
    float max_kmer_ratio;
    char adapter[10][128];
    /* -M */
    char refnt;
    int readnt_cnt;
    char readnts[5];
    int new_rule; /* !(readnt_cnt==1 && readnts[0]!='-')  align.cpp:300 */
    uint8_t alphabet[256], rev_alphabet[256], reg_alphabet[256];
    uint8_t alphabet_mread[256], rev_alphabet_mread[256];
    char useful_nt[9];
    uint32_t profile[ORC_MAXSNPS + 1][16];
    uint32_t max_kmer_num; /* filled by index build, refbase.cpp:363 */
    uint32_t total_ref_seq;
} orc_param;

void orc_param_defaults(orc_param *p);
void orc_param_set_seed_size(orc_param *p, int n);
/* returns 0, or -1 with a message in err (the reference exits instead) */
int orc_param_set_align(orc_param *p, const char *rule, char *err, size_t errlen);
void orc_param_init_mapping(orc_param *p);
/* -v parsing, main.cpp:324-338 */
void orc_param_set_v(orc_param *p, double v);

/* ---- reference + index (refbase.cpp) ---- */
typedef struct orc_block {
    uint32_t id, begin, end;
} orc_block;

typedef struct orc_ref {
    uint32_t ncontig;
    char **name;
    uint32_t *size;      /* title[2i].size */
    uint32_t *rc_offset; /* title[2i].rc_offset */
    uint32_t *nword;     /* bfa[2i].n */
    uint64_t sum_length;
    uint64_t nwords_total; /* s + 2*REF_MARGIN */
    uint64_t *xref[2];
    uint32_t *ref_anchor; /* ncontig+1 */
    orc_block *blocks;
    size_t nblocks;
    /* index (KmerLoc2 index2[] flattened): per k-mer n_tot=n[0], n_fwd=n[1], off into locs */
    uint32_t total_kmers;
    uint32_t *n_tot, *n_fwd;
    uint64_t *off;
    uint32_t *locs;
    uint64_t nlocs;
    int owns_arrays;
} orc_ref;

orc_ref *orc_ref_load_fasta_mem(const char *buf, size_t len, orc_param *p);
orc_ref *orc_ref_load_fasta(const char *path, orc_param *p); /* plain or .gz */
void orc_ref_build_index(orc_ref *r, orc_param *p);          /* sets p->max_kmer_num */
/* adopt arrays built elsewhere (used by bench.py to time the oracle on a GPU-built index) */
orc_ref *orc_ref_from_arrays(uint32_t ncontig, const char *const *names, const uint32_t *size,
                             uint64_t *xref_fwd, uint64_t *xref_rc, uint64_t nwords_total,
                             uint32_t total_kmers, uint32_t *n_tot, uint32_t *n_fwd, uint64_t *off,
                             uint32_t *locs, uint64_t nlocs);
void orc_ref_free(orc_ref *r);

/* ---- hits (param.h:35-42 gHit; fields widened, same value ranges) ---- */
typedef struct orc_hit {
    uint32_t loc;
    uint32_t chr;    /* 18-bit field in the reference */
    uint32_t strand; /* 2 bits: (ref_chain<<1)|read_chain */
    int32_t gap_size;
    uint32_t gap_pos; /* 9-bit field */
} orc_hit;

/* one record of the per-read hit log, in the order AddHit stored them */
typedef struct orc_loghit {
    orc_hit h;
    uint8_t level, chain, mode, pad;
} orc_loghit;

typedef struct orc_counters { /* SURVEY.md §8d algorithmic-bytes counters */
    uint64_t reads, hdr_lookups /*H*/, seed_lookups /*S*/, candidates /*C*/, ref_words /*W*/,
        read_bytes /*L*/, hit_records /*R*/, snp_calls, gap_calls;
} orc_counters;

typedef struct orc_aligner orc_aligner;
orc_aligner *orc_aligner_new(const orc_param *p, const orc_ref *r);
void orc_aligner_free(orc_aligner *a);
const orc_counters *orc_aligner_counters(const orc_aligner *a);
void orc_aligner_stats(const orc_aligner *a, uint32_t *n_aligned, uint32_t *n_unique, uint32_t *n_multiple);

/* one read in, as reads.cpp leaves it (seq/qual are modified in place by trimming) */
typedef struct orc_read {
    uint32_t index, readset;
    char *name, *seq, *qual;
} orc_read;

/* FilterReads (align.cpp:548-563): returns 1 if QC-failed. */
int orc_filter_read(orc_aligner *a, orc_read *rd);
/* RunAlign (align.cpp:446-466) on a filtered read. Returns 1 if any hit. */
int orc_run_align(orc_aligner *a, const orc_read *rd);
/* introspection after orc_run_align */
uint32_t orc_read_max_snp(const orc_aligner *a);
uint32_t orc_n_hit(const orc_aligner *a, int chain, int level);
const orc_hit *orc_hits(const orc_aligner *a, int chain, int level);
size_t orc_hit_log(const orc_aligner *a, const orc_loghit **log);
void orc_seed_state(const orc_aligner *a, uint32_t start_off[2], uint32_t start_arr[2][16],
                    int32_t seg_weight[2][16], int32_t seg_order[2][16], uint32_t *seedseg_num);
const uint64_t *orc_xseq(const orc_aligner *a, int chain); /* 48 words */
const uint32_t *orc_seed_array(const orc_aligner *a, int chain);

/* growable text buffer for SAM */
typedef struct orc_str {
    char *s;
    size_t n, cap;
} orc_str;
void orc_str_free(orc_str *s);

/* StringAlign + s_OutHit (align.cpp:583-669) for the read last aligned */
void orc_string_align(orc_aligner *a, const orc_read *rd, orc_str *os);
/* Do_Batch body for one read: filter, align, format (align.cpp:565-580) */
void orc_do_read(orc_aligner *a, orc_read *rd, orc_str *os);
/* SAM header (main.cpp:586-597); cmdline goes into @PG CL */
void orc_sam_header(const orc_ref *r, const char *cmdline, orc_str *os);

/* PE (pairs.cpp) */
typedef struct orc_pair_aligner orc_pair_aligner;
orc_pair_aligner *orc_pair_aligner_new(const orc_param *p, const orc_ref *r);
void orc_pair_aligner_free(orc_pair_aligner *pa);
void orc_do_pair(orc_pair_aligner *pa, orc_read *ra, orc_read *rb, orc_str *os);
void orc_pair_stats(const orc_pair_aligner *pa, uint32_t out[9]);
const orc_counters *orc_pair_counters(orc_pair_aligner *pa, int mate);

/* Batch driver used by bench.py's cpu_baseline leg and by at-scale spot checks: aligns n SE reads
 * that are already in memory with `threads` pthreads (one orc_aligner each, contiguous slices) and
 * returns, per read, what StringAlign would pick.  out[i] = {best_level (0xFF none), n_hit, n_chit,
 * chr, loc, gap_size, gap_pos, chain}.  Fills *seconds (wall, monotonic) and sums the counters. */
typedef struct orc_best {
    uint32_t best_level, n_hit, n_chit, chr, loc;
    int32_t gap_size;
    uint32_t gap_pos, chain;
} orc_best;
int orc_align_batch_mt(const orc_param *p, const orc_ref *r, const uint8_t *bases, const uint32_t *seq_off, const uint16_t *len,
                       const uint32_t *index, const uint8_t *max_snp, uint32_t n, int threads, orc_best *out, orc_counters *counters,
                       double *seconds);
#ifdef __cplusplus
}
#endif
#endif
