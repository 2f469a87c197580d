/*
 * oracle_main.c -- command-line driver for the CPU oracle (test infrastructure only).
 * Mirrors the reference CLI (main.cpp:272-364 flags, 56-92 batch threads, 555-613 SE driver)
 * closely enough to produce the same SAM from the same command line; used by the golden-SAM
 * tests and as bench.py's cpu_baseline "port" leg.  Prints "ORACLE_ALIGN_SECONDS <s>" (align
 * phase only, monotonic clock) and the §8d counters to stderr.
 */
#define _GNU_SOURCE
#include "basal_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <zlib.h>

#define BATCHNUM 50000 /* reads.h:14 */

typedef struct reader {
    gzFile f;
    int fastq;
    uint32_t index;
    char *buf; size_t len, pos; /* whole file in memory */
} reader;

static int is_ws(int c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }
static void skip_ws(reader *r) { while (r->pos < r->len && is_ws((unsigned char)r->buf[r->pos])) r->pos++; }
static char *token(reader *r, size_t *n) { skip_ws(r); size_t s = r->pos; while (r->pos < r->len && !is_ws((unsigned char)r->buf[r->pos])) r->pos++; *n = r->pos - s; return r->buf + s; }
static void rest_of_line(reader *r) { while (r->pos < r->len && r->buf[r->pos] != '\n') r->pos++; if (r->pos < r->len) r->pos++; }

static int reader_open(reader *r, const char *path) {
    gzFile f = gzopen(path, "rb");
    if (!f) return -1;
    size_t cap = 1 << 22, n = 0; char *b = malloc(cap);
    for (;;) { if (cap - n < (1 << 20)) { cap *= 2; b = realloc(b, cap); } int g = gzread(f, b + n, (unsigned)((cap - n) > (1u << 30) ? (1u << 30) : (cap - n))); if (g <= 0) break; n += (size_t)g; }
    gzclose(f);
    r->buf = b; r->len = n; r->pos = 0; r->index = 0;
    size_t p = 0; while (p < n && is_ws((unsigned char)b[p])) p++;
    r->fastq = (p < n && b[p] == '@');
    return 0;
}

typedef struct rec { uint32_t index, readset; char *name, *seq, *qual; } rec;

/* reads.cpp:42-83 (FASTA/FASTQ branch).  Strings are copied with room for in-place trimming. */
static int load_batch(reader *r, const orc_param *p, rec *out, int readset) {
    int num = 0;
    for (; num < BATCHNUM && r->index < p->read_end; num++, r->index++) {
        skip_ws(r);
        if (r->pos >= r->len) break;
        r->pos++; /* '>' or '@' */
        size_t nl, sl, ql = 0; char *nm = token(r, &nl); rest_of_line(r);
        char *sq = token(r, &sl); char *ql_p = NULL;
        if (r->fastq) { size_t tl; token(r, &tl); rest_of_line(r); ql_p = token(r, &ql); }
        rec *o = &out[num];
        o->index = r->index; o->readset = (uint32_t)readset;
        o->name = strndup(nm, nl);
        size_t cap = (sl > ql ? sl : ql) + 2;
        o->seq = malloc(cap); memcpy(o->seq, sq, sl); o->seq[sl] = 0;
        o->qual = malloc(cap);
        if (r->fastq) { memcpy(o->qual, ql_p, ql); o->qual[ql] = 0; }
        else { memset(o->qual, p->zero_qual + p->default_qual, sl); o->qual[sl] = 0; }
        if (sl > p->max_readlen) { o->seq[p->max_readlen] = 0; if (strlen(o->qual) > p->max_readlen) o->qual[p->max_readlen] = 0; }
    }
    return num;
}

static orc_param P;
static orc_ref *R;
static reader RA, RB;
static FILE *fout;
static pthread_mutex_t mfin = PTHREAD_MUTEX_INITIALIZER, mfout = PTHREAD_MUTEX_INITIALIZER;
static uint32_t g_aligned, g_unique, g_multiple;
static orc_counters g_c;
static uint32_t g_pe[9];

static void add_counters(const orc_counters *c) {
    g_c.reads += c->reads; g_c.hdr_lookups += c->hdr_lookups; g_c.seed_lookups += c->seed_lookups; g_c.candidates += c->candidates;
    g_c.ref_words += c->ref_words; g_c.read_bytes += c->read_bytes; g_c.hit_records += c->hit_records; g_c.snp_calls += c->snp_calls; g_c.gap_calls += c->gap_calls;
}

static void *t_single(void *arg) {
    (void)arg;
    orc_aligner *a = orc_aligner_new(&P, R);
    rec *batch = calloc(BATCHNUM, sizeof(rec));
    orc_str os = {0};
    for (;;) {
        pthread_mutex_lock(&mfin);
        int n = load_batch(&RA, &P, batch, 0);
        pthread_mutex_unlock(&mfin);
        if (!n) break;
        os.n = 0;
        for (int i = 0; i < n; i++) {
            orc_read rd = {batch[i].index, batch[i].readset, batch[i].name, batch[i].seq, batch[i].qual};
            orc_do_read(a, &rd, &os);
            free(batch[i].name); free(batch[i].seq); free(batch[i].qual);
        }
        pthread_mutex_lock(&mfout);
        if (os.n) fwrite(os.s, 1, os.n, fout);
        pthread_mutex_unlock(&mfout);
    }
    pthread_mutex_lock(&mfout);
    uint32_t x, y, z; orc_aligner_stats(a, &x, &y, &z);
    g_aligned += x; g_unique += y; g_multiple += z;
    add_counters(orc_aligner_counters(a));
    pthread_mutex_unlock(&mfout);
    orc_str_free(&os); free(batch); orc_aligner_free(a);
    return NULL;
}

static void *t_pair(void *arg) {
    (void)arg;
    orc_pair_aligner *pa = orc_pair_aligner_new(&P, R);
    rec *b1 = calloc(BATCHNUM, sizeof(rec)), *b2 = calloc(BATCHNUM, sizeof(rec));
    orc_str os = {0};
    for (;;) {
        pthread_mutex_lock(&mfin);
        int n1 = load_batch(&RA, &P, b1, 1);
        int n2 = load_batch(&RB, &P, b2, 2);
        pthread_mutex_unlock(&mfin);
        if (!n1 || n1 != n2) break;
        os.n = 0;
        for (int i = 0; i < n1; i++) {
            orc_read ra = {b1[i].index, b1[i].readset, b1[i].name, b1[i].seq, b1[i].qual};
            orc_read rb = {b2[i].index, b2[i].readset, b2[i].name, b2[i].seq, b2[i].qual};
            orc_do_pair(pa, &ra, &rb, &os);
            free(b1[i].name); free(b1[i].seq); free(b1[i].qual);
            free(b2[i].name); free(b2[i].seq); free(b2[i].qual);
        }
        pthread_mutex_lock(&mfout);
        if (os.n) fwrite(os.s, 1, os.n, fout);
        pthread_mutex_unlock(&mfout);
    }
    pthread_mutex_lock(&mfout);
    uint32_t st[9]; orc_pair_stats(pa, st);
    for (int i = 0; i < 9; i++) g_pe[i] += st[i];
    add_counters(orc_pair_counters(pa, 0)); add_counters(orc_pair_counters(pa, 1));
    pthread_mutex_unlock(&mfout);
    orc_str_free(&os); free(b1); free(b2); orc_pair_aligner_free(pa);
    return NULL;
}

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv) {
    const char *qa = NULL, *qb = NULL, *ref = NULL, *out = NULL, *rule = NULL;
    char cmdline[8192] = {0};
    orc_param_defaults(&P);
    strncat(cmdline, argv[0], sizeof cmdline - 1);
    for (int i = 1; i < argc; i++) { strncat(cmdline, " ", sizeof cmdline - strlen(cmdline) - 1); strncat(cmdline, argv[i], sizeof cmdline - strlen(cmdline) - 1); }
    /* main.cpp:272-364: "-x v" or "-x=v" */
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (a[0] != '-') { fprintf(stderr, "unknown option: %s\n", a); return i; }
        const char *v = NULL;
        char f = a[1];
        if (strchr("RHu3N", f)) { if (a[2]) { fprintf(stderr, "unknown option: %s\n", a); return i; } }
        else if (a[2] == 0) { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a); return i; } v = argv[++i]; }
        else if (a[2] == '=') v = a + 3;
        else { fprintf(stderr, "unknown option: %s\n", a); return i; }
        switch (f) {
        case 'a': qa = v; break;
        case 'b': qb = v; P.pairend = 1; break;
        case 'd': ref = v; break;
        case 's': { int n = atoi(v); if (n > 16 || n < 10) { fprintf(stderr, "seed size must be between 10 and 16\n"); return 1; } orc_param_set_seed_size(&P, n); break; }
        case 'o': out = v; break;
        case 'M': rule = v; break;
        case 'm': P.min_insert = (uint32_t)atoi(v); break;
        case 'n': P.chains = (uint32_t)atoi(v); break;
        case 'g': P.gap = (uint32_t)atoi(v); if (P.gap > ORC_MAXGAPS) P.gap = ORC_MAXGAPS; break;
        case 'x': P.max_insert = (uint32_t)atoi(v); break;
        case 'r': P.report_repeat_hits = (uint32_t)atoi(v); if (P.report_repeat_hits > 2) { fprintf(stderr, "invalid -r value\n"); return 1; } break;
        case 'V': break;
        case 'I': P.index_interval = (uint32_t)atoi(v); if (P.index_interval > 16) { fprintf(stderr, "index interval exceeds max value:16\n"); return 1; } break;
        case 'k': P.max_kmer_ratio = (float)atof(v); break;
        case 'v': orc_param_set_v(&P, atof(v)); break;
        case 'w': P.max_num_hits = (uint32_t)atoi(v); if (P.max_num_hits > ORC_MAXHITS) { fprintf(stderr, "number of multi-hits exceeds max value\n"); return 1; } break;
        case 'q': P.trim_qual_threshold = (uint32_t)atoi(v); break;
        case 'f': P.max_ns = (uint32_t)atoi(v); break;
        case 'z': P.zero_qual = (uint8_t)atoi(v); break;
        case 'p': P.num_procs = (uint32_t)atoi(v); break;
        case 'A': if (P.n_adapter < 10) { strncpy(P.adapter[P.n_adapter], v, 127); P.n_adapter++; } break;
        case 'R': P.out_ref = 1; break;
        case 'H': P.sam_header = 0; break;
        case 'u': P.out_unmap = 1; break;
        case 'B': { int b = atoi(v); P.read_start = (uint32_t)(b > 1 ? b : 1); break; }
        case 'E': P.read_end = (uint32_t)atoi(v); break;
        case 'L': P.max_readlen = (uint32_t)atoi(v); break;
        case 'N': P.N_mis = 1; break;
        case 'S': P.randseed = (uint32_t)atoi(v); break;
        default: fprintf(stderr, "unknown option: %s\n", a); return i;
        }
    }
    orc_param_init_mapping(&P);
    char err[256];
    if (!rule) { fprintf(stderr, "\n-M option is required\n"); return 1; }
    if (orc_param_set_align(&P, rule, err, sizeof err)) { fprintf(stderr, "%s\n", err); return 1; }
    if (!ref || !qa) { fprintf(stderr, "-a and -d are required\n"); return 1; }
    if (P.randseed == 0) fprintf(stderr, "warning: -S 0 is not reproducible in the reference; the oracle treats it as the hash RNG with seed 0\n");

    double t0 = now();
    R = orc_ref_load_fasta(ref, &P);
    if (!R) { fprintf(stderr, "failed to open reference file (check -d option): %s\n", ref); return 1; }
    double t1 = now();
    orc_ref_build_index(R, &P);
    double t2 = now();
    fprintf(stderr, "ORACLE_REF_SECONDS %.6f\nORACLE_INDEX_SECONDS %.6f\nORACLE_MAX_KMER_NUM %u\n", t1 - t0, t2 - t1, P.max_kmer_num);

    if (reader_open(&RA, qa)) { fprintf(stderr, "failed to open read file (check -a option): %s\n", qa); return 1; }
    if (qb && reader_open(&RB, qb)) { fprintf(stderr, "failed to open read file #2 (check -b option): %s\n", qb); return 1; }
    /* InitIndex reads.cpp:13-40: skip read_start-1 records */
    for (reader *r = &RA; r; r = (r == &RA && qb) ? &RB : NULL) {
        uint32_t maxi = (P.read_start - 1) * (2 + 2 * (uint32_t)r->fastq);
        for (uint32_t i = 0; i < maxi && r->pos < r->len; i++) rest_of_line(r);
        r->index = P.read_start - 1;
    }
    fout = out ? fopen(out, "w") : stdout;
    if (!fout) { fprintf(stderr, "failed to open output file (check -o option): %s\n", out); return 1; }
    if (P.sam_header) { orc_str h = {0}; orc_sam_header(R, cmdline, &h); fwrite(h.s, 1, h.n, fout); orc_str_free(&h); }

    double t3 = now();
    int np = (int)P.num_procs; if (np < 1) np = 1;
    pthread_t *th = calloc((size_t)np, sizeof(pthread_t));
    for (int i = 0; i < np; i++) pthread_create(&th[i], NULL, P.pairend ? t_pair : t_single, NULL);
    for (int i = 0; i < np; i++) pthread_join(th[i], NULL);
    double t4 = now();
    if (out) fclose(fout);
    uint32_t total = RA.index - P.read_start + 1;
    fprintf(stderr, "ORACLE_ALIGN_SECONDS %.6f\nORACLE_READS %u\nORACLE_THREADS %d\n", t4 - t3, total, np);
    if (P.pairend) fprintf(stderr, "ORACLE_PE_STATS %u %u %u %u %u %u %u %u %u\n", g_pe[0], g_pe[1], g_pe[2], g_pe[3], g_pe[4], g_pe[5], g_pe[6], g_pe[7], g_pe[8]);
    else fprintf(stderr, "ORACLE_STATS aligned %u unique %u multiple %u\n", g_aligned, g_unique, g_multiple);
    fprintf(stderr, "ORACLE_COUNTERS reads %llu H %llu S %llu C %llu W %llu L %llu R %llu snp_calls %llu gap_calls %llu\n",
            (unsigned long long)g_c.reads, (unsigned long long)g_c.hdr_lookups, (unsigned long long)g_c.seed_lookups, (unsigned long long)g_c.candidates,
            (unsigned long long)g_c.ref_words, (unsigned long long)g_c.read_bytes, (unsigned long long)g_c.hit_records, (unsigned long long)g_c.snp_calls, (unsigned long long)g_c.gap_calls);
    orc_ref_free(R);
    return 0;
}
