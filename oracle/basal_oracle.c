/*
 * basal_oracle.c -- CPU ORACLE (test infrastructure, NOT product code; see basal_oracle.h).
 *
 * Plain-C restatement of the BASAL seed-and-extend path.  Every function cites the
 * reference file:line (under /root/reference) whose behaviour it restates.  Written to be
 * literal and easy to audit, not fast.  Pinned against the real reference's output by
 * tests/test_oracle_golden.py (golden SAMs made by tools/make_golden.py).
 */
#define _GNU_SOURCE
#include "basal_oracle.h"

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ------------------------------------------------------------------ bit primitives */

/* param.h:104 */
uint64_t orc_XT64(uint64_t tt) {
    tt -= (tt << 1) & tt & 0xAAAAAAAAAAAAAAAAULL;
    return tt;
}

/* param.h:107-116: 16 2-bit digits, 11->01, read as a base-3 number */
uint32_t orc_XT(uint32_t tt) {
    uint32_t ss;
    tt -= (tt << 1) & tt & 0xAAAAAAAAUL;
    tt -= (tt >> 2) & 0x33333333UL;
    ss = (tt & 0xF0F0F0F0UL) >> 1;
    tt -= ss - (ss >> 3);
    ss = (tt & 0xFF00FF00UL) >> 2;
    tt = (tt & 0x00FF00FFUL) + ss + (ss >> 2) + (ss >> 6);
    return (tt & 0xFFFFUL) + (tt >> 16) * 6561;
}

/* param.h:119 */
uint64_t orc_XC64(uint64_t tt) { return ((~tt) << 1) | tt | 0x5555555555555555ULL; }

/* param.h:129-139: number of non-zero 2-bit pairs */
uint32_t orc_XM64(uint64_t tt) {
    tt |= tt >> 1;
    tt &= 0x5555555555555555ULL;
    tt += tt >> 2;
    tt &= 0x3333333333333333ULL;
    tt += tt >> 4;
    tt &= 0x0F0F0F0F0F0F0F0FULL;
    tt *= 0x0101010101010101ULL;
    tt >>= 56;
    return (uint32_t)tt;
}

/* param.h:142: 01 -> 00, 11 unchanged */
uint64_t orc_M2_judge(uint64_t tt) {
    tt &= ((tt & 0xAAAAAAAAAAAAAAAAULL) >> 1) | ((tt & 0x5555555555555555ULL) << 1);
    return tt;
}

/* utilities.cpp:38-48, the -S != 0 branch (the -S 0 branch is rand_r, non-reproducible) */
uint32_t orc_myrand(int i, uint32_t randseed) {
    uint64_t v;
    uint32_t s = randseed * 1000000u; /* bit32_t arithmetic, wraps */
    v = ((uint64_t)(int64_t)i + s) * 3935559000370003845ULL + 2691343689449507681ULL;
    v ^= v >> 21;
    v ^= v << 37;
    v ^= v >> 4;
    v *= 4768777513237032717ULL;
    v ^= v << 20;
    v ^= v >> 41;
    v ^= v << 5;
    return (uint32_t)(v & 0xffffffffUL);
}

/* ------------------------------------------------------------------ parameters */

static const char nt_code[5] = {'A', 'C', 'G', 'T', '-'};
static const char revnt_code[5] = {'T', 'G', 'C', 'A', '-'};

static int alphabet0(int c) { /* param.cpp:119-128 */
    switch (c) {
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 0;
    }
}

/* param.cpp:108-115 */
void orc_param_set_seed_size(orc_param *p, int n) {
    p->seed_size = (uint32_t)n;
    p->seed_bits_lz = (ORC_SEGLEN - p->seed_size) * 2;
    p->min_read_size = p->seed_size + p->index_interval - 1;
    p->seed_bits = 0;
    for (uint32_t i = 0; i < p->seed_size; i++) p->seed_bits |= 0x3u << (i * 2);
}

/* param.cpp:7-68 */
void orc_param_defaults(orc_param *p) {
    memset(p, 0, sizeof(*p));
    p->num_procs = 1;
    p->max_ns = 5;
    p->zero_qual = '!';
    p->trim_qual_threshold = 0;
    p->default_qual = 40;
    p->min_insert = 28;
    p->max_insert = 1000;
    p->index_interval = 4;
    orc_param_set_seed_size(p, 16);
    p->max_snp_num = 110;
    p->max_num_hits = ORC_MAXHITS > 100 ? 100 : ORC_MAXHITS;
    p->max_kmer_ratio = 5e-7f;
    p->min_read_size = p->seed_size; /* param.cpp:34 overrides SetSeedSize's value */
    p->report_repeat_hits = 1;
    memcpy(p->useful_nt, "ACGTacgt", 9);
    p->read_start = 1;
    p->read_end = ~0u;
    p->gap_edge = 6;
    p->max_readlen = (ORC_FIXELEMENT - 1) * ORC_SEGLEN;
    p->refnt = 'C';
    p->sam_header = 1;
    for (int c = 0; c < 256; c++) p->reg_alphabet[c] = 0; /* param.cpp:130-139 */
    for (const char *q = "ACGTacgt"; *q; q++) p->reg_alphabet[(unsigned char)*q] = 3;
}

/* param.cpp:70-74 */
void orc_param_init_mapping(orc_param *p) {
    for (uint32_t i = 0; i < p->index_interval; i++)
        for (uint32_t j = 0; j <= ORC_MAXSNPS; j++)
            p->profile[j][i] = ((j * p->seed_size + i + p->index_interval - 1) / p->index_interval) * p->index_interval;
}

/* main.cpp:324-338 */
void orc_param_set_v(orc_param *p, double v) {
    if (v < 1.0) {
        p->max_snp_num = (uint32_t)((int)(v * 100 + 0.5) + 100);
        if (p->max_snp_num == 100) p->max_snp_num = 0;
    } else {
        p->max_snp_num = (uint32_t)(int)(v + 0.5);
        if (p->max_snp_num > ORC_MAXSNPS) p->max_snp_num = ORC_MAXSNPS;
    }
}

/* param.cpp:163-263 */
int orc_param_set_align(orc_param *p, const char *rule, char *err, size_t errlen) {
    size_t rl = strlen(rule);
    if (rl < 2 || rule[1] != ':') {
        snprintf(err, errlen, "invalid -M, ref base(one letter in A/C/G/T) should be assigned first before :");
        return -1;
    }
    for (int i = 0; i < 5; i++) p->readnts[i] = ' ';
    p->refnt = (char)toupper((unsigned char)rule[0]);
    if (!p->reg_alphabet[(unsigned char)p->refnt]) {
        snprintf(err, errlen, "invalid -M, ref base %c not in A/C/G/T", rule[0]);
        return -1;
    }
    p->readnt_cnt = 0;
    for (size_t i = 2; i < rl; i++) {
        char readnt = (char)toupper((unsigned char)rule[i]);
        int valid = 0, used = 0;
        for (int j = 0; j < 5; j++) {
            if (nt_code[j] == readnt) valid = 1;
            if (p->readnts[j] == readnt) used = 1;
        }
        if (readnt == p->refnt) {
            snprintf(err, errlen, "invalid -M, read base %c should not be equal to ref base %c", rule[i], p->refnt);
            return -1;
        }
        if (!valid) {
            snprintf(err, errlen, "invalid -M, read base %c not in A/C/G/T/-", rule[i]);
            return -1;
        }
        if (!used && p->readnt_cnt < 5) p->readnts[p->readnt_cnt++] = readnt;
    }
    /* 202-215: Mread planes: 01 for any convert-to base, 11 for other ACGT, 00 else */
    for (int i = 0; i < 256; i++) {
        p->alphabet_mread[i] = p->reg_alphabet[i];
        p->rev_alphabet_mread[i] = p->reg_alphabet[i];
    }
    for (int i = 0; i < p->readnt_cnt; i++) {
        p->alphabet_mread[(unsigned char)p->readnts[i]] = 1;
        p->alphabet_mread[(unsigned char)tolower((unsigned char)p->readnts[i])] = 1;
        for (int j = 0; j < 5; j++) {
            if (nt_code[j] == p->readnts[i] && p->readnts[i] != '-') {
                p->rev_alphabet_mread[(unsigned char)revnt_code[j]] = 1;
                p->rev_alphabet_mread[(unsigned char)tolower((unsigned char)revnt_code[j])] = 1;
            }
        }
    }
    /* 216-233: 2-bit codes */
    uint8_t bit_nt[4] = {100, 100, 100, 100};
    bit_nt[alphabet0(p->refnt)] = 1;
    static const int other_bit[3] = {0, 2, 3}; /* the inner `int other_bit[2]` at 225 is dead */
    if (p->readnt_cnt == 1 && p->readnts[0] != '-') bit_nt[alphabet0(p->readnts[0])] = 3;
    for (int i = 0, j = 0; i < 4; i++)
        if (bit_nt[i] == 100) bit_nt[i] = (uint8_t)other_bit[j++];
    /* 238-253 */
    memset(p->alphabet, 0, 256);
    memset(p->rev_alphabet, 0, 256);
    static const char up[4] = {'A', 'C', 'G', 'T'}, lo[4] = {'a', 'c', 'g', 't'};
    for (int i = 0; i < 4; i++) {
        p->alphabet[(unsigned char)up[i]] = p->alphabet[(unsigned char)lo[i]] = bit_nt[i];
        p->rev_alphabet[(unsigned char)up[i]] = p->rev_alphabet[(unsigned char)lo[i]] = bit_nt[3 - i];
    }
    /* 259-260 */
    for (int i = 0; i < 4; i++) p->useful_nt[bit_nt[i]] = nt_code[i];
    for (int i = 0; i < 4; i++) p->useful_nt[bit_nt[i] + 4] = (char)tolower((unsigned char)nt_code[i]);
    p->useful_nt[8] = 0;
    p->new_rule = !(p->readnt_cnt == 1 && p->readnts[0] != '-');
    return 0;
}

/* ------------------------------------------------------------------ reference load */

static char *slurp(const char *path, size_t *len) {
    gzFile f = gzopen(path, "rb"); /* transparently reads plain files too */
    if (!f) return NULL;
    size_t cap = 1 << 20, n = 0;
    char *buf = (char *)malloc(cap);
    for (;;) {
        if (cap - n < (1 << 19)) {
            cap *= 2;
            buf = (char *)realloc(buf, cap);
        }
        int got = gzread(f, buf + n, (unsigned)(cap - n > (1u << 30) ? (1u << 30) : cap - n));
        if (got <= 0) break;
        n += (size_t)got;
    }
    gzclose(f);
    *len = n;
    return buf;
}

static int is_ws(int c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

typedef struct contig_tmp {
    char *name;
    char *seq;
    size_t len;
} contig_tmp;

static int in_set(const char *set, int c) { return c != 0 && strchr(set, c) != NULL; }

/* refbase.cpp:17-38 (LoadNextSeq: operator>> tokenisation), 63-128, 186-252 */
orc_ref *orc_ref_load_fasta_mem(const char *buf, size_t len, orc_param *p) {
    size_t pos = 0, ncap = 16, nct = 0;
    contig_tmp *ct = (contig_tmp *)malloc(ncap * sizeof(*ct));
    for (;;) {
        /* fin>>c : skip ws, take one char (assumed '>') */
        while (pos < len && is_ws((unsigned char)buf[pos])) pos++;
        if (pos >= len) break;
        pos++;
        /* fin>>_name */
        while (pos < len && is_ws((unsigned char)buf[pos])) pos++;
        size_t ns = pos;
        while (pos < len && !is_ws((unsigned char)buf[pos])) pos++;
        char *name = strndup(buf + ns, pos - ns);
        /* getline: rest of the header line */
        while (pos < len && buf[pos] != '\n') pos++;
        if (pos < len) pos++;
        size_t scap = 1 << 16, sl = 0;
        char *seq = (char *)malloc(scap);
        for (;;) {
            while (pos < len && is_ws((unsigned char)buf[pos])) pos++;
            if (pos >= len) break;
            if (buf[pos] == '>') break;
            size_t ts = pos;
            while (pos < len && !is_ws((unsigned char)buf[pos])) pos++;
            size_t tl = pos - ts;
            if (sl + tl + 1 > scap) {
                while (sl + tl + 1 > scap) scap *= 2;
                seq = (char *)realloc(seq, scap);
            }
            memcpy(seq + sl, buf + ts, tl);
            sl += tl;
        }
        if (sl == 0) { /* LoadNextSeq returns _length==0 -> the while loop at 192 ends */
            free(name);
            free(seq);
            break;
        }
        if (nct == ncap) {
            ncap *= 2;
            ct = (contig_tmp *)realloc(ct, ncap * sizeof(*ct));
        }
        ct[nct].name = name;
        ct[nct].seq = seq;
        ct[nct].len = sl;
        nct++;
    }

    orc_ref *r = (orc_ref *)calloc(1, sizeof(*r));
    r->owns_arrays = 1;
    r->ncontig = (uint32_t)nct;
    r->name = (char **)calloc(nct ? nct : 1, sizeof(char *));
    r->size = (uint32_t *)calloc(nct ? nct : 1, 4);
    r->rc_offset = (uint32_t *)calloc(nct ? nct : 1, 4);
    r->nword = (uint32_t *)calloc(nct ? nct : 1, 4);
    r->ref_anchor = (uint32_t *)calloc(nct + 1, 4);
    uint32_t s = 0;
    r->ref_anchor[0] = ORC_REF_MARGIN * ORC_SEGLEN; /* 223 */
    for (size_t i = 0; i < nct; i++) {
        uint32_t L = (uint32_t)ct[i].len;
        r->name[i] = ct[i].name;
        r->size[i] = L;
        r->nword[i] = (L + (ORC_SEGLEN - 1)) / ORC_SEGLEN + ORC_BINSEQPAD; /* 64 */
        r->rc_offset[i] = r->nword[i] * ORC_SEGLEN;                        /* 195 */
        r->sum_length += L;
        s += r->nword[i];
        r->ref_anchor[i + 1] = (s + ORC_REF_MARGIN) * ORC_SEGLEN; /* 225 */
    }
    r->nwords_total = (uint64_t)s + 2 * ORC_REF_MARGIN; /* 230; margins are uninitialised in the
        reference; hits that touch them are always rejected by AddHit's bounds check, so zero is safe */
    r->xref[0] = (uint64_t *)calloc(r->nwords_total, 8);
    r->xref[1] = (uint64_t *)calloc(r->nwords_total, 8);

    size_t bcap = 1024;
    r->blocks = (orc_block *)malloc(bcap * sizeof(orc_block));
    r->nblocks = 0;
    uint64_t woff = ORC_REF_MARGIN;
    for (size_t ci = 0; ci < nct; ci++) {
        uint32_t L = (uint32_t)ct[ci].len, n = r->nword[ci];
        uint32_t total_len = n * ORC_SEGLEN;
        /* BinSeq 63-83: pad with 'N' to n*32 */
        char *sq = (char *)realloc(ct[ci].seq, (size_t)total_len + 1);
        memset(sq + L, 'N', total_len - L);
        sq[total_len] = 0;
        uint64_t *fw = r->xref[0] + woff, *rc = r->xref[1] + woff;
        for (uint32_t i = 0; i < n; i++) {
            uint64_t w = 0;
            for (uint32_t j = 0; j < ORC_SEGLEN; j++) w = (w << 2) | p->alphabet[(unsigned char)sq[i * ORC_SEGLEN + j]];
            fw[i] = w;
        }
        /* cBinSeq 85-101 */
        for (uint32_t i = 0; i < n; i++) {
            uint64_t w = 0;
            const char *q = sq + total_len - 1 - (size_t)i * ORC_SEGLEN;
            for (uint32_t j = 0; j < ORC_SEGLEN; j++) w = (w << 2) | p->rev_alphabet[(unsigned char)*(q - j)];
            rc[i] = w;
        }
        /* UnmaskRegion 103-128.  find_first_of runs over the padded buffer. */
        uint32_t bb = 0, be = 0;
        while (be < L) {
            uint32_t x = be;
            while (x < total_len && !in_set("ACGTacgt", (unsigned char)sq[x])) x++;
            bb = x; /* npos if none: x==total_len > L */
            if (bb > L || x == total_len) break;
            x = bb;
            while (x < total_len && !in_set("NXnx", (unsigned char)sq[x])) x++;
            be = x <= L ? x : L;
            if (be - bb < 16) continue;
            /* the "merge if gap<5" test (116-118) never fires: the last pushed block is the RC twin */
            if (r->nblocks + 2 > bcap) {
                bcap *= 2;
                r->blocks = (orc_block *)realloc(r->blocks, bcap * sizeof(orc_block));
            }
            orc_block b = {(uint32_t)(2 * ci), bb, be};
            orc_block cb = {(uint32_t)(2 * ci + 1), total_len - be, total_len - bb};
            r->blocks[r->nblocks++] = b;
            r->blocks[r->nblocks++] = cb;
        }
        free(sq);
        woff += n;
    }
    free(ct);
    p->total_ref_seq = r->ncontig; /* 220 */
    return r;
}

orc_ref *orc_ref_load_fasta(const char *path, orc_param *p) {
    size_t len;
    char *buf = slurp(path, &len);
    if (!buf) return NULL;
    orc_ref *r = orc_ref_load_fasta_mem(buf, len, p);
    free(buf);
    return r;
}

/* refbase.cpp:254-255 */
static uint32_t make_seed(const orc_param *p, const uint64_t *m, int a) {
    uint64_t v = (m[0] << (a * 2)) | ((m[1] >> 1) >> (63 - a * 2));
    return orc_XT((uint32_t)(v >> p->seed_bits_lz));
}

static int block_cmp(const void *a, const void *b) { /* refbase.cpp:184 */
    const orc_block *x = (const orc_block *)a, *y = (const orc_block *)b;
    if (x->id != y->id) return x->id < y->id ? -1 : 1;
    if (x->begin != y->begin) return x->begin < y->begin ? -1 : 1;
    return 0;
}

static int u32_cmp(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

/* refbase.cpp:261-274 InitialIndex, 303-325 t_CalKmerFreq, 327-367 AllocIndex, 419-439 t_FillIndex */
void orc_ref_build_index(orc_ref *r, orc_param *p) {
    uint32_t K = p->seed_size, I = p->index_interval;
    uint32_t total = 1;
    for (uint32_t i = 0; i < K; i++) total *= 3;
    r->total_kmers = total;
    r->n_tot = (uint32_t *)calloc(total, 4);
    r->n_fwd = (uint32_t *)calloc(total, 4);
    r->off = (uint64_t *)calloc((size_t)total + 1, 8);
    qsort(r->blocks, r->nblocks, sizeof(orc_block), block_cmp); /* 219 */

    uint64_t *base[2];
    /* word offset of contig c inside xref */
    uint64_t *cw = (uint64_t *)malloc((r->ncontig + 1) * 8);
    cw[0] = ORC_REF_MARGIN;
    for (uint32_t c = 0; c < r->ncontig; c++) cw[c + 1] = cw[c] + r->nword[c];
    base[0] = r->xref[0];
    base[1] = r->xref[1];

    uint32_t *cnt[2];
    cnt[0] = r->n_fwd;                        /* n[0] during counting = fwd */
    cnt[1] = (uint32_t *)calloc(total, 4);    /* n[1] during counting = RC  */
    for (size_t b = 0; b < r->nblocks; b++) {
        const orc_block *bl = &r->blocks[b];
        const uint64_t *m = base[bl->id & 1] + cw[bl->id >> 1];
        uint32_t i2 = ((bl->end - K) / I) * I;
        for (uint32_t i = (bl->begin / I) * I; i <= i2; i += I) cnt[bl->id & 1][make_seed(p, m + i / ORC_SEGLEN, (int)(i % ORC_SEGLEN))]++;
    }
    uint64_t acc = 0;
    uint32_t *kc = (uint32_t *)malloc((size_t)total * 4);
    for (uint32_t k = 0; k < total; k++) {
        uint32_t t = cnt[0][k] + cnt[1][k];
        kc[k] = t;
        r->n_tot[k] = t;
        r->off[k] = acc;
        acc += t;
    }
    r->off[total] = acc;
    r->nlocs = acc;
    r->locs = (uint32_t *)malloc((acc ? acc : 1) * 4);
    /* fill: fwd entries first (ascending), then RC entries (ascending)  (358-359, 433) */
    uint32_t *cur_f = (uint32_t *)calloc(total, 4), *cur_r = (uint32_t *)calloc(total, 4);
    for (size_t b = 0; b < r->nblocks; b++) {
        const orc_block *bl = &r->blocks[b];
        int rcs = bl->id & 1;
        const uint64_t *m = base[rcs] + cw[bl->id >> 1];
        uint32_t i2 = ((bl->end - K) / I) * I;
        for (uint32_t i = (bl->begin / I) * I; i <= i2; i += I) {
            uint32_t sd = make_seed(p, m + i / ORC_SEGLEN, (int)(i % ORC_SEGLEN));
            uint32_t v = r->ref_anchor[bl->id >> 1] + i; /* hit2int 485-487 */
            if (!rcs) r->locs[r->off[sd] + cur_f[sd]++] = v;
            else r->locs[r->off[sd] + cnt[0][sd] + cur_r[sd]++] = v;
        }
    }
    free(cur_f);
    free(cur_r);
    free(cnt[1]);
    free(cw);
    /* 362-363: float arithmetic; sort covers total_kmers-1 elements */
    volatile float one_minus = 1 - p->max_kmer_ratio;
    volatile float prod = (float)total * one_minus;
    uint32_t idx = (uint32_t)prod - 1;
    /* kmer_count[idx] after sorting elements [0, total-1): the last element stays in place */
    if (idx == total - 1) p->max_kmer_num = kc[total - 1];
    else {
        uint32_t mx = 0;
        for (uint32_t k = 0; k + 1 < total; k++) if (kc[k] > mx) mx = kc[k];
        if (mx < (1u << 24)) { /* order statistic through a histogram == value at sorted position */
            uint64_t *hist = (uint64_t *)calloc((size_t)mx + 1, 8), run = 0;
            for (uint32_t k = 0; k + 1 < total; k++) hist[kc[k]]++;
            uint32_t v = 0;
            for (; v <= mx; v++) { run += hist[v]; if (run > idx) break; }
            p->max_kmer_num = v;
            free(hist);
        } else {
            qsort(kc, (size_t)total - 1, 4, u32_cmp);
            p->max_kmer_num = kc[idx];
        }
    }
    free(kc);
}

orc_ref *orc_ref_from_arrays(uint32_t ncontig, const char *const *names, const uint32_t *size,
                             uint64_t *xref_fwd, uint64_t *xref_rc, uint64_t nwords_total,
                             uint32_t total_kmers, uint32_t *n_tot, uint32_t *n_fwd, uint64_t *off,
                             uint32_t *locs, uint64_t nlocs) {
    orc_ref *r = (orc_ref *)calloc(1, sizeof(*r));
    r->ncontig = ncontig;
    r->name = (char **)calloc(ncontig, sizeof(char *));
    r->size = (uint32_t *)calloc(ncontig, 4);
    r->rc_offset = (uint32_t *)calloc(ncontig, 4);
    r->nword = (uint32_t *)calloc(ncontig, 4);
    r->ref_anchor = (uint32_t *)calloc(ncontig + 1, 4);
    uint32_t s = 0;
    r->ref_anchor[0] = ORC_REF_MARGIN * ORC_SEGLEN;
    for (uint32_t i = 0; i < ncontig; i++) {
        r->name[i] = strdup(names[i]);
        r->size[i] = size[i];
        r->nword[i] = (size[i] + (ORC_SEGLEN - 1)) / ORC_SEGLEN + ORC_BINSEQPAD;
        r->rc_offset[i] = r->nword[i] * ORC_SEGLEN;
        r->sum_length += size[i];
        s += r->nword[i];
        r->ref_anchor[i + 1] = (s + ORC_REF_MARGIN) * ORC_SEGLEN;
    }
    r->nwords_total = nwords_total;
    r->xref[0] = xref_fwd;
    r->xref[1] = xref_rc;
    r->total_kmers = total_kmers;
    r->n_tot = n_tot;
    r->n_fwd = n_fwd;
    r->off = off;
    r->locs = locs;
    r->nlocs = nlocs;
    r->owns_arrays = 0;
    return r;
}

void orc_ref_free(orc_ref *r) {
    if (!r) return;
    for (uint32_t i = 0; i < r->ncontig; i++) free(r->name[i]);
    free(r->name);
    free(r->size);
    free(r->rc_offset);
    free(r->nword);
    free(r->ref_anchor);
    free(r->blocks);
    if (r->owns_arrays) {
        free(r->xref[0]);
        free(r->xref[1]);
        free(r->n_tot);
        free(r->n_fwd);
        free(r->off);
        free(r->locs);
    }
    free(r);
}

/* ------------------------------------------------------------------ aligner state */

typedef struct locset { /* std::set<ref_loc_t> hitset[total_ref_seq] (align.h:94), as one hash set */
    uint64_t *key;
    uint32_t *stamp;
    uint32_t cap, gen;
} locset;

static void locset_init(locset *s) {
    s->cap = 1u << 16;
    s->key = (uint64_t *)calloc(s->cap, 8);
    s->stamp = (uint32_t *)calloc(s->cap, 4);
    s->gen = 1;
}
static void locset_clear(locset *s) {
    if (++s->gen == 0) {
        memset(s->stamp, 0, (size_t)s->cap * 4);
        s->gen = 1;
    }
}
/* returns 1 if newly inserted (like set::insert(...).second) */
static int locset_insert(locset *s, uint32_t contig, uint32_t loc) {
    uint64_t k = ((uint64_t)contig << 32) | loc;
    uint64_t h = k * 0x9E3779B97F4A7C15ULL;
    uint32_t i = (uint32_t)(h >> 40) & (s->cap - 1);
    for (;;) {
        if (s->stamp[i] != s->gen) {
            s->stamp[i] = s->gen;
            s->key[i] = k;
            return 1;
        }
        if (s->key[i] == k) return 0;
        i = (i + 1) & (s->cap - 1);
    }
}

struct orc_aligner {
    const orc_param *p;
    const orc_ref *r;
    /* align.h:69-98 */
    uint32_t tmp_snp, snp_thres, map_readlen, raw_readlen;
    uint32_t read_chain_index, ref_chain_index;
    uint32_t n_aligned, n_unique, n_multiple, read_max_snp_num, seedseg_num;
    uint32_t xflag_chain[2], xseed_start_offset[2];
    orc_hit _hit;
    uint32_t N_count;
    uint64_t xseq[2][ORC_FIXELEMENT * 3];
    uint32_t end_element, end_offset;
    uint32_t xseeds[2][ORC_MAXSNPS + 1][16];
    uint32_t xseed_array[2][ORC_FIXSIZE - ORC_SEGLEN], xseedreg_array[2][ORC_FIXSIZE - ORC_SEGLEN];
    uint32_t x_cur_n_hit[2][ORC_MAXSNPS + 1];
    uint32_t xseed_start_array[2][ORC_MAXSNPS + 1];
    int32_t seg_weight[2][ORC_MAXSNPS + 1], seg_order[2][ORC_MAXSNPS + 1]; /* xseedindex */
    uint32_t mm_index[2 * ORC_MAXGAPS + 1][ORC_MAXSNPS + 1];
    orc_hit (*xhits)[ORC_MAXSNPS + 1][ORC_MAXHITS + 1]; /* HitMatrix[2] */
    locset hitset, ghitset;
    char *outseq[2], *outqual[2]; /* Reverse_Seq */
    size_t outcap;
    orc_loghit *log;
    size_t nlog, logcap;
    orc_counters c;
    uint32_t cur_index; /* _pread->index */
};

orc_aligner *orc_aligner_new(const orc_param *p, const orc_ref *r) {
    orc_aligner *a = (orc_aligner *)calloc(1, sizeof(*a));
    a->p = p;
    a->r = r;
    a->xhits = calloc(2, sizeof(*a->xhits));
    locset_init(&a->hitset);
    locset_init(&a->ghitset);
    a->logcap = 1024;
    a->log = (orc_loghit *)malloc(a->logcap * sizeof(orc_loghit));
    /* xseed_start_offset is an uninitialised member in the reference (align.h:73); the
       restatement (and the product) define the initial value as 0 */
    return a;
}

void orc_aligner_free(orc_aligner *a) {
    if (!a) return;
    free(a->xhits);
    free(a->hitset.key);
    free(a->hitset.stamp);
    free(a->ghitset.key);
    free(a->ghitset.stamp);
    for (int i = 0; i < 2; i++) {
        free(a->outseq[i]);
        free(a->outqual[i]);
    }
    free(a->log);
    free(a);
}

const orc_counters *orc_aligner_counters(const orc_aligner *a) { return &a->c; }
void orc_aligner_stats(const orc_aligner *a, uint32_t *n_aligned, uint32_t *n_unique, uint32_t *n_multiple) {
    *n_aligned = a->n_aligned;
    *n_unique = a->n_unique;
    *n_multiple = a->n_multiple;
}
uint32_t orc_read_max_snp(const orc_aligner *a) { return a->read_max_snp_num; }
uint32_t orc_n_hit(const orc_aligner *a, int chain, int level) { return a->x_cur_n_hit[chain][level]; }
const orc_hit *orc_hits(const orc_aligner *a, int chain, int level) { return a->xhits[chain][level]; }
size_t orc_hit_log(const orc_aligner *a, const orc_loghit **log) {
    *log = a->log;
    return a->nlog;
}
void orc_seed_state(const orc_aligner *a, uint32_t start_off[2], uint32_t start_arr[2][16],
                    int32_t seg_weight[2][16], int32_t seg_order[2][16], uint32_t *seedseg_num) {
    for (int c = 0; c < 2; c++) {
        start_off[c] = a->xseed_start_offset[c];
        for (int i = 0; i < 16; i++) {
            start_arr[c][i] = a->xseed_start_array[c][i];
            seg_weight[c][i] = a->seg_weight[c][i];
            seg_order[c][i] = a->seg_order[c][i];
        }
    }
    *seedseg_num = a->seedseg_num;
}
const uint64_t *orc_xseq(const orc_aligner *a, int chain) { return a->xseq[chain]; }
const uint32_t *orc_seed_array(const orc_aligner *a, int chain) { return a->xseed_array[chain]; }

/* ------------------------------------------------------------------ FilterReads */

/* align.cpp:418-435 */
static int trim_adapter(orc_aligner *a, orc_read *rd) {
    const orc_param *p = a->p;
    size_t L = strlen(rd->seq);
    a->raw_readlen = (uint32_t)L;
    for (uint32_t i = 0; i < p->n_adapter; i++) {
        size_t al = strlen(p->adapter[i]);
        /* pos < size()-4 is an unsigned comparison in the reference */
        for (uint32_t pos = p->seed_size + p->index_interval - 1; (size_t)pos < L - 4 && L >= 4; pos++) {
            uint32_t m0 = 0, k;
            for (k = 0; k < al && k < 15 && pos + k < L; k++) {
                if ((m0 += (p->adapter[i][k] != rd->seq[pos + k])) > 4) break;
            }
            if (k >= m0 * 5 && k > 3) {
                rd->seq[pos] = 0;
                if (strlen(rd->qual) > pos) rd->qual[pos] = 0;
                return 1;
            }
        }
    }
    return 0;
}

/* align.cpp:51-76. qual buffers are owned by the caller with room for seq length. */
static int trim_lowqual(orc_aligner *a, orc_read *rd) {
    const orc_param *p = a->p;
    size_t L = strlen(rd->seq);
    if (L != strlen(rd->qual)) {
        memset(rd->qual, p->zero_qual + p->default_qual, L);
        rd->qual[L] = 0;
    }
    uint8_t qual_thres = (uint8_t)(p->zero_qual + p->trim_qual_threshold);
    if (p->zero_qual != '!') { /* out_sam is always set (main.cpp:411-415) */
        for (char *q = rd->qual; *q; q++) *q -= (p->zero_qual - '!');
        qual_thres -= (p->zero_qual - '!');
    }
    if (p->trim_qual_threshold == 0) return 0;
    uint32_t i = (uint32_t)L;
    for (; i > 0; i--)
        if ((uint8_t)rd->qual[i - 1] > qual_thres) break; /* *_sq is a (signed) char; quals are ASCII */
    if (i < p->seed_size + p->index_interval - 1) return 1;
    rd->qual[i] = 0;
    rd->seq[i] = 0;
    return 0;
}

/* align.cpp:548-563 */
int orc_filter_read(orc_aligner *a, orc_read *rd) {
    const orc_param *p = a->p;
    size_t L = strlen(rd->seq);
    if (p->max_snp_num < 100) a->read_max_snp_num = p->max_snp_num;
    else a->read_max_snp_num = (uint32_t)((p->max_snp_num - 100) / 100.0 * L + 0.5);
    if (p->gap > 0) a->read_max_snp_num = a->read_max_snp_num + 1 + p->gap;
    if (a->read_max_snp_num > ORC_MAXSNPS) a->read_max_snp_num = ORC_MAXSNPS;
    trim_adapter(a, rd);
    if (trim_lowqual(a, rd) != 0) return 1;
    L = strlen(rd->seq);
    if (L < p->min_read_size) return 1;
    /* CountNs 40-47 */
    uint32_t n = 0;
    for (const char *s = rd->seq; *s; s++)
        if (!p->reg_alphabet[(unsigned char)*s]) n++;
    if (p->N_mis) a->N_count = n;
    if (n > p->max_ns) return 1;
    a->read_max_snp_num = (a->read_max_snp_num + 1) * ((uint32_t)L - 1) / a->raw_readlen;
    return 0;
}

/* ------------------------------------------------------------------ read packing */

static char rev_char(int c) { /* param.cpp:146-156 */
    switch (c) {
    case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
    case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
    default: return 'N';
    }
}

/* align.h:349-355 */
static void reverse_seq(orc_aligner *a, const orc_read *rd) {
    size_t L = strlen(rd->seq), Q = strlen(rd->qual);
    size_t need = (L > Q ? L : Q) + 1;
    if (need > a->outcap) {
        a->outcap = need * 2;
        for (int i = 0; i < 2; i++) {
            a->outseq[i] = (char *)realloc(a->outseq[i], a->outcap);
            a->outqual[i] = (char *)realloc(a->outqual[i], a->outcap);
        }
    }
    memcpy(a->outseq[0], rd->seq, L + 1);
    for (size_t i = 0; i < L; i++) a->outseq[1][i] = rev_char((unsigned char)rd->seq[L - 1 - i]);
    a->outseq[1][L] = 0;
    memcpy(a->outqual[0], rd->qual, Q + 1);
    for (size_t i = 0; i < Q; i++) a->outqual[1][i] = rd->qual[Q - 1 - i];
    a->outqual[1][Q] = 0;
}

/* align.cpp:79-150 (ConvertBinaySeq) and 153-226 (ConvertBinarySeq): the two differ only in
   the third (Mread) plane, which the one-way rule never reads. */
static void convert_binary_seq(orc_aligner *a, const orc_read *rd) {
    const orc_param *p = a->p;
    size_t L = strlen(rd->seq);
    reverse_seq(a, rd);
    a->xflag_chain[0] = (p->chains == 1) || ((p->chains <= 1) == (rd->readset < 2));
    a->xflag_chain[1] = (p->chains == 1) || ((p->chains <= 1) == (rd->readset == 2));
    for (int c = 0; c < 2; c++) {
        if (!a->xflag_chain[c]) continue;
        const uint8_t *al = c ? p->rev_alphabet : p->alphabet;
        const uint8_t *am = c ? p->rev_alphabet_mread : p->alphabet_mread;
        uint32_t h = 0, s = 0, sb = 0, i;
        uint64_t _a = 0, _b = 0, _c = 0;
        for (i = 1; i <= L; i++) {
            unsigned char ch = (unsigned char)(c ? rd->seq[L - i] : rd->seq[i - 1]);
            _a <<= 2; _b <<= 2; _c <<= 2;
            _a |= al[ch];
            _b |= p->reg_alphabet[ch];
            _c |= am[ch];
            if (i > p->seed_size) {
                s <<= 2; s |= (uint32_t)(_a & 0x3);
                a->xseed_array[c][i - p->seed_size] = orc_XT(s & p->seed_bits);
                sb <<= 2; sb |= (uint32_t)(_b & 0x3);
                a->xseedreg_array[c][i - p->seed_size] = (~sb) & p->seed_bits;
            } else if (i == p->seed_size) {
                s = (uint32_t)_a;
                a->xseed_array[c][0] = orc_XT(s);
                sb = (uint32_t)_b;
                a->xseedreg_array[c][0] = (~sb) & p->seed_bits;
            }
            if (0 == i % ORC_SEGLEN) {
                a->xseq[c][h] = _a;
                a->xseq[c][h + ORC_FIXELEMENT] = _b;
                a->xseq[c][h + ORC_FIXELEMENT * 2] = _c;
                _a = _b = _c = 0;
                h++;
            }
        }
        for (; i != ORC_FIXSIZE + 1; i++) {
            _a <<= 2; _b <<= 2; _c <<= 2;
            if (0 == i % ORC_SEGLEN) {
                a->xseq[c][h] = _a;
                a->xseq[c][h + ORC_FIXELEMENT] = _b;
                a->xseq[c][h + ORC_FIXELEMENT * 2] = _c;
                _a = _b = _c = 0;
                h++;
            }
        }
    }
}

/* ------------------------------------------------------------------ seed ordering */

/* align.cpp:526-540 */
static int count_seeds(orc_aligner *a, int n, uint32_t start) {
    const orc_param *p = a->p;
    uint32_t total = 0, k = 0;
    for (uint32_t i = 0; i < p->index_interval; i++) {
        uint32_t pos = p->profile[n][i] + start - i;
        uint32_t s = a->xseed_array[a->read_chain_index][pos];
        uint32_t r = a->xseedreg_array[a->read_chain_index][pos];
        if (r) k = 12; /* sticky for the remaining phases */
        total += a->r->n_tot[s] << k;
        a->c.hdr_lookups++;
    }
    if (total == 0) total = 9999999;
    return (int)total;
}

/* align.cpp:542-546 */
static uint32_t get_total_seed_loc(orc_aligner *a, uint32_t start) {
    uint32_t total = 0;
    for (uint32_t i = 0; i < a->seedseg_num; i++) total += (uint32_t)count_seeds(a, (int)i, start);
    return total;
}

/* align.cpp:500-524 */
static void adjust_seed_start_array(orc_aligner *a) {
    const orc_param *p = a->p;
    uint32_t c = a->read_chain_index;
    uint32_t max_offset = (a->map_readlen - p->index_interval + 1) % p->seed_size;
    for (uint32_t i = 0; i < a->seedseg_num; i++) {
        uint32_t ptr = (i % 2 == 0) ? i / 2 : a->seedseg_num - 1 - i / 2;
        uint32_t total = 0xffffffffu;
        uint32_t start = (ptr == 0) ? 0 : a->xseed_start_array[c][ptr - 1];
        uint32_t end = (ptr == a->seedseg_num - 1) ? max_offset : a->xseed_start_array[c][ptr + 1];
        a->xseed_start_array[c][ptr] = start;
        for (uint32_t ii = start; ii <= end; ii++) {
            uint32_t tt = (uint32_t)count_seeds(a, (int)ptr, ii);
            if (tt < total) {
                total = tt;
                a->xseed_start_array[c][ptr] = ii;
            }
        }
    }
}

/* align.cpp:468-498 */
static void reorder_seed(orc_aligner *a) {
    const orc_param *p = a->p;
    for (a->read_chain_index = 0; a->read_chain_index < 2; a->read_chain_index++) {
        uint32_t c = a->read_chain_index;
        if (!a->xflag_chain[c]) continue;
        uint32_t total = 0xffffffffu;
        uint32_t ii = (a->map_readlen - p->index_interval + 1) % p->seed_size;
        for (uint32_t i = 0; i < ii; i++) { /* ii==0: the previous read's value is kept (475-480) */
            uint32_t tt = get_total_seed_loc(a, i);
            if (tt < total) {
                total = tt;
                a->xseed_start_offset[c] = i;
            }
        }
        for (uint32_t i = 0; i < a->seedseg_num; i++) a->xseed_start_array[c][i] = a->xseed_start_offset[c];
        adjust_seed_start_array(a);
        for (uint32_t i = 0; i < a->seedseg_num; i++) {
            for (uint32_t k = 0; k < p->index_interval; k++)
                a->xseeds[c][i][k] = a->xseed_array[c][p->profile[i][k] + a->xseed_start_array[c][i] - k];
            a->seg_weight[c][i] = count_seeds(a, (int)i, a->xseed_start_array[c][i]);
            a->seg_order[c][i] = (int32_t)i;
        }
        /* sort pair<int,int> ascending (495): keys are unique through .second, so any sort works */
        for (uint32_t i = 1; i < a->seedseg_num; i++) {
            int32_t w = a->seg_weight[c][i], o = a->seg_order[c][i];
            int j = (int)i - 1;
            while (j >= 0 && (a->seg_weight[c][j] > w || (a->seg_weight[c][j] == w && a->seg_order[c][j] > o))) {
                a->seg_weight[c][j + 1] = a->seg_weight[c][j];
                a->seg_order[c][j + 1] = a->seg_order[c][j];
                j--;
            }
            a->seg_weight[c][j + 1] = w;
            a->seg_order[c][j + 1] = o;
        }
    }
}

/* ------------------------------------------------------------------ candidate scoring */

/* number of 64-bit reference words that hold read bases for (len, offset) */
static uint32_t words_needed(const orc_aligner *a, uint32_t offset) {
    return (a->map_readlen + (offset >> 1) + ORC_SEGLEN - 1) / ORC_SEGLEN;
}

/* align.h:118-131 (one-way) / 199-239 (multi-way, "_new").  The reference walks all 16 words
   unless it exits early; words past the read contribute 0, so the restatement stops after the
   last word that holds read bases and counts only those in W. */
static void count_mismatch(orc_aligner *a, const uint64_t *q, uint32_t offset, const uint64_t *s) {
    const int nr = a->p->new_rule;
    uint32_t nw = words_needed(a, offset);
    a->tmp_snp = a->N_count;
    for (uint32_t i = 0; i < nw && i < ORC_FIXELEMENT; i++) {
        uint64_t rw, mw, cw;
        if (i == 0) {
            rw = q[0] >> offset;
            mw = q[ORC_FIXELEMENT] >> offset;
            cw = q[ORC_FIXELEMENT * 2] >> offset;
        } else {
            rw = ((q[i - 1] << 1) << (63 - offset)) | (q[i] >> offset);
            mw = ((q[ORC_FIXELEMENT - 1 + i] << 1) << (63 - offset)) | (q[i + ORC_FIXELEMENT] >> offset);
            cw = ((q[ORC_FIXELEMENT * 2 - 1 + i] << 1) << (63 - offset)) | (q[i + ORC_FIXELEMENT * 2] >> offset);
        }
        a->c.ref_words++;
        uint64_t x;
        if (!nr) x = ((rw & orc_XC64(s[i])) ^ s[i]) & mw;
        else {
            uint64_t M2 = orc_XC64(s[i]) | cw;
            uint64_t M3 = orc_M2_judge(M2);
            uint64_t M4 = (((~M3) & M2) | (M3 & rw)) ^ s[i];
            x = M4 & mw;
        }
        if ((a->tmp_snp += orc_XM64(x)) > a->snp_thres) return;
    }
}

static uint64_t gap_cmp_word(const orc_aligner *a, const uint64_t *q, uint64_t tmp, int i) {
    if (!a->p->new_rule) return tmp ^ (q[i] & orc_XC64(tmp)); /* align.h:143 */
    uint64_t M2 = orc_XC64(tmp) | q[i + ORC_FIXELEMENT * 2];   /* align.h:253-256 */
    uint64_t M3 = orc_M2_judge(M2);
    return tmp ^ (((~M3) & M2) | (M3 & q[i]));
}

/* align.h:133-168 / 241-288: positions (from the left) of the first snp_thres-1 mismatches */
static uint32_t mismatch_pattern0(orc_aligner *a, const uint64_t *q, const uint64_t *s, uint32_t offset) {
    uint32_t *mm = a->mm_index[0];
    int i, j, jj, ss = 0;
    uint64_t tmp;
    for (i = 0; i < (int)a->end_element; i++) {
        tmp = (s[i] << offset) | ((s[i + 1] >> (63 - offset)) >> 1);
        a->c.ref_words++;
        tmp = gap_cmp_word(a, q, tmp, i);
        j = (int)(i * ORC_SEGLEN) - 1;
        while (tmp) {
            jj = __builtin_clzll(tmp) >> 1;
            j += jj + 1;
            mm[ss++] = (uint32_t)j;
            if (ss > (int)a->snp_thres - 2) return (uint32_t)j;
            tmp <<= 2;
            tmp <<= (jj << 1);
        }
    }
    tmp = (s[i] << offset) | ((s[i + 1] >> (63 - offset)) >> 1);
    a->c.ref_words++;
    tmp = gap_cmp_word(a, q, tmp, i);
    tmp >>= a->end_offset;
    tmp <<= a->end_offset;
    j = (int)(i * ORC_SEGLEN) - 1;
    while (tmp) {
        jj = __builtin_clzll(tmp) >> 1;
        j += jj + 1;
        mm[ss++] = (uint32_t)j;
        if (ss > (int)a->snp_thres - 2) return (uint32_t)j;
        tmp <<= 2;
        tmp <<= (jj << 1);
    }
    for (; ss <= (int)a->snp_thres - 2; ss++) mm[ss] = a->map_readlen;
    return a->map_readlen;
}

/* align.h:170-196 / 289-327: distances from the right end, reference start shifted */
static void mismatch_pattern1(orc_aligner *a, const uint64_t *q, const uint64_t *s, uint32_t gap_index, uint32_t offset) {
    uint32_t *mm = a->mm_index[gap_index];
    int i, ii, j, jj, ss = 0;
    uint64_t tmp;
    for (i = (int)a->end_element, ii = 0; i >= 0; i--, ii += ORC_SEGLEN) {
        tmp = (s[i] << offset) | ((s[i + 1] >> (63 - offset)) >> 1);
        a->c.ref_words++;
        tmp = gap_cmp_word(a, q, tmp, i);
        uint32_t shift = a->end_offset * (i == (int)a->end_element);
        tmp = (tmp >> shift) << shift;
        j = ii - (int)(a->end_offset >> 1) - 1;
        while (tmp) {
            jj = __builtin_ctzll(tmp) >> 1;
            j += jj + 1;
            mm[ss++] = (uint32_t)j;
            if (ss > (int)a->snp_thres - 2) return;
            tmp >>= 2;
            tmp >>= (jj << 1);
        }
    }
    for (; ss <= (int)a->snp_thres - 2; ss++) mm[ss] = a->map_readlen;
}

/* align.cpp:319-346 */
static orc_hit int2hit(orc_aligner *a, orc_hit gh, int gap_size, uint32_t gap_pos) {
    const orc_ref *r = a->r;
    uint32_t left = 0, right = r->ncontig, mid;
    while (left < right - 1) {
        mid = (left + right) / 2;
        if (gh.loc >= r->ref_anchor[mid]) left = mid;
        else right = mid;
    }
    gh.chr = (left * 2 + a->ref_chain_index) & 0x3FFFF; /* bit32_t chr:18 */
    gh.loc -= r->ref_anchor[left];
    gh.gap_size = gap_size;
    gh.gap_pos = gap_pos & 0x1FF; /* bit16_t gap_pos:9 */
    gh.strand = ((a->ref_chain_index << 1) | a->read_chain_index) & 3;
    if (a->ref_chain_index) {
        gh.loc = r->rc_offset[gh.chr >> 1] - a->map_readlen - gh.loc;
        gh.gap_pos = (uint32_t)((int)a->map_readlen + (gh.gap_size < 0) * (gh.gap_size) - (int)gh.gap_pos) & 0x1FF;
        gh.loc -= (uint32_t)gh.gap_size;
    }
    return gh;
}

/* align.h:329-347 */
static uint32_t add_hit(orc_aligner *a, const orc_hit *gh, uint32_t w, uint32_t mode) {
    const orc_param *p = a->p;
    if ((int)gh->loc < 0) return 0;
    if (gh->loc + a->map_readlen > a->r->size[gh->chr >> 1]) return 0;
    if (gh->gap_size) {
        if (!locset_insert(&a->ghitset, gh->chr >> 1, gh->loc)) return 0;
    } else {
        if (!locset_insert(&a->hitset, gh->chr >> 1, gh->loc)) return 0;
    }
    uint32_t c = a->read_chain_index;
    a->xhits[c][w][a->x_cur_n_hit[c][w]++] = *gh;
    if (a->nlog == a->logcap) {
        a->logcap *= 2;
        a->log = (orc_loghit *)realloc(a->log, a->logcap * sizeof(orc_loghit));
    }
    orc_loghit lh = {*gh, (uint8_t)w, (uint8_t)c, (uint8_t)mode, 0};
    a->log[a->nlog++] = lh;
    if (a->x_cur_n_hit[0][w] + a->x_cur_n_hit[1][w] >= p->max_num_hits) {
        if (w == 0) return 1;
        else a->snp_thres = w - 1;
    }
    return 0;
}

/* align.cpp:348-410 */
static uint32_t gap_align(orc_aligner *a, uint32_t mode, uint32_t seed_pos) {
    const orc_param *p = a->p;
    uint32_t ghit_loc = a->_hit.loc, t, tt, ghit_loc1;
    int shift, shift1, clip;
    uint32_t i, j, gap_snp, gap_pos, m2, rl, *mmi1, *mmi2;
    if (a->snp_thres < 2) return 0;
    a->c.gap_calls++;
    const uint64_t *refseq = a->r->xref[a->ref_chain_index];
    const uint64_t *q = a->xseq[a->read_chain_index];
    if (mismatch_pattern0(a, q, refseq + ghit_loc / ORC_SEGLEN, (ghit_loc % ORC_SEGLEN) << 1) < seed_pos + p->seed_size) return 0;
    mmi1 = a->mm_index[0];
    for (tt = 1; tt <= p->gap * 2; tt++) {
        t = (tt + 1) / 2;
        shift = (1 - (int)(tt % 2) * 2) * (int)t;
        shift1 = shift * (shift < 0);
        if (a->snp_thres < 1 + t) break;
        ghit_loc1 = ghit_loc + (uint32_t)shift;
        mismatch_pattern1(a, q, refseq + ghit_loc1 / ORC_SEGLEN, tt, (ghit_loc1 % ORC_SEGLEN) << 1);
        rl = a->map_readlen - t - 1;
        mmi2 = a->mm_index[tt];
        for (i = 0; i < a->snp_thres - t; i++) {
            gap_pos = mmi1[i];
            gap_snp = 0;
            if (gap_pos < p->gap_edge || gap_pos >= rl) continue;
            for (j = 0; j < a->snp_thres - t - i; j++) {
                m2 = mmi2[j];
                if (m2 < p->gap_edge || m2 >= rl) continue;
                if ((int)gap_pos + (int)m2 - shift1 < (int)a->map_readlen) continue;
                gap_snp = i + j + t;
                clip = (int)gap_pos + (int)p->gap_edge - (int)a->map_readlen;
                clip -= shift1;
                if (clip > 0) gap_pos -= (uint32_t)clip;
                orc_hit gh = int2hit(a, a->_hit, shift, gap_pos);
                return add_hit(a, &gh, gap_snp, mode);
            }
        }
    }
    return 0;
}

/* align.cpp:228-317, WGBS branch 274-316 */
static void snp_align(orc_aligner *a, uint32_t mode) {
    const orc_param *p = a->p;
    const orc_ref *r = a->r;
    a->c.snp_calls++;
    for (a->read_chain_index = 0; a->read_chain_index < 2; a->read_chain_index++) {
        uint32_t c = a->read_chain_index;
        if (!a->xflag_chain[c]) continue;
        uint32_t modeindex = (uint32_t)a->seg_order[c][mode];
        for (uint32_t i = 0; i != p->index_interval; i++) {
            uint32_t seed = a->xseeds[c][modeindex][i];
            uint32_t m = r->n_tot[seed];
            a->c.seed_lookups++;
            if (m == 0 || m > p->max_kmer_num) continue;
            a->c.candidates += m;
            uint32_t mc = r->n_fwd[seed] - 1;
            uint32_t h = p->profile[modeindex][i] + a->xseed_start_array[c][modeindex] - i;
            uint32_t jj = orc_myrand((int)a->cur_index, p->randseed) % m;
            const uint32_t *loc0 = r->locs + r->off[seed];
            for (uint32_t j = 0; j != m; ++j, ++jj) {
                jj -= (jj >= m) * m;
                a->_hit.loc = loc0[jj] - h;
                a->ref_chain_index = (uint32_t)((int)mc - (int)jj) >> 31;
                count_mismatch(a, a->xseq[c], (a->_hit.loc % ORC_SEGLEN) << 1, r->xref[a->ref_chain_index] + a->_hit.loc / ORC_SEGLEN);
                if (a->tmp_snp <= a->snp_thres) {
                    orc_hit gh = int2hit(a, a->_hit, 0, 0);
                    if (add_hit(a, &gh, a->tmp_snp, mode)) return;
                }
                if (p->gap > 0)
                    if (gap_align(a, mode, h)) return;
            }
        }
    }
}

/* align.cpp:437-444 ClearHits + 446-466 RunAlign */
static void clear_hits(orc_aligner *a, const orc_read *rd) {
    for (int i = 0; i <= ORC_MAXSNPS; i++) a->x_cur_n_hit[0][i] = a->x_cur_n_hit[1][i] = 0;
    locset_clear(&a->hitset);
    locset_clear(&a->ghitset);
    a->nlog = 0;
    a->map_readlen = (uint32_t)strlen(rd->seq);
    a->end_element = (a->map_readlen - 1) / ORC_SEGLEN;
    a->end_offset = (ORC_SEGLEN - ((a->map_readlen - 1) % ORC_SEGLEN + 1)) << 1;
}

static uint32_t calc_seedseg_num(const orc_aligner *a) {
    int x = (int)((a->map_readlen - a->p->index_interval + 1) / a->p->seed_size);
    int y = (int)(a->read_max_snp_num + 1);
    return (uint32_t)(x < y ? x : y);
}

int orc_run_align(orc_aligner *a, const orc_read *rd) {
    a->cur_index = rd->index;
    clear_hits(a, rd);
    a->c.reads++;
    a->c.read_bytes += a->map_readlen;
    a->seedseg_num = calc_seedseg_num(a);
    convert_binary_seq(a, rd);
    a->snp_thres = a->read_max_snp_num;
    reorder_seed(a);
    for (uint32_t i = 0; i < a->seedseg_num; i++) {
        snp_align(a, i);
        for (uint32_t ii = 0; ii <= i; ii++)
            if (a->x_cur_n_hit[0][ii] || a->x_cur_n_hit[1][ii]) return 1;
    }
    for (uint32_t i = 0; i <= a->read_max_snp_num; i++)
        if (a->x_cur_n_hit[0][i] || a->x_cur_n_hit[1][i]) return 1;
    return 0;
}

/* ------------------------------------------------------------------ SAM text */

static void str_append(orc_str *s, const char *t, size_t n) {
    if (s->n + n + 1 > s->cap) {
        size_t cap = s->cap ? s->cap : 4096;
        while (s->n + n + 1 > cap) cap *= 2;
        s->s = (char *)realloc(s->s, cap);
        s->cap = cap;
    }
    memcpy(s->s + s->n, t, n);
    s->n += n;
    s->s[s->n] = 0;
}
static void str_printf(orc_str *s, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#include <stdarg.h>
static void str_printf(orc_str *s, const char *fmt, ...) {
    char stack[2048];
    va_list ap;
    va_start(ap, fmt);
    int n = vsnprintf(stack, sizeof stack, fmt, ap);
    va_end(ap);
    if (n < (int)sizeof stack) {
        str_append(s, stack, (size_t)n);
        return;
    }
    char *big = (char *)malloc((size_t)n + 1);
    va_start(ap, fmt);
    vsnprintf(big, (size_t)n + 1, fmt, ap);
    va_end(ap);
    str_append(s, big, (size_t)n);
    free(big);
}
void orc_str_free(orc_str *s) {
    free(s->s);
    s->s = NULL;
    s->n = s->cap = 0;
}

static const char chain_flag[2] = {'+', '-'};

static void fmt_cigar(char *cigar, size_t n, const orc_hit *hit, uint32_t map_readlen) {
    /* align.cpp:641-643 */
    if (hit->gap_size == 0) snprintf(cigar, n, "%uM", map_readlen);
    else if (hit->gap_size > 0) snprintf(cigar, n, "%dM%dD%dM", (int)hit->gap_pos, (int)hit->gap_size, (int)map_readlen - (int)hit->gap_pos);
    else snprintf(cigar, n, "%dM%dI%dM", (int)hit->gap_pos, -(int)hit->gap_size, (int)map_readlen - (int)hit->gap_pos + (int)hit->gap_size);
}

/* the XR:Z field, align.cpp:646-658 (same code in pairs.cpp:339-352; there the contig is
   (chr>>1)<<1 instead of chr&0xfffe) */
static void fmt_xr(const orc_aligner *a, const orc_hit *hit, uint32_t nbases, int mask16, orc_str *os) {
    const orc_ref *r = a->r;
    uint32_t contig = mask16 ? ((hit->chr & 0xfffeU) >> 1) : (hit->chr >> 1);
    uint64_t woff = ORC_REF_MARGIN;
    for (uint32_t c = 0; c < contig; c++) woff += r->nword[c];
    const uint64_t *s = r->xref[0] + woff;
    char mapseq[1024];
    int ptr = 0;
    for (uint32_t ii = 2; ii > 0; ii--) {
        if (hit->loc < ii) continue;
        uint32_t x = hit->loc - ii;
        mapseq[ptr++] = (char)(a->p->useful_nt[(s[x / ORC_SEGLEN] >> (ORC_SEGLEN * 2 - 2 - (x % ORC_SEGLEN) * 2)) & 0x3] + 32);
    }
    for (uint32_t ii = 0; ii < nbases + 2; ii++) {
        uint32_t x = hit->loc + ii;
        mapseq[ptr++] = a->p->useful_nt[(s[x / ORC_SEGLEN] >> (ORC_SEGLEN * 2 - 2 - (x % ORC_SEGLEN) * 2)) & 0x3];
    }
    mapseq[ptr] = 0;
    mapseq[ptr - 1] += 32;
    mapseq[ptr - 2] += 32;
    str_printf(os, "\tXR:Z:%s", mapseq);
}

/* align.cpp:616-669 */
static void s_out_hit(orc_aligner *a, const orc_read *rd, int chain, int n, uint32_t nsnps, const orc_hit *hit, orc_str *os) {
    const orc_param *p = a->p;
    uint32_t rev_seq = (uint32_t)chain ^ (hit->chr % 2);
    int flag = (int)(0x40 * rd->readset);
    if (n < 0) {
        if (!p->out_unmap) return;
        flag |= 0x204;
        str_printf(os, "%s\t%d\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n", rd->name, flag, rd->seq, rd->qual);
    } else if (n == 0) {
        if (!p->out_unmap) return;
        flag |= 0x4;
        str_printf(os, "%s\t%d\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n", rd->name, flag, rd->seq, rd->qual);
    } else {
        if (n != 1) flag |= 0x100;
        if (rev_seq && n) flag |= 0x010;
        char cigar[64];
        fmt_cigar(cigar, sizeof cigar, hit, a->map_readlen);
        str_printf(os, "%s\t%d\t%s\t%u\t255\t%s\t*\t0\t0\t%s\t%s\tNM:i:%d", rd->name, flag, a->r->name[hit->chr >> 1],
                   hit->loc + 1, cigar, a->outseq[rev_seq], a->outqual[rev_seq], (int)(uint8_t)nsnps);
        if (p->out_ref) fmt_xr(a, hit, a->map_readlen, 1, os);
        str_printf(os, "\tZS:Z:%c%c\n", chain_flag[hit->chr % 2], chain_flag[chain]);
    }
}

/* align.cpp:583-612 */
void orc_string_align(orc_aligner *a, const orc_read *rd, orc_str *os) {
    const orc_param *p = a->p;
    uint32_t ii, sum = 0, j = 0;
    uint32_t *nh = a->x_cur_n_hit[0], *nc = a->x_cur_n_hit[1];
    for (ii = 0; ii <= a->read_max_snp_num; ii++)
        if ((sum = nh[ii] + nc[ii]) > 0) break;
    if (sum == 0) {
        if (p->out_unmap) s_out_hit(a, rd, 0, 0, ii, &a->xhits[0][0][0], os);
    } else if (sum == 1) {
        ++a->n_aligned;
        ++a->n_unique;
        a->c.hit_records++;
        if (nh[ii]) s_out_hit(a, rd, 0, 1, ii, &a->xhits[0][ii][0], os);
        else s_out_hit(a, rd, 1, 1, ii, &a->xhits[1][ii][0], os);
    } else {
        ++a->n_multiple;
        if (p->report_repeat_hits == 1) {
            ++a->n_aligned;
            a->c.hit_records++;
            j = orc_myrand((int)rd->index, p->randseed) % sum;
            if (j < nh[ii]) s_out_hit(a, rd, 0, (int)sum, ii, &a->xhits[0][ii][j], os);
            else s_out_hit(a, rd, 1, (int)sum, ii, &a->xhits[1][ii][j - nh[ii]], os);
        } else if (p->report_repeat_hits == 2) {
            ++a->n_aligned;
            a->c.hit_records += sum;
            for (j = 0; j < nh[ii]; ++j) s_out_hit(a, rd, 0, (int)sum, ii, &a->xhits[0][ii][j], os);
            for (j = 0; j < nc[ii]; ++j) s_out_hit(a, rd, 1, (int)sum, ii, &a->xhits[1][ii][j], os);
        } else if (p->out_unmap) s_out_hit(a, rd, 0, 0, ii, &a->xhits[0][0][0], os);
    }
}

/* align.cpp:565-580, one iteration */
void orc_do_read(orc_aligner *a, orc_read *rd, orc_str *os) {
    if (orc_filter_read(a, rd)) {
        if (a->p->out_unmap) s_out_hit(a, rd, 0, -1, 0, &a->xhits[0][0][0], os);
    } else {
        orc_run_align(a, rd);
        orc_string_align(a, rd, os);
    }
}

/* main.cpp:586-597 */
void orc_sam_header(const orc_ref *r, const char *cmdline, orc_str *os) {
    str_printf(os, "@HD\tVN:1.0\n");
    for (uint32_t i = 0; i < r->ncontig; i++) str_printf(os, "@SQ\tSN:%s\tLN:%u\n", r->name[i], r->size[i]);
    str_printf(os, "@PG\tID:BASAL\tVN:%s\tCL:\"%s\"\n", "1.8.1", cmdline);
}

#include "basal_oracle_pe.inc"

/* ------------------------------------------------------------------ in-memory batch driver (bench/spot checks) */
#include <pthread.h>
#include <time.h>

typedef struct mt_job {
    const orc_param *p;
    const orc_ref *r;
    const uint8_t *bases;
    const uint32_t *seq_off, *index;
    const uint16_t *len;
    const uint8_t *max_snp;
    uint32_t begin, end;
    orc_best *out;
    orc_counters c;
} mt_job;

static void *mt_worker(void *arg) {
    mt_job *j = (mt_job *)arg;
    orc_aligner *a = orc_aligner_new(j->p, j->r);
    char seq[ORC_FIXSIZE + 8], qual[ORC_FIXSIZE + 8], name[4] = "r";
    for (uint32_t i = j->begin; i < j->end; i++) {
        uint32_t L = j->len[i];
        orc_best *o = &j->out[i];
        memset(o, 0, sizeof *o);
        o->best_level = 0xFF;
        if (L == 0 || L > ORC_FIXSIZE) continue;
        memcpy(seq, j->bases + j->seq_off[i], L);
        seq[L] = 0;
        memset(qual, 'I', L);
        qual[L] = 0;
        orc_read rd = {j->index[i], 0, name, seq, qual};
        /* the reads are post-FilterReads: take read_max_snp_num as given */
        a->read_max_snp_num = j->max_snp[i];
        a->raw_readlen = L;
        orc_run_align(a, &rd);
        uint32_t ii, sum = 0;
        for (ii = 0; ii <= a->read_max_snp_num; ii++)
            if ((sum = a->x_cur_n_hit[0][ii] + a->x_cur_n_hit[1][ii]) > 0) break;
        if (!sum) continue;
        uint32_t nh = a->x_cur_n_hit[0][ii], jj = sum == 1 ? 0 : orc_myrand((int)rd.index, j->p->randseed) % sum;
        const orc_hit *h = jj < nh ? &a->xhits[0][ii][jj] : &a->xhits[1][ii][jj - nh];
        o->best_level = ii;
        o->n_hit = nh;
        o->n_chit = a->x_cur_n_hit[1][ii];
        o->chr = h->chr;
        o->loc = h->loc;
        o->gap_size = h->gap_size;
        o->gap_pos = h->gap_pos;
        o->chain = jj < nh ? 0 : 1;
        a->c.hit_records++;
    }
    j->c = a->c;
    orc_aligner_free(a);
    return NULL;
}

int orc_align_batch_mt(const orc_param *p, const orc_ref *r, const uint8_t *bases, const uint32_t *seq_off, const uint16_t *len,
                       const uint32_t *index, const uint8_t *max_snp, uint32_t n, int threads, orc_best *out, orc_counters *counters,
                       double *seconds) {
    if (threads < 1) threads = 1;
    if ((uint32_t)threads > n && n) threads = (int)n;
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    mt_job *jobs = (mt_job *)calloc((size_t)threads, sizeof(mt_job));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        mt_job *j = &jobs[t];
        j->p = p; j->r = r; j->bases = bases; j->seq_off = seq_off; j->len = len; j->index = index; j->max_snp = max_snp; j->out = out;
        j->begin = (uint32_t)((uint64_t)n * t / threads);
        j->end = (uint32_t)((uint64_t)n * (t + 1) / threads);
        pthread_create(&th[t], NULL, mt_worker, j);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (seconds) *seconds = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    if (counters) {
        memset(counters, 0, sizeof *counters);
        for (int t = 0; t < threads; t++) {
            const orc_counters *c = &jobs[t].c;
            counters->reads += c->reads; counters->hdr_lookups += c->hdr_lookups; counters->seed_lookups += c->seed_lookups;
            counters->candidates += c->candidates; counters->ref_words += c->ref_words; counters->read_bytes += c->read_bytes;
            counters->hit_records += c->hit_records; counters->snp_calls += c->snp_calls; counters->gap_calls += c->gap_calls;
        }
    }
    free(th);
    free(jobs);
    return 0;
}
