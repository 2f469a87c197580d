// basal_host.cpp -- host side of libbasal_amd.so: everything either side of the GPU core.
//
//   parameters and -M code tables     Param::Param / SetSeedSize / SetAlign   (param.cpp:7-115,163-263)
//   FASTA -> 2-bit forward + RC       RefSeq::Run_ConvertBinseq                (refbase.cpp:17-252)
//   seed index (CPU build)            RefSeq::CreateIndex                      (refbase.cpp:261-439)
//   read QC                           SingleAlign::FilterReads                 (align.cpp:40-76,418-435,548-563)
//   SAM text                          StringAlign / s_OutHit                   (align.cpp:583-669)
//
// The index is kept flat (CSR offsets + forward counts + one location array) because that is the
// layout the GPU gathers from; the CPU build is a two-pass counting sort split over threads by
// k-mer range, which reproduces the reference's per-k-mer order (forward ascending, then RC
// ascending) without its 16-byte header table.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/basal_core.h"
#include "basal_bits.h"
#include "basal_internal.h"

using namespace basal;

// ------------------------------------------------------------------------------------------ params

static const char kNt[5] = {'A', 'C', 'G', 'T', '-'};
static const char kRevNt[5] = {'T', 'G', 'C', 'A', '-'};

static int base_rank(int c) {  // A C G T -> 0..3 (alphabet0, param.cpp:119-128)
    switch (c) {
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 0;
    }
}

extern "C" int basal_host_params_set_seed_size(basal_params *p, int n) {
    if (n > 16 || n < 10) {
        set_error("seed size must be between 10 and 16");
        return BASAL_EINVAL;
    }
    p->seed_size = (uint32_t)n;
    p->min_read_size = p->seed_size + p->index_interval - 1;  // param.cpp:112, with -I as parsed so far
    return BASAL_OK;
}

extern "C" void basal_host_params_defaults(basal_params *p) {
    memset(p, 0, sizeof *p);
    p->max_ns = 5;
    p->zero_qual = '!';
    p->default_qual = 40;
    p->min_insert = 28;
    p->max_insert = 1000;
    p->index_interval = 4;
    p->seed_size = 16;
    p->min_read_size = 16;  // param.cpp:34 overrides what SetSeedSize computed
    p->max_snp_num = 110;
    p->max_num_hits = 100;
    p->max_kmer_ratio = 5e-7f;
    p->report_repeat_hits = 1;
    p->gap_edge = 6;
    p->max_readlen = BASAL_MAXREADLEN;
    p->refnt = 'C';
    memcpy(p->useful_nt, "ACGTacgt", 9);
    for (const char *q = "ACGTacgt"; *q; q++) p->reg_alphabet[(unsigned char)*q] = 3;
}

extern "C" void basal_host_params_set_v(basal_params *p, double v) {
    if (v < 1.0) {
        p->max_snp_num = (uint32_t)((int)(v * 100 + 0.5) + 100);
        if (p->max_snp_num == 100) p->max_snp_num = 0;
    } else {
        p->max_snp_num = (uint32_t)(int)(v + 0.5);
        if (p->max_snp_num > BASAL_MAXSNPS) p->max_snp_num = BASAL_MAXSNPS;
    }
}

extern "C" int basal_host_params_set_align(basal_params *p, const char *rule) {
    std::string r = rule ? rule : "";
    if (r.size() < 2 || r[1] != ':') {
        set_error("invalid -M, ref base(one letter in A/C/G/T) should be assigned first before :");
        return BASAL_EINVAL;
    }
    char refnt = (char)toupper((unsigned char)r[0]);
    if (!p->reg_alphabet[(unsigned char)refnt]) {
        set_error(std::string("invalid -M, ref base ") + r[0] + " not in A/C/G/T");
        return BASAL_EINVAL;
    }
    std::string tos;
    for (size_t i = 2; i < r.size(); i++) {
        char t = (char)toupper((unsigned char)r[i]);
        if (t == refnt) {
            set_error(std::string("invalid -M, read base ") + r[i] + " should not be equal to ref base " + refnt);
            return BASAL_EINVAL;
        }
        if (!memchr(kNt, t, 5)) {
            set_error(std::string("invalid -M, read base ") + r[i] + " not in A/C/G/T/-");
            return BASAL_EINVAL;
        }
        if (tos.find(t) == std::string::npos && tos.size() < 5) tos.push_back(t);
    }
    p->refnt = refnt;
    memset(p->readnts, ' ', 5);
    memcpy(p->readnts, tos.data(), tos.size());
    p->readnt_cnt = (uint8_t)tos.size();
    const bool one_way = tos.size() == 1 && tos[0] != '-';
    p->new_rule = one_way ? 0 : 1;

    // convert-to plane: 01 for every convert-to base, 11 for the other ACGT, 00 otherwise
    memcpy(p->alphabet_mread, p->reg_alphabet, 256);
    memcpy(p->rev_alphabet_mread, p->reg_alphabet, 256);
    for (char t : tos) {
        p->alphabet_mread[(unsigned char)t] = 1;
        p->alphabet_mread[(unsigned char)tolower((unsigned char)t)] = 1;
        if (t == '-') continue;
        char rc = kRevNt[base_rank(t)];
        p->rev_alphabet_mread[(unsigned char)rc] = 1;
        p->rev_alphabet_mread[(unsigned char)tolower((unsigned char)rc)] = 1;
    }
    // base codes: convert-from = 01; a single convert-to base = 11; the rest take 0,2,3 in ACGT order
    int code[4] = {-1, -1, -1, -1};
    code[base_rank(refnt)] = 1;
    if (one_way) code[base_rank(tos[0])] = 3;
    const int spare[3] = {0, 2, 3};
    for (int b = 0, j = 0; b < 4; b++)
        if (code[b] < 0) code[b] = spare[j++];
    memset(p->alphabet, 0, 256);
    memset(p->rev_alphabet, 0, 256);
    for (int b = 0; b < 4; b++) {
        p->alphabet[(unsigned char)kNt[b]] = p->alphabet[(unsigned char)tolower((unsigned char)kNt[b])] = (uint8_t)code[b];
        p->rev_alphabet[(unsigned char)kNt[b]] = p->rev_alphabet[(unsigned char)tolower((unsigned char)kNt[b])] = (uint8_t)code[3 - b];
        p->useful_nt[code[b]] = kNt[b];
        p->useful_nt[code[b] + 4] = (char)tolower((unsigned char)kNt[b]);
    }
    p->useful_nt[8] = 0;
    return BASAL_OK;
}

// ------------------------------------------------------------------------------------------ reference

struct basal_ref {
    std::vector<std::string> name;
    std::vector<uint32_t> size, rc_offset, nword, anchor;
    std::vector<uint64_t> words[2];   // xref[0], xref[1] with the 400-word margins
    std::vector<uint32_t> blocks;     // (id, begin, end) triples sorted by (id, begin)
    std::vector<uint64_t> word_base;  // first word of each contig inside words[]
    uint64_t sum_length = 0;
    // index
    uint32_t total_kmers = 0, max_kmer_num = 0;
    std::vector<uint32_t> kmer_off, kmer_nfwd, locs;
};

static bool read_all(const char *path, std::string &out) {
    gzFile f = gzopen(path, "rb");
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    {   // a plain file's size is known: one allocation instead of a 3 GB string growing by doubling
        struct stat st;
        if (stat(path, &st) == 0 && st.st_size > 0) out.reserve((size_t)st.st_size + 1);
    }
    std::vector<char> buf(1 << 22);
    int got;
    while ((got = gzread(f, buf.data(), (unsigned)buf.size())) > 0) out.append(buf.data(), (size_t)got);
    gzclose(f);
    return true;
}

static inline bool ws(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }

// the sequence tokens of one record, from `pos` to the next token that starts with '>' (or the end): appended to sq; returns where it stopped
static size_t append_sequence(const char *buf, size_t pos, size_t len, std::string &sq) {
    for (;;) {
        while (pos < len && ws((unsigned char)buf[pos])) pos++;
        if (pos >= len || buf[pos] == '>') break;
        // a whole line at once when it is one token (what FASTA lines are): memchr for its end, then a check for white space inside
        const char *nl = (const char *)memchr(buf + pos, '\n', len - pos);
        const size_t le = nl ? (size_t)(nl - buf) : len;
        bool plain = true;
        for (size_t i = pos; i < le; i++)
            if ((unsigned char)buf[i] <= ' ') { plain = false; break; }
        if (plain) {
            sq.append(buf + pos, le - pos);
            pos = le;
            continue;
        }
        size_t b = pos;
        while (pos < len && !ws((unsigned char)buf[pos])) pos++;
        sq.append(buf + b, pos - b);
    }
    return pos;
}

// A FASTA file laid out the usual way -- it starts with '>', every record's '>' is the first byte of a line, any other '>' sits inside a
// header line, a name follows each '>' at once -- splits into records at those line starts, and the records' sequences are then gathered
// in parallel (RefSeq::LoadNextSeq's token reader would read them the same way). Returns false for anything else: the caller then
// walks the file token by token.
static bool split_records(const char *buf, size_t len, std::vector<size_t> &starts) {
    if (len == 0 || buf[0] != '>') return false;
    size_t hdr_end = 0;  // end of the current header line
    for (size_t p = 0; p < len;) {
        const char *g = (const char *)memchr(buf + p, '>', len - p);
        if (!g) break;
        const size_t q = (size_t)(g - buf);
        if (q == 0 || buf[q - 1] == '\n') {
            if (q + 1 >= len || ws((unsigned char)buf[q + 1])) return false;
            starts.push_back(q);
            const char *nl = (const char *)memchr(buf + q, '\n', len - q);
            hdr_end = nl ? (size_t)(nl - buf) : len;
            p = hdr_end;
        } else if (q < hdr_end) p = q + 1;  // inside a header line (unreachable: the scan resumes behind it; kept for clarity)
        else return false;                  // a '>' in the middle of a sequence line: leave it to the token reader
    }
    return !starts.empty();
}

extern "C" int basal_host_ref_load_mem(const basal_params *p, const char *buf, size_t len, basal_ref_t **out) {
    if (!p || !buf || !out) { set_error("ref_load: null argument"); return BASAL_EINVAL; }
    basal_ref *r = new basal_ref();
    std::vector<std::string> seqs;
    size_t pos = 0;
    std::vector<size_t> starts;
    if (len >= (1u << 20) && split_records(buf, len, starts)) {
        const size_t nr = starts.size();
        std::vector<std::string> names(nr);
        seqs.resize(nr);
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= nr) break;
                const size_t end = k + 1 < nr ? starts[k + 1] : len;
                size_t q = starts[k] + 1, b = q;
                while (q < end && !ws((unsigned char)buf[q])) q++;
                names[k].assign(buf + b, q - b);
                while (q < end && buf[q] != '\n') q++;
                seqs[k].reserve(end - q);
                append_sequence(buf, q, end, seqs[k]);
            }
        };
        unsigned nt = std::thread::hardware_concurrency();
        nt = nt < 1 ? 1 : nt > 16 ? 16 : nt;
        std::vector<std::thread> th;
        for (unsigned k = 1; k < nt && k < nr; k++) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
        size_t keep = 0;
        while (keep < nr && !seqs[keep].empty()) keep++;  // a zero-length record ends loading (refbase.cpp:192)
        seqs.resize(keep);
        names.resize(keep);
        r->name = std::move(names);
        pos = len;
    }
    // iostream-token semantics of RefSeq::LoadNextSeq (refbase.cpp:17-38)
    for (;;) {
        while (pos < len && ws((unsigned char)buf[pos])) pos++;
        if (pos >= len) break;
        pos++;  // the '>' (whatever it is)
        while (pos < len && ws((unsigned char)buf[pos])) pos++;
        size_t b = pos;
        while (pos < len && !ws((unsigned char)buf[pos])) pos++;
        std::string nm(buf + b, pos - b);
        while (pos < len && buf[pos] != '\n') pos++;
        if (pos < len) pos++;
        std::string sq;
        pos = append_sequence(buf, pos, len, sq);
        if (sq.empty()) break;  // a zero-length record ends loading (refbase.cpp:192)
        r->name.push_back(nm);
        seqs.push_back(std::move(sq));
    }
    const size_t nc = seqs.size();
    if (nc == 0) { delete r; set_error("ref_load: no sequence in FASTA"); return BASAL_EIO; }
    uint64_t s = 0;
    r->anchor.push_back(BASAL_REF_MARGIN * 32);
    for (size_t i = 0; i < nc; i++) {
        uint64_t L = seqs[i].size();
        if (L > 0xFFFFFF00ull) { delete r; set_error("ref_load: contig longer than 2^32"); return BASAL_EINVAL; }
        uint32_t n = (uint32_t)((L + 31) / 32 + 2);  // BINSEQPAD
        r->size.push_back((uint32_t)L);
        r->nword.push_back(n);
        r->rc_offset.push_back(n * 32);
        r->word_base.push_back(BASAL_REF_MARGIN + s);
        s += n;
        if ((s + BASAL_REF_MARGIN) * 32 > 0xFFFFFFFFull) { delete r; set_error("ref_load: reference exceeds the 32-bit coordinate space of the index"); return BASAL_EINVAL; }
        r->anchor.push_back((uint32_t)((s + BASAL_REF_MARGIN) * 32));
        r->sum_length += L;
    }
    const uint64_t nw = s + 2 * BASAL_REF_MARGIN;
    r->words[0].assign(nw, 0);
    r->words[1].assign(nw, 0);
    // Packing is per (contig, slice of 2^20 bases): a slice starts on a word boundary of the forward strand, and -- the slot being a
    // whole number of words -- of the reverse strand too, so slices write disjoint words. The unmasked blocks are found per contig.
    {
        struct Task { uint32_t ci, b, e; };
        std::vector<Task> tasks;
        for (size_t ci = 0; ci < nc; ci++)
            for (uint32_t b = 0; b < r->size[ci]; b += 1u << 20) tasks.push_back({(uint32_t)ci, b, std::min<uint32_t>(r->size[ci], b + (1u << 20))});
        std::vector<std::vector<uint32_t>> cblocks(nc);
        std::atomic<size_t> next_task{0}, next_contig{0};
        auto work = [&]() {
            for (;;) {
                const size_t k = next_task.fetch_add(1);
                if (k >= tasks.size()) break;
                const Task t = tasks[k];
                const std::string &sq = seqs[t.ci];
                const uint32_t tot = r->nword[t.ci] * 32;
                uint64_t *fw = r->words[0].data() + r->word_base[t.ci], *rc = r->words[1].data() + r->word_base[t.ci];
                // forward strand, MSB first; the tail is 'N' (code alphabet['N'] = 0)
                for (uint32_t i = t.b; i < t.e; i++) fw[i >> 5] |= (uint64_t)p->alphabet[(unsigned char)sq[i]] << (62 - 2 * (i & 31));
                // reverse complement of the padded sequence: base j of the RC array is the complement of base tot-1-j
                for (uint32_t i = t.b; i < t.e; i++) {
                    const uint32_t j = tot - 1 - i;
                    rc[j >> 5] |= (uint64_t)p->rev_alphabet[(unsigned char)sq[i]] << (62 - 2 * (j & 31));
                }
            }
            for (;;) {
                const size_t ci = next_contig.fetch_add(1);
                if (ci >= nc) break;
                const std::string &sq = seqs[ci];
                const uint32_t L = r->size[ci], tot = r->nword[ci] * 32;
                std::vector<uint32_t> &bl = cblocks[ci];
                // blocks: maximal runs starting at an ACGT and ending before the next N/X, kept if >= 16 long
                // (UnmaskRegion, refbase.cpp:103-128; its "merge" branch never fires)
                uint32_t e = 0;
                while (e < L) {
                    uint32_t b = e;
                    while (b < L && !p->reg_alphabet[(unsigned char)sq[b]]) b++;
                    if (b >= L) break;
                    e = b;
                    while (e < L) {
                        char ch = sq[e];
                        if (ch == 'N' || ch == 'X' || ch == 'n' || ch == 'x') break;
                        e++;
                    }
                    if (e - b < 16) continue;
                    bl.insert(bl.end(), {(uint32_t)(2 * ci), b, e});
                    bl.insert(bl.end(), {(uint32_t)(2 * ci + 1), tot - e, tot - b});
                }
            }
        };
        unsigned nt = std::thread::hardware_concurrency();
        nt = nt < 1 ? 1 : nt > 16 ? 16 : nt;
        if (tasks.size() < 4) nt = 1;
        std::vector<std::thread> th;
        for (unsigned k = 1; k < nt; k++) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
        for (size_t ci = 0; ci < nc; ci++) r->blocks.insert(r->blocks.end(), cblocks[ci].begin(), cblocks[ci].end());
    }
    // sort by (id, begin) (refbase.cpp:184,219)
    {
        size_t nb = r->blocks.size() / 3;
        std::vector<uint32_t> idx(nb);
        for (size_t i = 0; i < nb; i++) idx[i] = (uint32_t)i;
        const uint32_t *B = r->blocks.data();
        std::sort(idx.begin(), idx.end(), [B](uint32_t a, uint32_t b) {
            if (B[3 * a] != B[3 * b]) return B[3 * a] < B[3 * b];
            return B[3 * a + 1] < B[3 * b + 1];
        });
        std::vector<uint32_t> nbk(r->blocks.size());
        for (size_t i = 0; i < nb; i++) memcpy(&nbk[3 * i], &B[3 * idx[i]], 12);
        r->blocks.swap(nbk);
    }
    *out = r;
    return BASAL_OK;
}

extern "C" int basal_host_ref_load(const basal_params *p, const char *path, basal_ref_t **out) {
    if (path) {  // a plain (not gzip) file is parsed where the page cache has it, without a copy
        int fd = open(path, O_RDONLY);
        if (fd >= 0) {
            struct stat st;
            unsigned char magic[2] = {0, 0};
            if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= (1 << 20) && pread(fd, magic, 2, 0) == 2 && !(magic[0] == 0x1f && magic[1] == 0x8b)) {
                void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) {
                    madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
                    const int rc = basal_host_ref_load_mem(p, (const char *)m, (size_t)st.st_size, out);
                    munmap(m, (size_t)st.st_size);
                    close(fd);
                    return rc;
                }
            }
            close(fd);
        }
    }
    std::string buf;
    if (!path || !read_all(path, buf)) {
        set_error(std::string("failed to open reference file (check -d option): ") + (path ? path : "(null)"));
        return BASAL_EIO;
    }
    return basal_host_ref_load_mem(p, buf.data(), buf.size(), out);
}

extern "C" void basal_host_ref_free(basal_ref_t *r) { delete r; }
extern "C" uint32_t basal_host_ref_ncontig(const basal_ref_t *r) { return (uint32_t)r->name.size(); }
extern "C" const char *basal_host_ref_name(const basal_ref_t *r, uint32_t c) { return c < r->name.size() ? r->name[c].c_str() : ""; }
extern "C" const uint32_t *basal_host_ref_sizes(const basal_ref_t *r) { return r->size.data(); }
extern "C" const uint32_t *basal_host_ref_rc_offsets(const basal_ref_t *r) { return r->rc_offset.data(); }
extern "C" const uint32_t *basal_host_ref_anchors(const basal_ref_t *r) { return r->anchor.data(); }
extern "C" uint64_t basal_host_ref_nwords(const basal_ref_t *r) { return r->words[0].size(); }
extern "C" const uint64_t *basal_host_ref_words(const basal_ref_t *r, int strand) { return r->words[strand ? 1 : 0].data(); }
extern "C" uint64_t basal_host_ref_nblocks(const basal_ref_t *r) { return r->blocks.size() / 3; }
extern "C" const uint32_t *basal_host_ref_blocks(const basal_ref_t *r) { return r->blocks.data(); }
extern "C" uint32_t basal_host_ref_total_kmers(const basal_ref_t *r) { return r->total_kmers; }
extern "C" const uint32_t *basal_host_ref_kmer_off(const basal_ref_t *r) { return r->kmer_off.data(); }
extern "C" const uint32_t *basal_host_ref_kmer_nfwd(const basal_ref_t *r) { return r->kmer_nfwd.data(); }
extern "C" const uint32_t *basal_host_ref_locs(const basal_ref_t *r) { return r->locs.data(); }
extern "C" uint64_t basal_host_ref_nlocs(const basal_ref_t *r) { return r->locs.size(); }
extern "C" uint32_t basal_host_ref_max_kmer_num(const basal_ref_t *r) { return r->max_kmer_num; }

// the 3-letter hash of the k-mer at base `pos` of a packed strand (s_MakeSeed_1, refbase.cpp:254-255)
static inline uint32_t seed_at(const uint64_t *m, uint32_t pos, uint32_t K) {
    const uint64_t *w = m + (pos >> 5);
    uint32_t a = (pos & 31) * 2;
    uint64_t v = a ? (w[0] << a) | (w[1] >> (64 - a)) : w[0];
    return XT((uint32_t)(v >> (64 - 2 * K)));
}

// The over-represented k-mer cut-off (refbase.cpp:362-363): element [(u32)(total*(1-ratio)) - 1] of
// the counts after sorting all but the LAST k-mer id ascending; the product is single precision.
namespace basal {
uint32_t kmer_cutoff_index(uint32_t total_kmers, float ratio) {
    volatile float one_minus = 1 - ratio;
    volatile float prod = (float)total_kmers * one_minus;
    return (uint32_t)prod - 1;
}
}  // namespace basal

extern "C" int basal_host_ref_build_index(basal_ref_t *r, const basal_params *p, int threads) {
    if (!r || !p) { set_error("build_index: null argument"); return BASAL_EINVAL; }
    const uint32_t K = p->seed_size, I = p->index_interval;
    uint32_t total = 1;
    for (uint32_t i = 0; i < K; i++) total *= 3;
    r->total_kmers = total;
    if (threads < 1) threads = 1;
    const size_t nb = r->blocks.size() / 3;
    const uint32_t *B = r->blocks.data();
    // pass 1: counts per k-mer and strand. Threads own disjoint k-mer ranges and each scans every
    // block (hash work is repeated, memory writes are not shared) -- deterministic and lock-free.
    std::vector<uint32_t> cf(total, 0), cr(total, 0);
    auto scan = [&](int tid, int pass, std::vector<uint32_t> *curf, std::vector<uint32_t> *curr) {
        uint32_t lo = (uint32_t)((uint64_t)total * tid / threads), hi = (uint32_t)((uint64_t)total * (tid + 1) / threads);
        for (size_t b = 0; b < nb; b++) {
            uint32_t id = B[3 * b], beg = B[3 * b + 1], end = B[3 * b + 2];
            const uint64_t *m = r->words[id & 1].data() + r->word_base[id >> 1];
            uint32_t i2 = ((end - K) / I) * I;
            uint32_t anchor = r->anchor[id >> 1];
            for (uint32_t i = (beg / I) * I; i <= i2; i += I) {
                uint32_t sd = seed_at(m, i, K);
                if (sd < lo || sd >= hi) continue;
                if (pass == 0) {
                    if (id & 1) cr[sd]++; else cf[sd]++;
                } else {
                    if (id & 1) r->locs[(size_t)r->kmer_off[sd] + cf[sd] + (*curr)[sd]++] = anchor + i;
                    else r->locs[(size_t)r->kmer_off[sd] + (*curf)[sd]++] = anchor + i;
                }
            }
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) th.emplace_back(scan, t, 0, nullptr, nullptr);
        for (auto &t : th) t.join();
    }
    r->kmer_off.assign((size_t)total + 1, 0);
    uint64_t acc = 0;
    for (uint32_t k = 0; k < total; k++) {
        r->kmer_off[k] = (uint32_t)acc;
        acc += (uint64_t)cf[k] + cr[k];
    }
    if (acc >= 0xFFFFFFFFull) { set_error("build_index: more than 2^32-1 index entries"); return BASAL_EINVAL; }
    r->kmer_off[total] = (uint32_t)acc;
    r->locs.assign(acc, 0);
    {
        std::vector<uint32_t> curf(total, 0), curr(total, 0);
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) th.emplace_back(scan, t, 1, &curf, &curr);
        for (auto &t : th) t.join();
    }
    r->kmer_nfwd.swap(cf);
    // cut-off: order statistic over ids 0..total-2 via a histogram of counts
    uint32_t idx = kmer_cutoff_index(total, p->max_kmer_ratio);
    if (idx >= total - 1) r->max_kmer_num = r->kmer_off[total] - r->kmer_off[total - 1];
    else {
        uint32_t mx = 0;
        for (uint32_t k = 0; k + 1 < total; k++) mx = std::max(mx, r->kmer_off[k + 1] - r->kmer_off[k]);
        std::vector<uint64_t> hist((size_t)mx + 1, 0);
        for (uint32_t k = 0; k + 1 < total; k++) hist[r->kmer_off[k + 1] - r->kmer_off[k]]++;
        uint64_t run = 0;
        uint32_t v = 0;
        for (; v <= mx; v++) {
            run += hist[v];
            if (run > idx) break;
        }
        r->max_kmer_num = v;
    }
    return BASAL_OK;
}

extern "C" int basal_host_ref_upload(const basal_ref_t *r, basal_core_t *c, int build_on_gpu, uint32_t *max_kmer_num) {
    if (!r || !c) { set_error("ref_upload: null argument"); return BASAL_EINVAL; }
    int rc = basal_core_set_reference(c, r->words[0].data(), r->words[1].data(), r->words[0].size(), r->anchor.data(), r->size.data(),
                                      r->rc_offset.data(), (uint32_t)r->name.size());
    if (rc) return rc;
    if (build_on_gpu) {
        uint32_t mk = 0;
        rc = basal_core_build_index(c, r->blocks.data(), r->blocks.size() / 3, &mk);
        if (max_kmer_num) *max_kmer_num = mk;
        return rc;
    }
    if (r->kmer_off.empty()) { set_error("ref_upload: CPU index not built (basal_host_ref_build_index)"); return BASAL_ESTATE; }
    if (max_kmer_num) *max_kmer_num = r->max_kmer_num;
    return basal_core_set_index(c, r->kmer_off.data(), r->kmer_nfwd.data(), r->locs.data(), r->locs.size(), r->max_kmer_num);
}

// ------------------------------------------------------------------------------------------ read QC

extern "C" int basal_host_filter_read(const basal_params *p, char *seq, char *qual, uint32_t *read_max_snp_num) {
    size_t L = strlen(seq);
    uint32_t x;
    if (p->max_snp_num < 100) x = p->max_snp_num;
    else x = (uint32_t)((p->max_snp_num - 100) / 100.0 * L + 0.5);
    if (p->gap > 0) x = x + 1 + p->gap;
    if (x > BASAL_MAXSNPS) x = BASAL_MAXSNPS;
    const uint32_t raw_len = (uint32_t)L;
    // 3' adapter (TrimAdapter, align.cpp:418-435): <=4 mismatches in the first 15 adapter bases, 1 per 5 compared
    bool cut = false;
    for (uint32_t a = 0; a < p->n_adapter && !cut && L >= 4; a++) {
        size_t al = strlen(p->adapter[a]);
        for (uint32_t pos = p->seed_size + p->index_interval - 1; (size_t)pos < L - 4; pos++) {
            uint32_t mis = 0, k = 0;
            for (; k < al && k < 15 && pos + k < L; k++)
                if ((mis += (p->adapter[a][k] != seq[pos + k])) > 4) break;
            if (k >= mis * 5 && k > 3) {
                seq[pos] = 0;
                if (strlen(qual) > pos) qual[pos] = 0;
                cut = true;
                break;
            }
        }
    }
    // quality (TrimLowQual, align.cpp:51-76)
    L = strlen(seq);
    if (L != strlen(qual)) {
        memset(qual, p->zero_qual + p->default_qual, L);
        qual[L] = 0;
    }
    uint8_t thres = (uint8_t)(p->zero_qual + p->trim_qual_threshold);
    if (p->zero_qual != '!') {
        for (char *q = qual; *q; q++) *q -= (p->zero_qual - '!');
        thres -= (p->zero_qual - '!');
    }
    if (p->trim_qual_threshold != 0) {
        uint32_t i = (uint32_t)L;
        while (i > 0 && !((uint8_t)qual[i - 1] > thres)) i--;
        if (i < p->seed_size + p->index_interval - 1) return 1;
        qual[i] = 0;
        seq[i] = 0;
        L = i;
    }
    if (L < p->min_read_size) return 1;
    uint32_t ns = 0;
    for (size_t i = 0; i < L; i++) ns += !p->reg_alphabet[(unsigned char)seq[i]];
    if (ns > p->max_ns) return 1;
    *read_max_snp_num = (x + 1) * ((uint32_t)L - 1) / raw_len;
    return 0;
}

// ------------------------------------------------------------------------------------------ inherited state

struct basal_stale_tracker {
    basal_params p;
    struct Owner { uint32_t npos; std::string seq; };
    struct Slot {
        uint32_t last_def = BASAL_STALE_CARRY;  // most recent read of this batch that defined the start offset
        std::vector<Owner> stack;               // most recent on top; npos strictly increasing towards the bottom
    } slot[2];
};

extern "C" basal_stale_tracker_t *basal_host_stale_new(const basal_params *p) {
    basal_stale_tracker *t = new basal_stale_tracker();
    t->p = *p;
    return t;
}
extern "C" void basal_host_stale_free(basal_stale_tracker_t *t) { delete t; }
extern "C" void basal_host_stale_begin_batch(basal_stale_tracker_t *t) { t->slot[0].last_def = t->slot[1].last_def = BASAL_STALE_CARRY; }

// seed (xseed_array) and N flag (xseedreg_array) of `seq` at read offset pos on chain c (align.cpp:92-100)
static uint32_t host_seed_at(const basal_params &p, const std::string &seq, uint32_t pos, int c) {
    const uint32_t K = p.seed_size, L = (uint32_t)seq.size();
    uint32_t s = 0;
    bool n = false;
    for (uint32_t t = 0; t < K; t++) {
        unsigned char ch = (unsigned char)(c ? seq[L - 1 - (pos + t)] : seq[pos + t]);
        s = (s << 2) | (c ? p.rev_alphabet[ch] : p.alphabet[ch]);
        n |= !p.reg_alphabet[ch];
    }
    return XT(s) | (n ? 0x80000000u : 0);
}

extern "C" int basal_host_stale_visit(basal_stale_tracker_t *t, const char *seq, uint32_t len, uint32_t readset, int qc_failed,
                                      uint32_t read_number_in_batch, basal_stale *out) {
    if (qc_failed || len == 0) return 0;  // FilterReads failed: RunAlign never ran, nothing was written
    const basal_params &p = t->p;
    basal_stale_tracker::Slot &S = t->slot[readset == 2 ? 1 : 0];
    const uint32_t K = p.seed_size, I = p.index_interval;
    const uint32_t npos = len >= K ? len - K + 1 : 0;
    const bool flag[2] = {(p.chains == 1) || ((p.chains <= 1) == (readset < 2)), (p.chains == 1) || ((p.chains <= 1) == (readset == 2))};
    int stale = 0;
    if ((len - I + 1) % K == 0) {
        stale = 1;
        out->src = S.last_def;
        for (int c = 0; c < 2; c++)
            for (uint32_t j = 0; j < 15; j++) {
                uint32_t pos = npos + j, v = 0;  // never-written slots hold 0 (a fresh SingleAlign object)
                if (flag[c])
                    for (size_t k = S.stack.size(); k-- > 0;)
                        if (S.stack[k].npos > pos) { v = host_seed_at(p, S.stack[k].seq, pos, c); break; }
                out->overlay[c][j] = v;
            }
    } else S.last_def = read_number_in_batch;
    std::string keep;  // (the storage of a popped entry is used again: one entry is popped and one pushed per read of a fixed-length file)
    while (!S.stack.empty() && S.stack.back().npos <= npos) { keep.swap(S.stack.back().seq); S.stack.pop_back(); }
    keep.assign(seq, len);
    S.stack.push_back({npos, std::move(keep)});
    return stale;
}

// ------------------------------------------------------------------------------------------ SAM

namespace {
struct Out {
    char *p;
    size_t cap, n;
    bool ok;
    void put(const char *s, size_t l) {
        if (!ok || n + l > cap) { ok = false; return; }
        memcpy(p + n, s, l);
        n += l;
    }
    void str(const char *s) { put(s, strlen(s)); }
    void ch(char c) { put(&c, 1); }
    void num(long long v) {
        char b[24];
        int l = snprintf(b, sizeof b, "%lld", v);
        put(b, (size_t)l);
    }
};

char comp_char(char c) {  // rev_char, param.cpp:146-156
    switch (c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
        default: return 'N';
    }
}

void put_cigar(Out &o, const basal_hit &h, uint32_t len) {  // align.cpp:641-643
    if (h.gap_size == 0) { o.num(len); o.ch('M'); }
    else if (h.gap_size > 0) { o.num(h.gap_pos); o.ch('M'); o.num(h.gap_size); o.ch('D'); o.num((int)len - (int)h.gap_pos); o.ch('M'); }
    else { o.num(h.gap_pos); o.ch('M'); o.num(-(int)h.gap_size); o.ch('I'); o.num((int)len - (int)h.gap_pos + (int)h.gap_size); o.ch('M'); }
}

// XR:Z (align.cpp:646-658): 2 lower-case flank bases, len bases, 2 lower-case flank bases of the forward strand
void put_xr(Out &o, const basal_params *p, const basal_ref *r, uint32_t contig, uint32_t loc, uint32_t len) {
    if (contig >= r->name.size()) contig = 0;
    const uint64_t *s = r->words[0].data() + r->word_base[contig];
    auto base = [&](uint32_t x) { return p->useful_nt[(s[x >> 5] >> (62 - 2 * (x & 31))) & 3]; };
    o.str("\tXR:Z:");
    for (uint32_t k = 2; k > 0; k--)
        if (loc >= k) o.ch((char)(base(loc - k) + 32));
    for (uint32_t k = 0; k < len + 2; k++) o.ch((char)(base(loc + k) + (k >= len ? 32 : 0)));
}

void put_record(Out &o, const basal_params *p, const basal_ref *r, const char *name, const char *seq, const char *qual, uint32_t readset, int chain,
                int n, uint32_t level, const basal_hit *h) {  // s_OutHit, align.cpp:616-669
    int flag = (int)(0x40 * readset);
    if (n <= 0) {
        if (!p->out_unmap) return;
        flag |= n < 0 ? 0x204 : 0x4;
        o.str(name); o.ch('\t'); o.num(flag); o.str("\t*\t0\t0\t*\t*\t0\t0\t"); o.str(seq); o.ch('\t'); o.str(qual); o.ch('\n');
        return;
    }
    const uint32_t len = (uint32_t)strlen(seq), qlen = (uint32_t)strlen(qual);
    const bool rev = (chain ^ (int)(h->chr & 1)) != 0;
    if (n != 1) flag |= 0x100;
    if (rev) flag |= 0x10;
    o.str(name); o.ch('\t'); o.num(flag); o.ch('\t'); o.str(r->name[h->chr >> 1].c_str()); o.ch('\t'); o.num((long long)h->loc + 1);
    o.str("\t255\t");
    put_cigar(o, *h, len);
    o.str("\t*\t0\t0\t");
    if (!rev) { o.put(seq, len); o.ch('\t'); o.put(qual, qlen); }
    else {
        for (uint32_t i = 0; i < len; i++) o.ch(comp_char(seq[len - 1 - i]));
        o.ch('\t');
        for (uint32_t i = 0; i < qlen; i++) o.ch(qual[qlen - 1 - i]);
    }
    o.str("\tNM:i:"); o.num((uint8_t)level);
    if (p->out_ref) put_xr(o, p, r, (h->chr & 0xfffeu) >> 1, h->loc, len);
    o.str("\tZS:Z:"); o.ch((h->chr & 1) ? '-' : '+'); o.ch(chain ? '-' : '+'); o.ch('\n');
}
}  // namespace

namespace basal {
void put_xr_field(char *out, size_t cap, size_t *n, bool *ok, const basal_params *p, const basal_ref_t *r, uint32_t contig, uint32_t loc, uint32_t len) {
    Out o{out, cap, *n, *ok};
    put_xr(o, p, r, contig, loc, len);
    *n = o.n;
    *ok = o.ok;
}
const char *ref_contig_name(const basal_ref_t *r, uint32_t contig) { return contig < r->name.size() ? r->name[contig].c_str() : "*"; }
}  // namespace basal

extern "C" int64_t basal_host_format_se(const basal_params *p, const basal_ref_t *r, const char *name, const char *seq, const char *qual,
                                        uint32_t readset, int qc_failed, const basal_result *res, const basal_hit *stream, char *out, size_t cap) {
    Out o{out, cap, 0, true};
    static const basal_hit none = {};
    if (qc_failed) put_record(o, p, r, name, seq, qual, readset, 0, -1, 0, &none);
    else if (!res || res->best_level == 0xFF) {
        // StringAlign prints NM with ii = read_max_snp_num+1 there, but n==0 records carry no NM tag
        put_record(o, p, r, name, seq, qual, readset, 0, 0, 0, &none);
    } else {
        const uint32_t sum = (uint32_t)res->n_hit + res->n_chit, ii = res->best_level;
        if (sum == 1) put_record(o, p, r, name, seq, qual, readset, res->best.chain, 1, ii, &res->best);
        else if (p->report_repeat_hits == 1) put_record(o, p, r, name, seq, qual, readset, res->best.chain, (int)sum, ii, &res->best);
        else if (p->report_repeat_hits == 2) {
            if (!stream || res->status == BASAL_READ_OVERFLOW || res->stream_n != sum) { set_error("format_se: -r 2 needs the BASAL_STREAM_BEST hit stream"); return BASAL_EINVAL; }
            for (uint32_t j = 0; j < sum; j++) put_record(o, p, r, name, seq, qual, readset, stream[res->stream_first + j].chain, (int)sum, ii, &stream[res->stream_first + j]);
        } else put_record(o, p, r, name, seq, qual, readset, 0, 0, ii, &none);
    }
    if (!o.ok) { set_error("format_se: output buffer too small"); return BASAL_EOVERFLOW; }
    return (int64_t)o.n;
}

extern "C" int64_t basal_host_sam_header(const basal_ref_t *r, const char *cmdline, char *out, size_t cap) {
    Out o{out, cap, 0, true};
    o.str("@HD\tVN:1.0\n");
    for (size_t i = 0; i < r->name.size(); i++) {
        o.str("@SQ\tSN:"); o.str(r->name[i].c_str()); o.str("\tLN:"); o.num(r->size[i]); o.ch('\n');
    }
    o.str("@PG\tID:BASAL\tVN:1.8.1\tCL:\""); o.str(cmdline ? cmdline : ""); o.str("\"\n");
    if (!o.ok) { set_error("sam_header: output buffer too small"); return BASAL_EOVERFLOW; }
    return (int64_t)o.n;
}
