// basal_main.cpp -- the `basal` command line on top of libbasal_amd.so.
//
// Same flags, messages-to-stderr and SAM-to-file behaviour as the reference driver (main.cpp:272-364 option parser, 409-614
// RunProcess, 60-92 batch loop).  Single-end runs go through the batch pipeline (basal_pipe_*): a reader thread fills page-locked
// buffers with raw read text (plain FASTQ / FASTA files: the bytes as they are in the file, the GPU finds the records) or with
// decoded reads (gzip, BAM, SAM text, irregular text: parsed here with the reference's token semantics, reads.cpp:42-110), the GPU
// does FilterReads + alignment + SAM text, and the main thread writes the text in input order -- so the output equals the
// reference's `-p 1` output.  Paired-end runs align both mates on the GPU and pair them on the host (PairAlign, pairs.cpp).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <atomic>
#include <csignal>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/basal_core.h"

namespace {

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// -o to a regular file is written through a shared mapping of a file grown ahead of the data (Output): whatever ends the run cuts the padding off
static int g_out_fd = -1;
static std::atomic<unsigned long long> g_out_off{0};
static void trim_output() {
    if (g_out_fd >= 0 && ftruncate(g_out_fd, (off_t)g_out_off.load()) != 0) {}
}

void die(const std::string &m, int code = 1) {
    fprintf(stderr, "%s\n", m.c_str());
    trim_output();
    exit(code);
}
// a store into the output mapping that the file system cannot back (disk full, quota): a write error, not a crash
static void on_sigbus(int) {
    static const char msg[] = "write failed on the output file (no space left on the device?)\n";
    if (::write(2, msg, sizeof msg - 1) < 0) {}
    trim_output();
    _exit(1);
}

// ---- streaming input: a sliding window over the (gzip-transparent) byte stream; iostream-token semantics of ReadClass::LoadBatchReads ----
struct Reader {
    gzFile f = nullptr;
    std::vector<char> buf;
    size_t pos = 0, end = 0;  // window = buf[pos, end)
    bool eof = false, fastq = false, bam = false, sam = false;
    uint32_t index = 0;
    static bool ws(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }
    // at least n bytes in the window, or everything up to the end of the stream
    bool ensure(size_t n) {
        if (end - pos >= n || eof) return end - pos >= n;
        if (pos > 0) { memmove(buf.data(), buf.data() + pos, end - pos); end -= pos; pos = 0; }
        if (buf.size() < n + (1u << 22)) buf.resize(n + (1u << 22));
        while (end < buf.size() && !eof) {
            int got = gzread(f, buf.data() + end, (unsigned)std::min<size_t>(buf.size() - end, 1u << 30));
            if (got <= 0) eof = true;
            else end += (size_t)got;
        }
        return end - pos >= n;
    }
    bool open(const char *path, uint64_t offset = 0) {
        f = gzopen(path, "rb");
        if (!f) return false;
        gzbuffer(f, 1 << 20);
        if (offset) gzseek(f, (z_off_t)offset, SEEK_SET);
        ensure(1 << 16);
        size_t p = pos;
        while (p < end && ws((unsigned char)buf[p])) p++;
        fastq = p < end && buf[p] == '@';
        const bool fasta = p < end && buf[p] == '>';
        // main.cpp:386-405 tries FASTA, FASTQ, then BAM, then SAM text. BGZF is multi-member gzip, so the window already holds
        // the uncompressed BAM stream: magic, header text, reference table, then the alignment records
        if (end - pos >= 12 && memcmp(buf.data() + pos, "BAM\1", 4) == 0) {
            bam = true;
            size_t q = 4;
            ensure(q + 4);
            int32_t l_text = i32(q); q += 4 + (size_t)l_text;
            if (!ensure(q + 4)) return false;
            int32_t n_ref = i32(q); q += 4;
            for (int32_t k = 0; k < n_ref; k++) {
                if (!ensure(q + 4)) return false;
                int32_t l_name = i32(q);
                q += 4 + (size_t)l_name + 4;
            }
            if (!ensure(q)) return false;
            pos += q;
        } else if (!fastq && !fasta && p < end) sam = true;  // SAM text without a header (one that starts with @HD reads as FASTQ in the reference too)
        return true;
    }
    void close() { if (f) gzclose(f); f = nullptr; }
    int32_t i32(size_t at) const { int32_t v; memcpy(&v, buf.data() + pos + at, 4); return v; }
    // next BAM record: *at = its offset from pos BEFORE the call advanced pos ... returns pointer to the record body (valid until the next ensure)
    const unsigned char *bam_next(int32_t &bs) {
        if (!ensure(4)) return nullptr;
        bs = i32(0);
        if (bs < 32 || !ensure(4 + (size_t)bs)) return nullptr;
        const unsigned char *b = (const unsigned char *)buf.data() + pos + 4;
        pos += 4 + (size_t)bs;
        return b;
    }
    void skip_ws() {
        for (;;) {
            while (pos < end && ws((unsigned char)buf[pos])) pos++;
            if (pos < end || eof) return;
            ensure(1);
            if (pos >= end) return;
        }
    }
    // the next white-space delimited token (offsets relative to buf.data(), valid until the next ensure that moves the window)
    void token(size_t &b, size_t &l) {
        skip_ws();
        size_t s = pos;
        for (;;) {
            while (pos < end && !ws((unsigned char)buf[pos])) pos++;
            if (pos < end || eof) break;
            const size_t rel = pos - s;  // the token runs into the end of the window: pull more in, keep its start
            pos = s;
            ensure(rel + (1u << 16));
            s = pos;
            pos = s + rel;
            if (pos >= end) break;
        }
        b = s;
        l = pos - s;
    }
    void rest_of_line() {
        for (;;) {
            while (pos < end && buf[pos] != '\n') pos++;
            if (pos < end) { pos++; return; }
            if (eof) return;
            ensure(1);
            if (pos >= end) return;
        }
    }
};

struct Rec {
    uint32_t index, readset;
    std::string name;
    std::vector<char> seq, qual;  // NUL-terminated, trimmed in place by the filter
    int qc_failed = 0;
    uint32_t max_snp = 0;
    bool has_qual = true;
};

// any byte <= ' ' (or >= 0x80) among the n at p: the white space the token reader splits at, conservatively
inline bool has_space(const char *p, size_t n) {
    size_t i = 0;
#if defined(__SSE2__)
    const __m128i lim = _mm_set1_epi8(33);
    __m128i acc = _mm_setzero_si128();
    for (; i + 16 <= n; i += 16) acc = _mm_or_si128(acc, _mm_cmplt_epi8(_mm_loadu_si128((const __m128i *)(p + i)), lim));
    if (_mm_movemask_epi8(acc)) return true;
#endif
    for (; i < n; i++)
        if ((signed char)p[i] < 33) return true;
    return false;
}

const char kNt16[] = "=ACMGRSVTWYHKDBN";
// what bam_nt16_rev_table[bam_nt16_table[c]] gives (samtools 0.1.18 bam_import.c): the reference reads SAM text through it
char nt16_roundtrip(char c) {
    switch (c) {
        case '=': return '=';
        case 'A': case 'a': return 'A'; case 'C': case 'c': return 'C'; case 'M': case 'm': return 'M'; case 'G': case 'g': return 'G';
        case 'R': case 'r': return 'R'; case 'S': case 's': return 'S'; case 'V': case 'v': return 'V'; case 'T': case 't': return 'T';
        case 'W': case 'w': return 'W'; case 'Y': case 'y': return 'Y'; case 'H': case 'h': return 'H'; case 'K': case 'k': return 'K';
        case 'D': case 'd': return 'D'; case 'B': case 'b': return 'B';
        default: return 'N';
    }
}

// one read from any input form; false at the end of the input. readset != 0: the BAM/SAM mate conventions of reads.cpp:84-110
bool next_record(Reader &r, const basal_params &p, int readset, Rec &o) {
    o.qc_failed = 0; o.max_snp = 0; o.has_qual = true;
    if (r.bam) {
        int32_t bs;
        if (readset == 2 && !r.bam_next(bs)) return false;  // mate 2 skips a record, then reads one
        const unsigned char *b = r.bam_next(bs);
        if (!b) return false;
        uint32_t l_name = b[8], n_cigar, flag, l_qseq;
        uint16_t u16;
        memcpy(&u16, b + 12, 2); n_cigar = u16;
        memcpy(&u16, b + 14, 2); flag = u16;
        memcpy(&l_qseq, b + 16, 4);
        const char *name = (const char *)b + 32;
        const unsigned char *sq = b + 32 + l_name + 4 * n_cigar, *ql = sq + (l_qseq + 1) / 2;
        if (ql + l_qseq > b + bs) return false;  // truncated record
        uint32_t l = std::min<uint32_t>(l_qseq, p.max_readlen);
        o.index = r.index;
        o.readset = readset ? ((flag & 0x40) ? 1u : (flag & 0x80) ? 2u : (uint32_t)readset) : 0u;
        o.name.assign(name, strnlen(name, l_name));
        o.seq.assign((size_t)l + 2, 0);
        o.qual.assign((size_t)l + 2, 0);
        for (uint32_t i = 0; i < l; i++) {
            o.seq[i] = kNt16[(sq[i >> 1] >> ((~i & 1) << 2)) & 0xf];
            o.qual[i] = (char)(ql[i] + 33);
        }
        if (readset == 1 && !r.bam_next(bs)) return false;  // (the reference drops a mate 1 without a following record, too)
        return true;
    }
    if (r.sam) {  // one alignment line: QNAME FLAG RNAME POS MAPQ CIGAR RNEXT PNEXT TLEN SEQ QUAL ...
        auto line = [&](std::string &out) {
            out.clear();
            for (;;) {
                r.ensure(1);
                if (r.pos >= r.end) return !out.empty();
                size_t s = r.pos;
                while (r.pos < r.end && r.buf[r.pos] != '\n') r.pos++;
                out.append(r.buf.data() + s, r.pos - s);
                if (r.pos < r.end) { r.pos++; return true; }
                if (r.eof) return !out.empty();
            }
        };
        std::string ln, skip;
        if (readset == 2 && !line(skip)) return false;
        do { if (!line(ln)) return false; } while (ln.empty() || ln[0] == '@');
        while (!ln.empty() && (ln.back() == '\r')) ln.pop_back();
        std::vector<std::pair<size_t, size_t>> fld;
        size_t s = 0;
        for (size_t i = 0; i <= ln.size(); i++)
            if (i == ln.size() || ln[i] == '\t') { fld.push_back({s, i - s}); s = i + 1; }
        if (fld.size() < 11) return false;
        const uint32_t flag = (uint32_t)atoi(ln.c_str() + fld[1].first);
        size_t sl = fld[9].second, ql = fld[10].second;
        const char *sq = ln.data() + fld[9].first, *qq = ln.data() + fld[10].first;
        if (sl == 1 && sq[0] == '*') sl = 0;
        const bool noqual = ql == 1 && qq[0] == '*';
        uint32_t l = (uint32_t)std::min<size_t>(sl, p.max_readlen);
        o.index = r.index;
        o.readset = readset ? ((flag & 0x40) ? 1u : (flag & 0x80) ? 2u : (uint32_t)readset) : 0u;
        o.name.assign(ln.data() + fld[0].first, fld[0].second);
        o.seq.assign((size_t)l + 2, 0);
        o.qual.assign((size_t)l + 2, 0);
        for (uint32_t i = 0; i < l; i++) {
            o.seq[i] = nt16_roundtrip(sq[i]);
            o.qual[i] = noqual ? (char)(0xff + 33) : (char)(((unsigned char)(i < ql ? qq[i] : '!') - 33) + 33);
        }
        if (readset == 1 && !line(skip)) return false;
        return true;
    }
    // Fast path for what nearly every FASTQ file is: a record of exactly four lines, wholly inside the window, with nothing the token
    // reader would treat differently from a line reader (no white space inside or around the sequence and quality lines, no empty name).
    // Lines are found with memchr and checked 16 bytes at a time; anything else falls through to the token reader below, which is
    // ReadClass::LoadBatchReads (reads.cpp:42-81) to the letter.
    if (r.fastq) {
        if (r.end - r.pos < (1u << 14)) r.ensure(1u << 16);
        const char *p0 = r.buf.data() + r.pos, *e = r.buf.data() + r.end;
        if (p0 < e && *p0 == '@') {
            const char *l1 = (const char *)memchr(p0, '\n', (size_t)(e - p0));
            const char *l2 = l1 ? (const char *)memchr(l1 + 1, '\n', (size_t)(e - l1 - 1)) : nullptr;
            const char *l3 = l2 ? (const char *)memchr(l2 + 1, '\n', (size_t)(e - l2 - 1)) : nullptr;
            const char *l4 = l3 ? (const char *)memchr(l3 + 1, '\n', (size_t)(e - l3 - 1)) : nullptr;
            if (l4 && l2[1] == '+') {
                const char *nm = p0 + 1, *ne = nm;
                while (ne < l1 && !Reader::ws((unsigned char)*ne)) ne++;
                const size_t sl = (size_t)(l2 - l1 - 1), ql = (size_t)(l4 - l3 - 1);
                if (ne > nm && sl > 0 && ql > 0 && !has_space(l1 + 1, sl) && !has_space(l3 + 1, ql)) {
                    o.name.assign(nm, (size_t)(ne - nm));
                    o.seq.assign(sl + 2, 0);
                    memcpy(o.seq.data(), l1 + 1, sl);
                    o.qual.assign(std::max(sl, ql) + 2, 0);
                    memcpy(o.qual.data(), l3 + 1, ql);
                    o.index = r.index;
                    o.readset = (uint32_t)readset;
                    if (sl > p.max_readlen) {  // reads.cpp:63-65
                        o.seq[p.max_readlen] = 0;
                        if (strlen(o.qual.data()) > p.max_readlen) o.qual[p.max_readlen] = 0;
                    }
                    r.pos = (size_t)(l4 + 1 - r.buf.data());
                    return true;
                }
            }
        }
    }
    r.skip_ws();
    if (r.pos >= r.end) return false;
    r.pos++;
    size_t nb, nl, sb, sl, qb = 0, ql = 0, tb, tl;
    r.token(nb, nl);
    o.name.assign(r.buf.data() + nb, nl);
    r.rest_of_line();
    r.token(sb, sl);
    o.seq.assign(sl + 2, 0);
    memcpy(o.seq.data(), r.buf.data() + sb, sl);
    if (r.fastq) {
        r.token(tb, tl);
        r.rest_of_line();
        r.token(qb, ql);
        o.qual.assign(std::max(sl, ql) + 2, 0);
        memcpy(o.qual.data(), r.buf.data() + qb, ql);
    } else {
        o.qual.assign(sl + 2, 0);
        memset(o.qual.data(), p.zero_qual + p.default_qual, sl);
        o.has_qual = false;
    }
    o.index = r.index;
    o.readset = (uint32_t)readset;
    if (sl > p.max_readlen) {  // reads.cpp:63-65
        o.seq[p.max_readlen] = 0;
        if (strlen(o.qual.data()) > p.max_readlen) o.qual[p.max_readlen] = 0;
    }
    return true;
}

// The records of `out` are overwritten in place (a batch vector that comes back from the aligning thread keeps its strings' and vectors'
// storage: three heap blocks per read otherwise, allocated here and freed there, millions of times per second).
int load_batch(Reader &r, const basal_params &p, uint32_t read_end, size_t want, int readset, std::vector<Rec> &out, size_t max_bases = ~(size_t)0) {
    size_t bases = 0, n = 0;
    for (; n < want && r.index < read_end && bases < max_bases; r.index++) {
        if (out.size() <= n) out.emplace_back();
        Rec &o = out[n];
        o.qc_failed = 0; o.max_snp = 0; o.has_qual = true;
        if (!next_record(r, p, readset, o)) break;
        bases += o.seq.size();
        n++;
    }
    out.resize(n);
    return (int)n;
}

// ReadClass::InitIndex (reads.cpp:13-40): skip the reads before -B
void skip_reads(Reader &r, const basal_params &p, uint32_t read_start, int pairend) {
    if (read_start <= 1) { r.index = 0; return; }
    if (r.bam || r.sam) {
        Rec tmp;
        Reader *rp = &r;
        for (uint32_t i = 0; i < (read_start - 1) * (1u + (uint32_t)pairend); i++)
            if (!next_record(*rp, p, 0, tmp)) break;
    } else {
        const uint32_t maxi = (read_start - 1) * (2 + 2 * (uint32_t)r.fastq);
        for (uint32_t i = 0; i < maxi; i++) {
            r.ensure(1);
            if (r.pos >= r.end) break;
            r.rest_of_line();
        }
    }
    r.index = read_start - 1;
}

template <typename F>
void parallel_for(size_t n, int threads, F f) {
    if (threads <= 1 || n < 1024) { f(0, n, 0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(f, n * t / threads, n * (t + 1) / threads, t);
    for (auto &t : th) t.join();
}

struct Cli {
    basal_params P;
    std::string qa, qb, ref_file, out_file, rule, cmdline;
    int threads = 1, verbose = 1, device = 0, sam_header = 1;
    std::vector<int> devices;  // -G 0,1,...: the GPUs the reads of every batch are sharded over
    uint32_t read_start = 1, read_end = ~0u;
    size_t batch = 0;
    bool cpu_index = false;
};

// ---- output: a FILE* for pipes / stdout; regular files are written through a shared mapping ----
// (write()/pwrite() to one file take the inode lock, so threads do not add up; stores into a MAP_SHARED mapping fault their pages in
// in parallel. The file is grown ahead in large steps and cut to its final size when it is closed.)
struct Output {
    FILE *fo = stdout;
    bool piped = false, regular = false;
    int fd = -1;
    uint64_t off = 0, size = 0;
    int threads = 1;
    void write(const char *p, size_t n) {
        if (!n) return;
        if (!regular) {
            if (fwrite(p, 1, n, fo) != n) die("write failed on the output");
            return;
        }
        if (off + n > size) {
            // (grown with ftruncate: posix_fallocate makes tmpfs and the page cache touch every page twice, 21 -> 17 Mreads/s into /dev/shm. A full
            // disk or a quota then shows at the first store into the hole, as SIGBUS: on_sigbus reports it as the write error it is)
            size = off + n + (n < (64u << 20) ? (64u << 20) : 4 * (uint64_t)n);
            if (ftruncate(fd, (off_t)size) != 0) die(std::string("cannot grow the output file: ") + strerror(errno));
        }
        const uint64_t page = 4096, m0 = off & ~(page - 1);
        const size_t mlen = (size_t)(off + n - m0);
        char *m = (char *)mmap(nullptr, mlen, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)m0);
        if (m == MAP_FAILED) {  // a file system without shared mappings: plain positional writes
            size_t done = 0;
            while (done < n) {
                ssize_t w = pwrite(fd, p + done, n - done, (off_t)(off + done));
                if (w <= 0) die("write failed on the output file");
                done += (size_t)w;
            }
            off += n;
            g_out_off.store(off);
            return;
        }
        char *dst = m + (off - m0);
        const int nt = n < (4u << 20) ? 1 : threads;
        if (nt <= 1) memcpy(dst, p, n);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < nt; t++)
                th.emplace_back([=] {
                    const size_t b = n * (size_t)t / (size_t)nt, e = n * (size_t)(t + 1) / (size_t)nt;
                    memcpy(dst + b, p + b, e - b);
                });
            for (auto &t : th) t.join();
        }
        munmap(m, mlen);
        off += n;
        g_out_off.store(off);
    }
    void close() {
        if (piped) pclose(fo);
        else if (regular) { g_out_fd = -1; if (ftruncate(fd, (off_t)off) != 0) die("cannot set the size of the output file"); ::close(fd); }
        else if (fo != stdout) fclose(fo);
        else fflush(stdout);
    }
};

// =========================================================================== single-end: the GPU pipeline
struct SeStats {
    uint64_t n_reads = 0, n_aligned = 0, n_unique = 0, n_multiple = 0;
    double ms[5] = {0, 0, 0, 0, 0};
    double t_create = 0, t_read = 0, t_acquire = 0, t_write = 0, t_collect = 0;  // host side: where the wall clock went
};

// the last complete FASTQ / FASTA record in buf[0, n): returns the number of bytes that hold whole records. A FASTQ record
// starts at a line that begins with '@' and whose next-but-one line begins with '+' (a quality line may begin with '@', but the
// line two below it is then a base line, which cannot begin with '+').
size_t cut_at_record(const char *buf, size_t n, bool fastq, bool at_eof) {
    if (at_eof) return n;
    // the starts of the last lines, last first (the last one may be a partial line)
    size_t starts[24];
    int ns = 0;
    size_t p = n;
    while (ns < 24) {
        const void *q = p ? memrchr(buf, '\n', p) : nullptr;  // the last newline before byte p
        const size_t ls = q ? (size_t)((const char *)q - buf) + 1 : 0;
        if (ls < n) starts[ns++] = ls;
        if (!q) break;
        p = (size_t)((const char *)q - buf);
    }
    for (int i = 0; i < ns; i++) {
        const size_t s = starts[i];
        if (!fastq) { if (buf[s] == '>') return s; continue; }
        if (buf[s] == '@' && i >= 2 && buf[starts[i - 2]] == '+') return s;
    }
    return 0;
}

struct SePlan { bool plain = false; basal_pipe_opts po; uint64_t light_bytes = 0; uint32_t light_reads = 0; };  // light_*: the batch of an index without long lists

// the input form and the batch geometry (before anything is staged: the pipeline's page-locked buffers are set up meanwhile)
SePlan plan_se(const Cli &cli) {
    bool plain = false;
    uint64_t est_bytes = ~0ull;  // an estimate of the uncompressed input size, to size the batches of a small run
    {
        FILE *f = fopen(cli.qa.c_str(), "rb");
        if (!f) die("failed to open read file (check -a option): " + cli.qa);
        unsigned char m[4] = {0, 0, 0, 0};
        size_t got = fread(m, 1, 4, f);
        struct stat sb;
        const bool reg = fstat(fileno(f), &sb) == 0 && S_ISREG(sb.st_mode), gz = m[0] == 0x1f && m[1] == 0x8b;
        plain = got >= 1 && !gz && (m[0] == '@' || m[0] == '>') && reg;
        if (reg) est_bytes = (uint64_t)sb.st_size * (gz ? 8 : 1);
        fclose(f);
    }
    if (getenv("BASAL_HOST_PARSE")) plain = false;
    basal_pipe_opts po;
    memset(&po, 0, sizeof po);
    po.depth = 3;
    // batches of half a million reads: a batch costs a few tenths of a millisecond of fixed GPU time, while page-locking its
    // buffers costs host time in proportion to their size
    // The pipe's buffers are sized for 2 Mi reads per batch; how much of that a batch uses is decided when the index is there (run_se): a launch cannot
    // end before its longest read has, and on a repeat-rich index (over-represented-k-mer cut-off >= 32 768: any real genome) one read in 80 000
    // runs 14-28 ms -- batches of half a million reads then cost 16 ms each instead of 6 (26 against 45 Mreads/s on the hg38-like stand-in), while on
    // an index without long lists the smaller batch overlaps the stages better (87 against 80). -Z sets both.
    const uint32_t light_reads = cli.batch ? (uint32_t)std::min<size_t>(cli.batch, 16u << 20) : (512u << 10);
    po.max_reads = cli.batch ? light_reads : (2048u << 10);
    // bytes per batch: room for 100-base FASTQ records with short names at max_reads; longer records make batches of fewer reads
    po.max_bytes = std::min<uint64_t>((uint64_t)po.max_reads * 160 + (1u << 20), 0xF0000000ull);
    if (est_bytes < po.max_bytes) {  // a small input: small buffers (page-locking gigabytes takes longer than aligning a few thousand reads)
        po.max_bytes = std::max<uint64_t>(est_bytes + (64u << 10), 1u << 20);
        po.max_reads = (uint32_t)std::min<uint64_t>(po.max_reads, std::max<uint64_t>(po.max_bytes / 16, 4096));
    }
    if (const char *e = getenv("BASAL_PIPE_BYTES")) po.max_bytes = std::max<uint64_t>(4096, (uint64_t)atoll(e));  // (tests: many small batches)
    po.output = BASAL_PIPE_OUT_SAM;
    po.max_reads = (po.max_reads + 4095u) & ~4095u;
    po.max_bytes = (po.max_bytes + 4095ull) & ~4095ull;
    SePlan pl;
    pl.plain = plain;
    pl.po = po;
    pl.light_reads = std::min(light_reads, po.max_reads);
    pl.light_bytes = std::min<uint64_t>(((uint64_t)pl.light_reads * 160 + (1u << 20) + 4095ull) & ~4095ull, po.max_bytes);
    return pl;
}

void run_se(Cli &cli, basal_pipe_t *pipe, const SePlan &plan, bool long_lists, Output &out, SeStats &st, double &t_wait_gpu) {
    const basal_params &P = cli.P;
    bool plain = plan.plain;
    basal_pipe_opts po = plan.po;
    if (!long_lists) {  // (an index without long lists: the smaller batch -- see plan_se; the pipe's buffers hold either)
        po.max_bytes = std::min<uint64_t>(po.max_bytes, plan.light_bytes);
        po.max_reads = std::min(po.max_reads, plan.light_reads);
    }
    if (basal_pipe_set_read_range(pipe, cli.read_start - 1, cli.read_end)) die(basal_last_error());

    // reader thread -> pipe; the main thread collects and writes. A refused (irregular) text batch restarts the reader in
    // host-parse mode at the file offset of that batch.
    struct BatchInfo { uint64_t file_off; };
    std::mutex qm;
    std::deque<BatchInfo> submitted;
    std::atomic<bool> reader_done{false}, stop{false};
    std::string reader_err;
    auto reader_text = [&](uint64_t start_off) {
        int fd = open(cli.qa.c_str(), O_RDONLY);
        if (fd < 0) { reader_err = "failed to open read file (check -a option): " + cli.qa; reader_done = true; return; }
        struct stat sb;
        fstat(fd, &sb);
        const uint64_t fsize = (uint64_t)sb.st_size;
        uint64_t off = start_off;
        bool fastq = true, first = true;
        // -B: skip the lines of the reads before read_start (text form: on the host, it is a prefix of the file)
        if (cli.read_start > 1) {
            Reader r;
            if (!r.open(cli.qa.c_str())) { reader_err = "failed to open read file"; reader_done = true; ::close(fd); return; }
            skip_reads(r, P, cli.read_start, 0);
            off = (uint64_t)gztell(r.f) - (r.end - r.pos);
            r.close();
        }
        std::vector<char> left;
        const int rthreads = std::max(1, std::min(cli.threads, 8));
        while (!stop) {
            uint8_t *blob = nullptr;
            basal_rawread *raw = nullptr;
            const double ta0 = now();
            if (basal_pipe_acquire(pipe, &blob, &raw)) break;  // the pipe was stopped (a batch was refused)
            const double ta1 = now();
            st.t_acquire += ta1 - ta0;
            if (stop) { basal_pipe_cancel(pipe); break; }
            size_t have = left.size();
            memcpy(blob, left.data(), have);
            const uint64_t batch_off = off - have;
            const size_t want = (size_t)std::min<uint64_t>(po.max_bytes - 1 - have, fsize - off);
            if (rthreads > 1 && want > (32u << 20)) {  // the page cache gives a few GB/s per core
                std::vector<std::thread> th;
                std::atomic<bool> bad{false};
                for (int t = 0; t < rthreads; t++)
                    th.emplace_back([&, t] {
                        size_t b = want * (size_t)t / (size_t)rthreads, e = want * (size_t)(t + 1) / (size_t)rthreads;
                        while (b < e) {
                            ssize_t g = pread(fd, blob + have + b, e - b, (off_t)(off + b));
                            if (g <= 0) { bad = true; return; }
                            b += (size_t)g;
                        }
                    });
                for (auto &t : th) t.join();
                if (bad) { reader_err = "read failed on " + cli.qa; basal_pipe_cancel(pipe); break; }
            } else {
                size_t b = 0;
                while (b < want) {
                    ssize_t g = pread(fd, blob + have + b, want - b, (off_t)(off + b));
                    if (g <= 0) break;
                    b += (size_t)g;
                }
                if (b < want) { reader_err = "read failed on " + cli.qa; basal_pipe_cancel(pipe); break; }
            }
            off += want;
            have += want;
            st.t_read += now() - ta1;
            const bool at_eof = off >= fsize;
            if (first) {
                size_t p = 0;
                while (p < have && Reader::ws(blob[p])) p++;
                fastq = p < have && blob[p] == '@';
                first = false;
            }
            if (at_eof && have && blob[have - 1] != '\n') blob[have++] = '\n';  // a last line without a newline
            const size_t keep = cut_at_record((const char *)blob, have, fastq, at_eof);
            if (keep == 0 && have) {  // no record boundary in a whole buffer: not something the text path can take
                std::lock_guard<std::mutex> lk(qm);
                submitted.push_back({batch_off});
                // hand the whole buffer over: the device flags it irregular and the fallback takes it from there
                if (basal_pipe_submit_text(pipe, have, fastq ? BASAL_FMT_FASTQ : BASAL_FMT_FASTA, 0xFFFFFFFFu, 0)) break;
                left.clear();
                if (at_eof) break;
                continue;
            }
            if (have == 0) { basal_pipe_cancel(pipe); break; }
            left.assign((const char *)blob + keep, (const char *)blob + have);
            {
                std::lock_guard<std::mutex> lk(qm);
                submitted.push_back({batch_off});
            }
            if (basal_pipe_submit_text(pipe, keep, fastq ? BASAL_FMT_FASTQ : BASAL_FMT_FASTA, 0xFFFFFFFFu, 0)) {
                std::lock_guard<std::mutex> lk(qm);
                submitted.pop_back();
                break;
            }
            if (at_eof) break;
        }
        ::close(fd);
        reader_done = true;
    };
    auto reader_host = [&](uint64_t start_off, uint32_t first_index, bool skip_to_start) {
        Reader r;
        if (!r.open(cli.qa.c_str(), start_off)) { reader_err = "failed to open read file (check -a option): " + cli.qa; reader_done = true; return; }
        if (skip_to_start) skip_reads(r, P, cli.read_start, 0);
        else r.index = first_index;
        Rec rec;
        bool have_rec = false;  // a read that was parsed but did not fit the batch any more: it opens the next one
        while (!stop) {
            uint8_t *blob = nullptr;
            basal_rawread *raw = nullptr;
            if (basal_pipe_acquire(pipe, &blob, &raw)) break;
            if (stop) { basal_pipe_cancel(pipe); break; }
            uint32_t n = 0;
            uint64_t nb = 0;
            bool more = true;
            while (n < po.max_reads && r.index < cli.read_end) {
                if (!have_rec) {
                    if (!next_record(r, P, 0, rec)) { more = false; break; }
                    have_rec = true;
                }
                const size_t sl = strlen(rec.seq.data()), ql = rec.has_qual ? strlen(rec.qual.data()) : 0;
                if (nb + rec.name.size() + sl + ql > po.max_bytes) break;
                basal_rawread &w = raw[n];
                memset(&w, 0, sizeof w);
                w.name_off = (uint32_t)nb; w.name_len = (uint16_t)std::min<size_t>(rec.name.size(), 0xffff);
                memcpy(blob + nb, rec.name.data(), w.name_len); nb += w.name_len;
                w.seq_off = (uint32_t)nb; w.seq_len = (uint16_t)sl;
                memcpy(blob + nb, rec.seq.data(), sl); nb += sl;
                w.qual_off = (uint32_t)nb; w.qual_len = (uint16_t)ql;
                memcpy(blob + nb, rec.qual.data(), ql); nb += ql;
                w.readset = 0; w.index = r.index;
                n++; r.index++;
                have_rec = false;
            }
            if (n == 0) { basal_pipe_cancel(pipe); break; }
            {
                std::lock_guard<std::mutex> lk(qm);
                submitted.push_back({0});
            }
            if (basal_pipe_submit_records(pipe, nb, n)) break;
            if (!more || r.index >= cli.read_end) break;
        }
        r.close();
        reader_done = true;
    };

    std::thread rt;
    if (plain) rt = std::thread(reader_text, (uint64_t)0);
    else rt = std::thread(reader_host, (uint64_t)0, 0u, true);
    for (;;) {
        const void *data = nullptr;
        uint64_t nbytes = 0;
        basal_batch_stats bs;
        double w0 = now();
        int rc = basal_pipe_collect(pipe, &data, &nbytes, &bs);
        t_wait_gpu += now() - w0;
        if (rc == BASAL_ESTATE) {  // nothing in flight
            if (reader_done) {
                // the reader may have submitted its last batch between our collect and this test
                rc = basal_pipe_collect(pipe, &data, &nbytes, &bs);
                if (rc == BASAL_ESTATE) break;
            } else { std::this_thread::sleep_for(std::chrono::microseconds(200)); continue; }
        }
        if (rc == BASAL_EIO && plain) {
            // irregular text: stop the reader, drop what is in flight, go on in host-parse mode from this batch's first byte
            stop = true;
            basal_pipe_stop(pipe);
            rt.join();
            uint64_t foff;
            {
                std::lock_guard<std::mutex> lk(qm);
                foff = submitted.empty() ? 0 : submitted.front().file_off;
                submitted.clear();
            }
            if (basal_pipe_rewind(pipe)) die(basal_last_error());
            if (cli.verbose >= 1) fprintf(stderr, "[BASAL-MI355X] read text is not 4 regular lines per read from byte %llu on: parsing it on the host\n", (unsigned long long)foff);
            plain = false;
            stop = false;
            reader_done = false;
            rt = std::thread(reader_host, foff, (uint32_t)(cli.read_start - 1 + st.n_reads), false);
            continue;
        }
        if (rc) die(std::string("pipeline: ") + basal_last_error());
        {
            std::lock_guard<std::mutex> lk(qm);
            if (!submitted.empty()) submitted.pop_front();
        }
        const double tw0 = now();
        out.write((const char *)data, (size_t)nbytes);
        st.t_write += now() - tw0;
        st.n_reads += bs.n_reads; st.n_aligned += bs.n_aligned; st.n_unique += bs.n_unique; st.n_multiple += bs.n_multiple;
        st.ms[0] += bs.ms_h2d; st.ms[1] += bs.ms_prep; st.ms[2] += bs.ms_align; st.ms[3] += bs.ms_format; st.ms[4] += bs.ms_d2h;
        if (cli.verbose >= 2) fprintf(stderr, "[BASAL-MI355X] %llu reads finished.\n", (unsigned long long)st.n_reads);
    }
    rt.join();
    if (!reader_err.empty()) die(reader_err);
}

// one batch through one GPU (basal_core_align_batch) or sharded over several (basal_multi_align_batch: RCCL gather of the records)
struct Aligner {
    basal_core_t *core = nullptr;
    basal_multi_t *multi = nullptr;
    int run(const uint8_t *bases, uint64_t nbases, const basal_read *reads, uint32_t n, const basal_stale *stales, uint32_t nstale, int mode, basal_result *res,
            basal_hit *stream, uint64_t cap, uint64_t *used, uint8_t carry[2][2]) {
        return multi ? basal_multi_align_batch(multi, bases, nbases, reads, n, stales, nstale, mode, res, stream, cap, used, carry)
                     : basal_core_align_batch(core, bases, nbases, reads, n, stales, nstale, mode, res, stream, cap, used, carry);
    }
};

// =========================================================================== single-end with host-side QC and SAM text (several GPUs)
// The reads of every batch are sharded over the GPUs by read number; FilterReads and s_OutHit run here on `-p` host threads
// (basal_host_filter_read / basal_host_format_se), the way round 1's command line did on one GPU.
void run_se_host(Cli &cli, Aligner &al, basal_ref_t *R, Output &out, SeStats &st) {
    const basal_params &P = cli.P;
    const int threads = cli.threads;
    const size_t batch = cli.batch ? cli.batch : (1u << 20);
    Reader ra;
    if (!ra.open(cli.qa.c_str())) die("failed to open read file (check -a option): " + cli.qa);
    skip_reads(ra, P, cli.read_start, 0);
    std::vector<Rec> recs;
    std::vector<uint8_t> bases;
    std::vector<basal_read> descs;
    std::vector<basal_result> results;
    std::vector<basal_hit> stream;
    std::vector<basal_stale> stales;
    basal_stale_tracker_t *tracker = basal_host_stale_new(&P);
    uint8_t carry[2][2] = {{0, 0}, {0, 0}};
    const int smode = P.report_repeat_hits == 2 ? BASAL_STREAM_BEST : BASAL_STREAM_NONE;
    while (load_batch(ra, P, cli.read_end, batch, 0, recs, (size_t)3500 << 20)) {
        const size_t n = recs.size();
        parallel_for(n, threads, [&](size_t b, size_t e, int) {
            for (size_t i = b; i < e; i++) recs[i].qc_failed = basal_host_filter_read(&P, recs[i].seq.data(), recs[i].qual.data(), &recs[i].max_snp);
        });
        descs.assign(n, basal_read{});
        bases.clear();
        stales.clear();
        basal_host_stale_begin_batch(tracker);
        for (size_t i = 0; i < n; i++) {
            basal_read &d = descs[i];
            d.index = recs[i].index;
            d.readset = 0;
            d.stale_idx = BASAL_STALE_NONE;
            if (recs[i].qc_failed) { d.len = 0; continue; }
            uint32_t len = (uint32_t)strlen(recs[i].seq.data());
            d.len = (uint16_t)len;
            d.max_snp = (uint8_t)recs[i].max_snp;
            d.seq_off = (uint32_t)bases.size();
            bases.insert(bases.end(), recs[i].seq.begin(), recs[i].seq.begin() + len);
            basal_stale se;
            if (basal_host_stale_visit(tracker, recs[i].seq.data(), len, 0, 0, (uint32_t)i, &se)) {
                d.stale_idx = (uint32_t)stales.size();
                stales.push_back(se);
            }
        }
        results.assign(n, basal_result{});
        uint64_t cap = smode ? (uint64_t)n * 4 + 1024 : 0, used = 0;
        for (int attempt = 0;; attempt++) {
            stream.resize(cap ? cap : 1);
            uint8_t cy[2][2];
            memcpy(cy, carry, 4);
            int rc = al.run(bases.data(), bases.size(), descs.data(), (uint32_t)n, stales.data(), (uint32_t)stales.size(), smode, results.data(), stream.data(), cap, &used, cy);
            if (rc == BASAL_EOVERFLOW && attempt < 4) { cap = std::max(used, cap) + 1024; continue; }  // (used = the capacity that fits, one retry is enough; never loop for ever)
            if (rc) die(std::string("align_batch: ") + basal_last_error());
            memcpy(carry, cy, 4);
            break;
        }
        std::vector<std::string> chunks((size_t)std::max(threads, 1));
        std::vector<uint64_t> cnt((size_t)std::max(threads, 1) * 3, 0);
        parallel_for(n, threads, [&](size_t b, size_t e, int tid) {
            std::string &o = chunks[(size_t)tid];
            std::vector<char> line(1 << 16);
            for (size_t i = b; i < e; i++) {
                const Rec &rc_ = recs[i];
                const basal_result &rs = results[i];
                size_t need = 4096 + rc_.name.size() + 2 * rc_.seq.size() + (size_t)(rs.stream_n + 1) * (1024 + 2 * rc_.seq.size());
                if (line.size() < need) line.resize(need);
                int64_t w = basal_host_format_se(&P, R, rc_.name.c_str(), rc_.seq.data(), rc_.qual.data(), 0, rc_.qc_failed, &rs, stream.data(), line.data(), line.size());
                if (w < 0) die(std::string("format: ") + basal_last_error());
                o.append(line.data(), (size_t)w);
                if (!rc_.qc_failed && rs.best_level != 0xFF) {
                    uint32_t sum = (uint32_t)rs.n_hit + rs.n_chit;
                    if (sum == 1) { cnt[3 * tid]++; cnt[3 * tid + 1]++; }
                    else { cnt[3 * tid + 2]++; if (P.report_repeat_hits) cnt[3 * tid]++; }
                }
            }
        });
        for (auto &c : chunks) out.write(c.data(), c.size());
        for (size_t t = 0; t < chunks.size(); t++) { st.n_aligned += cnt[3 * t]; st.n_unique += cnt[3 * t + 1]; st.n_multiple += cnt[3 * t + 2]; }
        st.n_reads += n;
    }
    basal_host_stale_free(tracker);
    ra.close();
}

// =========================================================================== paired-end: GPU alignment, host pairing
// a bounded hand-over between a reader thread and the thread that aligns: at most `cap` batches wait
template <typename T>
struct BatchQueue {
    std::mutex m;
    std::condition_variable cv;
    std::deque<T> q;
    size_t cap = 2;
    bool closed = false;
    void push(T &&v) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return q.size() < cap || closed; });
        if (closed) return;
        q.push_back(std::move(v));
        cv.notify_all();
    }
    void try_push(T &&v) {  // never waits: what does not fit is dropped
        std::lock_guard<std::mutex> l(m);
        if (q.size() < cap && !closed) q.push_back(std::move(v));
    }
    bool pop(T &v) {  // false: the producer is done and nothing is left
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return !q.empty() || closed; });
        if (q.empty()) return false;
        v = std::move(q.front());
        q.pop_front();
        cv.notify_all();
        return true;
    }
    void close() {
        std::lock_guard<std::mutex> l(m);
        closed = true;
        cv.notify_all();
    }
};

// =========================================================================== paired-end through the device-side pipeline (one GPU)
// number of '\n' among the n bytes at p
inline size_t count_newlines(const char *p, size_t n) {
    size_t c = 0, i = 0;
#if defined(__SSE2__)
    const __m128i nl = _mm_set1_epi8('\n');
    for (; i + 16 <= n; i += 16) c += (size_t)__builtin_popcount((unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + i)), nl)));
#endif
    for (; i < n; i++) c += p[i] == '\n';
    return c;
}
// bytes up to and including the k-th '\n' (k >= 1) among the n bytes at p; 0 if there are fewer
inline size_t bytes_of_lines(const char *p, size_t n, size_t k) {
    size_t c = 0, i = 0;
#if defined(__SSE2__)
    const __m128i nl = _mm_set1_epi8('\n');
    for (; i + 16 <= n; i += 16) {
        const unsigned m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + i)), nl));
        const size_t here = (size_t)__builtin_popcount(m);
        if (c + here >= k) break;
        c += here;
    }
#endif
    for (; i < n; i++)
        if (p[i] == '\n' && ++c == k) return i + 1;
    return 0;
}

// Paired-end: FilterReads x 2, FixPairReadName, the mates' alignment, the pairing rounds and the text of s_OutHitPair / s_OutHitUnpair
// (pairs.cpp:179-507) all run on the GPU (basal_pipe_* with BASAL_PIPE_PAIRS); the host only cuts and writes. Plain FASTQ / FASTA mate files go
// to the GPU as text: mate 1's records of a batch, then the same number of mate 2's (the host counts newlines; the device finds the records
// and refuses a text whose lines and the reference's token reader would disagree on -- the run then continues in the second form from that
// batch's first bytes). Everything else (gz, BAM, irregular text) is parsed by two reader threads and handed over as a table of records.
void run_pe_pipe(Cli &cli, basal_pipe_t *pipe, const basal_pipe_opts &po, Output &out, uint32_t pst[9], uint64_t &n_pairs, double &t_gpu) {
    basal_params &P = cli.P;
    const int threads = cli.threads;
    bool plain = !getenv("BASAL_HOST_PARSE");
    bool fastq = true;
    if (plain) {
        int lead[2] = {0, 0};
        const std::string *paths[2] = {&cli.qa, &cli.qb};
        for (int m = 0; m < 2 && plain; m++) {
            FILE *f = fopen(paths[m]->c_str(), "rb");
            if (!f) die(std::string(m ? "failed to open read file #2 (check -b option): " : "failed to open read file (check -a option): ") + *paths[m]);
            struct stat sb;
            lead[m] = fgetc(f);
            plain = fstat(fileno(f), &sb) == 0 && S_ISREG(sb.st_mode) && (lead[m] == '@' || lead[m] == '>');
            fclose(f);
        }
        plain = plain && lead[0] == lead[1];
        fastq = lead[0] == '@';
    }
    struct BatchInfo { uint64_t off_a, off_b; uint32_t first_index; };
    std::mutex qm;
    std::deque<BatchInfo> submitted;
    std::atomic<bool> producer_done{false}, stop{false};
    std::string producer_err;

    auto producer_text = [&]() {
        int fd[2] = {open(cli.qa.c_str(), O_RDONLY), open(cli.qb.c_str(), O_RDONLY)};
        if (fd[0] < 0 || fd[1] < 0) { producer_err = "failed to open the read files"; producer_done = true; return; }
        uint64_t fsize[2], off[2] = {0, 0};
        for (int m = 0; m < 2; m++) { struct stat sb; fstat(fd[m], &sb); fsize[m] = (uint64_t)sb.st_size; }
        if (cli.read_start > 1)  // -B: skip the lines of the pairs before read_start (a prefix of both files)
            for (int m = 0; m < 2; m++) {
                Reader r;
                if (!r.open(m ? cli.qb.c_str() : cli.qa.c_str())) { producer_err = "failed to open read file"; producer_done = true; return; }
                skip_reads(r, P, cli.read_start, 1);
                off[m] = (uint64_t)gztell(r.f) - (r.end - r.pos);
                r.close();
            }
        const size_t lpr = fastq ? 4 : 2;
        uint32_t index = cli.read_start - 1;
        double est[2] = {2.0 * P.max_readlen + 64, 2.0 * P.max_readlen + 64};  // bytes per record of either file: from the batch before (first: generous)
        const int rthreads = std::max(1, std::min(threads, 8));
        auto pread_all = [&](int f, uint8_t *dst, size_t n, uint64_t at) {
            if (n > (8u << 20) && rthreads > 1) {  // the page cache gives a few GB/s per core
                std::vector<std::thread> th;
                std::atomic<bool> bad{false};
                for (int t = 0; t < rthreads; t++)
                    th.emplace_back([&, t] {
                        size_t b = n * (size_t)t / (size_t)rthreads, e = n * (size_t)(t + 1) / (size_t)rthreads;
                        while (b < e) {
                            ssize_t g = pread(f, dst + b, e - b, (off_t)(at + b));
                            if (g <= 0) { bad = true; return; }
                            b += (size_t)g;
                        }
                    });
                for (auto &t : th) t.join();
                return !bad.load();
            }
            size_t b = 0;
            while (b < n) {
                ssize_t g = pread(f, dst + b, n - b, (off_t)(at + b));
                if (g <= 0) return false;
                b += (size_t)g;
            }
            return true;
        };
        while (!stop) {
            if (off[0] >= fsize[0] || off[1] >= fsize[1] || index >= cli.read_end) break;
            uint8_t *blob = nullptr;
            basal_rawread *raw = nullptr;
            if (basal_pipe_acquire(pipe, &blob, &raw)) break;  // the pipe was stopped (a batch was refused)
            if (stop) { basal_pipe_cancel(pipe); break; }
            // mate 1: the batch's records out of a window sized from the records seen so far (at most half the buffer)
            const size_t want_pairs = std::min<size_t>(po.max_reads / 2, (size_t)(cli.read_end - index));
            size_t have_a = (size_t)std::min<uint64_t>(std::min<uint64_t>((uint64_t)(want_pairs * est[0] * 1.03) + (64u << 10), (po.max_bytes - 4096) / 2), fsize[0] - off[0]);
            if (!pread_all(fd[0], blob, have_a, off[0])) { producer_err = "read failed on " + cli.qa; basal_pipe_cancel(pipe); break; }
            size_t file_a = have_a;
            if (off[0] + have_a >= fsize[0] && blob[have_a - 1] != '\n') blob[have_a++] = '\n';  // a last line without a newline
            size_t np = want_pairs;
            size_t keep_a = bytes_of_lines((const char *)blob, have_a, np * lpr);
            if (!keep_a) {  // the window holds fewer records than asked for (the end of the file, or longer records than estimated): take what it holds
                np = count_newlines((const char *)blob, have_a) / lpr;
                keep_a = np ? bytes_of_lines((const char *)blob, have_a, np * lpr) : 0;
            }
            // mate 2: the same number of records, right behind
            size_t have_b = 0, keep_b = 0, file_b = 0;
            bool eof_b = false;
            if (np) {
                have_b = (size_t)std::min<uint64_t>(std::min<uint64_t>((uint64_t)(np * est[1] * 1.03) + (64u << 10), po.max_bytes - keep_a - 1), fsize[1] - off[1]);
                if (!pread_all(fd[1], blob + keep_a, have_b, off[1])) { producer_err = "read failed on " + cli.qb; basal_pipe_cancel(pipe); break; }
                file_b = have_b;
                eof_b = off[1] + have_b >= fsize[1];
                if (eof_b && blob[keep_a + have_b - 1] != '\n') blob[keep_a + have_b++] = '\n';
                keep_b = bytes_of_lines((const char *)blob + keep_a, have_b, np * lpr);
                if (!keep_b && !eof_b) {  // mate 2's window was too small for these records: fewer pairs this time, a better estimate next time
                    const size_t nb = count_newlines((const char *)blob + keep_a, have_b) / lpr;
                    if (nb) {
                        const size_t old_a = keep_a;
                        np = nb;
                        keep_a = bytes_of_lines((const char *)blob, have_a, np * lpr);
                        keep_b = bytes_of_lines((const char *)blob + old_a, have_b, np * lpr);
                        memmove(blob + keep_a, blob + old_a, keep_b);
                    }
                }
            }
            if (np && keep_a && keep_b) { est[0] = std::max(64.0, (double)keep_a / (double)np); est[1] = std::max(64.0, (double)keep_b / (double)np) * (keep_b ? 1.0 : 2.0); }
            else if (np && !eof_b) est[1] *= 2;
            if (!np || !keep_a || !keep_b) {
                basal_pipe_cancel(pipe);
                if (np && eof_b) {  // mate 2's file ends inside this batch: the reference stops at the batch whose two halves differ in size (main.cpp:105)
                    if (cli.verbose >= 1) fprintf(stderr, "[BASAL-MI355X] warning: the mate files do not hold the same number of reads; stopping at read pair %u\n", index);
                    break;
                }
                // no whole record in a whole window: not something the text path can take -- hand over to the host parser from here
                std::lock_guard<std::mutex> lk(qm);
                submitted.push_back({off[0], off[1], index});
                producer_err = "FALLBACK";
                break;
            }
            {
                std::lock_guard<std::mutex> lk(qm);
                submitted.push_back({off[0], off[1], index});
            }
            if (basal_pipe_submit_text_pairs(pipe, keep_a + keep_b, keep_a, (uint32_t)np, fastq ? BASAL_FMT_FASTQ : BASAL_FMT_FASTA, index)) {
                std::lock_guard<std::mutex> lk(qm);
                submitted.pop_back();
                break;
            }
            off[0] += std::min(keep_a, file_a);  // (an appended newline is not in the file)
            off[1] += std::min(keep_b, file_b);
            index += (uint32_t)np;
        }
        ::close(fd[0]);
        ::close(fd[1]);
        producer_done = true;
    };

    auto producer_records = [&](uint64_t off_a, uint64_t off_b, uint32_t first_index, bool skip_to_start) {
        Reader ra, rb;
        if (!ra.open(cli.qa.c_str(), off_a)) { producer_err = "failed to open read file (check -a option): " + cli.qa; producer_done = true; return; }
        if (!rb.open(cli.qb.c_str(), off_b)) { producer_err = "failed to open read file #2 (check -b option): " + cli.qb; producer_done = true; return; }
        if (skip_to_start) { skip_reads(ra, P, cli.read_start, 1); skip_reads(rb, P, cli.read_start, 1); }
        else ra.index = rb.index = first_index;
        BatchQueue<std::vector<Rec>> qa, qb, qfree;
        qfree.cap = 8;
        auto recycled = [&](std::vector<Rec> &v) {
            std::lock_guard<std::mutex> l(qfree.m);
            if (qfree.q.empty()) return;
            v = std::move(qfree.q.front());
            qfree.q.pop_front();
        };
        const size_t per = po.max_reads / 2;  // pairs per batch
        // (reads are cut to max_readlen as they are parsed: two records of a pair stay below max_bytes / per together)
        std::thread ta([&] {
            while (!stop.load()) {
                std::vector<Rec> v;
                recycled(v);
                if (!load_batch(ra, P, cli.read_end, per, 1, v)) break;
                qa.push(std::move(v));
            }
            qa.close();
        });
        std::thread tb([&] {
            while (!stop.load()) {
                std::vector<Rec> v;
                recycled(v);
                load_batch(rb, P, cli.read_end, per, 2, v);
                if (v.empty()) break;
                const bool short_file = v.size() != per;
                qb.push(std::move(v));
                if (short_file) break;
            }
            qb.close();
        });
        std::vector<Rec> ra_, rb_;
        std::vector<size_t> off;
        while (!stop) {
            if (!ra_.empty()) qfree.try_push(std::move(ra_));
            if (!rb_.empty()) qfree.try_push(std::move(rb_));
            ra_.clear(); rb_.clear();
            const bool ga = qa.pop(ra_), gb = ga && qb.pop(rb_);
            if (!ga || !gb || ra_.empty()) break;
            if (ra_.size() != rb_.size()) {  // the reference stops at the batch whose two halves differ in size (main.cpp:105)
                if (cli.verbose >= 1) fprintf(stderr, "[BASAL-MI355X] warning: the mate files do not hold the same number of reads; stopping at read pair %u\n", ra_[0].index);
                break;
            }
            const size_t np = ra_.size();
            uint8_t *blob = nullptr;
            basal_rawread *raw = nullptr;
            if (basal_pipe_acquire(pipe, &blob, &raw)) break;
            if (stop) { basal_pipe_cancel(pipe); break; }
            off.resize(np + 1);
            off[0] = 0;
            for (size_t i = 0; i < np; i++) {
                const Rec &x = ra_[i], &y = rb_[i];
                off[i + 1] = off[i] + std::min<size_t>(x.name.size(), 0xffff) + strlen(x.seq.data()) + (x.has_qual ? strlen(x.qual.data()) : 0) +
                             std::min<size_t>(y.name.size(), 0xffff) + strlen(y.seq.data()) + (y.has_qual ? strlen(y.qual.data()) : 0);
            }
            if (off[np] > po.max_bytes || 2 * np > po.max_reads) { producer_err = "paired-end batch larger than the pipeline's buffers"; basal_pipe_cancel(pipe); break; }
            parallel_for(np, threads, [&](size_t b, size_t e, int) {
                for (size_t i = b; i < e; i++) {
                    size_t nb = off[i];
                    for (int m = 0; m < 2; m++) {
                        const Rec &rec = m ? rb_[i] : ra_[i];
                        basal_rawread &w = raw[2 * i + m];
                        memset(&w, 0, sizeof w);
                        const size_t sl = strlen(rec.seq.data()), ql = rec.has_qual ? strlen(rec.qual.data()) : 0;
                        w.name_off = (uint32_t)nb; w.name_len = (uint16_t)std::min<size_t>(rec.name.size(), 0xffff);
                        memcpy(blob + nb, rec.name.data(), w.name_len); nb += w.name_len;
                        w.seq_off = (uint32_t)nb; w.seq_len = (uint16_t)sl;
                        memcpy(blob + nb, rec.seq.data(), sl); nb += sl;
                        w.qual_off = (uint32_t)nb; w.qual_len = (uint16_t)ql;
                        memcpy(blob + nb, rec.qual.data(), ql); nb += ql;
                        w.readset = (uint8_t)(m ? 2 : 1); w.index = rec.index;
                    }
                }
            });
            {
                std::lock_guard<std::mutex> lk(qm);
                submitted.push_back({0, 0, 0});
            }
            if (basal_pipe_submit_records(pipe, off[np], (uint32_t)(2 * np))) {
                std::lock_guard<std::mutex> lk(qm);
                submitted.pop_back();
                break;
            }
        }
        stop.store(true);
        qa.close(); qb.close(); qfree.close();
        ta.join();
        tb.join();
        producer_done = true;
    };

    std::thread pt;
    if (plain) pt = std::thread(producer_text);
    else pt = std::thread(producer_records, (uint64_t)0, (uint64_t)0, 0u, true);
    double t_wait = 0, ms[5] = {0, 0, 0, 0, 0};
    for (;;) {
        const void *data = nullptr;
        uint64_t nbytes = 0;
        basal_batch_stats bs;
        const double w0 = now();
        int rc = basal_pipe_collect(pipe, &data, &nbytes, &bs);
        t_wait += now() - w0;
        bool fall_back = false;
        if (rc == BASAL_ESTATE) {  // nothing in flight
            if (producer_done) {
                rc = basal_pipe_collect(pipe, &data, &nbytes, &bs);  // (the producer may have submitted its last batch between our collect and this test)
                if (rc == BASAL_ESTATE) {
                    if (plain && producer_err == "FALLBACK") fall_back = true;
                    else break;
                }
            } else { std::this_thread::sleep_for(std::chrono::microseconds(200)); continue; }
        }
        if ((rc == BASAL_EIO && plain) || fall_back) {
            // irregular text: stop the producer, drop what is in flight, go on with the host's parser from this batch's first bytes
            stop = true;
            basal_pipe_stop(pipe);
            pt.join();
            BatchInfo bi{0, 0, 0};
            {
                std::lock_guard<std::mutex> lk(qm);
                if (!submitted.empty()) bi = submitted.front();
                submitted.clear();
            }
            if (basal_pipe_rewind(pipe)) die(basal_last_error());
            if (cli.verbose >= 1)
                fprintf(stderr, "[BASAL-MI355X] mate files are not 4 regular lines per read from read pair %u on: parsing them on the host\n", bi.first_index);
            plain = false;
            stop = false;
            producer_done = false;
            producer_err.clear();
            pt = std::thread(producer_records, bi.off_a, bi.off_b, bi.first_index, false);
            continue;
        }
        if (rc) die(std::string("pipeline: ") + basal_last_error());
        {
            std::lock_guard<std::mutex> lk(qm);
            if (!submitted.empty()) submitted.pop_front();
        }
        out.write((const char *)data, (size_t)nbytes);
        for (int k = 0; k < 9; k++) pst[k] += bs.pe[k];
        n_pairs += bs.n_reads / 2;
        t_gpu += (bs.ms_prep + bs.ms_align + bs.ms_format) * 1e-3;
        ms[0] += bs.ms_h2d; ms[1] += bs.ms_prep; ms[2] += bs.ms_align; ms[3] += bs.ms_format; ms[4] += bs.ms_d2h;
        if (cli.verbose >= 2) fprintf(stderr, "[BASAL-MI355X] %llu read pairs finished.\n", (unsigned long long)n_pairs);
    }
    pt.join();
    if (!producer_err.empty() && producer_err != "FALLBACK") die(producer_err);
    if (cli.verbose >= 1)
        fprintf(stderr, "\tGPU stage sums: H2D %.3f, read prep %.3f, align %.3f, pairing + SAM %.3f, D2H %.3f s; host side: waiting for results %.3f s\n", ms[0] / 1e3, ms[1] / 1e3,
                ms[2] / 1e3, ms[3] / 1e3, ms[4] / 1e3, t_wait);
}

void run_pe(Cli &cli, Aligner &al, basal_ref_t *R, Output &out, uint32_t pst[9], uint64_t &n_pairs, double &t_gpu) {
    basal_params &P = cli.P;
    const int threads = cli.threads;
    // 131 072 pairs per batch by default: the three stages (two readers | QC, descriptors, GPU | text, writing) overlap batch by batch, and on
    // 4 M pairs this size ran 5.6 M pairs/s against 3.2 at 524 288 pairs and 4.0 at 65 536 (the GPU's share grows as launches shrink)
    const size_t batch = cli.batch ? cli.batch : (1u << 18);
    Reader ra, rb;
    if (!ra.open(cli.qa.c_str())) die("failed to open read file (check -a option): " + cli.qa);
    if (!rb.open(cli.qb.c_str())) die("failed to open read file #2 (check -b option): " + cli.qb);
    skip_reads(ra, P, cli.read_start, 1);
    skip_reads(rb, P, cli.read_start, 1);
    // PairAlign::Do_Batch (pairs.cpp:179-202): both mates in one GPU batch (a0,b0,a1,b1,...), pairing on the host
    std::vector<Rec> ra_, rb_;
    std::vector<uint8_t> bases;
    std::vector<basal_read> descs;
    std::vector<basal_result> results;
    std::vector<basal_hit> stream;
    std::vector<basal_stale> stales;
    basal_stale_tracker_t *tracker = basal_host_stale_new(&P);
    uint8_t carry[2][2] = {{0, 0}, {0, 0}};
    const bool device_pairing = al.multi == nullptr && !getenv("BASAL_PE_HOST_PAIRING");
    std::vector<basal_pe_pair> pe_pairs;
    std::vector<basal_pe_rec> pe_recs;
    double tm[6] = {0, 0, 0, 0, 0, 0};  // waiting for the readers, QC, descriptors, GPU, text, write
    const double t_begin = now();
    // Two reader threads, one per mate file, run ahead of the aligning thread by up to two batches. Mate 1's reader decides a batch's size
    // (a batch's bases must stay below 4 GiB -- 32-bit offsets -- so the base budget may close one early) and tells mate 2's reader.
    BatchQueue<std::vector<Rec>> qa, qb, qfree;  // qfree: batch vectors the aligning thread is done with, for the readers to fill again
    BatchQueue<size_t> qn;
    qn.cap = 4;
    qfree.cap = 8;
    auto recycled = [&](std::vector<Rec> &v) {
        std::lock_guard<std::mutex> l(qfree.m);
        if (qfree.q.empty()) return;
        v = std::move(qfree.q.front());
        qfree.q.pop_front();
    };
    std::atomic<bool> stop_readers{false};
    // (reads are cut to max_readlen when they are parsed, so below ~3.9 M pairs per batch the base budget cannot close a batch early and
    // mate 2's reader need not wait to be told the size)
    const bool same_size = (batch / 2 + 1) * (size_t)(P.max_readlen + 2) < ((size_t)1800 << 20);
    std::thread ta([&] {
        while (!stop_readers.load()) {
            std::vector<Rec> v;
            recycled(v);
            const int n1 = load_batch(ra, P, cli.read_end, batch / 2 + 1, 1, v, (size_t)1800 << 20);
            size_t n = (size_t)n1;
            if (!same_size) qn.push(std::move(n));
            if (!n1) break;
            qa.push(std::move(v));
        }
        qa.close();
    });
    std::thread tb([&] {
        size_t n1 = batch / 2 + 1;
        while (!stop_readers.load() && (same_size || (qn.pop(n1) && n1))) {
            std::vector<Rec> v;
            recycled(v);
            load_batch(rb, P, cli.read_end, n1, 2, v);
            if (same_size && v.empty()) break;
            const bool short_file = v.size() != n1;
            qb.push(std::move(v));
            if (short_file) break;
        }
        qb.close();
    });
    // Third stage (device pairing): the records of a batch are turned into text and written while the next batch is prepared and aligned.
    struct PeOut { std::vector<Rec> a, b; std::vector<basal_pe_pair> pairs; std::vector<basal_pe_rec> recs; };
    BatchQueue<PeOut> qfmt, qpool;  // qpool: pair/record vectors to use again
    std::thread tf([&] {
        PeOut po;
        while (qfmt.pop(po)) {
            const size_t np = po.a.size();
            double f0 = now();
            std::vector<std::string> chunks((size_t)std::max(threads, 1));
            parallel_for(np, threads, [&](size_t b, size_t e, int tid) {
                std::vector<char> line(1 << 16);
                for (size_t i = b; i < e; i++) {
                    const Rec &x = po.a[i], &y = po.b[i];
                    basal_mate ma{x.name.c_str(), x.seq.data(), x.qual.data(), 1, x.index, x.max_snp, x.qc_failed, nullptr};
                    basal_mate mb{y.name.c_str(), y.seq.data(), y.qual.data(), 2, y.index, y.max_snp, y.qc_failed, nullptr};
                    if (po.pairs[i].status) die("align_pairs_batch: a pair's records did not fit");
                    size_t need = 8192 + (size_t)(po.pairs[i].n + 2) * (2048 + 2 * (x.seq.size() + y.seq.size()));
                    if (line.size() < need) line.resize(need);
                    int64_t w = basal_host_format_pe_records(&P, R, &ma, &mb, po.recs.data() + po.pairs[i].first, po.pairs[i].n, line.data(), line.size());
                    if (w < 0) die(std::string("format_pe_records: ") + basal_last_error());
                    chunks[(size_t)tid].append(line.data(), (size_t)w);
                }
            });
            tm[4] += now() - f0;
            f0 = now();
            for (auto &c : chunks) out.write(c.data(), c.size());
            tm[5] += now() - f0;
            qfree.try_push(std::move(po.a));
            qfree.try_push(std::move(po.b));
            po.a.clear(); po.b.clear();
            qpool.try_push(std::move(po));
        }
    });
    for (;;) {
        double q0 = now();
        if (!ra_.empty()) qfree.try_push(std::move(ra_));
        if (!rb_.empty()) qfree.try_push(std::move(rb_));
        ra_.clear(); rb_.clear();
        const bool ga = qa.pop(ra_), gb = ga && qb.pop(rb_);
        tm[0] += now() - q0;
        if (ga && gb && !ra_.empty() && ra_.size() != rb_.size() && cli.verbose >= 1)
            fprintf(stderr, "[BASAL-MI355X] warning: the mate files do not hold the same number of reads; stopping at read pair %u\n", ra_[0].index);
        if (!ga || !gb || ra_.empty() || ra_.size() != rb_.size()) break;
        const size_t np = ra_.size();
        q0 = now();
        parallel_for(np, threads, [&](size_t b, size_t e, int) {
            for (size_t i = b; i < e; i++) {
                ra_[i].qc_failed = basal_host_filter_read(&P, ra_[i].seq.data(), ra_[i].qual.data(), &ra_[i].max_snp);
                rb_[i].qc_failed = basal_host_filter_read(&P, rb_[i].seq.data(), rb_[i].qual.data(), &rb_[i].max_snp);
            }
        });
        tm[1] += now() - q0;
        q0 = now();
        descs.assign(2 * np, basal_read{});
        bases.clear();
        stales.clear();
        basal_host_stale_begin_batch(tracker);
        parallel_for(np, threads, [&](size_t b, size_t e, int) {  // FixPairReadName (pairs.cpp:487-507), pair by pair
            std::vector<char> na, nb;
            for (size_t i = b; i < e; i++) {
                na.assign(ra_[i].name.begin(), ra_[i].name.end()); nb.assign(rb_[i].name.begin(), rb_[i].name.end());
                na.push_back(0); nb.push_back(0);
                if (basal_host_fix_pair_names(na.data(), nb.data())) die(basal_last_error());
                ra_[i].name = na.data(); rb_[i].name = nb.data();
            }
        });
        for (size_t i = 0; i < np; i++) {
            const bool both = !ra_[i].qc_failed && !rb_[i].qc_failed;
            for (int m = 0; m < 2; m++) {
                Rec &rc_ = m ? rb_[i] : ra_[i];
                basal_read &d = descs[2 * i + m];
                d.index = rc_.index;
                d.readset = (uint8_t)((m ? 2 : 1) | (both ? BASAL_READ_ALLMODES : 0));
                d.stale_idx = BASAL_STALE_NONE;
                if (rc_.qc_failed) { d.len = 0; continue; }
                uint32_t len = (uint32_t)strlen(rc_.seq.data());
                d.len = (uint16_t)len;
                d.max_snp = (uint8_t)rc_.max_snp;
                d.seq_off = (uint32_t)bases.size();
                bases.insert(bases.end(), rc_.seq.begin(), rc_.seq.begin() + len);
                basal_stale se;
                if (basal_host_stale_visit(tracker, rc_.seq.data(), len, m ? 2 : 1, 0, (uint32_t)(2 * i + m), &se)) {
                    d.stale_idx = (uint32_t)stales.size();
                    stales.push_back(se);
                }
            }
        }
        tm[2] += now() - q0;
        if (device_pairing) {
            // one GPU: the pairing rounds run on the device too (basal_pe.hip); what comes back is the list of records to print
            pe_pairs.resize(np);
            uint64_t cap = pe_recs.size() > 2 * np + 4096 ? pe_recs.size() : 2 * np + 4096, used = 0;
            double g0 = now();
            uint32_t st9[9] = {0};
            for (int attempt = 0;; attempt++) {
                pe_recs.resize(cap);
                uint8_t cy[2][2];
                memcpy(cy, carry, 4);
                memset(st9, 0, sizeof st9);
                int rc = basal_core_align_pairs_batch(al.core, bases.data(), bases.size(), descs.data(), (uint32_t)np, stales.data(), (uint32_t)stales.size(), pe_pairs.data(),
                                                      pe_recs.data(), cap, &used, st9, cy);
                if (rc == BASAL_EOVERFLOW && attempt < 4) { cap = std::max(used, cap) + used / 8 + 4096; continue; }
                if (rc) die(std::string("align_pairs_batch: ") + basal_last_error());
                memcpy(carry, cy, 4);
                break;
            }
            for (int k = 0; k < 9; k++) pst[k] += st9[k];
            t_gpu += now() - g0;
            tm[3] += now() - g0;
            {   // hand the batch to the text stage; its vectors come back through qfree / qpool
                PeOut po;
                po.a = std::move(ra_); po.b = std::move(rb_); po.pairs = std::move(pe_pairs); po.recs = std::move(pe_recs);
                ra_.clear(); rb_.clear();
                qfmt.push(std::move(po));
                PeOut spare;
                {
                    std::lock_guard<std::mutex> l(qpool.m);
                    if (!qpool.q.empty()) { spare = std::move(qpool.q.front()); qpool.q.pop_front(); }
                }
                pe_pairs = std::move(spare.pairs);
                pe_recs = std::move(spare.recs);
            }
            n_pairs += np;
            continue;
        }
        results.assign(2 * np, basal_result{});
        uint64_t cap = 16 * np + 4096, used = 0;
        double g0 = now();
        for (int attempt = 0;; attempt++) {
            stream.resize(cap);
            uint8_t cy[2][2];
            memcpy(cy, carry, 4);
            int rc = al.run(bases.data(), bases.size(), descs.data(), (uint32_t)(2 * np), stales.data(), (uint32_t)stales.size(), BASAL_STREAM_ALL, results.data(),
                            stream.data(), cap, &used, cy);
            if (rc == BASAL_EOVERFLOW && attempt < 4) { cap = std::max(used, cap) + 4096; continue; }
            if (rc) die(std::string("align_batch: ") + basal_last_error());
            memcpy(carry, cy, 4);
            break;
        }
        t_gpu += now() - g0;
        std::vector<std::string> chunks((size_t)std::max(threads, 1));
        std::vector<uint32_t> st((size_t)std::max(threads, 1) * 9, 0);
        parallel_for(np, threads, [&](size_t b, size_t e, int tid) {
            std::vector<char> line(1 << 16);
            for (size_t i = b; i < e; i++) {
                basal_mate ma{ra_[i].name.c_str(), ra_[i].seq.data(), ra_[i].qual.data(), 1, ra_[i].index, ra_[i].max_snp, ra_[i].qc_failed, &results[2 * i]};
                basal_mate mb{rb_[i].name.c_str(), rb_[i].seq.data(), rb_[i].qual.data(), 2, rb_[i].index, rb_[i].max_snp, rb_[i].qc_failed, &results[2 * i + 1]};
                size_t need = 8192 + (size_t)(results[2 * i].stream_n + results[2 * i + 1].stream_n + 2 + P.max_num_hits * 2) * (1024 + 2 * (ra_[i].seq.size() + rb_[i].seq.size()));
                if (line.size() < need) line.resize(need);
                int64_t w = basal_host_format_pe(&P, R, &ma, &mb, stream.data(), line.data(), line.size(), &st[9 * (size_t)tid]);
                if (w < 0) die(std::string("format_pe: ") + basal_last_error());
                chunks[(size_t)tid].append(line.data(), (size_t)w);
            }
        });
        for (auto &c : chunks) out.write(c.data(), c.size());
        for (size_t t = 0; t < chunks.size(); t++) for (int k = 0; k < 9; k++) pst[k] += st[9 * t + k];
        n_pairs += np;
    }
    // (a mate file that ended early, or an error above: let the readers run out)
    qfmt.close();  // (pop hands out what is still queued before it reports the end)
    tf.join();
    stop_readers.store(true);
    qa.close(); qb.close(); qn.close(); qfree.close();
    ta.join();
    tb.join();
    if (cli.verbose >= 1)
        fprintf(stderr, "\thost side (%.3f s): waiting for the two reader threads %.3f s, QC %.3f s, descriptors %.3f s, GPU batches %.3f s, SAM text %.3f s, writing %.3f s\n",
                now() - t_begin, tm[0], tm[1], tm[2], tm[3], tm[4], tm[5]);
    basal_host_stale_free(tracker);
    // The batch vectors hold three heap blocks per read, ten million of them by now; the process is about to exit and the run's clock is
    // still running, so they are left to the OS instead of being freed one by one (half a second per 4 M reads).
    if (!getenv("BASAL_CLEAN_EXIT")) {
        new std::vector<Rec>(std::move(ra_));
        new std::vector<Rec>(std::move(rb_));
        new std::deque<std::vector<Rec>>(std::move(qfree.q));
        new std::deque<std::vector<Rec>>(std::move(qa.q));
        new std::deque<std::vector<Rec>>(std::move(qb.q));
    }
    // (n_pairs: the pairs of the batches that were aligned and written -- not the mate-1 reader's position, which runs ahead and counts a
    // last batch that mate 2's file no longer matched)
    ra.close();
    rb.close();
}

}  // namespace

int main(int argc, char **argv) {
    // HIP maps streams of one priority onto a few hardware queues (4 by default), and two streams on one queue run in submission order. The batch
    // pipeline alternates consecutive batches' kernels between two streams so that the end of one align launch (a few waves finishing its longest
    // reads) overlaps the next batch's kernels: they need queues of their own. Read by the runtime when it starts, i.e. before the first HIP call.
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    Cli cli;
    basal_params &P = cli.P;
    basal_host_params_defaults(&P);
    cli.cmdline = argv[0];
    cli.cpu_index = getenv("BASAL_CPU_INDEX") != nullptr;
    for (int i = 1; i < argc; i++) cli.cmdline += std::string(" ") + argv[i];
    if (argc == 1) die("Usage: basal -a reads.fq [-b mates.fq] -d ref.fa -M C:T [options]   (options as in BASAL 1.8.1)");
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (a[0] != '-') die(std::string("unknown option: ") + a, i);
        char f = a[1];
        const char *v = nullptr;
        if (strchr("RHu3N", f)) {
            if (a[2]) die(std::string("unknown option: ") + a, i);
        } else if (a[2] == 0) {
            if (i + 1 >= argc) die(std::string("missing value for ") + a, i);
            v = argv[++i];
        } else if (a[2] == '=') v = a + 3;
        else die(std::string("unknown option: ") + a, i);
        switch (f) {
            case 'a': cli.qa = v; break;
            case 'b': cli.qb = v; P.pairend = 1; break;
            case 'd': cli.ref_file = v; break;
            case 's': if (basal_host_params_set_seed_size(&P, atoi(v))) die(basal_last_error()); break;
            case 'o': cli.out_file = v; break;
            case 'M': cli.rule = v; break;
            case 'm': P.min_insert = (uint32_t)atoi(v); break;
            case 'n': P.chains = (uint32_t)atoi(v); break;
            case 'g': P.gap = (uint32_t)atoi(v); if (P.gap > BASAL_MAXGAPS) { fprintf(stderr, "warning: gap length exceeds max value:%d\n", BASAL_MAXGAPS); P.gap = BASAL_MAXGAPS; } break;
            case 'x': P.max_insert = (uint32_t)atoi(v); break;
            case 'r': P.report_repeat_hits = (uint32_t)atoi(v); if (P.report_repeat_hits > 2) die("invalid -r value, must be 0, 1, or 2."); break;
            case 'V': cli.verbose = atoi(v); break;
            case 'I': P.index_interval = (uint32_t)atoi(v); if (P.index_interval > 16 || P.index_interval < 1) die("index interval exceeds max value:16"); break;
            case 'k': P.max_kmer_ratio = (float)atof(v); break;
            case 'v': basal_host_params_set_v(&P, atof(v)); break;
            case 'w': P.max_num_hits = (uint32_t)atoi(v); if (P.max_num_hits > BASAL_MAXHITS) die("number of multi-hits exceeds max value:1000"); break;
            case 'q': P.trim_qual_threshold = (uint32_t)atoi(v); break;
            case 'f': P.max_ns = (uint32_t)atoi(v); break;
            case 'z': P.zero_qual = (uint8_t)atoi(v); break;
            case 'p': cli.threads = atoi(v); break;
            case 'A': if (P.n_adapter < 10) { strncpy(P.adapter[P.n_adapter], v, 127); P.n_adapter++; } break;
            case 'R': P.out_ref = 1; break;
            case 'H': cli.sam_header = 0; break;
            case 'u': P.out_unmap = 1; break;
            case 'B': cli.read_start = (uint32_t)std::max(atoi(v), 1); break;
            case 'E': cli.read_end = (uint32_t)atoi(v); break;
            case 'L': P.max_readlen = (uint32_t)atoi(v); if (P.max_readlen > BASAL_MAXREADLEN) P.max_readlen = BASAL_MAXREADLEN; break;
            case 'N': P.n_mis = 1; break;
            case 'S': P.randseed = (uint32_t)atoi(v); break;
            case '3': die("-3 (3-nucleotide mode) is not supported by the MI355X build");
            case 'D': die("-D (RRBS digestion sites) is not supported by the MI355X build");
            case 'G': {  // extension: HIP device ordinal, or a list 0,1,2,... to shard every batch's reads over several GPUs
                cli.devices.clear();
                for (const char *q = v; *q;) { cli.devices.push_back(atoi(q)); while (*q && *q != ',') q++; if (*q == ',') q++; }
                if (cli.devices.empty()) die("-G needs a device ordinal or a comma-separated list of them");
                cli.device = cli.devices[0];
                break;
            }
            case 'Z': cli.batch = (size_t)atol(v); break;  // extension: reads per GPU batch
            case 'h': die("see the BASAL 1.8.1 usage text; this build accepts the same options");
            default: die(std::string("unknown option: ") + a, i);
        }
    }
    if (cli.rule.empty()) die("\n-M option is required");
    if (basal_host_params_set_align(&P, cli.rule.c_str())) die(basal_last_error());
    if (cli.ref_file.empty() || cli.qa.empty()) die("-a and -d are required");
    if (cli.threads < 1) cli.threads = 1;

    double t0 = now();
    {   // the reference reports an unreadable -d file before anything else happens; so does this, before the GPU is asked for
        FILE *f = fopen(cli.ref_file.c_str(), "rb");
        if (!f) die("failed to open reference file (check -d option): " + cli.ref_file);
        fclose(f);
    }
    basal_core_t *core = nullptr;
    basal_multi_t *multi = nullptr;
    std::vector<basal_core_t *> cores;  // single-end on several GPUs: one core per GPU behind ONE batch pipeline (basal_pipe_create_multi)
    if (cli.devices.empty()) cli.devices.push_back(cli.device);
    const bool several = cli.devices.size() > 1 || getenv("BASAL_FORCE_MULTI") != nullptr;  // (the variable: the several-GPU paths on a one-GPU box, for tests)
    // Whole batches fan out over the GPUs, each through the device-side pipeline (text or records in, SAM out; single- and paired-end); a GPU
    // may be listed twice (two cores on one GPU: how the tests drive two ranks on a one-GPU box). BASAL_MULTI_HOST=1: the batch is sharded over
    // the GPUs instead and the records come back through one RCCL gather (basal_multi_*: host-side QC and SAM text).
    // (paired-end with BASAL_PE_HOST_PAIRING / BASAL_PE_NO_PIPE keeps round 2's host-side path, which shards over several GPUs through basal_multi_*)
    const bool se_pipes = !getenv("BASAL_MULTI_HOST") && !(P.pairend && (getenv("BASAL_PE_HOST_PAIRING") || getenv("BASAL_PE_NO_PIPE")));
    if (several && !se_pipes) {
        if (basal_multi_create(&P, cli.devices.data(), (int)cli.devices.size(), &multi)) die(std::string("cannot set up the GPUs: ") + basal_last_error());
        core = basal_multi_core(multi, 0);
    } else {
        for (int dv : cli.devices) {
            basal_core_t *c1 = nullptr;
            if (basal_core_create(&P, dv, &c1)) die(std::string("cannot create the GPU core: ") + basal_last_error());
            cores.push_back(c1);
            if (!several) break;
        }
        core = cores[0];
    }
    // single-end: the pipeline's buffers are page-locked by a helper thread while the reference is read and staged
    SePlan plan;
    basal_pipe_t *pipe = nullptr;
    std::thread pipe_thread;
    std::string pipe_err;
    double t_pipe = 0;
    // paired-end on one GPU: the same pipeline in its paired-end mode (records in, the pairs' SAM text out); BASAL_PE_HOST_PAIRING /
    // BASAL_PE_NO_PIPE keep round 2's path (device or host pairing, text on the host) as the cross-check
    const bool pe_pipe = P.pairend && !multi && !getenv("BASAL_PE_HOST_PAIRING") && !getenv("BASAL_PE_NO_PIPE");
    if (!P.pairend && !multi) {
        plan = plan_se(cli);
        pipe_thread = std::thread([&] {
            const double a0 = now();
            if (basal_pipe_create_multi(cores.data(), (int)cores.size(), &plan.po, &pipe)) pipe_err = basal_last_error();
            t_pipe = now() - a0;
        });
    } else if (pe_pipe) {
        memset(&plan.po, 0, sizeof plan.po);
        const size_t pairs_per_batch = cli.batch ? std::max<size_t>(cli.batch / 2, 1) : (1u << 17);  // 131 072 pairs: the stages overlap from the first second on
        plan.po.depth = 3;
        plan.po.max_reads = (uint32_t)((2 * pairs_per_batch + 4095) & ~(size_t)4095);
        plan.po.max_bytes = std::min<uint64_t>((uint64_t)plan.po.max_reads * (2ull * P.max_readlen + 300) + (1u << 20), 0xF0000000ull);
        plan.po.output = BASAL_PIPE_OUT_SAM;
        plan.po.flags = BASAL_PIPE_PAIRS;
        pipe_thread = std::thread([&] {
            const double a0 = now();
            if (basal_pipe_create_multi(cores.data(), (int)cores.size(), &plan.po, &pipe)) pipe_err = basal_last_error();
            t_pipe = now() - a0;
        });
    }
    if (cli.verbose >= 1) fprintf(stderr, "[BASAL-MI355X] loading reference file: %s\n", cli.ref_file.c_str());
    basal_ref_t *R = nullptr;
    if (basal_host_ref_load(&P, cli.ref_file.c_str(), &R)) die(basal_last_error());
    double t1 = now();
    uint32_t mk = 0;
    if (cli.cpu_index && basal_host_ref_build_index(R, &P, cli.threads)) die(basal_last_error());
    const uint32_t nc_names = basal_host_ref_ncontig(R);
    std::vector<const char *> names(nc_names);
    for (uint32_t i = 0; i < nc_names; i++) names[i] = basal_host_ref_name(R, i);
    if (multi) {
        if (basal_multi_upload(multi, R, cli.cpu_index ? 0 : 1, &mk)) die(basal_last_error());
        if (basal_core_set_contig_names(core, names.data(), nc_names)) die(basal_last_error());
    } else {  // every GPU stages the reference and builds its index at the same time, one host thread each
        std::vector<std::string> errs(cores.size());
        std::vector<uint32_t> mks(cores.size(), 0);
        std::vector<std::thread> th;
        for (size_t g = 0; g < cores.size(); g++)
            th.emplace_back([&, g] {
                if (basal_host_ref_upload(R, cores[g], cli.cpu_index ? 0 : 1, &mks[g]) || basal_core_set_contig_names(cores[g], names.data(), nc_names)) errs[g] = basal_last_error();
            });
        for (auto &t : th) t.join();
        for (auto &e : errs) if (!e.empty()) die(e);
        mk = mks[0];
    }
    double t2 = now();
    if (cli.verbose >= 1)
        fprintf(stderr, "[BASAL-MI355X] %u reference seqs loaded in %.2f s; seed table (%s) in %.2f s, over-represented k-mer cut-off %u\n",
                basal_host_ref_ncontig(R), t1 - t0, cli.cpu_index ? "CPU" : "GPU", t2 - t1, mk);

    // -o x.bam pipes SAM through an external `samtools view -bS -`, like the reference (main.cpp:504-513)
    Output out;
    out.threads = std::max(1, std::min(cli.threads, 16));
    if (!cli.out_file.empty()) {
        if (cli.out_file.size() > 4 && cli.out_file.compare(cli.out_file.size() - 4, 4, ".bam") == 0) {
            std::string cmd = "samtools view -bS - >" + cli.out_file;
            out.fo = popen(cmd.c_str(), "w");
            out.piped = out.fo != nullptr;
            if (!out.fo) fprintf(stderr, "unable to creat samtools pipe, writing SAM instead.\n");
        }
        if (!out.piped) {
            out.fd = open(cli.out_file.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (out.fd < 0) die("failed to open output file (check -o option): " + cli.out_file);
            struct stat sb;
            out.regular = fstat(out.fd, &sb) == 0 && S_ISREG(sb.st_mode);
            if (out.regular) { g_out_fd = out.fd; signal(SIGBUS, on_sigbus); }
            if (!out.regular) { out.fo = fdopen(out.fd, "w"); if (!out.fo) die("failed to open output file (check -o option): " + cli.out_file); }
        }
    }
    if (cli.sam_header) {
        std::vector<char> hb(64 + cli.cmdline.size() + 160 * (size_t)basal_host_ref_ncontig(R) + 4096);
        int64_t n = basal_host_sam_header(R, cli.cmdline.c_str(), hb.data(), hb.size());
        if (n < 0) die(basal_last_error());
        out.write(hb.data(), (size_t)n);
    }

    double t3 = now();
    if (P.pairend) {
        uint32_t pst[9] = {0};
        uint64_t n_pairs = 0;
        double t_gpu = 0;
        Aligner al{multi ? nullptr : core, multi};
        if (pe_pipe) {
            pipe_thread.join();
            if (!pipe) die("cannot create the pipeline: " + pipe_err);
        }
        const double tp0 = now();
        if (pe_pipe) run_pe_pipe(cli, pipe, plan.po, out, pst, n_pairs, t_gpu);
        else run_pe(cli, al, R, out, pst, n_pairs, t_gpu);
        out.close();
        if (cli.verbose >= 1) {
            const double tp1 = now();
            fprintf(stderr, "[BASAL-MI355X] total read pairs: %llu \ttotal time:  %.2f secs (align phase %.3f s = %.2f Mpairs/s; GPU batches %.3f s)\n", (unsigned long long)n_pairs,
                    tp1 - t0, tp1 - tp0, n_pairs / (tp1 - tp0) / 1e6, t_gpu);
            fprintf(stderr, "\taligned pairs: %u, unique pairs: %u, non-unique pairs: %u\n\tunpaired read #1: %u, unique: %u, non-unique: %u\n\tunpaired read #2: %u, unique: %u, non-unique: %u\n",
                    pst[0], pst[1], pst[2], pst[3], pst[4], pst[5], pst[6], pst[7], pst[8]);
        }
    } else {
        SeStats st;
        double t_wait = 0;
        if (multi) {
            Aligner al{nullptr, multi};
            run_se_host(cli, al, R, out, st);
        } else {
            pipe_thread.join();
            if (!pipe) die("cannot create the pipeline: " + pipe_err);
            st.t_create = t_pipe;
            t3 = now();
            run_se(cli, pipe, plan, mk >= 32768 && !getenv("BASAL_HEAVY"), out, st, t_wait);
        }
        out.close();
        double t4 = now();
        if (cli.verbose >= 1) {
            const uint64_t tot = st.n_reads;
            fprintf(stderr, "[BASAL-MI355X] total reads: %llu \ttotal time:  %.2f secs (align phase %.3f s = %.2f Mreads/s; GPU stage sums: H2D %.3f, read prep %.3f, align %.3f, SAM %.3f, D2H %.3f s)\n",
                    (unsigned long long)tot, t4 - t0, t4 - t3, tot / (t4 - t3) / 1e6, st.ms[0] / 1e3, st.ms[1] / 1e3, st.ms[2] / 1e3, st.ms[3] / 1e3, st.ms[4] / 1e3);
            fprintf(stderr, "\thost side: pipeline set-up %.3f s (beside the reference load), reading %.3f s, waiting for a free batch slot %.3f s, waiting for results %.3f s, writing %.3f s\n", st.t_create, st.t_read,
                    st.t_acquire, t_wait, st.t_write);
            fprintf(stderr, "\taligned reads: %llu (%.1f%%), unique reads: %llu (%.1f%%), %snon-unique reads: %llu (%.1f%%)\n", (unsigned long long)st.n_aligned,
                    100.0 * st.n_aligned / (tot ? tot : 1), (unsigned long long)st.n_unique, 100.0 * st.n_unique / (tot ? tot : 1),
                    P.report_repeat_hits == 0 ? "suppressed " : "", (unsigned long long)st.n_multiple, 100.0 * st.n_multiple / (tot ? tot : 1));
        }
    }
    fflush(stdout);
    fflush(stderr);
    if (!getenv("BASAL_CLEAN_EXIT")) _exit(0);  // everything is written: leave the gigabytes of page-locked and device memory to the OS
    if (pipe) basal_pipe_destroy(pipe);
    if (multi) basal_multi_destroy(multi);
    else for (auto *c1 : cores) basal_core_destroy(c1);
    basal_host_ref_free(R);
    return 0;
}
