// basal_main.cpp -- the `basal` command line on top of libbasal_amd.so.
//
// Same flags, messages-to-stderr and SAM-to-file behaviour as the reference driver (main.cpp:272-364
// option parser, 409-614 RunProcess, 60-92 batch loop), with SingleAlign::Do_Batch replaced by
// basal_core_align_batch on an MI355X.  Reads are parsed and QC-filtered on the host
// (reads.cpp:42-83, align.cpp:548-563), aligned on the GPU in large batches, and formatted back
// to SAM on the host in input order, so the output equals the reference's `-p 1` output.
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/basal_core.h"

namespace {

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Reader {  // whole file in memory; iostream-token semantics of ReadClass::LoadBatchReads
    std::string buf;
    size_t pos = 0;
    bool fastq = false;
    uint32_t index = 0;
    static bool ws(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }
    bool open(const char *path) {
        gzFile f = gzopen(path, "rb");
        if (!f) return false;
        gzbuffer(f, 1 << 20);
        std::vector<char> tmp(1 << 22);
        int got;
        while ((got = gzread(f, tmp.data(), (unsigned)tmp.size())) > 0) buf.append(tmp.data(), (size_t)got);
        gzclose(f);
        size_t p = 0;
        while (p < buf.size() && ws((unsigned char)buf[p])) p++;
        fastq = p < buf.size() && buf[p] == '@';
        // BAM (main.cpp:386-405 tries FASTA, FASTQ, then BAM): BGZF is multi-member gzip, so buf already holds the
        // uncompressed stream -- magic, header text, reference table, then the alignment records
        if (buf.size() >= 12 && memcmp(buf.data(), "BAM\1", 4) == 0) {
            bam = true;
            size_t q = 4;
            int32_t l_text = i32(q); q += 4 + (size_t)l_text;
            if (q + 4 > buf.size()) return false;
            int32_t n_ref = i32(q); q += 4;
            for (int32_t k = 0; k < n_ref && q + 4 <= buf.size(); k++) { int32_t l_name = i32(q); q += 4 + (size_t)l_name + 4; }
            if (q > buf.size()) return false;
            pos = q;
        }
        return true;
    }
    bool bam = false;
    int32_t i32(size_t at) const { int32_t v; memcpy(&v, buf.data() + at, 4); return v; }
    // next BAM record: [at, at + 4 + block_size); false at the end of the file
    bool bam_next(size_t &at) {
        if (pos + 4 > buf.size()) return false;
        int32_t bs = i32(pos);
        if (bs < 32 || pos + 4 + (size_t)bs > buf.size()) return false;
        at = pos + 4;
        pos += 4 + (size_t)bs;
        return true;
    }
    void skip_ws() { while (pos < buf.size() && ws((unsigned char)buf[pos])) pos++; }
    void token(size_t &b, size_t &l) {
        skip_ws();
        b = pos;
        while (pos < buf.size() && !ws((unsigned char)buf[pos])) pos++;
        l = pos - b;
    }
    void rest_of_line() {
        while (pos < buf.size() && buf[pos] != '\n') pos++;
        if (pos < buf.size()) pos++;
    }
};

struct Rec {
    uint32_t index, readset;
    std::string name;
    std::vector<char> seq, qual;  // NUL-terminated, trimmed in place by the filter
    int qc_failed = 0;
    uint32_t max_snp = 0;
};

// BAM records (reads.cpp:84-110): name, 4-bit bases, phred + 33; with -b the two mates alternate in one file
// (mate 1 reads a record and skips one, mate 2 skips one and reads), and flags 0x40 / 0x80 name the mate
int load_batch_bam(Reader &r, const basal_params &p, uint32_t read_end, size_t want, int readset, std::vector<Rec> &out) {
    static const char nt16[] = "=ACMGRSVTWYHKDBN";
    out.clear();
    for (; out.size() < want && r.index < read_end; r.index++) {
        size_t at, skip;
        if (readset == 2 && !r.bam_next(skip)) break;
        if (!r.bam_next(at)) break;
        const unsigned char *b = (const unsigned char *)r.buf.data() + at;
        uint32_t l_name = b[8], n_cigar, flag, l_qseq;
        uint16_t u16;
        memcpy(&u16, b + 12, 2); n_cigar = u16;
        memcpy(&u16, b + 14, 2); flag = u16;
        memcpy(&l_qseq, b + 16, 4);
        const char *name = (const char *)b + 32;
        const unsigned char *sq = b + 32 + l_name + 4 * n_cigar, *ql = sq + (l_qseq + 1) / 2;
        if (ql + l_qseq > (const unsigned char *)r.buf.data() + r.pos) break;  // truncated record
        uint32_t l = std::min<uint32_t>(l_qseq, p.max_readlen);
        Rec o;
        o.index = r.index;
        o.readset = readset ? ((flag & 0x40) ? 1u : (flag & 0x80) ? 2u : (uint32_t)readset) : 0u;
        o.name.assign(name, strnlen(name, l_name));
        o.seq.assign((size_t)l + 2, 0);
        o.qual.assign((size_t)l + 2, 0);
        for (uint32_t i = 0; i < l; i++) {
            o.seq[i] = nt16[(sq[i >> 1] >> ((~i & 1) << 2)) & 0xf];
            o.qual[i] = (char)(ql[i] + 33);
        }
        if (readset == 1 && !r.bam_next(skip)) break;  // (the reference drops a mate 1 without a following record, too)
        out.push_back(std::move(o));
    }
    return (int)out.size();
}

int load_batch(Reader &r, const basal_params &p, uint32_t read_end, size_t want, int readset, std::vector<Rec> &out) {
    if (r.bam) return load_batch_bam(r, p, read_end, want, readset, out);
    out.clear();
    for (; out.size() < want && r.index < read_end; r.index++) {
        r.skip_ws();
        if (r.pos >= r.buf.size()) break;
        r.pos++;
        size_t nb, nl, sb, sl, qb = 0, ql = 0, tb, tl;
        r.token(nb, nl);
        r.rest_of_line();
        r.token(sb, sl);
        if (r.fastq) {
            r.token(tb, tl);
            r.rest_of_line();
            r.token(qb, ql);
        }
        out.emplace_back();
        Rec &o = out.back();
        o.index = r.index;
        o.readset = (uint32_t)readset;
        o.name.assign(r.buf.data() + nb, nl);
        size_t cap = std::max(sl, ql) + 2;
        o.seq.assign(cap, 0);
        o.qual.assign(cap, 0);
        memcpy(o.seq.data(), r.buf.data() + sb, sl);
        if (r.fastq) memcpy(o.qual.data(), r.buf.data() + qb, ql);
        else memset(o.qual.data(), p.zero_qual + p.default_qual, sl);
        if (sl > p.max_readlen) {
            o.seq[p.max_readlen] = 0;
            if (strlen(o.qual.data()) > p.max_readlen) o.qual[p.max_readlen] = 0;
        }
    }
    return (int)out.size();
}

void die(const std::string &m, int code = 1) {
    fprintf(stderr, "%s\n", m.c_str());
    exit(code);
}

template <typename F>
void parallel_for(size_t n, int threads, F f) {
    if (threads <= 1 || n < 1024) { f(0, n, 0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(f, n * t / threads, n * (t + 1) / threads, t);
    for (auto &t : th) t.join();
}

}  // namespace

int main(int argc, char **argv) {
    basal_params P;
    basal_host_params_defaults(&P);
    std::string qa, qb, ref_file, out_file, rule, cmdline = argv[0];
    int threads = 1, verbose = 1, device = 0, sam_header = 1;
    uint32_t read_start = 1, read_end = ~0u;
    size_t batch = 1u << 20;
    bool cpu_index = getenv("BASAL_CPU_INDEX") != nullptr;
    for (int i = 1; i < argc; i++) cmdline += std::string(" ") + argv[i];
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
            case 'a': qa = v; break;
            case 'b': qb = v; P.pairend = 1; break;
            case 'd': ref_file = v; break;
            case 's': if (basal_host_params_set_seed_size(&P, atoi(v))) die(basal_last_error()); break;
            case 'o': out_file = v; break;
            case 'M': rule = v; break;
            case 'm': P.min_insert = (uint32_t)atoi(v); break;
            case 'n': P.chains = (uint32_t)atoi(v); break;
            case 'g': P.gap = (uint32_t)atoi(v); if (P.gap > BASAL_MAXGAPS) { fprintf(stderr, "warning: gap length exceeds max value:%d\n", BASAL_MAXGAPS); P.gap = BASAL_MAXGAPS; } break;
            case 'x': P.max_insert = (uint32_t)atoi(v); break;
            case 'r': P.report_repeat_hits = (uint32_t)atoi(v); if (P.report_repeat_hits > 2) die("invalid -r value, must be 0, 1, or 2."); break;
            case 'V': verbose = atoi(v); break;
            case 'I': P.index_interval = (uint32_t)atoi(v); if (P.index_interval > 16 || P.index_interval < 1) die("index interval exceeds max value:16"); break;
            case 'k': P.max_kmer_ratio = (float)atof(v); break;
            case 'v': basal_host_params_set_v(&P, atof(v)); break;
            case 'w': P.max_num_hits = (uint32_t)atoi(v); if (P.max_num_hits > BASAL_MAXHITS) die("number of multi-hits exceeds max value:1000"); break;
            case 'q': P.trim_qual_threshold = (uint32_t)atoi(v); break;
            case 'f': P.max_ns = (uint32_t)atoi(v); break;
            case 'z': P.zero_qual = (uint8_t)atoi(v); break;
            case 'p': threads = atoi(v); break;
            case 'A': if (P.n_adapter < 10) { strncpy(P.adapter[P.n_adapter], v, 127); P.n_adapter++; } break;
            case 'R': P.out_ref = 1; break;
            case 'H': sam_header = 0; break;
            case 'u': P.out_unmap = 1; break;
            case 'B': read_start = (uint32_t)std::max(atoi(v), 1); break;
            case 'E': read_end = (uint32_t)atoi(v); break;
            case 'L': P.max_readlen = (uint32_t)atoi(v); if (P.max_readlen > BASAL_MAXREADLEN) P.max_readlen = BASAL_MAXREADLEN; break;
            case 'N': P.n_mis = 1; break;
            case 'S': P.randseed = (uint32_t)atoi(v); break;
            case '3': die("-3 (3-nucleotide mode) is not supported by the MI355X build");
            case 'D': die("-D (RRBS digestion sites) is not supported by the MI355X build");
            case 'G': device = atoi(v); break;  // extension: HIP device ordinal
            case 'Z': batch = (size_t)atol(v); break;  // extension: reads per GPU batch
            case 'h': die("see the BASAL 1.8.1 usage text; this build accepts the same options");
            default: die(std::string("unknown option: ") + a, i);
        }
    }
    if (rule.empty()) die("\n-M option is required");
    if (basal_host_params_set_align(&P, rule.c_str())) die(basal_last_error());
    if (ref_file.empty() || qa.empty()) die("-a and -d are required");
    if (threads < 1) threads = 1;
    if (batch < 1) batch = 1;

    double t0 = now();
    if (verbose >= 1) fprintf(stderr, "[BASAL-MI355X] loading reference file: %s\n", ref_file.c_str());
    basal_ref_t *R = nullptr;
    if (basal_host_ref_load(&P, ref_file.c_str(), &R)) die(basal_last_error());
    double t1 = now();
    basal_core_t *core = nullptr;
    if (basal_core_create(&P, device, &core)) die(std::string("cannot create the GPU core: ") + basal_last_error());
    uint32_t mk = 0;
    if (cpu_index) {
        if (basal_host_ref_build_index(R, &P, threads)) die(basal_last_error());
        if (basal_host_ref_upload(R, core, 0, &mk)) die(basal_last_error());
    } else if (basal_host_ref_upload(R, core, 1, &mk)) die(basal_last_error());
    double t2 = now();
    if (verbose >= 1)
        fprintf(stderr, "[BASAL-MI355X] %u reference seqs loaded in %.2f s; seed table (%s) in %.2f s, over-represented k-mer cut-off %u\n",
                basal_host_ref_ncontig(R), t1 - t0, cpu_index ? "CPU" : "GPU", t2 - t1, mk);

    Reader ra, rb;
    if (!ra.open(qa.c_str())) die("failed to open read file (check -a option): " + qa);
    if (P.pairend && !rb.open(qb.c_str())) die("failed to open read file #2 (check -b option): " + qb);
    for (Reader *r : {&ra, &rb}) {  // ReadClass::InitIndex (reads.cpp:13-40)
        if (r->buf.empty()) continue;
        uint32_t maxi = (read_start - 1) * (2 + 2 * (uint32_t)r->fastq);
        for (uint32_t i = 0; i < maxi && r->pos < r->buf.size(); i++) r->rest_of_line();
        r->index = read_start - 1;
    }
    // -o x.bam pipes SAM through an external `samtools view -bS -`, like the reference (main.cpp:504-513)
    bool piped = false;
    FILE *fo = stdout;
    if (!out_file.empty()) {
        if (out_file.size() > 4 && out_file.compare(out_file.size() - 4, 4, ".bam") == 0) {
            std::string cmd = "samtools view -bS - >" + out_file;
            fo = popen(cmd.c_str(), "w");
            piped = fo != nullptr;
            if (!fo) fprintf(stderr, "unable to creat samtools pipe, writing SAM instead.\n");
        }
        if (!piped) fo = fopen(out_file.c_str(), "w");
    }
    if (!fo) die("failed to open output file (check -o option): " + out_file);
    if (sam_header) {
        std::vector<char> hb(64 + cmdline.size() + 128 * (size_t)basal_host_ref_ncontig(R) + 4096);
        int64_t n = basal_host_sam_header(R, cmdline.c_str(), hb.data(), hb.size());
        if (n < 0) die(basal_last_error());
        fwrite(hb.data(), 1, (size_t)n, fo);
    }

    uint64_t n_total = 0, n_aligned = 0, n_unique = 0, n_multiple = 0;
    double t_gpu = 0, t_host = 0, t3 = now();
    uint8_t carry[2][2] = {{0, 0}, {0, 0}};
    if (P.pairend) {
        // PairAlign::Do_Batch (pairs.cpp:179-202): both mates in one GPU batch (a0,b0,a1,b1,...), pairing on the host
        std::vector<Rec> ra_, rb_;
        std::vector<uint8_t> bases;
        std::vector<basal_read> descs;
        std::vector<basal_result> results;
        std::vector<basal_hit> stream;
        std::vector<basal_stale> stales;
        basal_stale_tracker_t *tracker = basal_host_stale_new(&P);
        uint32_t pst[9] = {0};
        for (;;) {
            int n1 = load_batch(ra, P, read_end, batch / 2 + 1, 1, ra_);
            int n2 = load_batch(rb, P, read_end, batch / 2 + 1, 2, rb_);
            if (!n1 || n1 != n2) break;
            const size_t np = (size_t)n1;
            parallel_for(np, threads, [&](size_t b, size_t e, int) {
                for (size_t i = b; i < e; i++) {
                    ra_[i].qc_failed = basal_host_filter_read(&P, ra_[i].seq.data(), ra_[i].qual.data(), &ra_[i].max_snp);
                    rb_[i].qc_failed = basal_host_filter_read(&P, rb_[i].seq.data(), rb_[i].qual.data(), &rb_[i].max_snp);
                }
            });
            descs.assign(2 * np, basal_read{});
            bases.clear();
            stales.clear();
            basal_host_stale_begin_batch(tracker);
            for (size_t i = 0; i < np; i++) {
                std::vector<char> na(ra_[i].name.begin(), ra_[i].name.end()), nb(rb_[i].name.begin(), rb_[i].name.end());
                na.push_back(0); nb.push_back(0);
                if (basal_host_fix_pair_names(na.data(), nb.data())) die(basal_last_error());
                ra_[i].name = na.data(); rb_[i].name = nb.data();
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
            results.assign(2 * np, basal_result{});
            uint64_t cap = 16 * np + 4096, used = 0;
            double g0 = now();
            for (;;) {
                stream.resize(cap);
                uint8_t cy[2][2];
                memcpy(cy, carry, 4);
                int rc = basal_core_align_batch(core, bases.data(), bases.size(), descs.data(), (uint32_t)(2 * np), stales.data(), (uint32_t)stales.size(),
                                                BASAL_STREAM_ALL, results.data(), stream.data(), cap, &used, cy);
                if (rc == BASAL_EOVERFLOW) { cap = used + 4096; continue; }
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
            for (auto &c : chunks) fwrite(c.data(), 1, c.size(), fo);
            for (size_t t = 0; t < chunks.size(); t++) for (int k = 0; k < 9; k++) pst[k] += st[9 * t + k];
            n_total += np;
        }
        if (verbose >= 1) {
            uint32_t tot = ra.index - read_start + 1;
            fprintf(stderr, "[BASAL-MI355X] total read pairs: %u \ttotal time:  %.2f secs (GPU batches %.3f s)\n", tot, now() - t0, t_gpu);
            fprintf(stderr, "\taligned pairs: %u, unique pairs: %u, non-unique pairs: %u\n\tunpaired read #1: %u, unique: %u, non-unique: %u\n\tunpaired read #2: %u, unique: %u, non-unique: %u\n",
                    pst[0], pst[1], pst[2], pst[3], pst[4], pst[5], pst[6], pst[7], pst[8]);
        }
        basal_host_stale_free(tracker);
    } else {
        std::vector<Rec> recs;
        std::vector<uint8_t> bases;
        std::vector<basal_read> descs;
        std::vector<basal_result> results;
        std::vector<basal_hit> stream;
        std::vector<basal_stale> stales;
        basal_stale_tracker_t *tracker = basal_host_stale_new(&P);
        const int smode = P.report_repeat_hits == 2 ? BASAL_STREAM_BEST : BASAL_STREAM_NONE;
        while (load_batch(ra, P, read_end, batch, 0, recs)) {
            double h0 = now();
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
            double g0 = now();
            for (;;) {
                stream.resize(cap ? cap : 1);
                uint8_t cy[2][2];
                memcpy(cy, carry, 4);
                int rc = basal_core_align_batch(core, bases.data(), bases.size(), descs.data(), (uint32_t)n, stales.data(), (uint32_t)stales.size(), smode, results.data(), stream.data(), cap, &used, cy);
                if (rc == BASAL_EOVERFLOW) { cap = used + 1024; continue; }
                if (rc) die(std::string("align_batch: ") + basal_last_error());
                memcpy(carry, cy, 4);
                break;
            }
            double g1 = now();
            t_gpu += g1 - g0;
            // format in input order; one output chunk per thread slice, written in slice order
            std::vector<std::string> chunks((size_t)std::max(threads, 1));
            std::vector<uint64_t> st((size_t)std::max(threads, 1) * 3, 0);
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
                        if (sum == 1) { st[3 * tid]++; st[3 * tid + 1]++; }
                        else { st[3 * tid + 2]++; if (P.report_repeat_hits) st[3 * tid]++; }
                    }
                }
            });
            for (auto &c : chunks) fwrite(c.data(), 1, c.size(), fo);
            for (size_t t = 0; t < chunks.size(); t++) { n_aligned += st[3 * t]; n_unique += st[3 * t + 1]; n_multiple += st[3 * t + 2]; }
            n_total += n;
            t_host += (g0 - h0) + (now() - g1);
            if (verbose >= 2) fprintf(stderr, "[BASAL-MI355X] %llu reads finished. %.2f secs passed\n", (unsigned long long)n_total, now() - t0);
        }
    }
    double t4 = now();
    if (piped) pclose(fo);
    else if (fo != stdout) fclose(fo);
    if (verbose >= 1 && !P.pairend) {
        uint32_t tot = ra.index - read_start + 1;
        fprintf(stderr, "[BASAL-MI355X] total reads: %u \ttotal time:  %.2f secs (align %.3f s: GPU batches %.3f s, host QC+SAM %.3f s)\n", tot, t4 - t0, t4 - t3, t_gpu, t_host);
        fprintf(stderr, "\taligned reads: %llu (%.1f%%), unique reads: %llu (%.1f%%), %snon-unique reads: %llu (%.1f%%)\n", (unsigned long long)n_aligned,
                100.0 * n_aligned / (tot ? tot : 1), (unsigned long long)n_unique, 100.0 * n_unique / (tot ? tot : 1), P.report_repeat_hits == 0 ? "suppressed " : "",
                (unsigned long long)n_multiple, 100.0 * n_multiple / (tot ? tot : 1));
    }
    basal_core_destroy(core);
    basal_host_ref_free(R);
    return 0;
}
