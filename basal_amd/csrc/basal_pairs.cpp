// basal_pairs.cpp -- paired-end on the host: pairing rounds and pair/unpair SAM records (pairs.cpp).
//
// The GPU aligns each mate on its own (every SnpAlign mode, hits tagged with the mode that stored
// them); a mate's hit history does not depend on its partner, so PairAlign::RunAlign's loop
//     for i: _sa.SnpAlign(i); _sb.SnpAlign(i); SortHits4PE(i) x2; GetPairs(i,i); GetPairs(i,j)+GetPairs(j,i) for j<i; stop at first pair
// is replayed here over the logs: round i appends the mode-i hits to the per-chain, per-level arrays,
// sorts level i with std::sort + HitComp (the reference's own call, so ties fall the same way) and
// joins by (chr, insert size).  Everything after the round at which a pair appears never "happened".
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/basal_core.h"
#include "basal_bits.h"
#include "basal_internal.h"

using namespace basal;

namespace {

struct Buf {
    char *p;
    size_t cap, n;
    bool ok;
    void put(const char *s, size_t l) { if (!ok || n + l > cap) { ok = false; return; } memcpy(p + n, s, l); n += l; }
    void str(const char *s) { put(s, strlen(s)); }
    void ch(char c) { put(&c, 1); }
    void num(long long v) { char b[24]; int l = snprintf(b, sizeof b, "%lld", v); put(b, (size_t)l); }
};

char comp_char(char c) {
    switch (c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
        default: return 'N';
    }
}

bool hit_comp(const basal_hit &a, const basal_hit &b) { return (a.chr < b.chr) || ((a.chr == b.chr) && (a.loc < b.loc)); }  // HitComp, utilities.cpp:51-53

struct Mate {
    const basal_mate *m;
    uint32_t len, nseg;
    std::vector<basal_hit> h[2][BASAL_MAXSNPS + 1];  // hits[level], chits[level]
    uint32_t n(int c, uint32_t lvl) const { return (uint32_t)h[c][lvl].size(); }
};

struct PairHit {
    uint32_t chain, na, nb, insert;
    basal_hit a, b;
};

void put_seq(Buf &o, const char *seq, const char *qual, bool rev) {
    size_t len = strlen(seq), ql = strlen(qual);
    if (!rev) { o.put(seq, len); o.ch('\t'); o.put(qual, ql); return; }
    for (size_t i = 0; i < len; i++) o.ch(comp_char(seq[len - 1 - i]));
    o.ch('\t');
    for (size_t i = 0; i < ql; i++) o.ch(qual[ql - 1 - i]);
}

void put_cigar(Buf &o, const basal_hit &h, uint32_t len) {
    if (h.gap_size == 0) { o.num(len); o.ch('M'); }
    else if (h.gap_size > 0) { o.num(h.gap_pos); o.ch('M'); o.num(h.gap_size); o.ch('D'); o.num((int)len - (int)h.gap_pos); o.ch('M'); }
    else { o.num(h.gap_pos); o.ch('M'); o.num(-(int)h.gap_size); o.ch('I'); o.num((int)len - (int)h.gap_pos + (int)h.gap_size); o.ch('M'); }
}

}  // namespace

// defined in basal_host.cpp
namespace basal { void put_xr_field(char *out, size_t cap, size_t *n, bool *ok, const basal_params *p, const basal_ref_t *r, uint32_t contig, uint32_t loc, uint32_t len);
                  const char *ref_contig_name(const basal_ref_t *r, uint32_t contig); }

namespace {

void put_xr(Buf &o, const basal_params *p, const basal_ref_t *r, uint32_t contig, uint32_t loc, uint32_t len) {
    put_xr_field(o.p, o.cap, &o.n, &o.ok, p, r, contig, loc, len);
}

// s_OutHitPair, pairs.cpp:307-411
void out_pair(Buf &o, const basal_params *p, const basal_ref_t *r, const Mate &A, const Mate &B, const PairHit &pp, int n) {
    const bool rev_a = (pp.chain ^ (pp.a.chr & 1)) != 0, rev_b = ((!pp.chain) ^ (pp.b.chr & 1)) != 0;
    for (int side = 0; side < 2; side++) {
        const Mate &M = side ? B : A;
        const basal_hit &h = side ? pp.b : pp.a, &mh = side ? pp.a : pp.b;
        const bool rev = side ? rev_b : rev_a;
        uint32_t flag = 0x3;
        if (n > 1) flag |= 0x100;
        int ins;
        if (rev) { flag |= 0x10; ins = -(int)pp.insert; }
        else { flag |= 0x20; ins = (int)pp.insert; }
        flag |= 0x40 * M.m->readset;
        o.str(M.m->name); o.ch('\t'); o.num(flag); o.ch('\t'); o.str(ref_contig_name(r, h.chr >> 1)); o.ch('\t'); o.num((long long)h.loc + 1);
        o.str("\t255\t"); put_cigar(o, h, M.len); o.str("\t=\t"); o.num((long long)mh.loc + 1); o.ch('\t'); o.num(ins); o.ch('\t');
        put_seq(o, M.m->seq, M.m->qual, rev);
        o.str("\tNM:i:"); o.num((uint8_t)(side ? pp.nb : pp.na));
        if (p->out_ref) put_xr(o, p, r, h.chr >> 1, h.loc, M.len);
        o.str("\tZS:Z:"); o.ch((h.chr & 1) ? '-' : '+'); o.ch((side ? !pp.chain : pp.chain) ? '-' : '+'); o.ch('\n');
    }
}

// s_OutHitUnpair, pairs.cpp:418-485
void out_unpair(Buf &o, const basal_params *p, const basal_ref_t *r, const Mate &M, int chain_a, int chain_b, int ma, uint32_t na, const basal_hit &ha,
                int mb, const basal_hit &hb) {
    uint32_t flag = 1 | (0x40 * M.m->readset);
    const bool rev = (chain_a ^ (int)(ha.chr & 1)) != 0;
    if (ma <= 0) {
        if (ma < 0) flag |= 0x204;
        if (ma == 0) flag |= 0x004;
        if (mb <= 0) {
            flag |= 0x008;
            o.str(M.m->name); o.ch('\t'); o.num(flag); o.str("\t*\t0\t0\t*\t*\t0\t0\t"); o.str(M.m->seq); o.ch('\t'); o.str(M.m->qual); o.ch('\n');
        } else {
            if (chain_b ^ (int)(hb.chr & 1)) flag |= 0x020;
            o.str(M.m->name); o.ch('\t'); o.num(flag); o.str("\t*\t0\t0\t*\t"); o.str(ref_contig_name(r, hb.chr >> 1)); o.ch('\t'); o.num((long long)hb.loc + 1);
            o.str("\t0\t"); o.str(M.m->seq); o.ch('\t'); o.str(M.m->qual); o.ch('\n');
        }
        return;
    }
    if (ma > 1) flag |= 0x100;
    if (rev) flag |= 0x010;
    if (mb <= 0) flag |= 0x008;
    else if (chain_b ^ (int)(hb.chr & 1)) flag |= 0x020;
    o.str(M.m->name); o.ch('\t'); o.num(flag); o.ch('\t'); o.str(ref_contig_name(r, ha.chr >> 1)); o.ch('\t'); o.num((long long)ha.loc + 1); o.str("\t255\t");
    put_cigar(o, ha, M.len);
    if (mb <= 0) o.str("\t*\t0\t0\t");
    else { o.ch('\t'); o.str(ref_contig_name(r, hb.chr >> 1)); o.ch('\t'); o.num((long long)hb.loc + 1); o.str("\t0\t"); }
    put_seq(o, M.m->seq, M.m->qual, rev);
    o.str("\tNM:i:"); o.num((int)na);
    if (p->out_ref) put_xr(o, p, r, ha.chr >> 1, ha.loc, M.len);
    o.str("\tZS:Z:"); o.ch((ha.chr & 1) ? '-' : '+'); o.ch(chain_a ? '-' : '+'); o.ch('\n');
}

// GetPairs, pairs.cpp:29-130
int get_pairs(const basal_params *p, const Mate &A, const Mate &B, uint32_t na, uint32_t nb, std::vector<PairHit> *ph) {
    if (na > A.m->max_snp || nb > B.m->max_snp) return 0;
    int npair = 0;
    std::vector<PairHit> &dst = ph[na + nb];
    for (int chain = 0; chain < 2; chain++) {
        const std::vector<basal_hit> &av = A.h[chain][na], &bv = B.h[chain ^ 1][nb];
        uint32_t chra = ~0u, bstart = 0, bend = 0;
        for (size_t i = 0; i < av.size(); i++) {
            if (chra != av[i].chr) {
                chra = av[i].chr;
                for (bstart = bend; bstart < bv.size(); bstart++) if (bv[bstart].chr >= chra) break;
                for (bend = bstart; bend < bv.size(); bend++) if (bv[bend].chr > chra) break;
            }
            for (uint32_t j = bstart; j < bend; j++) {
                uint32_t s, e;
                const bool a_left = chain == 0 ? !(chra & 1) : (chra & 1);
                if (a_left) { s = av[i].loc; e = bv[j].loc + B.len; }
                else { s = bv[j].loc; e = av[i].loc + A.len; }
                uint32_t ins = e - s;
                if (ins >= p->min_insert && ins <= p->max_insert) {
                    dst.push_back(PairHit{(uint32_t)chain, na, nb, ins, av[i], bv[j]});
                    npair++;
                    if (dst.size() >= p->max_num_hits) return npair;
                }
            }
        }
    }
    return npair;
}

void load_mate(Mate &M, const basal_mate *m, const basal_params *p) {
    M.m = m;
    M.len = (uint32_t)strlen(m->seq);
    uint32_t x = (M.len - p->index_interval + 1) / p->seed_size, y = m->max_snp + 1;
    M.nseg = m->qc_failed ? 0 : (x < y ? x : y);
}

}  // namespace

extern "C" int basal_host_fix_pair_names(char *a, char *b) {
    if (strcmp(a, b) == 0) return 0;
    size_t la = strlen(a), lb = strlen(b), n = std::min(la, lb), i;
    int d = -1;
    for (i = 0; i < n; i++) {
        if (a[i] != b[i]) break;
        if (isdigit((unsigned char)a[i])) d = (int)i;
    }
    if (i == 0) { set_error(std::string("Error: Paired reads name not match:\n") + a + "\n" + b); return -1; }
    if (d < 0) d = (int)i - 1;
    if ((size_t)d + 1 < la) a[d + 1] = 0;
    if ((size_t)d + 1 < lb) b[d + 1] = 0;
    return 0;
}

extern "C" int64_t basal_host_format_pe_records(const basal_params *p, const basal_ref_t *r, const basal_mate *ma_, const basal_mate *mb_, const basal_pe_rec *recs,
                                                uint32_t n, char *out, size_t cap) {
    Buf o{out, cap, 0, true};
    Mate A, B;
    load_mate(A, ma_, p);
    load_mate(B, mb_, p);
    for (uint32_t k = 0; k < n; k++) {
        const basal_pe_rec &e = recs[k];
        if (e.kind == BASAL_PE_PAIR) {
            PairHit pp{e.chain_a, e.na, (uint32_t)e.mb, e.insert, e.ha, e.hb};
            out_pair(o, p, r, A, B, pp, e.ma);
        } else out_unpair(o, p, r, e.side ? B : A, e.chain_a, e.chain_b, e.ma, e.na, e.ha, e.mb, e.hb);
    }
    if (!o.ok) { set_error("format_pe_records: output buffer too small"); return BASAL_EOVERFLOW; }
    return (int64_t)o.n;
}

extern "C" int64_t basal_host_format_pe(const basal_params *p, const basal_ref_t *r, const basal_mate *ma_, const basal_mate *mb_, const basal_hit *stream,
                                        char *out, size_t cap, uint32_t stats[9]) {
    Buf o{out, cap, 0, true};
    Mate A, B;
    load_mate(A, ma_, p);
    load_mate(B, mb_, p);
    std::vector<PairHit> ph[2 * BASAL_MAXSNPS + 1];
    uint32_t st_local[9] = {0};
    uint32_t *st = stats ? stats : st_local;
    const bool both = !ma_->qc_failed && !mb_->qc_failed;
    auto log_of = [&](const basal_mate *m, const basal_hit *&first, uint32_t &n) {
        first = nullptr; n = 0;
        if (m->qc_failed || !m->res || m->res->best_level == 0xFF) return true;
        if (m->res->status == BASAL_READ_OVERFLOW) return false;
        first = stream + m->res->stream_first; n = m->res->stream_n;
        return true;
    };
    const basal_hit *la, *lb;
    uint32_t nla, nlb;
    if (!log_of(ma_, la, nla) || !log_of(mb_, lb, nlb)) { set_error("format_pe: hit stream overflowed for this pair"); return BASAL_EOVERFLOW; }
    int paired = 0;
    if (both) {
        // PairAlign::RunAlign, pairs.cpp:161-176
        uint32_t maxi = std::max(ma_->max_snp, mb_->max_snp), ia = 0, ib = 0, n = 0;
        for (uint32_t i = 0; i <= maxi && !paired; i++) {
            for (; ia < nla && la[ia].mode <= i; ia++) if (la[ia].level <= BASAL_MAXSNPS) A.h[la[ia].chain & 1][la[ia].level].push_back(la[ia]);
            for (; ib < nlb && lb[ib].mode <= i; ib++) if (lb[ib].level <= BASAL_MAXSNPS) B.h[lb[ib].chain & 1][lb[ib].level].push_back(lb[ib]);
            if (i <= ma_->max_snp) for (int c = 0; c < 2; c++) std::sort(A.h[c][i].begin(), A.h[c][i].end(), hit_comp);
            if (i <= mb_->max_snp) for (int c = 0; c < 2; c++) std::sort(B.h[c][i].begin(), B.h[c][i].end(), hit_comp);
            n += (uint32_t)get_pairs(p, A, B, i, i, ph);
            for (uint32_t j = 0; j < i; j++) n += (uint32_t)(get_pairs(p, A, B, i, j, ph) + get_pairs(p, A, B, j, i, ph));
            if (n > 0) paired = 1;
        }
    } else {
        // one mate failed QC: the other went through SingleAlign::RunAlign (insertion order, no sort)
        for (uint32_t k = 0; k < nla; k++) if (la[k].level <= BASAL_MAXSNPS) A.h[la[k].chain & 1][la[k].level].push_back(la[k]);
        for (uint32_t k = 0; k < nlb; k++) if (lb[k].level <= BASAL_MAXSNPS) B.h[lb[k].chain & 1][lb[k].level].push_back(lb[k]);
    }
    int pair_reported = 0;
    if (paired) {  // StringAlignPair, pairs.cpp:204-230
        uint32_t i = 0, sum = 0;
        for (; i <= 2 * BASAL_MAXSNPS; i++) if ((sum = (uint32_t)ph[i].size()) > 0) break;
        if (sum == 1) { st[1]++; st[0]++; out_pair(o, p, r, A, B, ph[i][0], 1); pair_reported = 1; }
        else if (sum > 1) {
            st[2]++;
            if (p->report_repeat_hits == 1) { st[0]++; out_pair(o, p, r, A, B, ph[i][myrand(ma_->index, p->randseed) % sum], (int)sum); pair_reported = 1; }
            else if (p->report_repeat_hits == 2) { st[0]++; for (uint32_t j = 0; j < sum; j++) out_pair(o, p, r, A, B, ph[i][j], (int)sum); pair_reported = 1; }
        }
    }
    if (!pair_reported || !paired) {  // StringAlignUnpair, pairs.cpp:232-305
        int ma = 0, mb = 0;
        uint32_t na = 0, nb = 0, ca = 0, cb = 0;
        basal_hit ha = {}, hb = {};
        if (ma_->qc_failed) ma = -1;
        else {
            for (na = 0; na <= ma_->max_snp; na++) if ((ma = (int)(A.n(0, na) + A.n(1, na))) > 0) break;
            if (ma > 0) { uint32_t ra = myrand(ma_->index, p->randseed) % (uint32_t)ma; ca = ra >= A.n(0, na); ha = ca ? A.h[1][na][ra - A.n(0, na)] : A.h[0][na][ra]; }
            na %= (ma_->max_snp + 1);
        }
        if (mb_->qc_failed) mb = -1;
        else {
            for (nb = 0; nb <= mb_->max_snp; nb++) if ((mb = (int)(B.n(0, nb) + B.n(1, nb))) > 0) break;
            if (mb > 0) { uint32_t rb = myrand(mb_->index, p->randseed) % (uint32_t)mb; cb = rb >= B.n(0, nb); hb = cb ? B.h[1][nb][rb - B.n(0, nb)] : B.h[0][nb][rb]; }
            nb %= (mb_->max_snp + 1);
        }
        const int ma1 = (ma > 1 && p->report_repeat_hits == 0) ? 0 : ma, mb1 = (mb > 1 && p->report_repeat_hits == 0) ? 0 : mb;
        if (ma <= 0) { if (p->out_unmap) out_unpair(o, p, r, A, 0, (int)cb, ma, 0, ha, mb1, hb); }
        else if (ma == 1) { st[3]++; st[4]++; out_unpair(o, p, r, A, (int)ca, (int)cb, 1, na, ha, mb1, hb); }
        else {
            st[5]++;
            if (p->report_repeat_hits == 1) { st[3]++; out_unpair(o, p, r, A, (int)ca, (int)cb, ma, na, ha, mb1, hb); }
            else if (p->report_repeat_hits == 2) {
                st[3]++;
                for (auto &h : A.h[0][na]) out_unpair(o, p, r, A, 0, (int)cb, ma, na, h, mb1, hb);
                for (auto &h : A.h[1][na]) out_unpair(o, p, r, A, 1, (int)cb, ma, na, h, mb1, hb);
            } else if (p->out_unmap) out_unpair(o, p, r, A, 0, (int)cb, 0, 0, ha, mb1, hb);
        }
        if (mb <= 0) { if (p->out_unmap) out_unpair(o, p, r, B, 0, (int)ca, mb, 0, hb, ma1, ha); }
        else if (mb == 1) { st[6]++; st[7]++; out_unpair(o, p, r, B, (int)cb, (int)ca, 1, nb, hb, ma1, ha); }
        else {
            st[8]++;
            if (p->report_repeat_hits == 1) { st[6]++; out_unpair(o, p, r, B, (int)cb, (int)ca, mb, nb, hb, ma1, ha); }
            else if (p->report_repeat_hits == 2) {
                st[6]++;
                for (auto &h : B.h[0][nb]) out_unpair(o, p, r, B, 0, (int)cb, mb, nb, h, ma1, ha);
                for (auto &h : B.h[1][nb]) out_unpair(o, p, r, B, 1, (int)cb, mb, nb, h, ma1, ha);
            } else if (p->out_unmap) out_unpair(o, p, r, B, 0, (int)ca, 0, 0, hb, ma1, ha);
        }
    }
    if (!o.ok) { set_error("format_pe: output buffer too small"); return BASAL_EOVERFLOW; }
    return (int64_t)o.n;
}
