// basal_pe.hip -- paired-end pairing on the device: PairAlign::RunAlign's pairing rounds (SortHits4PE + GetPairs,
// pairs.cpp:29-177, align.cpp:412-416) and the choices of StringAlignPair / StringAlignUnpair (pairs.cpp:204-305), one thread per read pair.
//
// The align kernel has run every SnpAlign mode of both mates (BASAL_READ_ALLMODES) and left each mate's hit log, in insertion order, every
// record tagged with level, chain and the mode that stored it (BASAL_STREAM_ALL).  A mate's history does not depend on its partner, so the
// reference's loop
//     for i: _sa.SnpAlign(i); _sb.SnpAlign(i); SortHits4PE(i) x 2; GetPairs(i,i); GetPairs(i,j) + GetPairs(j,i) for j < i; stop at the first pair
// is replayed over the logs: the hits of (chain, level) form one array whose first `cur` elements are those of modes <= i (the log is in mode
// order); round i sorts the current prefix of the level-i arrays -- with libstdc++'s std::sort restated step by step, because HitComp
// (utilities.cpp:51-53) compares (chr, loc) only and an ungapped and a gapped hit can tie: their order is then whatever introsort leaves --
// and joins mate 1's arrays with mate 2's opposite chain by contig and insert size.  The output is a list of records to PRINT, in order
// (pair records, unpaired-mate records); the host only turns them into text (basal_host_format_pe_records).
#include <hip/hip_runtime.h>

#include <string>

#include "basal_bits.h"
#include "basal_core_priv.h"
#include "basal_internal.h"

using namespace basal;

namespace {

#define HIP_TRYE(x)                                                    \
    do {                                                               \
        hipError_t e_ = (x);                                           \
        if (e_ != hipSuccess) {                                        \
            set_error(std::string(#x) + ": " + hipGetErrorString(e_)); \
            return BASAL_EDEVICE;                                      \
        }                                                              \
    } while (0)

struct PeConst {
    uint32_t min_insert, max_insert, max_num_hits, report_repeat_hits, randseed, out_unmap;
};

__device__ __forceinline__ bool hit_less(const basal_hit &a, const basal_hit &b) { return (a.chr < b.chr) || ((a.chr == b.chr) && (a.loc < b.loc)); }
__device__ __forceinline__ void hswap(basal_hit &a, basal_hit &b) { const basal_hit t = a; a = b; b = t; }

// ---- std::sort as libstdc++ performs it: introsort (median-of-3 quicksort down to runs of 16, heap sort past the depth limit), then insertion sort
__device__ void unguarded_linear_insert(basal_hit *last) {
    const basal_hit val = *last;
    basal_hit *next = last - 1;
    while (hit_less(val, *next)) { *last = *next; last = next; --next; }
    *last = val;
}
__device__ void insertion_sort(basal_hit *first, basal_hit *last) {
    if (first == last) return;
    for (basal_hit *i = first + 1; i != last; ++i) {
        if (hit_less(*i, *first)) {
            const basal_hit val = *i;
            for (basal_hit *q = i; q != first; --q) *q = *(q - 1);
            *first = val;
        } else unguarded_linear_insert(i);
    }
}
__device__ void sift_down(basal_hit *a, long start, long n) {
    long root = start;
    for (;;) {
        long child = 2 * root + 1;
        if (child >= n) break;
        if (child + 1 < n && hit_less(a[child], a[child + 1])) child++;
        if (hit_less(a[root], a[child])) { hswap(a[root], a[child]); root = child; } else break;
    }
}
__device__ void heap_sort(basal_hit *a, long n) {
    for (long s = n / 2 - 1; s >= 0; s--) sift_down(a, s, n);
    for (long e = n - 1; e > 0; e--) { hswap(a[0], a[e]); sift_down(a, 0, e); }
}
__device__ void std_sort_hits(basal_hit *first, uint32_t n) {
    if (n < 2) return;
    basal_hit *const end = first + n;
    if (n > 16) {
        long lg = 0;
        for (uint32_t t = n; t > 1; t >>= 1) lg++;
        // introsort_loop, its recursion on an explicit stack (the right part is pushed, the loop continues on the left one)
        struct Frame { basal_hit *first, *last; long depth; } stack[32];
        int sp = 0;
        stack[sp++] = Frame{first, end, lg * 2};
        while (sp > 0) {
            Frame f = stack[--sp];
            basal_hit *fi = f.first, *la = f.last;
            long depth = f.depth;
            while (la - fi > 16) {
                if (depth == 0) { heap_sort(fi, la - fi); break; }
                --depth;
                basal_hit *mid = fi + (la - fi) / 2, *a = fi + 1, *b = mid, *c = la - 1;
                if (hit_less(*a, *b)) {
                    if (hit_less(*b, *c)) hswap(*fi, *b);
                    else if (hit_less(*a, *c)) hswap(*fi, *c);
                    else hswap(*fi, *a);
                } else if (hit_less(*a, *c)) hswap(*fi, *a);
                else if (hit_less(*b, *c)) hswap(*fi, *c);
                else hswap(*fi, *b);
                basal_hit *lo = fi + 1, *hi = la;
                for (;;) {
                    while (hit_less(*lo, *fi)) ++lo;
                    --hi;
                    while (hit_less(*fi, *hi)) --hi;
                    if (!(lo < hi)) break;
                    hswap(*lo, *hi);
                    ++lo;
                }
                // introsort_loop(lo, last, depth) runs first in the reference; the two parts are disjoint, so the order does not matter
                if (sp < 32) stack[sp++] = Frame{lo, la, depth};
                la = lo;
            }
        }
        insertion_sort(first, first + 16);
        for (basal_hit *i = first + 16; i != end; ++i) unguarded_linear_insert(i);
    } else insertion_sort(first, end);
}

// one mate's hits as 32 arrays (chain, level) inside its own region of the work buffer
struct MateView {
    basal_hit *w;            // the mate's region of the work buffer (stream_n records)
    uint16_t off[2][16];     // start of array (chain, level)
    uint16_t tot[2][16];     // its final length
    uint16_t cur[2][16];     // elements of modes <= the current round
    uint32_t len, max_snp, index, failed, n_log;
    const basal_hit *log;
};

__device__ void load_mate(MateView &M, const basal_read &rd, const basal_result &rs, const basal_hit *stream, basal_hit *work) {
    M.len = rd.len; M.max_snp = rd.max_snp; M.index = rd.index; M.failed = rd.len == 0;
    M.n_log = 0; M.log = nullptr; M.w = nullptr;
    for (int c = 0; c < 2; c++) for (int l = 0; l < 16; l++) M.off[c][l] = M.tot[c][l] = M.cur[c][l] = 0;
    if (M.failed || rs.best_level == 0xFF || rs.status == BASAL_READ_OVERFLOW) return;
    M.log = stream + rs.stream_first;
    M.w = work + rs.stream_first;
    M.n_log = rs.stream_n;
    for (uint32_t k = 0; k < M.n_log; k++) { const basal_hit h = M.log[k]; if (h.level <= BASAL_MAXSNPS) M.tot[h.chain & 1][h.level]++; }
    uint32_t at = 0;
    for (int c = 0; c < 2; c++) for (int l = 0; l < 16; l++) { M.off[c][l] = (uint16_t)at; at += M.tot[c][l]; }
    uint16_t fill[2][16];
    for (int c = 0; c < 2; c++) for (int l = 0; l < 16; l++) fill[c][l] = 0;
    for (uint32_t k = 0; k < M.n_log; k++) {  // stream order inside every array = the order PairAlign appends them, mode by mode
        const basal_hit h = M.log[k];
        if (h.level > BASAL_MAXSNPS) continue;
        M.w[M.off[h.chain & 1][h.level] + fill[h.chain & 1][h.level]++] = h;
    }
}
// the arrays' current lengths for round i: elements of modes <= i (a prefix: the log is in mode order)
__device__ void advance_round(MateView &M, uint32_t i) {
    for (int c = 0; c < 2; c++)
        for (int l = 0; l < 16; l++) {
            uint32_t n = M.cur[c][l];
            const basal_hit *a = M.w + M.off[c][l];
            while (n < M.tot[c][l] && a[n].mode <= i) n++;
            M.cur[c][l] = (uint16_t)n;
        }
}

// GetPairs (pairs.cpp:29-130) over the current arrays: counts the pairs it would store at summed level na + nb (the list already holds
// `have`), and hands the k-th element of that level's list (counted from `have`) to the sink when asked for
struct PairSink {
    uint32_t want_level;  // only pairs of this summed level are emitted
    uint32_t want_lo, want_hi;  // list positions [lo, hi) of that level
    basal_pe_rec *out;    // where they go (consecutive)
    uint32_t n_total;     // the `n` printed with every pair record
};
__device__ uint32_t get_pairs(const PeConst &P, const MateView &A, const MateView &B, uint32_t na, uint32_t nb, uint32_t have, PairSink *sink) {
    if (na > A.max_snp || nb > B.max_snp) return 0;
    uint32_t npair = 0, size = have;
    for (uint32_t chain = 0; chain < 2; chain++) {
        const basal_hit *av = A.w + A.off[chain][na], *bv = B.w + B.off[chain ^ 1][nb];
        const uint32_t an = A.cur[chain][na], bn = B.cur[chain ^ 1][nb];
        uint32_t chra = ~0u, bstart = 0, bend = 0;
        for (uint32_t i = 0; i < an; i++) {
            const basal_hit ha = av[i];
            if (chra != ha.chr) {
                chra = ha.chr;
                for (bstart = bend; bstart < bn; bstart++) if (bv[bstart].chr >= chra) break;
                for (bend = bstart; bend < bn; bend++) if (bv[bend].chr > chra) break;
            }
            for (uint32_t j = bstart; j < bend; j++) {
                const basal_hit hb = bv[j];
                uint32_t s, e;
                const bool a_left = chain == 0 ? !(chra & 1) : (chra & 1);
                if (a_left) { s = ha.loc; e = hb.loc + B.len; }
                else { s = hb.loc; e = ha.loc + A.len; }
                const uint32_t ins = e - s;
                if (ins >= P.min_insert && ins <= P.max_insert) {
                    if (sink && na + nb == sink->want_level && size >= sink->want_lo && size < sink->want_hi) {
                        basal_pe_rec r;
                        r.kind = BASAL_PE_PAIR; r.side = 0; r.chain_a = (uint8_t)chain; r.chain_b = 0;
                        r.ma = (int32_t)sink->n_total; r.na = na; r.mb = (int32_t)nb; r.insert = ins; r.ha = ha; r.hb = hb;
                        sink->out[size - sink->want_lo] = r;
                    }
                    size++;
                    npair++;
                    if (size >= P.max_num_hits) return npair;
                }
            }
        }
    }
    return npair;
}

// the pairing rounds of round i (pairs.cpp:170-171): per summed level the number of pairs added, in the reference's order of calls
__device__ uint32_t pairing_round(const PeConst &P, const MateView &A, const MateView &B, uint32_t i, uint32_t cnt[2 * BASAL_MAXSNPS + 1], PairSink *sink) {
    uint32_t n = 0, a;
    a = get_pairs(P, A, B, i, i, cnt[2 * i], sink); cnt[2 * i] += a; n += a;
    for (uint32_t j = 0; j < i; j++) {
        a = get_pairs(P, A, B, i, j, cnt[i + j], sink); cnt[i + j] += a; n += a;
        a = get_pairs(P, A, B, j, i, cnt[i + j], sink); cnt[i + j] += a; n += a;
    }
    return n;
}

__device__ void emit_unpair(basal_pe_rec *out, uint32_t side, uint32_t chain_a, uint32_t chain_b, int ma, uint32_t na, const basal_hit &ha, int mb, const basal_hit &hb) {
    basal_pe_rec r;
    r.kind = BASAL_PE_UNPAIR; r.side = (uint8_t)side; r.chain_a = (uint8_t)chain_a; r.chain_b = (uint8_t)chain_b;
    r.ma = ma; r.na = na; r.mb = mb; r.insert = 0; r.ha = ha; r.hb = hb;
    *out = r;
}

__global__ __launch_bounds__(64) void pair_kernel(PeConst P, const basal_read *__restrict__ reads, const basal_result *__restrict__ results,
                                                   const basal_hit *__restrict__ stream, basal_hit *__restrict__ work, uint32_t npairs,
                                                   basal_pe_pair *__restrict__ pairs, basal_pe_rec *__restrict__ recs, unsigned long long recs_cap,
                                                   unsigned long long *__restrict__ recs_used, unsigned int *__restrict__ stats) {
    const uint32_t pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= npairs) return;
    MateView A, B;
    load_mate(A, reads[2 * pi], results[2 * pi], stream, work);
    load_mate(B, reads[2 * pi + 1], results[2 * pi + 1], stream, work);
    basal_pe_pair pr;
    pr.first = 0; pr.n = 0; pr.status = 0;
    if (results[2 * pi].status == BASAL_READ_OVERFLOW || results[2 * pi + 1].status == BASAL_READ_OVERFLOW) {  // the hit stream was too small: the batch is redone
        pr.status = BASAL_READ_OVERFLOW;
        pairs[pi] = pr;
        return;
    }
    const bool both = !A.failed && !B.failed;
    uint32_t cnt[2 * BASAL_MAXSNPS + 1];
    for (uint32_t l = 0; l <= 2 * BASAL_MAXSNPS; l++) cnt[l] = 0;
    uint32_t paired_round = 0xFFFFFFFFu;
    if (both) {  // PairAlign::RunAlign, pairs.cpp:161-176
        const uint32_t maxi = A.max_snp > B.max_snp ? A.max_snp : B.max_snp;
        for (uint32_t i = 0; i <= maxi; i++) {
            advance_round(A, i);
            advance_round(B, i);
            if (i <= A.max_snp) for (int c = 0; c < 2; c++) std_sort_hits(A.w + A.off[c][i], A.cur[c][i]);
            if (i <= B.max_snp) for (int c = 0; c < 2; c++) std_sort_hits(B.w + B.off[c][i], B.cur[c][i]);
            if (pairing_round(P, A, B, i, cnt, nullptr) > 0) { paired_round = i; break; }
        }
    } else {  // one mate failed QC: the other went through SingleAlign::RunAlign -- every array complete, in insertion order, never sorted
        advance_round(A, 0xFFu);
        advance_round(B, 0xFFu);
    }
    // what will be printed: count first
    uint32_t level = 0, sum = 0, n_pair_recs = 0;
    bool pair_reported = false;
    if (paired_round != 0xFFFFFFFFu) {  // StringAlignPair, pairs.cpp:204-230
        for (level = 0; level <= 2 * BASAL_MAXSNPS; level++) if ((sum = cnt[level]) > 0) break;
        if (sum == 1) { n_pair_recs = 1; pair_reported = true; }
        else if (sum > 1 && P.report_repeat_hits == 1) { n_pair_recs = 1; pair_reported = true; }
        else if (sum > 1 && P.report_repeat_hits == 2) { n_pair_recs = sum; pair_reported = true; }
    }
    // StringAlignUnpair's view of the two mates (pairs.cpp:232-305)
    int ma = 0, mb = 0;
    uint32_t na = 0, nb = 0, ca = 0, cb = 0;
    basal_hit ha, hb;
    memset(&ha, 0, sizeof ha);
    memset(&hb, 0, sizeof hb);
    uint32_t n_unpair_recs = 0;
    int ma1 = 0, mb1 = 0;
    if (!pair_reported) {
        if (A.failed) ma = -1;
        else {
            for (na = 0; na <= A.max_snp; na++) if ((ma = (int)(A.cur[0][na] + A.cur[1][na])) > 0) break;
            if (ma > 0) { const uint32_t ra = myrand(A.index, P.randseed) % (uint32_t)ma; ca = ra >= A.cur[0][na]; ha = ca ? A.w[A.off[1][na] + ra - A.cur[0][na]] : A.w[A.off[0][na] + ra]; }
            na %= (A.max_snp + 1);
        }
        if (B.failed) mb = -1;
        else {
            for (nb = 0; nb <= B.max_snp; nb++) if ((mb = (int)(B.cur[0][nb] + B.cur[1][nb])) > 0) break;
            if (mb > 0) { const uint32_t rb = myrand(B.index, P.randseed) % (uint32_t)mb; cb = rb >= B.cur[0][nb]; hb = cb ? B.w[B.off[1][nb] + rb - B.cur[0][nb]] : B.w[B.off[0][nb] + rb]; }
            nb %= (B.max_snp + 1);
        }
        ma1 = (ma > 1 && P.report_repeat_hits == 0) ? 0 : ma;
        mb1 = (mb > 1 && P.report_repeat_hits == 0) ? 0 : mb;
        auto count_side = [&](int m) -> uint32_t {
            if (m <= 0) return P.out_unmap ? 1u : 0u;
            if (m == 1 || P.report_repeat_hits == 1) return 1u;
            if (P.report_repeat_hits == 2) return (uint32_t)m;
            return P.out_unmap ? 1u : 0u;
        };
        n_unpair_recs = count_side(ma) + count_side(mb);
    }
    const uint32_t total = n_pair_recs + n_unpair_recs;
    unsigned long long first = 0;
    if (total) first = atomicAdd(recs_used, (unsigned long long)total);
    pr.first = (uint32_t)first;
    pr.n = total;
    if (first + total > recs_cap) { pr.status = BASAL_READ_OVERFLOW; pairs[pi] = pr; return; }
    basal_pe_rec *out = recs + first;
    // statistics (pairs.cpp's counters: aligned / unique / multiple for pairs, mate 1, mate 2)
    if (paired_round != 0xFFFFFFFFu) {
        if (sum == 1) { atomicAdd(&stats[1], 1u); atomicAdd(&stats[0], 1u); }
        else if (sum > 1) { atomicAdd(&stats[2], 1u); if (P.report_repeat_hits) atomicAdd(&stats[0], 1u); }
    }
    if (pair_reported) {
        // enumerate the final round again, now emitting the wanted positions of the best level's list
        uint32_t lo = 0, hi = sum;
        if (sum > 1 && P.report_repeat_hits == 1) { lo = myrand(A.index, P.randseed) % sum; hi = lo + 1; }
        PairSink sink{level, lo, hi, out, sum};
        uint32_t cnt2[2 * BASAL_MAXSNPS + 1];
        for (uint32_t l = 0; l <= 2 * BASAL_MAXSNPS; l++) cnt2[l] = 0;
        pairing_round(P, A, B, paired_round, cnt2, &sink);
    } else {
        basal_pe_rec *o = out;
        // mate 1's records, then mate 2's
        if (ma <= 0) { if (P.out_unmap) emit_unpair(o++, 0, 0, cb, ma, 0, ha, mb1, hb); }
        else if (ma == 1) { atomicAdd(&stats[3], 1u); atomicAdd(&stats[4], 1u); emit_unpair(o++, 0, ca, cb, 1, na, ha, mb1, hb); }
        else {
            atomicAdd(&stats[5], 1u);
            if (P.report_repeat_hits == 1) { atomicAdd(&stats[3], 1u); emit_unpair(o++, 0, ca, cb, ma, na, ha, mb1, hb); }
            else if (P.report_repeat_hits == 2) {
                atomicAdd(&stats[3], 1u);
                for (uint32_t c = 0; c < 2; c++) for (uint32_t k = 0; k < A.cur[c][na]; k++) emit_unpair(o++, 0, c, cb, ma, na, A.w[A.off[c][na] + k], mb1, hb);
            } else if (P.out_unmap) emit_unpair(o++, 0, 0, cb, 0, 0, ha, mb1, hb);
        }
        if (mb <= 0) { if (P.out_unmap) emit_unpair(o++, 1, 0, ca, mb, 0, hb, ma1, ha); }
        else if (mb == 1) { atomicAdd(&stats[6], 1u); atomicAdd(&stats[7], 1u); emit_unpair(o++, 1, cb, ca, 1, nb, hb, ma1, ha); }
        else {
            atomicAdd(&stats[8], 1u);
            if (P.report_repeat_hits == 1) { atomicAdd(&stats[6], 1u); emit_unpair(o++, 1, cb, ca, mb, nb, hb, ma1, ha); }
            else if (P.report_repeat_hits == 2) {
                atomicAdd(&stats[6], 1u);
                // (the reference passes cb, mate 2's own chain, as the OTHER mate's chain in these two loops: pairs.cpp:296-297)
                for (uint32_t c = 0; c < 2; c++) for (uint32_t k = 0; k < B.cur[c][nb]; k++) emit_unpair(o++, 1, c, cb, mb, nb, B.w[B.off[c][nb] + k], ma1, ha);
            } else if (P.out_unmap) emit_unpair(o++, 1, 0, ca, 0, 0, hb, ma1, ha);
        }
    }
    pairs[pi] = pr;
}

}  // namespace

// Queue the pairing of the npairs read pairs whose 2 * npairs mates (a0, b0, a1, b1, ...) were aligned into d_results / d_stream.
int basal_pe_enqueue(basal_core *c, const void *d_reads, const void *d_results, const void *d_stream, void *d_work, uint32_t npairs, void *d_pairs, void *d_recs,
                     uint64_t recs_cap, void *d_recs_used, void *d_stats, hipStream_t s) {
    if (npairs == 0) return BASAL_OK;
    PeConst P;
    P.min_insert = c->p.min_insert; P.max_insert = c->p.max_insert; P.max_num_hits = c->p.max_num_hits; P.report_repeat_hits = c->p.report_repeat_hits;
    P.randseed = c->p.randseed; P.out_unmap = c->p.out_unmap;
    hipLaunchKernelGGL(pair_kernel, dim3((npairs + 63) / 64), dim3(64), 0, s, P, (const basal_read *)d_reads, (const basal_result *)d_results, (const basal_hit *)d_stream,
                       (basal_hit *)d_work, npairs, (basal_pe_pair *)d_pairs, (basal_pe_rec *)d_recs, (unsigned long long)recs_cap, (unsigned long long *)d_recs_used,
                       (unsigned int *)d_stats);
    HIP_TRYE(hipGetLastError());
    return BASAL_OK;
}
