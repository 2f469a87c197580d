// basal_prep.hip -- everything the reference's host does to a read either side of the alignment, on the GPU:
//
//   text -> reads        ReadClass::LoadBatchReads for FASTQ / FASTA reads (reads.cpp:42-83): newline index of the batch's text,
//                        one raw read (name / bases / qualities as byte ranges of the text) per 4 (2) lines
//   FilterReads          align.cpp:548-563 with TrimAdapter (418-435), TrimLowQual (51-76), CountNs (40-47): the read descriptor
//                        the align kernel consumes, and what the SAM writer must print of the read
//   inherited state      what a read with (len - I + 1) % k == 0 inherits from EARLIER reads of the same SingleAlign object
//                        (align.cpp:475-480, align.h:73,90): the basal_stale table, computed from the batch itself plus a small
//                        carry state kept on the device from batch to batch
//   SAM text             StringAlign + s_OutHit (align.cpp:583-669): byte length of every read's records, exclusive scan, write
//
// All of it is byte/integer work over data that is already in HBM; nothing here touches the host.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstring>
#include <string>

#include "basal_bits.h"
#include "basal_internal.h"
#include "basal_prep.h"

namespace basal {

namespace {

#define HIP_TRYP(x)                                                    \
    do {                                                               \
        hipError_t e_ = (x);                                           \
        if (e_ != hipSuccess) {                                        \
            set_error(std::string(#x) + ": " + hipGetErrorString(e_)); \
            return BASAL_EDEVICE;                                      \
        }                                                              \
    } while (0)

__device__ __forceinline__ bool is_ws(uint32_t c) { return c == ' ' || (c >= 9 && c <= 13); }  // what iostream's >> skips

// Bytes of a global buffer served from aligned 16-byte chunks: a thread that walks a string byte by byte (in either direction)
// issues one load per 16 bytes instead of one per byte. The chunk holding a valid byte lies inside the buffer's allocation
// (hipMalloc aligns to 256 bytes and the buffers are padded).
struct Bytes16 {
    const uint8_t *base;
    unsigned long long cur = ~0ull;
    uint32_t w[4];
    __device__ __forceinline__ explicit Bytes16(const uint8_t *b) : base(b) {}
    __device__ __forceinline__ uint32_t get(uint32_t i) {
        const unsigned long long a = (unsigned long long)(base + i), c = a & ~15ull;
        if (c != cur) {
            const uint4 v = *(const uint4 *)c;
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
            cur = c;
        }
        const uint32_t k = (uint32_t)(a & 15u);
        const uint32_t d = k < 8 ? (k < 4 ? w[0] : w[1]) : (k < 12 ? w[2] : w[3]);
        return (d >> ((k & 3u) * 8)) & 0xffu;
    }
};

// ------------------------------------------------------------------------------------------------ text -> lines -> raw reads
constexpr int kTextBlock = 4096;  // bytes of text per 256-thread block, 16 per thread

__device__ __forceinline__ uint32_t newline_mask(const uint8_t *text, unsigned long long n, unsigned long long at) {
    uint32_t m = 0;
    if (at + 16 <= n) {
        const uint4 v = *(const uint4 *)(text + at);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int b = 0; b < 4; b++) m |= (uint32_t)(((w[k] >> (8 * b)) & 0xffu) == '\n') << (4 * k + b);
    } else
        for (int b = 0; b < 16 && at + b < n; b++) m |= (uint32_t)(text[at + b] == '\n') << b;
    return m;
}

__global__ __launch_bounds__(256) void nl_count(const uint8_t *__restrict__ text, unsigned long long n, uint32_t *__restrict__ blk_cnt) {
    typedef hipcub::BlockReduce<uint32_t, 256> Red;
    __shared__ typename Red::TempStorage tmp;
    const unsigned long long at = (unsigned long long)blockIdx.x * kTextBlock + threadIdx.x * 16;
    const uint32_t c = at < n ? (uint32_t)__popc(newline_mask(text, n, at)) : 0u;
    const uint32_t s = Red(tmp).Sum(c);
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void nl_write(const uint8_t *__restrict__ text, unsigned long long n, const uint32_t *__restrict__ blk_off, uint32_t nblk,
                                                uint32_t *__restrict__ nl, uint32_t cap, BatchCounters *__restrict__ cnt) {
    typedef hipcub::BlockScan<uint32_t, 256> Scan;
    __shared__ typename Scan::TempStorage tmp;
    const unsigned long long at = (unsigned long long)blockIdx.x * kTextBlock + threadIdx.x * 16;
    uint32_t m = at < n ? newline_mask(text, n, at) : 0u;
    uint32_t off, total;
    Scan(tmp).ExclusiveSum((uint32_t)__popc(m), off, total);
    off += blk_off[blockIdx.x];
    for (; m; m &= m - 1, off++)
        if (off < cap) nl[off] = (uint32_t)at + (uint32_t)__ffs(m) - 1;
    if (blockIdx.x == nblk - 1 && threadIdx.x == 0) cnt->n_lines = blk_off[blockIdx.x] + total;
}

// one record = 4 lines (FASTQ) or 2 (FASTA reads). A text on which line parsing and the reference's token parsing
// (`>>` skips white space, reads.cpp:52-74) would disagree is flagged irregular and left to the host parser.
__global__ __launch_bounds__(256) void build_raw(const uint8_t *__restrict__ text, unsigned long long nbytes, const uint32_t *__restrict__ nl, uint32_t nl_cap,
                                                 int fasta, uint32_t readset, uint32_t first_index, uint32_t read_end, uint32_t max_readlen, uint32_t max_reads,
                                                 uint32_t pair_split, uint32_t pair_n,
                                                 const CarryState *__restrict__ carry, basal_rawread *__restrict__ raw, BatchCounters *__restrict__ cnt) {
    if (first_index == 0xFFFFFFFFu) first_index = carry->next_index;
    const uint32_t lpr = fasta ? 2u : 4u, n_lines = cnt->n_lines;
    uint32_t nrec = n_lines / lpr;
    bool bad = n_lines % lpr != 0 || n_lines > nl_cap || nrec > max_reads || (nbytes && text[nbytes - 1] != '\n');
    // paired-end text: mate 1's pair_n records (the first pair_split bytes), then mate 2's pair_n records; record k of either half is mate
    // k's, and they come out interleaved (a0, b0, a1, b1, ...) with the pair's number as both mates' read number
    const bool pairs = pair_n != 0;
    if (pairs && !bad) bad = nrec != 2 * pair_n || nl[lpr * pair_n - 1] + 1 != pair_split;
    if (first_index >= read_end) nrec = 0;
    else if (nrec > read_end - first_index) nrec = read_end - first_index;  // -E: reads beyond read_end are not loaded (reads.cpp:45)
    if (bad) nrec = 0;
    const uint32_t t0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (t0 == 0) { cnt->n_reads = nrec; if (bad) cnt->irregular = 1; }
    for (uint32_t k = t0; k < nrec; k += gridDim.x * blockDim.x) {
        uint32_t lb[4], le[4];
        bool irr = false;
        for (uint32_t j = 0; j < lpr; j++) {
            const uint32_t li = k * lpr + j;
            lb[j] = li ? nl[li - 1] + 1 : 0;
            le[j] = nl[li];
            if (le[j] <= lb[j] || is_ws(text[lb[j]])) irr = true;  // empty line, or one that starts with white space
        }
        basal_rawread rr;
        memset(&rr, 0, sizeof rr);
        if (!irr) {
            if (text[lb[0]] != (fasta ? '>' : '@')) irr = true;
            Bytes16 tb(text);
            // name: the first token behind the marker (reads.cpp:53-55 `>>ch; >>name; getline`)
            uint32_t p = lb[0] + 1;
            while (p < le[0] && !is_ws(tb.get(p))) p++;
            const uint32_t nlen = p - (lb[0] + 1);
            if (nlen == 0 || nlen > 0xffffu) irr = true;
            rr.name_off = lb[0] + 1;
            rr.name_len = (uint16_t)nlen;
            // bases: one token, then only white space up to the end of the line
            p = lb[1];
            while (p < le[1] && !is_ws(tb.get(p))) p++;
            uint32_t sl = p - lb[1];
            for (; p < le[1]; p++) if (!is_ws(tb.get(p))) irr = true;
            uint32_t ql = 0;
            if (!fasta) {
                p = lb[3];
                while (p < le[3] && !is_ws(tb.get(p))) p++;
                ql = p - lb[3];
                for (; p < le[3]; p++) if (!is_ws(tb.get(p))) irr = true;
                rr.qual_off = lb[3];
            }
            if (sl > 0xffffu || ql > 0xffffu) irr = true;
            if (sl > max_readlen) {  // reads.cpp:63-65: both strings cut at max_readlen
                sl = max_readlen;
                if (ql > max_readlen) ql = max_readlen;
            }
            rr.seq_off = lb[1];
            rr.seq_len = (uint16_t)sl;
            rr.qual_len = (uint16_t)ql;
            rr.readset = (uint8_t)(pairs ? (k >= pair_n ? 2u : 1u) : readset);
            rr.index = first_index + (pairs && k >= pair_n ? k - pair_n : k);
        }
        if (irr) cnt->irregular = 1;
        raw[pairs ? (k >= pair_n ? 2 * (k - pair_n) + 1 : 2 * k) : k] = rr;
    }
}

// ------------------------------------------------------------------------------------------------ FilterReads
// the quality character the reference holds after TrimLowQual's shift (align.cpp:58-61)
__device__ __forceinline__ uint32_t qual_char(const PrepConst &k, Bytes16 &q, bool fill, uint32_t i) {
    const uint32_t shift = k.zero_qual - '!';
    return ((fill ? k.zero_qual + k.default_qual : q.get(i)) - shift) & 0xffu;
}

__global__ __launch_bounds__(256) void filter_reads(PrepConst k, const uint8_t *__restrict__ text, const uint8_t *__restrict__ reg_alphabet,
                                                    const basal_rawread *__restrict__ raw, const uint32_t *__restrict__ n_ptr, uint32_t n_host,
                                                    basal_read *__restrict__ desc, ReadAux *__restrict__ aux, uint16_t *__restrict__ npos, int32_t *__restrict__ defidx,
                                                    uint32_t *__restrict__ order, uint32_t max_reads, BatchCounters *__restrict__ cnt) {
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    const int lane = threadIdx.x & 63;
    for (uint32_t r0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63u; r0 < n; r0 += gridDim.x * blockDim.x) {
        const uint32_t r = r0 + (uint32_t)lane;
        uint32_t cls = 3;
        if (r < n) {
            const basal_rawread rr = raw[r];
            Bytes16 seq(text + rr.seq_off), qual(text + rr.qual_off);
            uint32_t L = rr.seq_len;
            uint32_t x = k.max_snp_num < 100 ? k.max_snp_num : (uint32_t)((k.max_snp_num - 100) / 100.0 * L + 0.5);  // align.cpp:550-551
            if (k.gap > 0) x += 1 + k.gap;
            if (x > BASAL_MAXSNPS) x = BASAL_MAXSNPS;
            const uint32_t raw_len = L;
            bool fill = rr.qual_len == 0;  // FASTA reads: the loader makes default qualities (reads.cpp:77)
            uint32_t qlen = fill ? L : rr.qual_len;
            // TrimAdapter (align.cpp:418-435): <= 4 mismatches in the first 15 adapter bases, 1 per 5 compared
            bool cut = false;
            for (uint32_t a = 0; a < k.n_adapter && !cut && L >= 4; a++) {
                const uint32_t al = k.adapter_len[a];
                for (uint32_t pos = k.K + k.I - 1; pos + 4 < L; pos++) {
                    uint32_t mis = 0, j = 0;
                    for (; j < al && pos + j < L; j++)
                        if ((mis += (k.adapter[a][j] != seq.get(pos + j))) > 4) break;
                    if (j >= mis * 5 && j > 3) {
                        L = pos;
                        if (qlen > pos) qlen = pos;
                        cut = true;
                        break;
                    }
                }
            }
            // TrimLowQual (align.cpp:51-76)
            if (L != qlen) { fill = true; qlen = L; }
            bool failed = false;
            uint32_t thres = (k.zero_qual + k.trim_qual) & 0xffu;
            if (k.zero_qual != '!') thres = (thres - (k.zero_qual - '!')) & 0xffu;
            if (k.trim_qual != 0) {
                uint32_t i = L;
                while (i > 0 && !(qual_char(k, qual, fill, i - 1) > thres)) i--;
                if (i < k.K + k.I - 1) failed = true;
                else { qlen = i; L = i; }
            }
            if (!failed && L < k.min_read_size) failed = true;
            if (!failed) {  // CountNs (align.cpp:40-47)
                uint32_t ns = 0;
                for (uint32_t i = 0; i < L; i++) ns += !reg_alphabet[seq.get(i)];
                if (ns > k.max_ns) failed = true;
            }
            basal_read d;
            d.seq_off = rr.seq_off;
            d.index = rr.index;
            d.len = failed ? 0 : (uint16_t)L;
            d.readset = rr.readset;
            d.max_snp = failed ? 0 : (uint8_t)((x + 1) * (L - 1) / raw_len);  // align.cpp:561
            d.stale_idx = BASAL_STALE_NONE;
            desc[r] = d;
            if (!failed) cls = L <= 128 ? 0 : L <= 256 ? 1 : 2;
            ReadAux a;
            a.seq_len = (uint16_t)L; a.qual_len = (uint16_t)qlen; a.qc_failed = failed; a.qual_fill = fill; a.cls = (uint8_t)cls; a.pad = 0;
            aux[r] = a;
            const uint32_t slot = rr.readset == 2 ? 1u : 0u;
            if (!failed) {
                npos[(size_t)slot * max_reads + r] = L >= k.K ? (uint16_t)(L - k.K + 1) : 0;
                if ((L - k.I + 1) % k.K != 0) defidx[(size_t)slot * max_reads + r] = (int32_t)r;
            }
        }
        const unsigned long long mf = __ballot(cls == 3 && r < n);
        if (mf && lane == 0) atomicAdd(&cnt->n_filtered, (unsigned long long)__popcll(mf));
    }
}

// The read-length classes' work lists (one align launch per class, each with the kernel instantiation for its lengths). A read that
// inherits its start offset from another read is aligned by the instantiation that can also pack THAT read (the kernel recomputes
// the inherited offset from the source read's bases), so it may go to a longer class than its own length asks for.
__global__ __launch_bounds__(256) void class_lists(const uint32_t *__restrict__ n_ptr, uint32_t n_host, const basal_read *__restrict__ desc,
                                                   const basal_stale *__restrict__ stales, uint32_t max_reads, uint32_t *__restrict__ order,
                                                   BatchCounters *__restrict__ cnt) {
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    const int lane = threadIdx.x & 63;
    for (uint32_t r0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63u; r0 < n; r0 += gridDim.x * blockDim.x) {
        const uint32_t r = r0 + (uint32_t)lane;
        uint32_t cls = 3;
        if (r < n) {
            const basal_read d = desc[r];
            uint32_t L = d.len;
            if (L && d.stale_idx != BASAL_STALE_NONE) {
                const uint32_t src = stales[d.stale_idx].src;
                if (src < max_reads + 2) { const uint32_t sl = desc[src].len; L = sl > L ? sl : L; }
            }
            if (L) cls = L <= 128 ? 0 : L <= 256 ? 1 : 2;
        }
        for (uint32_t c = 0; c < 3; c++) {  // wave-aggregated
            const unsigned long long m = __ballot(cls == c);
            if (!m) continue;
            uint32_t base = 0;
            if (lane == __ffsll((unsigned long long)m) - 1) base = atomicAdd(&cnt->cls_n[c], (uint32_t)__popcll(m));
            base = __shfl(base, __ffsll((unsigned long long)m) - 1);
            if (cls == c) order[(size_t)c * max_reads + base + (uint32_t)__popcll(m & ((1ULL << lane) - 1))] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------ inherited state
// maxima of npos over groups of 64 and 4096 reads (backward searches skip whole groups)
__global__ __launch_bounds__(256) void block_maxima(const uint16_t *__restrict__ npos, uint32_t max_reads, uint16_t *__restrict__ bmax1, uint16_t *__restrict__ bmax2) {
    // grid.y = slot; one block per 4096 reads, one thread per 16
    const uint32_t slot = blockIdx.y, g1 = max_reads / 64, g2 = max_reads / 4096;
    const uint16_t *p = npos + (size_t)slot * max_reads + (size_t)blockIdx.x * 4096 + threadIdx.x * 16;
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) m = max(m, (uint32_t)p[i]);
    m = max(m, (uint32_t)__shfl_xor((int)m, 1));
    m = max(m, (uint32_t)__shfl_xor((int)m, 2));
    if ((threadIdx.x & 3) == 0) bmax1[(size_t)slot * g1 + blockIdx.x * 64 + (threadIdx.x >> 2)] = (uint16_t)m;
    __shared__ uint32_t sm;
    if (threadIdx.x == 0) sm = 0;
    __syncthreads();
    atomicMax(&sm, m);
    __syncthreads();
    if (threadIdx.x == 0) bmax2[(size_t)slot * g2 + blockIdx.x] = (uint16_t)sm;
}

// nearest read number < from whose npos (of this slot) is > T, or -1
__device__ int32_t search_prev(const uint16_t *__restrict__ np, const uint16_t *__restrict__ b1, const uint16_t *__restrict__ b2, int32_t from, uint32_t T) {
    int32_t r = from - 1;
    while (r >= 0) {
        if ((r & 63) == 63) {  // at the end of a group: skip groups that hold nothing larger
            if ((r & 4095) == 4095 && b2[r >> 12] <= T) { r -= 4096; continue; }
            if (b1[r >> 6] <= T) { r -= 64; continue; }
        }
        if (np[r] > T) return r;
        r--;
    }
    return -1;
}

// xseed_array / xseedreg_array entry of `seq` at read offset pos on chain c (align.cpp:92-100)
__device__ uint32_t seed_at(const uint8_t *__restrict__ tables, const uint8_t *seq, uint32_t L, uint32_t pos, int c, uint32_t K) {
    uint32_t s = 0;
    bool nn = false;
    for (uint32_t t = 0; t < K; t++) {
        const uint32_t ch = c ? seq[L - 1 - (pos + t)] : seq[pos + t];
        s = (s << 2) | tables[(c ? 256 : 0) + ch];
        nn |= !tables[512 + ch];
    }
    return XT(s) | (nn ? 0x80000000u : 0u);
}

__global__ __launch_bounds__(256) void stale_table(PrepConst k, const uint8_t *__restrict__ text, const uint8_t *__restrict__ tables, const uint32_t *__restrict__ n_ptr,
                                                   uint32_t n_host, basal_read *__restrict__ desc, const uint16_t *__restrict__ npos, const uint16_t *__restrict__ bmax1,
                                                   const uint16_t *__restrict__ bmax2, const int32_t *__restrict__ defscan, uint32_t max_reads,
                                                   const CarryState *__restrict__ carry, basal_stale *__restrict__ stales, BatchCounters *__restrict__ cnt) {
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    const uint32_t g1 = max_reads / 64, g2 = max_reads / 4096;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const basal_read d = desc[r];
        if (d.len == 0 || (d.len - k.I + 1) % k.K != 0) continue;
        const uint32_t rs = d.readset & 0x7fu, slot = rs == 2 ? 1u : 0u;
        const uint16_t *np = npos + (size_t)slot * max_reads, *b1 = bmax1 + (size_t)slot * g1, *b2 = bmax2 + (size_t)slot * g2;
        basal_stale st;
        const int32_t src = defscan[(size_t)slot * max_reads + r];
        st.src = src >= 0 ? (uint32_t)src : carry->ghost[slot].valid ? max_reads + slot : BASAL_STALE_CARRY;
        const bool flag[2] = {(k.chains == 1) || ((k.chains <= 1) == (rs < 2)), (k.chains == 1) || ((k.chains <= 1) == (rs == 2))};
        const uint32_t my_npos = d.len >= k.K ? d.len - k.K + 1 : 0;
        int32_t from = (int32_t)r;  // the answer for a larger threshold lies at or before the answer for a smaller one
        uint32_t ce = 0;            // same for the carried stack (most recent first, npos increasing)
        for (uint32_t j = 0; j < 15; j++) {
            const uint32_t pos = my_npos + j;
            uint32_t v0 = 0, v1 = 0;  // never-written slots hold 0 (a fresh SingleAlign object)
            int32_t q = from >= 0 ? search_prev(np, b1, b2, from, pos) : -1;
            if (q >= 0) {
                from = q + 1;  // q itself may serve the next threshold too
                const basal_read s = desc[q];
                if (flag[0]) v0 = seed_at(tables, text + s.seq_off, s.len, pos, 0, k.K);
                if (flag[1]) v1 = seed_at(tables, text + s.seq_off, s.len, pos, 1, k.K);
            } else {
                from = -1;
                while (ce < carry->depth[slot] && carry->stack[slot][ce].npos <= pos) ce++;
                if (ce < carry->depth[slot]) {
                    const CarryRead &s = carry->stack[slot][ce];
                    if (flag[0]) v0 = seed_at(tables, s.seq, s.len, pos, 0, k.K);
                    if (flag[1]) v1 = seed_at(tables, s.seq, s.len, pos, 1, k.K);
                }
            }
            st.overlay[0][j] = v0;
            st.overlay[1][j] = v1;
        }
        stales[r] = st;
        desc[r].stale_idx = r;  // the table is indexed by read number
        atomicAdd(&cnt->n_stale, 1u);
    }
}

// reads that stay visible to later reads = right-to-left strict maxima of npos (the tracker's stack, basal_host_stale_visit)
__global__ __launch_bounds__(256) void stack_collect(const uint16_t *__restrict__ npos, const uint16_t *__restrict__ sufmax_rev, const uint32_t *__restrict__ n_ptr,
                                                     uint32_t n_host, uint32_t max_reads, uint32_t *__restrict__ list, uint32_t *__restrict__ list_n) {
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x)
        for (uint32_t slot = 0; slot < 2; slot++) {
            const uint32_t v = npos[(size_t)slot * max_reads + r];
            // sufmax_rev[max_reads - 1 - r] = max of npos over reads r+1.. (exclusive scan of the reversed array)
            if (v > 0 && v > sufmax_rev[(size_t)slot * max_reads + (max_reads - 1 - r)]) {
                const uint32_t i = atomicAdd(&list_n[slot], 1u);
                if (i < (uint32_t)kStackMax) list[slot * kStackMax + i] = r;
            }
        }
}

__global__ __launch_bounds__(256) void reverse_npos(const uint16_t *__restrict__ npos, uint32_t max_reads, uint16_t *__restrict__ rev) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < max_reads; i += gridDim.x * blockDim.x) {
        rev[i] = npos[max_reads - 1 - i];
        rev[(size_t)max_reads + i] = npos[(size_t)max_reads + max_reads - 1 - i];
    }
}

// the carry state the NEXT batch sees: new stack entries (this batch, most recent first) in front of the old entries that are
// still larger; the last read that defined each slot's start offset; one block
__global__ __launch_bounds__(512) void carry_update(const uint8_t *__restrict__ text, const basal_read *__restrict__ desc, const uint16_t *__restrict__ npos,
                                                    const int32_t *__restrict__ defscan, const int32_t *__restrict__ defidx_raw, const uint32_t *__restrict__ n_ptr,
                                                    uint32_t n_host, uint32_t max_reads, const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_n,
                                                    const CarryState *__restrict__ in, CarryState *__restrict__ out, uint32_t batch_no, BatchCounters *__restrict__ cnt) {
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    __shared__ uint32_t s_r[kStackMax], s_sorted[kStackMax];
    __shared__ uint32_t s_keep_from, s_new;
    if (threadIdx.x == 0) {
        out->next_index = n ? desc[n - 1].index + 1 : in->next_index;
        out->seq = batch_no + 1;
        if (in->seq != batch_no) atomicOr(&cnt->irregular, 4u);  // not the state batch_no - 1 left: the pipe reports it (basal_pipe_collect)
    }
    for (uint32_t slot = 0; slot < 2; slot++) {
        uint32_t m = list_n[slot];
        if (m > (uint32_t)kStackMax) m = kStackMax;
        for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) s_r[i] = list[slot * kStackMax + i];
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {  // rank sort, descending read number = most recent first
            uint32_t rank = 0;
            for (uint32_t j = 0; j < m; j++) rank += s_r[j] > s_r[i];
            s_sorted[rank] = s_r[i];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t top = m ? npos[(size_t)slot * max_reads + s_sorted[m - 1]] : 0;  // the largest npos of the new entries
            uint32_t e = 0;
            while (e < in->depth[slot] && in->stack[slot][e].npos <= top) e++;
            s_keep_from = e;
            uint32_t total = m + (in->depth[slot] - e);
            if (total > (uint32_t)kStackMax) total = kStackMax;  // cannot happen: npos values are distinct and <= 465
            s_new = total;
            out->depth[slot] = total;
        }
        __syncthreads();
        const uint32_t total = s_new, keep = s_keep_from;
        for (uint32_t ent = 0; ent < total; ent++) {
            CarryRead *o = &out->stack[slot][ent];
            if (ent < m) {
                const basal_read d = desc[s_sorted[ent]];
                if (threadIdx.x == 0) { o->npos = npos[(size_t)slot * max_reads + s_sorted[ent]]; o->len = d.len; o->readset = d.readset; o->valid = 1; }
                for (uint32_t b = threadIdx.x; b < d.len; b += blockDim.x) o->seq[b] = text[d.seq_off + b];
            } else {
                const CarryRead *s = &in->stack[slot][keep + (ent - m)];
                if (threadIdx.x == 0) { o->npos = s->npos; o->len = s->len; o->readset = s->readset; o->valid = 1; }
                for (uint32_t b = threadIdx.x; b < s->len; b += blockDim.x) o->seq[b] = s->seq[b];
            }
        }
        // ghost: the last defining read of this batch (the last element of the inclusive scan), else the one carried so far
        int32_t last = -1;
        if (n) {
            last = defscan[(size_t)slot * max_reads + n - 1];  // exclusive scan: covers reads 0..n-2
            const int32_t own = defidx_raw[(size_t)slot * max_reads + n - 1];
            if (own > last) last = own;
        }
        CarryRead *g = &out->ghost[slot];
        if (last >= 0) {
            const basal_read d = desc[last];
            if (threadIdx.x == 0) { g->npos = 0; g->len = d.len; g->readset = d.readset; g->valid = 1; }
            for (uint32_t b = threadIdx.x; b < d.len; b += blockDim.x) g->seq[b] = text[d.seq_off + b];
        } else {
            const CarryRead *s = &in->ghost[slot];
            if (threadIdx.x == 0) { g->npos = 0; g->len = s->len; g->readset = s->readset; g->valid = s->valid; }
            for (uint32_t b = threadIdx.x; b < s->len; b += blockDim.x) g->seq[b] = s->seq[b];
        }
        __syncthreads();
    }
}

// the ghost reads of THIS batch: bytes behind the batch's text, descriptors behind the batch's descriptors
__global__ __launch_bounds__(512) void ghost_install(const CarryState *__restrict__ carry, uint8_t *__restrict__ text, unsigned long long text_cap,
                                                     basal_read *__restrict__ desc, uint32_t max_reads) {
    for (uint32_t slot = 0; slot < 2; slot++) {
        const CarryRead *g = &carry->ghost[slot];
        const uint32_t len = g->valid ? g->len : 0;
        for (uint32_t b = threadIdx.x; b < len; b += blockDim.x) text[text_cap + slot * 512 + b] = g->seq[b];
        if (threadIdx.x == 0) {
            basal_read d;
            d.seq_off = (uint32_t)(text_cap + slot * 512);
            d.index = 0; d.len = (uint16_t)len; d.readset = g->readset; d.max_snp = 0; d.stale_idx = BASAL_STALE_NONE;
            desc[max_reads + slot] = d;
        }
    }
}

// ------------------------------------------------------------------------------------------------ SAM text (s_OutHit, align.cpp:616-669)
struct CountSink {
    unsigned long long n = 0;
    __device__ __forceinline__ void ch(uint32_t) { n++; }
    __device__ __forceinline__ void skip(uint32_t l) { n += l; }
    static constexpr bool kWrites = false;
};
// bytes are gathered into 16-byte chunks aligned at the destination and leave as one store each; the partial chunks at the two ends of
// a read's text (their other bytes belong to the neighbouring reads) leave byte by byte
struct WriteSink {
    uint8_t *p;                 // next byte to be written
    unsigned long long lo, hi;  // the chunk being gathered (bytes [p & ~15, p)); selects and shifts only -- the lanes of a wave are at
                                // different byte positions, and a branch per position would serialise them
    uint32_t have;              // 1: every byte of the current chunk from its start on was produced here (it may leave as one store)
    __device__ __forceinline__ explicit WriteSink(uint8_t *dst) : p(dst), lo(0), hi(0), have(((unsigned long long)dst & 15u) == 0) {}
    __device__ __forceinline__ void ch(uint32_t c) {
        const uint32_t k = (uint32_t)((unsigned long long)p & 15u);
        const unsigned long long v = (unsigned long long)(c & 0xffu) << ((k & 7u) * 8);
        lo |= k < 8 ? v : 0ull;
        hi |= k < 8 ? 0ull : v;
        if (!have) *p = (uint8_t)c;  // the head of the text, up to the first chunk boundary
        p++;
        if (k == 15) {
            if (have) *(uint4 *)(p - 16) = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
            have = 1;
            lo = hi = 0;
        }
    }
    __device__ __forceinline__ void flush() {  // the tail: the bytes gathered in an unfinished chunk
        if (!have) return;
        const uint32_t k = (uint32_t)((unsigned long long)p & 15u);
        uint8_t *q = p - k;
        for (uint32_t i = 0; i < k; i++) q[i] = (uint8_t)((i < 8 ? lo : hi) >> ((i & 7u) * 8));
    }
    static constexpr bool kWrites = true;
};

template <class S>
__device__ __forceinline__ void put_num(S &o, long long v) {  // |v| < 2^32; no digit buffer (a private array would live in scratch memory)
    if (v < 0) { o.ch('-'); v = -v; }
    uint32_t u = (uint32_t)v;
    bool started = false;
#pragma unroll
    for (uint32_t d = 1000000000u; d >= 1; d /= 10) {
        const uint32_t q = u / d;
        u -= q * d;
        if (q || started || d == 1) { o.ch('0' + q); started = true; }
    }
}
template <class S>
__device__ __forceinline__ void put_lit(S &o, const char *s) { for (; *s; s++) o.ch(*s); }
template <class S>
__device__ __forceinline__ void put_bytes(S &o, const uint8_t *s, uint32_t l) {
    if constexpr (!S::kWrites) o.skip(l);
    else {
        Bytes16 r(s);
        for (uint32_t i = 0; i < l; i++) o.ch(r.get(i));
    }
}

__device__ __forceinline__ uint32_t comp_char(uint32_t c) {  // rev_char, param.cpp:146-156
    switch (c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
        default: return 'N';
    }
}

struct FmtCtx {
    PrepConst k;
    const uint8_t *text;
    const basal_rawread *raw;
    const ReadAux *aux;
    const basal_result *res;
    const basal_hit *stream;
    const char *names;
    const uint32_t *name_off;
    const uint64_t *xref_fwd;
    const uint32_t *anchor;
    uint32_t ncontig;
};

template <class S>
__device__ __forceinline__ void put_seq_qual(S &o, const FmtCtx &f, const basal_rawread &rr, const ReadAux &a, bool rev) {
    const uint32_t len = a.seq_len, qlen = a.qual_len;
    if constexpr (!S::kWrites) { o.skip(len + 1 + qlen); return; }
    else {
        Bytes16 seq(f.text + rr.seq_off), qual(f.text + rr.qual_off);
        if (!rev) for (uint32_t i = 0; i < len; i++) o.ch(seq.get(i));
        else for (uint32_t i = 0; i < len; i++) o.ch(comp_char(seq.get(len - 1 - i)));
        o.ch('\t');
        if (!rev) for (uint32_t i = 0; i < qlen; i++) o.ch(qual_char(f.k, qual, a.qual_fill, i));
        else for (uint32_t i = 0; i < qlen; i++) o.ch(qual_char(f.k, qual, a.qual_fill, qlen - 1 - i));
    }
}

// XR:Z (align.cpp:646-658, pairs.cpp:338-350): 2 lower-case flank bases, len bases, 2 lower-case flank bases of the forward strand
template <class S>
__device__ __forceinline__ void put_xr(S &o, const FmtCtx &f, uint32_t contig, uint32_t loc, uint32_t len) {
    if (contig >= f.ncontig) contig = 0;
    const uint64_t *s = f.xref_fwd + f.anchor[contig] / 32;
    put_lit(o, "\tXR:Z:");
    if constexpr (!S::kWrites) o.skip((loc >= 2) + (loc >= 1) + len + 2);
    else {
        // the four letters as one word (a per-lane index into the by-value argument struct would put the struct into scratch memory)
        const uint32_t nt4 = (uint32_t)(uint8_t)f.k.useful_nt[0] | ((uint32_t)(uint8_t)f.k.useful_nt[1] << 8) | ((uint32_t)(uint8_t)f.k.useful_nt[2] << 16) |
                             ((uint32_t)(uint8_t)f.k.useful_nt[3] << 24);
        unsigned long long wcur = ~0ull, word = 0;  // one reference word serves 32 bases
        auto base_at = [&](uint32_t xx) {
            if ((xx >> 5) != wcur) { wcur = xx >> 5; word = s[wcur]; }
            return (nt4 >> (8 * (uint32_t)((word >> (62 - 2 * (xx & 31))) & 3))) & 0xffu;
        };
        for (uint32_t q = 2; q > 0; q--)
            if (loc >= q) o.ch(base_at(loc - q) + 32);
        for (uint32_t q = 0; q < len + 2; q++) o.ch(base_at(loc + q) + (q >= len ? 32 : 0));
    }
}

// n <= 0: unmapped (n < 0: failed QC); else one alignment record
template <class S>
__device__ __forceinline__ void put_record(S &o, const FmtCtx &f, const basal_rawread &rr, const ReadAux &a, uint32_t chain, int n, uint32_t level, const basal_hit &h) {
    int flag = (int)(0x40 * rr.readset);
    if (n <= 0) {
        if (!f.k.out_unmap) return;
        flag |= n < 0 ? 0x204 : 0x4;
        put_bytes(o, f.text + rr.name_off, rr.name_len); o.ch('\t'); put_num(o, flag); put_lit(o, "\t*\t0\t0\t*\t*\t0\t0\t");
        put_seq_qual(o, f, rr, a, false);
        o.ch('\n');
        return;
    }
    const uint32_t len = a.seq_len;
    const bool rev = (chain ^ (h.chr & 1u)) != 0;
    if (n != 1) flag |= 0x100;
    if (rev) flag |= 0x10;
    put_bytes(o, f.text + rr.name_off, rr.name_len); o.ch('\t'); put_num(o, flag); o.ch('\t');
    {
        const uint32_t ctg = h.chr >> 1;
        if (ctg < f.ncontig) put_bytes(o, (const uint8_t *)f.names + f.name_off[ctg], f.name_off[ctg + 1] - f.name_off[ctg]);
        else o.ch('*');
    }
    o.ch('\t'); put_num(o, (long long)h.loc + 1);
    put_lit(o, "\t255\t");
    if (h.gap_size == 0) { put_num(o, len); o.ch('M'); }  // align.cpp:641-643
    else if (h.gap_size > 0) { put_num(o, h.gap_pos); o.ch('M'); put_num(o, h.gap_size); o.ch('D'); put_num(o, (int)len - (int)h.gap_pos); o.ch('M'); }
    else { put_num(o, h.gap_pos); o.ch('M'); put_num(o, -(int)h.gap_size); o.ch('I'); put_num(o, (int)len - (int)h.gap_pos + (int)h.gap_size); o.ch('M'); }
    put_lit(o, "\t*\t0\t0\t");
    put_seq_qual(o, f, rr, a, rev);
    put_lit(o, "\tNM:i:"); put_num(o, (long long)(level & 0xffu));
    if (f.k.out_ref) put_xr(o, f, (h.chr & 0xfffeu) >> 1, h.loc, len);  // (the reference masks 16 bits here, align.cpp:646-658)
    put_lit(o, "\tZS:Z:"); o.ch((h.chr & 1) ? '-' : '+'); o.ch(chain ? '-' : '+'); o.ch('\n');
}

// StringAlign (align.cpp:583-612) with the GPU's choice of hit. One call site of put_record, everything inlined into the two
// kernels: a call that passes the by-value argument struct by reference would make every lane copy the struct to scratch memory.
template <class S>
__device__ __forceinline__ void put_read(S &o, const FmtCtx &f, uint32_t r, uint32_t *kind) {
    const basal_rawread rr = f.raw[r];
    const ReadAux a = f.aux[r];
    basal_hit h;
    memset(&h, 0, sizeof h);
    *kind = 0;
    int n = 0;
    uint32_t nrec = 1, level = 0, chain = 0, first = 0;
    bool from_stream = false;
    if (a.qc_failed) n = -1;
    else {
        const basal_result rs = f.res[r];
        if (rs.best_level != 0xFF) {
            const uint32_t sum = (uint32_t)rs.n_hit + rs.n_chit;
            level = rs.best_level;
            *kind = sum == 1 ? 1 : 2;
            if (sum == 1 || f.k.report_repeat_hits == 1) { n = (int)sum; h = rs.best; chain = rs.best.chain; }
            else if (f.k.report_repeat_hits == 2) {
                if (rs.status == BASAL_READ_OVERFLOW || rs.stream_n != sum) { *kind |= 4; return; }  // stream too small: the batch is redone
                n = (int)sum; nrec = sum; first = rs.stream_first; from_stream = true;
            }  // -r 0: one unmapped record (n = 0)
        }
    }
    for (uint32_t j = 0; j < nrec; j++) {
        if (from_stream) { h = f.stream[first + j]; chain = h.chain; }
        put_record(o, f, rr, a, chain, n, level, h);
    }
}

__global__ __launch_bounds__(256) void sam_lengths(FmtCtx f, const uint32_t *__restrict__ n_ptr, uint32_t n_host, uint32_t max_reads,
                                                   unsigned long long *__restrict__ out_off, BatchCounters *__restrict__ cnt) {
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r <= max_reads; r += gridDim.x * blockDim.x) {
        unsigned long long len = 0;
        uint32_t kind = 0;
        if (r < n) {
            CountSink o;
            put_read(o, f, r, &kind);
            len = o.n;
        }
        out_off[r] = len;
        // main.cpp:606-612
        const unsigned long long mu = __ballot((kind & 3) == 1), mm = __ballot((kind & 3) == 2), mo = __ballot(kind & 4);
        if ((threadIdx.x & 63) == 0) {
            if (mu) atomicAdd(&cnt->n_unique, (unsigned long long)__popcll(mu));
            if (mm) atomicAdd(&cnt->n_multiple, (unsigned long long)__popcll(mm));
            const unsigned long long al = __popcll(mu) + (f.k.report_repeat_hits ? __popcll(mm) : 0);
            if (al) atomicAdd(&cnt->n_aligned, al);
            if (mo) atomicOr(&cnt->irregular, 2u);
        }
    }
}

__global__ __launch_bounds__(256) void sam_write(FmtCtx f, const uint32_t *__restrict__ n_ptr, uint32_t n_host, const unsigned long long *__restrict__ out_off,
                                                 uint8_t *__restrict__ out, unsigned long long out_cap, BatchCounters *__restrict__ cnt) {
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt->out_bytes = out_off[n];
    if (out_off[n] > out_cap) return;  // the host grows the buffer and queues this kernel again
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        if (out_off[r + 1] == out_off[r]) continue;
        WriteSink o(out + out_off[r]);
        uint32_t kind;
        put_read(o, f, r, &kind);
        o.flush();
    }
}


// ---- paired-end: FixPairReadName, and the text of s_OutHitPair / s_OutHitUnpair (pairs.cpp:307-507) from the pairing kernel's records ----
// One thread per read pair (records 2 i and 2 i + 1 of the batch). FixPairReadName (pairs.cpp:487-507): names that differ are cut behind the
// last digit of their common prefix (behind the prefix if it holds none); names that differ from the first character on end the run in the
// reference (cnt->pair_err). Mates that both passed FilterReads are aligned with every mode (PairAlign::RunAlign drives them, pairs.cpp:132-177).
__global__ __launch_bounds__(256) void pair_fix(const uint8_t *__restrict__ text, basal_rawread *__restrict__ raw, basal_read *__restrict__ desc,
                                                const ReadAux *__restrict__ aux, uint32_t npairs, BatchCounters *__restrict__ cnt) {
    for (uint32_t pi = blockIdx.x * blockDim.x + threadIdx.x; pi < npairs; pi += gridDim.x * blockDim.x) {
        const basal_rawread a = raw[2 * pi], b = raw[2 * pi + 1];
        const uint32_t la = a.name_len, lb = b.name_len, i0 = la < lb ? la : lb;
        uint32_t i = 0;
        int d = -1;
        for (; i < i0; i++) {
            const uint32_t ca = text[a.name_off + i], cb = text[b.name_off + i];
            if (ca != cb) break;
            if (ca >= '0' && ca <= '9') d = (int)i;
        }
        if (!(la == lb && i == i0)) {
            if (i > 0) {
                if (d < 0) d = (int)i - 1;
                if ((uint32_t)d + 1 < la) raw[2 * pi].name_len = (uint16_t)(d + 1);
                if ((uint32_t)d + 1 < lb) raw[2 * pi + 1].name_len = (uint16_t)(d + 1);
            } else if (atomicOr(&cnt->pair_err, 1u) == 0) cnt->pair_err_at = pi;
        }
        if (!aux[2 * pi].qc_failed && !aux[2 * pi + 1].qc_failed) {
            desc[2 * pi].readset |= BASAL_READ_ALLMODES;
            desc[2 * pi + 1].readset |= BASAL_READ_ALLMODES;
        }
    }
}

struct PeFmt {
    const basal_pe_pair *pairs;
    const basal_pe_rec *recs;
};

template <class S>
__device__ __forceinline__ void put_contig(S &o, const FmtCtx &f, uint32_t ctg) {
    if (ctg < f.ncontig) put_bytes(o, (const uint8_t *)f.names + f.name_off[ctg], f.name_off[ctg + 1] - f.name_off[ctg]);
    else o.ch('*');
}
template <class S>
__device__ __forceinline__ void put_cigar(S &o, const basal_hit &h, uint32_t len) {  // pairs.cpp:329-331
    if (h.gap_size == 0) { put_num(o, len); o.ch('M'); }
    else if (h.gap_size > 0) { put_num(o, h.gap_pos); o.ch('M'); put_num(o, h.gap_size); o.ch('D'); put_num(o, (int)len - (int)h.gap_pos); o.ch('M'); }
    else { put_num(o, h.gap_pos); o.ch('M'); put_num(o, -(int)h.gap_size); o.ch('I'); put_num(o, (int)len - (int)h.gap_pos + (int)h.gap_size); o.ch('M'); }
}

// every line of one read pair, in the order the records list them
template <class S>
__device__ __forceinline__ void put_pair(S &o, const FmtCtx &f, const PeFmt &pf, uint32_t pi) {
    const basal_pe_pair pr = pf.pairs[pi];
    for (uint32_t k = 0; k < pr.n; k++) {
        const basal_pe_rec e = pf.recs[pr.first + k];
        if (e.kind == BASAL_PE_PAIR) {  // s_OutHitPair, pairs.cpp:307-411: mate 1's line, then mate 2's
            const uint32_t chain = e.chain_a;
            for (uint32_t side = 0; side < 2; side++) {
                const uint32_t r = 2 * pi + side;
                const basal_rawread rr = f.raw[r];
                const ReadAux ax = f.aux[r];
                const basal_hit h = side ? e.hb : e.ha, mh = side ? e.ha : e.hb;
                const bool rev = side ? (((chain == 0) ? 1u : 0u) ^ (h.chr & 1u)) != 0 : ((chain ^ (h.chr & 1u)) != 0);
                int flag = 0x3 | (e.ma > 1 ? 0x100 : 0) | (rev ? 0x10 : 0x20) | (int)(0x40 * rr.readset);
                put_bytes(o, f.text + rr.name_off, rr.name_len); o.ch('\t'); put_num(o, flag); o.ch('\t'); put_contig(o, f, h.chr >> 1); o.ch('\t');
                put_num(o, (long long)h.loc + 1); put_lit(o, "\t255\t"); put_cigar(o, h, ax.seq_len); put_lit(o, "\t=\t"); put_num(o, (long long)mh.loc + 1); o.ch('\t');
                put_num(o, rev ? -(long long)e.insert : (long long)e.insert); o.ch('\t');
                put_seq_qual(o, f, rr, ax, rev);
                put_lit(o, "\tNM:i:"); put_num(o, (long long)((side ? (uint32_t)e.mb : e.na) & 0xffu));
                if (f.k.out_ref) put_xr(o, f, h.chr >> 1, h.loc, ax.seq_len);
                put_lit(o, "\tZS:Z:"); o.ch((h.chr & 1) ? '-' : '+'); o.ch((side ? chain == 0 : chain != 0) ? '-' : '+'); o.ch('\n');
            }
        } else {  // s_OutHitUnpair, pairs.cpp:418-485: one line of mate e.side
            const uint32_t r = 2 * pi + (e.side ? 1u : 0u);
            const basal_rawread rr = f.raw[r];
            const ReadAux ax = f.aux[r];
            const basal_hit ha = e.ha, hb = e.hb;
            const int ma = e.ma, mb = e.mb;
            int flag = 1 | (int)(0x40 * rr.readset);
            const bool rev = ((uint32_t)e.chain_a ^ (ha.chr & 1u)) != 0;
            if (ma <= 0) {
                flag |= ma < 0 ? 0x204 : 0x004;
                if (mb <= 0) {
                    flag |= 0x008;
                    put_bytes(o, f.text + rr.name_off, rr.name_len); o.ch('\t'); put_num(o, flag); put_lit(o, "\t*\t0\t0\t*\t*\t0\t0\t");
                } else {
                    if ((uint32_t)e.chain_b ^ (hb.chr & 1u)) flag |= 0x020;
                    put_bytes(o, f.text + rr.name_off, rr.name_len); o.ch('\t'); put_num(o, flag); put_lit(o, "\t*\t0\t0\t*\t"); put_contig(o, f, hb.chr >> 1); o.ch('\t');
                    put_num(o, (long long)hb.loc + 1); put_lit(o, "\t0\t");
                }
                put_seq_qual(o, f, rr, ax, false);
                o.ch('\n');
                continue;
            }
            if (ma > 1) flag |= 0x100;
            if (rev) flag |= 0x010;
            if (mb <= 0) flag |= 0x008;
            else if ((uint32_t)e.chain_b ^ (hb.chr & 1u)) flag |= 0x020;
            put_bytes(o, f.text + rr.name_off, rr.name_len); o.ch('\t'); put_num(o, flag); o.ch('\t'); put_contig(o, f, ha.chr >> 1); o.ch('\t');
            put_num(o, (long long)ha.loc + 1); put_lit(o, "\t255\t"); put_cigar(o, ha, ax.seq_len);
            if (mb <= 0) put_lit(o, "\t*\t0\t0\t");
            else { o.ch('\t'); put_contig(o, f, hb.chr >> 1); o.ch('\t'); put_num(o, (long long)hb.loc + 1); put_lit(o, "\t0\t"); }
            put_seq_qual(o, f, rr, ax, rev);
            put_lit(o, "\tNM:i:"); put_num(o, (long long)(int)e.na);
            if (f.k.out_ref) put_xr(o, f, ha.chr >> 1, ha.loc, ax.seq_len);
            put_lit(o, "\tZS:Z:"); o.ch((ha.chr & 1) ? '-' : '+'); o.ch(e.chain_a ? '-' : '+'); o.ch('\n');
        }
    }
}

__global__ __launch_bounds__(256) void pe_lengths(FmtCtx f, PeFmt pf, uint32_t npairs, uint32_t max_reads, unsigned long long *__restrict__ out_off) {
    for (uint32_t pi = blockIdx.x * blockDim.x + threadIdx.x; pi <= max_reads; pi += gridDim.x * blockDim.x) {
        unsigned long long len = 0;
        if (pi < npairs && pf.pairs[pi].status == 0) {
            CountSink o;
            put_pair(o, f, pf, pi);
            len = o.n;
        }
        out_off[pi] = len;
    }
}

__global__ __launch_bounds__(256) void pe_write(FmtCtx f, PeFmt pf, uint32_t npairs, const unsigned long long *__restrict__ out_off, uint8_t *__restrict__ out,
                                                unsigned long long out_cap, BatchCounters *__restrict__ cnt) {
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt->out_bytes = out_off[npairs];
    if (out_off[npairs] > out_cap) return;  // the host grows the buffer and queues this kernel again
    for (uint32_t pi = blockIdx.x * blockDim.x + threadIdx.x; pi < npairs; pi += gridDim.x * blockDim.x) {
        if (out_off[pi + 1] == out_off[pi]) continue;
        WriteSink o(out + out_off[pi]);
        put_pair(o, f, pf, pi);
        o.flush();
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ host: queueing
int prep_make_const(const basal_params &p, PrepConst &k) {
    memset(&k, 0, sizeof k);
    k.K = p.seed_size; k.I = p.index_interval; k.max_readlen = p.max_readlen; k.min_read_size = p.min_read_size; k.max_ns = p.max_ns;
    k.trim_qual = p.trim_qual_threshold; k.zero_qual = p.zero_qual; k.default_qual = p.default_qual; k.n_adapter = p.n_adapter > 10 ? 10 : p.n_adapter;
    k.max_snp_num = p.max_snp_num; k.gap = p.gap; k.chains = p.chains; k.out_unmap = p.out_unmap; k.out_ref = p.out_ref;
    k.report_repeat_hits = p.report_repeat_hits;
    for (uint32_t a = 0; a < k.n_adapter; a++) {
        size_t al = strnlen(p.adapter[a], 127);
        if (al > 15) al = 15;
        k.adapter_len[a] = (uint8_t)al;
        memcpy(k.adapter[a], p.adapter[a], al);
    }
    memcpy(k.useful_nt, p.useful_nt, 8);
    return BASAL_OK;
}

static uint32_t grid_for(uint64_t items, uint32_t per_block, const basal_core *c) {
    uint64_t want = (items + per_block - 1) / per_block;
    uint64_t cap = (uint64_t)c->prop.multiProcessorCount * 16;
    if (want < 1) want = 1;
    return (uint32_t)(want < cap ? want : cap);
}

size_t prep_cub_tmp_bytes(uint32_t max_reads, uint64_t max_bytes) {
    size_t a = 0, b = 0, d = 0, e = 0;
    hipcub::DeviceScan::ExclusiveSum(nullptr, a, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)(max_bytes / kTextBlock + 2));
    hipcub::DeviceScan::ExclusiveScan(nullptr, b, (int32_t *)nullptr, (int32_t *)nullptr, hipcub::Max(), (int32_t)-1, (size_t)max_reads);
    hipcub::DeviceScan::ExclusiveScan(nullptr, d, (uint16_t *)nullptr, (uint16_t *)nullptr, hipcub::Max(), (uint16_t)0, (size_t)max_reads);
    hipcub::DeviceScan::ExclusiveSum(nullptr, e, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (size_t)max_reads + 1);
    size_t m = a > b ? a : b;
    m = m > d ? m : d;
    m = m > e ? m : e;
    return m + 256;
}

int prep_enqueue_index_text(basal_core *c, SlotDev &s, const PrepShared &sh, uint32_t batch_no, uint64_t nbytes, int format, uint32_t first_index, uint32_t read_end,
                            uint32_t readset, uint32_t max_reads, uint32_t pair_split, uint32_t pair_n, hipStream_t st) {
    const uint32_t nblk = (uint32_t)((nbytes + kTextBlock - 1) / kTextBlock);
    if (nblk == 0) return BASAL_OK;  // an empty text: counters stay zero
    hipLaunchKernelGGL(nl_count, dim3(nblk), dim3(256), 0, st, s.text, (unsigned long long)nbytes, s.blk_cnt);
    size_t tb = s.cub_tmp_bytes;
    HIP_TRYP(hipcub::DeviceScan::ExclusiveSum(s.cub_tmp, tb, s.blk_cnt, s.blk_cnt, (size_t)nblk, st));
    const uint32_t nl_cap = 4 * max_reads;
    hipLaunchKernelGGL(nl_write, dim3(nblk), dim3(256), 0, st, s.text, (unsigned long long)nbytes, s.blk_cnt, nblk, s.nl, nl_cap, s.cnt);
    hipLaunchKernelGGL(build_raw, dim3(grid_for(max_reads, 256, c)), dim3(256), 0, st, s.text, (unsigned long long)nbytes, s.nl, nl_cap, format == BASAL_FMT_FASTA ? 1 : 0,
                       readset, first_index, read_end, c->p.max_readlen, max_reads, pair_split, pair_n, sh.carry[batch_no % sh.ncarry], s.raw, s.cnt);
    HIP_TRYP(hipGetLastError());
    return BASAL_OK;
}

int prep_enqueue_filter(basal_core *c, const PrepConst &k, SlotDev &s, const PrepShared &sh, uint32_t batch_no, uint32_t max_reads, bool n_on_device, uint32_t n_host,
                        hipStream_t st) {
    const uint32_t *n_ptr = n_on_device ? &s.cnt->n_reads : nullptr;
    const CarryState *cin = sh.carry[batch_no % sh.ncarry];
    CarryState *cout = sh.carry[(batch_no + 1) % sh.ncarry];
    HIP_TRYP(hipMemsetAsync(s.npos, 0, (size_t)2 * max_reads * sizeof(uint16_t), st));
    HIP_TRYP(hipMemsetAsync(s.defidx, 0xFF, (size_t)2 * max_reads * sizeof(int32_t), st));
    hipLaunchKernelGGL(filter_reads, dim3(grid_for(max_reads, 256, c)), dim3(256), 0, st, k, s.text, c->d_tables + 512, s.raw, n_ptr, n_host, s.desc, s.aux, s.npos,
                       s.defidx, s.order, max_reads, s.cnt);
    // defscan = exclusive max-scan of defidx (per slot): the last earlier read that defined the start offset
    int32_t *defscan = s.defidx + (size_t)2 * max_reads;
    uint16_t *rev = s.npos + (size_t)2 * max_reads, *sufmax = s.npos + (size_t)4 * max_reads;
    for (int slot = 0; slot < 2; slot++) {
        size_t tb = s.cub_tmp_bytes;
        HIP_TRYP(hipcub::DeviceScan::ExclusiveScan(s.cub_tmp, tb, s.defidx + (size_t)slot * max_reads, defscan + (size_t)slot * max_reads, hipcub::Max(), (int32_t)-1,
                                                   (size_t)max_reads, st));
    }
    hipLaunchKernelGGL(block_maxima, dim3(max_reads / 4096, 2), dim3(256), 0, st, s.npos, max_reads, s.bmax1, s.bmax2);
    hipLaunchKernelGGL(ghost_install, dim3(1), dim3(512), 0, st, cin, s.text, (unsigned long long)s.text_cap, s.desc, max_reads);
    hipLaunchKernelGGL(stale_table, dim3(grid_for(max_reads, 256, c)), dim3(256), 0, st, k, s.text, c->d_tables, n_ptr, n_host, s.desc, s.npos, s.bmax1, s.bmax2, defscan,
                       max_reads, cin, s.stales, s.cnt);
    hipLaunchKernelGGL(class_lists, dim3(grid_for(max_reads, 256, c)), dim3(256), 0, st, n_ptr, n_host, s.desc, s.stales, max_reads, s.order, s.cnt);
    // next batch's carry state
    hipLaunchKernelGGL(reverse_npos, dim3(grid_for(max_reads, 256, c)), dim3(256), 0, st, s.npos, max_reads, rev);
    for (int slot = 0; slot < 2; slot++) {
        size_t tb = s.cub_tmp_bytes;
        HIP_TRYP(hipcub::DeviceScan::ExclusiveScan(s.cub_tmp, tb, rev + (size_t)slot * max_reads, sufmax + (size_t)slot * max_reads, hipcub::Max(), (uint16_t)0,
                                                   (size_t)max_reads, st));
    }
    uint32_t *list = s.order + (size_t)3 * max_reads, *list_n = list + 2 * kStackMax;
    HIP_TRYP(hipMemsetAsync(list_n, 0, 2 * sizeof(uint32_t), st));
    hipLaunchKernelGGL(stack_collect, dim3(grid_for(max_reads, 256, c)), dim3(256), 0, st, s.npos, sufmax, n_ptr, n_host, max_reads, list, list_n);
    hipLaunchKernelGGL(carry_update, dim3(1), dim3(512), 0, st, s.text, s.desc, s.npos, defscan, s.defidx, n_ptr, n_host, max_reads, list, list_n, cin, cout, batch_no, s.cnt);
    HIP_TRYP(hipGetLastError());
    return BASAL_OK;
}

int prep_enqueue_format(basal_core *c, const PrepConst &k, SlotDev &s, const PrepShared &sh, uint32_t max_reads, hipStream_t st) {
    FmtCtx f;
    f.k = k; f.text = s.text; f.raw = s.raw; f.aux = s.aux; f.res = s.results; f.stream = s.stream; f.names = sh.names; f.name_off = sh.name_off;
    f.xref_fwd = c->d_xref[0]; f.anchor = c->d_anchor; f.ncontig = c->ncontig;
    const uint32_t *n_ptr = &s.cnt->n_reads;
    hipLaunchKernelGGL(sam_lengths, dim3(grid_for((uint64_t)max_reads + 1, 256, c)), dim3(256), 0, st, f, n_ptr, 0u, max_reads, s.out_off, s.cnt);
    size_t tb = s.cub_tmp_bytes;
    HIP_TRYP(hipcub::DeviceScan::ExclusiveSum(s.cub_tmp, tb, s.out_off, s.out_off, (size_t)max_reads + 1, st));
    hipLaunchKernelGGL(sam_write, dim3(grid_for(max_reads, 256, c)), dim3(256), 0, st, f, n_ptr, 0u, s.out_off, s.out, (unsigned long long)s.out_cap, s.cnt);
    HIP_TRYP(hipGetLastError());
    return BASAL_OK;
}


// paired-end: FixPairReadName + the every-mode mark, right behind filter_reads (n = 2 * npairs records, known on the host)
int prep_enqueue_pair_fix(basal_core *c, SlotDev &s, uint32_t npairs, hipStream_t st) {
    if (!npairs) return BASAL_OK;
    hipLaunchKernelGGL(pair_fix, dim3(grid_for(npairs, 256, c)), dim3(256), 0, st, s.text, s.raw, s.desc, s.aux, npairs, s.cnt);
    HIP_TRYP(hipGetLastError());
    return BASAL_OK;
}

// paired-end: the records of the pairing kernel (s.pe_pairs / s.pe_recs) -> SAM text in s.out
int prep_enqueue_format_pe(basal_core *c, const PrepConst &k, SlotDev &s, const PrepShared &sh, uint32_t npairs, uint32_t max_reads, hipStream_t st) {
    FmtCtx f;
    f.k = k; f.text = s.text; f.raw = s.raw; f.aux = s.aux; f.res = s.results; f.stream = s.stream; f.names = sh.names; f.name_off = sh.name_off;
    f.xref_fwd = c->d_xref[0]; f.anchor = c->d_anchor; f.ncontig = c->ncontig;
    PeFmt pf{s.pe_pairs, s.pe_recs};
    hipLaunchKernelGGL(pe_lengths, dim3(grid_for((uint64_t)max_reads + 1, 256, c)), dim3(256), 0, st, f, pf, npairs, max_reads, s.out_off);
    size_t tb = s.cub_tmp_bytes;
    HIP_TRYP(hipcub::DeviceScan::ExclusiveSum(s.cub_tmp, tb, s.out_off, s.out_off, (size_t)max_reads + 1, st));
    hipLaunchKernelGGL(pe_write, dim3(grid_for(npairs ? npairs : 1, 256, c)), dim3(256), 0, st, f, pf, npairs, s.out_off, s.out, (unsigned long long)s.out_cap, s.cnt);
    HIP_TRYP(hipGetLastError());
    return BASAL_OK;
}

}  // namespace basal
