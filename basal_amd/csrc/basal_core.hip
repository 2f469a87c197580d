// basal_core.hip -- the GPU core of the BASAL seed-and-extend path for MI355X (gfx950 / CDNA4).
//
// One read per 64-lane wavefront, persistent waves pulling chunks of 8 reads from an atomic queue (descriptors in with
// one load, results out with one store, each read's bytes requested while the read before it is aligned).
// For each read a wave
//   1. packs the read into 2-bit planes with wave ballots          (ConvertBina(r)ySeq, align.cpp:79-226)
//   2. hashes the 3-letter seeds that can be asked for and gathers their index counts
//                                                                  (xseed_array + CountSeeds' index2[].n[0])
//   3. orders the seed segments from a CountSeeds(n, start) table  (ReorderSeed/AdjustSeedStartArray, align.cpp:468-540)
//   4. per mode, fans the candidate locations of all phases out over the lanes, 64 at a time, their loads issued one
//      chunk ahead: coalesced location + flank words, a conversion-tolerant flank pre-filter (a lower bound of the
//      mismatch count), one gather of reference words per surviving lane, XOR/AND + popcount scoring
//                                                                  (SnpAlign/CountMismatch*, align.cpp:274-316, align.h:118-239)
//      and the bit-parallel single-gap search                      (GapAlign/MismatchPattern*, align.cpp:348-410);
//      with -g the stream carries 64 + 32 reference bases per candidate as bit planes, and bounds of the ungapped count and of the
//      gap search at every shift drop the candidates neither can accept before the reference is touched
//   5. replays the accepted candidates IN VISITATION ORDER through the sequential hit state
//      machine (bounds, de-dup, per-level cap, threshold tightening; AddHit align.h:329-347,
//      int2hit align.cpp:319-346), which is what makes the result bit-identical; the first 64 hits of a read
//      live in registers, a Bloom filter in one more register says which keys the rest of the log cannot hold
//   6. selects what StringAlign (align.cpp:583-612) would print.
// All arithmetic is integer/bitwise (no MFMA). DESIGN.md section 4.1 has the measurements: memory gathers, instruction
// issue and per-read latency all bound it about equally.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/basal_core.h"
#include "basal_bits.h"
#include "basal_internal.h"
#include "basal_core_priv.h"

using namespace basal;

namespace {

thread_local std::string g_err;

#define HIP_TRY(x)                                                                                     \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) {                                                                        \
            g_err = std::string(#x) + ": " + hipGetErrorString(e_);                                    \
            return BASAL_EDEVICE;                                                                      \
        }                                                                                              \
    } while (0)

// ------------------------------------------------------------------------------------------
// device-side context (passed by value to the kernel)
struct DevCtx {
    const uint64_t *xref[2];
    const uint32_t *ref_anchor, *contig_size, *rc_offset;
    uint32_t ncontig;
    const uint32_t *kmer_off, *kmer_nfwd, *locs;
    const uint64_t *flank_a, *flank_b;  // 32 reference bases after / before each index entry's seed
    const uint32_t *seedw;              // HEAVY kernels: each index entry's own 16 bases
    uint32_t max_kmer_num;
    uint32_t heavy_m;     // HEAVY kernels: a list of at least this many entries is streamed on its own through the three-window test (heavy_mode)
    uint32_t win2_min_T;  // a mode's stream of at least this many candidates tests filter survivors against the second window (process_read); ~0u = never
    uint32_t K, I, max_num_hits, chains, randseed, gap, gap_edge, n_mis, stream_mode, report_repeat_hits;
    const uint8_t *tables;  // alphabet, rev_alphabet, reg_alphabet, alphabet_mread, rev_alphabet_mread
    const uint8_t *bases;
    const basal_read *reads;
    uint32_t n;
    const basal_stale *stales;
    uint32_t nstale;
    basal_result *results;
    basal_hit *stream;
    unsigned long long stream_cap;
    unsigned long long *stream_used;
    basal_hit *scratch;  // per-wave hit log
    uint32_t scratch_per_wave;
    unsigned int *work_counter;
    uint8_t carry[2][2];
    // bounds of the gathered arrays and a fault ledger: an index outside its array is clamped and
    // counted (per kind) instead of being dereferenced, so an internal error surfaces as
    // BASAL_EDEVICE on the host rather than as a GPU memory fault
    uint32_t total_kmers, nlocs;
    unsigned long long nwords, nbases;
    unsigned int *guard;  // [0..7] counts, [8..15] first offending value, [16..23] read number
    // batch pipeline (basal_pipe.hip): the reads to align as a list of read numbers (one list per read-length class, built on the
    // device) and its length in device memory, so that a batch whose size only the GPU knows yet can be queued without a host sync;
    // ghost_base: descriptors [ghost_base, ghost_base + 2) are the reads a stale read of this batch inherits its start offset from
    // when they lie in an EARLIER batch (their bytes are kept on the device)
    const uint32_t *order;
    const uint32_t *n_ptr;
    uint32_t ghost_base;
    // GAP cores: both strands once more as bit planes, one 16-byte block per 64 bases (.x = high bits of the base codes, .y = low bits, bit i =
    // base i of the block): a gap-search candidate's bitmaps at every shift are funnel shifts + three bit-selects per 64 bases (gap_flush)
    const ulonglong2 *xpl[2];
    uint32_t lds_poison;  // check build: 0x100 | byte = fill the LDS with that byte at kernel start (BASAL_POISON)
};

enum { G_KMER = 0, G_LOCS = 1, G_XREF = 2, G_BASES = 3, G_LDSPOS = 4, G_STALE = 5, G_KMER2 = 6, G_WATCHDOG = 7 };

// Diagnostic build only (-DBASAL_PHASE_TIMING, `make prof`): per-phase shader-clock totals, summed over all waves
// into d_counter[32..] and printed by basal_core_sync_check. The shipped library compiles these macros away.
enum { PH_QUEUE = 0, PH_PACK, PH_SEEDS, PH_REORDER, PH_MODE, PH_FILTER, PH_SCORE, PH_REPLAY, PH_FINAL, PH_CHUNK, PH_ENTRY, PH_BYTES, PH_E1, PH_E2, PH_E3, PH_N };
#ifdef BASAL_PHASE_TIMING
struct PhaseClock { uint64_t last; uint64_t acc[PH_N]; uint64_t n_chunks, n_alive, n_hits, n_bigchunks, n_bigalive; };
#define PH_PARAM , PhaseClock &phc
#define PH_ARG , phc
#define PH(k) do { uint64_t t_ = __builtin_readcyclecounter(); phc.acc[k] += t_ - phc.last; phc.last = t_; } while (0)
#else
#define PH_PARAM
#define PH_ARG
#define PH(k) do { } while (0)
#endif

// Cold kernel arguments are re-read from the kernarg segment where they are used (scalar loads through the
// constant cache) instead of being kept live in SGPRs for the whole kernel, where they were spilled to VGPR
// lanes (v_writelane/v_readlane are VALU work, and VALU issue is what bounds this kernel).
typedef const DevCtx __attribute__((address_space(4))) *ColdCtxPtr;
__device__ __forceinline__ ColdCtxPtr cold_ctx() {
    ColdCtxPtr p = (ColdCtxPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
#define COLD(f) (cold_ctx()->f)
// a pointer argument fetched that way: tell the compiler it points to global memory (otherwise: flat_load)
#define COLDP(T, f) ((T *)(T __attribute__((address_space(1))) *)(cold_ctx()->f))

__device__ __forceinline__ unsigned long long guard_idx(const DevCtx &cx, int kind, unsigned long long idx, unsigned long long lim, uint32_t r) {
    if (__builtin_expect(idx < lim, 1)) return idx;
    if (atomicAdd(&COLDP(unsigned int, guard)[kind], 1u) == 0) {
        COLDP(unsigned int, guard)[8 + kind] = (unsigned int)idx;
        COLDP(unsigned int, guard)[16 + kind] = r;
    }
    return 0;
}

__device__ __forceinline__ uint32_t guard_u32(const DevCtx &cx, int kind, uint32_t idx, uint32_t lim, uint32_t r) {
    if (__builtin_expect(idx < lim, 1)) return idx;
    return (uint32_t)guard_idx(cx, kind, idx, lim, r);
}

struct SeedEntBase {  // one (chain, phase) seed of the current mode
    uint32_t off, m, nfwd, jj0, pre;  // pre = number of candidates before this seed in the mode's stream
    uint32_t hcs;                     // h | chain << 16 | side << 17; side: 0 = test the flank after the seed, 1 = the one before it
};
template <bool GAP>
struct SeedEntT : SeedEntBase {};  // GAP kernels keep their windows as bit planes (SeedEntPl)
template <>
struct SeedEntT<false> : SeedEntBase {
    uint64_t fr, fm, fc;              // the read's bases / valid mask / convert-to plane opposite that flank
};

// 32 read bases starting at read position p (may be negative or run past the read: those bases come
// out with a zero valid mask) from a zero-padded MSB-first plane
template <int NWT>
__device__ __forceinline__ uint64_t plane_window(const uint64_t *q, int p) {
    if (p <= -32 || p >= NWT * 32) return 0;
    if (p < 0) return q[0] >> (2 * (-p));
    uint32_t w = (uint32_t)p >> 5, sh = ((uint32_t)p & 31) * 2;
    return sh ? (q[w] << sh) | (q[w + 1] >> (64 - sh)) : q[w];
}

// the same window of the three planes of one chain (bases, valid mask, convert-to), one control flow for all
template <int NWT, bool NEED_C>
__device__ __forceinline__ void plane_window3(const uint64_t (*q)[NWT + 1], int p, uint64_t &a, uint64_t &b, uint64_t &c) {
    a = b = c = 0;
    if (p <= -32 || p >= NWT * 32) return;
    if (p < 0) {
        const int s = 2 * (-p);
        a = q[0][0] >> s; b = q[1][0] >> s;
        if (NEED_C) c = q[2][0] >> s;
        return;
    }
    const uint32_t w = (uint32_t)p >> 5, sh = ((uint32_t)p & 31) * 2;  // (x >> 1) >> (63 - sh) is x >> (64 - sh), and 0 for sh = 0
    a = (q[0][w] << sh) | ((q[0][w + 1] >> 1) >> (63 - sh));
    b = (q[1][w] << sh) | ((q[1][w + 1] >> 1) >> (63 - sh));
    if (NEED_C) c = (q[2][w] << sh) | ((q[2][w + 1] >> 1) >> (63 - sh));
}

// Param::profile[j][i] (param.cpp:70-74), one copy per workgroup instead of an integer division per use
__shared__ uint16_t s_prof[16][16];
// the contig table of references with at most 64 contigs (int2hit then needs no memory access)
__shared__ uint32_t s_anchor[64], s_rcoff[64], s_csize[64];
// ceil(2^16 / d) for d = 1..32: x / d == (x * s_rcp[d]) >> 16 for x < 2048
__shared__ uint32_t s_rcp[33];

// GAP kernels: what the stream filter needs of the read opposite one seed's flanks, as bit planes (bit i = window base i, LSB first).
// ml/ms[X]: the read base there mismatches reference letter X (zero outside the read; N's compare as their code, as in MismatchPattern0/1);
// vl/vs: the base is a valid one inside the read. The long window (64 bases) lies on the side of the seed with more read bases
// (SeedEnt::side: 0 = from h+K on, 1 = the 64 bases ending at h), the short one (32 bases) on the other side, next to the seed.
struct SeedEntPl {
    uint64_t ml[4], vl;
    uint32_t ms[4], vs, pad;
};

__device__ __forceinline__ uint64_t bsel(uint64_t s, uint64_t a, uint64_t b) { return (s & a) | (~s & b); }  // v_bfi_b32 x2
__device__ __forceinline__ uint32_t bsel(uint32_t s, uint32_t a, uint32_t b) { return (s & a) | (~s & b); }
// mismatch bitmap of a window: per base the mask of the reference letter (2 * hi + lo) that is there
__device__ __forceinline__ uint64_t plane_mismatch(uint64_t hi, uint64_t lo, const uint64_t m[4]) { return bsel(lo, bsel(hi, m[3], m[1]), bsel(hi, m[2], m[0])); }
__device__ __forceinline__ uint32_t plane_mismatch(uint32_t hi, uint32_t lo, const uint32_t m[4]) { return bsel(lo, bsel(hi, m[3], m[1]), bsel(hi, m[2], m[0])); }
// The same for a SHIFTED window whose vacated bits are masked off afterwards: written as (s & a) | (~s & b) the compiler folds that mask into the
// outer select and spends four instructions per word on it (and, xor, and, or); the instruction itself keeps it at one.
__device__ __forceinline__ uint32_t bfi_hw(uint32_t s, uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(s), "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint64_t plane_mismatch_sh(uint64_t hi, uint64_t lo, const uint64_t m[4]) {
    const uint64_t x = bsel(hi, m[3], m[1]), y = bsel(hi, m[2], m[0]);
    return ((uint64_t)bfi_hw((uint32_t)(lo >> 32), (uint32_t)(x >> 32), (uint32_t)(y >> 32)) << 32) | bfi_hw((uint32_t)lo, (uint32_t)x, (uint32_t)y);
}

// bits [start, start + 64) of an NW-word bit string (LSB first) kept as pl[1..NW] between two zero words; zero outside it
template <int NW>
__device__ __forceinline__ uint64_t bits64(const uint64_t *pl, int start) {
    int s = start + 64;
    s = s < 0 ? 0 : s > (NW + 1) * 64 - 1 ? (NW + 1) * 64 - 1 : s;  // a start left of -64 or right of the string: all zero either way
    const uint32_t w = (uint32_t)s >> 6, sh = (uint32_t)s & 63;
    return (pl[w] >> sh) | ((pl[w + 1] << 1) << (63 - sh));
}

#ifndef WORK_CHUNK
#define WORK_CHUNK 8  // reads a wave takes from the queue per atomic
#endif
#ifndef BASAL_PE_ENT
#define BASAL_PE_ENT 32  // PE kernels: seed entries of a mode group. 32 = four modes at -I 4 in the LDS the standard kernels have (config 3: 16.4 ms per 1 M pairs);
#endif                   // 64 = eight modes, 1.5 KB more LDS per wave and a block less per CU (17.4 ms)
#ifndef BLOOM_WORDS
#define BLOOM_WORDS 256  // 8 192 bits: 123.0 -> 119.0 ms per 10 M reads against 4 096 (fewer look-ups of the memory log for reads with a thousand hits); 16 384 would cost the fifth block per CU
#endif
#ifndef LONG_WD
#define LONG_WD 2  // HEAVY kernels: chunks of 64 candidates of a long list per memory round trip (heavy_mode)
#endif
#ifndef LONG_WD_NWT
#define LONG_WD_NWT 8  // ... in the kernels for reads of up to 32 * this many bases
#endif

struct SurvEnt {  // a gap-eligible or still-alive candidate of the stream
    uint32_t loc;   // alignment start (bounds-checked)
    uint32_t meta;  // seed entry | strand << 8 | passed the ungapped flank bound << 9 | passed the gap-side bound << 10
};

// what only the GAP kernels keep per wave (an empty base otherwise: the other kernels' LDS is full at 8 blocks per CU)
template <int NWT, bool GAP>
struct GapLds {};
template <int NWT>
struct GapLds<NWT, true> {
    SeedEntPl entp[32];
    // per chain and reference letter, "this read base mismatches it" (1 bit per base, LSB first), and "valid base inside the read";
    // a zero word on either side of the NWT/2 data words, so that a window at any position is two loads and a funnel shift, no branches
    uint64_t mmp[2][4][NWT / 2 + 2];
    uint64_t valp[2][NWT / 2 + 2];
};
// HEAVY GAP kernels: per seed of the mode, which bits of its two windows are among the read's first / last gap_edge bases. GapAlign never puts
// the gap inside those (align.cpp:385, 391): the first gap_edge bases always align at the candidate's start and the last gap_edge at the
// shifted start, so there a mismatch counts even if the other start matches -- the stream's bound (iii) with teeth for diverged copies.
struct SeedEndMask {
    uint64_t fl, el;  // long window: bits among the read's first / last gap_edge bases
    uint32_t fs, es;  // short window
};
template <bool ON>
struct EndLds {};
template <>
struct EndLds<true> {
    SeedEndMask entx[32];
};
// the candidates the stream's tests could not rule out, in visitation order (GAP and HEAVY kernels), scored 64 at a time
template <bool ON>
struct SurvLds {};
template <>
struct SurvLds<true> {
    alignas(16) SurvEnt surv[128];  // GAP: see SurvEnt; HEAVY non-GAP: meta = reference strand | read chain << 1; counted from the stream already: | 4 | the count << 3 | the seed's read offset << 11, and loc = the index entry
};

// what only the HEAVY kernels keep per wave: the candidates that passed the stream's window tests, in visitation order, scored 64 at a time
template <bool HEAVY, int NWT>
struct HeavyLds {};
template <int NWT>
struct HeavyLds<true, NWT> {
    uint32_t bloom[NWT <= 8 ? BLOOM_WORDS : 128];  // a Bloom filter over the keys of ALL stored hits of the read (bulk_add); the 480-base kernels keep 4 096 bits, their LDS holds three blocks per CU only so
    uint32_t bucket[32];  // bulk_add: the lowest lane of each key-hash bucket (with 64 buckets the block's LDS would not fit six times into a CU)
};

// PE (the paired-end instantiations of the standard kernels): a mate runs every mode, so several modes' seeds (BASAL_PE_ENT entries: four modes at
// -I 4) are set up and streamed as ONE group, and the log remembers where its records were found (HitState::g)
// (its own base, PE only: eight bytes more per wave took the standard GAP kernels from 25 to 26 LDS allocation units of 1 280 B per block -- and from five
// blocks per CU to four, 101 -> 114 ms per 10 M reads of config 4)
template <bool PE>
struct StageLds {};
template <>
struct StageLds<true> {
    uint32_t stg_n, stg_mask;  // hit-stream records staged for the reads of this chunk (stream_flush), and which of the chunk's reads they belong to
};
template <int NWT, bool GAP, bool HEAVY = false, bool PE = false>
struct WaveLds : GapLds<NWT, GAP>, HeavyLds<HEAVY, NWT>, SurvLds<GAP || HEAVY>, EndLds<GAP && HEAVY>, StageLds<PE> {
    static constexpr int NW = NWT;
    static constexpr bool PEK = PE;
    // (seed positions of a read: fewer than BASAL_MAXREADLEN; 496 instead of 512 keeps the 16-word HEAVY kernels' block at 42 LDS allocation units, three blocks per CU)
    static constexpr int MAXPOS = NWT * 32 > BASAL_MAXREADLEN + 16 ? BASAL_MAXREADLEN + 16 : NWT * 32;
    uint64_t q[2][3][NWT + 1];  // [chain][bases, valid, convert-to][word]; last word always 0
    uint32_t seed[2][MAXPOS];   // XT hash; bit 31: seed window contains a non-ACGT base
    uint32_t cnt[2][MAXPOS];    // index2[seed].n[0]
    union {
        SeedEntT<GAP> ent[PE ? BASAL_PE_ENT : 32];  // the current mode's seeds (PE: of up to eight modes)
        uint32_t cs[16][16];  // before the first mode: CountSeeds(n, start) of the chain being ordered
    };
    static constexpr bool GAPK = GAP;
    uint32_t rno[WORK_CHUNK];      // their read numbers (consecutive, or taken from the pipeline's list)
    basal_read desc[WORK_CHUNK];   // the descriptors of the chunk of reads this wave took from the queue
    basal_result res[WORK_CHUNK];  // and their results, written out together when the chunk is done
    uint32_t nhit[2][16];  // x_cur_n_hit[chain][level]
    uint8_t start_arr[2][16];
    uint8_t order[2][16];
};

// LDS written by one lane and read by the others of the SAME wave: LDS ops of a wave execute in
// order, so only the compiler has to be told not to move or cache accesses across this point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// "Is this lane 0?" with the lane id laundered through an empty asm: LLVM must not correlate two
// such tests. Without this, jump threading joined the `if (lane0(lane)) store` that ends one read
// with the `if (lane0(lane)) atomicAdd` that fetches the next, so lane 0 and lanes 1..63 ran the
// work loop on separate paths and the wave-level operations (readfirstlane, ballot) saw partial
// waves: lanes 1..63 re-ran read 0 with lane 0 missing (found on MI355X, ROCm 7.2, -O2/-O3).
__device__ __forceinline__ bool lane0(int lane) {
    asm volatile("" : "+v"(lane));
    return lane == 0;
}

// HIP's __ballot takes an int and compares it with 0 again (v_cndmask + v_cmp on top of the compare that made it)
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint64_t rdlane64(uint64_t v, int l) {
    return ((uint64_t)rdlane((uint32_t)(v >> 32), l) << 32) | rdlane((uint32_t)v, l);
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t w = __shfl_xor(v, o);
        v = w < v ? w : v;
    }
    return v;
}

// Reductions / scans over lanes 0..15 with DPP row shifts (4 VALU steps instead of 6 ds_bpermute round trips).
// row_shr:n = dpp_ctrl 0x110|n: lane i reads lane i-n of its 16-lane row; lanes without a source keep `old`.
template <int N>
__device__ __forceinline__ uint32_t dpp_row_shr(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x110 | N, 0xf, 0xf, false);
}
// inclusive prefix sum within each 16-lane row
__device__ __forceinline__ uint32_t row16_scan_add(uint32_t v) {
    v += dpp_row_shr<1>(0, v);
    v += dpp_row_shr<2>(0, v);
    v += dpp_row_shr<4>(0, v);
    v += dpp_row_shr<8>(0, v);
    return v;
}
// min / sum of lanes 0..15, broadcast to the whole wave
__device__ __forceinline__ uint32_t row16_min(uint32_t v) {
    uint32_t t;
    t = dpp_row_shr<1>(v, v); v = t < v ? t : v;
    t = dpp_row_shr<2>(v, v); v = t < v ? t : v;
    t = dpp_row_shr<4>(v, v); v = t < v ? t : v;
    t = dpp_row_shr<8>(v, v); v = t < v ? t : v;
    return rdlane(v, 15);
}
__device__ __forceinline__ uint32_t row16_sum(uint32_t v) { return rdlane(row16_scan_add(v), 15); }

// every bit of a wave-uniform 32-bit value doubled (bit j -> bits 2j and 2j+1): one scalar instruction on gfx9
__device__ __forceinline__ uint64_t bitreplicate(uint32_t x) {
    uint64_t r;
    asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(r) : "s"(x));
    return r;
}

// 64 lanes each hold one 2-bit code; returns the two packed words (lanes 0-31, lanes 32-63),
// base of lane i at bits [63-2i, 62-2i] (MSB first, as ConvertBinaySeq packs, align.cpp:88-105)
__device__ __forceinline__ void pack_codes(uint32_t code, uint64_t &w0, uint64_t &w1) {
    uint64_t b0 = ballot(code & 1), b1 = ballot(code & 2);
    w0 = (bitreplicate(__brev((uint32_t)b1)) & ~kPairLo) | (bitreplicate(__brev((uint32_t)b0)) & kPairLo);
    w1 = (bitreplicate(__brev((uint32_t)(b1 >> 32))) & ~kPairLo) | (bitreplicate(__brev((uint32_t)(b0 >> 32))) & kPairLo);
}

// the word 32 lanes holding the same 2-bit code pack to
__device__ __forceinline__ uint64_t code_fill(uint32_t code) { return ((code & 2) ? ~kPairLo : 0) | ((code & 1) ? kPairLo : 0); }

struct ReadCtx {
    uint32_t len, index, readset, max_snp, seq_off;
    uint32_t nseg, ii, npos;
    uint32_t end_element, end_offset;
    uint32_t flags;  // bit c: read chain c is aligned (xflag_chain)
    uint32_t n_count;
    uint32_t rno;  // read number in the batch (diagnostics)
    __device__ __forceinline__ bool on(int c) const { return (flags >> c) & 1u; }
};

// xflag_chain (align.cpp:83-84): bit c set = read chain c is aligned
__device__ __forceinline__ uint32_t chain_flags(const DevCtx &cx, uint32_t rs) {
    return (((cx.chains == 1) || ((cx.chains <= 1) == (rs < 2))) ? 1u : 0u) | (((cx.chains == 1) || ((cx.chains <= 1) == (rs == 2))) ? 2u : 0u);
}

// the read's bytes as chain c sees them (c = 1: back to front), lane l of block b <- read position 64 b + l
template <int NWT>
__device__ __forceinline__ void load_bases(const DevCtx &cx, const basal_read &rd, uint32_t rno, int c, int lane, uint32_t chs[NWT / 2]) {
    uint32_t len = rd.len <= (uint32_t)NWT * 32 ? rd.len : 0;  // over-long reads are skipped by process_read
    // one wave-uniform bounds check for the whole read instead of one per byte
    if ((unsigned long long)rd.seq_off + len > COLD(nbases)) { guard_idx(cx, G_BASES, (unsigned long long)rd.seq_off + len, 0, rno); len = 0; }
    const uint8_t *p = cx.bases + rd.seq_off;
#pragma unroll
    for (uint32_t b = 0; b < (uint32_t)NWT / 2; b++) {
        uint32_t pos = b * 64 + lane;
        chs[b] = 0;
        if (pos < len) chs[b] = p[c ? len - 1 - pos : pos];
    }
}

// a read descriptor as the wave-uniform value it is (scalar registers)
__device__ __forceinline__ basal_read uniform_read(const basal_read &v) {
    basal_read u;
    u.seq_off = rfl(v.seq_off); u.index = rfl(v.index); u.stale_idx = rfl(v.stale_idx);
    uint32_t w = rfl((uint32_t)v.len | ((uint32_t)v.readset << 16) | ((uint32_t)v.max_snp << 24));
    u.len = (uint16_t)w; u.readset = (uint8_t)(w >> 16); u.max_snp = (uint8_t)(w >> 24);
    return u;
}

// ---- steps 1+2: pack, hash seeds, gather counts --------------------------------------------
// pre/pre_c: the bytes of chain pre_c fetched ahead by the caller (pre_c < 0: none)
// so0/so1: the start offsets the read inherits (they bound the seed positions that can be asked for)
template <bool NEWRULE, class LDS>
__device__ void prep_read(const DevCtx &cx, LDS &L, const uint8_t *tab, const basal_read &rd, ReadCtx &rc, int lane, const uint32_t *pre, int pre_c,
                          uint32_t so0, uint32_t so1 PH_PARAM) {
    constexpr int NWT = LDS::NW;
    rc.len = rd.len;
    rc.index = rd.index;
    const uint32_t rs = rd.readset & 0x7fu;
    rc.readset = rs;
    rc.max_snp = rd.max_snp;
    rc.seq_off = rd.seq_off;
    rc.flags = chain_flags(cx, rs);
    {  // seedseg_num (align.cpp:450)
        int x = (int)((rc.len - cx.I + 1) / cx.K), y = (int)(rc.max_snp + 1);
        rc.nseg = (uint32_t)(x < y ? x : y);
    }
    rc.ii = (rc.len - cx.I + 1) % cx.K;
    rc.npos = rc.len >= cx.K ? rc.len - cx.K + 1 : 0;
    rc.end_element = (rc.len - 1) / 32;                    // align.cpp:442
    rc.end_offset = (32 - ((rc.len - 1) % 32 + 1)) << 1;   // align.cpp:443
    uint32_t nblk = (rc.len + 63) / 64;
    uint32_t ncnt = 0;
    for (int c = 0; c < 2; c++) {
        if (!rc.on(c)) continue;
        const uint8_t *al = tab + (c ? 256 : 0), *am = tab + (c ? 1024 : 768), *rg = tab + 512;
        // all blocks' bytes are requested before the first is used (one memory round trip, not one per block)
        uint32_t chs[NWT / 2];
        if (c == pre_c) {
#pragma unroll
            for (int b = 0; b < NWT / 2; b++) chs[b] = pre[b];
        } else load_bases<NWT>(cx, rd, rc.rno, c, lane, chs);
#ifdef BASAL_PHASE_TIMING
        __builtin_amdgcn_s_waitcnt(0);
        PH(PH_BYTES);
#endif
#pragma unroll
        for (uint32_t b = 0; b <= (uint32_t)NWT / 2; b++) {
            uint64_t a0, a1, v0, v1, m0, m1;
            if (b < nblk) {
                const uint32_t ch = chs[b < (uint32_t)NWT / 2 ? b : 0], pos = b * 64 + lane;
                uint32_t valid = rg[ch];
                pack_codes(al[ch], a0, a1);
                pack_codes(valid, v0, v1);
                pack_codes(am[ch], m0, m1);
                if (c == 0) ncnt += __popcll(ballot(pos < rc.len && !valid));
                if constexpr (LDS::GAPK) {  // the stream filter's bit planes: this base against each reference letter (the comparison is per base)
                    const uint64_t code = al[ch], mcode = am[ch];
                    const bool in = pos < rc.len;
                    const uint64_t p0 = ballot(in && (cmp_word<NEWRULE>(code, mcode, 0) & 3)), p1 = ballot(in && (cmp_word<NEWRULE>(code, mcode, 1) & 3)),
                                   p2 = ballot(in && (cmp_word<NEWRULE>(code, mcode, 2) & 3)), p3 = ballot(in && (cmp_word<NEWRULE>(code, mcode, 3) & 3)),
                                   pv = ballot(in && (valid & 1));
                    if (lane0(lane)) { L.mmp[c][0][b + 1] = p0; L.mmp[c][1][b + 1] = p1; L.mmp[c][2][b + 1] = p2; L.mmp[c][3][b + 1] = p3; L.valp[c][b + 1] = pv; }
                }
            } else {  // past the read: what 64 lanes holding byte 0 would pack to
                a0 = a1 = code_fill(al[0]); v0 = v1 = code_fill(rg[0]); m0 = m1 = code_fill(am[0]);
                if constexpr (LDS::GAPK) if (b < (uint32_t)NWT / 2 && lane0(lane)) { L.mmp[c][0][b + 1] = L.mmp[c][1][b + 1] = L.mmp[c][2][b + 1] = L.mmp[c][3][b + 1] = 0; L.valp[c][b + 1] = 0; }
            }
            if (lane0(lane)) {
                if (2 * b < (uint32_t)NWT + 1) { L.q[c][0][2 * b] = a0; L.q[c][1][2 * b] = v0; L.q[c][2][2 * b] = m0; }
                if (2 * b + 1 < (uint32_t)NWT + 1) { L.q[c][0][2 * b + 1] = a1; L.q[c][1][2 * b + 1] = v1; L.q[c][2][2 * b + 1] = m1; }
            }
        }
    }
    if (!rc.on(0)) {  // count N's when only chain 1 is packed (CountNs, align.cpp:40-47)
        for (uint32_t b = 0; b < nblk; b++) {
            uint32_t pos = b * 64 + lane;
            uint32_t ch = pos < rc.len ? cx.bases[guard_idx(cx, G_BASES, (unsigned long long)rc.seq_off + pos, COLD(nbases), rc.rno)] : 'A';
            ncnt += __popcll(ballot(!tab[512 + ch]));
        }
    }
    rc.n_count = cx.n_mis ? ncnt : 0;
    wave_sync();
    PH(PH_PACK);
    // seeds: xseed_array / xseedreg_array (align.cpp:92-100) and their index counts.
    // Only the positions CountSeeds and SnpAlign can ask for are hashed and looked up: phase i of segment n with a
    // start offset st <= smax = max(ii, inherited offset) sits at profile[n][i] - i + st. With k = 16, I = 4 and
    // 100-base reads that is 30 distinct positions of 85, and each position not looked up is a random index access
    // saved. Duplicates (several (i, st) giving one position) fall on the same address and write the same values.
    const uint32_t kbits = 2 * cx.K;
    for (int c = 0; c < 2; c++) {
        if (!rc.on(c)) continue;
        uint32_t so = c ? so1 : so0;
        so = so < 16 ? so : 15;
        const uint32_t R = (rc.ii > so ? rc.ii : so) + 1, combos = rc.nseg * cx.I * R;
        const bool sparse = combos < rc.npos;
        const uint32_t rcpI = s_rcp[cx.I], rcpR = s_rcp[R];
        if (sparse && lane0(lane)) { L.seed[c][LDS::MAXPOS - 1] = 0x80000000u; L.cnt[c][LDS::MAXPOS - 1] = 0; }  // where CountSeeds' clip lands
        const uint32_t total = sparse ? combos : (uint32_t)LDS::MAXPOS;
        for (uint32_t base = 0; base < total; base += 64) {
            uint32_t p = base + lane;
            bool valid = p < total;
            if (sparse) {  // p -> (n, st, i) -> position
                uint32_t rest = (p * rcpI) >> 16, i = p - rest * cx.I;  // exact for p < 4096, divisor <= 16
                uint32_t n = (rest * rcpR) >> 16, st = rest - n * R;
                p = valid ? s_prof[n & 15][i & 15] + st - i : 0;
                valid = valid && p < (uint32_t)LDS::MAXPOS;
            }
            uint32_t sd = 0x80000000u, ct = 0;  // out-of-read positions: "contains N", count 0
            if (valid && p < rc.npos) {
                uint32_t w = p >> 5, sh = (p & 31) * 2;
                uint64_t a = L.q[c][0][w], b = L.q[c][1][w];
                if (sh) {
                    a = (a << sh) | (L.q[c][0][w + 1] >> (64 - sh));
                    b = (b << sh) | (L.q[c][1][w + 1] >> (64 - sh));
                }
                uint32_t s = (uint32_t)(a >> (64 - kbits)), sb = (uint32_t)(b >> (64 - kbits));
                uint32_t full = kbits == 32 ? 0xFFFFFFFFu : ((1u << kbits) - 1);
                sd = (uint32_t)guard_idx(cx, G_KMER, XT(s), COLD(total_kmers), rc.rno);
                ct = cx.kmer_off[sd + 1] - cx.kmer_off[sd];
                if ((~sb) & full) sd |= 0x80000000u;
            }
            if (valid) {
                L.seed[c][p] = sd;
                L.cnt[c][p] = ct;
            }
        }
    }
    wave_sync();
    PH(PH_SEEDS);
}

// CountSeeds (align.cpp:526-540)
template <class LDS>
__device__ __forceinline__ uint32_t count_seeds(const DevCtx &cx, const LDS &L, int c, uint32_t n, uint32_t start) {
    uint32_t total = 0, k = 0;
    for (uint32_t i = 0; i < cx.I; i++) {
        uint32_t pos = s_prof[n & 15][i] + start - i;
        pos = pos < (uint32_t)LDS::MAXPOS ? pos : (uint32_t)LDS::MAXPOS - 1;
        uint32_t s = L.seed[c][pos];
        if (s >> 31) k = 12;
        total += L.cnt[c][pos] << k;
    }
    return total == 0 ? 9999999u : total;
}

// the global start offset (align.cpp:475-480); only meaningful when rc.ii > 0
template <class LDS>
__device__ uint32_t best_start_offset(const DevCtx &cx, const LDS &L, const ReadCtx &rc, int c, int lane, uint32_t inherited) {
    uint32_t best = 0xffffffffu, so = inherited;
    for (uint32_t st = 0; st < rc.ii; st++) {
        uint32_t cs = (uint32_t)lane < rc.nseg ? count_seeds(cx, L, c, (uint32_t)lane, st) : 0;
        uint32_t tt = row16_sum(cs);  // nseg <= 16
        if (tt < best) { best = tt; so = st; }
    }
    return so;
}

// ---- step 3: the start offset, AdjustSeedStartArray and ReorderSeed -------------------------------
// CountSeeds(chain, n, start) is a pure function of (n, start) once the seeds are hashed, and the three
// users (the global start offset, align.cpp:475-480; AdjustSeedStartArray, align.cpp:500-524; the
// weights of ReorderSeed, align.cpp:492-495) only ever ask for start <= max(ii, inherited offset) < 16.
// All needed values are computed in one lane-parallel sweep into L.cs (which shares storage with the
// mode's seed entries, not live yet); the sequential walk over the segments then runs on table reads
// with start_arr held in lane registers.
template <class LDS>
__device__ void reorder_seed(const DevCtx &cx, LDS &L, const ReadCtx &rc, int lane, uint32_t &so0, uint32_t &so1) {
    const uint32_t max_offset = rc.ii;
    for (int c = 0; c < 2; c++) {
        if (!rc.on(c)) continue;
        uint32_t so = c ? so1 : so0;
        so = so < 16 ? so : 15;  // offsets are < K <= 16 by construction
        const uint32_t smax = max_offset > so ? max_offset : so;
        const uint32_t sh = smax ? 32u - (uint32_t)__builtin_clz(smax) : 0u;  // rows of 1 << sh lanes
        for (uint32_t base = 0; base < rc.nseg; base += 64u >> sh) {
            uint32_t n = base + ((uint32_t)lane >> sh), st = (uint32_t)lane & ((1u << sh) - 1);
            if (n < rc.nseg && st <= smax) L.cs[n][st] = count_seeds(cx, L, c, n, st);
        }
        wave_sync();
        if (max_offset > 0) {  // the global start offset: the first st < ii with the smallest total
            uint32_t tt = 0xffffffffu;
            if ((uint32_t)lane < max_offset) {
                tt = 0;
                for (uint32_t n = 0; n < rc.nseg; n++) tt += L.cs[n][lane];
            }
            uint32_t m = row16_min(tt);
            if (m != 0xffffffffu) so = (uint32_t)__ffsll((unsigned long long)ballot((uint32_t)lane < max_offset && tt == m)) - 1;
        }
        if (c) so1 = so; else so0 = so;
        // AdjustSeedStartArray: lane j holds start_arr[j]
        uint32_t sa = so;
        for (uint32_t i = 0; i < rc.nseg; i++) {
            uint32_t ptr = (i % 2 == 0) ? i / 2 : rc.nseg - 1 - i / 2;
            uint32_t start = (ptr == 0) ? 0 : rdlane(sa, (int)ptr - 1);
            uint32_t end = (ptr == rc.nseg - 1) ? max_offset : rdlane(sa, (int)ptr + 1);
            uint32_t pick = start;
            if (start < end) {  // (a single candidate, or none, is its own minimum: about half of the segments of a 100-base read)
                uint32_t cand = start + lane;
                bool valid = cand <= end && lane < 16;
                uint32_t tt = valid ? L.cs[ptr][cand & 15] : 0xffffffffu;
                uint32_t m = row16_min(tt);  // valid lanes are < 16
                if (m != 0xffffffffu) pick = start + (uint32_t)__ffsll((unsigned long long)(ballot(tt == m) & ballot(valid))) - 1;
            }
            if ((uint32_t)lane == ptr) sa = pick;
        }
        if (lane < 16) L.start_arr[c][lane] = (uint8_t)sa;
        // weights + sort ascending by (int weight, segment) (align.cpp:492-495)
        int32_t w = 0x7fffffff;
        if ((uint32_t)lane < rc.nseg) w = (int32_t)L.cs[lane][sa & 15];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < rc.nseg; j++) {
            int32_t wj = (int32_t)rdlane((uint32_t)w, (int)j);
            rank += (wj < w) || (wj == w && j < (uint32_t)lane);
        }
        if ((uint32_t)lane < rc.nseg) L.order[c][rank] = (uint8_t)lane;
        wave_sync();  // L.cs is rewritten for the other chain / by the seed entries
    }
}

// ---- candidate scoring ----------------------------------------------------------------------
// CountMismatch / CountMismatch_new (align.h:118-131, 199-239): the exact conversion-tolerant mismatch count
// (the reference's early exit only saves it work; the caller compares with the threshold).
template <int NWT, bool NEWRULE>
__device__ __forceinline__ uint32_t count_mismatch(const uint64_t *__restrict__ s, const uint64_t (*q)[NWT + 1], uint32_t off2, uint32_t nw,
                                                   uint32_t thr, uint32_t ncount) {
    // All reference words of the alignment are requested before any is used: after the flank pre-filter the
    // lanes that get here are mostly true hits that need every word, and one round trip beats nw of them.
    uint64_t sw[NWT + 1];
#pragma unroll
    for (int i = 0; i <= NWT; i++) sw[i] = (uint32_t)i < nw ? s[i] : 0;
    uint32_t mm = ncount;
    (void)thr;
#pragma unroll
    for (int i = 0; i <= NWT; i++) {
        if ((uint32_t)i < nw) {
            uint64_t rw, mw, cw = 0;
            if (i == 0) {
                rw = q[0][0] >> off2;
                mw = q[1][0] >> off2;
                if (NEWRULE) cw = q[2][0] >> off2;
            } else {
                rw = ((q[0][i - 1] << 1) << (63 - off2)) | (q[0][i] >> off2);
                mw = ((q[1][i - 1] << 1) << (63 - off2)) | (q[1][i] >> off2);
                if (NEWRULE) cw = ((q[2][i - 1] << 1) << (63 - off2)) | (q[2][i] >> off2);
            }
            mm += XM64(cmp_word<NEWRULE>(rw, cw, sw[i]) & mw);
        }
    }
    return mm;
}

// mismatch bitmap of the whole read against the reference starting at `loc` (reference shifted
// into the read frame, no N mask): the word loop of MismatchPattern0/1 (align.h:140-165, 179-193).
template <int NWT, bool NEWRULE>
__device__ __forceinline__ void mismatch_map(const uint64_t *__restrict__ xs, uint32_t loc, const uint64_t (*q)[NWT + 1], uint32_t end_element,
                                             uint32_t end_offset, uint64_t D[NWT]) {
    const uint64_t *s = xs + (loc >> 5);
    uint32_t off2 = (loc & 31) * 2;
    uint64_t cur = s[0];
#pragma unroll
    for (int i = 0; i < NWT; i++) {
        uint64_t d = 0;
        if ((uint32_t)i <= end_element) {
            uint64_t nxt = s[i + 1];
            uint64_t tmp = (cur << off2) | ((nxt >> (63 - off2)) >> 1);
            cur = nxt;
            if (!NEWRULE) tmp ^= q[0][i] & XC64(tmp);
            else {
                uint64_t M2 = XC64(tmp) | q[2][i];
                uint64_t M3 = M2_judge(M2);
                tmp ^= ((~M3) & M2) | (M3 & q[0][i]);
            }
            if ((uint32_t)i == end_element) tmp = (tmp >> end_offset) << end_offset;
            d = pair_mask(tmp);
        }
        D[i] = d;
    }
}

// The same bitmap cut from reference words already in registers: W[0..NWT+1] are the words from the one holding
// base `first` on, `rel` (< 64) is the alignment's first base relative to W[0]. GapAlign compares one candidate
// at up to 7 start positions loc-3..loc+3; they all lie in the same NWT+2 words, which are loaded once.
template <int NWT, bool NEWRULE>
__device__ __forceinline__ uint64_t mismatch_word_regs(const uint64_t W[NWT + 2], uint32_t rel, const uint64_t (*q)[NWT + 1], uint32_t end_element,
                                                       uint32_t end_offset, int i) {
    if ((uint32_t)i > end_element) return 0;
    const bool up = rel >= 32;
    const uint32_t off2 = (rel & 31) * 2;
    const uint64_t cur = up ? W[i + 1] : W[i], nxt = up ? W[i + 2] : W[i + 1];
    uint64_t tmp = (cur << off2) | ((nxt >> (63 - off2)) >> 1);
    if (!NEWRULE) tmp ^= q[0][i] & XC64(tmp);
    else {
        uint64_t M2 = XC64(tmp) | q[2][i];
        uint64_t M3 = M2_judge(M2);
        tmp ^= ((~M3) & M2) | (M3 & q[0][i]);
    }
    if ((uint32_t)i == end_element) tmp = (tmp >> end_offset) << end_offset;
    return pair_mask(tmp);
}
template <int NWT, bool NEWRULE>
__device__ __forceinline__ void mismatch_map_regs(const uint64_t W[NWT + 2], uint32_t rel, const uint64_t (*q)[NWT + 1], uint32_t end_element,
                                                  uint32_t end_offset, uint64_t D[NWT]) {
#pragma unroll
    for (int i = 0; i < NWT; i++) D[i] = mismatch_word_regs<NWT, NEWRULE>(W, rel, q, end_element, end_offset, i);
}

// GapAlign (align.cpp:348-410) for one candidate, bit-parallel: instead of the reference's
// position arrays mm_index[][] it keeps the mismatch bitmaps and answers "i-th mismatch from the
// left" / "first mismatch at distance >= X from the right" with popcounts. Same decisions.
template <int NWT, bool NEWRULE>
// W = the reference words from the one holding base loc-g on, rel0 = loc relative to W[0], D0 = the ungapped bitmap.
__device__ bool gap_align(const DevCtx &cx, const uint64_t W[NWT + 2], uint32_t rel0, const uint64_t D0[NWT], const uint64_t (*q)[NWT + 1], const ReadCtx &rc,
                          uint32_t thr, uint32_t seed_pos, uint32_t &gap_snp, uint32_t &gap_pos_out, int &shift_out) {
    if (thr < 2) return false;
    const int len = (int)rc.len;
    // MismatchPattern0 returns the position of mismatch #(thr-1) (or len); GapAlign gives up if
    // that lies before the end of the seed (align.cpp:365): >= thr-1 mismatches in [0, seed end)
    {
        int lim = (int)(seed_pos + cx.K);
        uint32_t c0 = 0;
#pragma unroll
        for (int w = 0; w < NWT; w++) c0 += popc64(D0[w] & prefix_pairs(lim - 32 * w));
        if (c0 >= thr - 1) return false;
    }
    // Per shift, first the cheapest test: a position that mismatches at both start positions costs one mismatch whichever
    // side of the gap it falls on (the t inserted read bases excepted), so i + j >= popc(D0 & D1) - (insertion ? t : 0) for
    // every (i, j). The shifted bitmap is built word by word and dropped as soon as that bound is reached: a random
    // candidate (they pass the seed-side test whenever the seed sits near the read start) dies after one or two words.
    // Then (lazily, once) the positions of the left-side mismatches with index thr-2, thr-3, thr-4 (the last one a gap of
    // size 1, 2, 3 may use) and of the last mismatch at all: an upper bound G of every usable gap position. A right side
    // that already holds thr-t mismatches behind the corresponding cut can never complete a hit, whatever i.
    // Pure pruning: the first (tt, i, j) found is unchanged.
    int g_at[3] = {-1, -1, -1}, lastpos = -1;
    uint32_t nleft = 0;
    bool walked = false;
    {
        uint32_t any = 0;
#pragma unroll
        for (int w = 0; w < NWT; w++) any |= (uint32_t)(D0[w] != 0);
        if (!any) return false;  // no mismatch to put the gap at: every mmi1[i] is map_readlen
    }
    for (uint32_t tt = 1; tt <= cx.gap * 2; tt++) {
        uint32_t t = (tt + 1) / 2;
        int shift = (tt & 1) ? -(int)t : (int)t;
        int shift1 = shift < 0 ? shift : 0;
        if (thr < 1 + t) break;
        uint64_t D1[NWT];
        {
            const uint32_t bound = thr - t + (shift < 0 ? t : 0), rel = (uint32_t)((int)rel0 + shift);
            uint32_t both = 0;
            bool dead = false;
#pragma unroll
            for (int v = 0; v < NWT; v++) {
                D1[v] = 0;
                if (!dead) {
                    D1[v] = mismatch_word_regs<NWT, NEWRULE>(W, rel, q, rc.end_element, rc.end_offset, v);
                    both += popc64(D0[v] & D1[v]);
                    dead = both >= bound;
                }
            }
            if (dead) continue;
        }
        if (!walked) {
            walked = true;
#pragma unroll
            for (int w = 0; w < NWT; w++) {
                uint64_t bits = D0[w];
                while (bits && nleft < thr - 1) {
                    int b = __clzll((long long)bits) >> 1;
                    bits &= ~(1ULL << (62 - 2 * b));
                    lastpos = w * 32 + b;
                    nleft++;
                    if (nleft == thr - 1) g_at[0] = lastpos;
                    if (nleft + 1 == thr - 1) g_at[1] = lastpos;
                    if (nleft + 2 == thr - 1) g_at[2] = lastpos;
                }
            }
        }
        int rl = len - (int)t - 1;
        {
            int G = nleft >= thr - t ? g_at[t - 1] : lastpos;  // >= every gap position the i-loop below can take
            if (G > rl - 1) G = rl - 1;
            int pcut_max = G - shift1;
            if (pcut_max > len - (int)cx.gap_edge) pcut_max = len - (int)cx.gap_edge;
            uint32_t J = 0;
#pragma unroll
            for (int v = 0; v < NWT; v++) J += popc64(D1[v] & ~prefix_pairs(pcut_max - 32 * v));
            if (J >= thr - t) continue;
        }
        uint32_t i = 0;  // index of the current left-side mismatch (mmi1[i])
#pragma unroll
        for (int w = 0; w < NWT; w++) {
            uint64_t bits = D0[w];
            while (bits && i < thr - t) {
                int b = __clzll((long long)bits) >> 1;
                bits &= ~(1ULL << (62 - 2 * b));
                int gp = w * 32 + b;
                if (gp >= (int)cx.gap_edge && gp < rl) {
                    int X = len + shift1 - gp;
                    if (X < (int)cx.gap_edge) X = (int)cx.gap_edge;
                    int pcut = len - X;  // right-side mismatches at positions >= pcut are at distance < X
                    uint32_t jstar = 0;
                    int plast = -1;      // highest mismatch position < pcut
#pragma unroll
                    for (int v = 0; v < NWT; v++) {
                        uint64_t pre = prefix_pairs(pcut - 32 * v);
                        jstar += popc64(D1[v] & ~pre);
                        uint64_t lo = D1[v] & pre;
                        if (lo) plast = v * 32 + 31 - (__ffsll((unsigned long long)lo) - 1) / 2;
                    }
                    if (plast >= 0 && jstar < thr - t - i) {
                        int m2 = len - 1 - plast;
                        if (m2 < rl) {  // m2 >= X >= gap_edge by construction
                            gap_snp = i + jstar + t;
                            int clip = gp + (int)cx.gap_edge - len - shift1;
                            if (clip > 0) gp -= clip;
                            gap_pos_out = (uint32_t)gp;
                            shift_out = shift;
                            return true;
                        }
                    }
                }
                i++;
            }
        }
    }
    return false;
}

// ---- the sequential hit state of one read (wave-uniform) ----------------------------------
// The hit log keeps every stored hit in insertion order. Its first 64 records live in registers, record i in
// lane i (the common read stores one or a few hits: no memory round trip to store, de-duplicate or pick one);
// records 64.. go to the wave's log in global memory.
struct HitState {
    uint32_t thr;     // snp_thres
    uint32_t nlog;    // records in the log
    uint32_t d0, d1, d2, d3;  // per lane: the four words of log record `lane`
    uint32_t bloom;           // per lane: 32 bits of a 2048-bit Bloom filter over the keys of records 64.. (the ones in memory)
    uint32_t g;               // per lane (PE kernels): the GLOBAL coordinate record `lane` was found at -- a mate of a pair runs every mode, and every mode finds
                              // the read's loci again: a candidate at a coordinate the log already holds is dropped before it is scored (process_read)
};

// the two filter positions of a key (wave-uniform)
__device__ __forceinline__ uint32_t bloom_hash(uint64_t key) {
    uint32_t x = (uint32_t)key ^ ((uint32_t)(key >> 32) * 0x9E3779B1u);
    x *= 0x85EBCA6Bu;
    return x ^ (x >> 15);
}

__device__ __forceinline__ uint64_t hit_key(uint32_t contig, uint32_t loc, bool gapped) {
    return ((uint64_t)contig << 33) | ((uint64_t)gapped << 32) | loc;
}

union HitWords {
    basal_hit h;
    uint32_t w[4];
};
// log record i of the read (all lanes get it); i < nlog
__device__ __forceinline__ basal_hit log_record(const HitState &st, const basal_hit *log, uint32_t i) {
    HitWords u;
    if (i < 64) { u.w[0] = rdlane(st.d0, (int)i); u.w[1] = rdlane(st.d1, (int)i); u.w[2] = rdlane(st.d2, (int)i); u.w[3] = rdlane(st.d3, (int)i); }
    else u.h = log[i];
    return u.h;
}
// this lane's record of the 64-record block starting at `base` (base + lane < nlog)
__device__ __forceinline__ basal_hit log_lane_record(const HitState &st, const basal_hit *log, uint32_t base, int lane) {
    HitWords u;
    if (base == 0) { u.w[0] = st.d0; u.w[1] = st.d1; u.w[2] = st.d2; u.w[3] = st.d3; }
    else u.h = log[base + lane];
    return u.h;
}

// int2hit + AddHit (align.cpp:319-346, align.h:329-347). All arguments wave-uniform.
// returns 1 when SnpAlign must stop (level-0 cap), else 0; may lower st.thr.
template <class LDS>
__device__ uint32_t add_hit(const DevCtx &cx, LDS &L, HitState &st, basal_hit *log, const ReadCtx &rc, uint32_t loc, uint32_t strand,
                            uint32_t chain, uint32_t w, uint32_t mode, int gap_size, uint32_t gap_pos, int lane) {
    // int2hit's binary search over ref_anchor (align.cpp:325-329), 64 probes per step: the largest
    // contig index whose anchor is <= loc (0 if none) -- one memory round trip for up to 64 contigs
    uint32_t left = 0, right = COLD(ncontig);
    const bool cached = right <= 64;
    if (cached) {
        uint32_t k = (uint32_t)__popcll(ballot((uint32_t)lane < right && s_anchor[lane] <= loc));
        left = k ? k - 1 : 0;
    } else
        while (right - left > 1) {
            uint32_t span = right - left, stride = (span + 63) / 64;
            uint32_t idx = left + (uint32_t)lane * stride;
            bool le = idx < right && COLDP(const uint32_t, ref_anchor)[idx] <= loc;
            uint32_t k = (uint32_t)__popcll(ballot(le));  // probes are ascending, so the true ones form a prefix
            if (k == 0) { right = left + 1; break; }
            left = left + (k - 1) * stride;
            right = left + stride < right ? left + stride : right;
        }
    uint32_t chr = (left * 2 + strand) & 0x3FFFF;
    // gHit.chr is an 18-bit field (param.h:35-42): beyond 131 071 contigs it wraps, and int2hit / AddHit read title[gh.chr] -- the
    // WRAPPED contig's rc_offset and size -- while the anchor subtracted is the true contig's (align.cpp:331-340, align.h:330-331)
    const uint32_t wc = chr >> 1;
    uint32_t anchor, rcoff = 0, csize;  // separate branches: a select between an LDS and a global pointer would make flat loads
    if (cached) { anchor = s_anchor[left]; csize = s_csize[left]; if (strand) rcoff = s_rcoff[left]; }
    else { anchor = COLDP(const uint32_t, ref_anchor)[left]; csize = COLDP(const uint32_t, contig_size)[wc]; if (strand) rcoff = COLDP(const uint32_t, rc_offset)[wc]; }
    uint32_t l = loc - anchor;
    uint32_t gp = gap_pos & 0x1FF;
    if (strand) {
        l = rcoff - rc.len - l;
        gp = (uint32_t)((int)rc.len + (gap_size < 0 ? gap_size : 0) - (int)gp) & 0x1FF;
        l -= (uint32_t)gap_size;
    }
#ifdef BASAL_COUNT_ADDHIT  // diagnostic build, -DBASAL_PHASE_TIMING -DBASAL_COUNT_ADDHIT: what AddHit's calls end in (printed by basal_core_sync_check; the atomics distort the clocks)
#define AH_COUNT(k, v) do { if (lane0(lane)) atomicAdd((unsigned long long *)(COLDP(unsigned int, guard) + 31) + PH_N + 4 + 32 + PH_N + (k), (unsigned long long)(v)); } while (0)
#else
#define AH_COUNT(k, v) do { } while (0)
#endif
    AH_COUNT(0, 1);
    AH_COUNT(5, st.nlog);
    if ((int)l < 0) { AH_COUNT(1, 1); return 0; }
    if (l + rc.len > csize) { AH_COUNT(1, 1); return 0; }
    uint64_t key = hit_key(chr >> 1, l, gap_size != 0);
    const uint64_t in_regs = st.nlog >= 64 ? ~0ULL : (1ULL << st.nlog) - 1;
    if (ballot(hit_key(st.d1 >> 1, st.d0, (st.d2 & 0xffu) != 0) == key) & in_regs) { AH_COUNT(2, 1); return 0; }
    // Long logs (reads from repeat families: hundreds of hits, nearly all of them new): a key the filter has not seen is in no
    // memory-resident record, and the scan -- a round trip per 256 records -- is skipped.
    const uint32_t bh = bloom_hash(key), b1 = bh & 2047u, b2 = (bh >> 11) & 2047u;
    const bool maybe_known = st.nlog > 64 && ((rdlane(st.bloom, (int)(b1 >> 5)) >> (b1 & 31)) & (rdlane(st.bloom, (int)(b2 >> 5)) >> (b2 & 31)) & 1u);
    if (maybe_known)
    for (uint32_t base = 64; base < st.nlog; base += 256) {  // scan the part in memory, four loads in flight per round trip
        bool d = false;
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t idx = base + 64 * u + (uint32_t)lane;
            const basal_hit h = log[idx < st.nlog ? idx : 0];  // (record 0 of the memory log is never written: its slot belongs to the registers)
            d |= idx < st.nlog && hit_key(h.chr >> 1, h.loc, h.gap_size != 0) == key;
        }
        AH_COUNT(6, 1);
        if (ballot(d)) { AH_COUNT(3, 1); return 0; }
    }
    AH_COUNT(4, 1);
    uint32_t n = st.nlog;
    if (n < COLD(scratch_per_wave)) {
        HitWords u;
        u.h.loc = l; u.h.chr = chr; u.h.gap_size = (int8_t)gap_size; u.h.strand = (uint8_t)(((strand << 1) | chain) & 3);
        u.h.gap_pos = (uint16_t)gp; u.h.level = (uint8_t)w; u.h.chain = (uint8_t)chain; u.h.mode = (uint8_t)mode; u.h.pad = 0;
        if (n < 64) {
            if ((uint32_t)lane == n) { st.d0 = u.w[0]; st.d1 = u.w[1]; st.d2 = u.w[2]; st.d3 = u.w[3]; if constexpr (LDS::PEK) st.g = loc; }
        } else {
            if (lane0(lane)) log[n] = u.h;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the other lanes read the log back later
            if ((uint32_t)lane == (b1 >> 5)) st.bloom |= 1u << (b1 & 31);
            if ((uint32_t)lane == (b2 >> 5)) st.bloom |= 1u << (b2 & 31);
        }
        st.nlog = n + 1;
    }
    uint32_t tot;
    {
        uint32_t a = L.nhit[chain][w] + 1u;
        if (lane0(lane)) L.nhit[chain][w] = a;
        tot = rfl(a + L.nhit[chain ^ 1][w]);  // the same in every lane; says so to the compiler (st.thr and the caller's loop exits stay scalar)
    }
    wave_sync();
    if (tot >= COLD(max_num_hits)) {
        if (w == 0) return 1;
        st.thr = w - 1;
    }
    return 0;
}

// k-th (0-based) log record with the given level and chain, in insertion order
__device__ uint32_t find_kth(const HitState &st, const basal_hit *log, uint32_t level, uint32_t chain, uint32_t k, int lane) {
    for (uint32_t base = 0; base < st.nlog; base += 64) {
        bool m = false;
        if (base + lane < st.nlog) {
            const basal_hit h = log_lane_record(st, log, base, lane);
            m = h.level == level && h.chain == chain;
        }
        uint64_t b = ballot(m);
        uint32_t c = (uint32_t)__popcll(b);
        if (k < c) {
            for (uint32_t i = 0; i < k; i++) b &= b - 1;
            return base + (uint32_t)__ffsll((unsigned long long)b) - 1;
        }
        k -= c;
    }
    return 0xffffffffu;
}

// The loads of one 64-candidate chunk of a mode's stream (non-GAP kernels), issued one chunk ahead: which list each
// candidate belongs to, its position in the list, its location and the flank word the filter compares.
struct ChunkLoads {
    uint32_t ei, jj, loc_raw;
    uint64_t f, fb, f2;  // non-GAP: f = the flank word on the entry's longer side; GAP: plane words after / before the seed, and the next 32 bases on the longer side
};
template <bool BOTH, class LDS>
__device__ __forceinline__ ChunkLoads issue_chunk(const DevCtx &cx, const LDS &L, uint32_t inc, uint64_t end_mask, uint32_t tb, uint32_t T, int lane,
                                                  uint32_t nlocs_u, unsigned long long flank_b_off, uint32_t r) {
    ChunkLoads c = {0, 0, 0, 0, 0, 0};
    // which seed's list candidate t belongs to = the number of list ends (inc[e], e < nent-1) that are <= t. The chunk is
    // 64 consecutive t, so that is the count at tb plus the ends inside the chunk (one or two, typically) -- cheaper than
    // comparing every lane against every end. (A ballot of one compare is one v_cmp; of a conjunction it is
    // v_cndmask + v_cmp on top: AND the masks instead.)
    uint32_t ei = (uint32_t)__popcll(ballot(inc <= tb) & end_mask);
    for (uint64_t inside = ballot(inc - tb - 1 < 63u) & end_mask; inside; inside &= inside - 1)  // tb < inc < tb + 64
        ei += ((uint32_t)lane >= rdlane(inc, __ffsll((unsigned long long)inside) - 1) - tb);
    const uint32_t t = tb + (uint32_t)lane;
    if (t < T) {
        const uint32_t e_off = L.ent[ei].off, e_m = L.ent[ei].m, e_jj0 = L.ent[ei].jj0, e_pre = L.ent[ei].pre, e_hcs = L.ent[ei].hcs;
        uint32_t jj = e_jj0 + (t - e_pre);
        if (jj >= e_m) jj -= e_m;
        const uint32_t x = e_off + jj;  // inside locs[] (the mode's set-up checked off + m per seed); kmer_off is 32-bit, so list positions are too
        // (the stream is read once: non-temporal loads in the non-GAP kernels, +1 % there; the GAP kernels lost 1.5 % with them)
        if (BOTH) { c.loc_raw = cx.locs[x]; c.f = cx.flank_a[x]; c.fb = cx.flank_a[flank_b_off + x]; c.f2 = cx.flank_a[(2ULL + ((e_hcs >> 17) & 1u)) * flank_b_off + x]; }
        else { c.loc_raw = __builtin_nontemporal_load(&cx.locs[x]); c.f = __builtin_nontemporal_load(&cx.flank_a[(unsigned long long)x + (((e_hcs >> 17) & 1u) ? flank_b_off : 0ULL)]); }
        c.ei = ei;
        c.jj = jj;
    }
    return c;
}


// ---- HEAVY kernels: bulk hit bookkeeping and the streaming of long lists ----------------------------------------------------------------
// Is `key` among the log records kept in memory (64 .. nlog-1)? Wave-uniform; four loads in flight per round trip.
__device__ bool log_has_key(const HitState &st, const basal_hit *log, uint64_t key, int lane) {
    for (uint32_t base = 64; base < st.nlog; base += 256) {
        bool d = false;
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t idx = base + 64 * u + (uint32_t)lane;
            const basal_hit h = log[idx < st.nlog ? idx : 0];  // (record 0 of the memory log is never written: its slot belongs to the registers)
            d |= idx < st.nlog && hit_key(h.chr >> 1, h.loc, h.gap_size != 0) == key;
        }
        if (ballot(d)) return true;
    }
    return false;
}

// int2hit + AddHit (align.cpp:319-346, align.h:329-347) for up to 64 scored candidates AT ONCE, lane order = visitation order.
// The sequential state machine is only order-sensitive at duplicates and where a level reaches the -w cap, so:
//   * every lane places its own hit (contig search, strand flip, bounds) and forms its key -- none of that depends on the state;
//   * of the lanes whose count is within the threshold, the ones whose key is already stored (the 64 records in registers by one compare
//     + ballot per lane, the memory part behind the Bloom filter) or equals the key of an earlier such lane are duplicates;
//   * per level present, the running total reaches the cap at the lane that holds the (cap - total)-th new hit of that level; the first
//     such lane over all levels ends the segment: everything up to and including it is appended with one permute / one store, the level
//     counts grow by popcounts, and the threshold drops to that level - 1 (level 0: SnpAlign stops);
//   * the lanes after it are looked at again under the new threshold (acceptance is monotone in it).
// Returns true when SnpAlign must stop.
template <class LDS>
__device__ bool bulk_add(const DevCtx &cx, LDS &L, HitState &st, basal_hit *log, const ReadCtx &rc, bool active, uint32_t loc, uint32_t strand, uint32_t chain,
                         uint32_t mm, uint32_t mode, uint32_t r, int lane) {
    uint64_t pend = ballot(active && mm <= st.thr);
    if (!pend) return false;
    const uint64_t lt = (1ULL << lane) - 1;
    // int2hit's search of ref_anchor (align.cpp:325-329): the largest contig whose anchor is <= loc, 0 if none. s_anchor holds the whole
    // table (<= 64 contigs) or 64 evenly spaced pivots of it (the search then ends in the stretch of the table between two pivots).
    const uint32_t nc = COLD(ncontig);
    const uint32_t pstride = (nc + 63) / 64;  // 1 for <= 64 contigs
    uint32_t left;
    {
        uint32_t lo = 0, hi = (nc + pstride - 1) / pstride;
#pragma unroll 1
        for (int it = 0; it < 6; it++) {
            const uint32_t mid = (lo + hi) >> 1;
            const bool go = hi - lo > 1, ge = s_anchor[mid & 63] <= loc;
            lo = go && ge ? mid : lo;
            hi = go && !ge ? mid : hi;
        }
        left = lo;
        if (nc > 64) {
            lo = left * pstride;
            hi = lo + pstride < nc ? lo + pstride : nc;
            while (ballot(hi - lo > 1)) {
                const uint32_t mid = (lo + hi) >> 1;
                const bool go = hi - lo > 1, ge = COLDP(const uint32_t, ref_anchor)[mid < nc ? mid : 0] <= loc;
                lo = go && ge ? mid : lo;
                hi = go && !ge ? mid : hi;
            }
            left = lo;
        }
    }
    const uint32_t chr = (left * 2 + strand) & 0x3FFFF;
    const uint32_t wc = chr >> 1;  // the 18-bit wrap of gHit.chr: title[] is read at the WRAPPED contig, the anchor at the true one (see add_hit)
    uint32_t anchor, csize, rcoff;
    // (each branch pins its loads with an empty asm: sunk into the `if (strand)` below, the choice between an LDS and a global pointer would be a flat load)
    if (nc <= 64) { anchor = s_anchor[left & 63]; csize = s_csize[left & 63]; rcoff = s_rcoff[left & 63]; asm volatile("" : "+v"(rcoff), "+v"(csize), "+v"(anchor)); }
    else {
        const uint32_t wci = wc < nc ? wc : 0;
        anchor = COLDP(const uint32_t, ref_anchor)[left]; csize = COLDP(const uint32_t, contig_size)[wci]; rcoff = COLDP(const uint32_t, rc_offset)[wci];
        asm volatile("" : "+v"(rcoff), "+v"(csize), "+v"(anchor));
    }
    uint32_t l = loc - anchor;
    if (strand) l = rcoff - rc.len - l;
    pend &= ballot((int)l >= 0 && l + rc.len <= csize);  // AddHit's two bounds (align.h:330-331)
    const uint64_t key = hit_key(chr >> 1, l, false);
    constexpr uint32_t kBloomBits = (uint32_t)(sizeof(L.bloom) * 8);
    const uint32_t bh = bloom_hash(key), b1 = bh & (kBloomBits - 1u), b2 = (bh >> 13) & (kBloomBits - 1u), bk = bh >> 27;
#ifdef BLOOM_K3
    const uint32_t b3 = ((bh * 0x9E3779B1u) >> 18) & (kBloomBits - 1u);
#endif
    HitWords u;
    u.h.loc = l; u.h.chr = chr; u.h.gap_size = 0; u.h.strand = (uint8_t)(((strand << 1) | chain) & 3);
    u.h.gap_pos = (uint16_t)(strand ? rc.len & 0x1FFu : 0u);  // int2hit mirrors gap_pos on the reverse strand even without a gap (align.cpp:341)
    u.h.level = (uint8_t)mm; u.h.chain = (uint8_t)chain; u.h.mode = (uint8_t)mode; u.h.pad = 0;
    const uint32_t cap = COLD(max_num_hits), spw = COLD(scratch_per_wave);
    for (uint32_t spin = 0;; spin++) {
        if (spin > 64) { guard_idx(cx, G_WATCHDOG, 0x40000u | spin, 0, r); return true; }
        const uint64_t acc = pend & ballot(mm <= st.thr);
        if (!acc) return false;
        uint64_t dup = 0;
        // Duplicates among the lanes themselves, in rounds: every lane still open aims at the bucket of its key's hash, the lowest lane of
        // a bucket is the first of its key; the others compare their key with that lane's -- equal: a duplicate of an earlier lane; not
        // equal (two keys in one bucket): open for the next round. Equal keys share a bucket, so nothing is missed.
        if (acc & (acc - 1)) {
            for (uint64_t open = acc; open;) {
                if (lane < 32) L.bucket[lane] = 0xFFFFFFFFu;
                wave_sync();
                const bool in = (open >> lane) & 1;
                if (in) atomicMin(&L.bucket[bk], (uint32_t)lane);
                wave_sync();
                const uint32_t w = in ? L.bucket[bk] & 63u : (uint32_t)lane;
                const uint32_t klo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(w << 2), (int)(uint32_t)key), khi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(w << 2), (int)(uint32_t)(key >> 32));
                const bool same = klo == (uint32_t)key && khi == (uint32_t)(key >> 32);
                const uint64_t first_m = ballot(w == (uint32_t)lane) & open, dup_m = ballot(w != (uint32_t)lane && same) & open;
                dup |= dup_m;
                open &= ~(first_m | dup_m);
            }
        }
        // Duplicates of stored hits: a key the Bloom filter (all stored keys of the read) has not seen is new; the few it may have seen
        // are looked up one by one -- the 64 records in registers with one compare + ballot, the rest of the log by scanning it.
        {
#ifdef BLOOM_K3
            const bool maybe = st.nlog && ((L.bloom[b1 >> 5] >> (b1 & 31)) & (L.bloom[b2 >> 5] >> (b2 & 31)) & (L.bloom[b3 >> 5] >> (b3 & 31)) & 1u);
#else
            const bool maybe = st.nlog && ((L.bloom[b1 >> 5] >> (b1 & 31)) & (L.bloom[b2 >> 5] >> (b2 & 31)) & 1u);
#endif
            uint64_t chk = ballot(maybe) & acc & ~dup;
            if (chk) {
                const uint64_t regk = hit_key(st.d1 >> 1, st.d0, (st.d2 & 0xffu) != 0);
                const uint64_t in_regs = st.nlog >= 64 ? ~0ULL : (1ULL << st.nlog) - 1;
                for (; chk; chk &= chk - 1) {
                    const int la = __ffsll((unsigned long long)chk) - 1;
                    const uint64_t kb = rdlane64(key, la);
                    bool known = (ballot(regk == kb) & in_regs) != 0;
                    if (!known && st.nlog > 64) known = log_has_key(st, log, kb, lane);
                    if (known) dup |= 1ULL << la;
                }
            }
        }
        const uint64_t newm = acc & ~dup;
        // the first lane at which a level's total reaches the cap
        uint32_t xlane = 64, xlevel = 0;
        {
            const uint32_t totv = lane < 16 ? L.nhit[0][lane] + L.nhit[1][lane] : 0;
            for (uint64_t rem = newm; rem;) {
                const uint32_t w = rdlane(mm, __ffsll((unsigned long long)rem) - 1);
                const uint64_t mw = ballot(mm == w) & newm;
                rem &= ~mw;
                const uint32_t tw = rdlane(totv, (int)(w & 15));
                const uint32_t room = tw < cap ? cap - tw : 1;
                if ((uint32_t)__popcll(mw) >= room) {
                    const uint64_t hm = ballot((uint32_t)__popcll(mw & lt) + 1 == room) & mw;
                    const uint32_t xl = (uint32_t)__ffsll((unsigned long long)hm) - 1;
                    if (xl < xlane) { xlane = xl; xlevel = w; }
                }
            }
        }
        const uint64_t upto = xlane < 63 ? (2ULL << xlane) - 1 : ~0ULL;
        const uint64_t segm = newm & upto;
        if (segm) {
            const bool mine = (segm >> lane) & 1;
            const uint32_t n0 = st.nlog, cnt = (uint32_t)__popcll(segm), pos = n0 + (uint32_t)__popcll(segm & lt);
            if (n0 < 64) {  // records 0..63 live in registers, record i in lane i: every new hit goes to the lane of its position
                // (lanes with nothing to send aim at a lane outside [n0, n0 + cnt): lane 0 if n0 > 0, else lane 63 -- when the range is all 64 lanes every lane sends)
                const uint32_t dst = (mine && pos < 64) ? pos : (n0 ? 0u : 63u);
                const uint32_t r0 = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)u.w[0]), r1 = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)u.w[1]),
                               r2 = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)u.w[2]), r3 = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)u.w[3]);
                if ((uint32_t)lane >= n0 && (uint32_t)lane < n0 + cnt) { st.d0 = r0; st.d1 = r1; st.d2 = r2; st.d3 = r3; }
            }
            const uint64_t inmem = ballot(mine && pos >= 64 && pos < spw);
            if (inmem) {
                if ((inmem >> lane) & 1) log[pos] = u.h;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the other lanes read the log back later
            }
            if (mine) {
                atomicOr(&L.bloom[b1 >> 5], 1u << (b1 & 31));
                atomicOr(&L.bloom[b2 >> 5], 1u << (b2 & 31));
#ifdef BLOOM_K3
                atomicOr(&L.bloom[b3 >> 5], 1u << (b3 & 31));
#endif
                atomicAdd(&L.nhit[chain][mm & 15], 1u);
            }
            st.nlog = n0 + cnt < spw ? n0 + cnt : spw;
            wave_sync();
        }
        if (xlane >= 64) return false;
        if (xlevel == 0) return true;
        st.thr = xlevel - 1;
        pend &= ~upto;
    }
}

// ---- HEAVY GAP kernels: the survivors' stage on bit planes -------------------------------------------------------------------------------
// Where long lists are the rule (an index with a high over-represented-k-mer cut-off: a repeat-rich genome) the GAP kernels meet hundreds of
// near-copies per read that no flank bound can drop; what they cost is the gap search itself and the one-by-one AddHit. These kernels
// (align_kernel<*, *, true, true>) keep the stream filter of the GAP kernels and replace the stage behind it (gap_flush):
//   * the survivor's reference bases come from a second copy of the reference kept as BIT PLANES (DevCtx::xpl), so its mismatch bitmap at the
//     candidate's start and at every shifted start is a funnel shift and three bit-selects per 64 bases, one bit per read base, LSB first;
//   * GapAlign (align.cpp:348-410) runs on those bitmaps (gap_search): the same decisions as gap_align above, at a third of the instructions;
//   * the ungapped hit and the gapped hit of up to 64 candidates are booked at once (bulk_add2), in the order the reference books them --
//     candidate by candidate, the ungapped hit first (align.cpp:308-312).

__device__ __forceinline__ uint64_t low_mask64(int n) { return n <= 0 ? 0ULL : n >= 64 ? ~0ULL : (1ULL << n) - 1; }
// set bits of an NW-word bitmap (bit p of word p / 64 = read position p) at positions < n / >= n
template <int NW>
__device__ __forceinline__ uint32_t popc_below(const uint64_t *D, int n) {
    uint32_t c = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) c += popc64(D[w] & low_mask64(n - 64 * w));
    return c;
}
template <int NW>
__device__ __forceinline__ uint32_t popc_from(const uint64_t *D, int n) {
    uint32_t c = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) c += popc64(D[w] & ~low_mask64(n - 64 * w));
    return c;
}
// bits [sh, sh + 64) of the 128-bit value hi:lo, sh < 64
__device__ __forceinline__ uint64_t funnel64(uint64_t lo, uint64_t hi, uint32_t sh) { return (lo >> sh) | ((hi << 1) << (63 - sh)); }

// The read's mismatch bitmap against the reference shifted by `sh` bits: Rh / Rl = the reference's bit planes from base (candidate start - g) on,
// M = L.mmp[chain] ("this read base mismatches letter X", one zero word in front; zero past the read): the word loop of MismatchPattern0/1
// (align.h:140-165, 179-193) for both rules -- the comparison is per base, so the masks are exact for every -M rule.
template <int NW>
__device__ __forceinline__ void plane_bitmap(const uint64_t Rh[NW + 1], const uint64_t Rl[NW + 1], uint32_t sh, const uint64_t (*M)[NW + 2], uint64_t D[NW]) {
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const uint64_t m[4] = {M[0][w + 1], M[1][w + 1], M[2][w + 1], M[3][w + 1]};
        D[w] = plane_mismatch(funnel64(Rh[w], Rh[w + 1], sh), funnel64(Rl[w], Rl[w + 1], sh), m);
    }
}

// position of the k-th set bit of x counted from the top (k >= 1; x holds at least k bits): a binary search on popcounts, no loop
__device__ __forceinline__ int select_from_top64(uint64_t x, uint32_t k) {
    uint32_t v = (uint32_t)(x >> 32), c = (uint32_t)__popc(v);
    int base = 32;
    if (k > c) { v = (uint32_t)x; base = 0; k -= c; }
#pragma unroll
    for (int h = 16; h >= 1; h >>= 1) {  // the upper half of the h*2 bits still in play holds c bits: the k-th is there, or it is the (k - c)-th of the lower half
        const uint32_t up = v >> h;
        c = (uint32_t)__popc(up);
        const bool in_up = k <= c;
        v = in_up ? up : v & ((1u << h) - 1u);
        base += in_up ? h : 0;
        k -= in_up ? 0u : c;
    }
    return base;
}
// the same over an NW-word bitmap; -1 if it holds fewer than k bits
template <int NW>
__device__ __forceinline__ int kth_from_top(const uint64_t *D, uint32_t k) {
    uint64_t x = 0;
    int wbase = -1;
#pragma unroll
    for (int w = NW - 1; w >= 0; w--) {
        const uint32_t c = popc64(D[w]);
        const bool here = wbase < 0 && k <= c;
        if (here) { x = D[w]; wbase = 64 * w; }
        if (wbase < 0) k -= c;
    }
    return wbase < 0 ? -1 : wbase + select_from_top64(x, k);
}

// GapAlign (align.cpp:348-410) for one candidate on one-bit-per-base bitmaps: D0 = the ungapped bitmap (no N mask, as MismatchPattern0), the
// shifted bitmaps are cut from the planes Rh / Rl (bit k = reference base start - g + k). The reference walks, per shift, the left-side mismatches
// mmi1[i] in order and, for each, the right-side mismatches mmi2[j] (align.cpp:383-404); a left-side mismatch at gp can only complete a hit if
// fewer than thr - t right-side mismatches lie at or behind pcut = min(gp - shift1, len - gap_edge) -- so with p* = one past the (thr - t)-th
// right-side mismatch from the read's end (one popcount search, no loop) only the left-side mismatches in [p* + shift1, len - t - 1) are looked
// at, usually none: the same first (tt, i, j) as the reference's loops (and as gap_align above), without walking the ones that cannot win.
template <int NW>
__device__ bool gap_search(const DevCtx &cx, const uint64_t Rh[NW + 1], const uint64_t Rl[NW + 1], const uint64_t (*M)[NW + 2], const uint64_t D0[NW], const ReadCtx &rc,
                           uint32_t thr, uint32_t seed_pos, uint32_t &gap_snp, uint32_t &gap_pos_out, int &shift_out) {
    if (thr < 2) return false;
    const int len = (int)rc.len, ge = (int)cx.gap_edge;
    // MismatchPattern0's return value against seed_pos + seed_size (align.cpp:365): thr-1 mismatches before the end of the seed end the search
    if (popc_below<NW>(D0, (int)(seed_pos + cx.K)) >= thr - 1) return false;
    for (uint32_t tt = 1; tt <= cx.gap * 2; tt++) {
        const uint32_t t = (tt + 1) / 2;
        const int shift = (tt & 1) ? -(int)t : (int)t, shift1 = shift < 0 ? shift : 0;
        if (thr < 1 + t) break;
        uint64_t D1[NW];
        plane_bitmap<NW>(Rh, Rl, (uint32_t)((int)cx.gap + shift), M, D1);
        {   // a base that mismatches at both starts costs one mismatch wherever the gap falls (the t inserted bases excepted)
            uint32_t both = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) both += popc64(D0[w] & D1[w]);
            if (both >= thr - t + (shift < 0 ? t : 0)) continue;
        }
        const int pstar = kth_from_top<NW>(D1, thr - t) + 1;  // popc(D1 at positions >= p) < thr - t  <=>  p >= pstar
        if (pstar > len - ge) continue;
        const int rl = len - (int)t - 1;
        int lo = pstar + shift1;
        if (lo < ge) lo = ge;
        uint64_t cand[NW], any = 0;  // the left-side mismatches worth a look: gap_edge <= gp < rl (align.cpp:385) and pcut >= p*
#pragma unroll
        for (int w = 0; w < NW; w++) { cand[w] = D0[w] & ~low_mask64(lo - 64 * w) & low_mask64(rl - 64 * w); any |= cand[w]; }
        if (!any) continue;
        uint32_t i = popc_below<NW>(D0, lo);  // index of the current left-side mismatch (mmi1[i])
#pragma unroll
        for (int w = 0; w < NW; w++) {
            uint64_t bits = cand[w];
            while (bits && i < thr - t) {
                int gp = w * 64 + __ffsll((unsigned long long)bits) - 1;
                bits &= bits - 1;
                int X = len + shift1 - gp;
                if (X < ge) X = ge;
                const int pcut = len - X;  // right-side mismatches at positions >= pcut are at distance < X from the read's end
                uint32_t jstar = 0;
                int plast = -1;            // the highest mismatch position < pcut
#pragma unroll
                for (int v = 0; v < NW; v++) {
                    const uint64_t pre = low_mask64(pcut - 64 * v);
                    jstar += popc64(D1[v] & ~pre);
                    const uint64_t lw = D1[v] & pre;
                    if (lw) plast = v * 64 + 63 - __clzll((long long)lw);
                }
                if (plast >= 0 && jstar < thr - t - i) {
                    const int m2 = len - 1 - plast;
                    if (m2 < rl) {  // m2 >= X >= gap_edge by construction
                        gap_snp = i + jstar + t;
                        const int clip = gp + ge - len - shift1;
                        if (clip > 0) gp -= clip;
                        gap_pos_out = (uint32_t)gp;
                        shift_out = shift;
                        return true;
                    }
                }
                i++;
            }
        }
    }
    return false;
}

// Which lanes of `acc` hold a key that an EARLIER lane of `acc` holds too. In rounds: every open lane aims at the bucket of its key's hash, the
// lowest lane of a bucket is the first of its key; the others compare their key with that lane's -- equal: a duplicate; not equal (two keys in
// one bucket): open for the next round. Equal keys share a bucket, so nothing is missed. (bulk_add's rounds, as a function.)
template <class LDS>
__device__ uint64_t dups_among(LDS &L, uint64_t acc, uint64_t key, uint32_t bk, int lane) {
    uint64_t dup = 0;
    if (!(acc & (acc - 1))) return 0;
    for (uint64_t open = acc; open;) {
        if (lane < 32) L.bucket[lane] = 0xFFFFFFFFu;
        wave_sync();
        const bool in = (open >> lane) & 1;
        if (in) atomicMin(&L.bucket[bk], (uint32_t)lane);
        wave_sync();
        const uint32_t w = in ? L.bucket[bk] & 63u : (uint32_t)lane;
        const uint32_t klo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(w << 2), (int)(uint32_t)key), khi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(w << 2), (int)(uint32_t)(key >> 32));
        const bool same = klo == (uint32_t)key && khi == (uint32_t)(key >> 32);
        const uint64_t first_m = ballot(w == (uint32_t)lane) & open, dup_m = ballot(w != (uint32_t)lane && same) & open;
        dup |= dup_m;
        open &= ~(first_m | dup_m);
    }
    return dup;
}
// Which lanes of `cand` hold a key that is already stored: the Bloom filter over all stored keys of the read says "certainly not" for most; the
// rest are looked up one by one (the 64 records in registers by compare + ballot, the memory part of the log by scanning it).
template <class LDS>
__device__ uint64_t dups_stored(LDS &L, const HitState &st, const basal_hit *log, uint64_t cand, uint64_t key, uint32_t b1, uint32_t b2, int lane) {
    const bool maybe = st.nlog && ((L.bloom[b1 >> 5] >> (b1 & 31)) & (L.bloom[b2 >> 5] >> (b2 & 31)) & 1u);
    uint64_t chk = ballot(maybe) & cand, dup = 0;
    if (chk) {
        const uint64_t regk = hit_key(st.d1 >> 1, st.d0, (st.d2 & 0xffu) != 0);
        const uint64_t in_regs = st.nlog >= 64 ? ~0ULL : (1ULL << st.nlog) - 1;
        for (; chk; chk &= chk - 1) {
            const int la = __ffsll((unsigned long long)chk) - 1;
            const uint64_t kb = rdlane64(key, la);
            bool known = (ballot(regk == kb) & in_regs) != 0;
            if (!known && st.nlog > 64) known = log_has_key(st, log, kb, lane);
            if (known) dup |= 1ULL << la;
        }
    }
    return dup;
}

// int2hit + AddHit (align.cpp:319-346, align.h:329-347) for up to 64 candidates AT ONCE, each with an ungapped hit (level mmU) and / or a gapped
// one (gfound: level gsnp, shift gshift, position gpos), lane order = visitation order, within a lane the ungapped hit first (SnpAlign books it
// before it calls GapAlign, align.cpp:308-312). bulk_add's scheme over 128 slots (2 * lane + kind): every lane places both hits and forms both
// keys (ungapped and gapped hits are de-duplicated apart, align.h:332-337, so the two kinds never meet); duplicates among the lanes and of
// stored hits per kind; per level present, the slot at which the running total reaches the -w cap; everything up to the first such slot is
// appended in slot order, the threshold drops to that level - 1 and the call returns: the slots after it must be looked at again under the new
// threshold, and the gap search -- whose result depends on the threshold -- run again for them (the caller does that).
// pendU / pendG: the lanes whose ungapped / gapped slot is still open (in and out). Returns BA_DONE (nothing open any more), BA_STOP (a level-0
// cap: SnpAlign must stop) or BA_RETHR (st.thr was lowered; pendU / pendG hold what is left).
enum { BA_DONE = 0, BA_STOP = 1, BA_RETHR = 2 };
template <class LDS>
__device__ int bulk_add2(const DevCtx &cx, LDS &L, HitState &st, basal_hit *log, const ReadCtx &rc, uint64_t &pendU, uint64_t &pendG, uint32_t loc, uint32_t strand,
                         uint32_t chain, uint32_t mmU, bool gfound, uint32_t gsnp, uint32_t gpos, int gshift, uint32_t mode, uint32_t r, int lane) {
    uint64_t accU = pendU & ballot(mmU <= st.thr), accG = pendG & ballot(gfound);
    if (!(accU | accG)) { pendU = pendG = 0; return BA_DONE; }
    const uint64_t lt = (1ULL << lane) - 1;
    // int2hit's search of ref_anchor (align.cpp:325-329), as in bulk_add: s_anchor holds the whole table (<= 64 contigs) or 64 pivots of it
    const uint32_t nc = COLD(ncontig);
    const uint32_t pstride = (nc + 63) / 64;
    uint32_t left;
    {
        uint32_t lo = 0, hi = (nc + pstride - 1) / pstride;
#pragma unroll 1
        for (int it = 0; it < 6; it++) {
            const uint32_t mid = (lo + hi) >> 1;
            const bool go = hi - lo > 1, ge = s_anchor[mid & 63] <= loc;
            lo = go && ge ? mid : lo;
            hi = go && !ge ? mid : hi;
        }
        left = lo;
        if (nc > 64) {
            lo = left * pstride;
            hi = lo + pstride < nc ? lo + pstride : nc;
            while (ballot(hi - lo > 1)) {
                const uint32_t mid = (lo + hi) >> 1;
                const bool go = hi - lo > 1, ge = COLDP(const uint32_t, ref_anchor)[mid < nc ? mid : 0] <= loc;
                lo = go && ge ? mid : lo;
                hi = go && !ge ? mid : hi;
            }
            left = lo;
        }
    }
    const uint32_t chr = (left * 2 + strand) & 0x3FFFF;
    const uint32_t wc = chr >> 1;  // the 18-bit wrap of gHit.chr (see add_hit)
    uint32_t anchor, csize, rcoff;
    if (nc <= 64) { anchor = s_anchor[left & 63]; csize = s_csize[left & 63]; rcoff = s_rcoff[left & 63]; asm volatile("" : "+v"(rcoff), "+v"(csize), "+v"(anchor)); }
    else {
        const uint32_t wci = wc < nc ? wc : 0;
        anchor = COLDP(const uint32_t, ref_anchor)[left]; csize = COLDP(const uint32_t, contig_size)[wci]; rcoff = COLDP(const uint32_t, rc_offset)[wci];
        asm volatile("" : "+v"(rcoff), "+v"(csize), "+v"(anchor));
    }
    // the two hits as int2hit leaves them (align.cpp:334-345)
    uint32_t lU = loc - anchor, lG = lU, gpU = 0, gpG = gpos & 0x1FFu;
    if (strand) {
        lU = rcoff - rc.len - lU;
        gpU = rc.len & 0x1FFu;
        gpG = (uint32_t)((int)rc.len + (gshift < 0 ? gshift : 0) - (int)gpG) & 0x1FFu;
        lG = lU - (uint32_t)gshift;
    }
    accU &= ballot((int)lU >= 0 && lU + rc.len <= csize);  // AddHit's two bounds (align.h:330-331)
    accG &= ballot((int)lG >= 0 && lG + rc.len <= csize);
    const uint64_t keyU = hit_key(chr >> 1, lU, false), keyG = hit_key(chr >> 1, lG, true);
    constexpr uint32_t kBloomBits = (uint32_t)(sizeof(L.bloom) * 8);
    const uint32_t bhU = bloom_hash(keyU), bhG = bloom_hash(keyG);
    const uint32_t u1 = bhU & (kBloomBits - 1u), u2 = (bhU >> 13) & (kBloomBits - 1u), g1 = bhG & (kBloomBits - 1u), g2 = (bhG >> 13) & (kBloomBits - 1u);
    HitWords hU, hG;
    hU.h.loc = lU; hU.h.chr = chr; hU.h.gap_size = 0; hU.h.strand = (uint8_t)(((strand << 1) | chain) & 3); hU.h.gap_pos = (uint16_t)gpU;
    hU.h.level = (uint8_t)mmU; hU.h.chain = (uint8_t)chain; hU.h.mode = (uint8_t)mode; hU.h.pad = 0;
    hG.h.loc = lG; hG.h.chr = chr; hG.h.gap_size = (int8_t)gshift; hG.h.strand = hU.h.strand; hG.h.gap_pos = (uint16_t)gpG;
    hG.h.level = (uint8_t)gsnp; hG.h.chain = (uint8_t)chain; hG.h.mode = (uint8_t)mode; hG.h.pad = 0;
    // duplicates: of an earlier lane, of a stored hit
    uint64_t dupU = dups_among(L, accU, keyU, bhU >> 27, lane), dupG = dups_among(L, accG, keyG, bhG >> 27, lane);
    dupU |= dups_stored(L, st, log, accU & ~dupU, keyU, u1, u2, lane);
    dupG |= dups_stored(L, st, log, accG & ~dupG, keyG, g1, g2, lane);
    const uint64_t newU = accU & ~dupU, newG = accG & ~dupG;
    // the first slot at which a level's total reaches the cap
    const uint32_t cap = COLD(max_num_hits), spw = COLD(scratch_per_wave);
    uint32_t xslot = 128, xlevel = 0;
    {
        const uint32_t totv = lane < 16 ? L.nhit[0][lane] + L.nhit[1][lane] : 0;
        for (uint64_t remU = newU, remG = newG; remU | remG;) {
            const uint32_t w = remU ? rdlane(mmU, __ffsll((unsigned long long)remU) - 1) : rdlane(gsnp, __ffsll((unsigned long long)remG) - 1);
            const uint64_t mwU = ballot(mmU == w) & newU, mwG = ballot(gsnp == w) & newG;
            remU &= ~mwU;
            remG &= ~mwG;
            const uint32_t tw = rdlane(totv, (int)(w & 15));
            const uint32_t room = tw < cap ? cap - tw : 1;
            if ((uint32_t)__popcll(mwU) + (uint32_t)__popcll(mwG) >= room) {
                const uint32_t pre = (uint32_t)__popcll(mwU & lt) + (uint32_t)__popcll(mwG & lt), inU = (uint32_t)(mwU >> lane) & 1u, inG = (uint32_t)(mwG >> lane) & 1u;
                const uint64_t hmU = ballot(inU && pre + 1 == room), hmG = ballot(inG && pre + inU + 1 == room);
                const uint32_t sU = hmU ? 2u * (uint32_t)(__ffsll((unsigned long long)hmU) - 1) : 128u, sG = hmG ? 2u * (uint32_t)(__ffsll((unsigned long long)hmG) - 1) + 1u : 128u;
                const uint32_t sx = sU < sG ? sU : sG;
                if (sx < xslot) { xslot = sx; xlevel = w; }
            }
        }
    }
    // everything up to and including that slot, in slot order
    const uint32_t xl = xslot >> 1;
    const uint64_t incl = xslot >= 128 ? ~0ULL : xl < 63 ? (2ULL << xl) - 1 : ~0ULL, excl = xslot >= 128 ? ~0ULL : (1ULL << xl) - 1;
    const uint64_t uptoU = incl, uptoG = (xslot & 1) || xslot >= 128 ? incl : excl;
    const uint64_t segU = newU & uptoU, segG = newG & uptoG;
    if (segU | segG) {
        const bool mineU = (segU >> lane) & 1, mineG = (segG >> lane) & 1;
        const uint32_t n0 = st.nlog, cnt = (uint32_t)__popcll(segU) + (uint32_t)__popcll(segG);
        const uint32_t posU = n0 + (uint32_t)__popcll(segU & lt) + (uint32_t)__popcll(segG & lt), posG = posU + (uint32_t)mineU;
        if (n0 < 64) {
            // records 0..63 live in registers, record i in lane i: they get there through the first 64 entries of the survivor list (the batch
            // being booked has been read into registers: 512 free bytes = 32 records at a time)
            uint4 *stage = (uint4 *)&L.surv[0];
#pragma unroll 1
            for (uint32_t p0 = n0; p0 < 64 && p0 < n0 + cnt; p0 += 32) {
                wave_sync();
                if (mineU && posU - p0 < 32u) stage[posU - p0] = make_uint4(hU.w[0], hU.w[1], hU.w[2], hU.w[3]);
                if (mineG && posG - p0 < 32u) stage[posG - p0] = make_uint4(hG.w[0], hG.w[1], hG.w[2], hG.w[3]);
                wave_sync();
                if ((uint32_t)lane - p0 < 32u && (uint32_t)lane < n0 + cnt) {
                    const uint4 v = stage[(uint32_t)lane - p0];
                    st.d0 = v.x; st.d1 = v.y; st.d2 = v.z; st.d3 = v.w;
                }
            }
            wave_sync();
        }
        const uint64_t inmem = ballot((mineU && posU >= 64 && posU < spw) || (mineG && posG >= 64 && posG < spw));
        if (inmem) {
            if (mineU && posU >= 64 && posU < spw) log[posU] = hU.h;
            if (mineG && posG >= 64 && posG < spw) log[posG] = hG.h;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the other lanes read the log back later
        }
        if (mineU) {
            atomicOr(&L.bloom[u1 >> 5], 1u << (u1 & 31));
            atomicOr(&L.bloom[u2 >> 5], 1u << (u2 & 31));
            atomicAdd(&L.nhit[chain][mmU & 15], 1u);
        }
        if (mineG) {
            atomicOr(&L.bloom[g1 >> 5], 1u << (g1 & 31));
            atomicOr(&L.bloom[g2 >> 5], 1u << (g2 & 31));
            atomicAdd(&L.nhit[chain][gsnp & 15], 1u);
        }
        st.nlog = n0 + cnt < spw ? n0 + cnt : spw;
        wave_sync();
    }
    if (xslot >= 128) { pendU = pendG = 0; return BA_DONE; }
    if (xlevel == 0) return BA_STOP;
    st.thr = xlevel - 1;
    pendU &= ~uptoU;
    pendG &= ~uptoG;
    return BA_RETHR;
}

// The HEAVY GAP kernels' survivor stage: the candidates in L.surv[0 .. min(nsurv, 64)) -- their reference planes in one round trip, the
// ungapped count (CountMismatch*: the bitmap under the valid mask + the N count), the gap search, both hits booked in bulk. Returns true when
// SnpAlign must stop.
template <int NWT, bool NEWRULE, class LDS>
__device__ bool gap_flush(const DevCtx &cx, LDS &L, HitState &st, basal_hit *log, const ReadCtx &rc, uint32_t mode, uint32_t &nsurv, uint32_t r, int lane PH_PARAM) {
    constexpr int NW = NWT / 2;
    const uint32_t batch = nsurv < 64 ? nsurv : 64;
    const bool active = (uint32_t)lane < batch;
    uint32_t loc = BASAL_REF_MARGIN * 32, strand = 0, hcs = 0, mm = 0xffff;
    bool alive = false, gap_ok = false;
#ifdef BASAL_CHECK_FILTER
    uint32_t chk = 3;
#endif
    if (active) {
        const SurvEnt sv = L.surv[lane];
        strand = (sv.meta >> 8) & 1;
        alive = (sv.meta >> 9) & 1;
        gap_ok = (sv.meta >> 10) & 1;
#ifdef BASAL_CHECK_FILTER
        chk = (sv.meta >> 11) & 3u;
#endif
        loc = sv.loc;
        if (((unsigned long long)(loc >> 5) + NWT + 4) >= COLD(nwords)) loc = (uint32_t)guard_idx(cx, G_XREF, loc, 0, r) + BASAL_REF_MARGIN * 32;
        hcs = L.ent[sv.meta & 0xff].hcs;
    }
    const uint32_t chain = (hcs >> 16) & 1;
    const uint64_t(*M)[NW + 2] = L.mmp[chain];
    // the candidate's reference bases from start - g on, as planes in the read's frame: NW + 2 blocks of 64 bases cover the read at every shift
    uint64_t Rh[NW + 1], Rl[NW + 1], D0[NW];
    {
        const uint32_t first = loc - cx.gap, rel = first & 63;  // loc >= 12320 by construction
        const ulonglong2 *P = (strand ? cx.xpl[1] : cx.xpl[0]) + (first >> 6);
        const uint32_t nblk = active ? ((rel + rc.len + 2 * cx.gap + 63) >> 6) : 0;
        ulonglong2 B[NW + 2];
#pragma unroll
        for (int i = 0; i < NW + 2; i++) {
            B[i] = make_ulonglong2(0, 0);
            if ((uint32_t)i < nblk) B[i] = P[i];
        }
#pragma unroll
        for (int w = 0; w <= NW; w++) { Rh[w] = funnel64(B[w].x, B[w + 1].x, rel); Rl[w] = funnel64(B[w].y, B[w + 1].y, rel); }
    }
    plane_bitmap<NW>(Rh, Rl, cx.gap, M, D0);
    if (alive) {
        mm = rc.n_count;
#pragma unroll
        for (int w = 0; w < NW; w++) mm += popc64(D0[w] & L.valp[chain][w + 1]);
    }
    PH(PH_SCORE);
#ifdef BASAL_CHECK_FILTER
    if (active && mm <= st.thr && !(chk & 1)) guard_idx(cx, G_WATCHDOG, 0x20000u | mm, 0, r);
#endif
    uint64_t pendU = ballot(active && alive), pendG = ballot(active && gap_ok);
    bool stop = false;
    for (uint32_t spin = 0;; spin++) {
        if (spin > 160) { guard_idx(cx, G_WATCHDOG, 0x50000u | spin, 0, r); stop = true; break; }
        bool gfound = false;
        uint32_t gsnp = 0, gpos = 0;
        int gshift = 0;
        if ((pendG >> lane) & 1) gfound = gap_search<NW>(cx, Rh, Rl, M, D0, rc, st.thr, hcs & 0xffffu, gsnp, gpos, gshift);
#ifdef BASAL_CHECK_FILTER
        if (gfound && !(chk & 2)) guard_idx(cx, G_WATCHDOG, 0x30000u | (gsnp << 8) | (uint32_t)(gshift & 0xff), 0, r);
#endif
        PH(PH_REPLAY);
        const int rcode = bulk_add2(cx, L, st, log, rc, pendU, pendG, loc, strand, chain, mm, gfound, gsnp, gpos, gshift, mode, r, lane);
        PH(PH_E1);
        if (rcode == BA_STOP) { stop = true; break; }
        if (rcode == BA_DONE) break;
    }
    const uint32_t rest = nsurv - batch;  // drop the batch from the front of the list
    SurvEnt v = {0, 0};
    if ((uint32_t)lane < rest) v = L.surv[batch + lane];
    wave_sync();
    if ((uint32_t)lane < rest) L.surv[lane] = v;
    nsurv = rest;
    wave_sync();
    return stop;
}

// A window of the HEAVY kernels' stream test. The flank words of a core that keeps long lists are stored as bit planes (basal_bits.h
// split_planes: the high bits of the 32 base codes in the upper half, the low bits in the lower), and what depends on the read alone is folded
// into four 32-bit words, so that cmp_word's rule (basal_bits.h; CountMismatch / CountMismatch_new, align.h:126-128, 210-219) costs a
// candidate four (old rule) or six (new rule) instructions per 32 bases and no shifts. With th / tl = the planes of s ^ r:
//   old rule: a mismatch unless s == r, or s == 01 and r == 11 (then th = 1, tl = 0):  (th & ~[r == 11]) | tl
//   new rule: a mismatch unless s == r, or s == 01 and the read base is convert-tolerant (high bit of the convert-to plane clear)
// rh, rl = the read's planes; m = valid read bases (the valid plane holds 00 or 11 per base: basal_core_create checks reg_alphabet for
// that); x = old rule: m & ~[r == 11] / new rule: the tolerant bases.
struct WinP { uint32_t rh, rl, x, m; };
template <bool NEWRULE>
__device__ __forceinline__ WinP win_make(uint64_t r, uint64_t valid, uint64_t c) {
    WinP k;
    k.rh = even_bits(r >> 1); k.rl = even_bits(r);
    k.m = even_bits(valid >> 1);
    k.x = NEWRULE ? even_bits(~c >> 1) : k.m & ~(k.rh & k.rl);
    return k;
}
template <bool NEWRULE>
__device__ __forceinline__ uint32_t win_count(const WinP &k, uint64_t s) {
    const uint32_t sh = (uint32_t)(s >> 32), sl = (uint32_t)s;
    if (!NEWRULE) return (uint32_t)__popc(((sh ^ k.rh) & k.x) | ((sl ^ k.rl) & k.m));
    return (uint32_t)__popc(((sh ^ k.rh) | (sl ^ k.rl)) & k.m & ~(~sh & sl & k.x));
}
// the same for the 16 bases of a seed (basal_bits.h split_planes16: high bits in bits 16-31, low bits in bits 0-15); r = both planes of the read
struct WinP16 { uint32_t r, x, m; };
template <bool NEWRULE>
__device__ __forceinline__ WinP16 win_make16(uint32_t r, uint32_t valid, uint32_t c) {
    WinP16 k;
    k.r = split_planes16(r);
    k.m = even_bits(valid >> 1);
    k.x = NEWRULE ? even_bits(~c >> 1) : k.m & ~((k.r >> 16) & k.r);
    return k;
}
template <bool NEWRULE>
__device__ __forceinline__ uint32_t win_count16(const WinP16 &k, uint32_t s) {
    const uint32_t t = s ^ k.r;
    if (!NEWRULE) return (uint32_t)__popc(((t >> 16) & k.x) | (t & k.m));
    return (uint32_t)__popc(((t >> 16) | t) & k.m & ~(~(s >> 16) & s & k.x));
}

// the candidates in L.surv[0 .. min(nsurv, 64)): all their reference words in one round trip, exact count (CountMismatch*), bulk bookkeeping
template <int NWT, bool NEWRULE, class LDS>
__device__ bool heavy_flush(const DevCtx &cx, LDS &L, HitState &st, basal_hit *log, const ReadCtx &rc, uint32_t mode, uint32_t &nsurv, uint32_t r, int lane PH_PARAM) {
    const uint32_t batch = nsurv < 64 ? nsurv : 64;
    const bool active = (uint32_t)lane < batch;
    uint32_t loc = 0, strand = 0, chain = 0, mm = 0xffff;
    if (active) {
        const SurvEnt sv = L.surv[lane];
        loc = sv.loc; strand = sv.meta & 1u; chain = (sv.meta >> 1) & 1u;
        if (sv.meta & 4u) loc = cx.locs[loc] - (sv.meta >> 11);  // counted from the stream: sv.loc is the index entry, bits 11.. its seed's read offset
        if (((unsigned long long)(loc >> 5) + NWT + 4) >= COLD(nwords)) loc = (uint32_t)guard_idx(cx, G_XREF, loc, 0, r) + BASAL_REF_MARGIN * 32;
        if (sv.meta & 4u) mm = (sv.meta >> 3) & 0xffu;  // a `full` long list's candidate: counted exactly from the stream (heavy_mode)
        else {
            const uint32_t off2 = (loc & 31) * 2, nw = (rc.len + (loc & 31) + 31) / 32;
            mm = count_mismatch<NWT, NEWRULE>((strand ? cx.xref[1] : cx.xref[0]) + (loc >> 5), L.q[chain], off2, nw, st.thr, rc.n_count);
        }
    }
    PH(PH_SCORE);
    const bool stop = bulk_add(cx, L, st, log, rc, active, loc, strand, chain, mm, mode, r, lane);
    PH(PH_E1);
    const uint32_t rest = nsurv - batch;  // drop the batch from the front of the list
    SurvEnt v = {0, 0};
    if ((uint32_t)lane < rest) v = L.surv[batch + lane];
    wave_sync();
    if ((uint32_t)lane < rest) L.surv[lane] = v;
    nsurv = rest;
    wave_sync();
    return stop;
}

// One mode of SnpAlign (align.cpp:274-316) in the HEAVY kernels. The mode's stream is cut into stretches of short lists, which go through
// the packed chunks of the standard kernel (64 consecutive stream positions, one window test), and long lists (>= cx.heavy_m entries),
// each streamed on its own: every lane of a chunk then shares the seed, so its read windows sit in scalar registers, the loads are
// straight runs of the list, and three windows (up to 96 of the read's bases off the seed) are tested from the coalesced stream. What
// passes is appended IN ORDER to L.surv and scored 64 at a time on full waves (heavy_flush). Returns true when SnpAlign must stop.
template <int NWT, bool NEWRULE, class LDS>
__device__ bool heavy_mode(const DevCtx &cx, LDS &L, HitState &st, basal_hit *log, const ReadCtx &rc, uint32_t mode, uint32_t inc, uint32_t e_m, uint32_t nent,
                           uint32_t T, uint32_t r, int lane PH_PARAM) {
    const uint64_t lt = (1ULL << lane) - 1;
    const uint64_t end_mask = nent > 1 ? (1ULL << (nent - 1)) - 1 : 0;
    const uint32_t nlocs_u = COLD(nlocs);
    const unsigned long long stride = (unsigned long long)nlocs_u + 64;
    uint32_t nsurv = 0;
    bool stop = false;
    uint64_t hv = ballot((uint32_t)lane < nent && e_m >= cx.heavy_m);
    uint32_t tcur = 0;
    for (uint32_t guard_it = 0; guard_it <= 64 && !stop; guard_it++) {
        const int eh = hv ? __ffsll((unsigned long long)hv) - 1 : -1;
        const uint32_t tend = eh >= 0 ? rdlane(inc - e_m, eh) : T;
        if (tcur < tend) {  // a stretch of short lists
            // (two chunks of 64 stream positions per iteration and memory round trip, as for the long lists below)
            constexpr int WP = NWT == 4 ? LONG_WD : 1;
            struct PackedLoads { uint32_t ei, jj, loc_raw; uint64_t f; };
            auto issue = [&](uint32_t t0, PackedLoads *o) {
#pragma unroll
                for (int u = 0; u < WP; u++) {
                    o[u] = PackedLoads{0, 0, 0, 0};
                    if (t0 + 64u * u < tend) {
                        const ChunkLoads c = issue_chunk<false>(cx, L, inc, end_mask, t0 + 64u * u, tend, lane, nlocs_u, stride, r);
                        o[u] = PackedLoads{c.ei, c.jj, c.loc_raw, c.f};
                    }
                }
            };
            PackedLoads nxt[WP];
            issue(tcur, nxt);
            for (uint32_t t0 = tcur; t0 < tend && !stop; t0 += 64u * WP) {
                PackedLoads cur[WP];
#pragma unroll
                for (int u = 0; u < WP; u++) cur[u] = nxt[u];
                issue(t0 + 64u * WP, nxt);
#pragma unroll
                for (int u = 0; u < WP; u++) {
                    if (stop) break;
                    const bool active = t0 + 64u * u + (uint32_t)lane < tend;
                    bool alive = false;
                    SurvEnt sv = {0, 0};
                    if (active) {
                        const uint32_t ei = cur[u].ei, hcs = L.ent[ei].hcs, e_nfwd = L.ent[ei].nfwd;
                        const uint64_t e_fr = L.ent[ei].fr, e_fm = L.ent[ei].fm;  // (as planes: process_read stores them that way for the HEAVY kernels)
                        const WinP k = {(uint32_t)(e_fr >> 32), (uint32_t)e_fr, (uint32_t)(e_fm >> 32), (uint32_t)e_fm};
                        alive = rc.n_count + win_count<NEWRULE>(k, cur[u].f) <= st.thr;
                        sv.loc = cur[u].loc_raw - (hcs & 0xffffu);
                        sv.meta = (uint32_t)(cur[u].jj >= e_nfwd) | (((hcs >> 16) & 1u) << 1);
                    }
                    PH(PH_FILTER);
                    const uint64_t mk = ballot(alive);
                    if (mk) {
                        if (alive) L.surv[nsurv + (uint32_t)__popcll(mk & lt)] = sv;
                        nsurv += (uint32_t)__popcll(mk);
                        wave_sync();
                        if (nsurv >= 64) stop = heavy_flush<NWT, NEWRULE>(cx, L, st, log, rc, mode, nsurv, r, lane PH_ARG);
                    }
                }
            }
            if (stop) break;
        }
        if (eh < 0) break;
        {   // one long list: entry eh
            const uint32_t l_off = rfl(L.ent[eh].off), l_m = rfl(L.ent[eh].m), l_nfwd = rfl(L.ent[eh].nfwd), l_jj0 = rfl(L.ent[eh].jj0), l_hcs = rfl(L.ent[eh].hcs);
            const uint32_t h = l_hcs & 0xffffu, chain = (l_hcs >> 16) & 1u;
            // The index keeps six flank words per entry (the 96 bases after the seed, the 96 before it) and the entry's own 16 bases. Up to four
            // windows are tested per candidate, the ones with the most read bases opposite them first (after0 >= after1 >= after2, likewise
            // before: a merge of two descending runs). When those four cover the whole read outside the seed -- every read of up to 100 bases
            // at K = 16, most seeds of longer ones -- the test IS CountMismatch* (align.h:118-131, 199-239; the count is a sum over disjoint
            // windows): such a list is `full`, its survivors carry their exact count and never touch the reference.
            const uint32_t nA = rc.len - h - cx.K, nB = h;
            WinP wk[4];
            uint32_t wsid[4], covered = 0;
            {
                uint32_t ia = 0, ib = 0;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const uint32_t ca = ia < 3 && nA > 32 * ia ? (nA - 32 * ia < 32 ? nA - 32 * ia : 32u) : 0;
                    const uint32_t cb = ib < 3 && nB > 32 * ib ? (nB - 32 * ib < 32 ? nB - 32 * ib : 32u) : 0;
                    int pos;
                    if (ca >= cb) { covered += ca; pos = (int)(h + cx.K + 32 * ia); wsid[w] = 2 * ia; ia += ca != 0; }
                    else { covered += cb; pos = (int)h - 32 * (int)(ib + 1); wsid[w] = 2 * ib + 1; ib++; }
                    uint64_t wr, wm, wc;
                    plane_window3<NWT, NEWRULE>(L.q[chain], pos, wr, wm, wc);
                    wk[w] = win_make<NEWRULE>(rdlane64(wr, 0), ca == 0 && cb == 0 ? 0 : rdlane64(wm, 0), NEWRULE ? rdlane64(wc, 0) : 0);
                }
            }
            const bool full = covered == nA + nB;
            // the seed's own bases: the k-mer code folds letters 01 and 11 (Param::XT, param.h:107-116), so an entry of this list may differ from
            // the read there; its 16 bases are fetched only when one of the two letters would count as a mismatch against a read base of the seed
            WinP16 sk = {0, 0, 0};
            bool need_seed = false;
            if (full) {
                uint64_t sr, sm, sc;
                plane_window3<NWT, NEWRULE>(L.q[chain], (int)h, sr, sm, sc);
                sk = win_make16<NEWRULE>((uint32_t)(rdlane64(sr, 0) >> 32), (uint32_t)((rdlane64(sm, 0) & (~0ULL << (64 - 2 * cx.K))) >> 32),
                                         NEWRULE ? (uint32_t)(rdlane64(sc, 0) >> 32) : 0);
                const uint32_t fold = (sk.r & 0xffffu) << 16;
                need_seed = (win_count16<NEWRULE>(sk, sk.r & ~fold) | win_count16<NEWRULE>(sk, sk.r | fold)) != 0;
            } else {
                // not exact anyway: a window with fewer than six read bases opposite it is not worth its 8 bytes per candidate, nor is a fourth one
                wk[3].m = wk[3].x = 0;
#pragma unroll
                for (int w = 0; w < 3; w++)
                    if (__popc(wk[w].m) < 6) wk[w].m = wk[w].x = 0;
            }
            const bool has0 = wk[0].m != 0, has1 = wk[1].m != 0, has2 = wk[2].m != 0, has3 = wk[3].m != 0;
            const uint32_t *Lc = cx.locs + l_off;
            const uint64_t *F0 = cx.flank_a + l_off + wsid[0] * stride, *F1 = cx.flank_a + l_off + wsid[1] * stride, *F2 = cx.flank_a + l_off + wsid[2] * stride,
                           *F3 = cx.flank_a + l_off + wsid[3] * stride;
            const uint32_t *Sw = need_seed ? COLDP(const uint32_t, seedw) + l_off : nullptr;
            // Two stages of loads per chunk, each issued ahead of its use: the two widest windows for every candidate; the other windows, the
            // seed word and the location only for the lanes the first two left alive -- a 128-byte line of those is fetched only if one of
            // its 16 (32) candidates is (half of those lines are not, on the hg38-like stand-in).
            // (Loads under branches make the compiler wait for all loads in flight at the first use of any, s_waitcnt vmcnt(0); issuing every
            // load from every lane -- idle lanes reading a line that is cached anyway -- gave exact wait counts and a slower kernel, 150 -> 157 ms
            // per 10 M reads: the address selects cost more vector instructions than the deeper pipeline won, with six waves per SIMD to switch to.)
            // LONG_WD chunks of 64 candidates per iteration, one exposed memory round trip for all of them: the stage-one words of an iteration
            // are requested during the one before; its stage-two words go out as soon as stage one has been looked at, the next iteration's
            // stage-one words right behind them, and the wave waits once for both. (A wave alone on the GPU -- the longest reads at the end of
            // a launch -- is bound by exactly these round trips.)
            constexpr int WD = NWT <= LONG_WD_NWT ? LONG_WD : 1;  // (the longest reads' kernels have no registers to spare)
            struct S1 { uint64_t a[WD], b[WD]; };
            struct S2 { uint32_t loc[WD], sw[WD]; uint64_t c[WD], d[WD]; };
            auto jj_of = [&](uint32_t p) { uint32_t jj = l_jj0 + p; return jj >= l_m ? jj - l_m : jj; };
            // (the stream is read once: non-temporal loads keep it from pushing the index tables and reference lines out of the caches, +1.5 %;
            // the same hint on the survivors' reference words, five loads into one line, cost 20 %)
            auto issue1 = [&](uint32_t p0) {
                S1 c;
#pragma unroll
                for (int u = 0; u < WD; u++) {
                    c.a[u] = c.b[u] = 0;
                    const uint32_t p = p0 + 64u * u + (uint32_t)lane;
                    if (p < l_m) {
                        const uint32_t jj = jj_of(p);
                        if (has0) c.a[u] = __builtin_nontemporal_load(&F0[jj]);
                        if (has1) c.b[u] = __builtin_nontemporal_load(&F1[jj]);
                    }
                }
                return c;
            };
            S1 n1 = issue1(0);
            for (uint32_t p0 = 0; p0 < l_m && !stop; p0 += 64u * WD) {
                uint32_t lb[WD];
                bool alive1[WD];
                S2 c2;
#pragma unroll
                for (int u = 0; u < WD; u++) {
                    lb[u] = rc.n_count + win_count<NEWRULE>(wk[0], n1.a[u]) + win_count<NEWRULE>(wk[1], n1.b[u]);
                    alive1[u] = p0 + 64u * u + (uint32_t)lane < l_m && lb[u] <= st.thr;
                    c2.loc[u] = c2.sw[u] = 0; c2.c[u] = c2.d[u] = 0;
                    if (alive1[u]) {
                        const uint32_t jj = jj_of(p0 + 64u * u + (uint32_t)lane);
                        if (!full) c2.loc[u] = __builtin_nontemporal_load(&Lc[jj]);  // (a full list's survivors fetch theirs when they are scored: 1 in 7 of these lanes)
                        if (has2) c2.c[u] = __builtin_nontemporal_load(&F2[jj]);
                        if (has3) c2.d[u] = __builtin_nontemporal_load(&F3[jj]);
                        if (need_seed) c2.sw[u] = __builtin_nontemporal_load(&Sw[jj]);
                    }
                }
                n1 = issue1(p0 + 64u * WD);  // (past the list: nothing is loaded)
#pragma unroll
                for (int u = 0; u < WD; u++) {
                    if (stop) break;
                    uint32_t mm = lb[u];
                    if (has2) mm += win_count<NEWRULE>(wk[2], c2.c[u]);
                    if (has3) mm += win_count<NEWRULE>(wk[3], c2.d[u]);
                    if (need_seed) mm += win_count16<NEWRULE>(sk, c2.sw[u]);
                    const bool alive = alive1[u] && mm <= st.thr;
#ifdef BASAL_PHASE_TIMING  // long-list chunks; lanes alive after the near windows; 16-lane groups (128-byte lines of the far words) with such a lane; survivors
                    {
                        const uint64_t ab = ballot(alive1[u]);
                        phc.n_chunks++; phc.n_alive += (uint32_t)__popcll(ab);
                        phc.n_bigchunks += ((ab & 0xffffULL) != 0) + ((ab & 0xffff0000ULL) != 0) + ((ab & 0xffff00000000ULL) != 0) + ((ab & 0xffff000000000000ULL) != 0);
                        phc.n_bigalive += (uint32_t)__popcll(ballot(alive));
                    }
#endif
                    PH(PH_FILTER);
                    const uint64_t mk = ballot(alive);
                    if (mk) {
                        if (alive) {
                            SurvEnt sv;
                            const uint32_t jj = jj_of(p0 + 64u * u + (uint32_t)lane);
                            sv.loc = full ? l_off + jj : c2.loc[u] - h;
                            sv.meta = (uint32_t)(jj >= l_nfwd) | (chain << 1) | (full ? 4u | ((mm & 0xffu) << 3) | (h << 11) : 0u);
                            L.surv[nsurv + (uint32_t)__popcll(mk & lt)] = sv;
                        }
                        nsurv += (uint32_t)__popcll(mk);
                        wave_sync();
                        if (nsurv >= 64) stop = heavy_flush<NWT, NEWRULE>(cx, L, st, log, rc, mode, nsurv, r, lane PH_ARG);
                    }
                }
            }
        }
        tcur = rdlane(inc, eh);
        hv &= hv - 1;
    }
    while (!stop && nsurv) stop = heavy_flush<NWT, NEWRULE>(cx, L, st, log, rc, mode, nsurv, r, lane PH_ARG);
    return stop;
}

// ---- the hit stream (-r 2: the best level's hits; paired-end: every mate's whole log) ---------------------------------------------------
// Its records are handed out by ONE counter, and a single memory word takes about 80 M atomic adds per second on this chip -- a paired-end batch,
// one log per mate, ran at exactly that rate whatever the kernel did (config 3: 25.5 ms per 2 M mates before and after its instruction count
// fell by a third). So a read's records (up to 64) are first STAGED in the wave's own scratch -- the first 64 records of its hit-log area, which
// live in registers and are never stored there -- and a chunk of reads takes its place in the stream with one atomic add: the same exact total,
// the same overflow rule per read, an eighth of the atomics. Logs of more than 64 records go straight to the stream as before.
template <class LDS>
__device__ __forceinline__ void stream_flush(const DevCtx &cx, LDS &L, basal_hit *log, int lane) {
    const uint32_t total = rfl(L.stg_n);
    if (!total) return;
    unsigned long long first = 0;
    if (lane0(lane)) first = atomicAdd(COLDP(unsigned long long, stream_used), (unsigned long long)total);
    first = ((unsigned long long)rfl((uint32_t)(first >> 32)) << 32) | rfl((uint32_t)first);
    const unsigned long long cap = COLD(stream_cap);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");  // (the staged records were stored by other lanes of this wave)
    if ((uint32_t)lane < total && first + (uint32_t)lane < cap) COLDP(basal_hit, stream)[first + (uint32_t)lane] = log[lane];
    const uint32_t mask = rfl(L.stg_mask);
    if (lane < WORK_CHUNK && ((mask >> lane) & 1u)) {
        const uint32_t rel = L.res[lane].stream_first;
        L.res[lane].stream_first = (uint32_t)(first + rel);
        if (first + rel + L.res[lane].stream_n > cap) L.res[lane].status = BASAL_READ_OVERFLOW;
    }
    if (lane0(lane)) { L.stg_n = 0; L.stg_mask = 0; }
    wave_sync();
}

// ---- one read ----------------------------------------------------------------------------------
template <int NWT, bool NEWRULE, bool GAP, bool HEAVY, bool PE>
__device__ __forceinline__ void process_read(const DevCtx &cx, WaveLds<NWT, GAP, HEAVY, PE> &L, const uint8_t *tab, basal_hit *log, uint32_t r, uint32_t chunk_slot, basal_read rd,
                             const uint32_t *pre, int pre_c, int lane PH_PARAM) {
    using LDS = WaveLds<NWT, GAP, HEAVY, PE>;
    static_assert(!PE || (!GAP && !HEAVY), "the paired-end instantiations are the standard kernels'");
    const bool allmodes = (rd.readset & BASAL_READ_ALLMODES) != 0;  // a PE mate: PairAlign::RunAlign drives the modes
    rd.readset &= 0x7f;
    basal_result res;
    memset(&res, 0, sizeof(res));
    res.best_level = 0xFF;
    if (rd.len == 0 || rd.len > (uint32_t)NWT * 32 || rd.len > BASAL_MAXREADLEN) {
        res.status = BASAL_READ_SKIPPED;
        if (lane0(lane)) L.res[chunk_slot] = res;
        return;
    }
#if defined(BASAL_PERTURB_VALU) || defined(BASAL_PERTURB_SALU) || defined(BASAL_PERTURB_LDS) || defined(BASAL_PERTURB_MEM)
#include "basal_core_diag.inc"  // sensitivity experiments only
#endif
    ReadCtx rc;
    uint32_t slot = rd.readset == 2 ? 1 : 0;
    // the four carry bytes as one scalar load (a byte indexed by `slot` would be a vector load, and waiting for it
    // would also wait for the bytes of the next read requested just before)
    const uint32_t carry_w = *(const uint32_t __attribute__((address_space(4))) *)&cold_ctx()->carry[0][0] >> (16 * slot);
    uint32_t so0 = carry_w & 0xff, so1 = (carry_w >> 8) & 0xff;
    const bool stale = rd.stale_idx < COLD(nstale);
    if (stale) {  // inherit xseed_start_offset from an earlier read of this batch (align.cpp:475-480)
        uint32_t srcno = COLDP(const basal_stale, stales)[rd.stale_idx].src;
        if (srcno < r || srcno - COLD(ghost_base) < 2u) {
            basal_read src = uniform_read(cx.reads[srcno]);
            rc.rno = r;
            prep_read<NEWRULE>(cx, L, tab, src, rc, lane, nullptr, -1, 0, 0 PH_ARG);
            if (rc.on(0)) so0 = best_start_offset(cx, L, rc, 0, lane, so0);
            if (rc.on(1)) so1 = best_start_offset(cx, L, rc, 1, lane, so1);
        }
    }
    rc.rno = r;
    PH(PH_ENTRY);
    prep_read<NEWRULE>(cx, L, tab, rd, rc, lane, pre, pre_c, so0, so1 PH_ARG);
    if (stale) {  // seed slots past this read's own seeds still hold an earlier read's values
        if (lane < 30) {
            uint32_t c = (uint32_t)lane / 15, j = (uint32_t)lane % 15, pos = rc.npos + j;
            if (rc.on(c) && pos < (uint32_t)LDS::MAXPOS) {
                uint32_t sd = COLDP(const basal_stale, stales)[rd.stale_idx].overlay[c][j];
                sd = (sd & 0x80000000u) | (uint32_t)guard_idx(cx, G_STALE, sd & 0x7fffffffu, COLD(total_kmers), r);
                L.seed[c][pos] = sd;
                sd &= 0x7fffffffu;
                L.cnt[c][pos] = cx.kmer_off[sd + 1] - cx.kmer_off[sd];
            }
        }
        wave_sync();
    }
    if (lane < 32) L.nhit[lane >> 4][lane & 15] = 0;
    if constexpr (HEAVY) {
#pragma unroll
        for (int i = 0; i < (int)(sizeof(L.bloom) / 256); i++) L.bloom[lane + 64 * i] = 0;
    }
    wave_sync();
    reorder_seed(cx, L, rc, lane, so0, so1);
    res.start_off[0] = (uint8_t)so0;
    res.start_off[1] = (uint8_t)so1;
    PH(PH_REORDER);
    HitState st;
    st.thr = rc.max_snp;
    st.nlog = 0;
    st.d0 = st.d1 = st.d2 = st.d3 = 0;
    st.bloom = 0;
    st.g = 0;
    const uint32_t rnd = myrand(rc.index, cx.randseed);
    const uint32_t n1 = rfl(2 * cx.I);  // seeds per mode
    // PE kernels: a mate of a pair runs every mode whatever the earlier ones found (PairAlign::RunAlign drives them, pairs.cpp:164-174), so its modes
    // are set up and streamed in GROUPS (BASAL_PE_ENT seed entries: four modes at -I 4): one set-up pass, one header round trip and a few full chunks
    // where four modes took four of each. The stream of a group is the modes' streams back to back -- the order SnpAlign visits the candidates in -- every
    // candidate knows its mode (bits 20.. of its seed entry), and a level-0 cap, which ends only the SnpAlign call it happens in, restarts the
    // grouping at the mode behind it. Everything else runs one mode at a time, as before.
    const uint32_t ent_mo = PE ? ((uint32_t)lane * s_rcp[n1]) >> 16 : 0u, ent_w = (uint32_t)lane - ent_mo * n1;  // lane -> (mode within the group, seed of the mode)
    const uint32_t ent_c = ent_w >= cx.I ? 1u : 0u, ent_i = ent_w >= cx.I ? ent_w - cx.I : ent_w;

    bool done = false;
    bool capped = false;  // PE: a level-0 cap has ended one of this read's modes (a read from a repeat family: its later modes run one at a time)
    for (uint32_t mode = 0; mode < rc.nseg && !done; mode++) {
        uint32_t G = 1;
        if constexpr (PE) {
            if (allmodes && !capped) {
                const uint32_t gmax = n1 * 8u <= (uint32_t)BASAL_PE_ENT ? 8u : (uint32_t)BASAL_PE_ENT / n1;
                G = rc.nseg - mode < gmax ? rc.nseg - mode : gmax;
            }
        }
        const uint32_t nent = PE ? G * n1 : n1;
        uint32_t stop_mode = 0xffffffffu;  // PE: the mode a level-0 cap ended
        // the seeds of this mode, chain-major then phase (the order SnpAlign visits them, align.cpp:275-279)
        uint32_t e_m = 0, e_off = 0, e_nfwd = 0, e_h = 0, e_jj0 = 0, e_chain = 0;
        if ((uint32_t)lane < nent) {
            const uint32_t c = ent_c, i = ent_i;
            if (rc.on(c)) {
                uint32_t seg = L.order[c][(mode + ent_mo) & 15];
                uint32_t pos = s_prof[seg & 15][i] + L.start_arr[c][seg & 15] - i;
                pos = (uint32_t)guard_idx(cx, G_LDSPOS, pos, LDS::MAXPOS, r);
                uint32_t sd = L.seed[c][pos] & 0x7fffffffu, m = L.cnt[c][pos];
                if (sd >= COLD(total_kmers)) sd = (uint32_t)guard_idx(cx, G_KMER2, 0x80000000u | pos | (seg << 16) | (c << 24) | (mode << 26), 0, r);
                if (m != 0 && m <= cx.max_kmer_num) {
                    e_m = m;
                    e_off = cx.kmer_off[sd];
                    e_nfwd = cx.kmer_nfwd[sd];
                    e_h = pos;
                    e_jj0 = rnd % m;
                    e_chain = c;
                }
            }
        }
        uint32_t inc = e_m;  // inclusive prefix sum over lanes 0..nent-1
        if (nent <= 16) inc = row16_scan_add(inc);
        else
            for (int o = 1; o < (PE ? 64 : 32); o <<= 1) {
                uint32_t v = __shfl_up(inc, o);
                if (lane >= o) inc += v;
            }
        if ((uint32_t)lane < nent) {
            // which flank of the seed has more read bases opposite it (the window tested is 32 bases; GAP kernels: 64 there, 32 on the other side)
            constexpr int WIN = GAP ? 64 : 32;
            int n_after = (int)rc.len - (int)(e_h + cx.K), n_before = (int)e_h;
            n_after = n_after < 0 ? 0 : n_after > WIN ? WIN : n_after;
            n_before = n_before > WIN ? WIN : n_before;
            const uint32_t side = (uint32_t)(n_before > n_after);
            uint64_t wr = 0, wm = 0, wc = 0;
            if (!GAP) plane_window3<NWT, NEWRULE>(L.q[e_chain], side ? (int)e_h - 32 : (int)(e_h + cx.K), wr, wm, wc);
            // the list must lie inside locs[]: checked here, once per seed, so that the stream's chunks need not check every position
            if ((unsigned long long)e_off + e_m > COLD(nlocs)) e_off = guard_u32(cx, G_LOCS, e_off + e_m, 0, r);
            SeedEntT<GAP> e;
            e.off = e_off; e.m = e_m; e.nfwd = e_nfwd; e.jj0 = e_jj0; e.pre = inc - e_m; e.hcs = e_h | (e_chain << 16) | (side << 17) | (PE ? ent_mo << 20 : 0u);
            if constexpr (HEAVY && !GAP) {  // the flank words of such a core are bit planes: the read's window likewise (win_make)
                const WinP k = win_make<NEWRULE>(wr, wm, wc);
                e.fr = (uint64_t)k.rh << 32 | k.rl; e.fm = (uint64_t)k.x << 32 | k.m; e.fc = 0;
            } else if constexpr (!GAP) { e.fr = wr; e.fm = wm; e.fc = wc; }
            L.ent[lane] = e;
            if constexpr (GAP) {
                const int pl = side ? (int)e_h - 64 : (int)(e_h + cx.K), ps = side ? (int)(e_h + cx.K) : (int)e_h - 32;
                SeedEntPl P;
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    P.ml[x] = bits64<NWT / 2>(L.mmp[e_chain][x], pl);
                    P.ms[x] = (uint32_t)bits64<NWT / 2>(L.mmp[e_chain][x], ps);
                }
                P.vl = bits64<NWT / 2>(L.valp[e_chain], pl);
                P.vs = (uint32_t)bits64<NWT / 2>(L.valp[e_chain], ps);
                P.pad = 0;
                L.entp[lane] = P;
                if constexpr (HEAVY) {  // window bit b of a window that starts at read position w0 is read base w0 + b
                    const int ge = (int)cx.gap_edge, len = (int)rc.len;
                    SeedEndMask X;
                    X.fl = low_mask64(ge - pl) & ~low_mask64(-pl);
                    X.el = low_mask64(len - pl) & ~low_mask64(len - ge - pl);
                    X.fs = (uint32_t)(low_mask64(ge - ps) & ~low_mask64(-ps));
                    X.es = (uint32_t)(low_mask64(len - ps) & ~low_mask64(len - ge - ps));
                    L.entx[lane] = X;
                }
            }
        }
        wave_sync();
        const uint32_t T = rdlane(inc, (int)nent - 1);
        PH(PH_MODE);

        if constexpr (HEAVY && !GAP) {
            done = heavy_mode<NWT, NEWRULE>(cx, L, st, log, rc, mode, inc, e_m, nent, T, r, lane PH_ARG);
        } else {
        // Non-GAP: 64 candidates of the stream per iteration. GAP: the flank tests run on the stream, the candidates they
        // cannot rule out are compacted (order kept) into L.surv and scored + gap-searched 64 at a time, so the expensive
        // part runs on full waves instead of on the ~quarter of the lanes that survive.
        const uint64_t lt = (1ULL << lane) - 1;
        const uint64_t end_mask = nent > 1 ? (1ULL << (nent - 1)) - 1 : 0;  // lanes holding a list end that counts (e < nent - 1)
        const uint32_t nlocs_u = COLD(nlocs);
        const unsigned long long flank_b_off = (unsigned long long)nlocs_u + 64;  // flank_b = flank_a + nlocs + 64
        uint32_t nsurv = 0, batch = 0;
        ChunkLoads nxt = {0, 0, 0, 0, 0, 0};
        if (T > 0) nxt = issue_chunk<GAP>(cx, L, inc, end_mask, 0, T, lane, nlocs_u, flank_b_off, r);
        for (uint32_t t0 = 0; (t0 < T || (GAP && nsurv > 0)) && !done;) {
            uint32_t t;
            bool active;
            if constexpr (GAP) {
                if (t0 < T && nsurv < 64) {
                    const uint32_t tf = t0 + lane;
                    const bool af = tf < T;
                    bool keep = false;
                    // this chunk's location and flank words were requested one chunk ago; the next chunk's go out now
                    const ChunkLoads cur = nxt;
                    if (t0 + 64 < T) nxt = issue_chunk<true>(cx, L, inc, end_mask, t0 + 64, T, lane, nlocs_u, flank_b_off, r);
                    SurvEnt sv = {0, 0};
                    if (af) {
                        // The candidate's reference bases around the seed came with its location, as bit planes: 64 bases on the side of the
                        // seed that has more read bases opposite it, 32 on the other. Three bounds, none of which touches the reference:
                        //  * the ungapped count (CountMismatch*, align.h:118-239) is at least the windows' mismatches at valid bases + the N count;
                        //  * the gap search gives up at once when the read prefix up to the seed end holds thr-1 mismatches (MismatchPattern0's
                        //    return value vs seed_pos+seed_size, align.cpp:365; no N mask there): the window(s) before the seed are part of it;
                        //  * a gapped hit at shift s (align.cpp:367-405) is a left part [0, gap_pos) at the candidate's start holding i mismatches
                        //    and a right part of m2 bases at start+s holding j, i + j + t <= thr - 1, which together cover the read (s > 0) or
                        //    all of it but t bases (s < 0): a base that mismatches at BOTH starts is counted in i or j wherever the gap falls, so
                        //    i + j >= popc(D0 & Ds) over any window (minus t for s < 0).
                        const uint32_t eif = cur.ei;
                        const uint32_t e_hcs = L.ent[eif].hcs, e_nfwd = L.ent[eif].nfwd;
                        const SeedEntPl P = L.entp[eif];
                        const bool before = ((e_hcs >> 17) & 1u) != 0;
                        const uint32_t lc = cur.loc_raw - (e_hcs & 0xffffu);
                        const uint32_t a_lo = (uint32_t)cur.f, a_hi = (uint32_t)(cur.f >> 32), b_lo = (uint32_t)cur.fb, b_hi = (uint32_t)(cur.fb >> 32);
                        const uint32_t f_lo = (uint32_t)cur.f2, f_hi = (uint32_t)(cur.f2 >> 32);
                        // long window: after the seed = [near | far << 32], before it = [far | near << 32] (bit i = base i of the 64); selects, not branches
                        const uint32_t bm = 0u - (uint32_t)before;
                        const uint64_t Llo = ((uint64_t)bsel(bm, b_lo, f_lo) << 32) | bsel(bm, f_lo, a_lo);
                        const uint64_t Lhi = ((uint64_t)bsel(bm, b_hi, f_hi) << 32) | bsel(bm, f_hi, a_hi);
                        const uint32_t Slo = bsel(bm, a_lo, b_lo), Shi = bsel(bm, a_hi, b_hi);
                        const uint64_t D0l = plane_mismatch(Lhi, Llo, P.ml);
                        const uint32_t D0s = plane_mismatch(Shi, Slo, P.ms);
                        const uint32_t lb = rc.n_count + popc64(D0l & P.vl) + (uint32_t)__popc(D0s & P.vs);
                        const bool al = lb <= st.thr;
                        const uint32_t pre_mm = bsel(bm, popc64(D0l), (uint32_t)__popc(D0s));
                        bool gk = false;
                        // (the lanes of a chunk mostly share one seed, hence one window geometry: where the prefix test fails it fails for the whole wave)
                        if (st.thr >= 2 && pre_mm < st.thr - 1) {
                            // HEAVY: among the read's first gap_edge bases a mismatch at the candidate's start counts whatever the shifted start says,
                            // among its last gap_edge bases one at the shifted start (SeedEndMask): D0 & (Ds | F) | Ds & E instead of D0 & Ds
                            [[maybe_unused]] uint64_t El = 0, F0l = 0;
                            [[maybe_unused]] uint32_t Es = 0, F0s = 0;
                            if constexpr (HEAVY) { const SeedEndMask X = L.entx[eif]; El = X.el; Es = X.es; F0l = D0l & X.fl; F0s = D0s & X.fs; }
#pragma unroll
                            for (uint32_t tg = 1; tg <= BASAL_MAXGAPS; tg++) {  // tt = 2 tg - 1 (shift -tg), then tt = 2 tg (shift +tg); wave-uniform conditions
                                if (tg <= cx.gap && st.thr >= 1 + tg) {
                                    // shift -t: read base i against window base i - t
                                    const uint64_t Dm = plane_mismatch_sh(Lhi << tg, Llo << tg, P.ml) & (~0ULL << tg);
                                    const uint32_t dm = plane_mismatch(Shi << tg, Slo << tg, P.ms) & (~0u << tg);
                                    if constexpr (HEAVY) gk |= popc64((Dm & (D0l | El)) | F0l) + (uint32_t)__popc((dm & (D0s | Es)) | F0s) <= st.thr - 1;
                                    else gk |= popc64(D0l & Dm) + (uint32_t)__popc(D0s & dm) <= st.thr - 1;
                                    // shift +t
                                    const uint64_t Dp = plane_mismatch_sh(Lhi >> tg, Llo >> tg, P.ml) & (~0ULL >> tg);
                                    const uint32_t dp = plane_mismatch(Shi >> tg, Slo >> tg, P.ms) & (~0u >> tg);
                                    if constexpr (HEAVY) gk |= popc64((Dp & (D0l | El)) | F0l) + (uint32_t)__popc((dp & (D0s | Es)) | F0s) <= st.thr - 1 - tg;
                                    else gk |= popc64(D0l & Dp) + (uint32_t)__popc(D0s & dp) <= st.thr - 1 - tg;
                                }
                            }
                        }
                        keep = al || gk;
                        sv.loc = lc;
                        sv.meta = eif | ((uint32_t)(cur.jj >= e_nfwd) << 8) | ((uint32_t)al << 9) | ((uint32_t)gk << 10);
#ifdef BASAL_CHECK_FILTER  // diagnostic build (`make chk`): every candidate is scored exactly, and one the bounds would have dropped must not be accepted
                        sv.meta = eif | ((uint32_t)(cur.jj >= e_nfwd) << 8) | (1u << 9) | ((uint32_t)(st.thr >= 2) << 10) | ((uint32_t)al << 11) | ((uint32_t)gk << 12);
                        keep = true;
#endif
                    }
                    PH(PH_FILTER);
                    uint64_t mk = ballot(keep);
#ifdef BASAL_PHASE_TIMING
                    phc.n_chunks += (uint32_t)__popcll(ballot(af)); phc.n_alive += (uint32_t)__popcll(mk);
                    phc.n_bigchunks += (uint32_t)__popcll(ballot(af && (sv.meta >> 9 & 1))); phc.n_bigalive += (uint32_t)__popcll(ballot(af && (sv.meta >> 10 & 1)));
#endif
                    if (keep) L.surv[nsurv + (uint32_t)__popcll(mk & lt)] = sv;
                    nsurv += (uint32_t)__popcll(mk);
                    t0 += 64;
                    wave_sync();
                    if (t0 < T && nsurv < 64) continue;
                }
                if constexpr (HEAVY) {  // the survivors' stage on bit planes, both hits of a candidate booked in bulk
                    if (nsurv) done = gap_flush<NWT, NEWRULE>(cx, L, st, log, rc, mode, nsurv, r, lane PH_ARG);
                    continue;
                }
                batch = nsurv < 64 ? nsurv : 64;
                if (batch == 0) continue;
                active = (uint32_t)lane < batch;
                t = 0;
            } else {
                t = t0 + lane;
                active = t < T;
                t0 += 64;
            }
            // which seed's list candidate t belongs to = the number of list ends (inc[e], e < nent-1) that are <= t.
            // Non-GAP: the chunk is 64 consecutive t, so that is the count at t0 plus the ends inside the chunk (one
            // or two, typically) -- cheaper than comparing every lane against every end.
            uint32_t ei = 0, loc = 0, strand = 0, mm = 0xffff, hcs = 0;
#ifdef BASAL_CHECK_FILTER
            uint32_t chk = 3;
#endif
            bool gap_ok = false;  // GAP: the flank tests leave the gap search a chance
            uint64_t W[GAP ? NWT + 2 : 1], D0[GAP ? NWT : 1];  // GAP: the candidate's reference words and ungapped mismatch bitmap
            uint32_t rel0 = 0;
            if constexpr (GAP) {
                // one round trip fetches the reference words of all 2g+1 start positions; the ungapped count (the bitmap under
                // the valid mask, plus the N count) and the gap search both work from these registers
                bool alive = false;
#ifdef BASAL_CHECK_FILTER
                chk = active ? (L.surv[lane].meta >> 11) & 3u : 3u;
#endif
                if (active) {
                    const SurvEnt sv = L.surv[lane];
                    ei = sv.meta & 0xff;
                    strand = (sv.meta >> 8) & 1;
                    alive = (sv.meta >> 9) & 1;
                    gap_ok = (sv.meta >> 10) & 1;
                    loc = sv.loc;
                    if (((unsigned long long)(loc >> 5) + NWT + 4) >= COLD(nwords)) loc = (uint32_t)guard_idx(cx, G_XREF, loc, 0, r) + BASAL_REF_MARGIN * 32;
                    hcs = L.ent[ei].hcs;
                }
                const uint64_t(*qg)[NWT + 1] = L.q[(hcs >> 16) & 1];
                const uint32_t first = loc - cx.gap;  // loc >= 12320 by construction
                rel0 = (first & 31) + cx.gap;
                if (active) {
                    const uint64_t *sp = (strand ? cx.xref[1] : cx.xref[0]) + (first >> 5);  // a select of two scalars, not an indexed load of the argument block
#pragma unroll
                    for (int i = 0; i < NWT + 2; i++) W[i] = (uint32_t)i <= rc.end_element + 2 ? sp[i] : 0;
                    mismatch_map_regs<NWT, NEWRULE>(W, rel0, qg, rc.end_element, rc.end_offset, D0);
                    if (alive) {
                        mm = rc.n_count;
#pragma unroll
                        for (int i = 0; i < NWT; i++) mm += popc64(D0[i] & qg[1][i] & kPairLo);
                    }
                }
            }
            const uint64_t(*q)[NWT + 1] = L.q[0];
            if constexpr (!GAP) {
                // this chunk's location and flank word were requested one chunk ago
                const ChunkLoads cur = nxt;
                // the next chunk's loads go out before this one is looked at: they overlap its filter, exact scoring and replay
                if (t0 < T) nxt = issue_chunk<false>(cx, L, inc, end_mask, t0, T, lane, nlocs_u, flank_b_off, r);
                ei = active ? cur.ei : 0;
                hcs = L.ent[ei].hcs;
                q = L.q[(hcs >> 16) & 1];
                bool alive = false;
                uint32_t lb_first = 0;
                if (active) {
                    const uint32_t e_nfwd = L.ent[ei].nfwd;
                    const uint64_t e_fr = L.ent[ei].fr, e_fm = L.ent[ei].fm, e_fc = NEWRULE ? L.ent[ei].fc : 0;
                    loc = cur.loc_raw - (hcs & 0xffffu);
                    strand = cur.jj >= e_nfwd;
                    // flank pre-filter on the coalesced stream: a lower bound of the mismatch count
                    const uint32_t lb = rc.n_count + XM64(cmp_word<NEWRULE>(e_fr, e_fc, cur.f) & e_fm);
                    alive = lb <= st.thr;
                    lb_first = lb;
                    PH(PH_FILTER);
                }
#ifdef BASAL_PHASE_TIMING
                {
                    uint32_t na = (uint32_t)__popcll(ballot(alive));
                    phc.n_chunks++; phc.n_alive += na;
                    if (T >= 1024) { phc.n_bigchunks++; phc.n_bigalive += na; }
                }
#endif
                if (ballot(alive) == 0) continue;  // most chunks: nothing passed the filter, nothing to score or replay
                // A mode with a long stream (a read from a repeat family: thousands of near-copies, half of which pass one 32-base window):
                // what passed is tested against the window on the seed's OTHER side too before it may touch the reference -- 8 coalesced
                // bytes per candidate against a 128-byte reference line per survivor. Fetched here, for the lanes still alive, rather
                // than with the stream (two more registers live across the loop cost the common read 4 %). It is one more dependent round
                // trip per chunk, which pays in throughput where long streams are the rule -- an index whose over-represented-k-mer
                // cut-off is high (107 091 on the repeat-realistic genome: +30 %) -- and costs latency where they are the exception (11 507 on
                // the uniform stand-in, whose one planted family then ends a 50 000-read launch 10-30 % later): hence the second condition.
                // (the launch folds both into cx.win2_min_T: 1 024, or never)
                if (__builtin_expect(T >= cx.win2_min_T, 0)) {
                    if (alive) {
                        const uint32_t h = hcs & 0xffffu;
                        const uint64_t fo = cx.flank_a[(unsigned long long)guard_u32(cx, G_LOCS, L.ent[ei].off + cur.jj, nlocs_u, r) + (((hcs >> 17) & 1u) ? 0ULL : flank_b_off)];
                        uint64_t wr, wm, wc;
                        plane_window3<NWT, NEWRULE>(q, ((hcs >> 17) & 1u) ? (int)(h + cx.K) : (int)h - 32, wr, wm, wc);
                        alive = lb_first + XM64(cmp_word<NEWRULE>(wr, wc, fo) & wm) <= st.thr;
                    }
                    if (ballot(alive) == 0) continue;
                }
                // A mate of a pair runs every mode (PairAlign::RunAlign drives them, pairs.cpp:164-174), and every mode's seeds find the read's loci again:
                // 25 filter survivors per mate on the transcriptome stand-in, 24 of them placements the log already holds -- each scored against the
                // reference and taken through int2hit + AddHit only to be refused as a duplicate (align.h:332-337). A survivor at a global coordinate
                // (same strand, same chain, ungapped) that one of the 64 records in registers was found at IS such a duplicate: AddHit's key is
                // (contig, coordinate), a function of exactly that -- dropped here, before it costs anything. (Records 64.. are not looked at: what this
                // misses goes the ordinary way.)
                // (chunks with a handful of survivors -- the unique read's case; a chunk full of a repeat family's copies goes the ordinary way)
                if (PE && allmodes && __popcll(ballot(alive)) <= 8) {
                    const uint64_t in_regs = st.nlog >= 64 ? ~0ULL : (1ULL << st.nlog) - 1;
                    const uint32_t sk = ((strand << 1) | ((hcs >> 16) & 1u)) << 8;  // basal_hit.strand next to gap_size 0, as word 2 of a record holds them
                    uint64_t known = 0;
                    for (uint64_t am = ballot(alive); am;) {
                        const int l = __ffsll((unsigned long long)am) - 1;
                        const uint32_t kl = rdlane(loc, l), ks = rdlane(sk, l);
                        // (a group's modes sit side by side in the stream, so the same placement comes several times in ONE chunk too: behind its first
                        // survivor the others are duplicates of it if it is stored, and refused for the reason it is refused if it is not)
                        const uint64_t same = ballot(alive && loc == kl && sk == ks);
                        const bool stored = (ballot(st.g == kl && (st.d2 & 0xffffu) == ks) & in_regs) != 0;
                        known |= stored ? same : same & ~(1ULL << l);
                        am &= ~same;
                    }
                    if ((known >> lane) & 1) alive = false;
                    if (ballot(alive) == 0) continue;
                }
                if (alive) {
                    // (the bounds of the reference are checked where it is about to be read, not for every candidate of the stream)
                    if (((unsigned long long)(loc >> 5) + NWT + 4) >= COLD(nwords)) loc = (uint32_t)guard_idx(cx, G_XREF, loc, 0, r) + BASAL_REF_MARGIN * 32;
                    uint32_t off2 = (loc & 31) * 2;
                    uint32_t nw = (rc.len + (loc & 31) + 31) / 32;
                    // (strand ? a : b selects between two scalar registers; cx.xref[strand] would be a vector load from the argument block and a
                    // wait for it -- and with it for the next chunk's loads, which are in flight by now)
                    mm = count_mismatch<NWT, NEWRULE>((strand ? cx.xref[1] : cx.xref[0]) + (loc >> 5), q, off2, nw, st.thr, rc.n_count);
                }
            } else q = L.q[(hcs >> 16) & 1];
            PH(PH_SCORE);
#ifdef BASAL_CHECK_FILTER
            if (GAP && active && mm <= st.thr && !(chk & 1)) guard_idx(cx, G_WATCHDOG, 0x20000u | mm, 0, r);
#endif
            uint64_t act = ballot(active);
            uint64_t ung_pending = act, gap_pending = GAP ? (act & ballot(gap_ok)) : 0;
            bool gfound = false;
            uint32_t gsnp = 0, gpos = 0;
            int gshift = 0;
            for (uint32_t spin = 0;; spin++) {
                if (spin > 2048) { guard_idx(cx, G_WATCHDOG, 0x10000u | spin, 0, r); done = true; break; }
                if (GAP) {
                    bool mine = (gap_pending >> lane) & 1;
                    gfound = false;
                    if (mine) gfound = gap_align<NWT, NEWRULE>(cx, W, rel0, D0, q, rc, st.thr, hcs & 0xffffu, gsnp, gpos, gshift);
#ifdef BASAL_CHECK_FILTER
                    if (gfound && !(chk & 2)) guard_idx(cx, G_WATCHDOG, 0x30000u | (gsnp << 8) | (uint32_t)(gshift & 0xff), 0, r);
#endif
                }
                uint64_t acc = ballot(mm <= st.thr) & ung_pending;  // ung_pending holds active lanes only
                uint64_t gm = GAP ? (ballot(gfound) & gap_pending) : 0;
                bool recompute = false;
                while (acc | gm) {
                    int l = __ffsll((unsigned long long)(acc | gm)) - 1;
                    uint64_t bit = 1ULL << l;
                    uint32_t thr_before = st.thr;
                    uint32_t lloc = rdlane(loc, l), lstrand = rdlane(strand, l), lchain = rdlane((hcs >> 16) & 1, l);
                    if (acc & bit) {
                        acc &= ~bit;
                        ung_pending &= ~bit;
                        uint32_t lmm = rdlane(mm, l);
                        PH(PH_REPLAY);
                        const uint32_t lmode = PE ? mode + (rdlane(hcs, l) >> 20) : mode;
                        const uint32_t stop = add_hit(cx, L, st, log, rc, lloc, lstrand, lchain, lmm, lmode, 0, 0, lane);
                        PH(PH_E1);  // diagnostic build: AddHit's own time, apart from the replay loop around it
                        if (stop) { done = true; stop_mode = lmode; break; }
                        if (st.thr != thr_before) {
                            acc = ballot(mm <= st.thr) & acc;
                            if (GAP) { gap_pending &= ~(bit - 1); recompute = true; break; }
                        }
                    }
                    if (gm & bit) {
                        gm &= ~bit;
                        gap_pending &= ~bit;
                        uint32_t lsnp = rdlane(gsnp, l), lgp = rdlane(gpos, l);
                        int lsh = (int)rdlane((uint32_t)gshift, l);
                        thr_before = st.thr;
                        if (add_hit(cx, L, st, log, rc, lloc, lstrand, lchain, lsnp, mode, lsh, lgp, lane)) { done = true; break; }
                        if (st.thr != thr_before) {
                            acc = ballot(mm <= st.thr) & acc;
                            gap_pending &= ~((bit << 1) - 1);
                            recompute = true;
                            break;
                        }
                    }
                }
                if (!recompute || done) break;
            }
            PH(PH_REPLAY);
            if constexpr (GAP) {  // drop the processed batch from the front of the list
                uint32_t rest = nsurv - batch;
                SurvEnt v = {0, 0};
                if ((uint32_t)lane < rest) v = L.surv[batch + lane];
                wave_sync();
                if ((uint32_t)lane < rest) L.surv[lane] = v;
                nsurv = rest;
                wave_sync();
            }
        }
        }  // !HEAVY
        // RunAlign: stop once any level <= mode holds a hit (align.cpp:462); `done` = AddHit said stop.
        // For a PE mate the stop only ends this SnpAlign call; the next mode still runs (pairs.cpp:164-174).
        if (allmodes) {
            done = false;
            if constexpr (PE) {  // (the loop's ++ steps behind the group, or behind the mode the cap ended)
                capped |= stop_mode != 0xffffffffu;
                mode = stop_mode != 0xffffffffu ? stop_mode : mode + G - 1;
            }
        } else {
            uint32_t any = 0;
            if ((uint32_t)lane <= mode && lane < 16) any = L.nhit[0][lane] | L.nhit[1][lane];
            if (ballot(any != 0)) done = true;
        }
    }

    // ---- StringAlign's choice (align.cpp:583-612) ----
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    uint32_t tot = 0;
    if (lane < 16 && (uint32_t)lane <= rc.max_snp) tot = (uint32_t)L.nhit[0][lane] + L.nhit[1][lane];
    uint64_t nz = ballot(tot != 0);
    if (nz) {
        uint32_t ii = (uint32_t)__ffsll((unsigned long long)nz) - 1;
        uint32_t nh = L.nhit[0][ii], nc = L.nhit[1][ii], sum = nh + nc;
        res.best_level = (uint8_t)ii;
        res.n_hit = (uint16_t)nh;
        res.n_chit = (uint16_t)nc;
        uint32_t j = sum == 1 ? 0 : rnd % sum;
        uint32_t idx = j < nh ? find_kth(st, log, ii, 0, j, lane) : find_kth(st, log, ii, 1, j - nh, lane);
        if (idx != 0xffffffffu) res.best = log_record(st, log, idx);
        if constexpr (PE) {
        if (COLD(stream_mode) == BASAL_STREAM_BEST || COLD(stream_mode) == BASAL_STREAM_ALL) {
            uint32_t need = COLD(stream_mode) == BASAL_STREAM_ALL ? st.nlog : sum;
            res.stream_n = need;
            // (PE kernels only -- every mate has a log there; the other instantiations keep one atomic per read with hits and their register budget)
            const bool staged = need <= 64;  // (stream_flush gives the chunk's staged records their place in the stream)
            unsigned long long first = 0;
            if (staged) {
                if (need > 64 - rfl(L.stg_n)) stream_flush(cx, L, log, lane);  // make room: the reads staged so far take their place now
                first = rfl(L.stg_n);
            } else {
                if (lane0(lane)) first = atomicAdd(COLDP(unsigned long long, stream_used), (unsigned long long)need);
                first = ((unsigned long long)rfl((uint32_t)(first >> 32)) << 32) | rfl((uint32_t)first);
                if (first + need > COLD(stream_cap)) res.status = BASAL_READ_OVERFLOW;
            }
            // (two stores under a wave-uniform branch: a select between the two pointers would compile to flat stores)
            auto put = [&](unsigned long long at, const basal_hit &h) { if (staged) log[at] = h; else COLDP(basal_hit, stream)[at] = h; };
            res.stream_first = (uint32_t)first;
            if (res.status == BASAL_READ_OVERFLOW) {
            } else if (COLD(stream_mode) == BASAL_STREAM_ALL) {
                for (uint32_t base = 0; base < st.nlog; base += 64)
                    if (base + lane < st.nlog) put(first + base + lane, log_lane_record(st, log, base, lane));
            } else {
                uint32_t outp = 0;
                for (uint32_t c = 0; c < 2; c++)
                    for (uint32_t base = 0; base < st.nlog; base += 64) {
                        bool m = false;
                        basal_hit h;
                        if (base + lane < st.nlog) {
                            h = log_lane_record(st, log, base, lane);
                            m = h.level == ii && h.chain == c;
                        }
                        uint64_t b = ballot(m);
                        if (m) put(first + outp + (uint32_t)__popcll(b & ((1ULL << lane) - 1)), h);
                        outp += (uint32_t)__popcll(b);
                    }
            }
            if (staged) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // other lanes read the staged records back (stream_flush)
                if (lane0(lane)) { L.stg_n = (uint32_t)first + need; L.stg_mask |= 1u << chunk_slot; }
                wave_sync();
            }
        }
        } else if (COLD(stream_mode) == BASAL_STREAM_BEST || COLD(stream_mode) == BASAL_STREAM_ALL) {
            uint32_t need = COLD(stream_mode) == BASAL_STREAM_ALL ? st.nlog : sum;
            unsigned long long first = 0;
            if (lane0(lane)) first = atomicAdd(COLDP(unsigned long long, stream_used), (unsigned long long)need);
            first = ((unsigned long long)rfl((uint32_t)(first >> 32)) << 32) | rfl((uint32_t)first);
            res.stream_first = (uint32_t)first;
            res.stream_n = need;
            if (first + need > COLD(stream_cap)) res.status = BASAL_READ_OVERFLOW;
            else if (COLD(stream_mode) == BASAL_STREAM_ALL) {
                for (uint32_t base = 0; base < st.nlog; base += 64)
                    if (base + lane < st.nlog) COLDP(basal_hit, stream)[first + base + lane] = log_lane_record(st, log, base, lane);
            } else {
                uint32_t outp = 0;
                for (uint32_t c = 0; c < 2; c++)
                    for (uint32_t base = 0; base < st.nlog; base += 64) {
                        bool m = false;
                        basal_hit h;
                        if (base + lane < st.nlog) {
                            h = log_lane_record(st, log, base, lane);
                            m = h.level == ii && h.chain == c;
                        }
                        uint64_t b = ballot(m);
                        if (m) COLDP(basal_hit, stream)[first + outp + (uint32_t)__popcll(b & ((1ULL << lane) - 1))] = h;
                        outp += (uint32_t)__popcll(b);
                    }
            }
        }
    } else if (COLD(stream_mode) == BASAL_STREAM_ALL && st.nlog) {
        // hits exist only above read_max_snp_num: cannot happen (levels are <= thr <= max_snp); kept for safety
        res.stream_n = 0;
    }
    if (lane0(lane)) L.res[chunk_slot] = res;
    PH(PH_FINAL);
}

// Resident waves per SIMD the register allocator is held to (= 256-thread blocks per CU). The kernel is
// bound by dependent memory round trips per read, so throughput ~ resident waves / per-read latency.
// Measured on the bench workload (NWT=4, no gap): 100 / 126 / 133 Mreads/s at 4 / 6 / 8 waves per SIMD -- the few
// registers spilled to scratch at 64 VGPRs cost less than the extra waves bring. Longer reads keep more planes and
// bitmaps in registers and get fewer waves.
#ifndef BASAL_W4NG
#define BASAL_W4NG 8
#endif
#ifndef BASAL_W4G
#define BASAL_W4G 5  // 96 VGPRs + 104 B of scratch: +4 % over 4 waves at 128 VGPRs (configs 4 and 5p); the LDS allows no sixth block
#endif
#ifndef BASAL_W8NG
#define BASAL_W8NG 5
#endif
#ifndef BASAL_W8G
#define BASAL_W8G 3  // (asked for four, the compiler kept 189 registers and the kernels ran two waves; three = 168 registers, one spill: 150-base -g 2 reads 50 -> 62.5 Mreads/s)
#endif
#ifndef BASAL_W16NG
#define BASAL_W16NG 4
#endif
#ifndef BASAL_W16G
#define BASAL_W16G 2
#endif
#ifndef BASAL_W4H
#define BASAL_W4H 5  // HEAVY: 96 registers for the two-chunk long-list loop (heavy_mode); six waves with 80: 132 against 125 ms per 10 M reads
#endif
#ifndef BASAL_W4GH
#define BASAL_W4GH 4  // HEAVY GAP: the Bloom filter and the buckets on top of the GAP kernels' LDS leave room for four blocks per CU
#endif
#ifndef BASAL_W8GH
#define BASAL_W8GH 3
#endif
constexpr int waves_per_simd(int nwt, bool gap, bool heavy = false, bool pe = false) {
    // (HEAVY with longer reads: the survivor list and the Bloom filter leave the LDS room for 4 / 3 blocks per CU)
    if (pe) return BASAL_PE_ENT > 32 ? (nwt == 4 ? 6 : nwt == 8 ? 4 : 3) : (nwt == 4 ? BASAL_W4NG : nwt == 8 ? BASAL_W8NG : BASAL_W16NG);  // (64 seed entries: 1.5 KB more LDS per wave)
    if (gap && heavy) return nwt == 4 ? BASAL_W4GH : nwt == 8 ? BASAL_W8GH : 2;
    return nwt == 4 ? (gap ? BASAL_W4G : heavy ? BASAL_W4H : BASAL_W4NG) : nwt == 8 ? (gap ? BASAL_W8G : heavy ? 4 : BASAL_W8NG) : (gap ? BASAL_W16G : heavy ? 3 : BASAL_W16NG);
}

template <int NWT, bool NEWRULE, bool GAP, bool HEAVY, bool PE = false>
__global__ __launch_bounds__(256, waves_per_simd(NWT, GAP, HEAVY, PE)) void align_kernel(DevCtx cx) {
    __shared__ uint8_t s_tab[5 * 256];
    __shared__ WaveLds<NWT, GAP, HEAVY, PE> s_w[4];
#ifdef BASAL_CHECK_FILTER  // the check build: BASAL_POISON=<byte> fills the block's LDS first -- no result may depend on what LDS held at launch
    if (COLD(lds_poison) >> 8) {
        const uint32_t pat = (COLD(lds_poison) & 0xffu) * 0x01010101u;
        for (uint32_t i = threadIdx.x; i < sizeof(s_w) / 4; i += 256) ((uint32_t *)s_w)[i] = pat;
        if (threadIdx.x < 64) { s_anchor[threadIdx.x] = s_rcoff[threadIdx.x] = s_csize[threadIdx.x] = pat; }
        __syncthreads();
    }
#endif
    for (int i = threadIdx.x; i < 5 * 256; i += 256) s_tab[i] = cx.tables[i];
    s_prof[threadIdx.x >> 4][threadIdx.x & 15] = (uint16_t)profile(threadIdx.x >> 4, threadIdx.x & 15, cx.K, cx.I);
    if (threadIdx.x >= 1 && threadIdx.x <= 32) s_rcp[threadIdx.x] = (65536u + threadIdx.x - 1) / threadIdx.x;
    if (threadIdx.x < 64 && threadIdx.x < cx.ncontig && cx.ncontig <= 64) {
        s_anchor[threadIdx.x] = cx.ref_anchor[threadIdx.x];
        s_rcoff[threadIdx.x] = cx.rc_offset[threadIdx.x];
        s_csize[threadIdx.x] = cx.contig_size[threadIdx.x];
    }
    if constexpr (HEAVY) {  // more than 64 contigs: 64 evenly spaced pivots of the anchor table (bulk_add's search starts there)
        const uint32_t ps = (cx.ncontig + 63) / 64;
        if (threadIdx.x < 64 && cx.ncontig > 64) s_anchor[threadIdx.x] = threadIdx.x * ps < cx.ncontig ? cx.ref_anchor[threadIdx.x * ps] : 0xFFFFFFFFu;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    WaveLds<NWT, GAP, HEAVY, PE> &L = s_w[wv];
    if constexpr (PE) if (lane0(lane)) L.stg_n = L.stg_mask = 0;
    if (lane <= NWT) {  // zero the pad words once
        for (int c = 0; c < 2; c++)
            for (int p = 0; p < 3; p++) L.q[c][p][lane] = 0;
    }
    if constexpr (GAP) {
        if (lane < 10) {  // the zero words around the bit planes: (chain, plane) = lane, plane 4 = the valid plane
            uint64_t *pl = lane % 5 < 4 ? L.mmp[lane / 5][lane % 5] : L.valp[lane / 5];
            pl[0] = pl[NWT / 2 + 1] = 0;
        }
    }
    wave_sync();
    basal_hit *log = cx.scratch + (size_t)(blockIdx.x * 4 + wv) * COLD(scratch_per_wave);
#ifdef BASAL_PHASE_TIMING
    PhaseClock phc;
    for (int i = 0; i < PH_N; i++) phc.acc[i] = 0;
    phc.n_chunks = phc.n_alive = phc.n_hits = phc.n_bigchunks = phc.n_bigalive = 0;
    phc.last = __builtin_readcyclecounter();
#endif
    // Work queue: a wave takes WORK_CHUNK consecutive reads per atomic. One atomic per read would cap the
    // whole GPU at the rate a single memory word can be incremented (~88 M/s measured on MI355X; the
    // kernel ran at exactly that ceiling, independent of occupancy, before reads were taken in chunks).
    // Every wave leaves this loop: the queue head only grows, and the iteration bound is a watchdog
    // against an internal error (a wave cannot legitimately take more than n reads).
    // (the pipeline's extras -- a list of read numbers, its length in device memory -- are re-read from the kernel arguments where they
    // are used, once per chunk of reads: held in registers across the work loop they cost the standard launch spills)
#define LISTED (COLD(order) != nullptr)
    const bool no_reads = COLD(n_ptr) && *COLDP(const uint32_t, n_ptr) == 0;  // (a read-length class without reads: no queue traffic)
    for (uint32_t iter = 0; !no_reads; iter++) {
        uint32_t n_items = cx.n;
        if (COLD(n_ptr)) { n_items = *COLDP(const uint32_t, n_ptr); n_items = n_items < cx.n ? n_items : cx.n; }  // cx.n: the capacity of the list
        // the whole wave must arrive here together (see lane0()); a partial wave is an internal error
        if (ballot(1) != ~0ULL) { guard_idx(cx, G_WATCHDOG, 0x20000u | (uint32_t)__popcll(ballot(1)), 0, iter); break; }
        uint32_t base = 0;
        if (lane0(lane)) base = atomicAdd(cx.work_counter, (unsigned int)WORK_CHUNK);
        base = rfl(base);
        PH(PH_QUEUE);
        if (base >= n_items) break;
        if (iter > n_items) { guard_idx(cx, G_WATCHDOG, iter, 0, base); break; }
        const uint32_t end = base + WORK_CHUNK < n_items ? base + WORK_CHUNK : n_items;
        // The chunk's descriptors come in with one load, and each read's bytes are requested while the read before it
        // is being aligned, so a read starts on data that is already in registers (2 memory round trips per chunk
        // instead of 2 per read).
        if ((uint32_t)lane < end - base) {
            const uint32_t rn = LISTED ? COLDP(const uint32_t, order)[base + lane] : base + (uint32_t)lane;
            L.rno[lane] = rn;
            L.desc[lane] = cx.reads[rn];
        }
        wave_sync();
        basal_read nrd = uniform_read(L.desc[0]);
        uint32_t npre[NWT / 2];
        int npc = (chain_flags(cx, nrd.readset & 0x7fu) & 1u) ? 0 : 1;
        load_bases<NWT>(cx, nrd, base, npc, lane, npre);
        PH(PH_CHUNK);
        for (uint32_t w = base; w < end; w++) {
            const uint32_t r = LISTED ? rfl(L.rno[w - base]) : w;  // the read's number in the batch
            if (ballot(1) != ~0ULL) { guard_idx(cx, G_WATCHDOG, 0x30000u | (uint32_t)__popcll(ballot(1)), 0, r); break; }
            const basal_read rd = nrd;
            const int pc = npc;
            uint32_t pre[NWT / 2];
#pragma unroll
            for (int b = 0; b < NWT / 2; b++) pre[b] = npre[b];
            PH(PH_E1);
            if (w + 1 < end) {
                nrd = uniform_read(L.desc[w + 1 - base]);
                npc = (chain_flags(cx, nrd.readset & 0x7fu) & 1u) ? 0 : 1;
                PH(PH_E2);
                load_bases<NWT>(cx, nrd, r, npc, lane, npre);
            }
            PH(PH_E3);
#ifdef BASAL_PHASE_TIMING
            const uint64_t t_read0 = __builtin_readcyclecounter();
            uint64_t snap[PH_N];
            for (int i = 0; i < PH_N; i++) snap[i] = phc.acc[i];
#endif
            process_read<NWT, NEWRULE, GAP, HEAVY, PE>(cx, L, s_tab, log, r, w - base, rd, pre, pc, lane PH_ARG);
#ifdef BASAL_PHASE_TIMING
            {   // histogram of per-read wave-clocks by power of two (diagnostic build)
                const uint64_t dtc = __builtin_readcyclecounter() - t_read0;
                const int bucket = 63 - __builtin_clzll(dtc | 1);
                if (lane0(lane)) atomicAdd((unsigned long long *)(COLDP(unsigned int, guard) + 31) + PH_N + 4 + (bucket < 31 ? bucket : 31), 1ull);
                if (bucket >= 21 && lane0(lane))  // the phase split of the long reads only
                    for (int i = 0; i < PH_N; i++) atomicAdd((unsigned long long *)(COLDP(unsigned int, guard) + 31) + PH_N + 4 + 32 + i, (unsigned long long)(phc.acc[i] - snap[i]));
            }
#endif
        }
        // the chunk's results leave in one coalesced store (a store per read would have every read wait for the
        // previous read's write acknowledgement at its first memory wait)
        wave_sync();
        if constexpr (PE) if (COLD(stream_mode) != BASAL_STREAM_NONE) stream_flush(cx, L, log, lane);  // the chunk's staged hit-stream records take their place (and the results learn it)
        static_assert(WORK_CHUNK * sizeof(basal_result) == 64 * sizeof(uint32_t), "one dword per lane");
        if ((uint32_t)lane < (end - base) * (uint32_t)(sizeof(basal_result) / 4)) {
            if (!LISTED) ((uint32_t *)(COLDP(basal_result, results) + base))[lane] = ((const uint32_t *)L.res)[lane];
            else ((uint32_t *)(COLDP(basal_result, results) + L.rno[lane >> 3]))[lane & 7] = ((const uint32_t *)L.res)[lane];  // still one store instruction per chunk
        }
    }
    // The last wave out leaves the queue head at zero for the next launch on this counter block (word 26 counts the waves that are through): a
    // memset in front of every launch queued behind the copy of the previous launch's results when both went through the same engine.
    if (lane0(lane)) {
        unsigned int *wc = cx.work_counter;
        if (atomicAdd(&wc[26], 1u) == gridDim.x * 4u - 1u) { atomicExch(&wc[26], 0u); atomicExch(&wc[0], 0u); }
    }
#ifdef BASAL_PHASE_TIMING
    if (lane0(lane)) {  // when this wave found the queue empty (100 MHz clock): latest, earliest (as the complement), sum, count -- the launch's tail
        const unsigned long long tx = __builtin_amdgcn_s_memrealtime();
        unsigned long long *ex = (unsigned long long *)(COLDP(unsigned int, guard) + 31) + PH_N + 4 + 32 + PH_N + 8;
        atomicMax(&ex[0], tx); atomicMax(&ex[1], ~tx); atomicAdd(&ex[2], tx & 0xFFFFFFFFFFull); atomicAdd(&ex[3], 1ull);
    }
    if (lane0(lane))
    {
        for (int i = 0; i < PH_N; i++) atomicAdd((unsigned long long *)(COLDP(unsigned int, guard) + 31) + i, (unsigned long long)phc.acc[i]);
        unsigned long long *cn = (unsigned long long *)(COLDP(unsigned int, guard) + 31) + PH_N;
        atomicAdd(cn + 0, (unsigned long long)phc.n_chunks); atomicAdd(cn + 1, (unsigned long long)phc.n_alive);
        atomicAdd(cn + 2, (unsigned long long)phc.n_bigchunks); atomicAdd(cn + 3, (unsigned long long)phc.n_bigalive);
    }
#endif
}

typedef void (*kernel_fn)(DevCtx);
template <int NWT>
kernel_fn pick_kernel(bool newrule, bool gap, bool heavy = false, bool pe = false) {
    if (pe && !gap && !heavy) return newrule ? align_kernel<NWT, true, false, false, true> : align_kernel<NWT, false, false, false, true>;
    if (heavy && gap) return newrule ? align_kernel<NWT, true, true, true> : align_kernel<NWT, false, true, true>;
    if (heavy) return newrule ? align_kernel<NWT, true, false, true> : align_kernel<NWT, false, false, true>;
    if (newrule) return gap ? align_kernel<NWT, true, true, false> : align_kernel<NWT, true, false, false>;
    return gap ? align_kernel<NWT, false, true, false> : align_kernel<NWT, false, false, false>;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host side of the core

extern "C" const char *basal_last_error(void) { return g_err.c_str(); }
namespace basal { void set_error(const std::string &s) { g_err = s; } }

// Instrumentation: for every align_kernel instantiation one text line "nwt newrule gap heavy pe assumed fit slack": the blocks per CU its launch bounds
// ask for (waves_per_simd: 256-thread blocks, one wave per SIMD each), the blocks that fit (registers by the runtime's occupancy answer, LDS by the
// chip's allocation unit), and how many more bytes of LDS a block could take before it loses one. Returns the number of kernels, or a
// negative error. (tests/test_gpu_parity.py pins the `fit` column: a kernel that silently loses a block per CU loses its share of the waves.)
namespace {
// LDS is handed out in units of 1 280 B on this chip (128 units per CU; tools/microbench_lds_alloc.hip: a block of 32 000 B is resident five times per
// CU, one of 32 001 B four times) -- which the occupancy API does not know (it answers 5 up to 32 768 B).
constexpr size_t LDS_UNIT = 1280, LDS_UNITS_PER_CU = 128;
int blocks_that_fit(kernel_fn k, size_t dyn_lds) {
    int fit = 0;
    hipFuncAttributes a;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, k, 256, dyn_lds) != hipSuccess || hipFuncGetAttributes(&a, (const void *)k) != hipSuccess) return -1;
    const size_t units = (a.sharedSizeBytes + dyn_lds + LDS_UNIT - 1) / LDS_UNIT;
    if (units && (int)(LDS_UNITS_PER_CU / units) < fit) fit = (int)(LDS_UNITS_PER_CU / units);
    return fit;
}
}  // namespace
extern "C" int basal_core_occupancy_report(char *text, size_t cap) {
    std::string out;
    int n = 0;
    for (int nwt : {4, 8, 16})
        for (int v = 0; v < 10; v++) {
            const bool nr = v & 1, gp = (v >> 1) == 1 || (v >> 1) == 3, hv = (v >> 1) >= 2 && (v >> 1) <= 3, pe = (v >> 1) == 4;
            kernel_fn k = nwt == 4 ? pick_kernel<4>(nr, gp, hv, pe) : nwt == 8 ? pick_kernel<8>(nr, gp, hv, pe) : pick_kernel<16>(nr, gp, hv, pe);
            const int fit = blocks_that_fit(k, 0);
            if (fit < 0) { g_err = "hipOccupancyMaxActiveBlocksPerMultiprocessor failed"; return BASAL_EDEVICE; }
            size_t lo = 0, hi = 65536;  // the largest dynamic LDS size that keeps `fit` blocks
            while (lo < hi) {
                const size_t mid = (lo + hi + 1) / 2;
                if (blocks_that_fit(k, mid) >= fit) lo = mid; else hi = mid - 1;
            }
            char line[96];
            snprintf(line, sizeof line, "%d %d %d %d %d %d %d %zu\n", nwt, (int)nr, (int)gp, (int)hv, (int)pe, waves_per_simd(nwt, gp, hv, pe), fit, lo);
            out += line;
            n++;
        }
    if (text && cap) snprintf(text, cap, "%s", out.c_str());
    return n;
}

// Where a buffer lies in HBM is worth up to 10 % of the align kernel's time on a repeat-rich index (steady for a given placement, different
// from one allocation to the next). The entry points below move the core's long-lived buffers to freshly allocated memory, so that a host with a
// representative batch can time a launch on two placements and keep the better one (bench.py does). Results do not depend on it. No launch of this core
// may be in flight. basal_core_move_buffers: one class of buffers -- new allocation, device copy, the old one freed (tools/probe_placement.py).
namespace {
int move_buffer(void **pp, bool *moved) {
    *moved = false;
    if (!*pp) return BASAL_OK;
    size_t n = 0;
    HIP_TRY(hipMemPtrGetInfo(*pp, &n));
    void *q = nullptr;
    if (hipMalloc(&q, n) != hipSuccess) { (void)hipGetLastError(); return BASAL_OK; }  // no room for a second copy: it stays where it is
    HIP_TRY(hipMemcpy(q, *pp, n, hipMemcpyDeviceToDevice));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipFree(*pp));
    *pp = q;
    *moved = true;
    return BASAL_OK;
}
}  // namespace
// which: 0 locs, 1 flank words, 2 seed words, 3 the two k-mer tables, 4 the reference strands (and their bit planes), 5 the per-wave hit logs. Returns the
// number of buffers moved (one that does not fit twice stays), or a negative error.
extern "C" int basal_core_move_buffers(basal_core_t *c, int which) {
    if (!c) { g_err = "null argument"; return BASAL_EINVAL; }
    if (hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { g_err = "basal_core_move_buffers: device"; return BASAL_EDEVICE; }
    void **cls[6][4] = {{(void **)&c->d_locs}, {(void **)&c->d_flank_a}, {(void **)&c->d_seedw}, {(void **)&c->d_koff, (void **)&c->d_knfwd},
                        {(void **)&c->d_xref[0], (void **)&c->d_xref[1], (void **)&c->d_xpl[0], (void **)&c->d_xpl[1]}, {(void **)&c->d_scratch}};
    if (which < 0 || which > 5) { g_err = "basal_core_move_buffers: which = 0..5"; return BASAL_EINVAL; }
    int n = 0;
    for (void **pp : cls[which]) {
        if (!pp) continue;
        bool moved = false;
        const int rc = move_buffer(pp, &moved);
        if (rc) return rc;
        n += moved;
    }
    if (which == 1) c->d_flank_b = c->d_flank_a ? c->d_flank_a + c->nlocs + 64 : nullptr;
    return n;
}
// A second placement kept beside the first, so that a host can time both and keep the better one:
//   fork   -- every long-lived buffer copied into freshly allocated memory; the core now runs on the copies, the originals are kept aside.
//             Returns the number of buffers, or 0 (nothing changed) when HBM has no room for a second set.
//   swap   -- the core goes back to the set kept aside (and that one becomes the current one).
//   commit -- the set kept aside is freed.
namespace {
constexpr int kPlaced = 10;
void placed_slots(basal_core *c, void **slot[kPlaced]) {
    void **s[kPlaced] = {(void **)&c->d_flank_a, (void **)&c->d_locs,    (void **)&c->d_seedw,  (void **)&c->d_koff,   (void **)&c->d_knfwd,
                         (void **)&c->d_xref[0], (void **)&c->d_xref[1], (void **)&c->d_xpl[0], (void **)&c->d_xpl[1], (void **)&c->d_scratch};
    for (int i = 0; i < kPlaced; i++) slot[i] = s[i];
}
}  // namespace
extern "C" int basal_core_placement_fork(basal_core_t *c) {
    if (!c) { g_err = "null argument"; return BASAL_EINVAL; }
    if (c->alt_valid) { g_err = "basal_core_placement_fork: a second placement is already kept (commit first)"; return BASAL_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    if (int rc = basal_ensure_launch_geometry(c)) return rc;  // (the hit logs exist from here on: both sets have theirs)
    void **slot[kPlaced];
    placed_slots(c, slot);
    void *copy[kPlaced] = {};
    int n = 0;
    for (int i = 0; i < kPlaced; i++) {
        if (!*slot[i]) continue;
        size_t bytes = 0;
        HIP_TRY(hipMemPtrGetInfo(*slot[i], &bytes));
        if (hipMalloc(&copy[i], bytes) != hipSuccess) {  // no room for a second set: leave everything as it was
            (void)hipGetLastError();
            for (int j = 0; j < i; j++) hipFree(copy[j]);
            return 0;
        }
        HIP_TRY(hipMemcpyAsync(copy[i], *slot[i], bytes, hipMemcpyDeviceToDevice, nullptr));
        n++;
    }
    HIP_TRY(hipDeviceSynchronize());
    for (int i = 0; i < kPlaced; i++) { c->alt[i] = *slot[i]; *slot[i] = copy[i]; }
    c->d_flank_b = c->d_flank_a ? c->d_flank_a + c->nlocs + 64 : nullptr;
    c->alt_valid = true;
    return n;
}
extern "C" int basal_core_placement_swap(basal_core_t *c) {
    if (!c || !c->alt_valid) { g_err = "basal_core_placement_swap: no second placement is kept"; return BASAL_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    void **slot[kPlaced];
    placed_slots(c, slot);
    for (int i = 0; i < kPlaced; i++) std::swap(c->alt[i], *slot[i]);
    c->d_flank_b = c->d_flank_a ? c->d_flank_a + c->nlocs + 64 : nullptr;
    return BASAL_OK;
}
extern "C" int basal_core_placement_commit(basal_core_t *c) {
    if (!c) { g_err = "null argument"; return BASAL_EINVAL; }
    if (!c->alt_valid) return BASAL_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    for (int i = 0; i < kPlaced; i++) { hipFree(c->alt[i]); c->alt[i] = nullptr; }
    c->alt_valid = false;
    return BASAL_OK;
}

static uint32_t pow3(uint32_t k) {
    uint32_t t = 1;
    for (uint32_t i = 0; i < k; i++) t *= 3;
    return t;
}

extern "C" int basal_core_create(const basal_params *p, int device, basal_core_t **out) {
    if (!p || !out) { g_err = "null argument"; return BASAL_EINVAL; }
    if (p->seed_size < 10 || p->seed_size > 16 || p->index_interval < 1 || p->index_interval > 16 || p->gap > BASAL_MAXGAPS ||
        p->max_num_hits < 1 || p->max_num_hits > BASAL_MAXHITS || p->chains > 2) {
        g_err = "parameter out of range (seed_size 10..16, index_interval 1..16, gap<=3, 1<=max_num_hits<=1000, chains 0..2)";
        return BASAL_EINVAL;
    }
    for (int i = 0; i < 256; i++)  // (the stream tests read "a valid base" as one bit per base)
        if (p->reg_alphabet[i] != 0 && p->reg_alphabet[i] != 3) { g_err = "reg_alphabet: every entry must be 0 or 3 (param.cpp:130-139)"; return BASAL_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "no HIP device: the BASAL core needs an MI355X (gfx950); there is no CPU fallback"; return BASAL_EDEVICE; }
    if (device < 0 || device >= ndev) { g_err = "bad device ordinal"; return BASAL_EINVAL; }
    basal_core *c = new basal_core();
    c->p = *p;
    c->device = device;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipGetDeviceProperties(&c->prop, device));
    HIP_TRY(hipMalloc(&c->d_tables, 5 * 256));
    uint8_t tabs[5 * 256];
    memcpy(tabs, p->alphabet, 256);
    memcpy(tabs + 256, p->rev_alphabet, 256);
    memcpy(tabs + 512, p->reg_alphabet, 256);
    memcpy(tabs + 768, p->alphabet_mread, 256);
    memcpy(tabs + 1024, p->rev_alphabet_mread, 256);
    HIP_TRY(hipMemcpy(c->d_tables, tabs, sizeof tabs, hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&c->d_counter, 256 * sizeof(unsigned int)));  // [0] queue head, [1..24] guard ledger, [32..] phase clocks (diagnostic build)
    HIP_TRY(hipMemset(c->d_counter, 0, 256 * sizeof(unsigned int)));
    HIP_TRY(hipMalloc(&c->lane0.d_used, sizeof(unsigned long long)));
    HIP_TRY(hipStreamCreate(&c->lane0.stream));
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    c->total_kmers = pow3(p->seed_size);
    c->scratch_per_wave = 16 * p->max_num_hits < 64 ? 64 : 16 * p->max_num_hits;  // (at least 64: the first 64 records of a wave's area stage its hit-stream records, stream_flush)
    *out = c;
    return BASAL_OK;
}

extern "C" void basal_core_destroy(basal_core_t *c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipFree(c->d_xref[0]); hipFree(c->d_xref[1]); hipFree(c->d_xpl[0]); hipFree(c->d_xpl[1]); hipFree(c->d_anchor); hipFree(c->d_size); hipFree(c->d_rcoff);
    hipFree(c->d_koff); hipFree(c->d_knfwd); hipFree(c->d_locs); hipFree(c->d_flank_a); hipFree(c->d_seedw); hipFree(c->d_tables); hipFree(c->d_scratch);
    hipFree(c->d_names); hipFree(c->d_name_off);
    if (c->alt_valid) for (void *q : c->alt) hipFree(q);  // (a second placement still kept aside)
    hipFree(c->d_pe_pairs); hipFree(c->d_pe_recs); hipFree(c->d_pe_work); hipFree(c->d_pe_misc);
    hipFree(c->d_counter);
    c->more_lanes.push_back(&c->lane0);
    for (CoreLane *L : c->more_lanes) {
        hipFree(L->d_bases); hipFree(L->d_reads); hipFree(L->d_stales); hipFree(L->d_results); hipFree(L->d_stream); hipFree(L->d_used); hipFree(L->d_counter); hipFree(L->d_scratch);
        if (L->stream) hipStreamDestroy(L->stream);
        if (L != &c->lane0) delete L;
    }
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->ev2) hipEventDestroy(c->ev2);
    if (c->ev3) hipEventDestroy(c->ev3);
    delete c;
}

// GAP cores: the staged strands once more as bit planes, one 16-byte block per 64 bases (DevCtx::xpl)
namespace {
__global__ __launch_bounds__(256) void ref_to_planes(const uint64_t *__restrict__ x, unsigned long long nwords_alloc, ulonglong2 *__restrict__ pl, unsigned long long nblk) {
    for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += (unsigned long long)gridDim.x * blockDim.x) {
        uint64_t hi = 0, lo = 0;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint64_t v = 2 * b + k < nwords_alloc ? x[2 * b + k] : 0;  // base j at bits 63-2j (high), 62-2j (low)
            const uint64_t rv = __brevll(v);                                     // base j: high bit at 2j, low bit at 2j+1
            hi |= (uint64_t)even_bits(rv) << (32 * k);
            lo |= (uint64_t)even_bits(rv >> 1) << (32 * k);
        }
        pl[b] = make_ulonglong2(hi, lo);
    }
}
}  // namespace

extern "C" int basal_core_set_reference(basal_core_t *c, const uint64_t *xref_fwd, const uint64_t *xref_rc, uint64_t nwords, const uint32_t *ref_anchor,
                                        const uint32_t *contig_size, const uint32_t *rc_offset, uint32_t ncontig) {
    if (!c || !xref_fwd || !xref_rc || !ref_anchor || !contig_size || !rc_offset || ncontig == 0 || nwords < 2 * BASAL_REF_MARGIN) {
        g_err = "set_reference: bad argument";
        return BASAL_EINVAL;
    }
    if (int rc = basal_core_placement_commit(c)) return rc;  // (a second placement kept aside would go stale)
    if ((uint64_t)ref_anchor[ncontig] != (nwords - BASAL_REF_MARGIN) * 32) {
        g_err = "set_reference: ref_anchor[ncontig] does not match nwords (layout: 400 margin words, contigs, 400 margin words)";
        return BASAL_EINVAL;
    }
    HIP_TRY(hipSetDevice(c->device));
    hipFree(c->d_xref[0]); hipFree(c->d_xref[1]); hipFree(c->d_anchor); hipFree(c->d_size); hipFree(c->d_rcoff);
    c->d_xref[0] = c->d_xref[1] = nullptr; c->d_anchor = c->d_size = c->d_rcoff = nullptr;
    // 64 extra words so a lane may always fetch one word beyond the alignment
    for (int s = 0; s < 2; s++) {
        HIP_TRY(hipMalloc(&c->d_xref[s], (nwords + 64) * 8));
        HIP_TRY(hipMemset(c->d_xref[s], 0, (nwords + 64) * 8));
        HIP_TRY(hipMemcpy(c->d_xref[s], s ? xref_rc : xref_fwd, nwords * 8, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc(&c->d_anchor, (ncontig + 1) * 4));
    HIP_TRY(hipMalloc(&c->d_size, ncontig * 4));
    HIP_TRY(hipMalloc(&c->d_rcoff, ncontig * 4));
    HIP_TRY(hipMemcpy(c->d_anchor, ref_anchor, (ncontig + 1) * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_size, contig_size, ncontig * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_rcoff, rc_offset, ncontig * 4, hipMemcpyHostToDevice));
    hipFree(c->d_xpl[0]); hipFree(c->d_xpl[1]);
    c->d_xpl[0] = c->d_xpl[1] = nullptr;
    if (c->p.gap > 0) {
        const unsigned long long nblk = (nwords + 64) / 2 + 8;
        for (int s = 0; s < 2; s++) {
            HIP_TRY(hipMalloc(&c->d_xpl[s], nblk * sizeof(ulonglong2)));
            hipLaunchKernelGGL(ref_to_planes, dim3((unsigned)std::min<unsigned long long>((nblk + 255) / 256, 65536ull)), dim3(256), 0, 0, c->d_xref[s],
                               (unsigned long long)(nwords + 64), c->d_xpl[s], nblk);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipDeviceSynchronize());
    }
    c->nwords = nwords;
    c->ncontig = ncontig;
    c->have_ref = true;
    return BASAL_OK;
}

extern "C" int basal_core_set_index(basal_core_t *c, const uint32_t *kmer_off, const uint32_t *kmer_nfwd, const uint32_t *locs, uint64_t nlocs,
                                    uint32_t max_kmer_num) {
    if (!c || !kmer_off || !kmer_nfwd || (!locs && nlocs)) { g_err = "set_index: bad argument"; return BASAL_EINVAL; }
    if (nlocs >= 0xFFFFFFFFull) { g_err = "set_index: more than 2^32-1 index entries (the reference's 32-bit counters overflow too)"; return BASAL_EINVAL; }
    if (kmer_off[c->total_kmers] != (uint32_t)nlocs) { g_err = "set_index: kmer_off[3^k] != nlocs"; return BASAL_EINVAL; }
    if (int rc = basal_core_placement_commit(c)) return rc;  // (a second placement kept aside would go stale)
    HIP_TRY(hipSetDevice(c->device));
    hipFree(c->d_koff); hipFree(c->d_knfwd); hipFree(c->d_locs);
    c->d_koff = c->d_knfwd = c->d_locs = nullptr;
    HIP_TRY(hipMalloc(&c->d_koff, ((size_t)c->total_kmers + 1) * 4));
    HIP_TRY(hipMalloc(&c->d_knfwd, (size_t)c->total_kmers * 4));
    HIP_TRY(hipMalloc(&c->d_locs, (nlocs + 64) * 4));
    HIP_TRY(hipMemcpy(c->d_koff, kmer_off, ((size_t)c->total_kmers + 1) * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_knfwd, kmer_nfwd, (size_t)c->total_kmers * 4, hipMemcpyHostToDevice));
    if (nlocs) HIP_TRY(hipMemcpy(c->d_locs, locs, nlocs * 4, hipMemcpyHostToDevice));
    c->nlocs = nlocs;
    c->max_kmer_num = max_kmer_num;
    if (!c->have_ref) { g_err = "set_index: stage the reference first (basal_core_set_reference)"; return BASAL_ESTATE; }
    if (int rc = basal_build_flanks(c, nullptr)) return rc;
    c->have_index = true;
    return BASAL_OK;
}

extern "C" int basal_core_get_index(basal_core_t *c, uint32_t *kmer_off, uint32_t *kmer_nfwd, uint32_t *locs, uint64_t *nlocs, uint32_t *max_kmer_num) {
    if (!c || !c->have_index) { g_err = "get_index: no index"; return BASAL_ESTATE; }
    HIP_TRY(hipSetDevice(c->device));
    if (kmer_off) HIP_TRY(hipMemcpy(kmer_off, c->d_koff, ((size_t)c->total_kmers + 1) * 4, hipMemcpyDeviceToHost));
    if (kmer_nfwd) HIP_TRY(hipMemcpy(kmer_nfwd, c->d_knfwd, (size_t)c->total_kmers * 4, hipMemcpyDeviceToHost));
    if (locs && c->nlocs) HIP_TRY(hipMemcpy(locs, c->d_locs, c->nlocs * 4, hipMemcpyDeviceToHost));
    if (nlocs) *nlocs = c->nlocs;
    if (max_kmer_num) *max_kmer_num = c->max_kmer_num;
    return BASAL_OK;
}

extern "C" int basal_core_set_timing(basal_core_t *c, int on) {
    if (!c) return BASAL_EINVAL;
    c->timing = on != 0;
    return BASAL_OK;
}

extern "C" float basal_core_last_kernel_ms(basal_core_t *c) {
    if (!c || !c->timed) return 0.f;
    float ms = 0.f;
    if (hipEventSynchronize(c->ev1) != hipSuccess) return 0.f;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return 0.f;
    return ms;
}

static int ensure_launch_geometry(basal_core *c) {
    if (c->grid) return BASAL_OK;
    uint32_t cus = (uint32_t)c->prop.multiProcessorCount;
    c->grid = cus * 8;  // scratch for the largest grid any instantiation uses (8 blocks of 4 waves per CU)
    HIP_TRY(hipMalloc(&c->d_scratch, (size_t)c->grid * 4 * c->scratch_per_wave * sizeof(basal_hit)));
    return BASAL_OK;
}

extern "C" float basal_core_last_pair_ms(basal_core_t *c) {
    if (!c || !c->pair_timed) return 0.f;
    float ms = 0.f;
    if (hipEventSynchronize(c->ev3) != hipSuccess || hipEventElapsedTime(&ms, c->ev2, c->ev3) != hipSuccess) return 0.f;
    return ms;
}

extern "C" int basal_core_launch_info(basal_core_t *c, uint32_t *blocks, uint32_t *threads, uint32_t *lds_bytes) {
    if (!c) return BASAL_EINVAL;
    int rc = ensure_launch_geometry(c);
    if (rc) return rc;
    if (blocks) *blocks = c->last_grid ? c->last_grid : c->grid;
    if (threads) *threads = 256;
    if (lds_bytes) {
        int nwt = c->nwt ? c->nwt : 4;
        bool nr = c->p.new_rule != 0, gp = c->p.gap > 0;
        const bool hv = c->heavy;
        const bool pe = c->p.pairend != 0;
        kernel_fn k = nwt == 4 ? pick_kernel<4>(nr, gp, hv, pe) : nwt == 8 ? pick_kernel<8>(nr, gp, hv, pe) : pick_kernel<16>(nr, gp, hv, pe);
        hipFuncAttributes at;
        HIP_TRY(hipFuncGetAttributes(&at, (const void *)k));
        *lds_bytes = (uint32_t)at.sharedSizeBytes;
    }
    return BASAL_OK;
}

static int report_guard(const unsigned int *guard) {
    static const char *kind[8] = {"k-mer id", "location-list index", "reference word", "base offset", "seed slot", "stale overlay k-mer",
                                  "k-mer id of a mode seed (value = pos|seg<<16|chain<<24|mode<<26)", "watchdog (a loop did not terminate, or a partial wave)"};
    for (int k = 0; k < 8; k++)
        if (guard[k]) {
            g_err = std::string("align: internal bounds violation (") + kind[k] + "): " + std::to_string(guard[k]) + " times, first value " +
                    std::to_string(guard[8 + k]) + " at read " + std::to_string(guard[16 + k]);
            return BASAL_EDEVICE;
        }
    return BASAL_OK;
}

// max_len: the longest read of the batch, selects the kernel instantiation
static int launch_align(basal_core *c, const void *d_bases, uint64_t nbases_dev, const void *d_reads, uint32_t n, const void *d_stales, uint32_t nstale, uint32_t max_len,
                        int stream_mode, void *d_results,
                        void *d_stream, uint64_t stream_cap, void *d_stream_used, const uint8_t carry[2][2], hipStream_t s, const basal_align_extra *ex = nullptr) {
    if (!c->have_ref || !c->have_index) { g_err = "align: reference/index not staged (call set_reference and set_index/build_index first)"; return BASAL_ESTATE; }
    if (stream_mode < 0 || stream_mode > 2) { g_err = "align: bad stream_mode"; return BASAL_EINVAL; }
    if (stream_mode != BASAL_STREAM_NONE && (!d_stream || !d_stream_used)) { g_err = "align: stream buffers required for this stream_mode"; return BASAL_EINVAL; }
    int rc = ensure_launch_geometry(c);
    if (rc) return rc;
    if (n == 0) return BASAL_OK;
    DevCtx cx;
    memset(&cx, 0, sizeof cx);
    cx.xref[0] = c->d_xref[0]; cx.xref[1] = c->d_xref[1];
    cx.ref_anchor = c->d_anchor; cx.contig_size = c->d_size; cx.rc_offset = c->d_rcoff; cx.ncontig = c->ncontig;
    cx.kmer_off = c->d_koff; cx.kmer_nfwd = c->d_knfwd; cx.locs = c->d_locs; cx.max_kmer_num = c->max_kmer_num;
    {   // BASAL_SECOND_WINDOW="<min stream length>:<min cut-off>" overrides the rule (tests run every fixture through the second window with "1:0")
        uint32_t min_T = 1024, min_cut = 32768;
        if (const char *e = getenv("BASAL_SECOND_WINDOW")) {
            unsigned a = 0, b = 0;
            if (sscanf(e, "%u:%u", &a, &b) == 2) { min_T = a; min_cut = b; }
        }
        cx.win2_min_T = c->max_kmer_num >= min_cut ? min_T : 0xFFFFFFFFu;
    }
    cx.flank_a = c->d_flank_a; cx.flank_b = c->d_flank_b; cx.seedw = c->d_seedw;
    cx.xpl[0] = c->d_xpl[0]; cx.xpl[1] = c->d_xpl[1];
    {
        static const char *pz = getenv("BASAL_POISON");
        cx.lds_poison = pz ? 0x100u | (uint32_t)(strtol(pz, nullptr, 0) & 0xff) : 0;
    }
    cx.K = c->p.seed_size; cx.I = c->p.index_interval; cx.max_num_hits = c->p.max_num_hits; cx.chains = c->p.chains;
    cx.randseed = c->p.randseed; cx.gap = c->p.gap; cx.gap_edge = c->p.gap_edge; cx.n_mis = c->p.n_mis;
    cx.stream_mode = (uint32_t)stream_mode; cx.report_repeat_hits = c->p.report_repeat_hits;
    cx.tables = c->d_tables; cx.bases = (const uint8_t *)d_bases; cx.reads = (const basal_read *)d_reads; cx.n = n;
    cx.stales = (const basal_stale *)d_stales; cx.nstale = d_stales ? nstale : 0;
    cx.results = (basal_result *)d_results; cx.stream = (basal_hit *)d_stream; cx.stream_cap = stream_cap;
    cx.stream_used = (unsigned long long *)d_stream_used;
    cx.scratch = c->d_scratch; cx.scratch_per_wave = c->scratch_per_wave; cx.work_counter = c->d_counter;
    cx.ghost_base = 0xFFFFFFF0u;  // no descriptor number gets there
    unsigned int *counter = c->d_counter;
    if (ex) {  // a pipeline slot: its own queue head, ledger and hit-log scratch (launches of different slots may overlap)
        cx.order = ex->order; cx.n_ptr = ex->n_ptr; cx.ghost_base = ex->ghost_base;
        if (ex->counter) { counter = ex->counter; cx.work_counter = counter; }
        if (ex->scratch) cx.scratch = ex->scratch;
    }
    memcpy(cx.carry, carry, 4);
    int nwt = max_len <= 128 ? 4 : max_len <= 256 ? 8 : 16;
    c->nwt = nwt;
    bool nr = c->p.new_rule != 0, gp = c->p.gap > 0;
    const bool hv = c->heavy;
    {
        const char *e = getenv("BASAL_HEAVY_M");  // (tests: 1 sends every list through the long-list loop)
        cx.heavy_m = e ? (uint32_t)atoi(e) : 128u;
        if (cx.heavy_m < 1) cx.heavy_m = 1;
    }
    // (a paired-end core's standard kernels are the PE instantiations: mode groups for the mates that run every mode)
    static const bool pe_off = getenv("BASAL_PE") && atoi(getenv("BASAL_PE")) == 0;  // (BASAL_PE=0: the standard kernels for paired-end cores too; A/B and tests)
    const bool pe = c->p.pairend != 0 && !gp && !hv && !pe_off;
    kernel_fn k = nwt == 4 ? pick_kernel<4>(nr, gp, hv, pe) : nwt == 8 ? pick_kernel<8>(nr, gp, hv, pe) : pick_kernel<16>(nr, gp, hv, pe);
    {   // the queue head is zero: the counter block was allocated so, and the last wave of a launch leaves it so (BASAL_HEAD_MEMSET=1: a memset as well)
        static const bool ms = getenv("BASAL_HEAD_MEMSET") && atoi(getenv("BASAL_HEAD_MEMSET")) != 0;
        if (ms) HIP_TRY(hipMemsetAsync(counter, 0, sizeof(unsigned int), s));
    }
    cx.guard = counter + 1;
    cx.total_kmers = c->total_kmers; cx.nlocs = (uint32_t)c->nlocs; cx.nwords = c->nwords + 64; cx.nbases = nbases_dev;
    const char *env = getenv("BASAL_BLOCKS_PER_CU");
    uint32_t per_cu = env ? (uint32_t)atoi(env) : (uint32_t)waves_per_simd(nwt, gp, hv, pe);
    if (!env) {  // (no more blocks than are resident at once: the kernels whose registers or LDS fit fewer than their launch bounds ask for)
        static std::atomic<int> fit_of[3][2][2][2][2];  // 0 = not asked yet (every GPU of a node answers the same)
        std::atomic<int> &slot = fit_of[nwt == 4 ? 0 : nwt == 8 ? 1 : 2][nr][gp][hv][pe];
        int fit = slot.load(std::memory_order_relaxed);
        if (!fit) { fit = blocks_that_fit(k, 0); slot.store(fit, std::memory_order_relaxed); }
        if (fit > 0 && (uint32_t)fit < per_cu) per_cu = (uint32_t)fit;
    }
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    uint32_t grid = (uint32_t)c->prop.multiProcessorCount * per_cu;
    c->last_grid = grid;
    uint32_t need = (n + 4 * WORK_CHUNK - 1) / (4 * WORK_CHUNK);
    if (grid > need) grid = need;
    if (c->timing && !ex) HIP_TRY(hipEventRecord(c->ev0, s));
    static const bool dbg = getenv("BASAL_DEBUG") != nullptr;
    if (dbg) {
        HIP_TRY(hipStreamSynchronize(s));
        fprintf(stderr, "[basal debug] launching align kernel NWT=%d newrule=%d gap=%d heavy=%d grid=%u n=%u nstale=%u max_kmer_num=%u nlocs=%llu\n", nwt, (int)nr,
                (int)gp, (int)hv, grid, n, cx.nstale, cx.max_kmer_num, (unsigned long long)c->nlocs);
        fprintf(stderr, "[basal debug] buffers: xref %p %p koff %p knfwd %p locs %p flank_a %p flank_b %p seedw %p scratch %p counter %p results %p bases %p\n", (void *)c->d_xref[0],
                (void *)c->d_xref[1], (void *)c->d_koff, (void *)c->d_knfwd, (void *)c->d_locs, (void *)c->d_flank_a, (void *)c->d_flank_b, (void *)c->d_seedw, (void *)cx.scratch,
                (void *)counter, d_results, d_bases);
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, s, cx);
    HIP_TRY(hipGetLastError());
    if (dbg) {
        HIP_TRY(hipStreamSynchronize(s));
        fprintf(stderr, "[basal debug] align kernel finished\n");
    }
    if (c->timing && !ex) { HIP_TRY(hipEventRecord(c->ev1, s)); c->timed = true; }
    if (!ex) c->last_stream = s;
    return BASAL_OK;
}

// A few words set to a value by a kernel on the caller's stream. (hipMemsetAsync in front of a launch queued behind a device-to-host copy that
// another stream had in flight -- 4-7 ms per 10 M-read step of bench.py -- where a kernel does not.)
namespace { __global__ void fill_words(uint32_t *p, uint32_t n, uint32_t v) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; } }
int basal_fill_async(void *p, size_t bytes, uint32_t value, hipStream_t s) {
    const uint32_t n = (uint32_t)(bytes / 4);
    if (!n) return BASAL_OK;
    hipLaunchKernelGGL(fill_words, dim3((n + 255) / 256), dim3(256), 0, s, (uint32_t *)p, n, value);
    HIP_TRY(hipGetLastError());
    return BASAL_OK;
}

int basal_launch_align(basal_core *c, const void *d_bases, uint64_t nbases_dev, const void *d_reads, uint32_t n, const void *d_stales, uint32_t nstale,
                       uint32_t max_len, int stream_mode, void *d_results, void *d_stream, uint64_t stream_cap, void *d_stream_used, hipStream_t s,
                       const basal_align_extra *ex) {
    static const uint8_t zero_carry[2][2] = {{0, 0}, {0, 0}};
    return launch_align(c, d_bases, nbases_dev, d_reads, n, d_stales, nstale, max_len, stream_mode, d_results, d_stream, stream_cap, d_stream_used, zero_carry, s, ex);
}
int basal_launch_align_carry(basal_core *c, const void *d_bases, uint64_t nbases_dev, const void *d_reads, uint32_t n, const void *d_stales, uint32_t nstale,
                             uint32_t max_len, int stream_mode, void *d_results, void *d_stream, uint64_t stream_cap, void *d_stream_used, const uint8_t carry[2][2],
                             hipStream_t s, const basal_align_extra *ex) {
    static const uint8_t zero_carry[2][2] = {{0, 0}, {0, 0}};
    return launch_align(c, d_bases, nbases_dev, d_reads, n, d_stales, nstale, max_len, stream_mode, d_results, d_stream, stream_cap, d_stream_used, carry ? carry : zero_carry, s, ex);
}
int basal_ensure_launch_geometry(basal_core *c) { return ensure_launch_geometry(c); }
int basal_report_guard(const unsigned int *guard) { return report_guard(guard); }

extern "C" int basal_core_sync_check(basal_core_t *c) {
    if (!c) { g_err = "sync_check: null argument"; return BASAL_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    unsigned int guard[24];
    HIP_TRY(hipMemcpyAsync(guard, c->d_counter + 1, sizeof guard, hipMemcpyDeviceToHost, c->last_stream));
    HIP_TRY(hipMemsetAsync(c->d_counter + 1, 0, sizeof guard, c->last_stream));
    HIP_TRY(hipStreamSynchronize(c->last_stream));
#ifdef BASAL_PHASE_TIMING
    {
        static const char *nm[PH_N] = {"queue", "pack", "seeds", "reorder", "mode", "filter", "score", "replay", "final", "chunk", "entry", "bytes", "addhit", "e2", "e3"};
        unsigned long long ph[PH_N + 4 + 32 + PH_N + 8 + 4], tot = 0;
        HIP_TRY(hipMemcpy(ph, c->d_counter + 32, sizeof ph, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(c->d_counter + 32, 0, sizeof ph));
        {
            const unsigned long long *ex = ph + PH_N + 4 + 32 + PH_N + 8;
            if (ex[3]) {
                const double last = (double)ex[0], first = (double)~ex[1], mean = (double)ex[2] / (double)ex[3];
                fprintf(stderr, "[basal wave exits, all launches since the last report] %llu waves; first %.2f ms and mean %.2f ms before the last one (100 MHz clock)\n", ex[3],
                        (last - first) / 1e5, ((double)(ex[0] & 0xFFFFFFFFFFull) - mean) / 1e5);
            }
        }
        fprintf(stderr, "[basal read-time histogram, log2(wave-clocks): count]");
        for (int b = 0; b < 32; b++) if (ph[PH_N + 4 + b]) fprintf(stderr, " %d:%llu", b, ph[PH_N + 4 + b]);
        fprintf(stderr, "\n");
        {
            unsigned long long ht = 0;
            for (int i = 0; i < PH_N; i++) ht += ph[PH_N + 4 + 32 + i];
            fprintf(stderr, "[basal phases of reads >= 2^21 clocks]");
            for (int i = 0; i < PH_N; i++) fprintf(stderr, " %s %.1f%%", nm[i], ht ? 100.0 * ph[PH_N + 4 + 32 + i] / ht : 0.0);
            fprintf(stderr, "  (total %llu)\n", ht);
        }
        fprintf(stderr, "[basal counts] chunks %llu alive %llu | in modes with >= 1024 candidates: chunks %llu alive %llu (HEAVY kernels: long-list chunks, lanes alive after the near windows | 16-lane groups with such a lane, survivors of all windows)\n", ph[PH_N], ph[PH_N + 1], ph[PH_N + 2], ph[PH_N + 3]);
        {
            const unsigned long long *ah = ph + PH_N + 4 + 32 + PH_N;
            fprintf(stderr, "[basal AddHit] calls %llu: off the contig %llu, known (registers) %llu, known (memory log) %llu, new %llu; mean log length at call %.1f, memory scan rounds %llu\n",
                    ah[0], ah[1], ah[2], ah[3], ah[4], ah[0] ? (double)ah[5] / ah[0] : 0.0, ah[6]);
        }
        for (int i = 0; i < PH_N; i++) tot += ph[i];
        fprintf(stderr, "[basal phases]");
        for (int i = 0; i < PH_N; i++) fprintf(stderr, " %s %.1f%%", nm[i], tot ? 100.0 * ph[i] / tot : 0.0);
        fprintf(stderr, "  (total %llu wave-clocks)\n", tot);
    }
#endif
    return report_guard(guard);
}

extern "C" int basal_core_align_batch_device(basal_core_t *c, const void *d_bases, const void *d_reads, uint32_t n, const void *d_stales, uint32_t nstale,
                                             int stream_mode, void *d_results, void *d_stream, uint64_t stream_cap, void *d_stream_used,
                                             const uint8_t carry[2][2], uint32_t max_len, void *hip_stream) {
    if (!c || (n && (!d_bases || !d_reads || !d_results))) { g_err = "align_batch_device: null argument"; return BASAL_EINVAL; }
    if (max_len == 0 || max_len > BASAL_MAXREADLEN) { g_err = "align_batch_device: max_len must be 1..480"; return BASAL_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    static const uint8_t zero_carry[2][2] = {{0, 0}, {0, 0}};
    return launch_align(c, d_bases, ~0ull, d_reads, n, d_stales, nstale, max_len, stream_mode, d_results, d_stream, stream_cap, d_stream_used,
                        carry ? carry : zero_carry, (hipStream_t)hip_stream);
}

// What every host-buffer entry point checks of a batch before anything is queued (one helper: basal_core_align_batch, _align_pairs_batch
// and basal_multi_align_batch must refuse the same descriptors). who: the entry point's name for the message. max_len: the longest read.
int basal_validate_batch(const basal_params &P, const basal_read *reads, uint32_t n, uint64_t nbases, const basal_stale *stales, uint32_t nstale, const char *who,
                         uint32_t *max_len_out) {
    uint32_t max_len = 0;
    const uint32_t K = P.seed_size, I = P.index_interval;
    auto bad = [&](const char *what) { g_err = std::string(who) + ": " + what; return BASAL_EINVAL; };
    // basal_read.seq_off is 32 bits: a batch whose bases do not fit would wrap the offsets and align reads against the wrong bytes
    if (nbases > 0xFFFFFFFFull) return bad("more than 4 GiB of bases in one batch (basal_read.seq_off is 32-bit); split the batch");
    for (uint32_t i = 0; i < n; i++) {
        const basal_read &r = reads[i];
        if (r.len == 0) continue;
        if (r.len > BASAL_MAXREADLEN || (uint64_t)r.seq_off + r.len > nbases) return bad("read descriptor out of range");
        if (r.max_snp > BASAL_MAXSNPS) return bad("max_snp > 15");
        if (r.stale_idx != BASAL_STALE_NONE) {
            if (!stales || r.stale_idx >= nstale) return bad("stale_idx outside the stale table");
            const uint32_t src = stales[r.stale_idx].src;
            if (src != BASAL_STALE_CARRY && (src >= i || reads[src].len < K + I - 1)) return bad("basal_stale.src must name an earlier aligned read");
        }
        if (r.len > max_len) max_len = r.len;
    }
    *max_len_out = max_len;
    return BASAL_OK;
}

template <typename T>
static int grow(T *&p, size_t &cap, size_t need) {
    if (need <= cap) return BASAL_OK;
    hipFree(p);
    p = nullptr;
    size_t nc = need + need / 4 + 1024;
    HIP_TRY(hipMalloc(&p, nc * sizeof(T)));
    cap = nc;
    return BASAL_OK;
}

// A lane for one basal_core_align_batch call: lane 0 if it is free, else one of up to kMaxLanes - 1 more (made on first use: a stream,
// a queue head + ledger, per-wave hit logs), else the call waits for one. The paired-end entry point always takes lane 0.
struct LaneHold {
    basal_core *c;
    CoreLane *L = nullptr;
    explicit LaneHold(basal_core *c_, bool lane0_only = false) : c(c_) {
        std::unique_lock<std::mutex> lk(c->lane_m);
        if (!c->grid && ensure_launch_geometry(c)) return;  // (sizes the per-wave hit logs; under the lock: the first calls may arrive together)
        for (;;) {
            if (!c->lane0.busy) { L = &c->lane0; break; }
            if (!lane0_only) {
                for (CoreLane *x : c->more_lanes) if (!x->busy) { L = x; break; }
                if (L) break;
                static const int max_lanes = getenv("BASAL_CORE_LANES") ? atoi(getenv("BASAL_CORE_LANES")) : kMaxLanes;
                if ((int)c->more_lanes.size() + 1 < max_lanes) {
                    CoreLane *x = new CoreLane();
                    const size_t scratch = (size_t)c->grid * 4 * c->scratch_per_wave * sizeof(basal_hit);
                    if (hipMalloc(&x->d_used, sizeof(unsigned long long)) != hipSuccess || hipStreamCreate(&x->stream) != hipSuccess ||
                        hipMalloc(&x->d_counter, 256 * sizeof(unsigned int)) != hipSuccess || hipMemset(x->d_counter, 0, 256 * sizeof(unsigned int)) != hipSuccess ||
                        hipMalloc(&x->d_scratch, scratch) != hipSuccess) {
                        g_err = "align_batch: cannot set up another lane on this core";
                        hipFree(x->d_used); hipFree(x->d_counter); hipFree(x->d_scratch);
                        if (x->stream) hipStreamDestroy(x->stream);
                        delete x;
                        return;
                    }
                    c->more_lanes.push_back(x);
                    L = x;
                    break;
                }
            }
            c->lane_cv.wait(lk);
        }
        L->busy = true;
    }
    ~LaneHold() {
        if (!L) return;
        std::lock_guard<std::mutex> lk(c->lane_m);
        L->busy = false;
        c->lane_cv.notify_all();
    }
};

extern "C" int basal_core_align_batch(basal_core_t *c, const uint8_t *bases, uint64_t nbases, const basal_read *reads, uint32_t n, const basal_stale *stales,
                                      uint32_t nstale, int stream_mode, basal_result *results, basal_hit *stream, uint64_t stream_cap,
                                      uint64_t *stream_used, uint8_t carry[2][2]) {
    if (!c || (n && (!bases || !reads || !results))) { g_err = "align_batch: null argument"; return BASAL_EINVAL; }
    if (stream_mode != BASAL_STREAM_NONE && (!stream || !stream_used)) { g_err = "align_batch: stream buffers required for this stream_mode"; return BASAL_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    if (stream_used) *stream_used = 0;
    if (n == 0) return BASAL_OK;
    uint32_t max_len = 0;
    if (nbases > 0xFFFFFFFFull) { g_err = "align_batch: more than 4 GiB of bases in one batch (basal_read.seq_off is 32-bit); split the batch"; return BASAL_EINVAL; }
    LaneHold hold(c);  // a free lane (this call may run beside others on the same core); given back when the call returns
    if (!hold.L) return BASAL_EDEVICE;
    CoreLane &L = *hold.L;
    basal_align_extra lane_ex;
    const basal_align_extra *ex = nullptr;
    unsigned int *ledger = c->d_counter;
    if (&L != &c->lane0) { lane_ex.counter = L.d_counter; lane_ex.scratch = L.d_scratch; ex = &lane_ex; ledger = L.d_counter; }
    int rc;
    size_t cap_res = L.cap_reads;
    if ((rc = grow(L.d_bases, L.cap_bases, nbases + 64))) return rc;
    if (n > L.cap_reads) {
        hipFree(L.d_results);
        L.d_results = nullptr;
    }
    if ((rc = grow(L.d_reads, L.cap_reads, n))) return rc;
    if (nstale && (rc = grow(L.d_stales, L.cap_stales, nstale))) return rc;
    if (!L.d_results) HIP_TRY(hipMalloc(&L.d_results, L.cap_reads * sizeof(basal_result)));
    (void)cap_res;
    if (stream_mode != BASAL_STREAM_NONE)
        if ((rc = grow(L.d_stream, L.cap_stream, stream_cap ? stream_cap : 1))) return rc;
    hipStream_t s = L.stream;
    HIP_TRY(hipMemcpyAsync(L.d_bases, bases, nbases, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(L.d_reads, reads, (size_t)n * sizeof(basal_read), hipMemcpyHostToDevice, s));
    if (nstale) HIP_TRY(hipMemcpyAsync(L.d_stales, stales, (size_t)nstale * sizeof(basal_stale), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(L.d_used, 0, sizeof(unsigned long long), s));
    // (the descriptors are checked while the copies run; nothing is launched on a batch that fails the check)
    if (int vrc = basal_validate_batch(c->p, reads, n, nbases, stales, nstale, "align_batch", &max_len)) { hipStreamSynchronize(s); return vrc; }
    static const uint8_t zero_carry[2][2] = {{0, 0}, {0, 0}};
    rc = launch_align(c, L.d_bases, nbases, L.d_reads, n, nstale ? L.d_stales : nullptr, nstale, max_len, stream_mode, L.d_results, L.d_stream, stream_cap, L.d_used,
                      carry ? (const uint8_t(*)[2])carry : zero_carry, s, ex);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(results, L.d_results, (size_t)n * sizeof(basal_result), hipMemcpyDeviceToHost, s));
    unsigned long long used = 0;
    unsigned int guard[24];
    HIP_TRY(hipMemcpyAsync(&used, L.d_used, sizeof used, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(guard, ledger + 1, sizeof guard, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemsetAsync(ledger + 1, 0, sizeof guard, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (int gr = report_guard(guard)) return gr;

    if (stream_used) *stream_used = used;
    int ret = BASAL_OK;
    if (stream_mode != BASAL_STREAM_NONE) {
        if (used > stream_cap) { g_err = "align_batch: hit stream too small; needed " + std::to_string(used); ret = BASAL_EOVERFLOW; }
        uint64_t ncopy = used < stream_cap ? used : stream_cap;
        if (ncopy) HIP_TRY(hipMemcpy(stream, L.d_stream, ncopy * sizeof(basal_hit), hipMemcpyDeviceToHost));
    }
    // carry: xseed_start_offset after the last read of each slot that defined it (align.cpp:475-480)
    if (carry) {
        for (int slot = 0; slot < 2; slot++) {
            for (uint32_t i = n; i-- > 0;) {
                const basal_read &r = reads[i];
                const uint32_t rs = r.readset & 0x7fu;
                if (r.len == 0 || (rs == 2 ? 1 : 0) != slot) continue;
                if (results[i].status == BASAL_READ_SKIPPED) continue;
                // any aligned read leaves a defined value behind (own or inherited)
                bool f0 = (c->p.chains == 1) || ((c->p.chains <= 1) == (rs < 2));
                bool f1 = (c->p.chains == 1) || ((c->p.chains <= 1) == (rs == 2));
                if (f0) carry[slot][0] = results[i].start_off[0];
                if (f1) carry[slot][1] = results[i].start_off[1];
                break;
            }
        }
    }
    return ret;
}


// basal_core_align_batch + the pairing kernel of basal_pe.hip, the hit streams never leaving the device
extern "C" int basal_core_align_pairs_batch(basal_core_t *c, const uint8_t *bases, uint64_t nbases, const basal_read *reads, uint32_t npairs, const basal_stale *stales,
                                            uint32_t nstale, basal_pe_pair *pairs_out, basal_pe_rec *recs_out, uint64_t recs_cap, uint64_t *recs_used, uint32_t stats[9],
                                            uint8_t carry[2][2]) {
    const uint32_t n = 2 * npairs;
    if (!c || (npairs && (!bases || !reads || !pairs_out || !recs_out || !recs_used))) { g_err = "align_pairs_batch: null argument"; return BASAL_EINVAL; }
    if (npairs > 0x7FFFFFFFu / 2) { g_err = "align_pairs_batch: too many pairs in one batch"; return BASAL_EINVAL; }
    *recs_used = 0;
    if (n == 0) return BASAL_OK;
    HIP_TRY(hipSetDevice(c->device));
    uint32_t max_len = 0;
    if (nbases > 0xFFFFFFFFull) { g_err = "align_pairs_batch: more than 4 GiB of bases in one batch (basal_read.seq_off is 32-bit); split the batch"; return BASAL_EINVAL; }
    LaneHold hold(c, true);  // (the pairing buffers are the core's: one paired-end call at a time)
    if (!hold.L) return BASAL_EDEVICE;
    int rc;
    if ((rc = grow(c->lane0.d_bases, c->lane0.cap_bases, nbases + 64))) return rc;
    if (n > c->lane0.cap_reads) { hipFree(c->lane0.d_results); c->lane0.d_results = nullptr; }
    if ((rc = grow(c->lane0.d_reads, c->lane0.cap_reads, n))) return rc;
    if (nstale && (rc = grow(c->lane0.d_stales, c->lane0.cap_stales, nstale))) return rc;
    if (!c->lane0.d_results) HIP_TRY(hipMalloc(&c->lane0.d_results, c->lane0.cap_reads * sizeof(basal_result)));
    if ((rc = grow(c->d_pe_pairs, c->cap_pe_pairs, npairs))) return rc;
    if ((rc = grow(c->d_pe_recs, c->cap_pe_recs, recs_cap ? recs_cap : 1))) return rc;
    if (!c->d_pe_misc) HIP_TRY(hipMalloc(&c->d_pe_misc, 16 * sizeof(unsigned long long)));
    hipStream_t s = c->lane0.stream;
    HIP_TRY(hipMemcpyAsync(c->lane0.d_bases, bases, nbases, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->lane0.d_reads, reads, (size_t)n * sizeof(basal_read), hipMemcpyHostToDevice, s));
    if (nstale) HIP_TRY(hipMemcpyAsync(c->lane0.d_stales, stales, (size_t)nstale * sizeof(basal_stale), hipMemcpyHostToDevice, s));
    // the descriptors are checked while the copies run (page-locked caller buffers: 2 M descriptors take the host as long as their bases take the link);
    // nothing is launched on a batch that fails the check
    if (int vrc = basal_validate_batch(c->p, reads, n, nbases, stales, nstale, "align_pairs_batch", &max_len)) { hipStreamSynchronize(s); return vrc; }
    if (max_len == 0) max_len = 1;
    static const uint8_t zero_carry[2][2] = {{0, 0}, {0, 0}};
    // the hit streams stay on the device: a first guess of their size, doubled until every mate's log fits
    uint64_t stream_cap = c->lane0.cap_stream > (uint64_t)n * 8 + 4096 ? c->lane0.cap_stream : (uint64_t)n * 8 + 4096;
    unsigned long long used = 0;
    for (int attempt = 0;; attempt++) {
        if ((rc = grow(c->lane0.d_stream, c->lane0.cap_stream, (size_t)stream_cap))) return rc;
        if ((rc = grow(c->d_pe_work, c->cap_pe_work, (size_t)stream_cap))) return rc;
        HIP_TRY(hipMemsetAsync(c->lane0.d_used, 0, sizeof(unsigned long long), s));
        rc = launch_align(c, c->lane0.d_bases, nbases, c->lane0.d_reads, n, nstale ? c->lane0.d_stales : nullptr, nstale, max_len, BASAL_STREAM_ALL, c->lane0.d_results, c->lane0.d_stream, stream_cap, c->lane0.d_used,
                          carry ? (const uint8_t(*)[2])carry : zero_carry, s);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(&used, c->lane0.d_used, sizeof used, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (used <= stream_cap) break;
        if (attempt > 3) { g_err = "align_pairs_batch: hit stream keeps overflowing"; return BASAL_EOVERFLOW; }
        stream_cap = used + used / 8 + 4096;
    }
    {
        unsigned int guard[24];
        HIP_TRY(hipMemcpyAsync(guard, c->d_counter + 1, sizeof guard, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemsetAsync(c->d_counter + 1, 0, sizeof guard, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (int gr = report_guard(guard)) return gr;
    }
    HIP_TRY(hipMemsetAsync(c->d_pe_misc, 0, 16 * sizeof(unsigned long long), s));
    if (c->timing) { if (!c->ev2) { HIP_TRY(hipEventCreate(&c->ev2)); HIP_TRY(hipEventCreate(&c->ev3)); } HIP_TRY(hipEventRecord(c->ev2, s)); }
    rc = basal_pe_enqueue(c, c->lane0.d_reads, c->lane0.d_results, c->lane0.d_stream, c->d_pe_work, npairs, c->d_pe_pairs, c->d_pe_recs, recs_cap, c->d_pe_misc, c->d_pe_misc + 1, s);
    if (rc) return rc;
    if (c->timing) { HIP_TRY(hipEventRecord(c->ev3, s)); c->pair_timed = true; }
    unsigned long long misc[16];
    HIP_TRY(hipMemcpyAsync(misc, c->d_pe_misc, sizeof misc, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(pairs_out, c->d_pe_pairs, (size_t)npairs * sizeof(basal_pe_pair), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *recs_used = misc[0];
    if (misc[0] > recs_cap) { g_err = "align_pairs_batch: record array too small; needed " + std::to_string(misc[0]); return BASAL_EOVERFLOW; }
    if (misc[0]) HIP_TRY(hipMemcpy(recs_out, c->d_pe_recs, (size_t)misc[0] * sizeof(basal_pe_rec), hipMemcpyDeviceToHost));
    if (stats) {
        const unsigned int *st = (const unsigned int *)(misc + 1);
        for (int k = 0; k < 9; k++) stats[k] += st[k];
    }
    // carry: the start offset after the last aligned read of each slot (basal_core_align_batch's rule), from the results on the device
    if (carry) {
        std::vector<basal_result> tail;
        for (int slot = 0; slot < 2; slot++)
            for (uint32_t i = n; i-- > 0;) {
                const basal_read &r = reads[i];
                const uint32_t rs = r.readset & 0x7fu;
                if (r.len == 0 || (rs == 2 ? 1 : 0) != slot) continue;
                basal_result one;
                HIP_TRY(hipMemcpy(&one, c->lane0.d_results + i, sizeof one, hipMemcpyDeviceToHost));
                if (one.status == BASAL_READ_SKIPPED) continue;
                const bool f0 = (c->p.chains == 1) || ((c->p.chains <= 1) == (rs < 2)), f1 = (c->p.chains == 1) || ((c->p.chains <= 1) == (rs == 2));
                if (f0) carry[slot][0] = one.start_off[0];
                if (f1) carry[slot][1] = one.start_off[1];
                break;
            }
    }
    return BASAL_OK;
}
