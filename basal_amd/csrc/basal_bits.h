// basal_bits.h -- bit primitives of the BASAL hot path, usable from host C++ and gfx950 device code.
//
// Each function restates one inline of the reference's Param class (param.h, cited per function)
// in the form that maps best onto CDNA4: 64-bit scalar/vector integer ops and v_bcnt popcounts
// instead of SWAR multiplies.  Results are bit-identical to the reference's (KATs in tests/).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BASAL_HD __host__ __device__ __forceinline__
#else
#define BASAL_HD inline
#endif

namespace basal {

constexpr uint64_t kPairLo = 0x5555555555555555ULL;  // low bit of every 2-bit base code
constexpr uint64_t kPairHi = 0xAAAAAAAAAAAAAAAAULL;

// Param::XT (param.h:107-116): 16 base codes, 11 -> 01, read as a base-3 number (MSB first).
BASAL_HD uint32_t XT(uint32_t tt) {
    uint32_t ss;
    tt -= (tt << 1) & tt & 0xAAAAAAAAu;
    tt -= (tt >> 2) & 0x33333333u;
    ss = (tt & 0xF0F0F0F0u) >> 1;
    tt -= ss - (ss >> 3);
    ss = (tt & 0xFF00FF00u) >> 2;
    tt = (tt & 0x00FF00FFu) + ss + (ss >> 2) + (ss >> 6);
    return (tt & 0xFFFFu) + (tt >> 16) * 6561u;
}

// Param::XC64 (param.h:119): per base, 01 stays 01, everything else becomes 11.
BASAL_HD uint64_t XC64(uint64_t tt) { return ((~tt) << 1) | tt | kPairLo; }

// Param::M2_judge (param.h:142): 01 -> 00, 11 stays 11.
BASAL_HD uint64_t M2_judge(uint64_t tt) { return tt & (((tt & kPairHi) >> 1) | ((tt & kPairLo) << 1)); }

// one bit (the low bit of the pair) per base whose 2-bit code is non-zero
BASAL_HD uint64_t pair_mask(uint64_t x) { return (x | (x >> 1)) & kPairLo; }

BASAL_HD uint32_t popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popcll(x);
#else
    return (uint32_t)__builtin_popcountll(x);
#endif
}

// Param::XM64 (param.h:129-139): number of non-zero 2-bit pairs.
BASAL_HD uint32_t XM64(uint64_t x) { return popc64(pair_mask(x)); }

// myrand (utilities.cpp:38-48), the -S != 0 branch: a stateless hash of the read index.
BASAL_HD uint32_t myrand(uint32_t index, uint32_t randseed) {
    uint32_t s = randseed * 1000000u;  // bit32_t arithmetic in the reference
    uint64_t v = ((uint64_t)(int64_t)(int32_t)index + s) * 3935559000370003845ULL + 2691343689449507681ULL;
    v ^= v >> 21; v ^= v << 37; v ^= v >> 4;
    v *= 4768777513237032717ULL;
    v ^= v << 20; v ^= v >> 41; v ^= v << 5;
    return (uint32_t)v;
}

// The even-numbered bits of x (bit 0, 2, 4 ...) packed into 32 bits.
BASAL_HD uint32_t even_bits(uint64_t x) {
    x &= kPairLo;
    x = (x | (x >> 1)) & 0x3333333333333333ULL;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0FULL;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFULL;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFULL;
    return (uint32_t)(x | (x >> 16));
}
// A 32-base word as two bit planes: the high bits of the 32 base codes in the upper half, the low bits in the lower (base 0 on top in both).
// The flank words of cores whose index keeps long lists are stored this way (basal_index.hip), so that the stream test needs no shifts.
BASAL_HD uint64_t split_planes(uint64_t w) { return ((uint64_t)even_bits(w >> 1) << 32) | even_bits(w); }
// 16 bases (a 32-bit word) likewise: high bits in the upper 16, low bits in the lower 16
BASAL_HD uint32_t split_planes16(uint32_t w) { return (even_bits(w >> 1) << 16) | even_bits(w); }

// Conversion-tolerant comparison of one 32-base word (the body of CountMismatch, align.h:126-128,
// and of CountMismatch_new, align.h:210-219): non-zero pairs of the result are mismatches.
// rw = read bases, cw = read "is a convert-to base" plane, s = reference bases (same frame).
template <bool NEWRULE>
BASAL_HD uint64_t cmp_word(uint64_t rw, uint64_t cw, uint64_t s) {
    if (!NEWRULE) return (rw & XC64(s)) ^ s;
    uint64_t M2 = XC64(s) | cw;
    uint64_t M3 = M2_judge(M2);
    return (((~M3) & M2) | (M3 & rw)) ^ s;
}

// Param::InitMapping (param.cpp:70-74): read offset of phase i of seed segment j, before the
// per-segment start shift: the smallest multiple of I that is >= j*k+i.
BASAL_HD uint32_t profile(uint32_t j, uint32_t i, uint32_t K, uint32_t I) { return ((j * K + i + I - 1) / I) * I; }

// mask of the first n bases (n in 0..32) of a word, in pair_mask form
BASAL_HD uint64_t prefix_pairs(int n) {
    if (n <= 0) return 0;
    if (n >= 32) return kPairLo;
    return kPairLo & (~0ULL << (64 - 2 * n));
}

}  // namespace basal
