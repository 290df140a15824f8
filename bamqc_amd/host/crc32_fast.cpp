// crc32_fast.cpp — see crc32_fast.h
//
// Method: the folding scheme of Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction"
// (Intel, 2009), for the bit-reflected gzip polynomial.  Written for this repository: the fold multipliers are DERIVED at
// start-up from the polynomial (fold_constant below) instead of being pasted in, the accumulators are an array, and the
// last 128 -> 32 bit step is an ordinary table update over the accumulator's 16 bytes (zlib's crc32) rather than the
// paper's Barrett reduction — 16 byte steps per 64 KiB block cost nothing.
#include "crc32_fast.h"

#include <zlib.h>

#if defined(__x86_64__)
#include <immintrin.h>

namespace {
// x^n mod P over GF(2) for P = x^32 + x^26 + ... + 1 (0x104C11DB7), bit-reflected into 32 bits and shifted left once:
// the multiplier that moves a 64-bit half of a reflected accumulator forward by n - 32 bits of message.
uint64_t fold_constant(unsigned n)
{
    uint64_t r = 1;
    for (unsigned i = 0; i < n; ++i) {
        r <<= 1;
        if (r & (1ull << 32)) r ^= 0x104C11DB7ull;
    }
    uint64_t v = 0;
    for (unsigned b = 0; b < 32; ++b) v |= ((r >> b) & 1ull) << (31 - b);
    return v << 1;
}

struct FoldKeys {
    __m128i by512, by128; // low half: multiplier of the accumulator's low 64 bits (distance + 32), high half: of its high 64 bits (distance - 32)
    FoldKeys()
    {
        by512 = _mm_set_epi64x((long long)fold_constant(512 - 32), (long long)fold_constant(512 + 32));
        by128 = _mm_set_epi64x((long long)fold_constant(128 - 32), (long long)fold_constant(128 + 32));
    }
};

__attribute__((target("pclmul,sse4.1"))) inline __m128i fold(__m128i acc, __m128i key, __m128i next)
{
    const __m128i lo = _mm_clmulepi64_si128(acc, key, 0x00), hi = _mm_clmulepi64_si128(acc, key, 0x11);
    return _mm_xor_si128(_mm_xor_si128(lo, hi), next);
}

// raw CRC state (no inversions) after `len` bytes, len >= 64 and a multiple of 16, starting from raw state `state`
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_fold(const uint8_t* p, size_t len, uint32_t state)
{
    static const FoldKeys K;
    const __m128i* v = (const __m128i*)p;
    __m128i acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = _mm_loadu_si128(v + i);
    acc[0] = _mm_xor_si128(acc[0], _mm_cvtsi32_si128((int)state));
    size_t blocks = len / 16 - 4;
    v += 4;
    for (; blocks >= 4; blocks -= 4, v += 4) // four independent chains, each 512 bits of message further on
        for (int i = 0; i < 4; ++i) acc[i] = fold(acc[i], K.by512, _mm_loadu_si128(v + i));
    __m128i one = acc[0]; // the four chains are 128 bits apart: fold them together, then take the remaining 16-byte blocks
    for (int i = 1; i < 4; ++i) one = fold(one, K.by128, acc[i]);
    for (; blocks; --blocks, ++v) one = fold(one, K.by128, _mm_loadu_si128(v));
    // `one` is congruent to the whole message: its CRC from state 0 is the state we want
    alignas(16) uint8_t tail[16];
    _mm_store_si128((__m128i*)tail, one);
    return (uint32_t)crc32(0xFFFFFFFFul, tail, 16) ^ 0xFFFFFFFFu; // zlib inverts on the way in and out
}
} // namespace
#endif

uint32_t bqc_crc32_fast(const uint8_t* p, size_t n)
{
#if defined(__x86_64__)
    static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (have && n >= 64) {
        const size_t body = n & ~(size_t)15;
        const uint32_t c = ~crc32_fold(p, body, ~0u); // gzip's convention: initial value and result are inverted
        return n == body ? c : (uint32_t)crc32(c, p + body, (uInt)(n - body));
    }
#endif
    return (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n);
}
