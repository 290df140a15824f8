// gpu_inflate.hip — raw DEFLATE (RFC 1951) of BGZF blocks and their CRC-32 on the GPU (row N2).  Used by the BAM reader on the card
// (gpu_bam.hip: the inflated bytes stay in device memory — the program's default for whole BAM files), and, opt-in and for
// comparison only (BQC_GPU_INFLATE=1 / bqc_gpu_inflate_device), by the host reader's BGZF layer, which copies the bytes back.
//
// BGZF blocks are independent, at most 64 KiB, and carry their uncompressed size.  The default path (round 3) takes a launch in TWO
// PHASES: k_inflate_wave (gpu_inflate_wave.inc) — a WAVE per block decodes the symbols, 64 pieces of the stream at once from guessed
// starts that are verified lane to lane, and writes literals and match descriptions — and k_inflate_resolve — a workgroup per block
// fills the matches in by pointer jumping over a 16-bit index per output byte in LDS; then k_gi_crc checks every block against its
// CRC-32.  45 K blocks (2.9 GB of output): 17.9 + 9 + 1.7 ms.  What it replaced stays selectable and tested: ONE LANE PER BLOCK
// (BQC_GI_WAVE=0: k_inflate with root tables in LDS — 2.1 KB per lane: 9-bit root table of the literal/length code, 7-bit of the
// distance code as u16 `symbol << 4 | code bits`, the canonical code itself for longer codes, every per-lane array laid out
// [entry][lane] — or k_inflate_lean with 356 bytes per lane), and the lanes copying their matches themselves (BQC_GI_TWO_PHASE=0:
// 32 / 16 / 4 / 1 bytes at a time from the lane's own output).
//
// History of the lane-per-block kernels (MI355X; why they are no longer the default): a run of 12.9 K blocks = 268 MB -> 844 MB took
// 46 / 40 / 35 / 32 ms with 64 / 32 / 16 / 8 lanes per workgroup; a lane spends ~1.5 us per symbol whatever the workgroup width — a
// match reads the lane's own output up to 32 KiB back, 13 K lanes x 64 KiB of output are far more than the L2 holds, and in every
// iteration some lane of the wave has a match: the kernel's time is the time of one block (22 ms for 3.4 K blocks, 32 ms for 13 K)
// until the card is full.  Measured in round 3: the copies are half of the time (FETCH_SIZE 20 GB for 2.7 GB of algorithmic bytes),
// literal stores are free; resolving the matches inside the lane after the symbols was 30-60 % slower, holding short matches in
// registers 8 % slower.  With the bytes copied BACK to the host the program does not get faster (10 M reads
// 0.59-0.69 s against 0.51-0.53 s: BQC_GPU_INFLATE=1 keeps that path for comparison); it pays once the records are walked and decoded
// on the card as well (gpu_bam.hip).
//
// Format: RFC 1951 (public); acceptance rules as the host decoder's (bamqc_amd/host/inflate_fast.cpp): over-subscribed or
// incomplete code sets, a missing end-of-block code, distances before the block's start, output other than ISIZE bytes, input
// beyond the block are errors — a corrupt file is reported, never followed out of bounds.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "kernels_common.h"
#include "gpu_inflate.h"

// per-lane arrays, u16 units, laid out [entry][lane]
#define GI_LIT_ROOT 9
#define GI_DIST_ROOT 7
#define GI_O_LIT   0                          // [512] root table literal/length
#define GI_O_DIST  (GI_O_LIT + 512)           // [128] root table distance
#define GI_O_LCNT  (GI_O_DIST + 128)          // [16] codes per length, literal/length
#define GI_O_DCNT  (GI_O_LCNT + 16)           // [16] ... distance
#define GI_O_LSYM  (GI_O_DCNT + 16)           // [288] literal/length symbols in canonical order
#define GI_O_DSYM  (GI_O_LSYM + 288)          // [32] distance symbols in canonical order
#define GI_O_LENS  (GI_O_DSYM + 32)           // [80] code lengths while the tables are built, four per u16 (320 nibbles)
#define GI_U16     (GI_O_LENS + 80)           // 1072 u16 = 2144 B per lane, 137 216 B per wave
#define GI_AT(k) ((k) * NL) // NL: lanes (= blocks) per workgroup

__constant__ uint8_t c_pre_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

typedef uint32_t __attribute__((aligned(1))) gi_u32_u;
typedef uint32_t gi_u32x4 __attribute__((ext_vector_type(4)));
typedef gi_u32x4 __attribute__((aligned(1))) gi_u32x4_u;

namespace {
typedef uint64_t __attribute__((aligned(1))) gi_u64_u;

struct Bits { // bit reader over [in, end): the two words behind the buffered ones are always on their way (one 8-byte load per 64 bits of
              // input; its latency hides behind the symbols decoded meanwhile); loads are clamped to `lim` (inside the compressed buffer),
              // what was consumed is checked at the end
    const uint8_t* in;  // address of the next prefetched word (the low half of `nw` while nwn == 2, else its high half)
    const uint8_t* lim; // last address a load may start at
    uint64_t bb;
    uint64_t nw;        // prefetched: nwn words of 32 bits, the next one in the low half
    uint32_t bc;
    uint32_t nwn;
    __device__ __forceinline__ void start(const uint8_t* p, const uint8_t* end)
    {
        lim = end + 4; // (the buffer holds at least 64 bytes behind the last block)
        bb = *(const gi_u32_u*)p | (uint64_t)(*(const gi_u32_u*)(p + 4)) << 32;
        bc = 64;
        in = p + 8;
        nw = *(const gi_u64_u*)(in < lim ? in : lim);
        nwn = 2;
    }
    __device__ __forceinline__ void refill()
    {
        if (bc <= 32u) {
            bb |= (uint64_t)(uint32_t)nw << bc;
            bc += 32u;
            in += 4;
            nw >>= 32;
            if (--nwn == 0u) { nw = *(const gi_u64_u*)(in < lim ? in : lim); nwn = 2u; }
        }
    }
    __device__ __forceinline__ const uint8_t* byte_pos() const { return in - (bc >> 3); } // address of the first unconsumed byte (bc a multiple of 8)
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t)bb & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(uint32_t n) { bb >>= n; bc -= n; }
    __device__ __forceinline__ uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
};

// The output side of a lane.  Literals are gathered in a register and stored eight at a time.  Matches: measured on a full card (45 K
// blocks in a launch, profiles/r3_inflate_*.json), copying them inside the lanes is HALF of the kernel's time — every lane copies from
// its own block's output, up to 32 KiB back, 45 K blocks x 64 KiB of output are 2.9 GB: the copies' loads miss every cache (FETCH_SIZE
// 20 GB for 2.7 GB of algorithmic bytes), and the launch stops scaling with the number of blocks (without the copies: 23 ms for 45 K
// blocks, 28-32 ms for 90 K; with them 46-49 and 93-96); literal stores cost nothing measurable.  So the work is done in TWO PHASES
// (the default; BQC_GI_TWO_PHASE=0: one): a lane writes its literals to their final places and, for a match, leaves distance and length
// in the first three of the bytes the match will fill (a match is at least three bytes long) and sets the bit of its first byte in a
// bitmap; k_inflate_resolve (below) fills the matches in, a workgroup per block, by pointer jumping in LDS: 36 ms for 45 K blocks
// instead of 49, 56-67 ms for 90 K instead of 98.  With bm == nullptr the lane copies its matches itself: up to 16 bytes at a
// distance of 16 or more are one 16-byte load and one store (most matches of a BGZF level-1 stream are short: 6.7 bytes on average in
// the synthetic files, and 90-96 % of the output comes from matches), longer ones 32 bytes per step, overlapping ones word- or
// byte-wise; stores may run past the current end of the output — never past the block's — and are overwritten by what follows.
struct Out {
    uint8_t* o0;
    uint32_t usize;
    uint32_t o;      // bytes produced (in two phases: matches counted, not copied), pending literals included
    uint32_t nlit;   // literals held in `lit` (0..7): the bytes o - nlit .. o - 1
    uint64_t lit;
    uint32_t* bm;    // two phases: this block's bitmap of match starts (bit p: a match begins at byte p); nullptr: one phase
    uint32_t bmi, bmw; // the bitmap word being filled and its index (the words are visited in ascending order; the bitmap was zeroed)
    uint32_t nmatch;
    __device__ __forceinline__ void flush()
    {
        if (!nlit) return;
        uint8_t* d = o0 + (o - nlit);
        if (o - nlit + 8u <= usize) *(gi_u64_u*)d = lit; // (runs past the literals into bytes that are written later, or by phase 2)
        else for (uint32_t k = 0; k < nlit; ++k) d[k] = (uint8_t)(lit >> (8u * k));
        nlit = 0; lit = 0;
    }
    __device__ __forceinline__ void literal(uint32_t sym) // (the caller has checked o < usize)
    {
        lit |= (uint64_t)sym << (8u * nlit);
        ++o;
        if (++nlit == 8u) { *(gi_u64_u*)(o0 + (o - 8u)) = lit; nlit = 0; lit = 0; }
    }
    __device__ __forceinline__ void raw(uint32_t n) { o += n; } // n bytes were written directly (a stored block)
    __device__ __forceinline__ void match(uint32_t length, uint32_t dist) // (the caller has checked dist <= o, o + length <= usize)
    {
        flush(); // (the literals in front of the match; those behind it start a new register)
        if (bm) {
            // the match is not copied: its description goes where its first three bytes will be (a match is at least three bytes long) —
            // distance - 1 in 15 bits, length - 3 in the third byte — and its start is marked in the block's bitmap
            const uint32_t tok = (dist - 1u) | ((length - 3u) << 16);
            uint8_t* d = o0 + o;
            if (o + 4u <= usize) *(gi_u32_u*)d = tok; // (the fourth byte belongs to what follows and is written after this)
            else { d[0] = (uint8_t)tok; d[1] = (uint8_t)(tok >> 8); d[2] = (uint8_t)(tok >> 16); }
            const uint32_t w = o >> 5;
            if (w != bmi) { if (bmw) bm[bmi] = bmw; bmi = w; bmw = 0; }
            bmw |= 1u << (o & 31u);
            ++nmatch;
            o += length;
            return;
        }
        uint8_t* dst = o0 + o;
        const uint8_t* src = dst - dist;
        o += length;
        if (dist >= 16u && length <= 16u && o - length + 16u <= usize) {
            *(gi_u32x4_u*)dst = *(const gi_u32x4_u*)src;
        } else if (dist >= 32u && o + 32u <= usize) { // 32 bytes at a time (both loads first); the overshoot stays inside this block's output
            for (uint32_t k = 0; k < length; k += 32u) {
                const gi_u32x4 a = *(const gi_u32x4_u*)(src + k), b = *(const gi_u32x4_u*)(src + k + 16);
                *(gi_u32x4_u*)(dst + k) = a;
                *(gi_u32x4_u*)(dst + k + 16) = b;
            }
        } else if (dist >= 16u && o + 16u <= usize) {
            for (uint32_t k = 0; k < length; k += 16u) *(gi_u32x4_u*)(dst + k) = *(const gi_u32x4_u*)(src + k);
        } else if (dist >= 4u && o + 4u <= usize) {
            for (uint32_t k = 0; k < length; k += 4u) *(gi_u32_u*)(dst + k) = *(const gi_u32_u*)(src + k);
        } else {
            for (uint32_t k = 0; k < length; ++k) dst[k] = src[k];
        }
    }
    __device__ __forceinline__ void finish() { flush(); if (bm && bmw) bm[bmi] = bmw; }
};

// where block bi's bitmap of match starts begins (in 32-bit words): the blocks of a launch lie back to back in the output; a block of u
// bytes gets at least ceil(u / 32) words of its own
__device__ __forceinline__ uint64_t gi_bm_base(const GiBlock* __restrict__ blocks, uint32_t bi) { return ((blocks[bi].uoff - blocks[0].uoff) >> 5) + (uint64_t)bi; }

template <int NL> __device__ __forceinline__ uint32_t lens_get(const uint16_t* L, uint32_t i) { const uint32_t v = L[GI_AT(GI_O_LENS + (i >> 2))]; return (v >> (4 * (i & 3))) & 15u; }
template <int NL> __device__ __forceinline__ void lens_set(uint16_t* L, uint32_t i, uint32_t len)
{
    uint16_t& w = L[GI_AT(GI_O_LENS + (i >> 2))];
    w = (uint16_t)((w & ~(15u << (4 * (i & 3)))) | (len << (4 * (i & 3))));
}

// Canonical code of `n` symbols whose lengths are lens[first .. first + n): counts per length, symbols in canonical order, root
// table of `root` bits (entry = symbol << 4 | length; 0 = longer than the root or unused).  zlib's acceptance rules.
template <int NL> __device__ bool build_code(uint16_t* L, uint32_t first, uint32_t n, uint32_t o_cnt, uint32_t o_sym, uint32_t o_root, uint32_t root)
{
    uint32_t count[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (uint32_t s = 0; s < n; ++s) {
        const uint32_t len = lens_get<NL>(L, first + s);
#pragma unroll
        for (int l = 0; l < 16; ++l) count[l] += len == (uint32_t)l; // (a register array cannot be indexed by a variable)
    }
    int left = 1;
    uint32_t maxl = 0;
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - (int)count[l];
        if (left < 0) return false;
        if (count[l]) maxl = (uint32_t)l;
    }
    if (left > 0 && maxl > 1) return false;
    uint32_t offs[16];
    offs[0] = 0; offs[1] = 0;
#pragma unroll
    for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + count[l];
#pragma unroll
    for (int l = 0; l < 16; ++l) L[GI_AT(o_cnt + l)] = (uint16_t)(l ? count[l] : 0);
    for (uint32_t k = 0; k < (1u << root); ++k) L[GI_AT(o_root + k)] = 0;
    // symbols in canonical order; the code of a symbol = first code of its length + its rank among the symbols of that length
    uint32_t next_code[16];
    {
        uint32_t code = 0;
        next_code[0] = 0;
#pragma unroll
        for (int l = 1; l <= 15; ++l) { code = (code + count[l - 1]) << 1; next_code[l] = code; }
    }
    for (uint32_t s = 0; s < n; ++s) {
        const uint32_t len = lens_get<NL>(L, first + s);
        if (!len) continue;
        uint32_t pos = 0, code = 0;
#pragma unroll
        for (int l = 1; l <= 15; ++l)
            if (len == (uint32_t)l) { pos = offs[l]++; code = next_code[l]++; }
        L[GI_AT(o_sym + pos)] = (uint16_t)s;
        if (len <= root) {
            const uint32_t r = __brev(code) >> (32u - len); // the code's bits in stream order (LSB first)
            const uint16_t e = (uint16_t)((s << 4) | len);
            for (uint32_t i = r; i < (1u << root); i += 1u << len) L[GI_AT(o_root + i)] = e;
        }
    }
    return true;
}

// one symbol of a canonical code: root table first, bit by bit for the longer codes; returns the symbol or 0xFFFF (invalid)
template <int NL> __device__ __forceinline__ uint32_t decode_sym(const uint16_t* L, Bits& B, uint32_t o_cnt, uint32_t o_sym, uint32_t o_root, uint32_t root)
{
    const uint32_t e = L[GI_AT(o_root + B.peek(root))];
    if (e & 15u) { B.drop(e & 15u); return e >> 4; }
    // longer than the root (or an unused pattern): puff-style walk over the lengths
    uint32_t code = 0, firstc = 0, index = 0;
    uint32_t bits = (uint32_t)B.bb;
    for (uint32_t len = 1; len <= 15; ++len) {
        code |= bits & 1u;
        bits >>= 1;
        const uint32_t cnt = L[GI_AT(o_cnt + len)];
        if (code - firstc < cnt) { B.drop(len); return L[GI_AT(o_sym + index + (code - firstc))]; }
        index += cnt;
        firstc = (firstc + cnt) << 1;
        code <<= 1;
    }
    return 0xFFFFu;
}
} // namespace

template <int NL> __global__ __launch_bounds__(NL) __attribute__((amdgpu_num_vgpr(96))) void k_inflate(const uint8_t* __restrict__ comp, const GiBlock* __restrict__ blocks, uint32_t n_blocks,
                                                       uint8_t* __restrict__ out, uint32_t* __restrict__ status, uint32_t* __restrict__ bitmap, uint32_t* __restrict__ ntok)
{
    extern __shared__ uint16_t lds16[];
    const uint32_t bi = blockIdx.x * NL + threadIdx.x;
    if (bi >= n_blocks) return;
    uint16_t* L = lds16 + threadIdx.x;
    const GiBlock blk = blocks[bi];
    const uint8_t* const c0 = comp + blk.coff;
    const uint8_t* const cend = c0 + blk.csize;
    Bits B;
    B.start(c0, cend);
    const uint32_t usize = blk.usize;
    Out O{out + blk.uoff, usize, 0u, 0u, 0ull, bitmap ? bitmap + gi_bm_base(blocks, bi) : nullptr, 0u, 0u, 0u};
    uint8_t* const o0 = O.o0;
    uint32_t st = 0; // 0 ok, else the reason (GI_ERR_*)
    for (;;) {
        B.refill();
        if (B.byte_pos() > cend) { st = GI_ERR_TRUNC; break; } // (also what ends a run of empty blocks decoded from beyond the stream)
        const uint32_t bfinal = B.take(1), btype = B.take(2);
        if (btype == 0u) { // stored: back to a byte boundary, LEN / NLEN, raw bytes
            B.drop(B.bc & 7u);
            const uint8_t* p = B.byte_pos();
            if (p + 4 > cend) { st = GI_ERR_TRUNC; break; }
            const uint32_t len = p[0] | (p[1] << 8), nlen = p[2] | (p[3] << 8);
            if ((len ^ nlen) != 0xFFFFu) { st = GI_ERR_DATA; break; }
            p += 4;
            if (p + len > cend || O.o + len > usize) { st = GI_ERR_TRUNC; break; }
            O.flush();
            uint32_t k = 0;
            for (; k + 4 <= len; k += 4) *(gi_u32_u*)(o0 + O.o + k) = *(const gi_u32_u*)(p + k);
            for (; k < len; ++k) o0[O.o + k] = p[k];
            O.raw(len);
            B.start(p + len, cend);
            if (bfinal) break;
            continue;
        }
        if (btype == 3u) { st = GI_ERR_DATA; break; }
        uint32_t hlit = 288, hdist = 30;
        if (btype == 1u) { // fixed code
            for (uint32_t i = 0; i < 80; ++i) L[GI_AT(GI_O_LENS + i)] = 0;
            for (uint32_t i = 0; i < 144; ++i) lens_set<NL>(L, i, 8);
            for (uint32_t i = 144; i < 256; ++i) lens_set<NL>(L, i, 9);
            for (uint32_t i = 256; i < 280; ++i) lens_set<NL>(L, i, 7);
            for (uint32_t i = 280; i < 288; ++i) lens_set<NL>(L, i, 8);
            for (uint32_t i = 0; i < 30; ++i) lens_set<NL>(L, 288 + i, 5);
            hdist = 30; // (the two unused 5-bit codes decode to symbols 30 / 31: rejected below)
            for (uint32_t i = 30; i < 32; ++i) lens_set<NL>(L, 288 + i, 5);
            hdist = 32;
        } else {
            B.refill();
            hlit = B.take(5) + 257u; hdist = B.take(5) + 1u;
            const uint32_t hclen = B.take(4) + 4u;
            if (hlit > 286u || hdist > 30u) { st = GI_ERR_DATA; break; }
            // code-length code: 19 symbols of up to 7 bits; its table lives where the literal/length root table will be
            for (uint32_t i = 0; i < 80; ++i) L[GI_AT(GI_O_LENS + i)] = 0;
            for (uint32_t i = 0; i < hclen; ++i) { B.refill(); lens_set<NL>(L, 300u + c_pre_order[i], B.take(3)); } // (precode lengths parked at 300..318)
            if (!build_code<NL>(L, 300, 19, GI_O_LCNT, GI_O_LSYM, GI_O_LIT, 7)) { st = GI_ERR_DATA; break; }
            {   // an incomplete precode is an error (zlib)
                int left = 1;
                for (uint32_t l = 1; l <= 15; ++l) left = (left << 1) - (int)L[GI_AT(GI_O_LCNT + l)];
                if (left > 0) { st = GI_ERR_DATA; break; }
            }
            for (uint32_t i = 0; i < 19; ++i) lens_set<NL>(L, 300u + i, 0);
            const uint32_t total = hlit + hdist;
            uint32_t i = 0, prev = 0;
            bool bad = false;
            while (i < total) {
                B.refill();
                const uint32_t sym = decode_sym<NL>(L, B, GI_O_LCNT, GI_O_LSYM, GI_O_LIT, 7);
                if (sym < 16u) { lens_set<NL>(L, i++, sym); prev = sym; continue; }
                uint32_t rep, val = 0;
                if (sym == 16u) { if (i == 0) { bad = true; break; } val = prev; rep = 3u + B.take(2); }
                else if (sym == 17u) rep = 3u + B.take(3);
                else if (sym == 18u) rep = 11u + B.take(7);
                else { bad = true; break; }
                if (i + rep > total) { bad = true; break; }
                for (uint32_t k = 0; k < rep; ++k) lens_set<NL>(L, i + k, val);
                i += rep;
                prev = val;
            }
            if (bad) { st = GI_ERR_DATA; break; }
            if (lens_get<NL>(L, 256) == 0u) { st = GI_ERR_DATA; break; } // no end-of-block code
            // the distance lengths follow the literal/length lengths directly: move them to 288.. (from the back: the ranges may overlap)
            for (uint32_t k = hdist; k-- > 0;) lens_set<NL>(L, 288u + k, lens_get<NL>(L, hlit + k));
            for (uint32_t k = hlit; k < 288u; ++k) lens_set<NL>(L, k, 0);
        }
        if (!build_code<NL>(L, 288, hdist, GI_O_DCNT, GI_O_DSYM, GI_O_DIST, GI_DIST_ROOT)) { st = GI_ERR_DATA; break; }
        if (!build_code<NL>(L, 0, hlit, GI_O_LCNT, GI_O_LSYM, GI_O_LIT, GI_LIT_ROOT)) { st = GI_ERR_DATA; break; }
        // ---- symbols of one block
        for (;;) {
            B.refill();
            const uint32_t sym = decode_sym<NL>(L, B, GI_O_LCNT, GI_O_LSYM, GI_O_LIT, GI_LIT_ROOT);
            if (sym < 256u) {
                if (O.o >= usize) { st = GI_ERR_SIZE; break; }
                O.literal(sym);
                continue;
            }
            if (sym == 256u) break;
            if (sym > 285u) { st = GI_ERR_DATA; break; }
            B.refill();
            // length symbols 257..264: 3..10; 265..284 in groups of four with 1..5 extra bits; 285: 258 (RFC 1951, 3.2.5) — computed, a
            // table would be a memory access in the middle of the chain
            const uint32_t ls = sym - 257u, le = ls < 8u || ls == 28u ? 0u : (ls - 4u) >> 2;
            const uint32_t length = (ls < 8u ? ls + 3u : ls == 28u ? 258u : ((4u + (ls & 3u)) << le) + 3u) + B.take(le);
            B.refill();
            const uint32_t ds = decode_sym<NL>(L, B, GI_O_DCNT, GI_O_DSYM, GI_O_DIST, GI_DIST_ROOT);
            if (ds > 29u) { st = GI_ERR_DATA; break; }
            B.refill();
            const uint32_t de = ds < 4u ? 0u : (ds >> 1) - 1u; // distance symbols 0..3: 1..4; then pairs with 1..13 extra bits
            const uint32_t dist = (ds < 4u ? ds + 1u : ((2u + (ds & 1u)) << de) + 1u) + B.take(de);
            if (dist > O.o || O.o + length > usize) { st = dist > O.o ? GI_ERR_DATA : GI_ERR_SIZE; break; }
            O.match(length, dist);
        }
        if (st || bfinal) break;
    }
    O.finish();
    if (!st && O.o != usize) st = GI_ERR_SIZE;
    if (!st && (uint64_t)(B.in - c0) * 8u - B.bc > (uint64_t)blk.csize * 8u) st = GI_ERR_TRUNC; // bits from beyond the stream were consumed
    if (bitmap) ntok[bi] = st ? 0u : O.nmatch; // (0: nothing for phase 2 to do)
    if (st) atomicOr(status, st);
}

// ---------------------------------------------------------------------------------------------------
// the same decoder with 356 bytes of LDS per block instead of 2144: whole files in one launch
// ---------------------------------------------------------------------------------------------------
// k_inflate's time is the time a lane needs for its block, and LDS decides how many lanes a launch may hold (72 per CU with the
// root tables).  Here a code is kept as what a canonical code IS — for every length the left-aligned limit `(first code + count)
// << (15 - length)` (the next length's first code) and the rank of its first symbol, in registers — and a symbol is found by
// comparing the next 15 bits, bit-reversed, against the limits (they are non-decreasing: its length is 1 + the number of limits
// it reaches) and reading symbol `rank + (bits - first) >> (15 - length)` of the canonical order, which is all that stays in
// LDS: a byte per literal/length symbol plus a bit for "256 and above", a byte per distance symbol.  More VALU work per symbol
// (~150 instructions), which the SIMDs have to spare: the loop waits for memory most of the time.  460 blocks per CU in flight.
#define GL_LS 0     // [288] low byte of the literal/length symbols in canonical order (during the code lengths: the precode's 19)
#define GL_LB 288   // [36]  bit i: symbol i of that order is >= 256
#define GL_DS 324   // [32]  distance symbols in canonical order
#define GL_BYTES 356
#define GL_AT(k) ((k) * NL)

namespace {
struct Canon { uint32_t lim[8], off[8]; }; // 16-bit halves: lim[l], off[l] for l = 0..15
__device__ __forceinline__ uint32_t cn_lim(const Canon& C, int l) { return (C.lim[l >> 1] >> (16 * (l & 1))) & 0xFFFFu; }
__device__ __forceinline__ uint32_t cn_off(const Canon& C, int l) { return (C.off[l >> 1] >> (16 * (l & 1))) & 0xFFFFu; }

// count[l] codes of length l -> limits and ranks; zlib's acceptance rules (kind 0: the code-length code must be complete)
__device__ __forceinline__ bool canon_make(Canon& C, const uint32_t (&count)[16], bool must_be_complete)
{
    int left = 1;
    uint32_t maxl = 0;
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - (int)count[l];
        if (left < 0) return false;
        if (count[l]) maxl = (uint32_t)l;
    }
    if (left > 0 && (must_be_complete || maxl > 1)) return false;
    uint32_t lim[16], off[16];
    lim[0] = 0; off[0] = 0; off[1] = 0;
    uint32_t code = 0;
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        code = (code + (l > 1 ? count[l - 1] : 0u)) << 1;
        lim[l] = (code + count[l]) << (15 - l);
        if (l < 15) off[l + 1] = off[l] + count[l];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { C.lim[k] = lim[2 * k] | (lim[2 * k + 1] << 16); C.off[k] = off[2 * k] | (off[2 * k + 1] << 16); }
    return true;
}

// index of the next symbol in canonical order and its code length (0: no such code)
__device__ __forceinline__ uint32_t canon_find(const Canon& C, uint32_t bits32, uint32_t& len)
{
    const uint32_t r = __brev(bits32) >> 17; // the next 15 bits, first bit on top
    uint32_t n = 1, first = 0, offv = cn_off(C, 1);
#pragma unroll
    for (int l = 1; l <= 14; ++l) {
        const uint32_t lm = cn_lim(C, l);
        const bool ge = r >= lm;
        n += ge;
        first = ge ? lm : first;
        offv = ge ? cn_off(C, l + 1) : offv;
    }
    len = r >= cn_lim(C, 15) ? 0u : n;
    return offv + ((r - first) >> (15u - n));
}
} // namespace

template <int NL> __global__ __launch_bounds__(NL) void k_inflate_lean(const uint8_t* __restrict__ comp, const GiBlock* __restrict__ blocks, uint32_t n_blocks,
                                                                      uint8_t* __restrict__ out, uint32_t* __restrict__ status, uint32_t* __restrict__ bitmap, uint32_t* __restrict__ ntok)
{
    extern __shared__ uint8_t lds8[];
    const uint32_t bi = blockIdx.x * NL + threadIdx.x;
    if (bi >= n_blocks) return;
    uint8_t* L = lds8 + threadIdx.x;
    const GiBlock blk = blocks[bi];
    const uint8_t* const c0 = comp + blk.coff;
    const uint8_t* const cend = c0 + blk.csize;
    Bits B;
    B.start(c0, cend);
    const uint32_t usize = blk.usize;
    Out O{out + blk.uoff, usize, 0u, 0u, 0ull, bitmap ? bitmap + gi_bm_base(blocks, bi) : nullptr, 0u, 0u, 0u};
    uint8_t* const o0 = O.o0;
    uint32_t st = 0;
    uint8_t lens[320]; // code lengths while the codes are built (private memory)
    for (;;) {
        B.refill();
        if (B.byte_pos() > cend) { st = GI_ERR_TRUNC; break; }
        const uint32_t bfinal = B.take(1), btype = B.take(2);
        if (btype == 0u) { // stored
            B.drop(B.bc & 7u);
            const uint8_t* p = B.byte_pos();
            if (p + 4 > cend) { st = GI_ERR_TRUNC; break; }
            const uint32_t len = p[0] | (p[1] << 8), nlen = p[2] | (p[3] << 8);
            if ((len ^ nlen) != 0xFFFFu) { st = GI_ERR_DATA; break; }
            p += 4;
            if (p + len > cend || O.o + len > usize) { st = GI_ERR_TRUNC; break; }
            O.flush();
            uint32_t k = 0;
            for (; k + 4 <= len; k += 4) *(gi_u32_u*)(o0 + O.o + k) = *(const gi_u32_u*)(p + k);
            for (; k < len; ++k) o0[O.o + k] = p[k];
            O.raw(len);
            B.start(p + len, cend);
            if (bfinal) break;
            continue;
        }
        if (btype == 3u) { st = GI_ERR_DATA; break; }
        uint32_t hlit = 288, hdist = 32;
        if (btype == 1u) { // fixed code (the two unused 5-bit distance codes decode to symbols 30 / 31: rejected below)
            for (uint32_t i = 0; i < 144; ++i) lens[i] = 8;
            for (uint32_t i = 144; i < 256; ++i) lens[i] = 9;
            for (uint32_t i = 256; i < 280; ++i) lens[i] = 7;
            for (uint32_t i = 280; i < 288; ++i) lens[i] = 8;
            for (uint32_t i = 0; i < 32; ++i) lens[288 + i] = 5;
        } else {
            B.refill();
            hlit = B.take(5) + 257u; hdist = B.take(5) + 1u;
            const uint32_t hclen = B.take(4) + 4u;
            if (hlit > 286u || hdist > 30u) { st = GI_ERR_DATA; break; }
            uint32_t pl[19];
#pragma unroll
            for (int i = 0; i < 19; ++i) pl[i] = 0;
            for (uint32_t i = 0; i < hclen; ++i) {
                B.refill();
                const uint32_t v = B.take(3), which = c_pre_order[i];
#pragma unroll
                for (int k = 0; k < 19; ++k) pl[k] = which == (uint32_t)k ? v : pl[k];
            }
            uint32_t cnt[16];
#pragma unroll
            for (int l = 0; l < 16; ++l) cnt[l] = 0;
#pragma unroll
            for (int k = 0; k < 19; ++k)
#pragma unroll
                for (int l = 1; l < 8; ++l) cnt[l] += pl[k] == (uint32_t)l;
            Canon P;
            if (!canon_make(P, cnt, true)) { st = GI_ERR_DATA; break; }
            { // the precode's symbols in canonical order
                uint32_t pos[8];
#pragma unroll
                for (int l = 0; l < 8; ++l) pos[l] = cn_off(P, l);
#pragma unroll
                for (int k = 0; k < 19; ++k)
#pragma unroll
                    for (int l = 1; l < 8; ++l)
                        if (pl[k] == (uint32_t)l) { L[GL_AT(GL_LS + pos[l])] = (uint8_t)k; ++pos[l]; }
            }
            const uint32_t total = hlit + hdist;
            uint32_t i = 0, prev = 0;
            bool bad = false;
            while (i < total) {
                B.refill();
                uint32_t cl;
                const uint32_t idx = canon_find(P, (uint32_t)B.bb, cl);
                if (!cl || idx >= 19u) { bad = true; break; }
                B.drop(cl);
                const uint32_t sym = L[GL_AT(GL_LS + idx)];
                if (sym < 16u) { lens[i++] = (uint8_t)sym; prev = sym; continue; }
                uint32_t rep, val = 0;
                if (sym == 16u) { if (i == 0) { bad = true; break; } val = prev; rep = 3u + B.take(2); }
                else if (sym == 17u) rep = 3u + B.take(3);
                else rep = 11u + B.take(7);
                if (i + rep > total) { bad = true; break; }
                for (uint32_t k = 0; k < rep; ++k) lens[i + k] = (uint8_t)val;
                i += rep;
                prev = val;
            }
            if (bad) { st = GI_ERR_DATA; break; }
            if (lens[256] == 0) { st = GI_ERR_DATA; break; } // no end-of-block code
            for (uint32_t k = hdist; k-- > 0;) lens[288 + k] = lens[hlit + k]; // (from the back: the ranges may overlap)
        }
        // ---- the two codes: limits and ranks in registers, symbols in canonical order in LDS
        Canon CL, CD;
        {
            uint32_t cnt[16];
#pragma unroll
            for (int l = 0; l < 16; ++l) cnt[l] = 0;
            for (uint32_t s = 0; s < hlit; ++s) {
                const uint32_t len = lens[s];
#pragma unroll
                for (int l = 1; l < 16; ++l) cnt[l] += len == (uint32_t)l;
            }
            if (!canon_make(CL, cnt, false)) { st = GI_ERR_DATA; break; }
            uint32_t pos[16];
#pragma unroll
            for (int l = 0; l < 16; ++l) pos[l] = cn_off(CL, l);
            for (uint32_t k = 0; k < 36; ++k) L[GL_AT(GL_LB + k)] = 0;
            for (uint32_t s = 0; s < hlit; ++s) {
                const uint32_t len = lens[s];
                if (!len) continue;
                uint32_t at = 0;
#pragma unroll
                for (int l = 1; l < 16; ++l)
                    if (len == (uint32_t)l) { at = pos[l]; ++pos[l]; }
                L[GL_AT(GL_LS + at)] = (uint8_t)s;
                if (s >= 256u) L[GL_AT(GL_LB + (at >> 3))] |= (uint8_t)(1u << (at & 7u));
            }
        }
        {
            uint32_t cnt[16];
#pragma unroll
            for (int l = 0; l < 16; ++l) cnt[l] = 0;
            for (uint32_t s = 0; s < hdist; ++s) {
                const uint32_t len = lens[288 + s];
#pragma unroll
                for (int l = 1; l < 16; ++l) cnt[l] += len == (uint32_t)l;
            }
            if (!canon_make(CD, cnt, false)) { st = GI_ERR_DATA; break; }
            uint32_t pos[16];
#pragma unroll
            for (int l = 0; l < 16; ++l) pos[l] = cn_off(CD, l);
            for (uint32_t s = 0; s < hdist; ++s) {
                const uint32_t len = lens[288 + s];
                if (!len) continue;
                uint32_t at = 0;
#pragma unroll
                for (int l = 1; l < 16; ++l)
                    if (len == (uint32_t)l) { at = pos[l]; ++pos[l]; }
                L[GL_AT(GL_DS + at)] = (uint8_t)s;
            }
        }
        // ---- symbols of one block
        for (;;) {
            B.refill();
            uint32_t cl;
            const uint32_t idx = canon_find(CL, (uint32_t)B.bb, cl);
            if (!cl || idx >= 288u) { st = GI_ERR_DATA; break; }
            B.drop(cl);
            const uint32_t sym = (uint32_t)L[GL_AT(GL_LS + idx)] | (((uint32_t)L[GL_AT(GL_LB + (idx >> 3))] >> (idx & 7u)) & 1u) << 8;
            if (sym < 256u) {
                if (O.o >= usize) { st = GI_ERR_SIZE; break; }
                O.literal(sym);
                continue;
            }
            if (sym == 256u) break;
            if (sym > 285u) { st = GI_ERR_DATA; break; }
            B.refill();
            const uint32_t ls = sym - 257u, le = ls < 8u || ls == 28u ? 0u : (ls - 4u) >> 2;
            const uint32_t length = (ls < 8u ? ls + 3u : ls == 28u ? 258u : ((4u + (ls & 3u)) << le) + 3u) + B.take(le);
            B.refill();
            uint32_t dl;
            const uint32_t didx = canon_find(CD, (uint32_t)B.bb, dl);
            if (!dl || didx >= 32u) { st = GI_ERR_DATA; break; }
            B.drop(dl);
            const uint32_t ds = L[GL_AT(GL_DS + didx)];
            if (ds > 29u) { st = GI_ERR_DATA; break; }
            B.refill();
            const uint32_t de = ds < 4u ? 0u : (ds >> 1) - 1u;
            const uint32_t dist = (ds < 4u ? ds + 1u : ((2u + (ds & 1u)) << de) + 1u) + B.take(de);
            if (dist > O.o || O.o + length > usize) { st = dist > O.o ? GI_ERR_DATA : GI_ERR_SIZE; break; }
            O.match(length, dist);
        }
        if (st || bfinal) break;
    }
    O.finish();
    if (!st && O.o != usize) st = GI_ERR_SIZE;
    if (!st && (uint64_t)(B.in - c0) * 8u - B.bc > (uint64_t)blk.csize * 8u) st = GI_ERR_TRUNC;
    if (bitmap) ntok[bi] = st ? 0u : O.nmatch; // (0: nothing for phase 2 to do)
    if (st) atomicOr(status, st);
}

#include "gpu_inflate_wave.inc"

// ---------------------------------------------------------------------------------------------------
// CRC-32 of the inflated blocks on the card (for readers that keep the inflated bytes there: csrc/gpu_bam.hip)
// ---------------------------------------------------------------------------------------------------
// A wave per block: lane 0 takes the first usize - 63 L bytes, every other lane L = usize / 64 bytes; each computes the standard
// CRC-32 of its piece (table in LDS), then the pieces are joined pairwise — crc(A || B) = crc(A) * x^(8 |B|) mod P  xor  crc(B)
// (the identity behind zlib's crc32_combine; reflected polynomial arithmetic) — where every right-hand piece has the length
// L * 2^level, so one squaring per level gives the multiplier.
namespace {
__device__ __forceinline__ uint32_t crc_mulmod(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
        p ^= (a & (0x80000000u >> i)) ? b : 0u;
        b = (b >> 1) ^ ((b & 1u) ? 0xEDB88320u : 0u);
    }
    return p;
}
// the same at compile time: the constants of k_inflate_resolve's CRC
constexpr uint32_t gi_cx_mulmod(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
        p ^= (a & (0x80000000u >> i)) ? b : 0u;
        b = (b >> 1) ^ ((b & 1u) ? 0xEDB88320u : 0u);
    }
    return p;
}
constexpr uint32_t gi_cx_xpow8(uint32_t n) // x^(8 n) mod P
{
    uint32_t p = 0x80000000u, base = 0x00800000u;
    while (n) {
        if (n & 1u) p = gi_cx_mulmod(p, base);
        base = gi_cx_mulmod(base, base);
        n >>= 1;
    }
    return p;
}
struct GiCrcK {
    static constexpr uint32_t x4096 = gi_cx_xpow8(4096u); // a row of the resolve kernel's gather pass further left
    // x^(32 * 2^lv): level lv of the tree that joins the 1024 columns (a column = 4 bytes)
    __device__ static constexpr uint32_t word_level(int lv)
    {
        constexpr uint32_t t[10] = {gi_cx_xpow8(4u), gi_cx_xpow8(8u), gi_cx_xpow8(16u), gi_cx_xpow8(32u), gi_cx_xpow8(64u), gi_cx_xpow8(128u), gi_cx_xpow8(256u), gi_cx_xpow8(512u),
                                    gi_cx_xpow8(1024u), gi_cx_xpow8(2048u)};
        return t[lv];
    }
    // x^(8 * 2^b), b < 12
    __device__ static uint32_t byte_pow2(uint32_t b)
    {
        constexpr uint32_t t[12] = {gi_cx_xpow8(1u), gi_cx_xpow8(2u), gi_cx_xpow8(4u), gi_cx_xpow8(8u), gi_cx_xpow8(16u), gi_cx_xpow8(32u), gi_cx_xpow8(64u), gi_cx_xpow8(128u),
                                    gi_cx_xpow8(256u), gi_cx_xpow8(512u), gi_cx_xpow8(1024u), gi_cx_xpow8(2048u)};
        uint32_t v = t[0];
#pragma unroll
        for (int k = 1; k < 12; ++k) v = b == (uint32_t)k ? t[k] : v; // (selects: no table in memory)
        return v;
    }
    // 0xFFFFFFFF * x^(8 * 4096 * rows), rows 1 .. 16: what the CRC's start value has become behind 4096 * rows bytes
    __device__ static uint32_t start_times_rows(uint32_t rows)
    {
        constexpr uint32_t t[16] = {gi_cx_mulmod(0xFFFFFFFFu, gi_cx_xpow8(4096u * 1)), gi_cx_mulmod(0xFFFFFFFFu, gi_cx_xpow8(4096u * 2)), gi_cx_mulmod(0xFFFFFFFFu, gi_cx_xpow8(4096u * 3)),
                                    gi_cx_mulmod(0xFFFFFFFFu, gi_cx_xpow8(4096u * 4)), gi_cx_mulmod(0xFFFFFFFFu, gi_cx_xpow8(4096u * 5)), gi_cx_mulmod(0xFFFFFFFFu, gi_cx_xpow8(4096u * 6)),
                                    gi_cx_mulmod(0xFFFFFFFFu, gi_cx_xpow8(4096u * 7)), gi_cx_mulmod(0xFFFFFFFFu, gi_cx_xpow8(4096u * 8)), gi_cx_mulmod(0xFFFFFFFFu, gi_cx_xpow8(4096u * 9)),
                                    gi_cx_mulmod(0xFFFFFFFFu, gi_cx_xpow8(4096u * 10)), gi_cx_mulmod(0xFFFFFFFFu, gi_cx_xpow8(4096u * 11)), gi_cx_mulmod(0xFFFFFFFFu, gi_cx_xpow8(4096u * 12)),
                                    gi_cx_mulmod(0xFFFFFFFFu, gi_cx_xpow8(4096u * 13)), gi_cx_mulmod(0xFFFFFFFFu, gi_cx_xpow8(4096u * 14)), gi_cx_mulmod(0xFFFFFFFFu, gi_cx_xpow8(4096u * 15)),
                                    gi_cx_mulmod(0xFFFFFFFFu, gi_cx_xpow8(4096u * 16))};
        uint32_t v = t[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) v = rows == (uint32_t)(k + 1) ? t[k] : v;
        return v;
    }
};
__device__ uint32_t crc_xpow8(uint32_t n) // x^(8 n) mod P
{
    uint32_t p = 0x80000000u, base = 0x00800000u;
    while (n) { // (n is the same for the whole wave)
        if (n & 1u) p = crc_mulmod(p, base);
        base = crc_mulmod(base, base);
        n >>= 1;
    }
    return p;
}
} // namespace

// ---------------------------------------------------------------------------------------------------
// phase 2: the matches of a block, resolved by pointer jumping
// ---------------------------------------------------------------------------------------------------
// A match says "byte p is byte p - distance"; following that from byte to byte ends at a literal, which phase 1 has written.  Phase 1
// leaves no list of matches: a match's distance and length sit in the first three bytes of its own (still empty) destination, and a
// bitmap with a bit per output byte marks where matches start (1/8 of the output's size; a list of 8-byte tokens was as large as
// the output: allocating and releasing 8 GB of it per run buffer cost the 10 M-read run 0.1 s and the next run 0.6 s).  A
// workgroup of 1024 threads per block keeps ONE 16-bit index per output byte in LDS (128 KB): a literal points at itself, a match
// byte at its source (one pass over the bitmap); then every byte's index is replaced by its index's index — idx[p] = idx[idx[p]] —
// until nothing changes: the length of every chain halves per round, so a block is done in at most 16 rounds whatever its chains look
// like (the dependence chains of the synthetic level-1 streams are ~775 matches long: the two kernels that FOLLOWED them — a round
// per link, between workgroup barriers or through a bitmap of final bytes — took 65 and 83 ms for 45 K blocks).  Reading an index
// that another thread is just replacing is harmless: old and new value are both ancestors.  Last, every match byte is fetched from
// its literal (a gather inside the block's 64 KiB, which the workgroup has just touched) and stored, four bytes per thread.
// The block's CRC-32 is checked HERE (round 4; k_gi_crc below stays for the one-phase kernels): every final byte passes through this
// workgroup's registers in the last pass, four bytes per thread and step at p = 4 tid + 4096 j — a layout a CRC does not like (a
// thread's words lie 4096 bytes apart), taken care of by linearity.  With raw(M) = the CRC register after M from a ZERO start (no
// final inversion): raw(A || B) = raw(A) x^(8|B|) + raw(B) (mod P, reflected arithmetic: crc_mulmod), leading zeros change nothing,
// and the message is the sum of its 1024 "columns" (column t: the words at 4t + 4096j, zeros elsewhere).  So thread t runs a Horner
// scheme down its column — acc = acc * x^(8*4096) + raw(word): the word's raw CRC by slicing-by-4 (four LDS look-ups), the
// multiplication by the constant through four more 256-entry tables — then the columns are joined, acc_t * x^(32 (1023 - t)), in a
// tree: six levels by shuffles, four through LDS, each level's multiplier the square of the one before.  That is raw(M || zeros up to
// the next multiple of 4096), = raw(M) x^(8 pad); and the block's CRC-32 is ~(raw(M) + 0xFFFFFFFF x^(8 |M|)) (the standard start value
// is linear too).  Both sides times x^(8 pad): no division is needed, two powers of x per block by a single lane.  8 KB of tables
// beside the 128 KB of indices; the separate kernel read the 2.9 GB of a 45 K-block launch once more (and fetched every line 4-8 times:
// a lane per kilobyte, profiles/r3_pmc_e2e_gpu_reader.json) for 1.7-3.3 ms.
#define GI_RESOLVE_LDS (131072 + 64)
#define GI_RESOLVE_LDS_CRC (131072 + 64 + 8192 + 128)
__global__ __launch_bounds__(1024) void k_inflate_resolve(const GiBlock* __restrict__ blocks, uint32_t n_blocks, uint8_t* __restrict__ out, const uint32_t* __restrict__ bitmap,
                                                           const uint32_t* __restrict__ ntok, unsigned long long* __restrict__ stats /* nullptr, or (BQC_GI_STATS) counters */,
                                                           const uint32_t* __restrict__ expect /* nullptr: no CRC check here */, uint32_t* __restrict__ status)
{
    extern __shared__ uint16_t ridx[]; // [65536], then (expect) the CRC tables
    const uint32_t bi = blockIdx.x, tid = threadIdx.x;
    if (bi >= n_blocks) return;
    const uint32_t n = ntok[bi];
    if (n == 0u && !expect) return; // no match in this block (or it failed): phase 1 has written all of it
    const GiBlock blk = blocks[bi];
    uint8_t* const o0 = out + blk.uoff;
    const uint32_t usize = blk.usize, upad = (usize + 7u) & ~7u;
    if (usize == 0u) { if (tid == 0u && expect && expect[bi] != 0u) atomicOr(status, (uint32_t)GI_ERR_CRC); return; }
    const uint32_t* const bm = bitmap + gi_bm_base(blocks, bi);
    uint32_t* const pair = (uint32_t*)ridx;
    uint32_t* const ctab = (uint32_t*)(ridx + 65536 + 32);   // [4][256] slicing by 4: ctab[k][b] = raw CRC of byte b followed by k zero bytes
    uint32_t* const mtab = ctab + 1024;                       // [4][256] mtab[k][b] = (b << 8k) * x^(8*4096)
    uint32_t* const wsum = mtab + 1024;                       // [16] the waves' partial results
    long long tc[5] = {stats ? clock64() : 0, 0, 0, 0, 0}; // (BQC_GI_STATS: where the workgroup's clocks go)
    const long long wc0 = stats ? wall_clock64() : 0;
    if (expect) { // the tables (before the first barrier below)
        const uint32_t b = tid & 255u, k = tid >> 8;
        uint32_t c = b;
        for (int i = 0; i < 8; ++i) c = (c >> 1) ^ ((c & 1u) ? 0xEDB88320u : 0u);
        for (uint32_t z = 0; z < k; ++z) { uint32_t c2 = c; for (int i = 0; i < 8; ++i) c2 = (c2 >> 1) ^ ((c2 & 1u) ? 0xEDB88320u : 0u); c = c2; } // k zero bytes behind it
        ctab[k * 256u + b] = c;
        mtab[k * 256u + b] = crc_mulmod(b << (8u * k), GiCrcK::x4096);
    }
    if (n != 0u) {
    for (uint32_t q = tid; q < upad / 2u; q += 1024u) pair[q] = (2u * q) | ((2u * q + 1u) << 16); // every byte its own root
    __syncthreads();
    if (stats) tc[1] = clock64();
    // Match bytes point at their sources: the matches that start in word w of the bitmap.  (Measured, none faster — the kernel is bound
    // by its LDS traffic, not by memory latency: the words' match descriptions fetched together before they are used, +0.4 ms per 45 K
    // blocks; the bytes beyond a match's first eight from a list, a wave per listed match: 78 K clocks for this pass instead of 45 K;
    // the descriptions in LDS and every thread walking its 32 bytes in order: 74 K.)
    for (uint32_t w = tid; w < (usize + 31u) / 32u; w += 1024u) {
        uint32_t bits = bm[w];
        while (bits) {
            const uint32_t d = 32u * w + (uint32_t)__ffs((int)bits) - 1u;
            bits &= bits - 1u;
            if (d + 3u > usize) continue;
            // phase 1 left the match's description in its first three bytes: one load of four where the block has them (three byte loads per match before)
            const uint32_t tok = d + 4u <= usize ? *(const gi_u32_u*)(o0 + d) & 0xFFFFFFu : (uint32_t)o0[d] | ((uint32_t)o0[d + 1u] << 8) | ((uint32_t)o0[d + 2u] << 16);
            const uint32_t dist = (tok & 0x7FFFu) + 1u, len = (tok >> 16) + 3u;
            if (dist > d || d + len > usize) continue; // (phase 1 has checked it; damaged memory must not reach outside)
            for (uint32_t j = 0; j < len; ++j) ridx[d + j] = (uint16_t)(d + j - dist);
        }
    }
    __syncthreads();
    if (stats) tc[2] = clock64();
    uint32_t rounds = 0;
    for (; rounds < 16u; ++rounds) { // (skipping the pairs that already point at roots — a mask per thread — changed nothing measurable)
        uint32_t changed = 0, nchg = 0;
#pragma unroll 4
        for (uint32_t q = tid; q < upad / 2u; q += 1024u) {
            const uint32_t v = pair[q], a = v & 0xFFFFu, b = v >> 16;
            const uint32_t a2 = ridx[a], b2 = ridx[b];
            changed |= (a2 ^ a) | (b2 ^ b);
            if (stats) nchg += (a2 != a ? 1u : 0u) + (b2 != b ? 1u : 0u);
            pair[q] = a2 | (b2 << 16);
        }
        if (stats) { // (BQC_GI_STATS: bytes whose index still moved in this round)
            for (int o = 32; o > 0; o >>= 1) nchg += __shfl_xor(nchg, o);
            if ((tid & 63u) == 0u && nchg) atomicAdd(&stats[21 + (rounds < 8u ? rounds : 8u)], (unsigned long long)nchg);
        }
        if (!__syncthreads_or(changed ? 1 : 0)) break;
    }
    if (stats) { tc[3] = clock64(); if (tid == 0u) atomicAdd(&stats[2], (unsigned long long)rounds); }
    } else __syncthreads(); // (a block without matches: only its CRC is looked at; the tables are complete behind this barrier)
    // every match byte from its literal: four bytes per thread and step (the block's tail byte by byte).
    // Round 4: the literals are gathered from LDS, not from memory.  A thread first takes the roots of its (up to 16) steps into
    // registers; then the index array — done with — is overwritten with the block's BYTES as phase 1 left them (coalesced 16-byte
    // loads: the literals are in place, what lies under the matches is never looked at); then four ds_read_u8 per word instead of four
    // byte loads from memory, which were 64 scattered addresses per instruction (~40 clocks of the CU's address path each, 1024 such
    // instructions per block: 31 K of the kernel's 150 K clocks per block in round 3).  The words that are all literals come from the same
    // copy (the CRC wants them too).
    uint32_t acc = 0;
#define GI_CRC_COL(x) do { if (expect) { const uint32_t w_ = (x); \
        acc = mtab[acc & 255u] ^ mtab[256u + ((acc >> 8) & 255u)] ^ mtab[512u + ((acc >> 16) & 255u)] ^ mtab[768u + (acc >> 24)] ^ \
              ctab[768u + (w_ & 255u)] ^ ctab[512u + ((w_ >> 8) & 255u)] ^ ctab[256u + ((w_ >> 16) & 255u)] ^ ctab[w_ >> 24]; } } while (0)
    if (n != 0u) {
        uint32_t r0[16], r1[16]; // the roots of this thread's words: step j at p = 4 tid + 4096 j
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t p = 4u * tid + 4096u * (uint32_t)j;
            if (p < usize) { r0[j] = pair[p / 2u]; r1[j] = pair[p / 2u + 1u]; } else { r0[j] = 0; r1[j] = 0; }
        }
        __syncthreads();
        uint8_t* const val = (uint8_t*)ridx; // [65536] the block's bytes
        for (uint32_t off = 16u * tid; off < usize; off += 16384u) *(gi_u32x4*)(val + off) = *(const gi_u32x4_u*)(o0 + off); // (runs up to 15 bytes over the block's end: inside the output buffer and its slack)
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t p = 4u * tid + 4096u * (uint32_t)j;
            if (p >= usize) break;
            const uint32_t i0 = r0[j] & 0xFFFFu, i1 = r0[j] >> 16, i2 = r1[j] & 0xFFFFu, i3 = r1[j] >> 16;
            if (p + 4u <= usize) {
                if (i0 == p && i1 == p + 1u && i2 == p + 2u && i3 == p + 3u) { GI_CRC_COL(*(const uint32_t*)(val + p)); continue; } // four literals: in place already
                const uint32_t w = (uint32_t)val[i0] | ((uint32_t)val[i1] << 8) | ((uint32_t)val[i2] << 16) | ((uint32_t)val[i3] << 24);
                *(gi_u32_u*)(o0 + p) = w;
                GI_CRC_COL(w);
            } else {
                const uint32_t ii[4] = {i0, i1, i2, i3};
                uint32_t w = 0;
                for (uint32_t k = 0; p + k < usize; ++k) {
                    const uint32_t by = val[ii[k]];
                    if (ii[k] != p + k) o0[p + k] = (uint8_t)by;
                    w |= by << (8u * k);
                }
                GI_CRC_COL(w); // (zeros behind the block's end)
            }
        }
    } else {
        for (uint32_t p = 4u * tid; p < usize; p += 4096u) { // no match in the block: only its CRC
            if (p + 4u <= usize) GI_CRC_COL(*(const gi_u32_u*)(o0 + p));
            else { uint32_t w = 0; for (uint32_t k = 0; p + k < usize; ++k) w |= (uint32_t)o0[p + k] << (8u * k); GI_CRC_COL(w); }
        }
    }
#undef GI_CRC_COL
    if (expect) {
        // rows this thread has not met (its column lies behind the block's end in the last row): the zero word of that row
        const uint32_t rows = (usize + 4095u) >> 12;
        if (4u * tid + 4096u * (rows - 1u) >= usize) acc = mtab[acc & 255u] ^ mtab[256u + ((acc >> 8) & 255u)] ^ mtab[512u + ((acc >> 16) & 255u)] ^ mtab[768u + (acc >> 24)];
        // the columns joined: c = c_left * x^(32 * 2^level) + c_right
        uint32_t c = acc;
#pragma unroll
        for (int lv = 0; lv < 6; ++lv) {
            const uint32_t right = __shfl_down(c, 1u << lv, 64);
            c = crc_mulmod(c, GiCrcK::word_level(lv)) ^ right; // (only the lanes whose index is a multiple of 2 << lv hold a joined piece; the others' values are not used)
        }
        if ((tid & 63u) == 0u) wsum[tid >> 6] = c;
        __syncthreads();
        if (tid < 64u) {
            c = tid < 16u ? wsum[tid] : 0u;
#pragma unroll
            for (int lv = 6; lv < 10; ++lv) {
                const uint32_t right = __shfl_down(c, 1u << (lv - 6), 64);
                c = crc_mulmod(c, GiCrcK::word_level(lv)) ^ right;
            }
            // c (lane 0) = raw(M || zeros up to 4096 * rows) = raw(M) x^(8 pad).  x^(8 pad), pad < 4096: the product of x^(8 * 2^b) over
            // the bits b of pad — twelve factors in twelve lanes, multiplied in a tree
            const uint32_t pad = 4096u * rows - usize;
            uint32_t f = tid < 12u && ((pad >> tid) & 1u) ? GiCrcK::byte_pow2(tid) : 0x80000000u; // (0x80000000 = 1)
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) f = crc_mulmod(f, __shfl_down(f, d, 64));
            if (tid == 0u) {
                const uint32_t lhs = c ^ GiCrcK::start_times_rows(rows);   // + 0xFFFFFFFF x^(8 * 4096 * rows)
                const uint32_t rhs = crc_mulmod(~expect[bi], f);
                if (lhs != rhs) atomicOr(status, (uint32_t)GI_ERR_CRC);
            }
        }
    }
    if (stats && n != 0u) {
        __syncthreads();
        if (tid == 0u) {
            tc[4] = clock64();
            atomicAdd(&stats[0], 1ull); atomicAdd(&stats[7], (unsigned long long)n);
            for (int k = 0; k < 4; ++k) atomicAdd(&stats[8 + k], (unsigned long long)(tc[k + 1] - tc[k]));
            atomicAdd(&stats[12], (unsigned long long)(wall_clock64() - wc0));
        }
    }
}

__global__ __launch_bounds__(256) void k_gi_crc(const uint8_t* __restrict__ out, const GiBlock* __restrict__ blocks, const uint32_t* __restrict__ expect,
                                                  uint32_t n_blocks, uint32_t* __restrict__ status)
{
    __shared__ uint32_t tab[4][256]; // slicing by 4: tab[k][b] = CRC of byte b followed by k zero bytes — four independent look-ups per word
    {
        uint32_t c = threadIdx.x;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? 0xEDB88320u : 0u);
        tab[0][threadIdx.x] = c;
    }
    __syncthreads();
    for (int k = 1; k < 4; ++k) {
        const uint32_t p = tab[k - 1][threadIdx.x];
        tab[k][threadIdx.x] = (p >> 8) ^ tab[0][p & 255u];
        __syncthreads();
    }
    const uint32_t bi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (bi >= n_blocks) return;
    const GiBlock blk = blocks[bi];
    const uint32_t L = blk.usize >> 6, first = blk.usize - 63u * L;
    const uint8_t* p = out + blk.uoff + (lane ? first + (lane - 1u) * L : 0u);
    const uint32_t n = lane ? L : first;
    uint32_t c = 0xFFFFFFFFu, k = 0;
#define GI_CRC_WORD(x) { const uint32_t w = c ^ (x); c = tab[3][w & 255u] ^ tab[2][(w >> 8) & 255u] ^ tab[1][(w >> 16) & 255u] ^ tab[0][w >> 24]; }
    for (; k + 16 <= n; k += 16) { // (a lane's piece is its own kilobyte: one 16-byte load per four steps)
        const gi_u32x4 v = *(const gi_u32x4_u*)(p + k);
        GI_CRC_WORD(v.x) GI_CRC_WORD(v.y) GI_CRC_WORD(v.z) GI_CRC_WORD(v.w)
    }
    for (; k + 4 <= n; k += 4) GI_CRC_WORD(*(const gi_u32_u*)(p + k))
    for (; k < n; ++k) c = tab[0][(c ^ p[k]) & 255u] ^ (c >> 8);
    c = ~c;
    uint32_t x = crc_xpow8(L); // x^(8 L): the multiplier for a right-hand piece of L bytes; squared level by level
#pragma unroll
    for (int lv = 0; lv < 6; ++lv) {
        const uint32_t right = __shfl_down(c, 1u << lv, 64);
        c = crc_mulmod(c, x) ^ right; // (only the lanes whose index is a multiple of 2 << lv hold a joined piece; the others' values are not used)
        x = crc_mulmod(x, x);
    }
    if (lane == 0 && c != expect[bi]) atomicOr(status, (uint32_t)GI_ERR_CRC);
}

// launches on device-resident operands (csrc/gpu_bam.hip): blocks[i].coff into comp, .uoff into out (the blocks' outputs back to
// back, total_out bytes from the first block's start to the last one's end); d_tok: 4 bytes x bqc_gpu_inflate_token_words() of
// scratch (the bitmap of match starts), d_ntok: 4 bytes per block
extern "C" int bqc_gpu_inflate_two_phase() { return getenv("BQC_GI_TWO_PHASE") && atoi(getenv("BQC_GI_TWO_PHASE")) == 0 ? 0 : 1; } // (default: two phases)
// 4-byte words of scratch for a launch: the bitmap of match starts, a bit per output byte and a word of slack per block (64 words when
// the two phases are off)
extern "C" size_t bqc_gpu_inflate_token_words(size_t inflated_bytes, size_t n_blocks) { return bqc_gpu_inflate_two_phase() ? inflated_bytes / 32 + n_blocks + 64 : 64; }

extern "C" void bqc_gpu_inflate_launch(const uint8_t* d_comp, const GiBlock* d_blocks, uint32_t n_blocks, uint64_t total_out, uint8_t* d_out, const uint32_t* d_crc,
                                       uint32_t* d_status, uint32_t* d_tok, uint32_t* d_ntok, void* stream)
{
    if (!n_blocks) return;
    const int lanes = getenv("BQC_GI_LANES") ? atoi(getenv("BQC_GI_LANES")) : 4; // (read at every launch: the tests switch kernels; 10 K blocks: 24 ms with 4 blocks per workgroup, 29 with 8, 34 with 16, 30 with 2)
    // more blocks than the kernel with root tables holds at once (72 per CU): the lean kernel takes them in one go; for fewer blocks the
    // root tables are faster.  BQC_GI_LEAN: 0 never, N always with N blocks per workgroup
    const int lean_env = getenv("BQC_GI_LEAN") ? atoi(getenv("BQC_GI_LEAN")) : -1;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inflate_resolve), hipFuncAttributeMaxDynamicSharedMemorySize, GI_RESOLVE_LDS_CRC);
    (void)attr;
    const bool two_phase = bqc_gpu_inflate_two_phase() != 0; // (see the comment at struct Out)
    if (!two_phase || !d_ntok) d_tok = nullptr;
    // (two phases, inflate + CRC: 9 K blocks 16 ms with the root tables against 22 lean; 18 K: 34 against 24; 45 K: 63 against 36; 90 K: lean 64-wide 56 against 68 32-wide)
    const int lean = lean_env >= 0 ? lean_env : (n_blocks > 60000u ? 64 : n_blocks > 12000u ? 32 : 0);
    uint32_t* const d_tok64 = d_tok; // (the bitmap of match starts: zeroed for every launch, phase 1 only stores the words that are not zero)
    if (d_tok) (void)hipMemsetAsync(d_tok, 0, 4 * (size_t)bqc_gpu_inflate_token_words(total_out, n_blocks), (hipStream_t)stream);
#define GI_LAUNCH_L(NL) hipLaunchKernelGGL(k_inflate_lean<NL>, dim3((n_blocks + NL - 1) / NL), dim3(NL), GL_BYTES * NL, (hipStream_t)stream, d_comp, d_blocks, n_blocks, d_out, d_status, d_tok64, d_ntok)
#define GI_LAUNCH_D(NL) hipLaunchKernelGGL(k_inflate<NL>, dim3((n_blocks + NL - 1) / NL), dim3(NL), GI_U16 * 2 * NL, (hipStream_t)stream, d_comp, d_blocks, n_blocks, d_out, d_status, d_tok64, d_ntok)
    static unsigned long long* d_stats = nullptr; // (BQC_GI_STATS: blocks resolved, chunks, rounds, matches, matched bytes, distances < 8, lengths > 32, tokens — printed at exit)
    static const bool want_stats = [] {
        if (!getenv("BQC_GI_STATS")) return false;
        if (hipMalloc((void**)&d_stats, 256) != hipSuccess || hipMemset(d_stats, 0, 256) != hipSuccess) { d_stats = nullptr; return false; }
        atexit([] {
            unsigned long long h[32] = {};
            if (hipMemcpy(h, d_stats, 256, hipMemcpyDeviceToHost) == hipSuccess) {
                if (h[4]) fprintf(stderr, "[gpu inflate] header clocks per deflate block: block header (lane 0) %.0f, code-length code built %.0f, code lengths (lane 0) %.0f, distance code built %.0f, literal/length code built %.0f\n",
                                  (double)h[16] / h[4], (double)h[17] / h[4], (double)h[18] / h[4], (double)h[19] / h[4], (double)h[20] / h[4]);
                fprintf(stderr, "[gpu inflate] resolve clocks per block: set-up %.0f, matches listed %.0f, pointer jumping %.0f, gather %.0f\n", (double)h[8] / (h[0] ? h[0] : 1),
                        (double)h[9] / (h[0] ? h[0] : 1), (double)h[10] / (h[0] ? h[0] : 1), (double)h[11] / (h[0] ? h[0] : 1));
                fprintf(stderr, "[gpu inflate] clock64 ticks per wall_clock64 tick (100 MHz): %.2f\n", (double)(h[8] + h[9] + h[10] + h[11]) / (h[12] ? h[12] : 1));
                if (h[0]) fprintf(stderr, "[gpu inflate] bytes per block whose index moved in jumping round 1, 2, ..: %.0f %.0f %.0f %.0f %.0f %.0f %.0f %.0f, later %.0f\n", (double)h[21] / h[0], (double)h[22] / h[0],
                                  (double)h[23] / h[0], (double)h[24] / h[0], (double)h[25] / h[0], (double)h[26] / h[0], (double)h[27] / h[0], (double)h[28] / h[0], (double)h[29] / h[0]);
            }
            if (h[0] || h[4])
                fprintf(stderr, "[gpu inflate] resolve: %llu blocks, %llu matches, %llu rounds; phase 1 by waves: %llu deflate blocks, %llu scan rounds; clocks per wave: header + tables %.0f, scan %.0f, write %.0f\n", h[0], h[7], h[2], h[4], h[3],
                        (double)h[5] / (h[4] ? h[4] : 1), (double)h[6] / (h[4] ? h[4] : 1), (double)h[1] / (h[4] ? h[4] : 1));
        });
        return true;
    }();
    (void)want_stats;
    // phase 1 by a wave per block (gpu_inflate_wave.inc) unless BQC_GI_WAVE=0 (a lane per block: the kernels above); one phase: always a lane per block
    const bool wave = d_tok && !(getenv("BQC_GI_WAVE") && atoi(getenv("BQC_GI_WAVE")) == 0) && lean_env < 0 && !getenv("BQC_GI_LANES");
    if (wave) hipLaunchKernelGGL(k_inflate_wave, dim3(getenv("BQC_GI_WAVE_GRID") && atoi(getenv("BQC_GI_WAVE_GRID")) > 0 ? std::min<uint32_t>(n_blocks, (uint32_t)atoi(getenv("BQC_GI_WAVE_GRID"))) : n_blocks), dim3(64), 0, (hipStream_t)stream, d_comp, d_blocks, n_blocks, d_out, d_status, d_tok64, d_ntok, d_stats);
    else if (lean) { if (lean == 8) GI_LAUNCH_L(8); else if (lean == 32) GI_LAUNCH_L(32); else if (lean == 16) GI_LAUNCH_L(16); else GI_LAUNCH_L(64); }
    else if (lanes == 2) GI_LAUNCH_D(2); else if (lanes == 4) GI_LAUNCH_D(4); else if (lanes == 16) GI_LAUNCH_D(16); else GI_LAUNCH_D(8);
    const bool no_resolve = getenv("BQC_GI_NO_RESOLVE") != nullptr; // (timing experiments: phase 1 alone; the output then lacks its matches)
    // the CRC-32s are checked by the resolve kernel, which has every final byte in its hands (BQC_GI_CRC_KERNEL=1, or one phase: k_gi_crc)
    const bool crc_in_resolve = d_crc && d_tok && !no_resolve && !(getenv("BQC_GI_CRC_KERNEL") && atoi(getenv("BQC_GI_CRC_KERNEL")) == 1);
    if (d_tok && !no_resolve) hipLaunchKernelGGL(k_inflate_resolve, dim3(n_blocks), dim3(1024), crc_in_resolve ? GI_RESOLVE_LDS_CRC : GI_RESOLVE_LDS, (hipStream_t)stream, d_blocks, n_blocks, d_out, d_tok64,
                                                 d_ntok, d_stats, crc_in_resolve ? d_crc : nullptr, d_status);
    if (d_crc && !crc_in_resolve) hipLaunchKernelGGL(k_gi_crc, dim3((n_blocks + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_out, d_blocks, d_crc, n_blocks, d_status);
}

// ---------------------------------------------------------------------------------------------------
// host side: one object per reader thread
// ---------------------------------------------------------------------------------------------------
struct GpuInflater {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr; // blocking: the waiting host thread sleeps (several readers wait at once)
    uint8_t* d_comp = nullptr; size_t comp_cap = 0;
    uint8_t* d_out = nullptr; size_t out_cap = 0;
    GiBlock* d_blocks = nullptr; size_t blocks_cap = 0;
    uint32_t* d_tok = nullptr; size_t tok_cap = 0;   // the two phases' scratch (bytes)
    uint32_t* d_ntok = nullptr; size_t ntok_cap = 0;
    uint32_t* d_status = nullptr;
    uint32_t* h_status = nullptr; // page-locked
};

extern "C" GpuInflater* bqc_gpu_inflater_create(int device)
{
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    auto* g = new GpuInflater();
    g->device = device;
    if (e == hipSuccess && (device < 0 || device >= ndev)) e = hipErrorInvalidDevice;
    if (e == hipSuccess) e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->done, hipEventBlockingSync | hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&g->d_status, 64);
    if (e == hipSuccess) e = hipHostMalloc((void**)&g->h_status, 64, hipHostMallocDefault);
    if (e != hipSuccess) {
        if (getenv("BQC_GI_TIMING")) fprintf(stderr, "[gpu inflate] not available: %s\n", hipGetErrorString(e));
        if (g->stream) (void)hipStreamDestroy(g->stream);
        if (g->done) (void)hipEventDestroy(g->done);
        (void)hipFree(g->d_status);
        if (g->h_status) (void)hipHostFree(g->h_status);
        delete g;
        return nullptr;
    }
    return g;
}

extern "C" void bqc_gpu_inflater_destroy(GpuInflater* g)
{
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->stream) { (void)hipStreamSynchronize(g->stream); (void)hipStreamDestroy(g->stream); }
    if (g->done) (void)hipEventDestroy(g->done);
    (void)hipFree(g->d_comp); (void)hipFree(g->d_out); (void)hipFree(g->d_blocks); (void)hipFree(g->d_status); (void)hipFree(g->d_tok); (void)hipFree(g->d_ntok);
    if (g->h_status) (void)hipHostFree(g->h_status);
    delete g;
}

// Inflates n_blocks blocks: comp[blocks[i].coff, + csize) -> out[blocks[i].uoff, + usize).  0 = every block inflated to exactly its
// usize bytes; > 0: GI_ERR_* bits of the blocks that did not (corrupt data); < 0: the GPU could not be used (fall back to the CPU).
extern "C" int bqc_gpu_inflate(GpuInflater* g, const uint8_t* comp, size_t comp_bytes, const GiBlock* blocks, size_t n_blocks, uint8_t* out, size_t out_bytes)
{
    if (!g) return -1;
    if (!n_blocks) return 0;
    if (hipSetDevice(g->device) != hipSuccess) return -1;
    auto grow = [](auto*& p, size_t& cap, size_t need) {
        if (cap >= need) return true;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        const size_t c = need + need / 4 + 4096;
        if (hipMalloc((void**)&p, c) != hipSuccess) { p = nullptr; return false; }
        cap = c;
        return true;
    };
    size_t bcap_bytes = g->blocks_cap * sizeof(GiBlock);
    if (!grow(g->d_comp, g->comp_cap, comp_bytes + 64) || !grow(g->d_out, g->out_cap, out_bytes + 64)) return -1;
    if (!grow(g->d_blocks, bcap_bytes, n_blocks * sizeof(GiBlock))) return -1;
    g->blocks_cap = bcap_bytes / sizeof(GiBlock);
    if (!grow(g->d_tok, g->tok_cap, 4 * bqc_gpu_inflate_token_words(out_bytes, n_blocks)) || !grow(g->d_ntok, g->ntok_cap, 4 * n_blocks + 64)) return -1;
    static const bool timing = getenv("BQC_GI_TIMING") != nullptr;
    hipEvent_t ev[4] = {};
    if (timing) for (auto& e : ev) (void)hipEventCreate(&e);
    if (timing) (void)hipEventRecord(ev[0], g->stream);
    if (hipMemsetAsync(g->d_status, 0, 4, g->stream) != hipSuccess) return -1;
    if (hipMemcpyAsync(g->d_comp, comp, comp_bytes, hipMemcpyHostToDevice, g->stream) != hipSuccess) return -1;
    if (hipMemcpyAsync(g->d_blocks, blocks, n_blocks * sizeof(GiBlock), hipMemcpyHostToDevice, g->stream) != hipSuccess) return -1;
    if (timing) (void)hipEventRecord(ev[1], g->stream);
    bqc_gpu_inflate_launch(g->d_comp, g->d_blocks, (uint32_t)n_blocks, out_bytes, g->d_out, nullptr, g->d_status, g->d_tok, g->d_ntok, g->stream); // (the CRC-32s are checked on the host here)
    if (timing) (void)hipEventRecord(ev[2], g->stream);
    if (hipMemcpyAsync(out, g->d_out, out_bytes, hipMemcpyDeviceToHost, g->stream) != hipSuccess) return -1;
    if (hipMemcpyAsync(g->h_status, g->d_status, 4, hipMemcpyDeviceToHost, g->stream) != hipSuccess) return -1;
    if (timing) (void)hipEventRecord(ev[3], g->stream);
    if (hipEventRecord(g->done, g->stream) != hipSuccess || hipEventSynchronize(g->done) != hipSuccess) return -1;
    if (timing) {
        float a = 0, b = 0, c = 0;
        (void)hipEventElapsedTime(&a, ev[0], ev[1]); (void)hipEventElapsedTime(&b, ev[1], ev[2]); (void)hipEventElapsedTime(&c, ev[2], ev[3]);
        fprintf(stderr, "[gpu inflate] %zu blocks, %.1f MB -> %.1f MB: H2D %.2f ms, kernel %.2f ms, D2H %.2f ms\n", n_blocks, comp_bytes / 1e6, out_bytes / 1e6, a, b, c);
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    return (int)*g->h_status;
}
