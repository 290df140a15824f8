// inflate_fast.cpp — see inflate_fast.h
#include "inflate_fast.h"

#include <cstring>

namespace {

// table entry:  [31:16] payload: literal ([23:16], and a second one in [31:24]) / length base / offset base / precode symbol /
//                       subtable start
//               [15] exceptional (one test covers end-of-block, subtable pointer and invalid code)
//               [14] subtable pointer   [13] invalid   [12] end of block
//               [11:8] code bits at this level of a length / offset entry, or index bits of the subtable of a pointer
//               [6] two literals   [5] literal
//               [4:0] bits to consume at this level: code bits (root bits for a pointer; both codes of a literal pair) plus,
//                     for a length / offset, its extra bits, which are then taken from the saved bit buffer in one step
const uint32_t E_LIT = 0x20u, E_LIT2 = 0x40u, E_EXC = 0x8000u, E_SUB = 0x4000u, E_INVALID = 0x2000u, E_EOB = 0x1000u, E_BITS = 31u;

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kOffBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kOffExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kPreOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

enum Kind { PRECODE, LITLEN, OFFSET };

inline uint32_t rev_bits(uint32_t v, int n) // the low n bits of v, reversed
{
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

inline uint32_t make_entry(int sym, Kind kind)
{
    if (kind == PRECODE) return (uint32_t)sym << 16;
    if (kind == LITLEN) {
        if (sym < 256) return E_LIT | ((uint32_t)sym << 16);
        if (sym == 256) return E_EXC | E_EOB;
        if (sym > 285) return E_EXC | E_INVALID;
        return ((uint32_t)kLenBase[sym - 257] << 16) | ((uint32_t)kLenExtra[sym - 257] << 8);
    }
    if (sym > 29) return E_EXC | E_INVALID;
    return ((uint32_t)kOffBase[sym] << 16) | ((uint32_t)kOffExtra[sym] << 8);
}

// completes a symbol's entry with the code bits at its table level: literals / precode symbols / end of block consume the
// code; lengths and offsets consume code + extra bits and remember where the extra bits start
inline uint32_t with_bits(uint32_t ent, uint32_t code_bits)
{
    if (ent & (E_LIT | E_EXC)) return ent | code_bits;
    const uint32_t extra = (ent >> 8) & 15u;
    return (ent & 0xFFFF0000u) | (code_bits << 8) | (code_bits + extra);
}

// Canonical Huffman decode table from code lengths (0 = unused), zlib's acceptance rules: an over-subscribed set is an
// error; an incomplete set is an error unless it is a single code of length 1 (or, for a table that is then never
// used, no code at all).  Slots no code reaches decode as E_INVALID.
bool build_table(uint32_t* table, int root, int enough, const uint8_t* lens, int nsyms, Kind kind)
{
    int count[16] = {0};
    for (int s = 0; s < nsyms; ++s) ++count[lens[s]];
    int maxl = 15;
    while (maxl > 0 && !count[maxl]) --maxl;
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return false;
    }
    if (left > 0 && maxl > 1) return false;
    if (left > 0 && kind == PRECODE) return false;
    const uint32_t inv = E_EXC | E_INVALID | 1u;
    for (int i = 0; i < (1 << root); ++i) table[i] = inv;
    if (maxl == 0) return true;
    // symbols in canonical order with their codes (MSB first)
    uint16_t sorted[320], code_of[320];
    int offs[17];
    offs[1] = 0;
    for (int l = 1; l <= 15; ++l) offs[l + 1] = offs[l] + count[l];
    const int ncodes = offs[16];
    {
        int pos[16];
        for (int l = 1; l <= 15; ++l) pos[l] = offs[l];
        for (int s = 0; s < nsyms; ++s) if (lens[s]) sorted[pos[lens[s]]++] = (uint16_t)s;
        uint32_t code = 0;
        int k = 0;
        for (int l = 1; l <= 15; ++l) {
            for (int c = 0; c < count[l]; ++c) code_of[k++] = (uint16_t)code++;
            code <<= 1;
        }
    }
    int next_free = 1 << root;
    int sub_start = 0, sub_bits = 0;
    uint32_t cur_prefix = 0xFFFFFFFFu;
    for (int k = 0; k < ncodes; ++k) {
        const int sym = sorted[k], l = lens[sym];
        const uint32_t code = code_of[k], ent = make_entry(sym, kind);
        if (l <= root) {
            const uint32_t r = rev_bits(code, l);
            const uint32_t full = with_bits(ent, (uint32_t)l);
            for (uint32_t i = r; i < (1u << root); i += 1u << l) table[i] = full;
            continue;
        }
        const uint32_t prefix = code >> (l - root);
        if (prefix != cur_prefix) { // codes with one root prefix are adjacent in canonical order: size the subtable by the longest
            cur_prefix = prefix;
            int longest = l;
            for (int k2 = k + 1; k2 < ncodes; ++k2) {
                const int l2 = lens[sorted[k2]];
                if ((uint32_t)(code_of[k2] >> (l2 - root)) != prefix) break;
                longest = l2;
            }
            sub_bits = longest - root;
            sub_start = next_free;
            next_free += 1 << sub_bits;
            if (next_free > enough) return false;
            for (int i = 0; i < (1 << sub_bits); ++i) table[sub_start + i] = inv;
            table[rev_bits(prefix, root)] = E_EXC | E_SUB | ((uint32_t)sub_start << 16) | ((uint32_t)sub_bits << 8) | (uint32_t)root;
        }
        const int rem = l - root;
        const uint32_t r = rev_bits(code & ((1u << rem) - 1u), rem);
        const uint32_t full = with_bits(ent, (uint32_t)rem);
        for (uint32_t i = r; i < (1u << sub_bits); i += 1u << rem) table[sub_start + i] = full;
    }
    return true;
}

// Root entries whose first code is a literal and leaves room for a second complete literal code decode both at once
// (base qualities and names are literal-heavy: the symbol-to-symbol dependency through the bit buffer is what bounds the rate).
void pair_literals(uint32_t* table, int root)
{
    uint32_t single[1 << Inflater::kLitBits];
    memcpy(single, table, sizeof(uint32_t) << root);
    for (uint32_t i = 0; i < (1u << root); ++i) {
        const uint32_t e1 = single[i];
        if (!(e1 & E_LIT)) continue;
        const uint32_t l1 = e1 & E_BITS;
        if (l1 >= (uint32_t)root) continue;
        const uint32_t e2 = single[i >> l1]; // the following bits, zero-extended: decisive only if the code fits into them
        if (!(e2 & E_LIT) || (e2 & E_BITS) > (uint32_t)root - l1) continue;
        table[i] = E_LIT | E_LIT2 | (e1 & 0x00FF0000u) | ((e2 & 0x00FF0000u) << 8) | (l1 + (e2 & E_BITS));
    }
}

struct FixedTables {
    uint32_t lit[Inflater::kLitEnough], off[Inflater::kOffEnough];
    bool ok;
    FixedTables()
    {
        uint8_t l[288], d[32];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        for (int i = 0; i < 32; ++i) d[i] = 5;
        ok = build_table(lit, Inflater::kLitBits, Inflater::kLitEnough, l, 288, LITLEN) &&
             build_table(off, Inflater::kOffBits, Inflater::kOffEnough, d, 32, OFFSET);
        if (ok) pair_literals(lit, Inflater::kLitBits);
    }
};

inline uint64_t load64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; } // little-endian host
inline void store64(uint8_t* p, uint64_t v) { memcpy(p, &v, 8); }

} // namespace

bool Inflater::run(const uint8_t* in0, size_t in_n, uint8_t* out0, size_t out_n)
{
    static const FixedTables fixed; // thread-safe one-time initialisation
    if (!fixed.ok) return false;
    const uint8_t* in = in0;              // a VIRTUAL position once it passes in_end: the stream continues with the 8
    const uint8_t* const in_end = in0 + in_n; // readable trailer bytes and then zeros; consuming any of them is an error (checked at the end)
    uint8_t* out = out0;
    uint8_t* const out_end = out0 + out_n;
    uint64_t bb = 0; // bit buffer, next bit at bit 0
    unsigned bc = 0; // valid bits in bb
    const size_t kFastRoom = 258 + 16 + 8; // longest match + the overshoot of its word-wise copy + literals decoded alongside

#define REFILL()                                                                  \
    do {                                                                          \
        if (in <= in_end) { bb |= load64(in) << bc; in += (63u - bc) >> 3; bc |= 56u; } \
        else while (bc <= 55u) { bc += 8u; ++in; }                                \
    } while (0)
#define CONSUME(n) do { bb >>= (n); bc -= (n); } while (0)

    for (;;) {
        REFILL();
        const unsigned bfinal = (unsigned)bb & 1u, btype = ((unsigned)bb >> 1) & 3u;
        CONSUME(3);
        if (btype == 0) { // stored: back to a byte boundary, LEN / NLEN, raw bytes
            CONSUME(bc & 7u);
            in -= bc >> 3;
            bb = 0; bc = 0;
            if (in > in_end || (size_t)(in_end - in) < 4) return false;
            const unsigned len = in[0] | (in[1] << 8), nlen = in[2] | (in[3] << 8);
            if ((len ^ nlen) != 0xFFFFu) return false;
            in += 4;
            if (len > (size_t)(in_end - in) || len > (size_t)(out_end - out)) return false;
            memcpy(out, in, len);
            in += len; out += len;
            if (bfinal) break;
            continue;
        }
        const uint32_t* lit;
        const uint32_t* off;
        if (btype == 1) { lit = fixed.lit; off = fixed.off; }
        else if (btype == 2) {
            const unsigned hlit = ((unsigned)bb & 31u) + 257u, hdist = (((unsigned)bb >> 5) & 31u) + 1u, hclen = (((unsigned)bb >> 10) & 15u) + 4u;
            CONSUME(14);
            if (hlit > 286 || hdist > 30) return false; // zlib: too many length or distance symbols
            uint8_t pl[19] = {0};
            for (unsigned i = 0; i < hclen; ++i) {
                if (bc < 3) REFILL();
                pl[kPreOrder[i]] = (uint8_t)(bb & 7u);
                CONSUME(3);
            }
            if (!build_table(pre_, kPreBits, 1 << kPreBits, pl, 19, PRECODE)) return false;
            const unsigned total = hlit + hdist;
            unsigned i = 0;
            while (i < total) {
                REFILL(); // up to 7 + 7 bits below
                const uint32_t e = pre_[bb & ((1u << kPreBits) - 1u)];
                if (e & E_EXC) return false;
                CONSUME(e & E_BITS);
                const unsigned sym = e >> 16;
                if (sym < 16) { lens_[i++] = (uint8_t)sym; continue; }
                unsigned rep, val = 0;
                if (sym == 16) {
                    if (i == 0) return false;
                    val = lens_[i - 1];
                    rep = 3 + ((unsigned)bb & 3u); CONSUME(2);
                } else if (sym == 17) { rep = 3 + ((unsigned)bb & 7u); CONSUME(3); }
                else { rep = 11 + ((unsigned)bb & 127u); CONSUME(7); }
                if (i + rep > total) return false;
                memset(lens_ + i, (int)val, rep);
                i += rep;
            }
            if (lens_[256] == 0) return false; // no end-of-block code
            if (!build_table(lit_, kLitBits, kLitEnough, lens_, (int)hlit, LITLEN)) return false;
            if (!build_table(off_, kOffBits, kOffEnough, lens_ + hlit, (int)hdist, OFFSET)) return false;
            pair_literals(lit_, kLitBits);
            lit = lit_; off = off_;
        } else return false;

        // ---- symbols of one block
        for (;;) {
            REFILL();
            uint32_t e = lit[bb & ((1u << kLitBits) - 1u)];
        have_entry: // >= 56 bits in the buffer: enough for a length (<= 20 bits) and its offset (<= 28) without another refill
            const bool fast = (size_t)(out_end - out) >= kFastRoom;
            if (fast) { // up to three literal entries (six literals) per refill: 3 x 11 bits <= 56
                if (e & E_LIT) {
                    CONSUME(e & E_BITS);
                    out[0] = (uint8_t)(e >> 16); out[1] = (uint8_t)(e >> 24); // (the second byte is overwritten if there is none)
                    out += 1 + ((e >> 6) & 1u);
                    e = lit[bb & ((1u << kLitBits) - 1u)];
                    if (e & E_LIT) {
                        CONSUME(e & E_BITS);
                        out[0] = (uint8_t)(e >> 16); out[1] = (uint8_t)(e >> 24);
                        out += 1 + ((e >> 6) & 1u);
                        e = lit[bb & ((1u << kLitBits) - 1u)];
                        if (e & E_LIT) {
                            CONSUME(e & E_BITS);
                            out[0] = (uint8_t)(e >> 16); out[1] = (uint8_t)(e >> 24);
                            out += 1 + ((e >> 6) & 1u);
                            continue;
                        }
                    }
                    REFILL();
                }
            }
            if (__builtin_expect((e & (E_EXC | E_LIT)) != 0, 0)) { // (a length code of the root table, the common case, tests one flag pair)
            if (e & E_EXC) {
                if (e & E_SUB) {
                    CONSUME(e & E_BITS);
                    e = lit[(e >> 16) + ((uint32_t)bb & ((1u << ((e >> 8) & 15u)) - 1u))];
                }
                if (e & E_EXC) {
                    if (e & E_EOB) { CONSUME(e & E_BITS); break; }
                    return false; // invalid code (a subtable never points to a subtable)
                }
            }
            if (e & E_LIT) {
                const unsigned nl = 1 + ((e >> 6) & 1u);
                if ((size_t)(out_end - out) < nl) return false;
                CONSUME(e & E_BITS);
                out[0] = (uint8_t)(e >> 16);
                if (nl == 2) out[1] = (uint8_t)(e >> 24);
                out += nl;
                continue;
            }
            }
            uint64_t saved = bb;
            CONSUME(e & E_BITS);
            const unsigned length = (e >> 16) + (unsigned)((saved & (((uint64_t)1 << (e & E_BITS)) - 1u)) >> ((e >> 8) & 15u));
            uint32_t d = off[bb & ((1u << kOffBits) - 1u)];
            if (d & E_EXC) {
                if (!(d & E_SUB)) return false;
                CONSUME(d & E_BITS);
                d = off[(d >> 16) + ((uint32_t)bb & ((1u << ((d >> 8) & 15u)) - 1u))];
                if (d & E_EXC) return false;
            }
            saved = bb;
            CONSUME(d & E_BITS);
            const size_t dist = (d >> 16) + (size_t)((saved & (((uint64_t)1 << (d & E_BITS)) - 1u)) >> ((d >> 8) & 15u));
            if (dist > (size_t)(out - out0) || (!fast && length > (size_t)(out_end - out))) return false; // (fast: room for any match)
            const uint8_t* src = out - dist;
            if (fast) {
                REFILL(); // the next symbol's table entry is fetched while the bytes are copied
                e = lit[bb & ((1u << kLitBits) - 1u)];
                uint8_t* dst = out;
                out += length;
                if (dist >= 8) { // most matches are short: 16 bytes without a length-dependent branch, the rest in a loop
                    store64(dst, load64(src));
                    store64(dst + 8, load64(src + 8));
                    if (length > 16) {
                        dst += 16; src += 16;
                        do { store64(dst, load64(src)); dst += 8; src += 8; } while (dst < out);
                    }
                } else if (dist == 1) {
                    const uint64_t v = 0x0101010101010101ull * src[0];
                    do { store64(dst, v); dst += 8; } while (dst < out);
                } else {
                    do { *dst++ = *src++; } while (dst < out);
                }
                goto have_entry;
            } else {
                for (unsigned i = 0; i < length; ++i) out[i] = src[i];
                out += length;
            }
        }
        if (bfinal) break;
    }
#undef REFILL
#undef CONSUME
    if (out != out_end) return false;
    const size_t bytes = (size_t)(in - in0);
    if (bytes * 8 < bc) return false;
    return bytes * 8 - bc <= in_n * 8; // nothing beyond the stream was consumed
}
