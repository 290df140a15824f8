// inflate_fast.h — raw DEFLATE (RFC 1951) decoder for whole in-memory blocks of known output size.
//
// BGZF blocks are at most 64 KiB, complete in memory and carry their uncompressed size, so the decoder needs none of
// zlib's streaming state: a 64-bit bit buffer refilled with unaligned 8-byte loads, one table lookup per symbol
// (11-bit litlen root, 8-bit offset root, second-level tables for longer codes), word-wise match copies.  Inflating is the
// largest consumer of host CPU time of a bamqualcheck run (N2: the records must be inflated before they can be decoded).
//
// Contract of Inflater::run(in, in_n, out, out_n):
//   * `in` holds one complete raw deflate stream; the 8 bytes after in + in_n must be READABLE (their value is ignored):
//     in a BGZF block they are the CRC32 / ISIZE trailer.
//   * exactly out_n bytes are produced; nothing outside [out, out + out_n) is written.
//   * returns false for anything else: corrupt or truncated data, a stream that produces fewer or more bytes.
// The decoder is the reader's only inflate; BgzfReader checks every block's CRC32 afterwards.
#pragma once
#include <cstddef>
#include <cstdint>

class Inflater {
public:
    bool run(const uint8_t* in, size_t in_n, uint8_t* out, size_t out_n);

    static constexpr int kLitBits = 11, kOffBits = 8, kPreBits = 7;
    static constexpr int kLitEnough = 2400, kOffEnough = 448; // root + worst-case second-level tables (2342 / 402), rounded up

private:
    uint32_t lit_[kLitEnough];
    uint32_t off_[kOffEnough];
    uint32_t pre_[1 << kPreBits];
    uint8_t lens_[288 + 32 + 160]; // litlen + offset code lengths (+ slack for an overlong repeat, rejected afterwards)
};
