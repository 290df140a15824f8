// crc32_fast.h — CRC-32 (IEEE 802.3, the gzip / BGZF checksum) of a buffer, by carry-less multiplication where the CPU
// has PCLMULQDQ (folding four 128-bit lanes, then Barrett reduction), else zlib's table code.  After the DEFLATE decoder
// (inflate_fast.h) the checksum was a third of the per-block host time.
#pragma once
#include <cstddef>
#include <cstdint>

uint32_t bqc_crc32_fast(const uint8_t* p, size_t n);
