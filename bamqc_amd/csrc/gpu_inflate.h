// gpu_inflate.h — raw DEFLATE of BGZF blocks on the GPU (gpu_inflate.hip); used by the BGZF reader (host/bgzf.cpp).
#pragma once
#include <stddef.h>
#include <stdint.h>

struct GiBlock {      // one BGZF block's deflate stream
    uint64_t coff;    // offset of the deflate data in the compressed buffer
    uint64_t uoff;    // offset of its output in the uncompressed buffer
    uint32_t csize;   // bytes of deflate data
    uint32_t usize;   // ISIZE: the bytes it must inflate to
};

enum { GI_ERR_DATA = 1, GI_ERR_TRUNC = 2, GI_ERR_SIZE = 4, GI_ERR_CRC = 8 };

struct GpuInflater;
extern "C" {
GpuInflater* bqc_gpu_inflater_create(int device); // nullptr: no such device
void bqc_gpu_inflater_destroy(GpuInflater* g);
// 0: every block inflated to exactly its usize bytes; > 0: GI_ERR_* bits (corrupt data); < 0: the GPU could not be used
// device-resident operands, asynchronous on `stream` (a hipStream_t): inflates, and with d_crc (the blocks' expected CRC-32s) checks;
// *d_status collects GI_ERR_* bits
// (the blocks' outputs lie back to back in d_out, in the order of the table); d_tok / d_ntok: scratch of the two phases — 4 bytes x
// bqc_gpu_inflate_token_words(inflated bytes, blocks) and 4 bytes x blocks
size_t bqc_gpu_inflate_token_words(size_t inflated_bytes, size_t n_blocks);
int bqc_gpu_inflate_two_phase(); // BQC_GI_TWO_PHASE
void bqc_gpu_inflate_launch(const uint8_t* d_comp, const GiBlock* d_blocks, uint32_t n_blocks, uint64_t total_out, uint8_t* d_out, const uint32_t* d_crc, uint32_t* d_status,
                            uint32_t* d_tok, uint32_t* d_ntok, void* stream);
int bqc_gpu_inflate(GpuInflater* g, const uint8_t* comp, size_t comp_bytes, const GiBlock* blocks, size_t n_blocks, uint8_t* out, size_t out_bytes);
}
