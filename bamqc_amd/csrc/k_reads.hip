// k_reads.hip — per-read statistics: flag cascade + scalar counters (reference
// src/bamqualcheck.cpp:318-434) and the per-read histograms of QualityCheck
// (src/QualityCheck.hpp:168-271: read_length, map_Q, insert_size, mis_match, cigar_count).
//
// Thread per read over the generic chunks, i.e. the reads k_long handles (k_short runs the same read_stats code on
// its own reads).  Counters are privatised in LDS per workgroup for the read group ("lane") the workgroup is
// currently in and flushed with one global atomic per non-zero bin; bins beyond the LDS capacity go to global memory
// directly.  Reads the fixed columns + the CIGAR words of a read.
#include "kernels_common.h"
#include "read_stats.h"

__global__ __launch_bounds__(256) void k_reads(DevBatch b, StateLayout sl, uint64_t* __restrict__ state, DevRefs refs,
                                                  uint32_t* __restrict__ err, uint32_t fast_table)
{
    __shared__ uint32_t lds[RS_WORDS];
    for (uint32_t i = threadIdx.x; i < RS_WORDS; i += blockDim.x) lds[i] = 0;
    uint32_t blane = 0xFFFFFFFFu;
    block_sync();
    // the generic chunks; or (profiling only, BQC_SHORT_PARTS without 8) the fast chunks, whose per-read statistics k_short then leaves out
    const Chunk* chunks = fast_table ? b.chunks_fast : b.chunks;
    const uint32_t n_chunks = fast_table ? b.desc->n_chunks_fast : b.desc->n_chunks_slow;
    for (uint32_t ci = blockIdx.x; ci < n_chunks; ci += gridDim.x) { // lane-uniform chunks
        const Chunk ch = chunks[ci];
        if (ch.lane != blane) { // block-uniform
            if (blane != 0xFFFFFFFFu) rs_flush(lds, sl, state, blane);
            blane = ch.lane;
        }
        for (uint32_t t0 = 0; t0 < ch.count; t0 += blockDim.x) {
            const uint32_t t = t0 + threadIdx.x;
            bool live = t < ch.count;
            uint32_t r = live ? b.perm[ch.first + t] : 0;
            if (r & BQC_ENTRY_SEG) { live = false; r = 0; } // padding / triplet-segment entry of a fast chunk
            if (__ballot(live)) read_stats(b, sl, state, refs, err, lds, r, live, live);
        }
    }
    if (blane != 0xFFFFFFFFu) rs_flush(lds, sl, state, blane);
}

// further integer NM tags of a record: mis_match counts once per tag (QualityCheck.hpp:201-218)
__global__ void k_nm_extra(DevBatch b, StateLayout sl, uint64_t* __restrict__ state, DevRefs refs, uint32_t* __restrict__ err)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= b.n_nm_extra || b.desc->fatal) return;
    const uint32_t r = b.nm_extra_read[e];
    const uint32_t flag = b.flag[r];
    if (flag & 0x900) return;
    const bool first = flag & 0x40, last = !first && (flag & 0x80);
    if (!first && !last) return;
    const int32_t rid = b.rid[r];
    if (!(rid >= 0 && (uint32_t)rid < refs.n_refs && refs.main_chrom[rid]) || (flag & 0x4)) return;
    const uint32_t* cg = b.cigar + b.cigar_off[r];
    uint32_t del = 0, ins = 0;
    for (uint32_t k = 0; k < b.n_cigar[r]; ++k) {
        const uint32_t c = cg[k], op = c & 15u;
        if (op == 2u) del += c >> 4; else if (op == 1u) ins += c >> 4;
    }
    const uint32_t mm = (uint32_t)b.nm_extra_val[e] - del - ins;
    if (mm >= sl.hcap) { atomicOr(err, BQC_DEVERR_RANGE); return; }
    gadd(state + sl.mate_base(b.lane[r], first ? 0u : 1u) + sl.m_mismatch + mm, 1);
}

extern "C" void bqc_launch_reads_chunks(const DevBatch& b, const StateLayout& sl, uint64_t* state, const DevRefs& refs, uint32_t* err,
                                        uint32_t grid, uint32_t fast_table, hipStream_t s)
{
    if (grid == 0) return;
    hipLaunchKernelGGL(k_reads, dim3(grid), dim3(256), 0, s, b, sl, state, refs, err, fast_table);
}

extern "C" void bqc_launch_nm_extra(const DevBatch& b, const StateLayout& sl, uint64_t* state, const DevRefs& refs, uint32_t* err, hipStream_t s)
{
    if (b.n_nm_extra)
        hipLaunchKernelGGL(k_nm_extra, dim3((b.n_nm_extra + 255) / 256), dim3(256), 0, s, b, sl, state, refs, err);
}
