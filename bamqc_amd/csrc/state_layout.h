// state_layout.h — flat uint64 state vector shared by host and device code.
//
// Every word is a pure sum over reads (or a difference array whose prefix sum is
// taken at finalisation), so per-GPU vectors add: one RCCL all-reduce/reduce of
// `words` uint64 is the whole multi-GPU merge.  The vector mirrors
// `struct Counts` (reference src/bamqualcheck.cpp:14-38) lane by lane.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BQC_HD __host__ __device__
#else
#define BQC_HD
#endif

struct StateLayout {
    // capacities
    uint32_t n_lanes, lcap, hcap, icap; // icap = isize + 1
    // per-lane offsets (words, relative to the lane base)
    uint32_t o_scalars;   // 13 scalars (OverallNumbers.hpp:12-24)
    uint32_t o_covstart;  // 1: number of shards in which this lane saw a coverage read
    uint32_t o_poscov;    // 101 (OverallNumbers.hpp:54)
    uint32_t o_eightmer;  // 65536
    uint32_t o_triplet;   // 64*4*4
    uint32_t o_mate[2];   // QualityCheck r1 / r2
    // per-mate offsets (relative to o_mate[m])
    uint32_t m_dnacount;  // 5 * lcap   [code][cycle]
    uint32_t m_qualcount; // lcap
    uint32_t m_sc5hist;   // lcap + 1   histogram of min(leading clip, L); sc5[j] = sum_{n>j}
    uint32_t m_sc3diff;   // lcap + 1   +1 at L-n, -1 at L; sc3 = prefix sum
    uint32_t m_readnr;    // 1
    uint32_t m_ncount;    // lcap + 1
    uint32_t m_gccount;   // lcap + 1
    uint32_t m_avgqual;   // 256   bin round(mean)
    uint32_t m_avgceil;   // 256   presence of ceil(mean)  (array length tracker)
    uint32_t m_mapq;      // 256
    uint32_t m_readlen;   // lcap + 1
    uint32_t m_mismatch;  // hcap
    uint32_t m_delhist;   // hcap
    uint32_t m_inshist;   // hcap
    uint32_t m_insert;    // icap
    uint32_t mate_words;
    uint32_t lane_words;
    uint64_t words;       // n_lanes * lane_words

    BQC_HD uint64_t lane_base(uint32_t lane) const { return (uint64_t)lane * lane_words; }
    BQC_HD uint64_t mate_base(uint32_t lane, uint32_t m) const { return lane_base(lane) + o_mate[m]; }
};

static inline uint32_t bqc_align8(uint32_t x) { return (x + 7u) & ~7u; }

static inline StateLayout make_state_layout(uint32_t n_lanes, uint32_t lcap, uint32_t hcap, uint32_t icap)
{
    StateLayout s;
    s.n_lanes = n_lanes; s.lcap = lcap; s.hcap = hcap; s.icap = icap;
    uint32_t o = 0;
    s.o_scalars = o; o += 13;
    s.o_covstart = o; o += 3;
    s.o_poscov = o; o += bqc_align8(101);
    s.o_eightmer = o; o += 65536;
    s.o_triplet = o; o += 1024;
    uint32_t m = 0;
    s.m_dnacount = m; m += bqc_align8(5 * lcap);
    s.m_qualcount = m; m += bqc_align8(lcap);
    s.m_sc5hist = m; m += bqc_align8(lcap + 1);
    s.m_sc3diff = m; m += bqc_align8(lcap + 1);
    s.m_readnr = m; m += 8;
    s.m_ncount = m; m += bqc_align8(lcap + 1);
    s.m_gccount = m; m += bqc_align8(lcap + 1);
    s.m_avgqual = m; m += 256;
    s.m_avgceil = m; m += 256;
    s.m_mapq = m; m += 256;
    s.m_readlen = m; m += bqc_align8(lcap + 1);
    s.m_mismatch = m; m += bqc_align8(hcap);
    s.m_delhist = m; m += bqc_align8(hcap);
    s.m_inshist = m; m += bqc_align8(hcap);
    s.m_insert = m; m += bqc_align8(icap);
    s.mate_words = m;
    s.o_mate[0] = o; o += m;
    s.o_mate[1] = o; o += m;
    s.lane_words = o;
    s.words = (uint64_t)n_lanes * o;
    return s;
}
