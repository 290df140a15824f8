// device_types.h — structures shared by the host side of the library and the HIP kernels.
#pragma once
#include <stdint.h>
#include "state_layout.h"

// host annotations in the device flag column (see include/bamqc.h for 0x1000 / 0x8000)
#define BQC_FLAG_TRIPLET 0x2000u // read passed checkFlagsAndQuality (TripletCounting.hpp:136-168)
#define BQC_FLAG_COV     0x4000u // read enters OverallNumbers::coverage (bamqualcheck.cpp:430-433)

#define BQC_TILE_STRIDE 56       // k_long: positions owned per wave step (64 loaded: 8 look-ahead for 8-mers)
#define BQC_CT 304               // read_stats: read lengths / clip lengths up to here are counted in LDS, longer ones with global atomics
#define BQC_CHUNK_READS 128      // max reads per generic chunk
#define BQC_CHUNK_BASES 262144   // base budget per generic chunk (load balance only; a chunk always holds at least one read)
#define BQC_FAST_WAVES 16        // waves per workgroup of k_short = tiles per fast chunk (host chunk layout and kernel must agree)
#define BQC_FAST_NH 2            // 8-cycle halves a lane of k_short owns (2: 16 cycles, 4: 32 cycles); also the pad dwords of the nibble tables
#define BQC_T8_SPW 8             // rows (64 KiB images of the packed 8-mer counters) a workgroup of k_short can write per launch
#define BQC_FAST_MAXLEN 255      // reads up to this length take the short-read fast path (k_short); 255: per-read N / GC counts fit 8 bits
#define BQC_COV_TILE_WINDOWS 4   // coverage tile = 4 windows of 1000 positions
#define BQC_COV_TILE (BQC_COV_TILE_WINDOWS * 1000)

// device error word bits
#define BQC_DEVERR_QUAL  1u      // Phred byte > 222 (q+33 wraps in the reference's char arithmetic)
#define BQC_DEVERR_RANGE 2u      // mismatch / deletion / insertion count >= hist_cap
#define BQC_DEVERR_MATE  4u      // neither first nor last flag
#define BQC_DEVERR_INTERNAL 8u   // a kernel found its own layout assumptions violated (never expected)

struct Chunk {        // lane-uniform run of reads (indices into perm, or read ids when perm == nullptr)
    uint32_t first, count, lane;
    uint32_t aux;  // fast chunks: entries of the first-mate part (the chunk is [first-mate part | second-mate part]); else unused
    uint32_t pad0, pad1, pad2, pad3;
};

// Triplet segment of a short read whose CIGAR has several operations: read positions [ia, ib) are one match-like operation
// (other than the first, which the read's own record covers) aligned at chromPos = posv + i (TripletCounting.hpp:203-232).
// A mate part of a fast chunk lists them as entries 0x80000000 | index behind its reads: k_short evaluates triplets only for them.
struct TripSeg {
    uint32_t r;     // read
    int32_t posv;   // virtual alignment start: segment's reference start - its first read position
    uint32_t range; // ia | ib << 8
    uint32_t pad;
};
#define BQC_ENTRY_SEG 0x80000000u // perm entry: TripSeg index (0xFFFFFFFF = padding)

struct CovEntry {     // one covered interval: positions [win * 1000 + off, + len) in window coordinates, off + len <= 2000
    uint32_t win;     // batch-relative index of the read's first live window
    uint32_t off_len; // off | len << 16
};

struct CovTile {      // BQC_COV_TILE_WINDOWS consecutive coverage windows of one lane
    uint32_t lane;
    uint32_t win_lo;      // first window (batch-relative index)
    uint32_t list_begin;  // candidate intervals: cov_list[list_begin, list_end)
    uint32_t list_end;
    uint32_t win_final;   // windows < win_final are complete -> histogram; win_final, win_final+1 -> carry out
    uint32_t pad0, pad1, pad2;
};

struct DevBatch {
    uint32_t n_reads;
    // fixed columns, 48 B / read
    const uint16_t* flag;
    const uint8_t* mapq;
    const uint8_t* lane;
    const int32_t* rid;
    const int32_t* pos;
    const int32_t* tlen;
    const int32_t* nm;
    const int32_t* as_;
    const uint32_t* l_seq;
    const uint16_t* n_cigar;
    const uint32_t* seq_off;
    const uint32_t* qual_off;
    const uint32_t* cigar_off;
    // variable-length payload
    const uint8_t* seq;    // 4-bit packed
    const uint8_t* qual;   // raw Phred
    const uint32_t* cigar; // len<<4|op
    // work decomposition (host pre-pass)
    const uint32_t* perm;  // processing order: reads grouped by lane (and by mate inside fast chunks); entries with
                           // the top bit set are not reads (0xFFFFFFFF padding, else TripSeg index); nullptr = identity
    uint32_t n_perm;       // entries in perm (>= n_reads because of padding), n_reads when perm == nullptr
    const Chunk* chunks;
    uint32_t n_chunks;
    const uint32_t* nm_extra_read;
    const int32_t* nm_extra_val;
    uint32_t n_nm_extra;
    const CovEntry* cov_list; // covered intervals per read group, in stream order (host pre-pass)
    const CovTile* cov_tiles;
    uint32_t n_cov_tiles;
    // short-read fast path (k_short): lane-uniform chunks of reads with L <= 8 * BQC_FAST_NH * fast_w
    const Chunk* chunks_fast;
    uint32_t n_chunks_fast;
    uint32_t fast_w;           // lanes per read = ceil(max fast read length / (8 * BQC_FAST_NH))
    const TripSeg* segs;       // triplet segments of fast reads with several CIGAR operations
};

struct DevRefs {
    const uint8_t* const* ref; // [n_refs] Dna5 codes, nullptr if not loaded
    const uint64_t* len;
    const uint8_t* main_chrom; // [n_refs]
    uint32_t n_refs;
    // same contigs as nibbles r1 r0 ~r0 ~r1 (r = code & 3, N -> A as Dna5 -> Dna does), 8 bases per dword, first base in
    // the top nibble, BQC_FAST_NH zero dwords in front and at least as many behind (see k_ref_nibbles)
    const uint32_t* const* refn;
};
