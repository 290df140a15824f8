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
#define BQC_T8_USED 4            // words per slot of the rows' directory: rows written, the read group of rows 0-3 (a byte each), of rows 4-7, unused
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

// Coverage entry of read i (cov[i]).  The host's anchor pass (OverallNumbers.hpp:84-110) writes {win, pos} for every read that
// enters coverage(): win = batch-relative index of the read's first live window, pos = beginPos - shift in [0, 2000]
// (BQC_COV_NONE in win for every other read); k_prep_reads walks the CIGAR and rewrites it as the covered interval
// [win * 1000 + off, + len), off + len <= 2000, len = 0: nothing.
struct CovEntry {
    uint32_t win;
    uint32_t off_len; // host: pos; device: off | len << 16
};
#define BQC_COV_NONE 0xFFFFFFFFu
// Shard of a stream that does not start at the stream's first record (multi-GPU): the window of a read group's first reads is
// unknown until the predecessor shard's final state arrives.  The host marks such reads BQC_COV_PENDING with their index in
// the batch's pending log (off_len); k_prep_reads stores the covered run(s) relative to beginPos there instead of an interval.
#define BQC_COV_PENDING 0xFFFFFFFEu
struct PendRun { uint32_t c0, len; };                 // first covered run: starts c0 positions behind beginPos (len = 0: none)
struct PendExtra { uint32_t idx, c0, len, pad; };     // further runs of pending read idx
struct CovExtra {     // second, third ... covered interval of a read whose clips sit between match operations (rare)
    uint32_t win, off_len, lane, pad;
};

struct CovTile {      // BQC_COV_TILE_WINDOWS consecutive coverage windows of one lane
    uint32_t lane;
    uint32_t win_lo;      // first window (batch-relative index)
    uint32_t list_begin;  // candidate reads: cov[list_begin, list_end) (stream order; entries of other lanes are skipped when `mixed`)
    uint32_t list_end;
    uint32_t win_final;   // windows < win_final are complete -> histogram; win_final, win_final+1 -> carry out
    uint32_t mixed;       // the batch holds several read groups: check lane[i] == lane
    uint32_t pad1, pad2;
};

// A run of reads of one read group in processing order (`order`, or the stream itself): the unit the chunk builder works on
// is a super-window = up to BQC_SW_READS consecutive positions of one stretch.
#define BQC_SW_READS 4096
struct SuperWindow {
    uint32_t lane;
    uint32_t begin, count; // positions [begin, begin + count) of the processing order
    uint32_t stretch;      // index of the lane stretch (chunks never span two stretches)
};
struct SwCounts {          // k_build_count -> k_build_plan
    uint32_t n0, n1;       // fast reads by mate slot (first mate / everything else)
    uint32_t n_slow;       // reads for the generic kernels
    uint32_t n_seg;        // triplet segments of the fast reads
};
struct SwPlan {            // k_build_plan -> k_build_scatter
    uint32_t group_base;   // perm index of the super-window's first read group
    uint32_t n_groups;
    uint32_t seg_base;     // perm index of its first segment entry
    uint32_t slow_base;    // perm index of its first generic-path read
};

// What only the device knows about a batch (written by k_build_plan, read by every later kernel of the batch).
struct BatchDesc {
    uint32_t n_chunks_fast;  // read chunks and segment chunks of k_short
    uint32_t n_chunks_slow;  // chunks of k_reads / k_long
    uint32_t fast_w;         // lanes per read of k_short = ceil(longest fast read / (8 * BQC_FAST_NH))
    uint32_t n_perm;
    uint32_t long_max_len;   // longest read of the generic path
    uint32_t n_cov_extra;
    uint32_t fatal;          // the pre-pass found a read that ends the reference's run: the hot kernels leave the batch alone
    uint32_t pad1;
};

// Error record of a batch: the first failing read in stream order (key = read << 3 | order of the check within a read,
// 1 length, 2 lane, 3 offsets, 4 AS tag, 5 FASTA, 6 mate flag) and the unordered BQC_DEVERR_* flags of the hot kernels.
struct ErrRec {
    unsigned long long first_key; // ~0ull: none
    uint32_t flags;
    uint32_t aux0, aux1;          // values for the message (read length / lane / reference id)
    uint32_t pad;
};
#define BQC_ERRKEY_NONE (~0ull)

struct DevBatch {
    uint32_t n_reads;
    // fixed columns, 48 B / read
    const uint16_t* flag;      // BAM flag | host annotations (0x1000, 0x8000) | device annotations (BQC_FLAG_TRIPLET / _COV, NO_QUAL): k_prep_reads
    const uint8_t* mapq;
    const uint8_t* lane;
    const int32_t* rid;
    const int32_t* pos;
    const int32_t* tlen;
    const int32_t* nm;
    const int32_t* as_;
    const uint32_t* l_seq;
    const uint16_t* n_cigar;
    const uint32_t* seq_off;   // payload offsets: prefix sums over the reads (k_prep_sizes / _scan / _reads)
    const uint32_t* qual_off;
    const uint32_t* cigar_off;
    // variable-length payload
    const uint8_t* seq;    // 4-bit packed
    const uint8_t* qual;   // raw Phred
    const uint32_t* cigar; // len<<4|op
    // work decomposition (device: k_build_*)
    const uint32_t* order; // processing order of the reads: grouped by read group (host, only when the batch holds several); nullptr = stream order
    const uint32_t* perm;  // entries of the chunks: read ids grouped by mate slot inside fast chunks; entries with
                           // the top bit set are not reads (0xFFFFFFFF padding, else TripSeg index)
    const Chunk* chunks;       // generic chunks (k_reads / k_long)
    const Chunk* chunks_fast;  // k_short: lane-uniform chunks of reads with L <= 8 * BQC_FAST_NH * fast_w, and of their triplet segments
    const BatchDesc* desc;     // counts of the tables above (device memory)
    const uint32_t* nm_extra_read;
    const int32_t* nm_extra_val;
    uint32_t n_nm_extra;
    const CovEntry* cov;       // [n_reads] covered interval per read (k_prep_reads)
    const CovExtra* cov_extra; // [desc->n_cov_extra]
    const CovTile* cov_tiles;
    uint32_t n_cov_tiles;
    const TripSeg* segs;       // triplet segments of fast reads with several CIGAR operations, at cigar_off[r] + j
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
