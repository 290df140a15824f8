// anchor.h — the order-dependent part of OverallNumbers::coverage (OverallNumbers.hpp:84-110) ON THE CARD (k_anchor.hip), for
// batches whose fixed columns already live in device memory (the reader on the card, gpu_bam.hip): the columns then never come back
// to the host, which only sees a summary of a few hundred bytes and the list of reads at which the window index changes.
#pragma once
#include <stdint.h>

#include "device_types.h"

// the window state machine of one read group between two reads (host mirror: LaneCov, bqc_ctx.h)
struct AnchorState {
    uint32_t first;   // no read has entered coverage() yet
    int32_t id;       // chromosome of the live windows
    int32_t shift;    // position of the first live window's first base
    uint32_t pad;
    uint64_t win;     // absolute index (flush order) of the first live window
    // A shard that starts inside the stream (bqc_options.shard_tail): its reads are SET ASIDE (BQC_COV_PENDING) up to the first one
    // that resets the windows whatever their state — another chromosome, or more than 2000 positions from the read before
    // (bqc_pipeline.cpp: host_pass) —, from which on the shard runs as a stream of its own.
    uint32_t pending;  // still setting aside
    uint32_t has_prev; // a read has been seen (prev_rid / prev_bp are the last one's)
    int32_t prev_rid;
    uint32_t prev_bp;
};

// what the host needs from a batch besides the anchors themselves
struct AnchorSummary {
    uint32_t n_cand;        // reads that enter coverage()
    uint32_t n_breaks;      // candidates that are not < 1000 positions behind the candidate before them on the same chromosome
    uint32_t n_bound;       // (unused since round 4: the boundary list is the dense array AnchorArgs::first_of)
    uint32_t flags;         // AN_FLAG_*
    uint32_t n_slow;        // reads of the generic path (longer than BQC_FAST_MAXLEN, or every read with no_fast)
    uint32_t max_len_slow;
    uint32_t n_noqual;      // primary first / last records without qualities (check_read_len's message, QualityCheck.hpp:70-79)
    uint32_t last_rel;      // window (relative to before.win) of the last candidate
    int32_t rid_min, rid_max; // range of the reference ids in [0, n_refs) the batch holds (rid_min > rid_max: none)
    uint32_t n_pending;     // candidates set aside (the first n_pending of the batch's candidates: a shard_tail context)
    uint32_t first_certain; // candidate index of the first read that resets whatever the state (0xFFFFFFFF: none; only looked for while setting aside)
    unsigned long long seq_bytes, qual_bytes, cigar_words; // payload sizes: sums of ceil(l_seq / 2), l_seq, n_cigar
    AnchorState before, after;
};
#define AN_FLAG_TOO_MANY_BREAKS 1u // not anchored: the state is untouched, the caller takes the host's recurrence for this batch
#define AN_FLAG_BOUND_OVERFLOW  2u // never expected (the list is sized for every candidate)

#define AN_NO_READ 0xFFFFFFFFu // first_of[rel]: no candidate's window changes TO rel (a reset skips one)

// a run = a break and the candidates behind it up to the next break: inside it every gap is in [0, 1000) on one chromosome, so a
// read's state follows in closed form from the state the run was entered with
struct AnchorRun {
    uint32_t b_e;     // beginPos of the run's first read
    uint32_t s_e;     // shift after that read
    uint32_t rel_e;   // window (relative to before.win) after that read
    uint32_t stuck;   // that read sits at offset 2000 exactly (neither slide nor reset): reads at the same position stay there, the first
                      // one further right resets
    uint32_t b_star;  // stuck: beginPos of that first read further right (if the run has one)
};

// breaks a batch may hold for the card's chain (ONE thread walks them, ~0.2 us each: 3 ms at this limit, what the host's pass over a
// million reads takes); a batch with more — sparse data: every other read is a break — is left to the host
#define AN_MAX_BREAKS 16384u

struct AnchorPart { // what a workgroup of k_an_count (1024 reads) knows; summed by k_an_scan
    uint32_t n_slow, max_slow, n_noqual, first_certain;
    int32_t rid_min, rid_max;
    unsigned long long s1, s2, s3;
};

struct AnchorArgs {
    uint32_t n, n_refs, n_lanes, no_fast;
    const uint16_t* flag; const uint8_t* lane; const int32_t* rid; const int32_t* pos; const uint32_t* l_seq; const uint16_t* n_cigar;
    const uint8_t* main_chrom;
    CovEntry* cov_out;          // [n]
    AnchorState* state;         // the read group's state: read by the chain, replaced when the batch is anchored
    AnchorSummary* sum;
    // first_of[rel] = the first read (index in the batch) whose window is `rel`, written by the candidate whose window differs from its
    // predecessor's (windows only grow along the candidates, by one per slide and two per reset: rel <= 2 n_cand + 2); preset to AN_NO_READ.
    // (Round 4, first version: a list appended to with an atomic counter — ~5 000 returning atomics on ONE word per million reads,
    // 60 of k_an_apply's 83 us.)
    uint32_t* first_of; uint32_t first_cap;
    // scratch
    uint32_t* cpos; int32_t* crid; uint32_t* cidx; uint32_t* crun; // [n] candidates in stream order
    uint32_t* bj;               // [AN_MAX_BREAKS] candidate index of every break
    AnchorRun* runs;            // [AN_MAX_BREAKS]
    uint32_t* blk_a; uint32_t* blk_b; // [n / 1024 + 2] block counts / offsets of the two compactions
    AnchorPart* parts;          // [n / 1024 + 2]
};

extern "C" void bqc_launch_anchor(const AnchorArgs& a, hipStream_t s);
