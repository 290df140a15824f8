// prep.h — arguments of the device-side pre-pass (k_prep.hip), filled by the host pipeline (bqc_pipeline.cpp).
#pragma once
#include <stdint.h>
#include "device_types.h"

struct Stretch {          // the reads of one read group, contiguous in processing order: super-windows [sw_begin, sw_end)
    uint32_t lane, sw_begin, sw_end, pad;
};

struct PrepArgs {
    uint32_t n;               // reads
    uint32_t n_lanes, max_read_len;
    uint32_t no_fast;         // BQC_NO_FAST=1: every read takes the generic kernels
    uint32_t replay;          // the batch has been processed before (bqc_process on a resident batch): start from its saved FASTA cursor
    // raw columns (as uploaded)
    const uint16_t* flag_in;
    const uint8_t* mapq;
    const uint8_t* lane;
    const int32_t* rid;
    const int32_t* pos;
    const int32_t* as_;
    const uint32_t* l_seq;
    const uint16_t* n_cigar;
    const uint8_t* qual;
    const uint32_t* cigar;
    const CovEntry* cov_in;   // host anchor pass: {win, pos} or BQC_COV_NONE
    const uint32_t* order;    // processing order (reads grouped by read group) or nullptr
    const SuperWindow* sws;
    uint32_t n_sw;
    const Stretch* stretches;
    uint32_t n_stretch;
    const int32_t* fasta_index; // [n_refs] position of the contig in FASTA order (-1: absent) or nullptr = identity
    // outputs
    uint16_t* flag_out;
    uint32_t* seq_off;
    uint32_t* qual_off;
    uint32_t* cigar_off;
    CovEntry* cov_out;
    CovExtra* cov_extra;
    uint32_t cov_extra_cap;
    PendRun* pend;            // shard mode: covered runs of the reads marked BQC_COV_PENDING (retained until bqc_shard_resolve)
    PendExtra* pend_extra;
    uint32_t* pend_extra_n;
    uint32_t pend_extra_cap;
    uint16_t* cls;            // per read: class << 8 | triplet segments (class 0 / 1: fast read by mate slot, 2: generic path)
    TripSeg* segs;            // at cigar_off[r] + j
    uint32_t* perm;
    uint32_t perm_cap;
    Chunk* chunks_fast;
    uint32_t chunks_fast_cap;
    Chunk* chunks_slow;
    uint32_t chunks_slow_cap;
    BatchDesc* desc;
    ErrRec* err;
    int32_t* cursor;          // FASTA cursor of the stream (TripletCounting.hpp:254-259), device resident
    int32_t* cursor_save;     // its value before this batch
    // scratch
    unsigned long long* blk_sizes; // [blocks][3]
    uint32_t* blk_tgt;        // [blocks][2] max / min FASTA position + 1 of the block's eligible reads
    uint32_t* blk_maxfast;    // [blocks][2] longest fast read, longest generic read
    SwCounts* sw_counts;
    SwPlan* sw_plan;
};

extern "C" void bqc_launch_prep(const PrepArgs& a, const DevRefs& refs, hipStream_t s);
// shard mode: intervals of a batch's pending reads once the host knows their windows (cov_in = {win, pos} per pending read)
extern "C" void bqc_launch_pend_cov(uint32_t n, const CovEntry* cov_in, const PendRun* pend, const PendExtra* extra, const uint32_t* extra_n, uint32_t extra_cap,
                                    const uint8_t* lane, CovEntry* cov_out, CovExtra* cov_extra, BatchDesc* desc, hipStream_t s);
