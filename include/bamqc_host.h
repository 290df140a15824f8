/*
 * bamqc_host.h — host-side helpers that surround the aggregation path of bamqc.h:
 * the seeded synthetic-input generator, and (N2) the BGZF/BAM/FASTA readers and the
 * `bamqualcheck` driver that replace SeqAn's I/O (reference src/bamqualcheck.cpp:252-292,
 * 303-315; src/TripletCounting.hpp:71-104).  Plain C ABI, same conventions as bamqc.h.
 */
#ifndef BAMQC_HOST_H_
#define BAMQC_HOST_H_
#include "bamqc.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic inputs ------------------------------------- */
typedef struct bqc_synth_params {
    uint64_t seed;
    uint64_t first_read_index; /* global index of read 0 (lets ranks draw disjoint streams) */
    uint32_t n_reads;
    uint32_t read_len;         /* 150 for the PE configs, 10000 for the long-read config    */
    uint32_t n_refs;
    const uint32_t* ref_len;
    uint32_t n_lanes;
    int32_t isize;
    int32_t long_reads;        /* 1: 20-60 CIGAR ops, indel/soft-clip heavy                 */
} bqc_synth_params;

/* Deterministic reference contig as Dna5 codes (uniform ACGT, 0.1 % N runs). */
int bqc_synth_reference(uint64_t seed, int32_t rid, uint64_t len, uint8_t* out);
/* Coordinate-sorted reads drawn from `refs` (may be NULL: random bases).  The returned batch and
 * its arrays are owned by the library; release with bqc_synth_batch_free. */
int bqc_synth_batch(const bqc_synth_params* p, const uint8_t* const* refs, bqc_batch** out);
void bqc_synth_batch_free(bqc_batch* b);

#ifdef __cplusplus
}
#endif
#endif
