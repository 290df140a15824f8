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
/* Reads [lo, lo + count) of the plan bqc_synth_batch would generate for p->n_reads reads (the same reads at the same
 * indices): lets a generator stream a plan that does not fit in memory.  Release with bqc_synth_batch_free. */
int bqc_synth_slice(const bqc_synth_params* p, uint64_t lo, uint32_t count, const uint8_t* const* refs, bqc_batch** out);

/* Writes a coordinate-sorted synthetic BAM (+ FASTA when fasta_path != NULL) for the parameters:
 * header with @SQ per contig and one @RG (ID:L<n>, SM:SYN) per lane; tags RG:Z, NM:i, AS:i. */
int bqc_synth_write(const bqc_synth_params* p, const char* const* ref_names, const char* bam_path, const char* fasta_path,
                    uint32_t batch_reads);

/* The same BAM, generated and written slice by slice (slice_reads reads in memory at a time; records serialised and BGZF
 * blocks compressed on all host threads; level 0 = stored blocks): for plans of hundreds of millions of reads.  bam_path
 * may be a FIFO that a reader drains meanwhile.  The FASTA is written first when fasta_path != NULL. */
int bqc_synth_stream(const bqc_synth_params* p, const char* const* ref_names, const char* bam_path, const char* fasta_path,
                     uint32_t slice_reads, int level);

/* Any batch as a BAM file with the header bqc_synth_write uses (@SQ per contig, @RG ID:L<n> SM:SYN per lane; read names
 * r<first_read_index + i>; tags RG:Z, NM:i, AS:i). */
int bqc_bam_write(const char* path, const bqc_batch* b, uint32_t n_refs, const char* const* ref_names, const uint32_t* ref_lens,
                  uint32_t n_lanes, uint64_t first_read_index, int level);

/* ---- BAM / FASTA input (replaces SeqAn BamStream / SequenceStream) --------- */
typedef struct bqc_bam bqc_bam;
int bqc_bam_open(const char* path, bqc_bam** out);  /* on failure *out still holds the message */
/* One shard of the record stream (multi-GPU): the records that START in the BGZF blocks between the first block boundary at
 * or behind the compressed offset begin_hint and the first one at or behind end_hint (0 / UINT64_MAX: the file's ends).  A
 * shard in the middle of a file locates its first record by a plausibility test; shards verify each other afterwards:
 * bqc_bam_range_over of a shard must equal bqc_bam_range_first of its successor (else: process the file unsharded). */
int bqc_bam_open_range(const char* path, uint64_t begin_hint, uint64_t end_hint, bqc_bam** out);
/* The same reader with the file inflated and its records decoded on GPU `device` (the batches of bqc_bam_next are fetched back for
 * the caller).  A batch that holds a record the card does not decode (a read group missing from the header, a second NM tag, no
 * RG tag, ...) is decoded by the host reader's rules from the bytes on the card (bqc_bam_batches_handed_over counts them);
 * bqc_bam_next returns -1000 when the FILE needs the host reader (a record walk that cannot be verified, a corrupt record, a
 * file that ends inside a record): open it with bqc_bam_open then. */
int bqc_bam_open_gpu(const char* path, int device, bqc_bam** out);
uint64_t bqc_bam_batches_handed_over(const bqc_bam* b);
/* A shard (as bqc_bam_open_range) read, inflated and decoded on GPU `device`: only the shard's bytes of the file are touched. */
int bqc_bam_open_gpu_range(const char* path, int device, uint64_t begin_hint, uint64_t end_hint, bqc_bam** out);
uint64_t bqc_bam_range_begin_block(const bqc_bam* b);
uint64_t bqc_bam_range_end_block(const bqc_bam* b);   /* UINT64_MAX: end of the file */
uint64_t bqc_bam_range_first(const bqc_bam* b);       /* valid after the first bqc_bam_next */
uint64_t bqc_bam_range_over(const bqc_bam* b);        /* valid after the last bqc_bam_next  */
uint64_t bqc_file_size(const char* path);
/* device >= 0: BAM readers inflate their BGZF blocks on that GPU from their next run of blocks on (the CRC-32 of every block is
 * still checked on the host; a card that cannot be used hands the work back to the CPU decoder); -1: on the CPU (the default). */
void bqc_gpu_inflate_device(int device);
uint64_t bqc_gpu_inflated_blocks(void);               /* BGZF blocks inflated on a GPU so far in this process */
void bqc_bam_close(bqc_bam* b);
const char* bqc_bam_error(const bqc_bam* b);
uint32_t bqc_bam_n_refs(const bqc_bam* b);
const char* bqc_bam_ref_name(const bqc_bam* b, uint32_t i);
uint32_t bqc_bam_ref_len(const bqc_bam* b, uint32_t i);
const char* bqc_bam_sample_id(const bqc_bam* b);      /* SM of the last @RG (bamqualcheck.cpp:58-61)        */
uint32_t bqc_bam_lane_count(const bqc_bam* b);        /* number of @RG IDs in the header                    */
uint32_t bqc_bam_n_lane_names(bqc_bam* b);            /* lane names in output (lexicographic) order,        */
const char* bqc_bam_lane_name(const bqc_bam* b, uint32_t i); /* including IDs first seen on a read (:86)  */
uint32_t bqc_bam_lane_index(const bqc_bam* b, uint32_t i);
int bqc_bam_set_main_chrom(bqc_bam* b, const uint8_t* main_chrom); /* [n_refs], for BQC_FLAG_MATE_MAIN  */
/* Keep only records whose refID is selected (refID -1 follows keep_unplaced): lets each rank of a
 * multi-GPU run take whole chromosomes of a coordinate-sorted BAM. */
int bqc_bam_set_rid_filter(bqc_bam* b, const uint8_t* keep, int keep_unplaced);
/* Next batch of decoded records: 1 = batch returned (owned by the reader until the next call),
 * 0 = end of file, < 0 = -(BQC_ERR_*) with the message in bqc_bam_error. */
int bqc_bam_next(bqc_bam* b, uint32_t max_reads, uint64_t max_bases, const bqc_batch** out);

/* Whole FASTA as Dna5 codes; ids cut at the first space/tab (TripletCounting.hpp:99-102). */
int bqc_fasta_load(const char* path, uint32_t* n_records, char*** names, uint8_t*** codes, uint64_t** lens);
void bqc_fasta_free(uint32_t n_records, char** names, uint8_t** codes, uint64_t* lens);

/* The BGZF reader's DEFLATE decoder on one raw deflate stream (host/inflate_fast.h; exposed for its tests): 1 when the
 * stream is valid, ends within in_n bytes and yields exactly out_n bytes, else 0.  The 8 bytes after in + in_n must be
 * readable (in a BGZF block: the CRC32 / ISIZE trailer). */
int bqc_inflate_raw(const uint8_t* in, uint64_t in_n, uint8_t* out, uint64_t out_n);
/* The reader's CRC-32 (gzip polynomial; host/crc32_fast.h), as zlib's crc32(0, p, n). */
uint32_t bqc_crc32(const uint8_t* p, uint64_t n);

/* Profiling aid: stream `bytes` of device memory `repeat` times with 4-byte-per-lane loads (kernel k_calib_read4),
 * used to calibrate the rocprofv3 FETCH_SIZE counter on a known byte count. */
int bqc_calib_read4(uint64_t bytes, int repeat);

/* ---- the program: drop-in for the reference's main() (bamqualcheck.cpp:239-457) ---- */
int bqc_main(int argc, const char** argv);

/* The program as one of shard_count processes (one per GPU) that split the BAM file's byte stream between them: this process
 * takes the records that start in its part of the compressed file (bqc_bam_open_range with hints i / n of the file size), its
 * context is a shard_tail context unless it is the first.  After its record loop it calls `hook` once — also when it failed,
 * so that all processes can agree — and the hook does what needs the other processes (launcher's business: torch.distributed
 * over RCCL in bamqc_amd/distributed.py): verify the split (range_over of a shard == range_first of its successor), hand the
 * coverage state down the chain (bqc_shard_export / bqc_shard_resolve), sum the state vectors onto the first process
 * (bqc_state_export / _import) and merge the lane names.  The hook returns BQC_SHARD_WRITE (this process finalises and writes
 * the output, with the lane names it put into `out`), BQC_SHARD_DONE (nothing more to do here), BQC_SHARD_FALLBACK (the split
 * could not be verified: this process runs the whole file again, unsharded) or BQC_SHARD_FAIL (exit status 1). */
typedef struct bqc_shard_info {
    bqc_ctx* ctx;                     /* NULL when the program failed before it was created                  */
    int32_t status;                   /* exit status so far: 0, or 1 after an error (message already printed) */
    uint64_t begin_block, end_block;  /* bqc_bam_range_*                                                      */
    uint64_t first, over;
    const char* sample_id;
    uint32_t n_lane_names;            /* lane names known to this process, incl. ids first seen on a read     */
    const char* const* lane_names;
    const uint32_t* lane_index;
} bqc_shard_info;
typedef struct bqc_shard_result {     /* filled by the hook for BQC_SHARD_WRITE: the merged lane names, output order */
    uint32_t n_lane_names;
    const char* const* lane_names;
    const uint32_t* lane_index;
} bqc_shard_result;
enum { BQC_SHARD_FAIL = 0, BQC_SHARD_DONE = 1, BQC_SHARD_FALLBACK = 2, BQC_SHARD_WRITE = 3 }; /* (0 — what a hook that died returns — never means "write") */
typedef int (*bqc_shard_hook)(void* user, const bqc_shard_info* info, bqc_shard_result* out);
int bqc_main_shard(int argc, const char** argv, uint32_t shard_index, uint32_t shard_count, bqc_shard_hook hook, void* user);

/* The program over n_gpus GPUs of one node as one binary (`bamqualcheck --gpus N`; reference: the single main() of
 * bamqualcheck.cpp:239-457): forks one worker per GPU — call it before anything in the process has touched a GPU — each of
 * which runs bqc_main_shard over its byte range of the BAM file; the shard hook is implemented in C++ over socket pairs for the
 * few words of agreement and the 8 KB coverage hand-over, and over librccl (loaded at run time) for the ONE reduce of the state
 * vectors: ncclReduce(uint64, sum) on device pointers.  BQC_REDUCE=pipe sums on the host instead (and so do workers that share
 * one card, BQC_GPUS_SHARE_DEVICE=1).  argv: the program's arguments (without --gpus).  Returns the exit status. */
int bqc_main_multi(int argc, const char** argv, int n_gpus);

/* The program's command line, parsed as bqc_main parses it (CommandLineParser.hpp:43-149) without running anything: 0 = valid
 * (the input path — a file name or "-" — is copied to input_path), 1 = usage error (message printed on stderr), 2 = -h /
 * --version answered on stdout.  bqc_main_multi calls it once before it forks its workers. */
int bqc_program_args(int argc, const char** argv, char* input_path, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
