/*
 * bamqc.h — C ABI of the MI355X-native per-read aggregation path of BamQC.
 *
 * The reference (DecodeGenetics/BamQC) has no FFI: its hot path is the body of
 * the `while (!atEnd(inStream))` loop in src/bamqualcheck.cpp:303-444, which
 * calls member functions on `struct Counts` (src/bamqualcheck.cpp:14-38).
 * This header is the seam "loop body <-> Counts": a host that has decoded BAM
 * records hands them over as structure-of-arrays batches, the library
 * aggregates them on the GPU, and `bqc_finalize` returns a host mirror of
 * `Counts` that a writer turns into the `.bamqc` text
 * (src/bamqualcheck.cpp:156-233).
 *
 * Plain C, no exceptions cross the boundary, every entry point returns
 * 0 = ok / non-zero = error with a message retrievable via bqc_last_error()
 * (mirrors the reference's "message on stderr + return 1",
 * src/bamqualcheck.cpp:265,281,308,315,341,388).
 *
 * Threading: one submitting thread per context (the reference is
 * single-threaded).
 */
#ifndef BAMQC_H_
#define BAMQC_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BQC_ABI_VERSION 2

/* ---- error codes ---------------------------------------------------------- */
enum {
    BQC_OK = 0,
    BQC_ERR_ARG = 1,          /* bad argument / inconsistent batch                              */
    BQC_ERR_DEVICE = 2,       /* HIP runtime error (no GPU, OOM, launch failure)                */
    BQC_ERR_NO_MATE_FLAG = 3, /* neither 0x40 nor 0x80 set   (bamqualcheck.cpp:385-389)         */
    BQC_ERR_AS_TAG = 4,       /* AS tag missing/negative on a read that passed flags+mapQ
                                 (TripletCounting.hpp:116-127,154-155 -> bamqualcheck.cpp:340)  */
    BQC_ERR_FASTA = 5,        /* contig needed for triplets is missing from / behind the FASTA
                                 cursor (TripletCounting.hpp:85-104,254-259)                    */
    BQC_ERR_RANGE = 6,        /* value exceeds a configured capacity (read length, histogram)   */
    BQC_ERR_IO = 7,           /* file could not be opened / parsed                              */
    BQC_ERR_STATE = 8         /* call sequence error                                            */
};

/* ---- scalar counters of OverallNumbers (OverallNumbers.hpp:12-24) ---------- */
enum {
    BQC_S_SUPPLEMENTARY = 0,
    BQC_S_DUPLICATES,
    BQC_S_QCFAILED,
    BQC_S_NOT_PRIMARY,
    BQC_S_READCOUNT,
    BQC_S_TOTALBPS,
    BQC_S_BOTHUNMAPPED,
    BQC_S_FIRSTUNMAPPED,
    BQC_S_SECONDUNMAPPED,
    BQC_S_FIRST_AND_OR_SECOND_MAPPED,
    BQC_S_FF_RR,
    BQC_S_PROPERPAIR,
    BQC_S_AUTO_PROPERPAIR,
    BQC_N_SCALARS
};

/* Host annotations carried in the spare high bits of the BAM flag column.     */
#define BQC_FLAG_MATE_MAIN 0x1000u /* rNextId is a main chromosome (bamqualcheck.cpp:404) */
#define BQC_FLAG_NO_QUAL   0x8000u /* the record has no qualities (BAM: quality block starts with 0xFF; SAM: QUAL is "*")
                                      => record.qual empty.  Set by whoever decodes the record: the library does not look
                                      at the quality bytes to find out (it would cost a cache line per read).             */

#define BQC_NM_ABSENT (-1)
#define BQC_AS_ABSENT INT32_MIN

#define BQC_COVSIZE 100   /* OverallNumbers.hpp:51 covsize */
#define BQC_VSIZE   1000  /* OverallNumbers.hpp:51 vsize   */
#define BQC_N_8MER  65536 /* OverallNumbers.hpp:53         */
#define BQC_N_TRIPLET (64 * 4 * 4)

/* ---- options -------------------------------------------------------------- */
typedef struct bqc_sketch_options { /* N1: ReadQualityHasher/StreamCounter (CommandLineParser.hpp:73-81) */
    uint32_t n_k;          /* number of k values (0 = sketch disabled)          */
    const int32_t* klist;  /* -k, default {32}; 1 <= k <= 63                     */
    uint32_t n_q;
    const uint32_t* qlist; /* -q, default {17}                                   */
    double e;              /* -e, default 0.01                                   */
    int32_t seed;          /* -s, default 1; 0 = table seeded from time(NULL), as RepHash.cpp:5-7 (not reproducible;
                              shards of one run must then be given the same non-zero seed by their driver)   */
} bqc_sketch_options;

typedef struct bqc_options {
    uint32_t struct_size;      /* sizeof(bqc_options), for ABI evolution                 */
    uint32_t n_lanes;          /* number of @RG IDs (bamqualcheck.cpp:296-298), >= 1     */
    uint32_t n_refs;           /* number of BAM reference sequences                      */
    int32_t isize;             /* -i (CommandLineParser.hpp:66-67), default 1000         */
    uint32_t max_read_len;     /* capacity of the per-cycle arrays (reference grows on
                                  demand, QualityCheck.hpp:85-105); longer read => BQC_ERR_RANGE */
    uint32_t hist_cap;         /* capacity of mismatch/deletion/insertion histograms     */
    const uint8_t* main_chrom; /* [n_refs] 1 = rID in chrIdset (bamqualcheck.cpp:106-123) */
    const int32_t* fasta_index;/* [n_refs] position of the contig in FASTA order, -1 =
                                  absent (TripletCounting.hpp:254-259); NULL = identity   */
    int32_t device;            /* HIP device ordinal                                     */
    bqc_sketch_options sketch;
    uint32_t shard_tail;       /* 1: this context processes a shard of the record stream that does NOT begin with the
                                  stream's first record (multi-GPU, see bqc_shard_resolve)              */
} bqc_options;

/* ---- input batch (host, structure of arrays) -------------------------------
 * Reads are stored back to back: read i owns
 *   seq   [ sum_{j<i} (l_seq[j]+1)/2 , +(l_seq[i]+1)/2 )   BAM-native 4-bit, high nibble first
 *   qual  [ sum_{j<i} l_seq[j]       , +l_seq[i] )          raw Phred bytes (no +33)
 *   cigar [ sum_{j<i} n_cigar[j]     , +n_cigar[i] )        BAM-native len<<4|op ("MIDNSHP=X")
 * All columns are in BAM orientation (as stored); the library applies the
 * reference's reverse-complement transform (bamqualcheck.cpp:345-350) itself.
 */
typedef struct bqc_batch {
    uint32_t n_reads;
    const uint16_t* flag;   /* BAM flag | BQC_FLAG_* host annotations                       */
    const uint8_t* mapq;
    const uint8_t* lane;    /* index from the RG:Z tag (bamqualcheck.cpp:72-100)            */
    const int32_t* rid;     /* refID, -1 = none                                             */
    const int32_t* pos;     /* 0-based leftmost position                                    */
    const int32_t* tlen;
    const int32_t* nm;      /* first integer-typed NM tag, BQC_NM_ABSENT if none            */
    const int32_t* as;      /* AS tag value, BQC_AS_ABSENT if none                          */
    const uint32_t* l_seq;
    const uint16_t* n_cigar;
    const uint8_t* seq;
    const uint8_t* qual;
    const uint32_t* cigar;
    /* further integer NM tags of a record (QualityCheck.hpp:201-218 loops over all of them) */
    uint32_t n_nm_extra;
    const uint32_t* nm_extra_read; /* read index, ascending */
    const int32_t* nm_extra_val;
} bqc_batch;

/* ---- output: host mirror of `struct Counts` -------------------------------
 * Every array is uint64_t; where the reference type is `unsigned`
 * (32-bit, OverallNumbers.hpp:12-24, QualityCheck.hpp:12-26) the value is
 * already reduced mod 2^32.  `n_*` are the reference's printed lengths
 * (arrays grow on demand there).
 */
typedef struct bqc_mate_counts {          /* QualityCheck (QualityCheck.hpp:8-56), r1 / r2        */
    uint32_t n_cycles;                    /* longest read seen = length of the per-cycle arrays   */
    const uint64_t* dnacount[5];          /* [n_cycles] A C G T N (Dna5 ordinal)                  */
    const uint64_t* qualcount;            /* [n_cycles] sum of (q-33) per cycle                   */
    uint64_t qualcount_readnr;            /* reads seen (unsigned)                                */
    const uint64_t* sc5;                  /* [n_cycles] scposcount_5prime                         */
    const uint64_t* sc3;                  /* [n_cycles] scposcount_3prime                         */
    uint32_t n_Ncount;      const uint64_t* Ncount;      /* n_cycles+1 (0 if no read)            */
    uint32_t n_GCcount;     const uint64_t* GCcount;
    uint32_t n_averageQual; const uint64_t* averageQual;
    uint32_t n_insertSize;  const uint64_t* insertSize;  /* isize+1, always                      */
    uint32_t n_mapQ;        const uint64_t* mapQ;
    uint32_t n_readLength;  const uint64_t* readLength;
    uint32_t n_mismatch;    const uint64_t* mismatch;
    uint32_t n_delhist;     const uint64_t* delhist;
    uint32_t n_inshist;     const uint64_t* inshist;
} bqc_mate_counts;

typedef struct bqc_sketch_counts {        /* one per (q,k): ReadQualityHasher (N1)                */
    uint32_t q, k;
    uint64_t sumCount, F0, f1, F2;
} bqc_sketch_counts;

typedef struct bqc_lane_counts {
    uint64_t scalars[BQC_N_SCALARS];
    uint64_t poscov[BQC_COVSIZE + 1];     /* genome_coverage_histogram                            */
    const uint64_t* eightmer;             /* [65536]                                              */
    bqc_mate_counts mate[2];
    const uint64_t* triplet;              /* [64 ctx][4: fwd1st, fwd2nd, rev1st, rev2nd][4 base]  */
    uint32_t n_sketch;                    /* n_q * n_k, q-major                                   */
    const bqc_sketch_counts* sketch;
} bqc_lane_counts;

typedef struct bqc_counts {
    uint32_t n_lanes;
    const bqc_lane_counts* lanes;
} bqc_counts;

typedef struct bqc_ctx bqc_ctx;
typedef struct bqc_dbatch bqc_dbatch; /* a batch resident in device memory */

/* ---- aggregation (replaces bamqualcheck.cpp:303-453) ----------------------- */
int bqc_abi_version(void);
int bqc_create(const bqc_options* opt, bqc_ctx** out);
/* Starts the HIP runtime on `device` (~0.06 s on a cold process) and makes the streams ahead that a context (and the program's reader)
 * will ask for (~0.05 s more, which then overlap with the caller's other set-up): callable from any thread, once per process, e.g.
 * while the inputs are opened. */
int bqc_warmup(int32_t device);
/* The FASTA order of the contigs (bqc_options.fasta_index) may also be given after creation, before the first batch: a
 * program can then create the context while it still reads the FASTA file. */
int bqc_set_fasta_index(bqc_ctx* ctx, const int32_t* fasta_index);
void bqc_destroy(bqc_ctx* ctx);
const char* bqc_last_error(const bqc_ctx* ctx); /* ctx may be NULL: last create error */

/* Reference chromosome as Dna5 codes (0..4 = A C G T N), one byte per base; the
 * library copies it to device memory (replaces Genome/readFastaRecord,
 * TripletCounting.hpp:60-104). */
int bqc_set_reference(bqc_ctx* ctx, int32_t rid, const uint8_t* dna5, uint64_t len);
/* Optional, before the first bqc_set_reference: one device allocation for contigs of total_bases bases in all, which
 * bqc_set_reference then carves from (an allocation made while kernels run waits for them). */
int bqc_reserve_references(bqc_ctx* ctx, uint64_t total_bases, uint32_t n_contigs);
/* (ONE uploader thread may call bqc_set_reference while another thread submits batches — the program's non-default BQC_BG_REFS=1 mode —
 * under these rules: the contig is one no submitted batch has a read on; the uploader is the only caller of bqc_set_reference /
 * bqc_reserve_references at that time; and its error is read with bqc_last_error only after the submitting thread has stopped.  The call
 * waits for the context's compute stream: it returns when the batches queued before it are through.  No other concurrent use of a
 * context is supported, bqc_anchor_* beside bqc_submit_anchored excepted.) */

/* One batch of decoded records into the pipeline: host pass (coverage anchors), copy into a page-locked staging slot,
 * host-to-device copy, device pre-pass and kernels — three batches in flight, the call returns when the batch is queued.
 * The caller's buffers may be reused on return.  What the device finds wrong with a batch (the first failing read in
 * stream order, as the reference would have met it) surfaces at a later call: bqc_submit, bqc_sync, bqc_flush, bqc_finalize. */
int bqc_submit(bqc_ctx* ctx, const bqc_batch* batch);

/* The same without the staging copy, for callers whose columns live in page-locked memory (bqc_host_register, or
 * hipHostMalloc): the columns are copied to the device straight from the caller's memory and must stay untouched until
 * bqc_batch_uploaded(ctx, ticket, ...) returns 1.  The payload columns (seq, qual, cigar) may also live in the device's own
 * memory (a caller that inflates and decodes there); the fixed columns are read by the host as well and stay host pointers. */
int bqc_submit_async(bqc_ctx* ctx, const bqc_batch* batch, uint64_t* ticket);

/* ---- batches that live on the card entirely (a caller that decodes the records there: the program's reader, csrc/gpu_bam.hip) ----
 * With the FIXED columns in device memory too, the one order-dependent step of the reference's loop — the window state machine of
 * OverallNumbers::coverage, OverallNumbers.hpp:84-110, which bqc_submit* runs on the host over the fixed columns — runs on the card
 * as well (csrc/k_anchor.hip), and the columns never come to the host: it sees a summary of a few hundred bytes per batch.
 *   bqc_anchor_enqueue   queues the anchor kernels and the copy of their summary on `stream` (a hipStream_t on the context's device, on
 *                        which the columns are complete); d_cov: 8 bytes per read of device memory for the anchors, valid — like the
 *                        columns — until the batch's ticket is reported uploaded.  Returns 0 and a handle; 1: not available — the
 *                        context has several read groups, or a batch has gone through bqc_submit* / bqc_upload before (the host
 *                        then keeps the state for the rest of the stream); < 0: -BQC_ERR_*.  (A shard_tail context sets its first
 *                        reads aside on the card exactly as bqc_submit* does on the host.)
 *   bqc_anchor_complete  after the caller has synchronised `stream`: 0 = anchored; 1 = not anchored (more position breaks in the batch
 *                        than the card's serial chain takes: the card has left its state untouched; this batch and every later one
 *                        go through bqc_submit* with host columns); < 0: -BQC_ERR_* (bqc_anchor_error).  `info` (optional): what a
 *                        program wants to know about a batch whose columns it does not have.
 *   bqc_submit_anchored  the batch into the pipeline as bqc_submit_async does (every column of `batch` a device pointer); consumes the handle.
 *                        The kernels read the columns IN PLACE (no copy into the pipeline's own buffers): they and d_cov must stay
 *                        untouched until bqc_batch_uploaded(ticket) says 1, which for such a batch means "its kernels are through";
 *                        at least 512 bytes of the same allocation must lie in front of and behind each of seq, qual and cigar (the
 *                        kernels' vector loads run over the ends).
 * Batches must be anchored in stream order and submitted in the same order; the anchor calls may be made by another thread than the
 * submit calls (the program's decode thread and its submitting thread). */
typedef struct bqc_anchored bqc_anchored;
typedef struct bqc_anchor_info {
    uint32_t n_noqual;        /* primary first / last records without qualities (check_read_len's message, QualityCheck.hpp:70-79) */
    int32_t rid_min, rid_max; /* range of the reference ids in [0, n_refs) the batch holds; rid_min > rid_max: none */
} bqc_anchor_info;
int bqc_anchor_enqueue(bqc_ctx* ctx, const bqc_batch* batch, void* d_cov, void* stream, bqc_anchored** out);
int bqc_anchor_complete(bqc_ctx* ctx, bqc_anchored* a, bqc_anchor_info* info);
int bqc_submit_anchored(bqc_ctx* ctx, const bqc_batch* batch, bqc_anchored* a, uint64_t* ticket);
void bqc_anchor_discard(bqc_ctx* ctx, bqc_anchored* a);
const char* bqc_anchor_error(const bqc_ctx* ctx);
/* 1: the batch's columns have been copied to the device (or the ticket is older than every batch in flight), 0: not yet
 * (only with wait == 0), < 0: -(BQC_ERR_*). */
int bqc_batch_uploaded(bqc_ctx* ctx, uint64_t ticket, int wait);
/* Page-lock / release host memory a caller decodes into (hipHostRegister; ~40 ms per GB once the pages are touched). */
int bqc_host_register(void* p, uint64_t bytes);
int bqc_host_unregister(void* p);

/* Same, split: keep the batch resident in HBM and run the hot path over it
 * any number of times (used by the benchmark, and by callers that overlap
 * upload with compute). */
int bqc_upload(bqc_ctx* ctx, const bqc_batch* batch, bqc_dbatch** out);
int bqc_process(bqc_ctx* ctx, bqc_dbatch* db);
void bqc_dbatch_free(bqc_ctx* ctx, bqc_dbatch* db);
uint64_t bqc_dbatch_bytes(const bqc_dbatch* db); /* algorithmic bytes resident for this batch */

int bqc_sync(bqc_ctx* ctx);  /* wait for all submitted work */
int bqc_reset(bqc_ctx* ctx); /* zero every counter (keeps references) */

/* End of stream: flush the two live coverage windows (bamqualcheck.cpp:447-453)
 * into the state vector.  Idempotent. */
int bqc_flush(bqc_ctx* ctx);

/* ---- shards of one record stream (multi-GPU) ------------------------------------------------------------------
 * Every statistic of the path is a sum over reads except the coverage-depth histogram: its window state machine
 * (OverallNumbers.hpp:84-110) depends on the reads before.  The stream may still be cut anywhere: a context created with
 * shard_tail = 1 sets aside, per read group, its reads up to the first one at which the state machine resets WHATEVER its
 * state (another chromosome, or more than 2000 positions away from the read before) and works normally from there on.
 * When all its batches are in: bqc_shard_resolve(predecessor's exported state) runs the reads set aside from the state the
 * predecessor ended in; bqc_shard_export yields this shard's own final state (its two live windows per read group are then
 * the successor's to flush: the state vector of a shard that has exported holds complete windows only).  The last shard
 * ends with bqc_flush / bqc_finalize as a whole stream does.  The exported block also carries the shard's first and last
 * FASTA position of triplet-eligible reads (words 0 and 1 of the block, int32, -1: none), so that the forward-only FASTA
 * scan (TripletCounting.hpp:254-259) can be checked across shards. */
int bqc_shard_fasta_span(bqc_ctx* ctx, int32_t span[2]); /* first / last FASTA position of the triplet-eligible reads so far (-1: none); waits for the batches in flight */
uint64_t bqc_shard_state_bytes(const bqc_ctx* ctx);
int bqc_shard_resolve(bqc_ctx* ctx, const void* predecessor_state);
int bqc_shard_export(bqc_ctx* ctx, void* state);

/* Flat state vector (uint64 words, pure sums => additive across shards).
 * export/import take DEVICE pointers (e.g. a torch tensor's data_ptr) so a
 * caller can reduce them with RCCL between the two calls. */
uint64_t bqc_state_words(const bqc_ctx* ctx);
int bqc_state_export(bqc_ctx* ctx, void* dev_dst_u64);
int bqc_state_import(bqc_ctx* ctx, const void* dev_src_u64);
int bqc_state_export_host(bqc_ctx* ctx, uint64_t* host_dst);
int bqc_state_import_host(bqc_ctx* ctx, const uint64_t* host_src);

/* bqc_flush + finalisation (lengths, soft-clip prefix sums, empty-lane coverage
 * rule, sketch estimators).  The returned structure is owned by the context and
 * valid until the next finalize/destroy. */
int bqc_finalize(bqc_ctx* ctx, const bqc_counts** out);

/* Timing of the last bqc_process call measured with HIP events on the
 * library's own stream: total and per kernel (ms).  names/ms arrays are owned by ctx. */
int bqc_last_timing(bqc_ctx* ctx, uint32_t* n, const char* const** names, const float** ms);
int bqc_set_timing(bqc_ctx* ctx, int enable);

/* ---- `.bamqc` text (replaces writeOutput, bamqualcheck.cpp:156-233) -------- */
typedef struct bqc_header_info {
    const char* sample_id;          /* SM of the last @RG                                   */
    uint32_t n_names;               /* lane names in output order (lexicographic by ID)     */
    const char* const* lane_names;
    const uint32_t* lane_index;     /* index into counts->lanes for each name               */
} bqc_header_info;
int bqc_write_bamqc(const bqc_counts* counts, const bqc_header_info* hdr, const char* path);

#ifdef __cplusplus
}
#endif
#endif /* BAMQC_H_ */
