// gpu_bam.h — BAM input decoded on the GPU (gpu_bam.hip): the compressed file goes to the card, the inflated bytes never come
// back.  Same interface as the host reader (host/bam_io.h: RecordReader), same columns: the fixed ones arrive in the batch's
// host arrays (the host pass of the aggregation needs them), the payload (bases, qualities, CIGARs) stays in device memory
// (HostBatch::d_seq / d_qual / d_cigar — bqc_submit_async copies from wherever the columns live).
#pragma once
#include <atomic>
#include <cstdint>
#include <string>
#include <vector>

#include "../host/bam_io.h"

class GpuBamReader : public RecordReader {
public:
    // err_code of next_batch when this input needs the host reader instead (a record walk that cannot be verified, a corrupt
    // record, a file that ends inside a record): nothing has been reported to the user, the caller starts over with BamReader, which
    // decides what is an error and what is not.  A BATCH with a record the card does not decode (a read group that is not in the
    // header, a second NM tag, no RG tag, ...) is not such a case: that batch is decoded by the host reader's rules
    // (bam_decode_records) from the bytes on the card, and the run continues.
    static const int kUnsupported = -1000;
    GpuBamReader();
    ~GpuBamReader() override;
    // hdr / first_record_u: the header as the host reader parsed it, and the offset of the first record in the uncompressed stream
    // batch_reads / batch_bases: the limits next_batch will be called with (buffers are allocated once, for full batches)
    bool open(const char* path, int device, const BamHeader& hdr, uint64_t first_record_u, size_t batch_reads, size_t batch_bases, std::string& err);
    // One shard of the record stream (multi-GPU; same meaning as BamReader::open_range, host/bam_io.h): call before open().  Only the
    // bytes of the compressed file from `begin_block` on are read (a BGZF block boundary: bgzf_find_block), and the records taken
    // are those that START before the block at `end_block` (UINT64_MAX: to the end of the file); the last one runs on into the
    // successor's blocks.  begin_block 0: from the first record (first_record_u of open()); otherwise the first record is located
    // by the walk's own guess (three records in a row that look like records) — neighbouring shards verify it afterwards:
    // range_over() of a shard must equal range_first() of its successor.
    void set_range(uint64_t begin_block, uint64_t end_block) { range_b0_ = begin_block; range_b1_ = end_block; ranged_ = true; }
    uint64_t range_first() const { return range_first_; } // uncompressed offset of the shard's first record, relative to its begin block (valid after the first batch)
    uint64_t range_over() const { return range_over_; }   // where the record after the shard's last one starts, relative to the end block (valid after the last batch)
    // The file is read and copied to the card from open() on; the first inflate kernel waits for this call: a caller that still has
    // device set-up of its own to do (allocations, synchronous copies wait for a running kernel) makes it when that is done.
    void allow_kernels();
    BamHeader& header() override { return hdr_; }
    void set_main_chrom(const std::vector<uint8_t>& mc) override { main_ = mc; }
    int next_batch(HostBatch& out, size_t max_reads, size_t max_bases, std::string& err, int& err_code) override;
    uint64_t records() const { return nrec_; }
    uint64_t batches_handed_over() const { return n_handed_over_; } // batches decoded by the host reader's rules (see next_batch)
    // The program's context, once it exists: from then on the batches' coverage anchors are made on the card (include/bamqc.h:
    // bqc_anchor_*) and their fixed columns stay there — HostBatch::anchored — as long as the stream allows it (one read group, not a
    // shard in the middle of a stream, no batch handed over to the host decoder so far).  Without it: columns on the host, as ever.
    void set_anchor_context(bqc_ctx* ctx) { anchor_ctx_ = ctx; }
    uint64_t batches_anchored() const { return n_anchored_; }
    double seconds_reading() const { return t_read_; } // time spent in pread, summed over the reader threads
    double seconds_waiting_for_runs() const { return t_wait_run_; } // time the consumer waited for the next inflated run (file, copy or inflate behind)
    double seconds_producer_waiting_for_chunks() const { return t_wait_chunk_; } // time the producer waited for the ring's reader threads
    // set by open() once its device buffers are allocated (also when it fails before that): a caller that creates its own device
    // context in another thread starts doing so from here on
    std::atomic<bool> buffers_allocated{false};

private:
    struct Impl;
    Impl* p_ = nullptr;
    BamHeader hdr_;
    std::vector<uint8_t> main_;
    uint64_t nrec_ = 0, n_handed_over_ = 0, n_anchored_ = 0;
    std::atomic<bqc_ctx*> anchor_ctx_{nullptr};
    bool anchors_ok_ = true; // (decode thread)
    double t_read_ = 0, t_wait_run_ = 0, t_wait_chunk_ = 0;
    bool ranged_ = false;
    uint64_t range_b0_ = 0, range_b1_ = UINT64_MAX, range_first_ = 0, range_over_ = 0;
};
