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
    // err_code of next_batch when this input needs the host reader instead (a read group that is not in the header, a record the
    // host reader would report, a record walk that cannot be verified, ...): nothing has been reported to the user, the caller
    // starts over with BamReader, which decides what is an error and what is not.
    static const int kUnsupported = -1000;
    GpuBamReader();
    ~GpuBamReader() override;
    // hdr / first_record_u: the header as the host reader parsed it, and the offset of the first record in the uncompressed stream
    // batch_reads / batch_bases: the limits next_batch will be called with (buffers are allocated once, for full batches)
    bool open(const char* path, int device, const BamHeader& hdr, uint64_t first_record_u, size_t batch_reads, size_t batch_bases, std::string& err);
    // The file is read and copied to the card from open() on; the first inflate kernel waits for this call: a caller that still has
    // device set-up of its own to do (allocations, synchronous copies wait for a running kernel) makes it when that is done.
    void allow_kernels();
    BamHeader& header() override { return hdr_; }
    void set_main_chrom(const std::vector<uint8_t>& mc) override { main_ = mc; }
    int next_batch(HostBatch& out, size_t max_reads, size_t max_bases, std::string& err, int& err_code) override;
    uint64_t records() const { return nrec_; }
    double seconds_reading() const { return t_read_; } // time spent in fread
    // set by open() once its device buffers are allocated (also when it fails before that): a caller that creates its own device
    // context in another thread starts doing so from here on
    std::atomic<bool> buffers_allocated{false};

private:
    struct Impl;
    Impl* p_ = nullptr;
    BamHeader hdr_;
    std::vector<uint8_t> main_;
    uint64_t nrec_ = 0;
    double t_read_ = 0;
};
