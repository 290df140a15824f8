// bam_io.h — BAM header / record decoding straight into the SoA batch of include/bamqc.h, and the
// matching writer used by the synthetic generator.  Replaces SeqAn's BamStream / readRecord /
// BamTagsDict (reference src/bamqualcheck.cpp:44-100, 252-267, 303-315; QualityCheck.hpp:201-209;
// TripletCounting.hpp:110-130).  Formats: public SAM/BAM specification.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <utility>
#include <string>
#include <vector>

#include "../../include/bamqc.h"
#include "bgzf.h"
#include "raw_vector.h"

struct BamHeader {
    std::string text;
    std::vector<std::string> ref_names;
    std::vector<uint32_t> ref_lens;
    // getSampleIdAndLaneNames (bamqualcheck.cpp:44-66)
    std::string sample_id;                      // SM of the last @RG that has one
    std::map<std::string, unsigned> lane_names; // @RG ID -> lane index, iterated lexicographically at output
    unsigned lane_count = 0;                    // laneNames.size() after the header (bamqualcheck.cpp:297)
};

struct HostBatch { // owning storage behind a bqc_batch
    raw_vector<uint16_t> flag, n_cigar;
    raw_vector<uint8_t> mapq, lane, seq, qual;
    raw_vector<int32_t> rid, pos, tlen, nm, as;
    raw_vector<uint32_t> l_seq, cigar;
    std::vector<int32_t> nm_extra_val;
    std::vector<uint32_t> nm_extra_read;
    // A batch decoded on the GPU (csrc/gpu_bam.hip) keeps its payload in device memory: these replace seq / qual / cigar in the
    // view.  dev_mem is the batch's device buffer (kept when the batch is recycled), released through dev_free.
    const uint8_t* d_seq = nullptr;
    const uint8_t* d_qual = nullptr;
    const uint32_t* d_cigar = nullptr;
    void* dev_mem = nullptr;
    size_t dev_cap = 0;
    void (*dev_free)(void*) = nullptr;
    // A batch that also keeps its FIXED columns on the card and was anchored there (include/bamqc.h: bqc_anchor_*): the host vectors
    // above are empty, dev holds the device pointers (n_reads and all), `anchored` the handle bqc_submit_anchored takes; what the
    // program wants to know about the reads without seeing them: records without qualities, range of reference ids.
    void* anchored = nullptr;
    bqc_batch dev{};
    uint32_t n_noqual = 0;
    int32_t rid_min = 0, rid_max = -1;
    HostBatch() = default;
    HostBatch(const HostBatch&) = delete;
    HostBatch& operator=(const HostBatch&) = delete;
    ~HostBatch() { if (dev_mem && dev_free) dev_free(dev_mem); }
    bqc_batch view() const;
    void clear();
    size_t n() const { return anchored ? dev.n_reads : flag.size(); }
};

// what the driver needs from an input stream of alignment records
class RecordReader {
public:
    virtual ~RecordReader() {}
    virtual BamHeader& header() = 0;
    virtual void set_main_chrom(const std::vector<uint8_t>& mc) = 0;
    virtual int next_batch(HostBatch& out, size_t max_reads, size_t max_bases, std::string& err, int& err_code) = 0;
};

void parse_read_groups(BamHeader& h); // getSampleIdAndLaneNames, bamqualcheck.cpp:44-66

// a record found by a walk over the block_size chain; off: relative to the walked range; so / qo / co: where its packed bases,
// qualities and CIGAR words go in the batch's payload columns
struct BamRec { size_t off; uint32_t bs, l_seq, n_cig; size_t so, qo, co; uint64_t nrec; };
// The decoder behind BamReader::next_batch, for any byte range whose records are known (also used by the reader on the card for
// a batch it hands over: csrc/gpu_bam.hip): see bam_io.cpp.
bool bam_decode_records(const uint8_t* base, const std::vector<BamRec>& recs, BamHeader& hdr, const std::vector<uint8_t>& main_chrom, unsigned threads, HostBatch& o,
                        std::string& err, int& err_code);

class BamReader : public RecordReader {
public:
    // header_first: the first run of BGZF blocks is a small one (the caller may only want the header: the program when the records
    // are to be decoded on the GPU)
    bool open(const char* path, std::string& err, bool header_first = false);
    // One shard of the record stream (multi-GPU): the header is read from the file's start as always; records are then taken
    // from the first record that STARTS in a BGZF block at or behind bgzf_find_block(begin_hint) — located by the same test
    // the parallel record walk uses for its guesses — up to the last record that starts before the block at
    // bgzf_find_block(end_hint) (records run on into the next shard's blocks).  begin_hint 0: from the first record;
    // end_hint UINT64_MAX: to the end.  Neighbouring shards verify afterwards that the first ended where the second began:
    // range_over() of a shard (bytes of its last record beyond its end block) == range_first() of its successor.
    bool open_range(const char* path, uint64_t begin_hint, uint64_t end_hint, std::string& err);
    uint64_t range_begin_block() const { return range_b0_; }
    uint64_t range_end_block() const { return range_b1_; }
    uint64_t range_first() const { return range_first_; }   // uncompressed offset of the shard's first record in its first block
    uint64_t range_over() const { return range_over_; }     // valid after the last batch: where the next record starts, relative to the end block
    const BamHeader& header() const { return hdr_; }
    BamHeader& header() override { return hdr_; }
    void set_main_chrom(const std::vector<uint8_t>& mc) override { main_ = mc; }
    // keep only records whose refID is selected (keep[rid] != 0; refID -1 follows keep_unplaced): used to shard
    // a coordinate-sorted BAM by chromosome across ranks
    void set_rid_filter(const std::vector<uint8_t>& keep, bool keep_unplaced) { keep_ = keep; keep_unplaced_ = keep_unplaced; filter_ = true; }
    // Decodes up to max_reads records (and at most max_bases bases) into `out`.
    // Returns 1 = batch filled (maybe partially, more may follow), 0 = end of file and nothing read,
    // -1 = error (err set; code in err_code: BQC_ERR_IO for a corrupt file, BQC_ERR_ARG for the RG-tag rule).
    int next_batch(HostBatch& out, size_t max_reads, size_t max_bases, std::string& err, int& err_code) override;
    uint64_t records() const { return nrec_; }
    void close() { bg_.stop(); raw_vector<uint8_t>().swap(buf_); raw_vector<uint8_t>().swap(chunk_); eof_ = true; } // the header stays; no more records
    uint64_t stream_pos() const { return base_u_ + cur_; } // offset of the next record in the uncompressed stream (after open: the first record's)

private:
    bool fill(size_t need, std::string& err); // ensure `need` bytes are available at cur_
    BgzfReader bg_;
    BamHeader hdr_;
    raw_vector<uint8_t> buf_, chunk_;
    std::vector<BamRec> recs_; // (kept from batch to batch: 64 MB per million records)
    struct WalkSeg { size_t first = 0, end = 0; uint64_t n_all = 0; std::vector<BamRec> recs; };
    std::vector<WalkSeg> segs_;
    void walk_segment(const uint8_t* base, size_t avail, size_t a, size_t b, bool exact_start, WalkSeg& out) const;
    void parallel_prewalk(size_t max_reads, size_t max_bases, size_t& rel, size_t& bases, size_t& so, size_t& qo, size_t& co);
    double avg_rec_bytes_ = 0, avg_rec_bases_ = 0; // bytes / bases per record of the previous batch (0: unknown)
    std::string sticky_err_;    // error of the inflating reader, reported whenever more data is asked for
    size_t cur_ = 0;
    bool eof_ = false;
    std::vector<uint8_t> main_, keep_;
    bool filter_ = false, keep_unplaced_ = true;
    uint64_t nrec_ = 0;
    double t_wait_ = 0, t_copy_ = 0; // BQC_TIMING=3
    // shard of the stream (open_range)
    bool ranged_ = false, need_locate_ = false, range_done_ = false;
    uint64_t range_b0_ = 0, range_b1_ = UINT64_MAX, range_first_ = 0, range_over_ = 0;
    uint64_t base_u_ = 0; // position of buf_[0] in the uncompressed stream of bg_
    bool locate_first_record(std::string& err);
    bool beyond_range(size_t rel) const { const uint64_t m = bg_.mark_u(); return m != UINT64_MAX && base_u_ + cur_ + rel >= m; }
};

// SAM text from a stream (the reference reads SAM from stdin when the input is "-": bamqualcheck.cpp:252-260,
// CommandLineParser.hpp:92-107).  Fills the same batch as BamReader: same tag rules, POS - 1, QUAL '*' -> 0xFF bytes.
// Format: public SAM specification.
class SamReader : public RecordReader {
public:
    bool open(FILE* f, std::string& err); // reads the header (@ lines)
    BamHeader& header() override { return hdr_; }
    void set_main_chrom(const std::vector<uint8_t>& mc) override { main_ = mc; }
    int next_batch(HostBatch& out, size_t max_reads, size_t max_bases, std::string& err, int& err_code) override;

private:
    bool getline(std::string& line);
    FILE* f_ = nullptr;
    BamHeader hdr_;
    std::map<std::string, int32_t> ref_index_;
    std::vector<uint8_t> main_;
    std::string pending_; // first record line, read while looking for the end of the header
    bool have_pending_ = false, eof_ = false;
    uint64_t nrec_ = 0;
};

// minimal BAM writer (used by the synthetic generator and tests)
class BamWriter {
public:
    bool open(const char* path, const std::string& header_text, const std::vector<std::string>& ref_names,
              const std::vector<uint32_t>& ref_lens, std::string& err, int level = 1);
    // writes every read of the batch; tags: RG:Z:<lane_ids[lane]>, NM:i (if present), AS:i (if present)
    bool write_batch(const bqc_batch& b, const std::vector<std::string>& lane_ids, uint64_t first_read_index);
    // the same bytes, with the records serialised by all host threads (for the streaming generator)
    bool write_batch_parallel(const bqc_batch& b, const std::vector<std::string>& lane_ids, uint64_t first_read_index);
    bool close() { return bg_.close(); }

private:
    BgzfWriter bg_;
    std::vector<uint8_t> rec_;
    raw_vector<uint8_t> big_;
    std::vector<uint64_t> at_;
};
