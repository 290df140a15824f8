// bgzf.h — BGZF block reader (parallel inflate, inflate_fast.h) and writer (parallel deflate, zlib).
// Replaces SeqAn's BAM stream layer (reference src/bamqualcheck.cpp:262, readRecord :306).
// Format: public SAM/BAM specification (gzip members <= 64 KiB with a "BC" extra field).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "raw_vector.h"

class BgzfReader {
public:
    ~BgzfReader();
    void stop(); // ends the read-ahead and closes the file (what the destructor does); the reader yields nothing more
    bool open(const char* path, std::string& err, unsigned threads = 0);
    // Start at `begin` (the file offset of a BGZF block, see bgzf_find_block) instead of the file's start.  `mark` (a later block
    // boundary, or UINT64_MAX): mark_u() becomes the number of uncompressed bytes this reader yields before that block once it
    // gets there, and from there on it reads in small steps (the caller only needs the rest of a record).
    bool open_at(const char* path, uint64_t begin, uint64_t mark, std::string& err, unsigned threads = 0);
    uint64_t mark_u() const { return mark_u_.load(); }      // UINT64_MAX: not reached yet
    bool mark_missed() const { return mark_missed_.load(); } // the block chain stepped over `mark`: it is not a block boundary
    // Fills `out` with the next run of uncompressed bytes (many blocks at once); returns false at EOF.
    // On a malformed stream sets err and returns false.  A read-ahead thread inflates the following runs meanwhile.
    bool next_chunk(raw_vector<uint8_t>& out, std::string& err);
    uint64_t compressed_bytes_read() const { return cbytes_; }
    void gpu_run_bytes(size_t n) { gpu_run_bytes_ = n; } // compressed bytes per run when the GPU inflates
    // (before the first next_chunk) a small first run, and nothing is read ahead behind it until somebody asks for more: for a
    // caller that may only want the file's first bytes (the header)
    void first_run_bytes(size_t n) { first_run_bytes_ = n; lazy_ = true; }
    unsigned threads() const { return threads_; }

private:
    struct BlockRef { size_t off, csize, usize, uoff; };
    struct Run { std::vector<BlockRef> blocks; size_t consumed = 0, utotal = 0, read_bytes = 0; double read_ms = 0; bool last = false; };
    bool plan_run(raw_vector<uint8_t>& raw, Run& run, std::string& err, bool on_card);
    bool inflate_run(const raw_vector<uint8_t>& raw, const Run& run, raw_vector<uint8_t>& out, std::string& err, struct GpuInflater** gpu, raw_vector<uint8_t>& bounce);
    void read_ahead(int worker);
    bool stopping();
    struct Item { raw_vector<uint8_t> data; std::string err; bool ok = false; };
    std::vector<std::thread> ra_;
    std::mutex plan_m_;                      // plan_run: one worker at a time; guards what follows
    uint64_t plan_seq_ = 0;
    bool plan_done_ = false;
    std::vector<uint8_t> tail_;              // the partial block behind the last planned run
    uint64_t pub_seq_ = 0;                   // (under m_) runs are handed on in the order they were planned
    std::map<uint64_t, Item> done_;          // (under m_) inflated runs waiting for their turn
    uint64_t planned_ = 0, popped_ = 0;      // (under m_) runs planned / taken by the consumer
    static const uint64_t kMaxAhead = 10;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Item> q_;
    std::vector<raw_vector<uint8_t>> spare_; // buffers handed back by next_chunk
    bool ra_started_ = false, ra_done_ = false, stop_ = false;
    FILE* f_ = nullptr;
    unsigned threads_ = 1;
    uint64_t cbytes_ = 0;
    bool eof_ = false;
    uint64_t file_pos_ = 0;          // file offset of raw_[0]
    uint64_t mark_ = UINT64_MAX;     // file offset whose uncompressed position is wanted
    uint64_t u_total_ = 0;           // uncompressed bytes yielded so far
    std::atomic<uint64_t> mark_u_{UINT64_MAX};
    std::atomic<bool> mark_missed_{false};
    std::atomic<bool> gpu_failed_{false};    // the card could not be used: the CPU decoder has taken over for good
    size_t gpu_run_bytes_ = 64u << 20;
    size_t first_run_bytes_ = 0;
    bool lazy_ = false;                      // (under m_) see first_run_bytes
};

// device >= 0: readers inflate their runs on that GPU from their next run on (csrc/gpu_inflate.hip; the CRC-32 of every block is
// still checked on the host); -1 (the default): on the CPU.  Set by the command-line driver once the card is up.
void bgzf_gpu_inflate_device(int device);
uint64_t bgzf_gpu_inflated_blocks(); // blocks inflated on a GPU so far in this process

// The offset of the first BGZF block at or behind `hint` (a gzip member header with the BC subfield whose BSIZE leads to two
// more such headers, or to the end of the file); the file size if there is none.  Every caller gets the same answer for the
// same hint, so two readers that split a file at bgzf_find_block(hint) agree on the boundary.
uint64_t bgzf_find_block(const char* path, uint64_t hint, std::string& err);
uint64_t bgzf_file_size(const char* path);

class BgzfWriter {
public:
    ~BgzfWriter();
    bool open(const char* path, std::string& err, int level = 1, unsigned threads = 0);
    bool write(const void* data, size_t n); // buffered; compresses in parallel when enough data is pending
    bool close();                           // flushes and appends the 28-byte EOF block

private:
    bool flush_pending(bool all);
    size_t put_blocks(const uint8_t* p, size_t n, bool all);
    raw_vector<uint8_t> out_;
    bool failed_ = false;
    FILE* f_ = nullptr;
    int level_ = 1;
    unsigned threads_ = 1;
    std::vector<uint8_t> pend_;
};
