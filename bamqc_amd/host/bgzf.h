// bgzf.h — BGZF block reader (parallel inflate, inflate_fast.h) and writer (parallel deflate, zlib).
// Replaces SeqAn's BAM stream layer (reference src/bamqualcheck.cpp:262, readRecord :306).
// Format: public SAM/BAM specification (gzip members <= 64 KiB with a "BC" extra field).
#pragma once
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "raw_vector.h"

class BgzfReader {
public:
    ~BgzfReader();
    bool open(const char* path, std::string& err, unsigned threads = 0);
    // Fills `out` with the next run of uncompressed bytes (many blocks at once); returns false at EOF.
    // On a malformed stream sets err and returns false.  A read-ahead thread inflates the following runs meanwhile.
    bool next_chunk(raw_vector<uint8_t>& out, std::string& err);
    uint64_t compressed_bytes_read() const { return cbytes_; }
    unsigned threads() const { return threads_; }

private:
    bool next_chunk_sync(raw_vector<uint8_t>& out, std::string& err);
    void read_ahead();
    struct Item { raw_vector<uint8_t> data; std::string err; bool ok = false; };
    std::thread ra_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Item> q_;
    std::vector<raw_vector<uint8_t>> spare_; // buffers handed back by next_chunk
    bool ra_started_ = false, ra_done_ = false, stop_ = false;
    FILE* f_ = nullptr;
    unsigned threads_ = 1;
    uint64_t cbytes_ = 0;
    bool eof_ = false;
    raw_vector<uint8_t> raw_;
};

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
