// bgzf.cpp — see bgzf.h
#include "bgzf.h"
#include "crc32_fast.h"
#include "inflate_fast.h"
#include "parallel.h"
#include "../csrc/gpu_inflate.h"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>

namespace {
const size_t kMaxBlock = 65536;
const size_t kWriteBlock = 65280; // uncompressed payload per block (leaves room for incompressible data)


unsigned default_threads() { return bqc_host_threads(); }

template <typename F>
void parallel_for(size_t n, unsigned threads, F f)
{
    if (threads <= 1 || n <= 1) { for (size_t i = 0; i < n; ++i) f(i); return; }
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    const unsigned nt = (unsigned)std::min<size_t>(threads, n);
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&]() { for (size_t i; (i = next.fetch_add(1)) < n;) f(i); });
    for (auto& t : th) t.join();
}
} // namespace

static std::atomic<int> g_gpu_device{-1};
static std::atomic<uint64_t> g_gpu_blocks{0};
void bgzf_gpu_inflate_device(int device) { g_gpu_device = device; }
uint64_t bgzf_gpu_inflated_blocks() { return g_gpu_blocks.load(); }

BgzfReader::~BgzfReader() { stop(); }

void BgzfReader::stop()
{
    if (ra_started_) {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; } // (from here on no worker is added)
        cv_.notify_all();
        for (auto& t : ra_) if (t.joinable()) t.join();
    }
    if (f_) fclose(f_);
    f_ = nullptr;
    std::lock_guard<std::mutex> lk(m_);
    ra_done_ = true; // (a next_chunk after this finds the end of the stream)
    q_.clear(); done_.clear(); spare_.clear();
}

// A worker of the read-ahead: plans a run (file read + block walk: one worker at a time, in file order), inflates it, and files
// it; runs reach the consumer in file order whatever order they finish in.  Worker 0 inflates on the host (every host thread
// on one run).  While a GPU inflates (bgzf_gpu_inflate_device) further workers run beside it, each with its own stream, device
// buffers and page-locked bounce buffers: the card's share comes ON TOP of the host's — the kernel's time is the time a lane
// needs for its block whatever the number of blocks, so several moderate runs in flight fill the card, and the waiting workers
// sleep.  Measured, it does not pay yet (csrc/gpu_inflate.hip): the inflated bytes have to come back, and what the card's
// workers add the host's lose to them (runtime locks, copies); the switch stays off by default.
void BgzfReader::read_ahead(int worker)
{
    raw_vector<uint8_t> raw, bounce;
    GpuInflater* gpu = nullptr;
    struct Bye { GpuInflater*& g; ~Bye() { if (g) bqc_gpu_inflater_destroy(g); } } bye{gpu};
    const bool on_card = worker > 0;
    for (;;) {
        Item it;
        { // reuse a buffer the consumer has handed back: no fresh pages to fault in for every run
            std::lock_guard<std::mutex> lk(m_);
            if (!spare_.empty()) { it.data.swap(spare_.back()); spare_.pop_back(); }
        }
        if (worker == 0 && g_gpu_device.load() >= 0 && !gpu_failed_.load()) { // a GPU has been switched on: its workers start
            static const int workers = getenv("BQC_GI_WORKERS") ? std::min(31, std::max(0, atoi(getenv("BQC_GI_WORKERS")))) : 2;
            std::lock_guard<std::mutex> lk(m_);
            if (!stop_ && ra_.size() == 1) for (int w = 1; w <= workers; ++w) ra_.emplace_back([this, w] { read_ahead(w); });
        }
        { // not too far ahead of the consumer (memory: every run in flight is ~100-200 MB)
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return stop_ || ra_done_ || (planned_ - popped_ < (ra_.size() > 1 ? kMaxAhead : 4u) && !(lazy_ && planned_ >= 1)); });
            if (stop_ || ra_done_) return;
        }
        Run run;
        uint64_t seq;
        {
            std::unique_lock<std::mutex> lk(plan_m_);
            if (on_card) { // (they only take part while the card inflates)
                while (!plan_done_ && !stopping() && (g_gpu_device.load() < 0 || gpu_failed_.load())) { lk.unlock(); std::this_thread::sleep_for(std::chrono::milliseconds(1)); lk.lock(); }
            }
            if (plan_done_ || stopping()) return;
            seq = plan_seq_++;
            { std::lock_guard<std::mutex> lk2(m_); planned_ = plan_seq_; }
            it.ok = plan_run(raw, run, it.err, on_card);
            if (!it.ok) plan_done_ = true;
        }
        if (it.ok) it.ok = inflate_run(raw, run, it.data, it.err, on_card ? &gpu : nullptr, bounce);
        if (it.ok && it.data.empty() && run.last) it.ok = false; // (end of the file, nothing left)
        const bool last = !it.ok;
        {
            std::lock_guard<std::mutex> lk(m_);
            if (stop_ || ra_done_) return; // (ra_done_: an earlier run has failed, what follows it is dropped)
            done_.emplace(seq, std::move(it));
            while (!done_.empty() && done_.begin()->first == pub_seq_ && !ra_done_) { // in file order
                const bool fin = !done_.begin()->second.ok;
                q_.push_back(std::move(done_.begin()->second));
                done_.erase(done_.begin());
                ++pub_seq_;
                if (fin) { ra_done_ = true; done_.clear(); }
            }
            cv_.notify_all();
        }
        if (last) {
            std::lock_guard<std::mutex> lk2(plan_m_);
            plan_done_ = true;
            return;
        }
    }
}

bool BgzfReader::stopping()
{
    std::lock_guard<std::mutex> lk(m_);
    return stop_;
}

bool BgzfReader::next_chunk(raw_vector<uint8_t>& out, std::string& err)
{
    if (!ra_started_) { ra_started_ = true; ra_.reserve(32); ra_.emplace_back([this] { read_ahead(0); }); }
    std::unique_lock<std::mutex> lk(m_);
    if (lazy_ && q_.empty() && !ra_done_ && planned_ >= 1) { lazy_ = false; cv_.notify_all(); } // (somebody does want more than the first run)
    cv_.wait(lk, [&] { return !q_.empty() || ra_done_; });
    if (q_.empty()) { out.clear(); return false; } // (the failing / final item was already consumed)
    Item it = std::move(q_.front());
    q_.pop_front();
    ++popped_;
    cv_.notify_all();
    lk.unlock();
    out.swap(it.data);
    if (it.data.capacity()) { // the caller's previous buffer goes back to the read-ahead thread
        std::lock_guard<std::mutex> lk2(m_);
        if (spare_.size() < (ra_.size() > 1 ? 12u : 4u)) spare_.emplace_back(std::move(it.data));
    }
    if (!it.ok) { err = it.err; out.clear(); return false; }
    return true;
}

bool BgzfReader::open(const char* path, std::string& err, unsigned threads)
{
    f_ = fopen(path, "rb");
    if (!f_) { err = std::string("could not open ") + path; return false; }
    setvbuf(f_, nullptr, _IOFBF, 1 << 22);
    threads_ = threads ? threads : default_threads();
    return true;
}

bool BgzfReader::open_at(const char* path, uint64_t begin, uint64_t mark, std::string& err, unsigned threads)
{
    if (!open(path, err, threads)) return false;
    if (begin && fseeko(f_, (off_t)begin, SEEK_SET) != 0) { err = std::string("could not seek in ") + path; return false; }
    file_pos_ = begin;
    mark_ = mark;
    if (mark_ <= begin) mark_u_ = 0;
    return true;
}

uint64_t bgzf_file_size(const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f) return 0;
    fseeko(f, 0, SEEK_END);
    const uint64_t n = (uint64_t)ftello(f);
    fclose(f);
    return n;
}

uint64_t bgzf_find_block(const char* path, uint64_t hint, std::string& err)
{
    FILE* f = fopen(path, "rb");
    if (!f) { err = std::string("could not open ") + path; return 0; }
    fseeko(f, 0, SEEK_END);
    const uint64_t size = (uint64_t)ftello(f);
    if (hint >= size) { fclose(f); return size; }
    // a window that holds every candidate start within 64 KiB of the hint (a block is at most 64 KiB) and three blocks behind it
    std::vector<uint8_t> w((size_t)std::min<uint64_t>(size - hint, 5 * kMaxBlock));
    fseeko(f, (off_t)hint, SEEK_SET);
    const size_t got = fread(w.data(), 1, w.size(), f);
    fclose(f);
    w.resize(got);
    auto block_at = [&](size_t p, size_t& bsize) { // a BGZF header at w[p]?
        if (p + 18 > w.size()) return false;
        const uint8_t* h = w.data() + p;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return false;
        const size_t xlen = h[10] | (h[11] << 8);
        if (p + 12 + xlen > w.size()) return false;
        size_t x = 12;
        bsize = 0;
        while (x + 4 <= 12 + xlen) {
            const size_t slen = h[x + 2] | (h[x + 3] << 8);
            if (x + 4 + slen > 12 + xlen) return false;
            if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2) bsize = (size_t)(h[x + 4] | (h[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        return bsize >= 12 + xlen + 8;
    };
    for (size_t p = 0; p < w.size(); ++p) {
        size_t q = p, bs = 0;
        int ok = 0;
        for (; ok < 3; ++ok) { // the chain of three headers, or the end of the file
            if (hint + q == size) { ok = 3; break; }
            if (!block_at(q, bs)) break;
            q += bs;
        }
        if (ok == 3) return hint + p;
    }
    return size;
}

// The next run of whole blocks: reads the file (behind what the last run left over), walks the block headers.  false: nothing
// more to read (err empty) or a malformed stream.
bool BgzfReader::plan_run(raw_vector<uint8_t>& raw, Run& run, std::string& err, bool on_card)
{
    run = Run();
    if (eof_) { run.last = true; return false; }
    size_t want = std::max<size_t>((size_t)threads_ * 16 * kMaxBlock, 32u << 20); // compressed bytes per round
    // On the GPU a lane inflates a block: a run has to hold thousands of blocks to fill the card.
    if (on_card) want = std::max<size_t>(want, gpu_run_bytes_);
    if (mark_u_.load() != UINT64_MAX) want = 4 * kMaxBlock;                          // behind the mark: only the rest of a record is wanted
    if (first_run_bytes_) { want = std::max<size_t>(first_run_bytes_, 4 * kMaxBlock); first_run_bytes_ = 0; }
    // the tail of the previous round (a partial block) goes to the front
    const size_t have = tail_.size();
    {
        const size_t cap = raw.capacity();
        if (on_card && cap < have + want) raw.reserve(have + want + (1u << 20)); // (page-locked once: it should not move again)
        raw.resize(have + want);
        if (raw.capacity() != cap) advise_huge(raw);
    }
    if (have) memcpy(raw.data(), tail_.data(), have);
    const auto tt0 = std::chrono::steady_clock::now();
    size_t got = fread(raw.data() + have, 1, want, f_);
    run.read_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tt0).count();
    run.read_bytes = got;
    cbytes_ += got;
    raw.resize(have + got);
    if (got < want) eof_ = true;
    run.last = eof_;
    std::vector<BlockRef>& blocks = run.blocks;
    size_t p = 0, utotal = 0;
    while (p + 18 <= raw.size()) {
        const uint8_t* h = raw.data() + p;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { err = "not a BGZF stream (bad gzip member header)"; return false; }
        const size_t xlen = h[10] | (h[11] << 8);
        if (p + 12 + xlen > raw.size()) break;
        size_t bsize = 0, x = 12;
        while (x + 4 <= 12 + xlen) {
            const size_t slen = h[x + 2] | (h[x + 3] << 8);
            if (x + 4 + slen > 12 + xlen) { err = "corrupt BGZF block (extra subfield runs past the extra field)"; return false; }
            if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2) bsize = (size_t)(h[x + 4] | (h[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        if (!bsize) { err = "BGZF block without BC extra field"; return false; }
        if (bsize < 12 + xlen + 8) { err = "corrupt BGZF block (BSIZE smaller than header + trailer)"; return false; } // (else csize underflows, the trailer lies before the block)
        if (p + bsize > raw.size()) break;
        const uint8_t* t = raw.data() + p + bsize - 8;
        const size_t isize = t[4] | (t[5] << 8) | (t[6] << 16) | ((size_t)t[7] << 24);
        if (isize > kMaxBlock) { err = "BGZF block larger than 64 KiB"; return false; }
        if (mark_u_.load() == UINT64_MAX) { // where the wanted block boundary lies in the uncompressed stream
            if (file_pos_ + p == mark_) mark_u_ = u_total_ + utotal;
            else if (file_pos_ + p > mark_) { mark_missed_ = true; err = "the split point of the file is not a BGZF block boundary"; return false; }
        }
        blocks.push_back(BlockRef{p + 12 + xlen, bsize - 12 - xlen - 8, isize, utotal});
        utotal += isize;
        p += bsize;
    }
    if (mark_u_.load() == UINT64_MAX && eof_ && p == raw.size() && file_pos_ + p == mark_) mark_u_ = u_total_ + utotal; // (mark = end of the file)
    if (eof_ && p != raw.size()) { err = "truncated BGZF file"; return false; }
    tail_.assign(raw.begin() + p, raw.end());
    run.consumed = p;
    run.utotal = utotal;
    file_pos_ += p;
    u_total_ += utotal;
    return true;
}

// Inflates the blocks of a planned run into `out` and checks their CRC-32.  gpu != nullptr: on the card — the inflated bytes
// arrive in the worker's page-locked bounce buffer and are copied to `out` (a buffer that circulates between the workers and
// the consumer) while their CRC is taken.
bool BgzfReader::inflate_run(const raw_vector<uint8_t>& raw, const Run& run, raw_vector<uint8_t>& out, std::string& err, GpuInflater** gpu, raw_vector<uint8_t>& bounce)
{
    static const bool gi_timing = getenv("BQC_GI_TIMING") != nullptr;
    const auto tt1 = std::chrono::steady_clock::now();
    const std::vector<BlockRef>& blocks = run.blocks;
    { const size_t cap = out.capacity(); out.resize(run.utotal); if (out.capacity() != cap) advise_huge(out); }
    std::atomic<bool> bad{false};
    bool on_gpu = false;
    const int gpu_dev = g_gpu_device.load();
    if (gpu && gpu_dev >= 0 && !*gpu && !gpu_failed_.load()) { *gpu = bqc_gpu_inflater_create(gpu_dev); if (!*gpu) gpu_failed_ = true; }
    if (gpu && *gpu && gpu_dev >= 0 && blocks.size() >= 64) {
        if (bounce.capacity() < run.utotal + 64) { bounce.clear(); bounce.reserve(run.utotal + run.utotal / 4 + (1u << 20)); advise_huge(bounce); }
        bounce.resize(run.utotal);
        if (auto pin = bqc_raw_vector_pin_hook.load(std::memory_order_acquire)) { // page-locked: the copies run at the link's speed and the waiting thread sleeps
            pin(raw.data(), raw.capacity());
            pin(bounce.data(), bounce.capacity());
        }
        std::vector<GiBlock> gb;
        gb.reserve(blocks.size());
        for (const BlockRef& b : blocks) if (b.usize) gb.push_back(GiBlock{b.off, b.uoff, (uint32_t)b.csize, (uint32_t)b.usize});
        const int rc = bqc_gpu_inflate(*gpu, raw.data(), run.consumed, gb.data(), gb.size(), bounce.data(), run.utotal);
        if (rc > 0) { err = "BGZF block failed to inflate (corrupt data)"; return false; }
        if (rc < 0) { bqc_gpu_inflater_destroy(*gpu); *gpu = nullptr; gpu_failed_ = true; } // the card cannot be used: the CPU decoder takes over
        else { on_gpu = true; g_gpu_blocks += gb.size(); }
    }
    const auto tt2 = std::chrono::steady_clock::now();
    parallel_for(blocks.size(), on_gpu ? std::min(threads_, 2u) : threads_, [&](size_t i) { // (the host's threads belong to worker 0)
        const BlockRef& b = blocks[i];
        if (b.usize == 0) return;
        const uint8_t* t = raw.data() + b.off + b.csize;
        const uint32_t crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
        if (on_gpu) {
            if (bqc_crc32_fast(bounce.data() + b.uoff, b.usize) != crc) bad = true;
            memcpy(out.data() + b.uoff, bounce.data() + b.uoff, b.usize);
            return;
        }
        static thread_local Inflater inf;
        // (the 8 bytes after the deflate data, which the decoder may load but not use, are the block's CRC32 / ISIZE)
        if (!inf.run(raw.data() + b.off, b.csize, out.data() + b.uoff, b.usize)) { bad = true; return; }
        if (bqc_crc32_fast(out.data() + b.uoff, b.usize) != crc) bad = true;
    });
    if (bad) { err = "BGZF block failed to inflate (corrupt data)"; return false; }
    if (gi_timing) {
        const auto tt3 = std::chrono::steady_clock::now();
        fprintf(stderr, "[bgzf run] read %.1f MB in %.2f ms, %s %.2f ms, %s %.2f ms\n", run.read_bytes / 1e6, run.read_ms, on_gpu ? "gpu inflate" : "-",
                std::chrono::duration<double, std::milli>(tt2 - tt1).count(), on_gpu ? "crc + copy" : "cpu inflate + crc", std::chrono::duration<double, std::milli>(tt3 - tt2).count());
    }
    return true;
}

BgzfWriter::~BgzfWriter() { if (f_) close(); }

bool BgzfWriter::open(const char* path, std::string& err, int level, unsigned threads)
{
    f_ = fopen(path, "wb");
    if (!f_) { err = std::string("could not open ") + path + " for writing"; return false; }
    setvbuf(f_, nullptr, _IOFBF, 1 << 22);
    level_ = level;
    threads_ = threads ? threads : default_threads();
    return true;
}

static size_t compress_block(const uint8_t* src, size_t n, uint8_t* dst, int level)
{
    static const uint8_t hdr[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, 0, 0};
    memcpy(dst, hdr, 18);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = (Bytef*)src; zs.avail_in = (uInt)n;
    zs.next_out = dst + 18; zs.avail_out = (uInt)(kMaxBlock - 18 - 8);
    int rc = deflate(&zs, Z_FINISH);
    size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) { // incompressible: store
        memset(&zs, 0, sizeof zs);
        deflateInit2(&zs, 0, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = (Bytef*)src; zs.avail_in = (uInt)n;
        zs.next_out = dst + 18; zs.avail_out = (uInt)(kMaxBlock - 18 - 8);
        deflate(&zs, Z_FINISH);
        clen = zs.total_out;
        deflateEnd(&zs);
    }
    const size_t bsize = 18 + clen + 8;
    dst[16] = (uint8_t)((bsize - 1) & 0xFF); dst[17] = (uint8_t)((bsize - 1) >> 8);
    const uint32_t crc = bqc_crc32_fast(src, n);
    uint8_t* t = dst + 18 + clen;
    t[0] = crc & 0xFF; t[1] = (crc >> 8) & 0xFF; t[2] = (crc >> 16) & 0xFF; t[3] = (crc >> 24) & 0xFF;
    t[4] = n & 0xFF; t[5] = (n >> 8) & 0xFF; t[6] = (n >> 16) & 0xFF; t[7] = (n >> 24) & 0xFF;
    return bsize;
}

// compresses the whole blocks of [p, p + n) (and the last partial one when `all`) in parallel and writes them; returns the bytes consumed
size_t BgzfWriter::put_blocks(const uint8_t* p, size_t n, bool all)
{
    const size_t nblk = all ? (n + kWriteBlock - 1) / kWriteBlock : n / kWriteBlock;
    if (!nblk) return 0;
    out_.resize(nblk * kMaxBlock);
    std::vector<size_t> sz(nblk);
    parallel_for(nblk, threads_, [&](size_t i) {
        const size_t off = i * kWriteBlock, m = std::min(kWriteBlock, n - off);
        sz[i] = compress_block(p + off, m, out_.data() + i * kMaxBlock, level_);
    });
    // neighbouring blocks are packed together first: one write per run instead of one per block
    size_t w = 0;
    for (size_t i = 0; i < nblk; ++i) { if (w != i * kMaxBlock) memmove(out_.data() + w, out_.data() + i * kMaxBlock, sz[i]); w += sz[i]; }
    if (fwrite(out_.data(), 1, w, f_) != w) { failed_ = true; return 0; }
    return std::min(n, nblk * kWriteBlock);
}

bool BgzfWriter::flush_pending(bool all)
{
    const size_t used = put_blocks(pend_.data(), pend_.size(), all);
    if (failed_) return false;
    pend_.erase(pend_.begin(), pend_.begin() + used);
    return true;
}

bool BgzfWriter::write(const void* data, size_t n)
{
    const uint8_t* p = (const uint8_t*)data;
    if (n >= (size_t)threads_ * 16 * kWriteBlock) { // a large buffer: top up the pending block, then compress straight from the caller's memory
        if (!pend_.empty()) {
            const size_t fill = std::min(n, (kWriteBlock - pend_.size() % kWriteBlock) % kWriteBlock);
            pend_.insert(pend_.end(), p, p + fill);
            p += fill; n -= fill;
            if (!flush_pending(false)) return false;
        }
        if (pend_.empty()) {
            const size_t used = put_blocks(p, n, false);
            if (failed_) return false;
            p += used; n -= used;
        }
    }
    pend_.insert(pend_.end(), p, p + n);
    if (pend_.size() >= (size_t)threads_ * 16 * kWriteBlock) return flush_pending(false);
    return true;
}

bool BgzfWriter::close()
{
    if (!f_) return true;
    bool ok = flush_pending(true);
    static const uint8_t eof[28] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    ok = ok && fwrite(eof, 1, 28, f_) == 28;
    ok = (fclose(f_) == 0) && ok;
    f_ = nullptr;
    return ok;
}
