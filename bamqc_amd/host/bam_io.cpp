// bam_io.cpp — see bam_io.h
#include "bam_io.h"
#include "parallel.h"

#include <algorithm>
#include <cctype>
#include <climits>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <thread>

bqc_batch HostBatch::view() const
{
    if (anchored) return dev;
    bqc_batch b;
    memset(&b, 0, sizeof b);
    b.n_reads = (uint32_t)flag.size();
    b.flag = flag.data(); b.mapq = mapq.data(); b.lane = lane.data(); b.rid = rid.data(); b.pos = pos.data();
    b.tlen = tlen.data(); b.nm = nm.data(); b.as = as.data(); b.l_seq = l_seq.data(); b.n_cigar = n_cigar.data();
    b.seq = d_seq ? d_seq : seq.data(); b.qual = d_qual ? d_qual : qual.data(); b.cigar = d_cigar ? d_cigar : cigar.data();
    b.n_nm_extra = (uint32_t)nm_extra_read.size();
    b.nm_extra_read = nm_extra_read.data(); b.nm_extra_val = nm_extra_val.data();
    return b;
}
void HostBatch::clear()
{
    flag.clear(); n_cigar.clear(); mapq.clear(); lane.clear(); seq.clear(); qual.clear(); rid.clear(); pos.clear();
    tlen.clear(); nm.clear(); as.clear(); nm_extra_val.clear(); l_seq.clear(); cigar.clear(); nm_extra_read.clear();
    d_seq = d_qual = nullptr; d_cigar = nullptr; // (dev_mem stays: the next batch reuses it)
    anchored = nullptr; memset(&dev, 0, sizeof dev); n_noqual = 0; rid_min = 0; rid_max = -1;
}

static inline uint32_t rd32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

bool BamReader::fill(size_t need, std::string& err)
{
    while (buf_.size() - cur_ < need) {
        if (eof_) { if (!sticky_err_.empty()) err = sticky_err_; return false; }
        if (cur_ > (1u << 22)) { buf_.erase(buf_.begin(), buf_.begin() + cur_); base_u_ += cur_; cur_ = 0; }
        std::string e;
        const auto w0 = std::chrono::steady_clock::now();
        const bool got = bg_.next_chunk(chunk_, e);
        const auto w1 = std::chrono::steady_clock::now();
        t_wait_ += std::chrono::duration<double>(w1 - w0).count();
        if (!got) {
            eof_ = true;
            if (!e.empty()) { sticky_err_ = e; err = e; return false; }
            continue;
        }
        if (buf_.capacity() < buf_.size() + chunk_.size()) { // grow rarely and far: reallocation copies the whole batch read so far
            try { buf_.reserve(std::max<size_t>(768u << 20, 2 * (buf_.size() + chunk_.size()))); advise_huge(buf_); } catch (const std::bad_alloc&) {}
        }
        { // append the run: a plain copy of ~100 MB, split over a few threads (one thread manages ~10 GB/s)
            const size_t at = buf_.size(), n = chunk_.size();
            buf_.resize(at + n);
            parallel_ranges(n, std::min(4u, bg_.threads()), 8u << 20, [&](unsigned, size_t lo, size_t hi) { memcpy(buf_.data() + at + lo, chunk_.data() + lo, hi - lo); });
        }
        t_copy_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - w1).count();
    }
    return true;
}

// getSampleIdAndLaneNames, bamqualcheck.cpp:44-66
void parse_read_groups(BamHeader& h)
{
    size_t p = 0;
    const std::string& t = h.text;
    while (p < t.size()) {
        size_t e = t.find('\n', p);
        if (e == std::string::npos) e = t.size();
        if (e - p >= 3 && t.compare(p, 3, "@RG") == 0) {
            size_t q = p + 3;
            while (q < e) {
                if (t[q] == '\t') { ++q; continue; }
                size_t f = t.find('\t', q);
                if (f == std::string::npos || f > e) f = e;
                if (f - q >= 3 && t[q + 2] == ':') {
                    const std::string key = t.substr(q, 2);
                    std::string val = t.substr(q + 3, f - q - 3);
                    if (!val.empty() && val.back() == '\r') val.pop_back();
                    if (key == "ID") { const unsigned l = (unsigned)h.lane_names.size(); h.lane_names[val] = l; }
                    if (key == "SM") h.sample_id = val;
                }
                q = f;
            }
        }
        p = e + 1;
    }
    h.lane_count = (unsigned)h.lane_names.size();
}

bool BamReader::open(const char* path, std::string& err, bool header_first)
{
    if (!bg_.open(path, err)) return false;
    if (header_first) bg_.first_run_bytes(1u << 20);
    std::string e;
    if (!fill(12, e) || memcmp(buf_.data(), "BAM\1", 4) != 0) { err = e.empty() ? "not a BAM file" : e; return false; }
    const uint32_t l_text = rd32(buf_.data() + 4);
    if (!fill(12 + (size_t)l_text, e)) { err = "truncated BAM header"; return false; }
    hdr_.text.assign((const char*)buf_.data() + 8, l_text);
    while (!hdr_.text.empty() && hdr_.text.back() == '\0') hdr_.text.pop_back();
    cur_ = 8 + l_text;
    const uint32_t n_ref = rd32(buf_.data() + cur_);
    cur_ += 4;
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (!fill(4, e)) { err = "truncated BAM header"; return false; }
        const uint32_t l_name = rd32(buf_.data() + cur_);
        if (!fill(8 + (size_t)l_name, e)) { err = "truncated BAM header"; return false; }
        std::string name((const char*)buf_.data() + cur_ + 4, l_name ? l_name - 1 : 0);
        hdr_.ref_names.push_back(name);
        hdr_.ref_lens.push_back(rd32(buf_.data() + cur_ + 4 + l_name));
        cur_ += 8 + l_name;
    }
    parse_read_groups(hdr_);
    return true;
}

// One record's variable part, decoded by parse_record.
namespace {
struct RecErr { size_t index = SIZE_MAX; std::string msg; int code = 0; };

} // namespace

// ---- parallel record walk ---------------------------------------------------------------------------------------------
// The block_size chain is a dependent walk (~45 ns per record: one cache miss each).  With the batch's bytes already in the
// buffer it is split into segments walked by several threads: every thread but the first has to GUESS where a record
// starts in its segment (the first offset at which three records in a row look like records) — a guess that is only used
// if the walk of the previous segment ends exactly there; otherwise that segment is walked again from the known position.
// The result is the serial walk's, whatever the data.
static inline bool plausible_record(const uint8_t* base, size_t avail, size_t p, int32_t n_ref, size_t& next)
{
    if (p + 36 > avail) return false;
    const uint32_t bs = rd32(base + p);
    if (bs < 32 || bs > (1u << 28)) return false;
    const uint8_t* r = base + p + 4;
    const int32_t rid = (int32_t)rd32(r), pos = (int32_t)rd32(r + 4), rnext = (int32_t)rd32(r + 20), pnext = (int32_t)rd32(r + 24);
    if (rid < -1 || rid >= n_ref || rnext < -1 || rnext >= n_ref || pos < -1 || pnext < -1) return false;
    const uint32_t l_name = r[8], n_cig = rd16(r + 12), l_seq = rd32(r + 16);
    if (l_name == 0 || l_seq > (1u << 28)) return false;
    const size_t var = 32 + (size_t)l_name + 4ull * n_cig + (l_seq + 1) / 2 + l_seq;
    if (var > bs) return false;
    if (p + 4 + 32 + l_name <= avail && r[32 + l_name - 1] != 0) return false; // read name is NUL-terminated
    next = p + 4 + (size_t)bs;
    return true;
}

bool BamReader::open_range(const char* path, uint64_t begin_hint, uint64_t end_hint, std::string& err)
{
    if (!open(path, err)) return false; // header (from the file's start)
    ranged_ = true;
    const uint64_t size = bgzf_file_size(path);
    range_b1_ = end_hint >= size ? UINT64_MAX : bgzf_find_block(path, end_hint, err);
    if (!err.empty()) return false;
    if (range_b1_ >= size) range_b1_ = UINT64_MAX;
    if (begin_hint == 0) { // the first shard keeps reading where the header ended; its stream started at offset 0
        range_b0_ = 0;
        if (range_b1_ != UINT64_MAX) { // (reopen with the mark: the header is parsed again from the same bytes)
            const size_t keep_cur = cur_;
            bg_.~BgzfReader();
            new (&bg_) BgzfReader();
            buf_.clear(); cur_ = 0; base_u_ = 0; eof_ = false; sticky_err_.clear();
            if (!bg_.open_at(path, 0, range_b1_, err)) return false;
            std::string e;
            if (!fill(keep_cur, e)) { err = e.empty() ? "truncated BAM header" : e; return false; }
            cur_ = keep_cur;
        }
        return true;
    }
    range_b0_ = bgzf_find_block(path, begin_hint, err);
    if (!err.empty()) return false;
    bg_.~BgzfReader();
    new (&bg_) BgzfReader();
    buf_.clear(); cur_ = 0; base_u_ = 0; eof_ = false; sticky_err_.clear();
    if (range_b0_ >= size || (range_b1_ != UINT64_MAX && range_b0_ >= range_b1_)) { range_done_ = true; eof_ = true; range_b0_ = std::min(range_b0_, size); return true; } // empty shard
    if (!bg_.open_at(path, range_b0_, range_b1_, err)) return false;
    need_locate_ = true;
    return true;
}

// first record of a shard that starts in the middle of the file: the first offset at which three records in a row look like
// records (the test of the parallel record walk); the predecessor shard confirms it afterwards (range_over / range_first)
bool BamReader::locate_first_record(std::string& err)
{
    need_locate_ = false;
    std::string e;
    (void)fill(4u << 20, e); // best effort: a small shard has less
    if (!e.empty()) { err = e; return false; }
    const size_t avail = buf_.size() - cur_;
    const uint8_t* base = buf_.data() + cur_;
    const int32_t n_ref = (int32_t)hdr_.ref_names.size();
    static const bool skew = getenv("BQC_TEST_SHARD_SKEW") != nullptr; // tests: a wrong guess (the second record found), to exercise the fallback
    for (size_t p = 0; p < avail; ++p) {
        if (beyond_range(p)) break; // no record starts inside this shard
        size_t q1, q2, q3;
        const bool hit = plausible_record(base, avail, p, n_ref, q1) &&
                         (q1 >= avail || (plausible_record(base, avail, q1, n_ref, q2) && (q2 >= avail || plausible_record(base, avail, q2, n_ref, q3))));
        if (!hit) continue;
        if (skew && q1 < avail) p = q1;
        range_first_ = p;
        cur_ += p;
        return true;
    }
    range_first_ = avail; // nothing starts here: the shard is empty, the predecessor's last record covers it
    range_done_ = true;
    range_over_ = 0;
    return true;
}

void BamReader::walk_segment(const uint8_t* base, size_t avail, size_t a, size_t b, bool exact_start, WalkSeg& out) const
{
    out.recs.clear();
    out.n_all = 0;
    size_t p = a;
    if (!exact_start) {
        const int32_t n_ref = (int32_t)hdr_.ref_names.size();
        for (; p < b; ++p) {
            size_t q1, q2, q3;
            if (plausible_record(base, avail, p, n_ref, q1) && plausible_record(base, avail, q1, n_ref, q2) && plausible_record(base, avail, q2, n_ref, q3)) break;
        }
        static const bool skew = getenv("BQC_TEST_WALK_SKEW") != nullptr; // tests: make every guess wrong (start at the second record found)
        if (skew && p < b) { size_t q; if (plausible_record(base, avail, p, n_ref, q)) p = q; }
        if (p >= b) { out.first = out.end = SIZE_MAX; return; }
    }
    out.first = p;
    while (p < b) {
        if (p + 36 > avail) break;
        const uint32_t bs = rd32(base + p);
        if (bs < 32) break; // (the serial walk reports it when it gets here)
        const uint8_t* r = base + p + 4;
        const int32_t rid = (int32_t)rd32(r);
        const uint32_t l_name = r[8], n_cig = rd16(r + 12), l_seq = rd32(r + 16);
        const size_t var = 32 + (size_t)l_name + 4ull * n_cig + (l_seq + 1) / 2 + l_seq;
        if (var > bs || p + 4 + (size_t)bs > avail) break;
        bool keep = true;
        if (filter_) keep = rid < 0 ? keep_unplaced_ : ((size_t)rid < keep_.size() && keep_[rid]);
        if (keep) out.recs.push_back(BamRec{p, bs, l_seq, n_cig, 0, 0, 0, out.n_all});
        ++out.n_all;
        p += 4 + (size_t)bs;
    }
    out.end = p;
}

void BamReader::parallel_prewalk(size_t max_reads, size_t max_bases, size_t& rel, size_t& bases, size_t& so, size_t& qo, size_t& co)
{
    const unsigned T = std::min(16u, bg_.threads());
    if (avg_rec_bytes_ <= 0 || T < 2 || max_reads < 65536) return;
    {
        // records this batch will hold: its read limit or, for long reads, its base limit
        const double n_exp = std::min((double)max_reads, avg_rec_bases_ > 0 ? (double)max_bases / avg_rec_bases_ + 1.0 : (double)max_reads);
        double want = std::min(n_exp * avg_rec_bytes_ * 1.02, 3.0e9);
        if (ranged_ && bg_.mark_u() != UINT64_MAX) { // a shard: nothing is wanted far behind its end
            const uint64_t here = base_u_ + cur_, m = bg_.mark_u();
            want = std::min(want, (double)(m > here ? m - here : 0) + (double)(1u << 20));
        }
        std::string e;
        (void)fill((size_t)want, e); // best effort: at the end of the file (or a damaged block) less is there
    }
    const size_t avail = buf_.size() - cur_;
    if (avail < ((size_t)T << 22)) return;
    const uint8_t* base = buf_.data() + cur_;
    if (segs_.size() < T) segs_.resize(T);
    const size_t step = avail / T;
    parallel_ranges(T, T, 1, [&](unsigned, size_t lo, size_t hi) {
        for (size_t t = lo; t < hi; ++t) walk_segment(base, avail, t * step, t + 1 == T ? avail : (t + 1) * step, t == 0, segs_[t]);
    });
    std::vector<BamRec>& recs = recs_;
    size_t pos = 0;
    uint64_t n_all = 0;
    for (unsigned t = 0; t < T; ++t) {
        const size_t b = t + 1 == T ? avail : (t + 1) * step;
        WalkSeg& S = segs_[t];
        if (S.first != pos) { // the guess was not where the chain arrives: this segment again, from the known position
            if (pos < b) walk_segment(base, avail, pos, b, true, S);
            else { S.recs.clear(); S.n_all = 0; S.first = S.end = pos; }
        }
        for (const BamRec& r : S.recs) { recs.push_back(r); recs.back().nrec = nrec_ + n_all + r.nrec; }
        n_all += S.n_all;
        pos = S.end;
        if (pos < b) break; // an incomplete or invalid record: the serial walk continues (and reports) from here
    }
    // the batch's limits and the payload offsets, in record order
    size_t n_take = 0;
    for (; n_take < recs.size() && n_take < max_reads && bases < max_bases; ++n_take) {
        BamRec& r = recs[n_take];
        if (ranged_ && beyond_range(r.off)) break; // starts behind the shard's end: the serial walk below notes where
        r.so = so; r.qo = qo; r.co = co;
        so += (r.l_seq + 1) / 2; qo += r.l_seq; co += r.n_cig;
        bases += r.l_seq;
    }
    if (n_take < recs.size()) { // a limit was reached inside the window: the rest is walked again by the next batch
        recs.resize(n_take);
        rel = n_take ? recs.back().off + 4 + (size_t)recs.back().bs : 0;
        nrec_ = n_take ? recs.back().nrec + 1 : nrec_;
    } else {
        rel = pos;
        nrec_ += n_all;
    }
}

// Decoding is done in two steps so that it can use every host core: (1) a serial walk over the block_size chain that
// finds the record boundaries and fixes where every record's seq / qual / CIGAR goes in the batch (prefix sums);
// (2) the records are decoded into the columns in parallel (tag scan, copies).  Unknown RG ids (std::map::operator[]
// inserts them with lane 0, bamqualcheck.cpp:86) and the per-record error rules are resolved in record order afterwards.
int BamReader::next_batch(HostBatch& o, size_t max_reads, size_t max_bases, std::string& err, int& err_code)
{
    o.clear();
    err_code = 0;
    static const bool timing = getenv("BQC_TIMING") && getenv("BQC_TIMING")[0] == '3';
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<BamRec>& recs = recs_;
    recs.clear();
    if (range_done_) return 0;
    if (need_locate_) {
        if (!locate_first_record(err)) { err_code = BQC_ERR_IO; return -1; }
        if (range_done_) return 0;
    }
    recs.reserve(std::min<size_t>(max_reads, 1u << 20));
    size_t bases = 0, so = 0, qo = 0, co = 0;
    size_t rel = 0; // bytes of this batch walked so far; cur_ stays at the batch start, so fill() never drops them
    size_t released = 0; // bytes of skipped records given back before the first kept one (refID filter)
    bool io_error = false;
    const uint64_t nrec_at_start = nrec_;
    parallel_prewalk(max_reads, max_bases, rel, bases, so, qo, co); // (does nothing for the first batch: record size unknown)
    while (recs.size() < max_reads && bases < max_bases) {
        // With a refID filter nothing may have been kept yet: those bytes are released at once (cur_ moves on; fill() drops what
        // lies before it), otherwise a rank whose chromosomes sit late in the file would hold everything before them in memory.
        if (recs.empty() && rel > (8u << 20)) { cur_ += rel; released += rel; rel = 0; }
        if (ranged_ && beyond_range(rel)) { // the next record starts in the successor's blocks: this shard ends here
            range_over_ = base_u_ + cur_ + rel - bg_.mark_u();
            range_done_ = true;
            break;
        }
        // fast path: the whole record is already in the buffer (fill() is only called when it is not)
        size_t avail = buf_.size() - cur_;
        if (avail < rel + 4 || avail < rel + 4 + (size_t)rd32(buf_.data() + cur_ + rel)) {
            std::string e;
            if (!fill(rel + 4, e)) {
                if (!e.empty() || buf_.size() != cur_ + rel) { err = e.empty() ? "truncated BAM record" : e; io_error = true; }
                break;
            }
            const uint32_t bs0 = rd32(buf_.data() + cur_ + rel);
            if (bs0 < 32 || !fill(rel + 4 + (size_t)bs0, e)) { err = e.empty() ? "truncated BAM record" : e; io_error = true; break; }
        }
        const uint8_t* r = buf_.data() + cur_ + rel + 4;
        const uint32_t bs = rd32(r - 4);
        if (bs < 32) { err = "truncated BAM record"; io_error = true; break; }
        const int32_t rid = (int32_t)rd32(r);
        const uint32_t l_name = r[8], n_cig = rd16(r + 12), l_seq = rd32(r + 16);
        const size_t var = 32 + (size_t)l_name + 4ull * n_cig + (l_seq + 1) / 2 + l_seq;
        if (var > bs) { err = "corrupt BAM record"; io_error = true; break; }
        bool keep = true;
        if (filter_) keep = rid < 0 ? keep_unplaced_ : ((size_t)rid < keep_.size() && keep_[rid]);
        if (keep) {
            recs.push_back(BamRec{rel, bs, l_seq, n_cig, so, qo, co, nrec_});
            so += (l_seq + 1) / 2; qo += l_seq; co += n_cig;
            bases += l_seq;
        }
        rel += 4 + (size_t)bs;
        ++nrec_;
    }
    if (nrec_ > nrec_at_start) {
        avg_rec_bytes_ = (double)(released + rel) / (double)(nrec_ - nrec_at_start);
        avg_rec_bases_ = recs.empty() ? 0.0 : (double)bases / (double)recs.size();
    }
    const auto t1 = std::chrono::steady_clock::now();
    const size_t n = recs.size();
    std::string derr;
    int dcode = 0;
    const bool decoded = bam_decode_records(buf_.data() + cur_, recs, hdr_, main_, bg_.threads(), o, derr, dcode);
    cur_ += rel; // the decoded records may now be dropped from the buffer
    if (timing) {
        const auto t2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[timing] batch of %zu records: walk %.3f s (of which waiting for inflated data %.3f s, appending it %.3f s), parallel decode %.3f s\n", n,
                std::chrono::duration<double>(t1 - t0).count(), t_wait_, t_copy_, std::chrono::duration<double>(t2 - t1).count());
        t_wait_ = t_copy_ = 0;
    }
    // the first error in record order wins; an I/O error of the walk comes after every indexed record
    if (!decoded) { err = derr; err_code = dcode; return -1; }
    if (io_error) { err_code = BQC_ERR_IO; return -1; }
    return n ? 1 : 0;
}

// The records `recs` of the byte range at `base` into the batch's columns, by all host threads: tag scan (RG -> lane through the
// header's table, every integer NM, first AS) and copies of the payload to where the walk has placed it.  Unknown RG ids
// (std::map::operator[] inserts them with lane 0, bamqualcheck.cpp:86) and the per-record error rules are resolved in record
// order afterwards.  false: a record the reference's rules end the run at (the first one in record order; err / err_code).
bool bam_decode_records(const uint8_t* base, const std::vector<BamRec>& recs, BamHeader& hdr, const std::vector<uint8_t>& main_chrom, unsigned threads, HostBatch& o,
                        std::string& err, int& err_code)
{
    const size_t n = recs.size();
    size_t so = 0, qo = 0, co = 0;
    if (n) { const BamRec& L = recs.back(); so = L.so + (L.l_seq + 1) / 2; qo = L.qo + L.l_seq; co = L.co + L.n_cig; }
    o.flag.resize(n); o.mapq.resize(n); o.lane.resize(n); o.rid.resize(n); o.pos.resize(n); o.tlen.resize(n);
    o.nm.resize(n); o.as.resize(n); o.l_seq.resize(n); o.n_cigar.resize(n);
    { // (batches are recycled: the capacities settle after the first ones)
        const size_t c1 = o.seq.capacity(), c2 = o.qual.capacity();
        o.seq.resize(so); o.qual.resize(qo); o.cigar.resize(co);
        if (o.seq.capacity() != c1) advise_huge(o.seq);
        if (o.qual.capacity() != c2) advise_huge(o.qual);
    }
    const unsigned nt_max = std::max(1u, threads);
    std::vector<RecErr> errs(nt_max);
    std::vector<std::vector<std::pair<uint32_t, int32_t>>> extra(nt_max);          // further NM tags: (read, value)
    std::vector<std::vector<std::pair<uint32_t, std::string>>> unknown_rg(nt_max); // reads whose RG id is not in the header
    const auto& lane_names = hdr.lane_names;
    parallel_ranges(n, nt_max, 4096, [&](unsigned t, size_t lo, size_t hi) {
        std::string last_id;
        int last_lane = -1;
        bool have_last = false;
        for (size_t i = lo; i < hi; ++i) {
            const BamRec& R = recs[i];
            const uint8_t* r = base + R.off + 4;
            const int32_t rid = (int32_t)rd32(r), pos = (int32_t)rd32(r + 4);
            const uint32_t l_name = r[8], mapq = r[9];
            const uint32_t n_cig = R.n_cig, flag = rd16(r + 14), l_seq = R.l_seq;
            const int32_t rnext = (int32_t)rd32(r + 20), tlen = (int32_t)rd32(r + 28);
            const uint8_t* cig = r + 32 + l_name;
            const uint8_t* sq = cig + 4ull * n_cig;
            const uint8_t* ql = sq + (l_seq + 1) / 2;
            const uint8_t* tg = ql + l_seq;
            const uint8_t* te = r + R.bs;
            // one linear tag scan: RG (getLane, bamqualcheck.cpp:72-100), every integer NM (QualityCheck.hpp:201-209),
            // first AS (TripletCounting.hpp:113-127)
            int lane = -1;
            bool rg_seen = false, rg_bad = false, as_seen = false, nm_seen = false, bad_tags = false;
            int32_t nm = BQC_NM_ABSENT, as = BQC_AS_ABSENT;
            while (tg + 3 <= te) {
                const char k0 = (char)tg[0], k1 = (char)tg[1], ty = (char)tg[2];
                const uint8_t* v = tg + 3;
                size_t len = 0;
                switch (ty) {
                case 'A': case 'c': case 'C': len = 1; break;
                case 's': case 'S': len = 2; break;
                case 'i': case 'I': case 'f': len = 4; break;
                case 'Z': case 'H': { const void* z = memchr(v, 0, (size_t)(te - v)); len = z ? (size_t)((const uint8_t*)z - v) + 1 : (size_t)(te - v); break; }
                case 'B': {
                    if (v + 5 > te) { len = (size_t)(te - v); break; }
                    const char st = (char)v[0];
                    const size_t cnt = rd32(v + 1);
                    const size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
                    len = 5 + cnt * es;
                    break;
                }
                default: len = (size_t)(te - v); break;
                }
                if (v + len > te) { bad_tags = true; break; }
                if (k0 == 'R' && k1 == 'G' && !rg_seen) {
                    rg_seen = true;
                    if (ty == 'Z') {
                        const size_t idl = len ? len - 1 : 0;
                        if (have_last && last_id.size() == idl && memcmp(last_id.data(), v, idl) == 0) lane = last_lane;
                        else {
                            std::string id((const char*)v, idl);
                            auto it = lane_names.find(id);
                            if (it == lane_names.end()) { unknown_rg[t].emplace_back((uint32_t)i, id); lane = 0; } // operator[] would insert 0 (:86)
                            else { lane = (int)it->second; last_id.swap(id); last_lane = lane; have_last = true; }
                        }
                    } else rg_bad = true;
                } else if (k0 == 'N' && k1 == 'M' && (ty == 'c' || ty == 'C' || ty == 's' || ty == 'S' || ty == 'i' || ty == 'I')) {
                    uint32_t x = 0;
                    switch (ty) {
                    case 'c': x = (uint32_t)(int32_t)(int8_t)v[0]; break;
                    case 'C': x = v[0]; break;
                    case 's': x = (uint32_t)(int32_t)(int16_t)rd16(v); break;
                    case 'S': x = rd16(v); break;
                    default: x = rd32(v); break;
                    }
                    if (!nm_seen) { nm = (int32_t)x; nm_seen = true; }
                    else extra[t].emplace_back((uint32_t)i, (int32_t)x);
                } else if (k0 == 'A' && k1 == 'S' && !as_seen) {
                    as_seen = true;
                    switch (ty) {
                    case 'A': as = (int32_t)(char)v[0]; break;
                    case 'c': as = (int8_t)v[0]; break;
                    case 'C': as = v[0]; break;
                    case 's': as = (int16_t)rd16(v); break;
                    case 'S': as = rd16(v); break;
                    case 'i': case 'I': as = (int32_t)rd32(v); break;
                    case 'f': { float f; uint32_t u = rd32(v); memcpy(&f, &u, 4); as = (int32_t)f; break; }
                    default: as = BQC_AS_ABSENT; break; // extractTagValue fails -> "Could not read AS tag"
                    }
                }
                tg = v + len;
            }
            RecErr& E = errs[t];
            if (E.index == SIZE_MAX) { // the first failing record of this range, by the reference's order of checks
                if (bad_tags) { E.index = i; E.msg = "corrupt BAM tags"; E.code = BQC_ERR_IO; }
                else if (rg_bad) { E.index = i; E.msg = "Read does not have Z"; E.code = BQC_ERR_ARG; }
                else if (!rg_seen) { // DEFINED: the reference falls off the end of getLane() (undefined behaviour)
                    E.index = i; E.msg = "ERROR: read without RG tag (record " + std::to_string(R.nrec) + ")"; E.code = BQC_ERR_ARG;
                } else if ((unsigned)lane >= hdr.lane_count) { E.index = i; E.msg = "ERROR: read group index out of range (no @RG lines in the header?)"; E.code = BQC_ERR_ARG; }
                else if (nm_seen && nm == BQC_NM_ABSENT) { E.index = i; E.msg = "NM tag value 0xFFFFFFFF is not representable"; E.code = BQC_ERR_RANGE; }
            }
            uint32_t f = flag & 0x0FFFu;
            if (rnext >= 0 && (size_t)rnext < main_chrom.size() && main_chrom[rnext]) f |= BQC_FLAG_MATE_MAIN;
            if (l_seq > 0 && ql[0] == 0xFF) f |= BQC_FLAG_NO_QUAL;
            o.flag[i] = (uint16_t)f; o.mapq[i] = (uint8_t)mapq; o.lane[i] = (uint8_t)(lane < 0 ? 0 : lane); o.rid[i] = rid;
            o.pos[i] = pos; o.tlen[i] = tlen; o.nm[i] = nm; o.as[i] = as; o.l_seq[i] = l_seq; o.n_cigar[i] = (uint16_t)n_cig;
            if (n_cig) memcpy(o.cigar.data() + R.co, cig, 4ull * n_cig);
            memcpy(o.seq.data() + R.so, sq, (l_seq + 1) / 2);
            memcpy(o.qual.data() + R.qo, ql, l_seq);
        }
    });
    // unknown read groups, in record order: the first use inserts the id with lane 0
    for (auto& u : unknown_rg)
        for (auto& kv : u)
            if (hdr.lane_names.find(kv.second) == hdr.lane_names.end()) hdr.lane_names[kv.second] = 0;
    for (auto& x : extra)
        for (auto& kv : x) { o.nm_extra_read.push_back(kv.first); o.nm_extra_val.push_back(kv.second); }
    // the first error in record order wins
    const RecErr* first = nullptr;
    for (auto& E : errs)
        if (E.index != SIZE_MAX && (!first || E.index < first->index)) first = &E;
    if (first) { err = first->msg; err_code = first->code; return false; }
    return true;
}
// ---------------------------------------------------------------------------------------------------
// SAM text
// ---------------------------------------------------------------------------------------------------
bool SamReader::getline(std::string& line)
{
    line.clear();
    if (eof_) return false;
    char buf[1 << 16];
    for (;;) {
        if (!fgets(buf, sizeof buf, f_)) { eof_ = true; return !line.empty(); }
        const size_t n = strlen(buf);
        line.append(buf, n);
        if (n && buf[n - 1] == '\n') { line.pop_back(); if (!line.empty() && line.back() == '\r') line.pop_back(); return true; }
    }
}

bool SamReader::open(FILE* f, std::string& err)
{
    f_ = f;
    if (!f_) { err = "no input stream"; return false; }
    std::string line;
    while (getline(line)) {
        if (line.empty()) continue;
        if (line[0] != '@') { pending_ = line; have_pending_ = true; break; }
        hdr_.text += line; hdr_.text += '\n';
        if (line.compare(0, 3, "@SQ") == 0) {
            std::string name;
            uint32_t len = 0;
            size_t p = 3;
            while (p < line.size()) {
                size_t e = line.find('\t', p + 1);
                if (e == std::string::npos) e = line.size();
                const std::string fld = line.substr(p + 1, e - p - 1);
                if (fld.compare(0, 3, "SN:") == 0) name = fld.substr(3);
                if (fld.compare(0, 3, "LN:") == 0) len = (uint32_t)strtoul(fld.c_str() + 3, nullptr, 10);
                p = e;
            }
            ref_index_[name] = (int32_t)hdr_.ref_names.size();
            hdr_.ref_names.push_back(name);
            hdr_.ref_lens.push_back(len);
        }
    }
    parse_read_groups(hdr_);
    return true;
}

int SamReader::next_batch(HostBatch& o, size_t max_reads, size_t max_bases, std::string& err, int& err_code)
{
    static const char kOps[] = "MIDNSHP=X";
    static const char kNib[] = "=ACMGRSVTWYHKDBN";
    o.clear();
    err_code = 0;
    size_t bases = 0;
    std::string line;
    std::vector<std::string> f;
    auto fail = [&](const std::string& m, int code) { err = m; err_code = code; return -1; };
    while (o.n() < max_reads && bases < max_bases) {
        if (have_pending_) { line.swap(pending_); have_pending_ = false; }
        else if (!getline(line)) break;
        if (line.empty() || line[0] == '@') continue;
        f.clear();
        for (size_t p = 0;;) {
            const size_t e = line.find('\t', p);
            f.push_back(line.substr(p, e == std::string::npos ? std::string::npos : e - p));
            if (e == std::string::npos) break;
            p = e + 1;
        }
        if (f.size() < 11) return fail("corrupt SAM record (fewer than 11 fields)", BQC_ERR_IO);
        auto ref_of = [&](const std::string& name) -> int32_t {
            if (name == "*") return -1;
            auto it = ref_index_.find(name);
            return it == ref_index_.end() ? -1 : it->second;
        };
        const uint32_t flag = (uint32_t)strtoul(f[1].c_str(), nullptr, 10);
        const int32_t rid = ref_of(f[2]);
        const int32_t pos = (int32_t)strtol(f[3].c_str(), nullptr, 10) - 1;
        const uint32_t mapq = (uint32_t)strtoul(f[4].c_str(), nullptr, 10);
        const int32_t rnext = f[6] == "=" ? rid : ref_of(f[6]);
        const int32_t tlen = (int32_t)strtol(f[8].c_str(), nullptr, 10);
        const size_t c0 = o.cigar.size();
        uint32_t n_cig = 0;
        if (f[5] != "*") {
            const char* p = f[5].c_str();
            while (*p) {
                char* e = nullptr;
                const unsigned long n = strtoul(p, &e, 10);
                const char* opc = e && *e ? strchr(kOps, *e) : nullptr;
                if (e == p || !opc || n >= (1ul << 28)) return fail("corrupt SAM record (CIGAR)", BQC_ERR_IO);
                o.cigar.push_back((uint32_t)(n << 4) | (uint32_t)(opc - kOps));
                ++n_cig;
                p = e + 1;
            }
            if (n_cig > 65535) return fail("corrupt SAM record (more than 65535 CIGAR operations)", BQC_ERR_IO);
        }
        const std::string& sq = f[9];
        const std::string& ql = f[10];
        const uint32_t l_seq = sq == "*" ? 0u : (uint32_t)sq.size();
        if (ql != "*" && ql.size() != l_seq) return fail("corrupt SAM record (SEQ and QUAL differ in length)", BQC_ERR_IO);
        const size_t s0 = o.seq.size(), q0 = o.qual.size();
        o.seq.resize(s0 + (l_seq + 1) / 2);
        o.qual.resize(q0 + l_seq);
        for (uint32_t i = 0; i < l_seq; i += 2) {
            auto code = [&](char ch) -> uint32_t { const char* z = strchr(kNib, toupper((unsigned char)ch)); return z && ch ? (uint32_t)(z - kNib) : 15u; };
            o.seq[s0 + i / 2] = (uint8_t)((code(sq[i]) << 4) | (i + 1 < l_seq ? code(sq[i + 1]) : 0u));
        }
        for (uint32_t i = 0; i < l_seq; ++i) o.qual[q0 + i] = ql == "*" ? (uint8_t)0xFF : (uint8_t)(ql[i] - 33);
        // tags: RG (getLane, bamqualcheck.cpp:72-100), every integer NM (QualityCheck.hpp:201-209), first AS (TripletCounting.hpp:113-127)
        int lane = -1;
        bool rg_seen = false, rg_bad = false, as_seen = false, nm_seen = false;
        int32_t nm = BQC_NM_ABSENT, as = BQC_AS_ABSENT;
        const uint32_t idx = (uint32_t)o.n();
        for (size_t t = 11; t < f.size(); ++t) {
            const std::string& tg = f[t];
            if (tg.size() < 5 || tg[2] != ':' || tg[4] != ':') continue;
            const char k0 = tg[0], k1 = tg[1], ty = tg[3];
            const char* v = tg.c_str() + 5;
            if (k0 == 'R' && k1 == 'G' && !rg_seen) {
                rg_seen = true;
                if (ty == 'Z') {
                    const std::string id(v);
                    auto it = hdr_.lane_names.find(id);
                    if (it == hdr_.lane_names.end()) { hdr_.lane_names[id] = 0; lane = 0; } // std::map::operator[] inserts 0 (:86)
                    else lane = (int)it->second;
                } else rg_bad = true;
            } else if (k0 == 'N' && k1 == 'M' && ty == 'i') {
                const uint32_t x = (uint32_t)strtoll(v, nullptr, 10);
                if (!nm_seen) { nm = (int32_t)x; nm_seen = true; }
                else { o.nm_extra_read.push_back(idx); o.nm_extra_val.push_back((int32_t)x); }
            } else if (k0 == 'A' && k1 == 'S' && !as_seen) {
                as_seen = true;
                if (ty == 'i') as = (int32_t)strtoll(v, nullptr, 10);
                else if (ty == 'A') as = (int32_t)v[0];
                else if (ty == 'f') as = (int32_t)strtof(v, nullptr);
                else as = BQC_AS_ABSENT;
            }
        }
        if (rg_bad) return fail("Read does not have Z", BQC_ERR_ARG);
        if (!rg_seen) return fail("ERROR: read without RG tag (record " + std::to_string(nrec_) + ")", BQC_ERR_ARG);
        if ((unsigned)lane >= hdr_.lane_count) return fail("ERROR: read group index out of range (no @RG lines in the header?)", BQC_ERR_ARG);
        if (nm_seen && nm == BQC_NM_ABSENT) return fail("NM tag value 0xFFFFFFFF is not representable", BQC_ERR_RANGE);
        uint32_t fl = flag & 0x0FFFu;
        if (rnext >= 0 && (size_t)rnext < main_.size() && main_[rnext]) fl |= BQC_FLAG_MATE_MAIN;
        if (l_seq > 0 && ql == "*") fl |= BQC_FLAG_NO_QUAL;
        o.flag.push_back((uint16_t)fl); o.mapq.push_back((uint8_t)mapq); o.lane.push_back((uint8_t)lane); o.rid.push_back(rid);
        o.pos.push_back(pos); o.tlen.push_back(tlen); o.nm.push_back(nm); o.as.push_back(as); o.l_seq.push_back(l_seq);
        o.n_cigar.push_back((uint16_t)n_cig);
        (void)c0;
        bases += l_seq;
        ++nrec_;
    }
    return o.n() ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------
// writer
// ---------------------------------------------------------------------------------------------------
static void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x & 0xFF); v.push_back((x >> 8) & 0xFF); v.push_back((x >> 16) & 0xFF); v.push_back((x >> 24) & 0xFF); }
static void put16(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x & 0xFF); v.push_back((x >> 8) & 0xFF); }

bool BamWriter::open(const char* path, const std::string& text, const std::vector<std::string>& names, const std::vector<uint32_t>& lens,
                     std::string& err, int level)
{
    if (!bg_.open(path, err, level)) return false;
    std::vector<uint8_t> h;
    h.insert(h.end(), {'B', 'A', 'M', 1});
    put32(h, (uint32_t)text.size());
    h.insert(h.end(), text.begin(), text.end());
    put32(h, (uint32_t)names.size());
    for (size_t i = 0; i < names.size(); ++i) {
        put32(h, (uint32_t)names[i].size() + 1);
        h.insert(h.end(), names[i].begin(), names[i].end());
        h.push_back(0);
        put32(h, lens[i]);
    }
    return bg_.write(h.data(), h.size());
}

static uint32_t reg2bin(int64_t beg, int64_t end)
{
    --end;
    if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

bool BamWriter::write_batch(const bqc_batch& b, const std::vector<std::string>& lane_ids, uint64_t first)
{
    uint64_t so = 0, qo = 0, co = 0;
    for (uint32_t i = 0; i < b.n_reads; ++i) {
        rec_.clear();
        char name[32];
        const int nl = snprintf(name, sizeof name, "r%llu", (unsigned long long)(first + i)) + 1;
        const uint32_t L = b.l_seq[i], nc = b.n_cigar[i];
        int64_t reflen = 0;
        for (uint32_t k = 0; k < nc; ++k) { const uint32_t op = b.cigar[co + k] & 15u; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) reflen += b.cigar[co + k] >> 4; }
        put32(rec_, 0); // block size, patched below
        put32(rec_, (uint32_t)b.rid[i]); put32(rec_, (uint32_t)b.pos[i]);
        rec_.push_back((uint8_t)nl); rec_.push_back(b.mapq[i]);
        put16(rec_, b.pos[i] >= 0 ? reg2bin(b.pos[i], b.pos[i] + (reflen ? reflen : 1)) : 4680);
        put16(rec_, nc); put16(rec_, b.flag[i] & 0x0FFFu);
        put32(rec_, L); put32(rec_, (uint32_t)b.rid[i]); put32(rec_, (uint32_t)b.pos[i]); put32(rec_, (uint32_t)b.tlen[i]);
        rec_.insert(rec_.end(), name, name + nl);
        for (uint32_t k = 0; k < nc; ++k) put32(rec_, b.cigar[co + k]);
        rec_.insert(rec_.end(), b.seq + so, b.seq + so + (L + 1) / 2);
        rec_.insert(rec_.end(), b.qual + qo, b.qual + qo + L);
        const std::string& rg = lane_ids[b.lane[i]];
        rec_.insert(rec_.end(), {'R', 'G', 'Z'});
        rec_.insert(rec_.end(), rg.begin(), rg.end());
        rec_.push_back(0);
        if (b.nm[i] != BQC_NM_ABSENT) { rec_.insert(rec_.end(), {'N', 'M', 'i'}); put32(rec_, (uint32_t)b.nm[i]); }
        if (b.as[i] != BQC_AS_ABSENT) { rec_.insert(rec_.end(), {'A', 'S', 'i'}); put32(rec_, (uint32_t)b.as[i]); }
        const uint32_t bs = (uint32_t)rec_.size() - 4;
        rec_[0] = bs & 0xFF; rec_[1] = (bs >> 8) & 0xFF; rec_[2] = (bs >> 16) & 0xFF; rec_[3] = (bs >> 24) & 0xFF;
        if (!bg_.write(rec_.data(), rec_.size())) return false;
        so += (L + 1) / 2; qo += L; co += nc;
    }
    return true;
}

bool BamWriter::write_batch_parallel(const bqc_batch& b, const std::vector<std::string>& lane_ids, uint64_t first)
{
    const size_t n = b.n_reads;
    // pass 1 (serial, a few ns per read): where every record and its payload start
    at_.resize(3 * (n + 1));
    uint64_t* rec_at = at_.data();
    uint64_t* so_at = rec_at + n + 1;
    uint64_t* co_at = so_at + n + 1;
    auto digits = [](uint64_t v) { int d = 1; while (v >= 10) { v /= 10; ++d; } return d; };
    uint64_t at = 0, so = 0, qo = 0, co = 0;
    std::vector<uint64_t> qo_at(n + 1);
    for (size_t i = 0; i < n; ++i) {
        rec_at[i] = at; so_at[i] = so; qo_at[i] = qo; co_at[i] = co;
        const uint32_t L = b.l_seq[i], nc = b.n_cigar[i];
        at += 4 + 32 + (1 + digits(first + i) + 1) + 4ull * nc + (L + 1) / 2 + L + (3 + lane_ids[b.lane[i]].size() + 1) +
              (b.nm[i] != BQC_NM_ABSENT ? 7 : 0) + (b.as[i] != BQC_AS_ABSENT ? 7 : 0);
        so += (L + 1) / 2; qo += L; co += nc;
    }
    rec_at[n] = at;
    big_.resize(at);
    parallel_ranges(n, bqc_host_threads(), 4096, [&](unsigned, size_t lo, size_t hi) {
        auto w32 = [](uint8_t* p, uint32_t x) { p[0] = x & 0xFF; p[1] = (x >> 8) & 0xFF; p[2] = (x >> 16) & 0xFF; p[3] = (x >> 24) & 0xFF; };
        auto w16 = [](uint8_t* p, uint32_t x) { p[0] = x & 0xFF; p[1] = (x >> 8) & 0xFF; };
        for (size_t i = lo; i < hi; ++i) {
            uint8_t* r = big_.data() + rec_at[i];
            const uint32_t L = b.l_seq[i], nc = b.n_cigar[i];
            const uint32_t* cg = b.cigar + co_at[i];
            int64_t reflen = 0;
            for (uint32_t k = 0; k < nc; ++k) { const uint32_t op = cg[k] & 15u; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) reflen += cg[k] >> 4; }
            char name[32];
            const int nl = snprintf(name, sizeof name, "r%llu", (unsigned long long)(first + i)) + 1;
            w32(r, (uint32_t)(rec_at[i + 1] - rec_at[i] - 4));
            w32(r + 4, (uint32_t)b.rid[i]); w32(r + 8, (uint32_t)b.pos[i]);
            r[12] = (uint8_t)nl; r[13] = b.mapq[i];
            w16(r + 14, b.pos[i] >= 0 ? reg2bin(b.pos[i], b.pos[i] + (reflen ? reflen : 1)) : 4680);
            w16(r + 16, nc); w16(r + 18, b.flag[i] & 0x0FFFu);
            w32(r + 20, L); w32(r + 24, (uint32_t)b.rid[i]); w32(r + 28, (uint32_t)b.pos[i]); w32(r + 32, (uint32_t)b.tlen[i]);
            uint8_t* q = r + 36;
            memcpy(q, name, (size_t)nl); q += nl;
            for (uint32_t k = 0; k < nc; ++k, q += 4) w32(q, cg[k]);
            memcpy(q, b.seq + so_at[i], (L + 1) / 2); q += (L + 1) / 2;
            memcpy(q, b.qual + qo_at[i], L); q += L;
            const std::string& rg = lane_ids[b.lane[i]];
            *q++ = 'R'; *q++ = 'G'; *q++ = 'Z';
            memcpy(q, rg.data(), rg.size()); q += rg.size();
            *q++ = 0;
            if (b.nm[i] != BQC_NM_ABSENT) { *q++ = 'N'; *q++ = 'M'; *q++ = 'i'; w32(q, (uint32_t)b.nm[i]); q += 4; }
            if (b.as[i] != BQC_AS_ABSENT) { *q++ = 'A'; *q++ = 'S'; *q++ = 'i'; w32(q, (uint32_t)b.as[i]); q += 4; }
        }
    });
    return bg_.write(big_.data(), big_.size());
}
