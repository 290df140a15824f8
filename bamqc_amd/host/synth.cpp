// synth.cpp — seeded synthetic input generator (SURVEY.md §8d "configs as concrete synthetic inputs").
//
// Counter-based: every read is derived from (seed, read index) alone, so generation is parallel and
// the output does not depend on the number of threads.  Reads come out coordinate-sorted by
// construction (read j of a contig starts in the j-th of n equal slots).  Used by bench.py, the
// large-size parity tests and the `bamqc_synth` tool; it is part of the host tooling around the
// hot path, not of the hot path itself.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/bamqc.h"
#include "host_tools.h"

namespace {
struct Rng { // xoshiro256** seeded by splitmix64
    uint64_t s[4];
    static uint64_t splitmix(uint64_t& x)
    {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed) { for (auto& v : s) v = splitmix(seed); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next()
    {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
    double normal()
    {
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

const uint8_t NIB_OF_CODE[5] = {1, 2, 4, 8, 15};

struct CigarBuf { // a read's CIGAR without a heap allocation per read (long reads: at most 29 events -> 61 operations)
    uint32_t n = 0;
    uint32_t v[64];
    bool empty() const { return n == 0; }
    size_t size() const { return n; }
    void clear() { n = 0; }
    uint32_t& back() { return v[n - 1]; }
    uint32_t front() const { return v[0]; }
    uint32_t back() const { return v[n - 1]; }
    void push_back(uint32_t x) { if (n < 64) v[n++] = x; }
    const uint32_t* begin() const { return v; }
    const uint32_t* end() const { return v + n; }
    const uint32_t* data() const { return v; }
};
struct ReadPlan { // everything about one read except its bases
    uint32_t flag = 0, L = 0;
    int32_t rid = -1, pos = -1, tlen = 0, nm = BQC_NM_ABSENT, as = BQC_AS_ABSENT;
    uint8_t mapq = 0, lane = 0;
    CigarBuf cigar;
};
} // namespace

extern "C" int bqc_synth_reference(uint64_t seed, int32_t rid, uint64_t len, uint8_t* out)
{
    if (!out) return BQC_ERR_ARG;
    const uint64_t BLK = 1 << 20;
    const uint64_t nblk = (len + BLK - 1) / BLK;
    unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([=]() {
            for (uint64_t b = t; b < nblk; b += nt) {
                Rng r(seed * 0x100000001B3ull + (uint64_t)(rid + 1) * 0x9E3779B97F4A7C15ull + b);
                const uint64_t lo = b * BLK, hi = std::min(len, lo + BLK);
                for (uint64_t i = lo; i < hi; i += 32) {
                    uint64_t w = r.next();
                    for (uint64_t k = i; k < std::min(hi, i + 32); ++k) { out[k] = w & 3; w >>= 2; }
                }
                // 0.1 % of the bases in N runs of 1..200
                uint64_t n_runs = (hi - lo) / 100000 + 1;
                for (uint64_t k = 0; k < n_runs; ++k) {
                    uint64_t s = lo + r.below((uint32_t)(hi - lo)), e = std::min(hi, s + 1 + r.below(200));
                    for (uint64_t i = s; i < e; ++i) out[i] = 4;
                }
            }
        });
    for (auto& t : th) t.join();
    return 0;
}

static void plan_read(const bqc_synth_params& p, uint64_t gi, uint32_t rid, uint32_t pos, Rng& r, ReadPlan& o)
{
    const uint32_t L = p.read_len;
    o.L = L;
    o.lane = p.n_lanes > 1 ? (uint8_t)r.below(p.n_lanes) : 0;
    uint32_t f = 0x1 | ((gi & 1) ? 0x80 : 0x40);
    if (r.next() & 1) f |= 0x10;
    if (r.next() & 1) f |= 0x20;
    bool unmapped = false;
    if (!p.long_reads) {
        const double u = r.uni();
        if (u < 0.01) { unmapped = true; f |= 0x4; }           // one mate unmapped (carries the mate's rid/pos)
        else if (u < 0.013) { unmapped = true; f |= 0x4 | 0x8; } // both unmapped
        else if (u < 0.023) f |= 0x8;                          // mate unmapped
        if (!unmapped && !(f & 0x8) && r.uni() < 0.96 / 0.977) f |= 0x2;
        if (r.uni() < 0.01) f |= 0x400;
        if (r.uni() < 0.005) f |= 0x200;
        if (r.uni() < 0.002) f |= 0x100;
        if (r.uni() < 0.002) f |= 0x800;
    } else {
        if (r.uni() < 0.9) f |= 0x2;
    }
    f |= BQC_FLAG_MATE_MAIN;
    o.flag = f;
    o.rid = (int32_t)rid;
    o.pos = (int32_t)pos;
    if ((f & 0xC) == 0xC) { o.rid = -1; o.pos = -1; }
    const double mq = r.uni();
    o.mapq = unmapped ? 0 : (mq < 0.9 ? 60 : (mq < 0.93 ? 0 : (uint8_t)(1 + r.below(59))));
    double tl = 400.0 + 80.0 * r.normal();
    const double lim = (double)p.isize + 200.0;
    if (tl > lim) tl = lim;
    if (tl < 0) tl = 0;
    o.tlen = (int32_t)tl * ((r.next() & 1) ? 1 : -1);
    o.cigar.clear();
    o.nm = BQC_NM_ABSENT;
    o.as = BQC_AS_ABSENT;
    if (unmapped) return;
    auto push = [&](uint32_t n, uint32_t op) {
        if (!n) return;
        if (!o.cigar.empty() && (o.cigar.back() & 15u) == op) o.cigar.back() += n << 4;
        else o.cigar.push_back((n << 4) | op);
    };
    uint32_t lead = 0, trail = 0, indel = 0;
    if (!p.long_reads) {
        if (r.uni() < 0.03) {
            const uint32_t w = r.below(3);
            if (w == 0 || w == 2) lead = 1 + r.below(30);
            if (w == 1 || w == 2) trail = 1 + r.below(30);
        }
        uint32_t body = L - lead - trail;
        push(lead, 4);
        if (r.uni() < 0.02 && body > 40) {
            const uint32_t k = 1 + r.below(3), at = 10 + r.below(body - 30);
            push(at, 0);
            if (r.next() & 1) { push(k, 1); push(body - at - k, 0); } // insertion
            else { push(k, 2); push(body - at, 0); }                 // deletion
            indel = k;
        } else {
            push(body, 0);
        }
        push(trail, 4);
    } else { // 20-60 ops, indel heavy, 10 % clipped >= 500 bp
        if (r.uni() < 0.10) { if (r.next() & 1) lead = 500 + r.below(1500); else trail = 500 + r.below(1500); }
        uint32_t body = L - lead - trail;
        push(lead, 4);
        const uint32_t n_ev = 10 + r.below(20);
        for (uint32_t e = 0; e < n_ev && body > 200; ++e) {
            const uint32_t m = 20 + r.below(2 * body / (n_ev - e + 1));
            const uint32_t mm = std::min(m, body - 100);
            push(mm, 0);
            body -= mm;
            const uint32_t k = 1 + r.below(8);
            if (r.next() & 1) { if (body > k + 50) { push(k, 1); body -= k; indel += k; } }
            else { push(k, 2); indel += k; }
        }
        push(body, 0);
        push(trail, 4);
    }
    o.nm = (int32_t)indel; // substitutions are added by the caller
}

// Fill one read's bases/qualities from the plan and the reference; returns the number of substitutions.
// `code` is scratch of L bytes; `qthr` the quality thresholds a | b << 8 | c << 16 per cycle (qual_thresholds).
static uint32_t fill_read(const bqc_synth_params& p, const ReadPlan& o, const uint8_t* ref, uint64_t reflen, Rng& r, uint8_t* seq, uint8_t* qual,
                          uint8_t* code, const uint32_t* qthr)
{
    const uint32_t L = o.L;
    uint32_t rp = 0;
    uint64_t cp = o.pos < 0 ? 0 : (uint64_t)o.pos;
    const bool mapped = !(o.flag & 0x4) && ref;
    if (!mapped || o.cigar.empty()) {
        for (uint32_t i = 0; i < L; i += 32) {
            uint64_t w = r.next();
            for (uint32_t k = i; k < std::min(L, i + 32); ++k) { code[k] = w & 3; w >>= 2; }
        }
    } else {
        for (uint32_t cw : o.cigar) {
            const uint32_t op = cw & 15u, n = cw >> 4;
            if (op == 0) {
                const uint32_t m = std::min(n, L - rp);
                if (cp + m <= reflen) { memcpy(code + rp, ref + cp, m); rp += m; cp += m; }
                else for (uint32_t k = 0; k < m; ++k, ++rp, ++cp) code[rp] = cp < reflen ? ref[cp] : 0;
            }
            else if (op == 1 || op == 4) { for (uint32_t k = 0; k < n && rp < L; ++k, ++rp) code[rp] = (uint8_t)r.below(4); }
            else if (op == 2) cp += n;
        }
    }
    static const double kLogSub = std::log(1.0 - 0.005), kLogN = std::log(1.0 - 0.001);
    uint32_t subs = 0;
    if (mapped && !o.cigar.empty()) { // substitutions at 0.5 % on aligned bases via geometric gaps
        uint32_t lead = (o.cigar.front() & 15u) == 4 ? o.cigar.front() >> 4 : 0;
        uint32_t trail = (o.cigar.back() & 15u) == 4 && o.cigar.size() > 1 ? o.cigar.back() >> 4 : 0;
        double i = lead;
        for (;;) {
            i += std::floor(std::log(1.0 - r.uni()) / kLogSub) + 1;
            if (i >= (double)(L - trail)) break;
            const uint32_t k = (uint32_t)i - 1;
            if (code[k] < 4) { code[k] = (code[k] + 1 + r.below(3)) & 3; ++subs; }
        }
    }
    { // read N at 0.1 %
        double i = 0;
        for (;;) {
            i += std::floor(std::log(1.0 - r.uni()) / kLogN) + 1;
            if (i > (double)L) break;
            code[(uint32_t)i - 1] = 4;
        }
    }
    for (uint32_t i = 0; i < L; i += 2) {
        const uint8_t hi = NIB_OF_CODE[code[i]], lo = (i + 1 < L) ? NIB_OF_CODE[code[i + 1]] : 0;
        seq[i >> 1] = (uint8_t)((hi << 4) | lo);
    }
    // qualities from {2,12,23,37} with cycle dependent weights (worse towards the end of the read)
    static const uint8_t QL[4] = {2, 12, 23, 37};
    const bool rc = o.flag & 0x10;
    for (uint32_t i = 0; i < L; i += 8) {
        uint64_t w = r.next();
        for (uint32_t k = i; k < std::min(L, i + 8); ++k, w >>= 8) {
            const uint32_t th = qthr[rc ? L - 1 - k : k], u = w & 255;
            qual[k] = QL[(u >= (th & 255u)) + (u >= ((th >> 8) & 255u)) + (u >= (th >> 16))];
        }
    }
    return subs;
}

// thresholds of the quality draw per cycle: P(q=37) from 0.80 down to 0.55 along the read
static std::vector<uint32_t> qual_thresholds(uint32_t L)
{
    std::vector<uint32_t> v(L);
    for (uint32_t cyc = 0; cyc < L; ++cyc) {
        const uint32_t t = (uint32_t)(((uint64_t)cyc << 8) / L); // 0..255
        const uint32_t a = 5 + t / 10, b = a + 10 + t / 10, c = b + 36 + t / 24;
        v[cyc] = a | (b << 8) | (c << 16);
    }
    return v;
}

extern "C" int bqc_synth_batch(const bqc_synth_params* pp, const uint8_t* const* refs, bqc_batch** out)
{
    return pp ? bqc_synth_slice(pp, 0, pp->n_reads, refs, out) : BQC_ERR_ARG;
}

// Reads [lo, lo + count) of the plan of p.n_reads reads: every read depends on (seed, global index, n_reads) only, so a slice
// holds exactly the reads the whole plan holds at those indices.
extern "C" int bqc_synth_slice(const bqc_synth_params* pp, uint64_t lo, uint32_t count, const uint8_t* const* refs, bqc_batch** out)
{
    if (!pp || !out || pp->n_refs == 0 || !pp->ref_len || pp->read_len < 8 || lo + count > pp->n_reads) return BQC_ERR_ARG;
    const bqc_synth_params p = *pp;
    const uint32_t n_all = p.n_reads, n = count, L = p.read_len;
    // reads per contig proportional to length
    long double total = 0;
    for (uint32_t c = 0; c < p.n_refs; ++c) total += p.ref_len[c];
    std::vector<uint64_t> start(p.n_refs + 1, 0);
    for (uint32_t c = 0; c < p.n_refs; ++c) {
        long double acc = 0;
        for (uint32_t k = 0; k <= c; ++k) acc += p.ref_len[k];
        start[c + 1] = (uint64_t)((long double)n_all * acc / total);
    }
    start[p.n_refs] = n_all;
    // pass 1: plans (cigars are variable length)
    std::vector<ReadPlan> plans(n);
    const unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    auto contig_of = [&](uint64_t gi) { return (uint32_t)(std::upper_bound(start.begin(), start.end(), gi) - start.begin() - 1); };
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t]() {
                for (uint64_t k = (uint64_t)n * t / nt; k < (uint64_t)n * (t + 1) / nt; ++k) {
                    const uint64_t gi = lo + k; // index in the plan
                    Rng r(p.seed * 0xD6E8FEB86659FD93ull + (p.first_read_index + gi) * 0x9E3779B97F4A7C15ull + 1);
                    const uint32_t c = std::min(contig_of(gi), p.n_refs - 1);
                    const uint64_t nc = start[c + 1] - start[c], j = gi - start[c];
                    const uint64_t margin = (uint64_t)L * 2 + 4096;
                    const uint64_t span = p.ref_len[c] > margin ? p.ref_len[c] - margin : 1;
                    const double slot = (double)span / (double)std::max<uint64_t>(nc, 1);
                    const uint32_t pos = (uint32_t)(((double)j + r.uni()) * slot);
                    plan_read(p, p.first_read_index + gi, c, pos, r, plans[k]);
                }
            });
        for (auto& t : th) t.join();
    }
    const std::vector<uint32_t> qthr = qual_thresholds(L);
    std::vector<uint64_t> coff(n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) coff[i + 1] = coff[i] + plans[i].cigar.size();
    const uint64_t sbytes = (uint64_t)n * ((L + 1) / 2), qbytes = (uint64_t)n * L;
    auto* hb = (bqc_synth_owned*)calloc(1, sizeof(bqc_synth_owned));
    hb->flag = (uint16_t*)malloc(2ull * n + 8); hb->mapq = (uint8_t*)malloc(n + 8); hb->lane = (uint8_t*)malloc(n + 8);
    hb->rid = (int32_t*)malloc(4ull * n + 8); hb->pos = (int32_t*)malloc(4ull * n + 8); hb->tlen = (int32_t*)malloc(4ull * n + 8);
    hb->nm = (int32_t*)malloc(4ull * n + 8); hb->as = (int32_t*)malloc(4ull * n + 8); hb->l_seq = (uint32_t*)malloc(4ull * n + 8);
    hb->n_cigar = (uint16_t*)malloc(2ull * n + 8); hb->seq = (uint8_t*)malloc(sbytes + 8); hb->qual = (uint8_t*)malloc(qbytes + 8);
    hb->cigar = (uint32_t*)malloc(4 * coff[n] + 8);
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t]() {
                std::vector<uint8_t> code(L);
                for (uint64_t gi = (uint64_t)n * t / nt; gi < (uint64_t)n * (t + 1) / nt; ++gi) { // (gi: index in the slice from here on)
                    ReadPlan& o = plans[gi];
                    Rng r(p.seed * 0xA0761D6478BD642Full + (p.first_read_index + lo + gi) * 0xE7037ED1A0B428DBull + 7);
                    const uint8_t* ref = (o.rid >= 0 && refs) ? refs[o.rid] : nullptr;
                    const uint32_t subs = fill_read(p, o, ref, o.rid >= 0 ? p.ref_len[o.rid] : 0, r, hb->seq + gi * ((L + 1) / 2), hb->qual + gi * L,
                                                    code.data(), qthr.data());
                    if (!(o.flag & 0x4)) {
                        uint32_t indel = (uint32_t)o.nm;
                        o.nm = (int32_t)(indel + subs);
                        int64_t as = (int64_t)L - 5ll * subs - 6ll * indel;
                        o.as = (int32_t)std::max<int64_t>(0, as);
                        if (r.uni() < 0.005) o.nm = BQC_NM_ABSENT;
                    }
                    hb->flag[gi] = (uint16_t)o.flag; hb->mapq[gi] = o.mapq; hb->lane[gi] = o.lane; hb->rid[gi] = o.rid;
                    hb->pos[gi] = o.pos; hb->tlen[gi] = o.tlen; hb->nm[gi] = o.nm; hb->as[gi] = o.as; hb->l_seq[gi] = L;
                    hb->n_cigar[gi] = (uint16_t)o.cigar.size();
                    if (!o.cigar.empty()) memcpy(hb->cigar + coff[gi], o.cigar.data(), 4 * o.cigar.size());
                }
            });
        for (auto& t : th) t.join();
    }
    bqc_batch& b = hb->batch;
    b.n_reads = n;
    b.flag = hb->flag; b.mapq = hb->mapq; b.lane = hb->lane; b.rid = hb->rid; b.pos = hb->pos; b.tlen = hb->tlen; b.nm = hb->nm;
    b.as = hb->as; b.l_seq = hb->l_seq; b.n_cigar = hb->n_cigar; b.seq = hb->seq; b.qual = hb->qual; b.cigar = hb->cigar;
    b.n_nm_extra = 0; b.nm_extra_read = nullptr; b.nm_extra_val = nullptr;
    *out = &hb->batch;
    return 0;
}

extern "C" void bqc_synth_batch_free(bqc_batch* b)
{
    if (!b) return;
    auto* hb = (bqc_synth_owned*)b; // batch is the first member
    free(hb->flag); free(hb->mapq); free(hb->lane); free(hb->rid); free(hb->pos); free(hb->tlen); free(hb->nm); free(hb->as);
    free(hb->l_seq); free(hb->n_cigar); free(hb->seq); free(hb->qual); free(hb->cigar);
    free(hb);
}
