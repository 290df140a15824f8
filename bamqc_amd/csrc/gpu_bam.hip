// gpu_bam.hip — BAM input decoded on the GPU (row N2 of the scope table, step 2: see gpu_bam.h).
//
//   file --pread, three threads--> ring of page-locked chunks --H2D--> k_inflate + k_gi_crc (gpu_inflate.hip) --> the uncompressed stream, in device memory
//        --> k_gb_walk: where the records are        --> k_gb_decode: the fixed columns + where each record's payload goes
//        --> k_gb_copy: bases / qualities / CIGARs into packed columns (device)   --> D2H of the fixed columns only (26 B per read)
//
// The record chain (every record's start follows from the previous one's block_size) is walked the way the host reader walks it
// in parallel (host/bam_io.cpp: parallel_prewalk): the window is cut into 16 KiB segments, a lane per segment GUESSES the first
// record start in its segment (the first offset at which three records in a row look like records) and walks from there to the
// segment's end; the host then checks, segment by segment, that every guess is exactly where the previous segment's walk
// arrived — by induction from the known first record the chain is then the serial walk's; a segment whose guess is not there
// is walked again, alone, from the known position.  A BATCH that holds a record the host reader has a rule for beyond the plain case
// (a read group that is not in the header, a second NM tag, no RG tag, ...) is decoded by the host reader's own code
// (bam_decode_records) from the bytes on the card; a walk that cannot be verified, a corrupt record or a file that ends inside a
// record make next_batch return kUnsupported: the caller starts over with the host reader, which is the one to decide what the
// user is told.
//
// The file — or one worker's byte range of it (set_range: the multi-GPU program) — is taken in few, large RUNS of BGZF blocks (large
// inflate launches are the cheapest per block: 10 ms for 9 K blocks, 29 ms for 45 K, gpu_inflate.hip): reader threads fill a ring of page-locked chunks, a
// producer thread parses the block headers, copies and launches on one stream (the first run is read while the HIP runtime is
// still starting); next_batch walks and decodes the current run's window on another; what a run leaves over (an unfinished
// record) is copied in front of the next run's bytes.  Every buffer is allocated in open(): allocations, releases and page-locking
// behind a running inflate kernel wait for it.
//
// Record layout, tag types: the public SAM/BAM specification; the rules applied to a record are those of the host reader
// (bam_io.cpp: next_batch, citing bamqualcheck.cpp:72-100 getLane, QualityCheck.hpp:201-209 NM, TripletCounting.hpp:113-127 AS).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "../../include/bamqc.h"
#include "gpu_bam.h"
#include "gpu_inflate.h"
#include "../host/parallel.h"

hipStream_t bqc_pool_stream(int device, int rank); // bqc_api.cpp: a stream bqc_warmup has made ahead (rank: the order of need), or a new one

#define GB_SEG 16384u // (4 KiB segments measured slower: 28 walk launches of 0.20 ms per 10 M reads instead of 18 of 0.27 — more segments whose guess has to be redone — and k_gb_decode 0.18 instead of 0.14 ms per batch)
#define GB_MAXR (GB_SEG / 36u + 1u)

enum { GB_INCOMPLETE = 1, GB_CORRUPT = 2, GB_NO_START = 4 };
// exceptions a record can raise (status word): the host reader handles all of them
enum { GBX_TAGS = 1 << 8, GBX_RG_MISSING = 1 << 9, GBX_RG_TYPE = 1 << 10, GBX_RG_UNKNOWN = 1 << 11, GBX_NM_EXTRA = 1 << 12, GBX_NM_VALUE = 1 << 13, GBX_LANE = 1 << 14 };

struct GbSeg { uint32_t first, exit, count, flags, seq_bytes, qual_bytes, cigar_words, pad; }; // offsets relative to the window
struct GbRec { uint32_t off, so, qo, co; };                                                     // payload prefix inside the segment
struct GbBase { uint64_t so, qo, co; uint32_t rec, take; };                                    // where a taken segment's records / payload start in the batch
struct GbCols { // device copies of the fixed columns + per-record payload placement
    uint16_t* flag; uint8_t* mapq; uint8_t* lane; int32_t* rid; int32_t* pos; int32_t* tlen; int32_t* nm; int32_t* as; uint32_t* l_seq; uint16_t* n_cigar;
    uint32_t* rec_off; uint64_t* so; uint64_t* qo; uint64_t* co;
};
struct GbLanes { const uint8_t* blob; const uint32_t* off; const uint32_t* len; const uint32_t* index; uint32_t n, lane_count; };

typedef uint32_t __attribute__((aligned(1))) gb_u32_u;
typedef uint32_t gb_u32x4 __attribute__((ext_vector_type(4)));
typedef gb_u32x4 __attribute__((aligned(1))) gb_u32x4_u;
typedef uint16_t __attribute__((aligned(1))) gb_u16_u;

namespace {
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { return *(const gb_u32_u*)p; }
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { return *(const gb_u16_u*)p; }

// host/bam_io.cpp: plausible_record
__device__ bool gb_plausible(const uint8_t* base, uint64_t avail, uint64_t p, int32_t n_ref, uint64_t& next)
{
    if (p + 36 > avail) return false;
    const uint32_t bs = ld32(base + p);
    if (bs < 32u || bs > (1u << 28)) return false;
    const uint8_t* r = base + p + 4;
    const int32_t rid = (int32_t)ld32(r), pos = (int32_t)ld32(r + 4), rnext = (int32_t)ld32(r + 20), pnext = (int32_t)ld32(r + 24);
    if (rid < -1 || rid >= n_ref || rnext < -1 || rnext >= n_ref || pos < -1 || pnext < -1) return false;
    const uint32_t l_name = r[8], n_cig = ld16(r + 12), l_seq = ld32(r + 16);
    if (l_name == 0 || l_seq > (1u << 28)) return false;
    const uint64_t var = 32ull + l_name + 4ull * n_cig + (l_seq + 1u) / 2u + l_seq;
    if (var > bs) return false;
    if (p + 4 + 32 + l_name <= avail && r[32 + l_name - 1] != 0) return false; // read name is NUL-terminated
    next = p + 4 + bs;
    return true;
}
} // namespace

// lane per segment: first record start (segment 0: the window's start, which is one), then the chain up to the segment's end
// (seg0, exact: the segments from seg0 on, the first of them from the known record start `exact` — the whole window is
// (0, 0); a segment whose guess turned out wrong is walked again alone from where the chain arrives; exact == UINT64_MAX:
// a shard that starts in the middle of the file does not know its first record: segment seg0 guesses like the others).
// `limit`: a shard's end — no record that starts at or behind it is listed (S.exit is then where that record starts).
__global__ __launch_bounds__(64) void k_gb_walk(const uint8_t* __restrict__ base, uint64_t avail, uint32_t seg0, uint32_t nseg, uint64_t exact, uint64_t limit, int32_t n_ref,
                                                 GbSeg* __restrict__ segs, GbRec* __restrict__ recs)
{
    const uint32_t s = seg0 + blockIdx.x * 64 + threadIdx.x;
    if (s >= seg0 + nseg) return;
    const uint64_t a = (uint64_t)s * GB_SEG, b = min(avail, a + GB_SEG);
    GbSeg S{};
    const bool known = s == seg0 && exact != UINT64_MAX;
    uint64_t p = known ? exact : a;
    if (!known) {
        const uint64_t stop = min(b, limit);
        for (; p < stop; ++p) {
            uint64_t q1, q2, q3;
            if (gb_plausible(base, avail, p, n_ref, q1) && gb_plausible(base, avail, q1, n_ref, q2) && gb_plausible(base, avail, q2, n_ref, q3)) break;
        }
        if (p >= stop) { S.first = S.exit = 0xFFFFFFFFu; S.flags = GB_NO_START; segs[s] = S; return; }
    }
    S.first = (uint32_t)p;
    GbRec* out = recs + (size_t)s * GB_MAXR;
    uint32_t so = 0, qo = 0, co = 0, n = 0;
    // The chain is one dependent load per record, and every record of a lane's own 16 KiB lies in a line nobody has touched: a miss
    // each (~2.5 us; 60 records per segment: what the kernel's 0.26 ms per launch were, with a quarter of the card's SIMDs holding a
    // wave).  So every step also asks for the line 1.5 KB further on — a load nothing waits for — and finds its own line on its way
    // or there when it gets to it.  (`ahead` only keeps the loads alive.)
    uint32_t ahead = 0;
    const uint64_t last_line = avail >= 4 ? avail - 4 : 0;
    while (p < b && p < limit) {
        if (p + 36 > avail) { S.flags |= GB_INCOMPLETE; break; }
        ahead ^= ld32(base + min(p + 1536u, last_line));
        const uint32_t bs = ld32(base + p);
        if (bs < 32u) { S.flags |= GB_CORRUPT; break; }
        const uint8_t* r = base + p + 4;
        const uint32_t l_name = r[8], n_cig = ld16(r + 12), l_seq = ld32(r + 16);
        const uint64_t var = 32ull + l_name + 4ull * n_cig + ((uint64_t)l_seq + 1u) / 2u + l_seq;
        if (var > bs) { S.flags |= GB_CORRUPT; break; }
        if (p + 4 + bs > avail) { S.flags |= GB_INCOMPLETE; break; }
        out[n] = GbRec{(uint32_t)p, so, qo, co};
        so += (l_seq + 1u) / 2u; qo += l_seq; co += n_cig;
        ++n;
        p += 4ull + bs;
    }
    S.exit = (uint32_t)min(p, (uint64_t)0xFFFFFFFEu);
    S.count = n; S.seq_bytes = so; S.qual_bytes = qo; S.cigar_words = co;
    if (n == 0xFFFFFFFFu) S.pad = ahead; // (never: a segment holds at most GB_MAXR records)
    segs[s] = S;
}

// The same walk by a WAVE per segment (round 4; BQC_GB_WALK=lane: the kernel above).  With a lane per segment a batch's window is 280 waves
// on 1024 SIMDs, and a lane's 60 records are 60 dependent loads from lines nobody has touched (~2 us each under the inflate kernels'
// traffic) behind ~135 serial plausibility tests: 0.26-0.37 ms per launch, 9 % of a run's GPU time.  Here the wave first brings its
// segment (and 192 bytes beyond it: the header of a record that starts at its very end) into LDS with ONE round of coalesced 16-byte
// loads; the guess tests 64 candidate starts at a time (ballot: the first one that holds), and the chain reads a record's header
// from LDS — every lane the same addresses, a broadcast — ~150 clocks per record instead of a memory latency.  What lies beyond the
// staged bytes (the records a plausibility test follows past the segment's end) is read from memory as before.  Same results, field
// for field: tests/test_gpu_reader.py compares both with the host reader's walk.
#define GBW_STAGE (GB_SEG + 192u)
namespace {
struct GbwView {
    const uint32_t* buf;   // LDS copy of base[a, a + staged)
    const uint8_t* base;
    uint64_t a, staged;
    // four bytes at window offset p (unaligned): from the staged copy when they lie inside it (two aligned words, v_alignbyte)
    __device__ __forceinline__ uint32_t u32(uint64_t p) const
    {
        if (p >= a && p + 4 <= a + staged) {
            const uint32_t r = (uint32_t)(p - a), w = r >> 2, sh = r & 3u;
            const uint32_t lo = buf[w], hi = buf[w + 1]; // (buf has a spare word behind the staged bytes)
            return sh ? __builtin_amdgcn_alignbyte(hi, lo, sh) : lo;
        }
        return ld32(base + p);
    }
    __device__ __forceinline__ uint32_t u8(uint64_t p) const { return p >= a && p < a + staged ? (buf[(uint32_t)(p - a) >> 2] >> (8u * ((uint32_t)(p - a) & 3u))) & 255u : base[p]; }
};
// gb_plausible over a view
__device__ __forceinline__ bool gbw_plausible(const GbwView& V, uint64_t avail, uint64_t p, int32_t n_ref, uint64_t& next)
{
    if (p + 36 > avail) return false;
    const uint32_t bs = V.u32(p);
    if (bs < 32u || bs > (1u << 28)) return false;
    const int32_t rid = (int32_t)V.u32(p + 4), pos = (int32_t)V.u32(p + 8), rnext = (int32_t)V.u32(p + 24), pnext = (int32_t)V.u32(p + 28);
    if (rid < -1 || rid >= n_ref || rnext < -1 || rnext >= n_ref || pos < -1 || pnext < -1) return false;
    const uint32_t w12 = V.u32(p + 12), l_name = w12 & 255u, n_cig = V.u32(p + 16) & 0xFFFFu, l_seq = V.u32(p + 20);
    if (l_name == 0 || l_seq > (1u << 28)) return false;
    const uint64_t var = 32ull + l_name + 4ull * n_cig + (l_seq + 1u) / 2u + l_seq;
    if (var > bs) return false;
    if (p + 4 + 32 + l_name <= avail && V.u8(p + 4 + 32 + l_name - 1) != 0) return false; // read name is NUL-terminated
    next = p + 4 + bs;
    return true;
}
} // namespace

__global__ __launch_bounds__(64) void k_gb_walk_wave(const uint8_t* __restrict__ base, uint64_t avail, uint32_t seg0, uint32_t nseg, uint64_t exact, uint64_t limit, int32_t n_ref,
                                                      GbSeg* __restrict__ segs, GbRec* __restrict__ recs)
{
    __shared__ uint32_t buf[GBW_STAGE / 4 + 4];
    const uint32_t s = seg0 + blockIdx.x, lane = threadIdx.x;
    if (s >= seg0 + nseg) return;
    const uint64_t a = (uint64_t)s * GB_SEG, b = min(avail, a + GB_SEG);
    const uint64_t staged = min(avail - a, (uint64_t)GBW_STAGE) & ~(uint64_t)3; // whole words (a tail of 1-3 bytes at the window's end is read from memory)
    { // (round 4: every load of the stage issued before the first is waited for — as a loop of load-then-store the seventeen 16-byte loads
      // of a lane were seventeen round trips one after the other, ~30 of the wave's ~40 us)
        constexpr uint32_t kSteps = (GBW_STAGE + 1023u) / 1024u;
        gb_u32x4 v[kSteps];
#pragma unroll
        for (uint32_t k = 0; k < kSteps; ++k) {
            const uint32_t off = lane * 16u + 1024u * k;
            v[k] = gb_u32x4{0, 0, 0, 0};
            if (off + 16u <= (uint32_t)staged) v[k] = *(const gb_u32x4_u*)(base + a + off);
        }
#pragma unroll
        for (uint32_t k = 0; k < kSteps; ++k) {
            const uint32_t off = lane * 16u + 1024u * k;
            if (off + 16u <= (uint32_t)staged) { buf[off / 4] = v[k].x; buf[off / 4 + 1] = v[k].y; buf[off / 4 + 2] = v[k].z; buf[off / 4 + 3] = v[k].w; }
            else if (off < (uint32_t)staged) for (uint32_t q = off; q < (uint32_t)staged; q += 4u) buf[q / 4] = ld32(base + a + q);
        }
    }
    if (lane == 0) buf[staged / 4] = 0;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __syncthreads();
    const GbwView V{buf, base, a, staged};
    GbSeg S{};
    const bool known = s == seg0 && exact != UINT64_MAX;
    uint64_t p = known ? exact : a;
    if (!known) { // 64 candidate starts at a time: the first at which three records in a row look like records
        const uint64_t stop = min(b, limit);
        bool found = false;
        for (uint64_t r0 = a; r0 < stop && !found; r0 += 64u) {
            const uint64_t c = r0 + lane;
            uint64_t q1, q2, q3;
            const bool ok = c < stop && gbw_plausible(V, avail, c, n_ref, q1) && gbw_plausible(V, avail, q1, n_ref, q2) && gbw_plausible(V, avail, q2, n_ref, q3);
            const uint64_t m = __ballot(ok);
            if (m) { p = r0 + (uint64_t)(__ffsll((unsigned long long)m) - 1); found = true; }
        }
        if (!found) { if (lane == 0) { S.first = S.exit = 0xFFFFFFFFu; S.flags = GB_NO_START; segs[s] = S; } return; }
    }
    S.first = (uint32_t)p;
    GbRec* out = recs + (size_t)s * GB_MAXR;
    uint32_t so = 0, qo = 0, co = 0, n = 0;
    // (every lane the same record: the header's words are broadcast reads.)  Round 4: the four header words of a record come from six LDS
    // words requested together (four reads of two words each, each waited for, before), and the records' entries leave 64 at a time —
    // lane k keeps the entry of record 64 m + k — instead of one 16-byte store per record by lane 0: behind that store the next record's
    // `s_waitcnt vmcnt(0)` (the compiler's, for the path that reads the header from memory) waited for the store to retire, ~2 000 of the
    // ~2 200 clocks a record took.
    GbRec mine{0, 0, 0, 0};
    // (positions relative to the segment's start inside the loop: 32-bit compares; the data at hand beyond 2^31 bytes behind the segment's
    // start is as good as endless for a record that starts inside the segment)
    const uint64_t stop64 = min(b, limit);
    const uint32_t stop_r = stop64 > a ? (uint32_t)(stop64 - a) : 0u;                         // <= GB_SEG
    const uint32_t avail_r = (uint32_t)min(avail - a, (uint64_t)0x7FFFFFFFu), staged_r = (uint32_t)staged;
    uint32_t r = p >= a && p - a < (uint64_t)stop_r ? (uint32_t)(p - a) : 0xFFFFFFFFu;        // (outside: nothing to walk; p stays what it is)
    while (r < stop_r) {
        if (r + 36u > avail_r) { S.flags |= GB_INCOMPLETE; break; }
        uint32_t bs, w12, w16, l_seq;
        if (r + 24u <= staged_r) {
            const uint32_t w = r >> 2, sh = r & 3u;
            const uint32_t x0 = buf[w], x1 = buf[w + 1], x3 = buf[w + 3], x4 = buf[w + 4], x5 = buf[w + 5], x6 = buf[w + 6]; // (buf has a spare word behind the staged bytes)
            bs = __builtin_amdgcn_alignbyte(x1, x0, sh); w12 = __builtin_amdgcn_alignbyte(x4, x3, sh);
            w16 = __builtin_amdgcn_alignbyte(x5, x4, sh); l_seq = __builtin_amdgcn_alignbyte(x6, x5, sh);
        } else { const uint64_t q = a + r; bs = V.u32(q); w12 = V.u32(q + 12); w16 = V.u32(q + 16); l_seq = V.u32(q + 20); } // (the last record of the data at hand)
        if (bs < 32u) { S.flags |= GB_CORRUPT; break; }
        const uint32_t l_name = w12 & 255u, n_cig = w16 & 0xFFFFu;
        const uint64_t var = 32ull + l_name + 4ull * n_cig + ((uint64_t)l_seq + 1u) / 2u + l_seq;
        if (var > bs) { S.flags |= GB_CORRUPT; break; }
        const uint64_t next = (uint64_t)r + 4u + bs; // (64 bits: a block_size near 2^32 must not wrap into the segment)
        if (next > avail - a) { S.flags |= GB_INCOMPLETE; break; }
        if (lane == (n & 63u)) mine = GbRec{(uint32_t)(a + r), so, qo, co};
        so += (l_seq + 1u) / 2u; qo += l_seq; co += n_cig;
        ++n;
        if ((n & 63u) == 0u) out[n - 64u + lane] = mine;
        if (next >= stop_r) { r = 0xFFFFFFFFu; p = a + next; break; } // the walk leaves the segment (or the data to look at)
        r = (uint32_t)next;
    }
    if (r != 0xFFFFFFFFu) p = a + r; // (stopped at a record it could not take)
    if (lane < (n & 63u)) out[(n & ~63u) + lane] = mine;
    if (lane == 0) {
        S.exit = (uint32_t)min(p, (uint64_t)0xFFFFFFFEu);
        S.count = n; S.seq_bytes = so; S.qual_bytes = qo; S.cigar_words = co;
        segs[s] = S;
    }
}

// workgroup per taken segment, thread per record: the fixed columns and the tag scan of host/bam_io.cpp
#define GBD_STAGE 64u // bytes of a record's optional fields staged in LDS (k_gb_decode)
__global__ __launch_bounds__(64) void k_gb_decode(const uint8_t* __restrict__ base, const GbSeg* __restrict__ segs, const GbRec* __restrict__ recs,
                                                   const GbBase* __restrict__ bases, GbCols C, GbLanes LN, const uint8_t* __restrict__ main_chrom, uint32_t n_main,
                                                   uint32_t* __restrict__ status)
{
    __shared__ uint32_t tagbuf[64][GBD_STAGE / 4 + 1]; // a record's first optional fields, per thread (an odd stride: the lanes' words in different banks)
    const uint32_t s = blockIdx.x;
    const GbBase B = bases[s];
    if (!B.take) return;
    const uint32_t count = segs[s].count;
    uint32_t exc = 0;
    for (uint32_t slot = threadIdx.x; slot < count; slot += 64) {
        const GbRec R = recs[(size_t)s * GB_MAXR + slot];
        const uint32_t i = B.rec + slot;
        const uint8_t* r = base + R.off + 4;
        const uint32_t bs = ld32(r - 4);
        const int32_t rid = (int32_t)ld32(r), pos = (int32_t)ld32(r + 4);
        const uint32_t l_name = r[8], mapq = r[9], n_cig = ld16(r + 12), flag = ld16(r + 14), l_seq = ld32(r + 16);
        const int32_t rnext = (int32_t)ld32(r + 20), tlen = (int32_t)ld32(r + 28);
        const uint8_t* ql = r + 32 + l_name + 4ull * n_cig + (l_seq + 1u) / 2u;
        const uint8_t* const tg0 = ql + l_seq;
        const uint8_t* te = r + bs;
        int lane = -1;
        bool rg_seen = false, as_seen = false, nm_seen = false;
        int32_t nm = BQC_NM_ABSENT, as = BQC_AS_ABSENT;
        // The optional fields are walked byte by byte, every byte the address of the next: from memory that was a chain of dependent
        // loads (the kernel's whole time).  Round 4: a record's first GBD_STAGE bytes of fields come into LDS with four 16-byte loads
        // issued together (what lies behind them, if anything, is read from memory as before; the buffer's slack covers the over-read).
        const uint64_t tlen_all = te > tg0 ? (uint64_t)(te - tg0) : 0;
        uint32_t* const tb = tagbuf[threadIdx.x];
        {
            gb_u32x4 v[GBD_STAGE / 16];
#pragma unroll
            for (uint32_t k = 0; k < GBD_STAGE / 16; ++k) v[k] = 16u * k < tlen_all ? *(const gb_u32x4_u*)(tg0 + 16u * k) : gb_u32x4{0, 0, 0, 0};
#pragma unroll
            for (uint32_t k = 0; k < GBD_STAGE / 16; ++k) { tb[4 * k] = v[k].x; tb[4 * k + 1] = v[k].y; tb[4 * k + 2] = v[k].z; tb[4 * k + 3] = v[k].w; }
        }
        auto t8 = [&](uint64_t o) -> uint32_t { return o < GBD_STAGE ? (tb[(uint32_t)o >> 2] >> (8u * ((uint32_t)o & 3u))) & 255u : (uint32_t)tg0[o]; };
        auto t16 = [&](uint64_t o) -> uint32_t { return t8(o) | (t8(o + 1) << 8); };
        auto t32 = [&](uint64_t o) -> uint32_t {
            if (o + 4 <= GBD_STAGE) { const uint32_t w = (uint32_t)o >> 2, sh = (uint32_t)o & 3u; return __builtin_amdgcn_alignbyte(tb[w + 1], tb[w], sh); } // (a spare word behind the stage)
            return t8(o) | (t8(o + 1) << 8) | (t8(o + 2) << 16) | (t8(o + 3) << 24);
        };
        uint64_t tg = 0; // offset of the next field from tg0
        while (tg + 3 <= tlen_all) {
            const char k0 = (char)t8(tg), k1 = (char)t8(tg + 1), ty = (char)t8(tg + 2);
            const uint64_t v = tg + 3;
            uint64_t len = 0;
            switch (ty) {
            case 'A': case 'c': case 'C': len = 1; break;
            case 's': case 'S': len = 2; break;
            case 'i': case 'I': case 'f': len = 4; break;
            case 'Z': case 'H': {
                uint64_t z = v;
                while (z < tlen_all && t8(z)) ++z;
                len = z < tlen_all ? (z - v) + 1 : tlen_all - v;
                break;
            }
            case 'B': {
                if (v + 5 > tlen_all) { len = tlen_all - v; break; }
                const char st = (char)t8(v);
                const uint64_t cnt = t32(v + 1);
                const uint64_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
                len = 5 + cnt * es;
                break;
            }
            default: len = tlen_all - v; break;
            }
            if (len > tlen_all - v) { exc |= GBX_TAGS; break; }
            if (k0 == 'R' && k1 == 'G' && !rg_seen) {
                rg_seen = true;
                if (ty == 'Z') {
                    const uint32_t idl = len ? (uint32_t)len - 1 : 0;
                    for (uint32_t l = 0; l < LN.n && lane < 0; ++l) {
                        if (LN.len[l] != idl) continue;
                        const uint8_t* id = LN.blob + LN.off[l];
                        uint32_t k = 0;
                        while (k < idl && id[k] == t8(v + k)) ++k;
                        if (k == idl) lane = (int)LN.index[l];
                    }
                    if (lane < 0) { exc |= GBX_RG_UNKNOWN; lane = 0; }
                } else exc |= GBX_RG_TYPE;
            } else if (k0 == 'N' && k1 == 'M' && (ty == 'c' || ty == 'C' || ty == 's' || ty == 'S' || ty == 'i' || ty == 'I')) {
                uint32_t x;
                switch (ty) {
                case 'c': x = (uint32_t)(int32_t)(int8_t)t8(v); break;
                case 'C': x = t8(v); break;
                case 's': x = (uint32_t)(int32_t)(int16_t)t16(v); break;
                case 'S': x = t16(v); break;
                default: x = t32(v); break;
                }
                if (!nm_seen) { nm = (int32_t)x; nm_seen = true; }
                else exc |= GBX_NM_EXTRA;
            } else if (k0 == 'A' && k1 == 'S' && !as_seen) {
                as_seen = true;
                switch (ty) {
                case 'A': as = (int32_t)(char)t8(v); break;
                case 'c': as = (int8_t)t8(v); break;
                case 'C': as = (int32_t)t8(v); break;
                case 's': as = (int16_t)t16(v); break;
                case 'S': as = (int32_t)t16(v); break;
                case 'i': case 'I': as = (int32_t)t32(v); break;
                case 'f': as = (int32_t)__uint_as_float(t32(v)); break;
                default: as = BQC_AS_ABSENT; break;
                }
            }
            tg = v + len;
        }
        if (!rg_seen) exc |= GBX_RG_MISSING;
        else if ((uint32_t)lane >= LN.lane_count) exc |= GBX_LANE;
        if (nm_seen && nm == BQC_NM_ABSENT) exc |= GBX_NM_VALUE;
        uint32_t f = flag & 0x0FFFu;
        if (rnext >= 0 && (uint32_t)rnext < n_main && main_chrom[rnext]) f |= BQC_FLAG_MATE_MAIN;
        if (l_seq > 0 && ql[0] == 0xFF) f |= BQC_FLAG_NO_QUAL;
        C.flag[i] = (uint16_t)f; C.mapq[i] = (uint8_t)mapq; C.lane[i] = (uint8_t)(lane < 0 ? 0 : lane); C.rid[i] = rid; C.pos[i] = pos; C.tlen[i] = tlen;
        C.nm[i] = nm; C.as[i] = as; C.l_seq[i] = l_seq; C.n_cigar[i] = (uint16_t)n_cig;
        C.rec_off[i] = R.off; C.so[i] = B.so + R.so; C.qo[i] = B.qo + R.qo; C.co[i] = B.co + R.co;
    }
    if (exc) atomicOr(status, exc);
}

// CIGAR words, packed bases and qualities into the batch's packed columns: 16 lanes per record, 16 bytes per lane and step.  A 150 bp
// record is 10 pieces of qualities, 5 of bases and one of CIGAR: one step with every lane busy (a wave per record moved the three
// arrays one after the other, 4 bytes per lane, with 19, 38 and 1 of 64 lanes busy: 0.67 ms per million records, three dependent round
// trips per wave).  A last piece of fewer than 16 bytes is moved as the array's last 16 (it overlaps the piece before it: the same
// bytes twice); an array shorter than 16 bytes by dwords and bytes.  Long reads: the 16 lanes stride over the pieces.
__global__ __launch_bounds__(256) void k_gb_copy(const uint8_t* __restrict__ base, GbCols C, uint32_t n, uint8_t* __restrict__ seq, uint8_t* __restrict__ qual,
                                                  uint8_t* __restrict__ cigar /* bytes */)
{
    const uint32_t i = blockIdx.x * 16 + (threadIdx.x >> 4), gl = threadIdx.x & 15u;
    if (i >= n) return;
    const uint8_t* r = base + C.rec_off[i] + 4;
    const uint32_t l_name = r[8], n_cig = C.n_cigar[i], l_seq = C.l_seq[i];
    const uint8_t* cg = r + 32 + l_name;
    const uint8_t* sq = cg + 4ull * n_cig;
    const uint8_t* ql = sq + (l_seq + 1u) / 2u;
    uint8_t* dc = cigar + 4ull * C.co[i];
    uint8_t* ds = seq + C.so[i];
    uint8_t* dq = qual + C.qo[i];
    const uint32_t nc = 4u * n_cig, ns = (l_seq + 1u) / 2u, nq = l_seq;
    const uint32_t pq = (nq + 15u) >> 4, ps = (ns + 15u) >> 4, pc = (nc + 15u) >> 4;
    for (uint32_t k = gl; k < pq + ps + pc; k += 16u) {
        const uint8_t* s_;
        uint8_t* d;
        uint32_t len, piece;
        if (k < pq) { s_ = ql; d = dq; len = nq; piece = k; }
        else if (k < pq + ps) { s_ = sq; d = ds; len = ns; piece = k - pq; }
        else { s_ = cg; d = dc; len = nc; piece = k - pq - ps; }
        const uint32_t at = 16u * piece;
        if (at + 16u <= len) *(gb_u32x4_u*)(d + at) = *(const gb_u32x4_u*)(s_ + at);
        else if (len >= 16u) *(gb_u32x4_u*)(d + len - 16u) = *(const gb_u32x4_u*)(s_ + len - 16u);
        else {
            uint32_t b_ = 0;
            for (; b_ + 4u <= len; b_ += 4u) *(gb_u32_u*)(d + b_) = *(const gb_u32_u*)(s_ + b_);
            for (; b_ < len; ++b_) d[b_] = s_[b_];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
namespace {
template <typename T> struct DevBuf { // grows, never shrinks
    T* p = nullptr;
    size_t cap = 0;
    bool need(size_t n, bool exact = false)
    {
        if (cap >= n) return true;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        const size_t c = exact ? n : n + n / 4 + 64;
        if (hipMalloc((void**)&p, c * sizeof(T)) != hipSuccess) { p = nullptr; return false; }
        cap = c;
        return true;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
};
template <typename T> struct PinBuf {
    T* p = nullptr;
    size_t cap = 0;
    bool need(size_t n)
    {
        if (cap >= n) return true;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t c = n + n / 4 + 64;
        if (hipHostMalloc((void**)&p, c * sizeof(T), hipHostMallocDefault) != hipSuccess) { p = nullptr; return false; }
        cap = c;
        return true;
    }
    ~PinBuf() { if (p) (void)hipHostFree(p); }
};
// The batches' payload buffers: allocated when a reader opens (before its first kernel runs: hipMalloc and hipFree behind a running
// 30 ms inflate kernel were measured to wait for it), all of one size, handed back here when a batch object dies.
struct PayloadPool {
    std::mutex m;
    std::vector<void*> free_list;          // buffers of `cap` bytes
    std::vector<std::pair<void*, size_t>> owned; // every buffer the pool has allocated and not released, with its size
    size_t cap = 0;
    void* take(size_t need)
    {
        std::lock_guard<std::mutex> lk(m);
        if (need > cap || free_list.empty()) return nullptr;
        void* p = free_list.back();
        free_list.pop_back();
        return p;
    }
    void fill(size_t bytes, int n)
    {
        std::lock_guard<std::mutex> lk(m);
        if (bytes > cap) { // larger batches than before: the smaller buffers go (those handed out are released when they come back)
            for (void* p : free_list) { forget(p); (void)hipFree(p); }
            free_list.clear();
            cap = bytes;
        }
        while ((int)free_list.size() < n) {
            void* p = nullptr;
            if (hipMalloc(&p, cap) != hipSuccess) break;
            free_list.push_back(p);
            owned.emplace_back(p, cap);
        }
    }
    void give(void* p) // a batch object lets go of its buffer (the pool's or its own)
    {
        std::lock_guard<std::mutex> lk(m);
        size_t bytes = 0;
        for (auto& kv : owned) if (kv.first == p) bytes = kv.second;
        if (bytes == cap && bytes && free_list.size() < 16) { free_list.push_back(p); return; }
        forget(p);
        (void)hipFree(p);
    }
    void forget(void* p)
    {
        for (size_t i = 0; i < owned.size(); ++i) if (owned[i].first == p) { owned[i] = owned.back(); owned.pop_back(); return; }
    }
};
PayloadPool g_pool;
void dev_free_hook(void* p) { g_pool.give(p); }
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
} // namespace

// One run of BGZF blocks in flight: its own stream, page-locked input buffer and device buffers.  The uncompressed bytes land behind
// `head` spare bytes, which take the unfinished record of the run before (the record stream is walked run by run).
struct GbRun {
    hipStream_t s = nullptr;
    hipEvent_t ready = nullptr, copied = nullptr;
    DevBuf<uint8_t> d_comp;
    std::vector<GiBlock> hb; DevBuf<GiBlock> d_blocks; // block table, host / device
    std::vector<uint32_t> hc; DevBuf<uint32_t> d_crc;  // expected CRC-32s
    DevBuf<uint8_t> d_out;
    DevBuf<uint32_t> d_tok, d_ntok; // scratch of the inflate kernels' two phases
    uint32_t* d_status = nullptr;
    size_t utotal = 0;
    uint64_t abs0 = 0;   // where this run's first inflated byte lies in the reader's uncompressed stream (which starts at its begin block)
    int state = 0;       // 0 free, 1 being filled, 2 ready (under Impl::m)
    bool final = false;  // nothing follows (this one may be empty)
    int rc = 1;          // 1 ok, -1 malformed (err), -2 device
    std::string err;
};

struct GpuBamReader::Impl {
    int device = 0;
    hipStream_t s = nullptr;    // the consumer's: walk, decode, copies
    hipStream_t ps = nullptr;   // the producer's: inflate kernels of every run (and the first run's copy)
    hipStream_t cs = nullptr;   // the producer's copies from the second run on (made when that run starts: off the start-up's path).  On ONE stream the
                                // chunks of run k + 1 queued behind the inflate kernels of run k (17 ms of copies behind 25-37 ms of kernels per 832 MB run),
                                // their slots came back late and the reader threads stood still: on the 53.6 GB file the producer waited 2.4 s of a 3.1 s
                                // loop for chunks (profiles/r4_config3_file.json, BQC_GB_COPY_STREAM=0)
    hipEvent_t ev = nullptr;    // blocking
    int fd = -1;                // the file (pread from several threads)
    uint64_t skip_u = 0;        // uncompressed bytes in front of the first record, still to be dropped
    // a shard of the file (GpuBamReader::set_range): the producer reads from begin_off on and notes where the block at mark_off
    // lies in the uncompressed stream; behind it only small runs follow (the consumer wants the rest of one record)
    uint64_t begin_off = 0, mark_off = UINT64_MAX;
    uint64_t parsed_off = 0;    // (producer) file offset of the next block header to parse
    uint64_t u_produced = 0;    // (producer) uncompressed bytes of the runs before the current one
    std::atomic<uint64_t> mark_abs{UINT64_MAX}; // the mark in the uncompressed stream, once the producer has got there
    static constexpr uint64_t kBeyond = 2u << 20; // compressed bytes a run takes behind the mark
    bool need_locate = false, range_done = false; // (consumer)
    uint64_t win_abs0 = 0;      // (consumer) position of win[head] in the uncompressed stream
    uint64_t abs_of(size_t x) const { return win_abs0 + x - head; } // (x < head: what the run before left over)
    uint64_t stop_off() const { return mark_off == UINT64_MAX ? UINT64_MAX : (parsed_off < mark_off ? mark_off : parsed_off) + kBeyond; }
    size_t run_bytes = 832u << 20; // (large inflate launches are the cheapest per block: few, large runs; the first one is 64 MB)
    size_t head = 16u << 20;    // room in front of a run's output for the unfinished record before it
    size_t out_cap = 0;         // bytes of a run's output buffer (head included): set in open() before the producer starts
    static const int kRuns = 3; // one being walked, one being inflated, one being read
    GbRun runs[kRuns];
    // producer thread: file -> runs
    std::thread producer;
    std::mutex m;
    std::condition_variable cv;
    bool stop = false;
    uint64_t produced = 0, taken = 0, freed = 0; // runs handed over / taken by the consumer / given back (run k lives in runs[k % kRuns])
    bool file_eof = false;
    // consumer: the window [cur, end) of the current run's buffer
    GbRun* cur_run = nullptr;
    uint8_t* win = nullptr;
    size_t cur = 0, end = 0;
    bool stream_done = false;
    uint32_t* h_status = nullptr;
    DevBuf<GbSeg> d_seg; PinBuf<GbSeg> h_seg;
    DevBuf<GbRec> d_rec;
    DevBuf<GbBase> d_base; PinBuf<GbBase> h_base;
    DevBuf<uint8_t> d_cols;
    DevBuf<uint8_t> d_lane_blob; DevBuf<uint32_t> d_lane_tab; uint32_t n_lane_ids = 0, lane_count = 0;
    DevBuf<uint8_t> d_main; uint32_t n_main = 0; bool main_set = false;
    uint32_t* d_status = nullptr; // of the decode kernels
    int32_t n_ref = 0;
    bool timing = false;
    double t_read = 0, t_wait_run = 0;
    std::atomic<double> t_wait_chunk{0}; // (producer thread writes, consumer reads)
    uint64_t n_rewalk = 0;
    raw_vector<uint8_t> handover_raw;      // a batch the host decoder takes: its bytes
    std::vector<BamRec> handover_recs;
    double avg_rec_bytes = 0, avg_rec_bases = 0;
    uint64_t grow_window = 0;

    ~Impl()
    {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv.notify_all();
        { std::lock_guard<std::mutex> lk(rm); rstop = true; }
        rcv.notify_all();
        if (producer.joinable()) producer.join();
        for (std::thread& t : readers) if (t.joinable()) t.join();
        (void)hipSetDevice(device);
        if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); }
        if (ps) { (void)hipStreamSynchronize(ps); (void)hipStreamDestroy(ps); }
        if (ring_alloc.joinable()) ring_alloc.join();
        for (Slot& C : slots) { if (C.done) (void)hipEventDestroy(C.done); if (C.p) { if (C.registered) (void)hipHostUnregister(C.p); free(C.p); } }
        for (GbRun& R : runs) {
            if (R.ready) (void)hipEventDestroy(R.ready);
            if (R.copied) (void)hipEventDestroy(R.copied);
            if (R.d_status) (void)hipFree(R.d_status);
        }
        if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
        if (ev) (void)hipEventDestroy(ev);
        if (d_status) (void)hipFree(d_status);
        if (h_status) (void)hipHostFree(h_status);
        if (fd >= 0) close(fd);
    }
    bool sync() { return hipEventRecord(ev, s) == hipSuccess && hipEventSynchronize(ev) == hipSuccess; }
    // read-group ids -> lane index, column-wise: off[n] len[n] index[n] (again when a batch handed over to the host decoder has met
    // ids that are not in the header: they are lane 0 from then on, bamqualcheck.cpp:86)
    bool upload_lanes(const BamHeader& hdr)
    {
        std::vector<uint8_t> blob;
        n_lane_ids = (uint32_t)hdr.lane_names.size();
        lane_count = hdr.lane_count;
        std::vector<uint32_t> cols(3 * (size_t)n_lane_ids + 1);
        uint32_t l = 0;
        for (const auto& kv : hdr.lane_names) {
            cols[l] = (uint32_t)blob.size(); cols[n_lane_ids + l] = (uint32_t)kv.first.size(); cols[2 * n_lane_ids + l] = kv.second;
            blob.insert(blob.end(), kv.first.begin(), kv.first.end());
            ++l;
        }
        if (!d_lane_blob.need(blob.size() + 1) || !d_lane_tab.need(cols.size())) return false;
        if (!blob.empty() && hipMemcpy(d_lane_blob.p, blob.data(), blob.size(), hipMemcpyHostToDevice) != hipSuccess) return false;
        return hipMemcpy(d_lane_tab.p, cols.data(), cols.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    }
    // The file behind the first run comes through a RING of page-locked chunks filled by several reader threads (one thread's
    // copy out of the page cache manages 8-9 GB/s, which a card that inflates 20+ GB/s of compressed input waits for): chunk i
    // is the file's bytes [ring_base + i * chunk_bytes, + chunk_bytes); the producer parses them in order, copies the whole
    // blocks to the card and gives the chunk back.  A block cut by a chunk's end is completed in the HEADROOM in front of the next
    // chunk's bytes, so that a block is always contiguous.
    struct Slot { uint8_t* p = nullptr; size_t len = 0; uint64_t index = UINT64_MAX; bool filled = false, used = false, registered = false; hipEvent_t done = nullptr; };
    std::thread ring_alloc; // allocates and touches the ring's chunks while the runtime starts, then page-locks them (hipHostRegister)
    int ring_state = 0;     // (under rm) 1: the chunks are there and page-locked, -1: that failed
    static const int kMaxSlots = 16;
    int kSlots = 6, kReaders = 3; // (BQC_GB_READERS=N: N reader threads and 2 N chunks, N <= 8)
    static constexpr size_t kHeadroom = 1u << 17;
    Slot slots[kMaxSlots];
    size_t chunk_bytes = 16u << 20; // (six of them are page-locked in open(): 16 ms; with 8 MB chunks the loop of a 54 GB file was 0.2 s longer)
    std::vector<std::thread> readers;
    std::mutex rm;
    std::condition_variable rcv;
    uint64_t ring_base = 0;          // file offset of chunk 0 (set when the first run has been parsed)
    bool ring_open = false;          // (under rm) ring_base is valid: the readers may start
    uint64_t next_claim = 0;         // (under rm) next chunk a reader takes
    uint64_t released = 0;           // (under rm) chunks below this one have been given back
    uint64_t read_limit = UINT64_MAX; // (under rm) chunks that begin at or behind this file offset are not read (a shard: nothing far behind its end)
    bool rstop = false;              // (under rm)
    uint64_t ring_i = 0;             // (producer) the chunk being parsed ...
    size_t ring_at = kHeadroom;      // ... and the first unparsed byte in its slot buffer (below kHeadroom: a block's head, carried over)
    void reader_loop();
    Slot* wait_chunk(uint64_t i);
    bool release_chunk(uint64_t i, hipStream_t st);
    size_t kMaxRunOut = (size_t)3200 << 20; // uncompressed bytes of a run (BQC_GB_MAX_RUN_OUT_MB: tests)
    raw_vector<uint8_t> first_raw; // the first run: read while the device is still starting
    bool dev_ready = false;        // (under m) streams and buffers exist: the producer may touch the device
    bool kernels_ok = false;       // (under m) GpuBamReader::allow_kernels(): the first inflate kernel may be launched
    void produce();
    void fill_run(GbRun& R);
    size_t parse_blocks(GbRun& R, const uint8_t* raw, size_t have, size_t d_off, size_t& nb, size_t& utotal, uint64_t stop_at, size_t out_limit, bool& stopped);
    bool wait_ready();
    // the next run becomes the window (what is left of the current one goes in front of it); 1: done, 0: no more runs, -1 / -2: see GbRun::rc
    int advance(std::string& err);
};

GpuBamReader::GpuBamReader() {}
void GpuBamReader::allow_kernels()
{
    if (!p_) return;
    { std::lock_guard<std::mutex> lk(p_->m); p_->kernels_ok = true; }
    p_->cv.notify_all();
}
GpuBamReader::~GpuBamReader() { delete p_; }

bool GpuBamReader::open(const char* path, int device, const BamHeader& hdr, uint64_t first_record_u, size_t batch_reads, size_t batch_bases, std::string& err)
{
    const double t_open0 = now_s();
    if (&hdr != &hdr_) hdr_ = hdr; // (a caller that has filled header() itself passes it: nothing is written while others read it)
    delete p_;
    p_ = new Impl();
    Impl& I = *p_;
    I.device = device;
    I.timing = getenv("BQC_GB_TIMING") != nullptr;
    if (const char* e = getenv("BQC_GB_RUN_MB")) I.run_bytes = (size_t)std::max(1, atoi(e)) << 20;
    if (const char* e = getenv("BQC_GB_READERS")) { I.kReaders = std::min(8, std::max(1, atoi(e))); I.kSlots = 2 * I.kReaders; }
    if (const char* e = getenv("BQC_GB_MAX_RUN_OUT_MB")) I.kMaxRunOut = (size_t)std::max(1, atoi(e)) << 20;
    I.fd = ::open(path, O_RDONLY);
    if (I.fd < 0) { err = std::string("could not open ") + path; return false; }
    if (ranged_) {
        I.begin_off = range_b0_; I.mark_off = range_b1_;
        I.need_locate = I.begin_off != 0;
        if (I.begin_off) first_record_u = 0; // (the header lies in the first shard)
    }
    I.parsed_off = I.begin_off;
    I.skip_u = first_record_u;
    I.n_ref = (int32_t)hdr.ref_names.size();
    { // (the sizes the producer works with are final before it starts)
        struct stat st;
        if (fstat(I.fd, &st) == 0 && st.st_size > 0) {
            const uint64_t upto = std::min<uint64_t>((uint64_t)st.st_size, I.mark_off == UINT64_MAX ? UINT64_MAX : I.mark_off + Impl::kBeyond + (1u << 17));
            I.run_bytes = std::min<size_t>(I.run_bytes, (size_t)(upto > I.begin_off ? upto - I.begin_off : 0) + (1u << 20));
        }
        I.out_cap = I.head + I.run_bytes / 5 * 18 + 64;
    }
    I.producer = std::thread([&I] { I.produce(); }); // reads the first run of the file while the device is set up below
    // the ring's chunks: allocated and touched here, by a thread of their own, while the runtime starts; page-locked below (measured,
    // tools/micro/startup_probe.cpp: hipHostMalloc of 6 x 16 MB 17-26 ms; touching them 18 ms — hidden behind the runtime's start — and
    // hipHostRegister 3-4 ms)
    I.ring_alloc = std::thread([&I, device] { // (... and page-locks them itself when the runtime is up: the ring is the SECOND run's business, open() does not wait for it)
        bool ok = true;
        for (int k_ = 0; k_ < I.kSlots; ++k_) {
            Impl::Slot& C = I.slots[k_];
            const size_t bytes = (Impl::kHeadroom + I.chunk_bytes + 64 + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
            C.p = (uint8_t*)aligned_alloc((size_t)2 << 20, bytes);
            if (C.p) memset(C.p, 0, bytes); else ok = false;
        }
        ok = ok && hipSetDevice(device) == hipSuccess;
        for (int k_ = 0; ok && k_ < I.kSlots; ++k_) {
            Impl::Slot& C = I.slots[k_];
            ok = hipHostRegister(C.p, Impl::kHeadroom + I.chunk_bytes + 64, hipHostRegisterDefault) == hipSuccess;
            C.registered = ok;
            ok = ok && hipEventCreateWithFlags(&C.done, hipEventBlockingSync | hipEventDisableTiming) == hipSuccess;
        }
        { std::lock_guard<std::mutex> lk(I.rm); I.ring_state = ok ? 1 : -1; }
        I.rcv.notify_all();
    });
    hipError_t he = hipSetDevice(device);
    const double t_open1 = now_s();
    if (he != hipSuccess) { err = std::string("GPU reader: ") + hipGetErrorString(he); return false; }
    // Everything the reader will need, now: an allocation (or a release) behind a running inflate kernel waits for that kernel.
    {
        // (a run inflates to ~3.3 x its size with BGZF level 1-6 on BAM records; a run that needs more grows its buffer, once.)  Sized
        // for what this file can need: a 20 GB set-up is 20 GB to hand back when the process ends.
        const size_t reads0 = std::min<size_t>(std::max<size_t>(batch_reads, 1), 1u << 22);
        const size_t out_cap = I.out_cap, nb_cap = I.run_bytes / 2048;
        const size_t seg_cap = std::min<size_t>(out_cap, std::min<size_t>(reads0 * 440, batch_bases * 2) + (8u << 20)) / GB_SEG + 2; // (a walk covers a batch's worth of the window)
        bool ok = true;
        const double t_a = now_s();
        for (int k = 0; k < 2; ++k) { // (the third run buffer is allocated when a third run comes)
            GbRun& R = I.runs[k];
            ok = ok && R.d_comp.need(I.run_bytes + I.chunk_bytes + (1u << 17) + 64, true) && R.d_blocks.need(nb_cap) && R.d_crc.need(nb_cap) && R.d_out.need(out_cap, true) &&
                 R.d_tok.need(bqc_gpu_inflate_token_words(out_cap, nb_cap), true) && R.d_ntok.need(nb_cap);
        }
        const double t_b = now_s();
        ok = ok && I.d_seg.need(seg_cap) && I.h_seg.need(seg_cap) && I.d_rec.need(seg_cap * GB_MAXR, true) && I.d_base.need(seg_cap) && I.h_base.need(seg_cap);
        const size_t reads = std::min<size_t>(std::max<size_t>(batch_reads, 1), 1u << 22);
        ok = ok && I.d_cols.need(((reads + 63 + GB_MAXR) & ~(size_t)63) * (7 * 4 + 3 * 8 + 2 * 2 + 2) + 256);
        if (!ok) { err = "GPU reader: out of device memory"; return false; }
        const size_t typical = std::min<size_t>(reads * 400, batch_bases / 2 * 3 + reads * 40 + (64u << 20)) + (1u << 20);
        const double t_c = now_s();
        g_pool.fill(typical, 10); // (a batch's buffer comes back when its kernels are through: three in the pipeline, three queued, the decoder's, spares)
        buffers_allocated = true; // (the caller creates its context from here on: side by side with these allocations the two were measured to hold each other up)
        if (I.timing) fprintf(stderr, "[gpu reader] open: run buffers %.1f, walk buffers %.1f, batch pool %.1f ms\n", (t_b - t_a) * 1e3, (t_c - t_b) * 1e3, (now_s() - t_c) * 1e3);
    }

    const double t_s0 = now_s();
    if (he == hipSuccess) { I.ps = bqc_pool_stream(device, 1); if (!I.ps) he = hipErrorOutOfMemory; } // (made ahead by bqc_warmup when the program runs; the consumer's stream: at the first batch)
    const double t_s1 = now_s();
    if (he == hipSuccess) he = hipEventCreateWithFlags(&I.ev, hipEventBlockingSync | hipEventDisableTiming);
    if (he == hipSuccess) he = hipMalloc((void**)&I.d_status, 64);
    if (he == hipSuccess) he = hipHostMalloc((void**)&I.h_status, 64, hipHostMallocDefault);
    if (he == hipSuccess) he = hipMemsetAsync(I.d_status, 0, 64, I.ps);
    // ONE stream for all runs: a process gets a handful of hardware queues, streams beyond them share one, and a 30 ms inflate
    // kernel in a shared queue holds up whatever else is in it
    for (GbRun& R : I.runs) {
        R.s = I.ps;
        if (he == hipSuccess) he = hipEventCreateWithFlags(&R.ready, hipEventBlockingSync | hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&R.copied, hipEventBlockingSync | hipEventDisableTiming);
        if (he == hipSuccess) he = hipMalloc((void**)&R.d_status, 64);
    }
    const double t_s2 = now_s();

    for (int t = 0; t < I.kReaders; ++t) I.readers.emplace_back([&I] { I.reader_loop(); }); // (they wait for the first run to be parsed)
    if (he != hipSuccess) { err = std::string("GPU reader: ") + hipGetErrorString(he); return false; }
    if (!I.upload_lanes(hdr)) { err = "GPU reader: out of device memory"; return false; }
    { std::lock_guard<std::mutex> lk(I.m); I.dev_ready = true; }
    I.cv.notify_all();
    if (I.timing) fprintf(stderr, "[gpu reader] open: %.1f ms (of which device set-up and buffers %.1f ms: the two streams %.1f, events + status words %.1f, page-locking the chunks + tables %.1f)\n", (now_s() - t_open0) * 1e3, (now_s() - t_open1) * 1e3, (t_s1 - t_s0) * 1e3, (t_s2 - t_s1) * 1e3, (now_s() - t_s2) * 1e3);
    return true;
}

void GpuBamReader::Impl::produce()
{
    for (;;) {
        GbRun* R;
        {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return stop || produced - freed < (uint64_t)kRuns; });
            if (stop) return;
            R = &runs[produced % kRuns];
            R->state = 1;
        }
        fill_run(*R);
        const bool last = R->final || R->rc != 1;
        {
            std::lock_guard<std::mutex> lk(m);
            R->state = 2;
            ++produced;
        }
        cv.notify_all();
        if (last) return;
    }
}

// reads the next run of whole BGZF blocks and starts its inflation (block headers: host/bgzf.cpp plan_run — same checks)
// whole BGZF blocks of raw[0, have): appended to the run's block table (their deflate data will lie at d_off + ... in the run's
// compressed buffer); returns the bytes they span, SIZE_MAX on a malformed stream (host/bgzf.cpp plan_run — same checks)
size_t GpuBamReader::Impl::parse_blocks(GbRun& R, const uint8_t* raw, size_t have, size_t d_off, size_t& nb, size_t& utotal, uint64_t stop_at, size_t out_limit, bool& stopped)
{
    const size_t kMaxBlock = 65536;
    size_t p = 0;
    while (p + 18 <= have) {
        if (mark_off != UINT64_MAX) { // a shard: where its end block lies in the uncompressed stream; the run ends a little behind it
            const uint64_t at = parsed_off + p;
            if (mark_abs.load() == UINT64_MAX) {
                if (at == mark_off) mark_abs = R.abs0 + utotal;
                else if (at > mark_off) { R.err = "the split point of the file is not a BGZF block boundary"; return SIZE_MAX; }
            }
            if (at >= stop_at) { stopped = true; break; }
        }
        const uint8_t* h = raw + p;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { R.err = "not a BGZF stream (bad gzip member header)"; return SIZE_MAX; }
        const size_t xlen = h[10] | (h[11] << 8);
        if (p + 12 + xlen > have) break;
        size_t bsize = 0, x = 12;
        while (x + 4 <= 12 + xlen) {
            const size_t slen = h[x + 2] | (h[x + 3] << 8);
            if (x + 4 + slen > 12 + xlen) { R.err = "corrupt BGZF block (extra subfield runs past the extra field)"; return SIZE_MAX; }
            if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2) bsize = (size_t)(h[x + 4] | (h[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        if (!bsize) { R.err = "BGZF block without BC extra field"; return SIZE_MAX; }
        if (bsize < 12 + xlen + 8) { R.err = "corrupt BGZF block (BSIZE smaller than header + trailer)"; return SIZE_MAX; }
        if (p + bsize > have) break;
        const uint8_t* t = raw + p + bsize - 8;
        const size_t isize = t[4] | (t[5] << 8) | (t[6] << 16) | ((size_t)t[7] << 24);
        if (isize > kMaxBlock) { R.err = "BGZF block larger than 64 KiB"; return SIZE_MAX; }
        if (utotal + isize > out_limit) break; // (a window's offsets are 32-bit, and the run's output buffer is what it is: the rest belongs to the next run)
        if (isize) {
            if (nb + 1 > R.hb.size()) { R.hb.resize(2 * nb + 1024); R.hc.resize(2 * nb + 1024); }
            R.hb[nb] = GiBlock{d_off + p + 12 + xlen, head + utotal, (uint32_t)(bsize - 12 - xlen - 8), (uint32_t)isize};
            R.hc[nb] = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
            ++nb;
        }
        utotal += isize;
        p += bsize;
    }
    return p;
}

bool GpuBamReader::Impl::wait_ready()
{
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return stop || dev_ready; });
    if (stop) return false;
    lk.unlock();
    return hipSetDevice(device) == hipSuccess;
}

// A reader of the chunk ring: takes the next chunk index, waits until its slot has been given back (and the card has copied what
// it held), reads the chunk's bytes of the file.
void GpuBamReader::Impl::reader_loop()
{
    bool dev_set = false;
    for (;;) {
        uint64_t i;
        {
            std::unique_lock<std::mutex> lk(rm);
            rcv.wait(lk, [&] { return rstop || ring_state < 0 || (ring_state > 0 && ring_open && next_claim < released + (uint64_t)kSlots && ring_base + next_claim * chunk_bytes < read_limit); });
            if (rstop || ring_state < 0) return;
            i = next_claim++;
        }
        Slot& S = slots[i % kSlots];
        if (S.used) { // (the copy of the chunk this slot held before)
            if (!dev_set) { (void)hipSetDevice(device); dev_set = true; }
            (void)hipEventSynchronize(S.done);
        }
        const uint64_t at = ring_base + i * chunk_bytes;
        size_t got = 0;
        const double t0 = now_s();
        while (got < chunk_bytes) {
            const ssize_t k = pread(fd, S.p + kHeadroom + got, chunk_bytes - got, (off_t)(at + got));
            if (k < 0) { if (errno == EINTR) continue; break; } // (a read error shows as a short chunk: "truncated BGZF file")
            if (k == 0) break;
            got += (size_t)k;
        }
        {
            std::lock_guard<std::mutex> lk(rm);
            t_read += now_s() - t0;
            S.len = got; S.index = i; S.filled = true;
        }
        rcv.notify_all();
    }
}

GpuBamReader::Impl::Slot* GpuBamReader::Impl::wait_chunk(uint64_t i)
{
    Slot& S = slots[i % kSlots];
    const double t0 = now_s();
    std::unique_lock<std::mutex> lk(rm);
    rcv.wait(lk, [&] { return rstop || ring_state < 0 || (S.filled && S.index == i); });
    t_wait_chunk = t_wait_chunk.load() + (now_s() - t0);
    return rstop || ring_state < 0 ? nullptr : &S;
}

// the producer is done with chunk i: its slot may be read into again once the copies queued on `st` so far have run
bool GpuBamReader::Impl::release_chunk(uint64_t i, hipStream_t st)
{
    Slot& S = slots[i % kSlots];
    if (hipEventRecord(S.done, st) != hipSuccess) return false;
    {
        std::lock_guard<std::mutex> lk(rm);
        S.used = true; S.filled = false;
        released = i + 1;
    }
    rcv.notify_all();
    return true;
}

// Reads the next run of whole BGZF blocks and starts its inflation.  The first run is read before the device is up (into pageable
// memory, by the reader threads' worth of plain threads, copied staged); the others come through the chunk ring.  A shard's run
// stops kBeyond bytes behind its end block, and no chunk is read that begins behind what a run can want.
void GpuBamReader::Impl::fill_run(GbRun& R)
{
    R.rc = 1; R.final = false; R.utotal = 0; R.err.clear();
    R.abs0 = u_produced;
    size_t nb = 0, utotal = 0, d_off = 0;
    hipError_t he = hipSuccess;
    const uint64_t stop_at = stop_off();
    bool stopped = false;
    // A run ends where its output buffer does (sized in open() for 3.6 x the compressed bytes: a file that inflates further gets
    // more, smaller runs — never a buffer released and allocated again behind a running kernel).
    const size_t out_limit = std::min(kMaxRunOut, out_cap - head - 64);
    if (produced == 0) {
        // the first run: at least 64 MB (the first batch is there when the device is), and whatever more can be read until the device is up
        static const size_t first_cap = getenv("BQC_GB_FIRST_MB") ? (size_t)std::max(64, atoi(getenv("BQC_GB_FIRST_MB"))) << 20 : (size_t)320 << 20; // (10 M-read file, 896 MB: 0.29 s with 640, 0.26 with 320, 0.27 with 192, 0.28 with 128 — tools/first_run_mb.py)
        const size_t piece = 32u << 20;
        size_t cap = std::min<size_t>(run_bytes, first_cap);
        if (stop_at != UINT64_MAX) cap = (size_t)std::min<uint64_t>(cap, stop_at + (1u << 17) - begin_off);
        first_raw.resize(cap + 64);
        const size_t n_pieces = (cap + piece - 1) / piece;
        std::atomic<size_t> next_piece{0};
        std::vector<size_t> got_of(n_pieces, 0);
        const double t0 = now_s();
        auto read_pieces = [&]() {
            for (;;) {
                const size_t k = next_piece.load();
                if (k >= n_pieces) return;
                if (k * piece >= (64u << 20)) { std::lock_guard<std::mutex> lk(m); if (dev_ready || stop) return; }
                size_t mine = k;
                if (!next_piece.compare_exchange_strong(mine, k + 1)) continue;
                const size_t want = std::min(piece, cap - k * piece);
                size_t g = 0;
                while (g < want) {
                    const ssize_t r = pread(fd, first_raw.data() + k * piece + g, want - g, (off_t)(begin_off + k * piece + g));
                    if (r < 0) { if (errno == EINTR) continue; break; }
                    if (r == 0) break;
                    g += (size_t)r;
                }
                got_of[k] = g;
                if (g < want) { next_piece = n_pieces; return; } // the file ends here
            }
        };
        {
            std::vector<std::thread> th;
            for (int t = 1; t < kReaders; ++t) th.emplace_back(read_pieces);
            read_pieces();
            for (std::thread& t : th) t.join();
        }
        size_t got = 0;
        for (size_t k = 0; k < n_pieces; ++k) { got += got_of[k]; if (got_of[k] < std::min(piece, cap - k * piece)) break; }
        t_read += now_s() - t0;
        if (got < cap) file_eof = true;
        const size_t p = parse_blocks(R, first_raw.data(), got, 0, nb, utotal, stop_at, out_limit, stopped);
        if (p == SIZE_MAX) { R.rc = -1; return; }
        if (file_eof && !stopped && p != got && utotal + 65536 <= out_limit) { R.err = "truncated BGZF file"; R.rc = -1; return; }
        parsed_off += p;
        file_eof = file_eof && p == got; // (what the first run has left over is read again, through the ring)
        { // the ring starts at the first byte no run has taken
            std::lock_guard<std::mutex> lk(rm);
            ring_base = parsed_off;
            ring_open = true;
            read_limit = stop_at == UINT64_MAX ? UINT64_MAX : stop_off() + (1u << 17);
        }
        rcv.notify_all();
        if (!wait_ready()) { R.rc = -2; return; }
        if (!R.d_comp.need(p + 64) || !R.d_out.need(head + utotal + 64)) { R.rc = -2; return; }
        he = hipMemsetAsync(R.d_status, 0, 4, R.s);
        if (he == hipSuccess && p) he = hipMemcpyAsync(R.d_comp.p, first_raw.data(), p, hipMemcpyHostToDevice, R.s);
        d_off = p;
    } else {
        if (!wait_ready()) { R.rc = -2; return; }
        // (no-ops for the two buffers open() has allocated; the third one is allocated here, whole: the chunks are copied in as they are read)
        if (!R.d_comp.need(run_bytes + chunk_bytes + (1u << 17) + 64, true) || !R.d_out.need(out_cap, true) || !R.d_blocks.need(run_bytes / 2048) || !R.d_crc.need(run_bytes / 2048) ||
            !R.d_tok.need(bqc_gpu_inflate_token_words(out_cap, run_bytes / 2048), true) || !R.d_ntok.need(run_bytes / 2048)) { R.rc = -2; return; }
        he = hipMemsetAsync(R.d_status, 0, 4, R.s);
        static const bool own_copy_stream = !(getenv("BQC_GB_COPY_STREAM") && getenv("BQC_GB_COPY_STREAM")[0] == '0');
        if (own_copy_stream && !cs && hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); cs = nullptr; }
        hipStream_t const copy_s = own_copy_stream && cs ? cs : R.s;
        if (stop_at != UINT64_MAX) { // the readers may go as far as this run can want
            { std::lock_guard<std::mutex> lk(rm); read_limit = std::max(read_limit, stop_at + (1u << 17)); }
            rcv.notify_all();
        }
        while (he == hipSuccess && !stopped && !file_eof && d_off < run_bytes && utotal + 65536 <= out_limit) {
            Slot* S = wait_chunk(ring_i);
            if (!S) { R.rc = -2; return; }
            const size_t have = kHeadroom + S->len - ring_at;
            const size_t p = parse_blocks(R, S->p + ring_at, have, d_off, nb, utotal, stop_at, out_limit, stopped);
            if (p == SIZE_MAX) { R.rc = -1; return; }
            if (!R.d_comp.need(d_off + p + 64) || !R.d_out.need(head + utotal + 64)) { R.rc = -2; return; } // (sized at open: grows only for unusual files)
            if (p) he = hipMemcpyAsync(R.d_comp.p + d_off, S->p + ring_at, p, hipMemcpyHostToDevice, copy_s);
            if (he != hipSuccess) break;
            d_off += p; parsed_off += p; ring_at += p;
            if (stopped || utotal + 65536 > out_limit) break; // (the rest of this chunk is the next run's)
            const size_t rem = kHeadroom + S->len - ring_at; // a block cut by the chunk's end (or nothing)
            if (S->len < chunk_bytes) { // the file ends in this chunk
                if (rem) { R.err = "truncated BGZF file"; R.rc = -1; return; }
                file_eof = true;
                break;
            }
            if (rem > kHeadroom) { R.err = "corrupt BGZF block (larger than 64 KiB)"; R.rc = -1; return; }
            Slot* N = wait_chunk(ring_i + 1);
            if (!N) { R.rc = -2; return; }
            if (rem) memcpy(N->p + kHeadroom - rem, S->p + ring_at, rem);
            if (!release_chunk(ring_i, copy_s)) { R.rc = -2; return; }
            ++ring_i;
            ring_at = kHeadroom - rem;
        }
    }
    if (he == hipSuccess && cs && produced != 0) { // (the run's chunks went through the copy stream: its kernels wait for the last of them)
        he = hipEventRecord(R.copied, cs);
        if (he == hipSuccess) he = hipStreamWaitEvent(R.s, R.copied, 0);
    }
    R.final = file_eof;
    R.utotal = utotal;
    u_produced += utotal;
    if (he == hipSuccess && (!R.d_blocks.need(nb + 1) || !R.d_crc.need(nb + 1) || !R.d_ntok.need(nb + 1) || !R.d_tok.need(bqc_gpu_inflate_token_words(utotal, nb)))) { R.rc = -2; return; }
    if (he == hipSuccess && nb) he = hipMemcpyAsync(R.d_blocks.p, R.hb.data(), nb * sizeof(GiBlock), hipMemcpyHostToDevice, R.s); // (pageable, small: staged at once)
    if (produced == 0) { // the caller says when the card may get busy: its own set-up (context, tables) behind a running 50 ms kernel was measured to take 0.2-0.65 s instead of 0.1
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return stop || kernels_ok; });
        if (stop) { R.rc = -2; return; }
    }
    if (he == hipSuccess && nb) he = hipMemcpyAsync(R.d_crc.p, R.hc.data(), nb * 4, hipMemcpyHostToDevice, R.s);
    if (he != hipSuccess) { R.rc = -2; return; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timing = this->timing && atoi(getenv("BQC_GB_TIMING")) >= 2; // (2: the kernels of every run, waited for)
    if (timing) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, R.s); }
    bqc_gpu_inflate_launch(R.d_comp.p, R.d_blocks.p, (uint32_t)nb, utotal, R.d_out.p, R.d_crc.p, R.d_status, R.d_tok.p, R.d_ntok.p, R.s);
    if (timing) (void)hipEventRecord(e1, R.s);
    if (hipEventRecord(R.ready, R.s) != hipSuccess) R.rc = -2;
    if (timing) { // (waits: only with BQC_GB_TIMING)
        float ms = 0;
        (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        fprintf(stderr, "[gpu reader] run of %zu blocks, %.1f MB -> %.1f MB: inflate + crc %.1f ms, done at %.3f\n", nb, d_off / 1e6, utotal / 1e6, ms, now_s());
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
}

int GpuBamReader::Impl::advance(std::string& err)
{
    if (stream_done) return 0;
    GbRun* N;
    {
        const double t0 = now_s();
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return produced > taken; });
        N = &runs[taken % kRuns];
        t_wait_run += now_s() - t0;
    }
    if (N->rc != 1) { err = N->err; stream_done = true; return N->rc; }
    const size_t left = end - cur;
    if (left > head) { err = "a record larger than the reader's spare room"; stream_done = true; return -3; }
    // behind the run's kernels, on the consumer's stream: what is left of the old window goes in front of the new run's bytes
    hipError_t he = hipStreamWaitEvent(s, N->ready, 0);
    if (he == hipSuccess && left) he = hipMemcpyAsync(N->d_out.p + head - left, win + cur, left, hipMemcpyDeviceToDevice, s);
    if (he == hipSuccess) he = hipMemcpyAsync(h_status, N->d_status, 4, hipMemcpyDeviceToHost, s);
    if (he != hipSuccess || !sync()) { stream_done = true; return -2; }
    if (*h_status & 15u) { err = "BGZF block failed to inflate (corrupt data)"; stream_done = true; return -1; }
    GbRun* old = cur_run;
    cur_run = N;
    win = N->d_out.p;
    win_abs0 = N->abs0;
    cur = head - left;
    end = head + N->utotal;
    if (N->final) stream_done = true; // (no run follows this one)
    {
        std::lock_guard<std::mutex> lk(m);
        ++taken;
        if (old) { old->state = 0; ++freed; } // the OLD buffer is free now; the new one stays the consumer's until the next advance
    }
    cv.notify_all();
    if (skip_u) { const size_t k = (size_t)std::min<uint64_t>(skip_u, end - cur); cur += k; skip_u -= k; }
    return 1;
}

int GpuBamReader::next_batch(HostBatch& o, size_t max_reads, size_t max_bases, std::string& err, int& err_code)
{
    o.clear();
    err_code = 0;
    Impl& I = *p_;
    if (hipSetDevice(I.device) != hipSuccess) { err = "GPU reader: device lost"; err_code = BQC_ERR_DEVICE; return -1; }
    if (!I.s) { // the consumer's stream (the third one the program needs: bqc_pool_stream)
        I.s = bqc_pool_stream(I.device, 2);
        if (!I.s || hipStreamSynchronize(I.ps) != hipSuccess) { err = "GPU reader: no stream"; err_code = BQC_ERR_DEVICE; return -1; } // (the status word's memset is on the producer's)
    }
    auto fail_dev = [&](const char* what) { err = std::string("GPU reader: ") + what; err_code = BQC_ERR_DEVICE; return -1; };
    auto unsupported = [&](const char* why) { err = std::string("GPU reader hands over to the host reader: ") + why; err_code = kUnsupported; return -1; };
    if (!I.main_set) {
        I.n_main = (uint32_t)main_.size();
        if (!I.d_main.need(main_.size() + 1)) return fail_dev("out of device memory");
        if (!main_.empty() && hipMemcpy(I.d_main.p, main_.data(), main_.size(), hipMemcpyHostToDevice) != hipSuccess) return fail_dev("copy failed");
        I.main_set = true;
    }
    const double t0 = now_s();
    double t_adv = 0, t_walk = 0, t_dec = 0;
    if (I.range_done) return 0;
    auto end_of_range = [&](uint64_t next_record_abs) { // the shard is over: where the successor's first record starts, relative to the end block
        const uint64_t mk = I.mark_abs.load();
        range_over_ = mk != UINT64_MAX && next_record_abs >= mk ? next_record_abs - mk : 0;
        I.range_done = true;
    };
    for (;;) {
        const double ta = now_s();
        if (I.cur == I.end || I.skip_u) { // nothing (left) in the window
            const bool was_done = I.stream_done;
            const int rc = I.advance(err);
            if (rc == 0) {
                if (I.skip_u && was_done) return unsupported("the file ends inside its header");
                if (I.need_locate) { range_first_ = 0; I.need_locate = false; } // (a shard without a byte: nothing starts here)
                end_of_range(I.abs_of(I.cur));
                return 0;
            }
            if (rc == -1) { err_code = BQC_ERR_IO; return -1; }
            if (rc == -2) return fail_dev("a failed copy or out of memory");
            if (rc == -3) return unsupported(err.c_str());
            t_adv += now_s() - ta;
            continue;
        }
        // a shard ends where its end block begins in the uncompressed stream (known once the producer has got there)
        uint64_t limit = UINT64_MAX;
        {
            const uint64_t mk = I.mark_abs.load(), here = I.abs_of(I.cur);
            if (mk != UINT64_MAX) limit = mk > here ? mk - here : 0;
        }
        if (limit == 0 && !I.need_locate) { end_of_range(I.abs_of(I.cur)); return 0; }
        // the walk covers what a batch is expected to need (a window may hold many batches); a record cut off by that is where the batch ends
        uint64_t avail = I.end - I.cur;
        {
            const double per_rec = I.avg_rec_bytes > 0 ? I.avg_rec_bytes : 400.0;
            double want = (double)std::min<size_t>(max_reads, 1u << 22) * per_rec;
            if (I.avg_rec_bases > 0) want = std::min(want, ((double)max_bases / I.avg_rec_bases + 1.0) * per_rec);
            want = want * 1.1 + (double)(4u << 20) + (double)I.grow_window;
            if ((double)avail > want) avail = (uint64_t)want;
        }
        const uint8_t* base = I.win + I.cur;
        // (segments that begin behind the shard's end hold nothing for it)
        const uint32_t nseg = (uint32_t)((std::min<uint64_t>(avail, limit == UINT64_MAX ? avail : limit + 1) + GB_SEG - 1) / GB_SEG);
        const size_t seg_cap = nseg; // (sized in open() for a batch's worth of the window: grows only for unusual records)
        if (!I.d_seg.need(seg_cap) || !I.h_seg.need(seg_cap) || !I.d_rec.need(seg_cap * GB_MAXR) || !I.d_base.need(seg_cap) || !I.h_base.need(seg_cap)) return fail_dev("out of device memory");
        static const bool walk_by_lanes = getenv("BQC_GB_WALK") && !strcmp(getenv("BQC_GB_WALK"), "lane"); // (the round-3 kernel: a lane per segment)
        if (walk_by_lanes) hipLaunchKernelGGL(k_gb_walk, dim3((nseg + 63) / 64), dim3(64), 0, I.s, base, avail, 0u, nseg, I.need_locate ? UINT64_MAX : (uint64_t)0, limit, I.n_ref, I.d_seg.p, I.d_rec.p);
        else hipLaunchKernelGGL(k_gb_walk_wave, dim3(nseg), dim3(64), 0, I.s, base, avail, 0u, nseg, I.need_locate ? UINT64_MAX : (uint64_t)0, limit, I.n_ref, I.d_seg.p, I.d_rec.p);
        if (hipMemcpyAsync(I.h_seg.p, I.d_seg.p, (size_t)nseg * sizeof(GbSeg), hipMemcpyDeviceToHost, I.s) != hipSuccess || !I.sync()) return fail_dev("walk failed");
        // the chain, segment by segment; whole segments are taken while the batch has room
        uint64_t pos = 0, n = 0, bases = 0, so = 0, qo = 0, co = 0;
        uint32_t last_taken = 0;
        bool over = false; // the chain has reached the shard's end
        if (I.need_locate) { // a shard in the middle of the file: its first record is the first guess of the walk (verified by the predecessor shard afterwards)
            uint32_t s0 = 0;
            while (s0 < nseg && (I.h_seg.p[s0].flags & GB_NO_START)) ++s0;
            if (s0 == nseg) {
                if (avail < I.end - I.cur && (limit == UINT64_MAX || avail < limit)) { I.grow_window += 2 * avail + (64u << 20); continue; }
                if (limit != UINT64_MAX && limit <= avail) { // no record starts in this shard: the predecessor's last one covers it
                    return unsupported("no record starts in this part of the file");
                }
                if (I.stream_done) return unsupported("no record starts in this part of the file");
                return unsupported("the first record of this part of the file was not found");
            }
            pos = I.h_seg.p[s0].first;
            static const bool skew = getenv("BQC_TEST_SHARD_SKEW") != nullptr; // tests: a wrong guess (the second record found), to exercise the fallback
            if (skew) {
                uint32_t bs = 0;
                if (hipMemcpy(&bs, base + pos, 4, hipMemcpyDeviceToHost) != hipSuccess) return fail_dev("copy failed");
                if (pos + 4 + (uint64_t)bs < avail && pos + 4 + (uint64_t)bs < limit) pos += 4 + (uint64_t)bs;
            }
            range_first_ = I.abs_of(I.cur + pos);
            I.need_locate = false;
            if (pos) { I.cur += pos; continue; } // (the window now starts at the record: walked again from there, as every other batch)
        }
        for (uint32_t s = 0; s < nseg; ++s) {
            GbBase& B = I.h_base.p[s];
            B = GbBase{so, qo, co, (uint32_t)n, 0};
            const uint64_t seg_end = std::min<uint64_t>(avail, (uint64_t)(s + 1) * GB_SEG);
            if (pos >= limit) { over = true; break; }
            if (pos >= seg_end) continue; // the previous record runs through this segment
            GbSeg& S = I.h_seg.p[s];
            if ((S.flags & GB_NO_START) || S.first != pos) { // the guess is not where the chain arrives (or there was none): this segment again, from there
                ++I.n_rewalk;
                if (walk_by_lanes) hipLaunchKernelGGL(k_gb_walk, dim3(1), dim3(64), 0, I.s, base, avail, s, 1u, pos, limit, I.n_ref, I.d_seg.p, I.d_rec.p);
                else hipLaunchKernelGGL(k_gb_walk_wave, dim3(1), dim3(64), 0, I.s, base, avail, s, 1u, pos, limit, I.n_ref, I.d_seg.p, I.d_rec.p);
                if (hipMemcpyAsync(&S, I.d_seg.p + s, sizeof(GbSeg), hipMemcpyDeviceToHost, I.s) != hipSuccess || !I.sync()) return fail_dev("walk failed");
                if (S.first != pos) return unsupported("the record walk could not be verified");
            }
            if (S.flags & GB_CORRUPT) return unsupported("a record the host reader will report");
            if (n && (n + S.count > max_reads || bases >= max_bases)) break;
            B.take = 1;
            last_taken = s + 1;
            n += S.count; bases += S.qual_bytes; so += S.seq_bytes; qo += S.qual_bytes; co += S.cigar_words;
            pos = S.exit;
            if (S.flags & GB_INCOMPLETE) break;
        }
        if (pos >= limit) over = true;
        if (n == 0 && over) { I.cur += pos; end_of_range(I.abs_of(I.cur)); return 0; }
        if (n == 0 && avail < I.end - I.cur) { I.grow_window += 2 * avail + (64u << 20); continue; } // (a record longer than the walked part of the window)
        I.grow_window = 0;
        if (n == 0) { // not one complete record in the window: the next run's bytes behind it
            if (I.stream_done) return unsupported("the file ends inside a record");
            const int rc = I.advance(err);
            if (rc == 0) return unsupported("the file ends inside a record");
            if (rc == -1) { err_code = BQC_ERR_IO; return -1; }
            if (rc == -2) return fail_dev("a failed copy or out of memory");
            if (rc == -3) return unsupported(err.c_str());
            continue;
        }
        if (n > 0xFFFFFFF0ull) return unsupported("batch too large");
        t_walk += now_s() - ta;
        const double td = now_s();
        // columns
        const size_t N = (size_t)n;
        // scratch columns of the copy kernel: [so qo co](8 B) [rec_off](4 B) per record
        const size_t Np = (N + 63) & ~(size_t)63, Np_cap = std::max(Np, (std::min<size_t>(max_reads, 1u << 22) + 63 + GB_MAXR) & ~(size_t)63);
        if (!I.d_cols.need(Np_cap * (3 * 8 + 4) + 256)) return fail_dev("out of device memory");
        // the batch's own buffer on the device: [seq][qual][cigar] (512 spare bytes behind each: the kernels' vector loads), then the fixed
        // columns [rid pos tlen nm as l_seq](4 B) [flag n_cigar](2 B) [mapq lane](1 B) and 8 bytes per read for the coverage anchors — a
        // batch that is anchored on the card (bqc_anchor_*) is submitted from here without its columns ever visiting the host
        const size_t o_seq = 512, o_qual = (o_seq + so + 512 + 255) & ~(size_t)255, o_cig = o_qual + ((qo + 512 + 255) & ~(size_t)255),
                     o_fix = (o_cig + 4 * co + 512 + 255) & ~(size_t)255, o_cov = o_fix + Np * (6 * 4 + 2 * 2 + 2), total = o_cov + 8 * Np + 256;
        if (o.dev_cap < total) {
            if (o.dev_mem) dev_free_hook(o.dev_mem);
            o.dev_mem = g_pool.take(total);
            o.dev_cap = o.dev_mem ? g_pool.cap : 0;
            if (!o.dev_mem) { // larger than the pool's buffers (or the pool is empty): its own allocation
                const size_t cap = total + total / 8 + 4096;
                if (hipMalloc(&o.dev_mem, cap) != hipSuccess) { o.dev_mem = nullptr; return fail_dev("out of device memory"); }
                o.dev_cap = cap;
            }
            o.dev_free = dev_free_hook;
        }
        uint8_t* pay = (uint8_t*)o.dev_mem;
        GbCols C;
        {
            uint8_t* q = I.d_cols.p;
            C.so = (uint64_t*)q; q += Np * 8; C.qo = (uint64_t*)q; q += Np * 8; C.co = (uint64_t*)q; q += Np * 8; C.rec_off = (uint32_t*)q;
            q = pay + o_fix;
            C.rid = (int32_t*)q; q += Np * 4; C.pos = (int32_t*)q; q += Np * 4; C.tlen = (int32_t*)q; q += Np * 4; C.nm = (int32_t*)q; q += Np * 4;
            C.as = (int32_t*)q; q += Np * 4; C.l_seq = (uint32_t*)q; q += Np * 4;
            C.flag = (uint16_t*)q; q += Np * 2; C.n_cigar = (uint16_t*)q; q += Np * 2;
            C.mapq = q; q += Np; C.lane = q;
        }
        GbLanes LN{I.d_lane_blob.p, I.d_lane_tab.p, I.d_lane_tab.p + I.n_lane_ids, I.d_lane_tab.p + 2 * (size_t)I.n_lane_ids, I.n_lane_ids, I.lane_count};
        hipError_t he = hipMemcpyAsync(I.d_base.p, I.h_base.p, (size_t)last_taken * sizeof(GbBase), hipMemcpyHostToDevice, I.s);
        if (he != hipSuccess) return fail_dev("copy failed");
        hipLaunchKernelGGL(k_gb_decode, dim3(last_taken), dim3(64), 0, I.s, base, I.d_seg.p, I.d_rec.p, I.d_base.p, C, LN, I.d_main.p, I.n_main, I.d_status);
        hipLaunchKernelGGL(k_gb_copy, dim3((uint32_t)((N + 15) / 16)), dim3(256), 0, I.s, base, C, (uint32_t)N, pay + o_seq, pay + o_qual, pay + o_cig);
        // The anchors of the coverage statistic on the card (k_anchor.hip), when the program has handed its context over and the stream
        // allows it: the fixed columns then stay here, and a summary comes back instead of 26 bytes per read
        bqc_batch dv;
        memset(&dv, 0, sizeof dv);
        dv.n_reads = (uint32_t)N; dv.flag = C.flag; dv.mapq = C.mapq; dv.lane = C.lane; dv.rid = C.rid; dv.pos = C.pos; dv.tlen = C.tlen; dv.nm = C.nm; dv.as = C.as;
        dv.l_seq = C.l_seq; dv.n_cigar = C.n_cigar; dv.seq = pay + o_seq; dv.qual = pay + o_qual; dv.cigar = (const uint32_t*)(pay + o_cig);
        bqc_anchored* ah = nullptr;
        bqc_ctx* const actx = anchor_ctx_.load();
        if (actx && anchors_ok_) {
            const int arc = bqc_anchor_enqueue(actx, &dv, pay + o_cov, I.s, &ah);
            if (arc < 0) return fail_dev(bqc_anchor_error(actx));
            if (arc > 0) { anchors_ok_ = false; ah = nullptr; } // (several read groups, a shard in the middle of the stream, or the host has kept the state so far)
        }
        auto columns_to_host = [&]() -> hipError_t {
            o.flag.resize(N); o.mapq.resize(N); o.lane.resize(N); o.rid.resize(N); o.pos.resize(N); o.tlen.resize(N);
            o.nm.resize(N); o.as.resize(N); o.l_seq.resize(N); o.n_cigar.resize(N);
            hipError_t e = hipMemcpyAsync(o.flag.data(), C.flag, N * 2, hipMemcpyDeviceToHost, I.s);
            if (e == hipSuccess) e = hipMemcpyAsync(o.n_cigar.data(), C.n_cigar, N * 2, hipMemcpyDeviceToHost, I.s);
            if (e == hipSuccess) e = hipMemcpyAsync(o.mapq.data(), C.mapq, N, hipMemcpyDeviceToHost, I.s);
            if (e == hipSuccess) e = hipMemcpyAsync(o.lane.data(), C.lane, N, hipMemcpyDeviceToHost, I.s);
            if (e == hipSuccess) e = hipMemcpyAsync(o.rid.data(), C.rid, N * 4, hipMemcpyDeviceToHost, I.s);
            if (e == hipSuccess) e = hipMemcpyAsync(o.pos.data(), C.pos, N * 4, hipMemcpyDeviceToHost, I.s);
            if (e == hipSuccess) e = hipMemcpyAsync(o.tlen.data(), C.tlen, N * 4, hipMemcpyDeviceToHost, I.s);
            if (e == hipSuccess) e = hipMemcpyAsync(o.nm.data(), C.nm, N * 4, hipMemcpyDeviceToHost, I.s);
            if (e == hipSuccess) e = hipMemcpyAsync(o.as.data(), C.as, N * 4, hipMemcpyDeviceToHost, I.s);
            if (e == hipSuccess) e = hipMemcpyAsync(o.l_seq.data(), C.l_seq, N * 4, hipMemcpyDeviceToHost, I.s);
            return e;
        };
        he = ah ? hipSuccess : columns_to_host();
        if (he == hipSuccess) he = hipMemcpyAsync(I.h_status, I.d_status, 4, hipMemcpyDeviceToHost, I.s);
        if (he != hipSuccess || !I.sync()) return fail_dev("decode failed");
        if (ah) {
            bqc_anchor_info info{};
            // a batch the host decoder takes (below), or one with more breaks than the card's chain walks: the host keeps the window
            // state from this batch on (it is current there: every anchored batch before this one is submitted before it)
            const int arc = *I.h_status ? 1 : bqc_anchor_complete(actx, ah, &info);
            if (*I.h_status) bqc_anchor_discard(actx, ah);
            if (arc < 0) return fail_dev(bqc_anchor_error(actx));
            if (arc > 0) {
                anchors_ok_ = false; ah = nullptr;
                if (!*I.h_status && (columns_to_host() != hipSuccess || !I.sync())) return fail_dev("decode failed");
            } else { o.anchored = ah; o.dev = dv; o.n_noqual = info.n_noqual; o.rid_min = info.rid_min; o.rid_max = info.rid_max; ++n_anchored_; }
        }
        if (*I.h_status) {
            // A record the card does not decode (a read group that is not in the header, a second NM tag, no RG tag, ...): THIS
            // batch is decoded by the host reader's rules — its bytes come back from the window, its records are listed by a serial
            // walk, bam_decode_records (host/bam_io.cpp) fills the batch's host columns and decides what is an error — and the run goes
            // on from the card with the next batch.
            if (hipMemsetAsync(I.d_status, 0, 4, I.s) != hipSuccess) return fail_dev("memset failed");
            try { I.handover_raw.resize((size_t)pos); } catch (const std::bad_alloc&) { return unsupported("no host memory for a batch handed over"); }
            if (hipMemcpy(I.handover_raw.data(), base, (size_t)pos, hipMemcpyDeviceToHost) != hipSuccess) return fail_dev("copy failed");
            std::vector<BamRec>& recs = I.handover_recs;
            recs.clear();
            recs.reserve(N);
            size_t hso = 0, hqo = 0, hco = 0;
            for (uint64_t q = 0; q + 36 <= pos && recs.size() < N;) {
                const uint8_t* r = I.handover_raw.data() + q;
                const uint32_t bs = r[0] | (r[1] << 8) | (r[2] << 16) | ((uint32_t)r[3] << 24);
                const uint32_t n_cig = r[16] | (r[17] << 8), l_seq = r[20] | (r[21] << 8) | (r[22] << 16) | ((uint32_t)r[23] << 24);
                recs.push_back(BamRec{(size_t)q, bs, l_seq, n_cig, hso, hqo, hco, nrec_ + recs.size()});
                hso += (l_seq + 1u) / 2u; hqo += l_seq; hco += n_cig;
                q += 4ull + bs;
            }
            if (recs.size() != N || hso != so || hqo != qo || hco != co) return unsupported("the record walk could not be verified");
            o.d_seq = o.d_qual = nullptr; o.d_cigar = nullptr; // (a host batch; its device buffer stays for the next one)
            const bool fine = bam_decode_records(I.handover_raw.data(), recs, hdr_, main_, bqc_host_threads(), o, err, err_code);
            ++n_handed_over_;
            if (hdr_.lane_names.size() != I.n_lane_ids && !I.upload_lanes(hdr_)) return fail_dev("out of device memory");
            I.cur += pos;
            if (over) end_of_range(I.abs_of(I.cur));
            nrec_ += n;
            I.avg_rec_bytes = (double)pos / (double)n;
            I.avg_rec_bases = (double)bases / (double)n;
            if (I.timing) fprintf(stderr, "[gpu reader] batch of %zu records handed over to the host decoder (%.1f ms)\n", N, (now_s() - t0) * 1e3);
            return fine ? 1 : -1;
        }
        o.d_seq = pay + o_seq; o.d_qual = pay + o_qual; o.d_cigar = (const uint32_t*)(pay + o_cig);
        I.cur += pos;
        if (over) end_of_range(I.abs_of(I.cur));
        nrec_ += n;
        I.avg_rec_bytes = (double)pos / (double)n;
        I.avg_rec_bases = (double)bases / (double)n;
        t_read_ = I.t_read; t_wait_run_ = I.t_wait_run; t_wait_chunk_ = I.t_wait_chunk;
        t_dec = now_s() - td;
        if (I.timing)
            fprintf(stderr, "[gpu reader] batch of %zu records (%.1f MB): %.1f ms = next run %.1f + walk %.1f + decode %.1f (reading so far %.3f s, waiting for runs %.3f s, rewalked %llu) at %.3f\n", N,
                    pos / 1e6, (now_s() - t0) * 1e3, t_adv * 1e3, t_walk * 1e3, t_dec * 1e3, I.t_read, I.t_wait_run, (unsigned long long)I.n_rewalk, now_s());
        return 1;
    }
}
