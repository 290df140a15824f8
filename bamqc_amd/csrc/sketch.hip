// sketch.hip — k-mer sketch on the GPU: ReadQualityHasher::operator()
// (reference src/ReadQualityHasher.hpp:30-68) feeding RepHash (src/kmerstream/RepHash.hpp:28-122,
// seeded as RepHash.cpp:4-17) and StreamCounter (src/kmerstream/StreamCounter.hpp:23-356).
//
// Thread per read walks the read in sequencing orientation (reverse strand: index L-1-j, complemented
// char), keeps the 128-bit cyclic-polynomial state of both strands in registers and emits
// hash = h.lo ^ ht.lo for every window of k consecutive valid bases.  StreamCounter state is a
// commutative monoid (StreamCounter::join :95-112), so per hash we do: sumCount (ballot), F2 table
// (exact counts, privatised in LDS as u32, 32768 bins), and the level-w 4-bit saturating counter as
// a u32 global counter that is only incremented while its value is < 15 (value = min(15, raw)).
// Levels whose 524288 counters are all saturated are skipped entirely (the reference's M[w] early-out,
// StreamCounter.hpp:81-83), which removes almost all global traffic on large inputs.
// Estimators F0 / f1 / F2 (doubles, log/pow) run on the host exactly as the reference's.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "kernels_common.h"
#include "sketch.h"
#include <ctime>

namespace {
struct u128 { uint64_t hi, lo; };

// ---- host: MT19937 seeding of the character table (mersennetwister.h:194-208,298-329; RepHash.cpp:4-17)
struct MT {
    uint32_t st[624];
    int next = 0, left = 0;
    explicit MT(uint32_t seed)
    {
        st[0] = seed;
        for (int i = 1; i < 624; ++i) st[i] = 1812433253u * (st[i - 1] ^ (st[i - 1] >> 30)) + (uint32_t)i;
    }
    // one generation of the recurrence x[i+624] = x[i+397] ^ A((x[i] & upper bit) | (x[i+1] & lower 31 bits)), all indices mod 624,
    // computed in place in index order (entries behind i are already the new generation, as the recurrence wants)
    void reload()
    {
        for (int i = 0; i < 624; ++i) {
            const uint32_t y = (st[i] & 0x80000000u) | (st[(i + 1) % 624] & 0x7fffffffu);
            st[i] = st[(i + 397) % 624] ^ (y >> 1) ^ ((st[(i + 1) % 624] & 1u) ? 0x9908b0dfu : 0u);
        }
        left = 624; next = 0;
    }
    uint32_t rand_int()
    {
        if (left == 0) reload();
        --left;
        uint32_t s1 = st[next++];
        s1 ^= (s1 >> 11);
        s1 ^= (s1 << 7) & 0x9d2c5680u;
        s1 ^= (s1 << 15) & 0xefc60000u;
        return s1 ^ (s1 >> 18);
    }
};

size_t round_up_pow2(size_t size) // smallest power of two >= size (size >= 1)
{
    size_t p = 1;
    while (p < size) p <<= 1;
    return p;
}

const unsigned char kTwin[32] = {0, 20, 2, 7, 4, 5, 6, 3, 8, 9, 10, 11, 12, 13, 14, 15,
                                 16, 17, 18, 19, 1, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31}; // RepHash.hpp:8-13
// nibble -> (char & 31) for "=ACMGRSVTWYHKDBN" and for the complemented character (reverse strand)
const unsigned char kIdxFwd[16] = {29, 1, 3, 13, 7, 18, 19, 22, 20, 23, 25, 8, 11, 4, 2, 14};
const unsigned char kIdxRc[16] = {29, 20, 7, 11, 3, 25, 19, 2, 1, 23, 18, 4, 13, 8, 22, 14};

struct PairParams { // one (q, k) sketch
    uint32_t k, q;
    int32_t q_thr;      // (signed char)(33 + q): base valid iff (signed char)(phred + 33) >= q_thr
    uint32_t f2_mask;   // F2size - 1
    uint32_t ctr_mask;  // size*16 - 1
    uint32_t levels;    // MAX_TABLE = 32 (k_sketch counts on it: a level index is five bits)
    uint32_t ctr_shift; // log2(ctr_per_level)
    uint64_t ctr_per_level;
};

struct DevSketch { // device pointers of one (lane, pair)
    uint32_t* counters; // [levels][ctr_per_level] raw counts (value = min(15, raw))
    uint64_t* f2;       // [F2size]
    uint64_t* misc;     // [0] sumCount, [1..32] M[w] = successful increments, [40] saturated-level mask
};
} // namespace

struct SketchDevice {
    bqc_sketch_options so{};
    std::vector<int32_t> ks;
    std::vector<uint32_t> qs;
    uint32_t n_lanes = 0, n_pairs = 0;
    size_t ctr_per_level = 0, f2size = 0;
    std::vector<PairParams> pp;
    std::vector<DevSketch> ds; // [lane][pair]
    DevSketch* d_ds = nullptr;
    PairParams* d_pp = nullptr;
    uint4* d_hv = nullptr; // [pair][2][32]: hvals and rotl_k(hvals) as (hi.x hi.y lo.x lo.y) -> stored as two u64 {hi, lo}
    uint8_t* d_idx = nullptr; // [2][16] nibble -> char&31, twin table [32]
    void* arena = nullptr;
    size_t arena_bytes = 0;
    std::vector<uint8_t> h_ctr; // finalize scratch
};

// ---------------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------------
// 128-bit rotation by one bit as four 32-bit funnel shifts (v_alignbit_b32) instead of 64-bit shifts and ORs
__device__ __forceinline__ void rotl1_128(uint64_t& hi, uint64_t& lo)
{
    const uint32_t w3 = (uint32_t)(hi >> 32), w2 = (uint32_t)hi, w1 = (uint32_t)(lo >> 32), w0 = (uint32_t)lo;
    const uint32_t n3 = __builtin_amdgcn_alignbit(w3, w2, 31), n2 = __builtin_amdgcn_alignbit(w2, w1, 31);
    const uint32_t n1 = __builtin_amdgcn_alignbit(w1, w0, 31), n0 = __builtin_amdgcn_alignbit(w0, w3, 31);
    hi = ((uint64_t)n3 << 32) | n2; lo = ((uint64_t)n1 << 32) | n0;
}
__device__ __forceinline__ void rotr1_128(uint64_t& hi, uint64_t& lo)
{
    const uint32_t w3 = (uint32_t)(hi >> 32), w2 = (uint32_t)hi, w1 = (uint32_t)(lo >> 32), w0 = (uint32_t)lo;
    const uint32_t n0 = __builtin_amdgcn_alignbit(w1, w0, 1), n1 = __builtin_amdgcn_alignbit(w2, w1, 1);
    const uint32_t n2 = __builtin_amdgcn_alignbit(w3, w2, 1), n3 = __builtin_amdgcn_alignbit(w0, w3, 1);
    hi = ((uint64_t)n3 << 32) | n2; lo = ((uint64_t)n1 << 32) | n0;
}
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x3 __attribute__((aligned(1))) u32x3_u;
typedef u32x4 __attribute__((aligned(1))) u32x4_u;

#define SK_F2 32768
#define SK_PT (SK_F2 + 256 + 64 + 16)                 // pair tables behind the small tables (16-byte aligned)
#define SK_PAIRS (2 * 17 * 16)                     // [2 strands][16 leaving nibbles + none][16 entering nibbles]
#define SK_WORDS (SK_PT + 2 * SK_PAIRS * 4)          // two tables of 16-byte entries
#define SK_THREADS 1024

__global__ __launch_bounds__(SK_THREADS) void k_sketch(DevBatch b, const DevSketch* __restrict__ dsk, const PairParams* __restrict__ pps,
                                                          const uint64_t* __restrict__ hv_all, const uint8_t* __restrict__ idx_tab,
                                                          uint32_t n_pairs, uint32_t per_block)
{
    extern __shared__ uint32_t lds[];
    uint32_t* f2 = lds;                                   // [SK_F2]
    uint64_t* hv = (uint64_t*)(lds + SK_F2);              // [2][32][2]: table, table rotated by k
    uint32_t* lm = lds + SK_F2 + 256;                     // [32] successful increments per level, [32] = sumCount
    uint8_t* ix = (uint8_t*)(lds + SK_F2 + 256 + 64);     // [2][16] nibble -> idx, [32] twin
    const uint32_t pair = blockIdx.y;
    const PairParams P = pps[pair];
    for (uint32_t i = threadIdx.x; i < SK_F2; i += blockDim.x) f2[i] = 0;
    for (uint32_t i = threadIdx.x; i < 128; i += blockDim.x) hv[i] = hv_all[pair * 128 + i];
    for (uint32_t i = threadIdx.x; i < 64; i += blockDim.x) lm[i] = 0;
    for (uint32_t i = threadIdx.x; i < 64; i += blockDim.x) ix[i] = idx_tab[i];
    block_sync();
    // RepHash::update(out, in) XORs two table values into each strand's state; with 16 x 16 nibble pairs per strand of the
    // read the two are one 16-byte LDS read:  pf[s][out][in] = rotl_k(hvals[out]) ^ hvals[in],
    // pt[s][out][in] = hvals[twin[out]] ^ rotl_k(hvals[twin[in]])   (s: reverse reads look bases up complemented).
    // Row out = 16 ("nothing leaves") holds hvals[in] and rotl_k(hvals[twin[in]]): with it the first k bases of a k-mer
    // take the same step as the rolling update.  For the forward strand that is RepHash::init literally (:85-97); for the
    // twin, init's sum_u rotl^u(x_u) equals k steps of  ht = rotr1(ht ^ rotl_k(x_u))  once all k bases are in, and no
    // hash is emitted before that.
    uint4* pf = (uint4*)(lds + SK_PT);
    uint4* pt = pf + SK_PAIRS;
    for (uint32_t i = threadIdx.x; i < SK_PAIRS; i += blockDim.x) {
        const uint8_t* nx = ix + (i >= 272u ? 16 : 0);
        const uint32_t o = (i % 272u) >> 4, c = nx[i & 15u], tc = ix[32 + c];
        uint64_t fh = hv[2 * c], fl = hv[2 * c + 1], th_ = hv[64 + 2 * tc], tl_ = hv[64 + 2 * tc + 1];
        if (o < 16u) {
            const uint32_t co = nx[o], tco = ix[32 + co];
            fh ^= hv[64 + 2 * co]; fl ^= hv[64 + 2 * co + 1];
            th_ ^= hv[2 * tco]; tl_ ^= hv[2 * tco + 1];
        }
        pf[i] = make_uint4((uint32_t)fl, (uint32_t)(fl >> 32), (uint32_t)fh, (uint32_t)(fh >> 32));
        pt[i] = make_uint4((uint32_t)tl_, (uint32_t)(tl_ >> 32), (uint32_t)th_, (uint32_t)(th_ >> 32));
    }
    block_sync();
    if (b.desc->fatal) return; // (a batch the pre-pass rejected: its lanes may be out of range)
    const uint32_t lo_r = blockIdx.x * per_block, hi_r = min(b.n_reads, lo_r + per_block); // positions of the processing order (reads grouped by read group)
    uint32_t blane = 0xFFFFFFFFu;
    uint32_t& s_lane = lm[63]; // (no static __shared__: it would mis-align the dynamic LDS base)
    uint32_t sat_mask = 0; // (32 levels)
    for (uint32_t base = lo_r; base < hi_r; base += blockDim.x) {
        const uint32_t kk = base + threadIdx.x;
        bool live = kk < hi_r;
        const uint32_t r = live ? (b.order ? b.order[kk] : kk) : 0;
        const uint32_t lane = live ? b.lane[r] : 0;
        if (threadIdx.x == 0) s_lane = lane;
        block_sync();
        const uint32_t fl = s_lane;
        if (fl != blane) { // block-uniform: flush the privatised F2 / M counters of the previous lane
            if (blane != 0xFFFFFFFFu) {
                const DevSketch D = dsk[blane * n_pairs + pair];
                for (uint32_t i = threadIdx.x; i < SK_F2; i += blockDim.x) { const uint32_t v = f2[i]; if (v) { gadd(D.f2 + i, v); f2[i] = 0; } }
                if (threadIdx.x < 33) { const uint32_t v = lm[threadIdx.x]; if (v) { gadd(D.misc + (threadIdx.x == 32 ? 0 : 1 + threadIdx.x), v); lm[threadIdx.x] = 0; } }
            }
            blane = fl;
            sat_mask = (uint32_t)dsk[blane * n_pairs + pair].misc[40];
            block_sync();
        }
        const DevSketch D = dsk[lane * n_pairs + pair];
        const bool mine = live && lane == blane; // reads of another lane inside a mixed block use global memory only
        uint32_t flag = live ? b.flag[r] : 0x900u;
        const uint32_t L = live ? b.l_seq[r] : 0;
        // RunBamStream only for !QCfail && !dup primary records (bamqualcheck.cpp:439-442); needs first/last like every
        // record that reaches that line; DEFINED: records without qualities are skipped
        const bool run = live && !(flag & 0x900) && !(flag & 0x600) && (flag & 0xC0) && !(flag & BQC_FLAG_NO_QUAL) && L >= P.k;
        if (run) {
            const bool rc = flag & 0x10;
            const uint8_t* __restrict__ sq = b.seq + b.seq_off[r];
            const uint8_t* __restrict__ ql = b.qual + b.qual_off[r];
            uint64_t hh = 0, hl = 0, th = 0, tl = 0;
            uint32_t t = 0; // number of consecutive valid bases ending at the current one
            uint32_t n_hash = 0;
            // The read is walked in sequencing order, 16 bases at a time from registers: one 16-byte load of qualities and
            // one 12-byte load of packed bases per chunk (a reverse read: the mirrored window, reversed in registers) instead
            // of three dependent byte loads per base.  hist[d] = the nibbles of the chunk d chunks ago, for the base that
            // leaves the k-mer (k <= 63: at most 4 chunks back).
            uint64_t hist[5] = {0, 0, 0, 0, 0};
            const uint32_t kq = P.k >> 4, kr = P.k & 15u; // position j - k lies kq chunks back (and one more), kr nibbles into it
            for (uint32_t cb = 0; cb < L; cb += 16) {
                const int32_t i_lo = rc ? (int32_t)L - 16 - (int32_t)cb : (int32_t)cb; // first BAM position of the chunk's window (may be < 0)
                const u32x4 qv = *(const __attribute__((address_space(1))) u32x4_u*)(uintptr_t)(ql + i_lo);
                const int32_t sb = i_lo >> 1;                                            // floor
                const u32x3 sv = *(const __attribute__((address_space(1))) u32x3_u*)(uintptr_t)(sq + sb);
                // qualities in sequencing order, position jj in byte jj (little-endian over q[0..3])
                uint64_t qlo, qhi; // positions 0..7 / 8..15, one byte each
                if (rc) {
                    qlo = (uint64_t)__builtin_bswap32(qv.w) | ((uint64_t)__builtin_bswap32(qv.z) << 32);
                    qhi = (uint64_t)__builtin_bswap32(qv.y) | ((uint64_t)__builtin_bswap32(qv.x) << 32);
                } else { qlo = (uint64_t)qv.x | ((uint64_t)qv.y << 32); qhi = (uint64_t)qv.z | ((uint64_t)qv.w << 32); }
                // nibbles in sequencing order, position jj in bits [63 - 4jj : 60 - 4jj]
                uint64_t ns;
                {
                    const uint64_t b0 = __builtin_bswap32(sv.x), b1 = __builtin_bswap32(sv.y), b2 = __builtin_bswap32(sv.z);
                    uint64_t w = (b0 << 32) | b1;                             // nibbles 2*sb .. 2*sb+15
                    if (i_lo & 1) w = (w << 4) | (b2 >> 28);                   // window starts at an odd nibble
                    if (rc) {                                                 // reverse the order of the 16 nibbles
                        w = __builtin_bswap64(w);
                        w = ((w & 0x0F0F0F0F0F0F0F0Full) << 4) | ((w >> 4) & 0x0F0F0F0F0F0F0F0Full);
                    }
                    ns = w;
                }
                hist[4] = hist[3]; hist[3] = hist[2]; hist[2] = hist[1]; hist[1] = hist[0]; hist[0] = ns;
                // nibbles of the positions k back: the 16-nibble window ending k positions before this chunk's end
                uint64_t outs;
                {
                    const uint64_t hi = kq == 0 ? hist[1] : kq == 1 ? hist[2] : kq == 2 ? hist[3] : hist[4];
                    const uint64_t lo = kq == 0 ? hist[0] : kq == 1 ? hist[1] : kq == 2 ? hist[2] : hist[3];
                    outs = kr == 0 ? lo : (hi << (64u - 4u * kr)) | (lo >> (4u * kr));
                }
                const uint32_t nb = min(16u, L - cb);
                // Which of the chunk's 16 positions restart the k-mer ('N', or (signed char)(phred + 33) below the threshold:
                // ReadQualityHasher.hpp:61-66) — for all 16 at once: the nibbles that are 15 by two AND-shifts, the qualities byte-wise in
                // four dwords — v = phred + 33 mod 256 without a carry between the bytes, valid iff bit 7 of v is clear and its low seven bits
                // reach the threshold (1..127; other thresholds: sixteen compares) — gathered into one 16-bit mask V (bit jj: position jj
                // takes part in a k-mer).
                uint32_t V;
                {
                    uint64_t y = ns & (ns >> 1);
                    y = y & (y >> 2) & 0x1111111111111111ull;              // position jj: bit 60 - 4 jj
                    auto gather4 = [](uint32_t w) {                          // flags at bits 0, 4, .., 28 -> bits 0..7
                        w = (w | (w >> 3)) & 0x03030303u;
                        w = (w | (w >> 6)) & 0x000F000Fu;
                        return (w | (w >> 12)) & 0xFFu;
                    };
                    const uint32_t n16 = gather4((uint32_t)y) | (gather4((uint32_t)(y >> 32)) << 8); // bit i: position 15 - i
                    uint32_t bad = __brev(n16) >> 16;
                    if (P.q_thr >= 1 && P.q_thr <= 127) {
                        const uint64_t add = 0x0101010101010101ull * (uint64_t)(0x80u - (uint32_t)P.q_thr);
                        auto bad8 = [&](uint64_t w) { // bit 8 j + 7: position j of the half is below the threshold
                            const uint64_t v = ((w & 0x7F7F7F7F7F7F7F7Full) + 0x2121212121212121ull) ^ (w & 0x8080808080808080ull);
                            const uint64_t tt = (v & 0x7F7F7F7F7F7F7F7Full) + add; // bit 7 of a byte: its low seven bits reach the threshold
                            return ~(tt & ~v) & 0x8080808080808080ull;
                        };
                        auto gather8 = [](uint32_t w) {                      // flags at bits 7, 15, 23, 31 -> bits 0..3
                            w = (w >> 7) & 0x01010101u;
                            w = (w | (w >> 7)) & 0x00030003u;
                            return (w | (w >> 14)) & 0xFu;
                        };
                        const uint64_t bl = bad8(qlo), bh = bad8(qhi);
                        bad |= gather8((uint32_t)bl) | (gather8((uint32_t)(bl >> 32)) << 4) | (gather8((uint32_t)bh) << 8) | (gather8((uint32_t)(bh >> 32)) << 12);
                    } else {
#pragma unroll
                        for (uint32_t jj = 0; jj < 16u; ++jj)
                            bad |= ((int32_t)(int8_t)(uint8_t)((uint32_t)((jj < 8 ? qlo : qhi) >> (8 * (jj & 7))) + 33u) < P.q_thr ? 1u : 0u) << jj;
                    }
                    V = ~bad & (nb >= 16u ? 0xFFFFu : (1u << nb) - 1u);
                }
                // What a position does follows from V and the count t the chunk is entered with, without walking the positions one after
                // the other (round 4).  T = number of consecutive valid positions ending at jj, t included where the run reaches the
                // chunk's start: the states are cleared where T == 1 (First), nothing leaves the k-mer while T <= k (Fill), a hash is
                // emitted where T >= k (Emit) — the serial form's  if (t == 0) clear;  out = t < k ? none : ..;  if (t < k && ++t < k) return.
                uint32_t First, Fill, Emit;
                {
                    auto low = [](uint32_t n) { return n >= 32u ? 0xFFFFFFFFu : (1u << n) - 1u; };
                    const uint32_t z = min((uint32_t)__ffs((int)~V) - 1u, 16u);      // first position that is not valid
                    const uint32_t A = low(z);                                       // the run from the chunk's start
                    const uint32_t Bm = V & ~A;
                    Emit = A & ~low(P.k - 1u > t ? P.k - 1u - t : 0u);
                    Fill = (A & low(P.k > t ? P.k - t : 0u)) | Bm;
                    if (P.k <= 16u) { // (runs inside the chunk can reach k)
                        uint32_t ge = V;
                        for (uint32_t sft = 1; sft < P.k; ++sft) ge &= V << sft;
                        Emit |= Bm & ge;
                        Fill &= ~(Bm & ge & (V << P.k));
                    }
                    First = V & ~((V << 1) | (t != 0u ? 1u : 0u));
                    t = z == 16u ? min(P.k, t + 16u) : min(P.k, (uint32_t)__clz((int)~(V << 16)));
                    Emit &= 0xFFFFu;
                }
                n_hash += (uint32_t)__popc(Emit);
                // The two 16-byte table entries a position needs depend on V, Fill and the bases only: they are requested three positions
                // ahead of the serial chain  h = rotl1(h) ^ a;  ht = rotr1(ht ^ b)  (round 4: each step waited for its own LDS round trip).
                const uint32_t rcbase = rc ? 272u : 0u;
                auto entry = [&](const uint32_t jj) __attribute__((always_inline)) {
                    const uint32_t nib = (uint32_t)(ns >> (60 - 4 * jj)) & 15u;
                    const uint32_t outn = (Fill >> jj) & 1u ? 16u : (uint32_t)(outs >> (60 - 4 * jj)) & 15u;
                    return rcbase + (outn << 4) + nib;
                };
                uint4 ea[4], eb[4]; // entries of positions jj .. jj + 3 at [jj & 3]
#pragma unroll
                for (uint32_t jj = 0; jj < 3u; ++jj) { const uint32_t pi = entry(jj); ea[jj] = pf[pi]; eb[jj] = pt[pi]; }
#pragma unroll
                for (uint32_t jj = 0; jj < 16u; ++jj) { // unrolled: the nibble / leaving-base extractions get constant shifts
                    if (jj + 3u < 16u) { const uint32_t pi = entry(jj + 3u); ea[(jj + 3u) & 3u] = pf[pi]; eb[(jj + 3u) & 3u] = pt[pi]; }
                    const uint4 a = ea[jj & 3u], bt = eb[jj & 3u];
                    if ((V >> jj) & 1u) {
                        if ((First >> jj) & 1u) { hh = hl = th = tl = 0; }
                        // RepHash::init built incrementally (:85-97) and RepHash::update(out, in) (:99-113) as one step: nothing
                        // leaves the k-mer while it is still filling
                        rotl1_128(hh, hl);
                        hl ^= (uint64_t)a.x | ((uint64_t)a.y << 32); hh ^= (uint64_t)a.z | ((uint64_t)a.w << 32);   // h = rotl1(h) ^ rotl_k(hvals[out]) ^ hvals[in]
                        tl ^= (uint64_t)bt.x | ((uint64_t)bt.y << 32); th ^= (uint64_t)bt.z | ((uint64_t)bt.w << 32); // ht = rotr1(ht ^ hvals[twin[out]] ^ rotl_k(hvals[twin[in]]))
                        rotr1_128(th, tl);
                        if ((Emit >> jj) & 1u) {
                            // ---- StreamCounter::operator()(hash), StreamCounter.hpp:68-93
                            const uint64_t hash = hl ^ tl;
                            if (mine) atomicAdd(&f2[(uint32_t)hash & P.f2_mask], 1u);
                            else gadd(D.f2 + ((uint32_t)hash & P.f2_mask), 1);
                            // bitScanForward (lsb.cpp:26-29), clamped to the last of the 32 levels: the lowest set bit of the hash's low word,
                            // 31 when that word is zero (__ffs(0) - 1 wraps to 0xFFFFFFFF)
                            const uint32_t w = min((uint32_t)__ffs((int)(uint32_t)hash) - 1u, 31u);
                            if (!(mine && ((sat_mask >> w) & 1u))) { // (else M[w] == size*countsPerLong*maxVal: every counter is 15)
                                const uint32_t index = (uint32_t)(hash >> (w + 1u)) & P.ctr_mask;
                                // fire and forget: the counter's value is min(15, raw); k_sketch_levels clamps the raw counts after every
                                // batch and finds the levels in which every counter has reached 15
                                gadd32(D.counters + ((w << P.ctr_shift) + index), 1u); // (32 levels x ctr_per_level counters: far below 2^32)
                            }
                        }
                    }
                }
            }
            if (mine) atomicAdd(&lm[32], n_hash); else if (n_hash) gadd(D.misc, n_hash);
        }
        block_sync();
    }
    if (blane != 0xFFFFFFFFu) {
        const DevSketch D = dsk[blane * n_pairs + pair];
        block_sync();
        for (uint32_t i = threadIdx.x; i < SK_F2; i += blockDim.x) { const uint32_t v = f2[i]; if (v) gadd(D.f2 + i, v); }
        if (threadIdx.x < 33) { const uint32_t v = lm[threadIdx.x]; if (v) gadd(D.misc + (threadIdx.x == 32 ? 0 : 1 + threadIdx.x), v); }
    }
}

// after every batch: clamp the raw counters to 15 and mark a level as saturated once every counter of it has reached 15
// (the reference's M[w] early-out, StreamCounter.hpp:81-83); saturated levels are not scanned again.
// Workgroup per (sketch, level, slice of 16 384 counters); misc[41] collects the levels that still have a counter below 15.
#define SK_SLICE 16384u
__global__ __launch_bounds__(256) void k_sketch_levels(const DevSketch* __restrict__ dsk, const PairParams* __restrict__ pps, uint32_t n_pairs)
{
    const DevSketch D = dsk[blockIdx.x];
    const PairParams P = pps[blockIdx.x % n_pairs];
    const uint32_t w = blockIdx.y;
    if (w >= P.levels || ((D.misc[40] >> w) & 1ull)) return;
    const uint32_t lo = blockIdx.z * SK_SLICE, hi = min(P.ctr_per_level, lo + SK_SLICE);
    uint4* ctr = (uint4*)(D.counters + (uint64_t)w * P.ctr_per_level); // (ctr_per_level is a power of two >= 8192)
    bool below = false;
    for (uint32_t i = lo / 4 + threadIdx.x; i < hi / 4; i += blockDim.x) {
        uint4 v = ctr[i];
        below |= v.x < 15u || v.y < 15u || v.z < 15u || v.w < 15u;
        if (v.x > 15u || v.y > 15u || v.z > 15u || v.w > 15u) {
            v.x = min(v.x, 15u); v.y = min(v.y, 15u); v.z = min(v.z, 15u); v.w = min(v.w, 15u);
            ctr[i] = v;
        }
    }
    if (__syncthreads_or(below) && threadIdx.x == 0) atomicOr((unsigned long long*)(D.misc + 41), 1ull << w);
}
__global__ void k_sketch_levels_commit(const DevSketch* __restrict__ dsk, const PairParams* __restrict__ pps, uint32_t n_pairs, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DevSketch D = dsk[i];
    const uint32_t levels = pps[i % n_pairs].levels;
    const uint64_t all = levels >= 64 ? ~0ull : (1ull << levels) - 1ull;
    D.misc[40] |= all & ~D.misc[41]; // no counter below 15 seen in this scan (levels already marked were not scanned: bit stays)
    D.misc[41] = 0;
}

// state vector <-> device tables: [sumCount][F2 table][counters as saturated bytes, 8 per word]
__global__ void k_sketch_export(DevSketch D, uint64_t* __restrict__ dst, uint64_t f2size, uint64_t n_ctr)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) dst[0] = D.misc[0];
    if (i < f2size) dst[1 + i] = D.f2[i];
    if (i < n_ctr / 8) {
        uint64_t w = 0;
        for (int k = 0; k < 8; ++k) { const uint32_t v = D.counters[i * 8 + k]; w |= (uint64_t)(v > 15u ? 15u : v) << (8 * k); }
        dst[1 + f2size + i] = w;
    }
}
__global__ void k_sketch_import(DevSketch D, const uint64_t* __restrict__ src, uint64_t f2size, uint64_t n_ctr, uint64_t ctr_per_level)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) D.misc[0] = src[0];
    if (i < f2size) D.f2[i] = src[1 + i];
    if (i < n_ctr / 8) {
        const uint64_t w = src[1 + f2size + i];
        for (int k = 0; k < 8; ++k) D.counters[i * 8 + k] = (uint32_t)((w >> (8 * k)) & 255u); // sums of per-rank min(15, .) ; clamped on use
    }
    if (i < 32) D.misc[1 + i] = 0; // successful-increment counts are not part of the vector: recomputed lazily (levels stay enabled)
    if (i == 0) { D.misc[40] = 0; D.misc[41] = 0; }
}

// ---------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------
SketchDevice* sketch_create(const bqc_sketch_options& so, uint32_t n_lanes, hipStream_t s, std::string& err)
{
    if (!(so.e > 0)) { err = "error rate must be > 0"; return nullptr; }
    auto* sk = new SketchDevice();
    sk->so = so;
    sk->ks.assign(so.klist, so.klist + so.n_k);
    sk->qs.assign(so.qlist, so.qlist + so.n_q);
    for (int k : sk->ks) if (k < 1 || k > 63) { err = "k must be in 1..63"; delete sk; return nullptr; }
    sk->n_lanes = n_lanes;
    sk->n_pairs = so.n_q * so.n_k;
    // StreamCounter ctor, StreamCounter.hpp:25-46
    size_t numcounts = (size_t)(48.0 / (so.e * so.e) + 1);
    sk->f2size = round_up_pow2((size_t)(2.0 / (so.e * so.e) + 1));
    if (numcounts < 8192) numcounts = 8192;
    size_t size = round_up_pow2((numcounts + 15) / 16);
    sk->ctr_per_level = size * 16;
    if (sk->f2size != SK_F2) { err = "only error rates with a 32768-entry F2 table (e.g. the default 0.01) are supported on the GPU"; delete sk; return nullptr; }
    // character tables
    std::vector<uint64_t> hv((size_t)sk->n_pairs * 128);
    // seed 0: RepHash::seed(0) takes (int)time(NULL) (RepHash.cpp:5-7; ReadQualityHasher.hpp:16-18 leaves the default-constructed,
    // time-seeded table in place) — the sketch lines then differ from run to run, as the reference's do
    MT mt((uint32_t)(so.seed != 0 ? so.seed : (int32_t)time(nullptr)));
    u128 base[32];
    for (int i = 0; i < 32; ++i) { // (randInt()<<32)|randInt(): g++ evaluates the left operand first (pinned in tests)
        uint64_t a = mt.rand_int(), b = mt.rand_int(), c = mt.rand_int(), d = mt.rand_int();
        base[i].hi = (a << 32) | b;
        base[i].lo = (c << 32) | d;
    }
    for (uint32_t qi = 0; qi < so.n_q; ++qi)
        for (uint32_t ki = 0; ki < so.n_k; ++ki) {
            const uint32_t p = qi * so.n_k + ki, k = (uint32_t)sk->ks[ki];
            PairParams P{};
            P.k = k; P.q = sk->qs[qi];
            P.q_thr = (int32_t)(int8_t)(uint8_t)(33u + P.q);
            P.f2_mask = (uint32_t)sk->f2size - 1; P.ctr_mask = (uint32_t)sk->ctr_per_level - 1; P.levels = 32;
            P.ctr_per_level = sk->ctr_per_level;
            P.ctr_shift = 0;
            while (((size_t)1 << P.ctr_shift) < sk->ctr_per_level) ++P.ctr_shift; // (a power of two: round_up_pow2 above)
            sk->pp.push_back(P);
            for (int i = 0; i < 32; ++i) {
                hv[p * 128 + 2 * i] = base[i].hi; hv[p * 128 + 2 * i + 1] = base[i].lo;
                const uint64_t h = base[i].hi, l = base[i].lo; // fastleftshiftk, RepHash.hpp:56-60
                hv[p * 128 + 64 + 2 * i] = (h << k) | (l >> (64 - k));
                hv[p * 128 + 64 + 2 * i + 1] = (l << k) | (h >> (64 - k));
            }
        }
    uint8_t idx[64];
    memcpy(idx, kIdxFwd, 16); memcpy(idx + 16, kIdxRc, 16); memcpy(idx + 32, kTwin, 32);
    // one arena: per (lane, pair): counters u32[32*ctr], f2 u64[f2size], misc u64[64]
    const size_t per = sk->ctr_per_level * 32 * 4 + sk->f2size * 8 + 64 * 8;
    const size_t n = (size_t)n_lanes * sk->n_pairs;
    sk->arena_bytes = per * n + hv.size() * 8 + 64 + sizeof(DevSketch) * n + sizeof(PairParams) * sk->n_pairs + 4096;
    if (hipMalloc(&sk->arena, sk->arena_bytes) != hipSuccess) { err = "hipMalloc failed for the sketch tables"; delete sk; return nullptr; }
    char* p = (char*)sk->arena;
    (void)hipMemsetAsync(sk->arena, 0, sk->arena_bytes, s);
    for (size_t i = 0; i < n; ++i) {
        DevSketch d;
        d.counters = (uint32_t*)p; p += sk->ctr_per_level * 32 * 4;
        d.f2 = (uint64_t*)p; p += sk->f2size * 8;
        d.misc = (uint64_t*)p; p += 64 * 8;
        sk->ds.push_back(d);
    }
    sk->d_hv = (uint4*)p; p += hv.size() * 8;
    sk->d_idx = (uint8_t*)p; p += 64;
    p = (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    sk->d_ds = (DevSketch*)p; p += sizeof(DevSketch) * n;
    p = (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    sk->d_pp = (PairParams*)p;
    (void)hipStreamSynchronize(s);
    bool ok = hipMemcpy(sk->d_hv, hv.data(), hv.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(sk->d_idx, idx, 64, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(sk->d_ds, sk->ds.data(), sizeof(DevSketch) * n, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(sk->d_pp, sk->pp.data(), sizeof(PairParams) * sk->n_pairs, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sketch), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   SK_WORDS * 4) == hipSuccess;
    if (!ok) { err = "sketch table upload failed"; sketch_destroy(sk); return nullptr; }
    return sk;
}

void sketch_destroy(SketchDevice* sk)
{
    if (!sk) return;
    (void)hipFree(sk->arena);
    delete sk;
}

void sketch_reset(SketchDevice* sk, hipStream_t s)
{
    const size_t per = sk->ctr_per_level * 32 * 4 + sk->f2size * 8 + 64 * 8;
    (void)hipMemsetAsync(sk->arena, 0, per * sk->n_lanes * sk->n_pairs, s);
}

void sketch_process(SketchDevice* sk, const DevBatch& b, hipStream_t s)
{
    if (b.n_reads == 0) return;
    uint32_t grid = 256;
    uint32_t per = (b.n_reads + grid - 1) / grid;
    per = ((per + SK_THREADS - 1) / SK_THREADS) * SK_THREADS;
    grid = (b.n_reads + per - 1) / per;
    hipLaunchKernelGGL(k_sketch, dim3(grid, sk->n_pairs), dim3(SK_THREADS), SK_WORDS * 4, s, b, sk->d_ds, sk->d_pp,
                       (const uint64_t*)sk->d_hv, sk->d_idx, sk->n_pairs, per);
    const uint32_t n = sk->n_lanes * sk->n_pairs;
    hipLaunchKernelGGL(k_sketch_levels, dim3(n, 32, (uint32_t)((sk->ctr_per_level + SK_SLICE - 1) / SK_SLICE)), dim3(256), 0, s, sk->d_ds, sk->d_pp, sk->n_pairs);
    hipLaunchKernelGGL(k_sketch_levels_commit, dim3((n + 63) / 64), dim3(64), 0, s, sk->d_ds, sk->d_pp, sk->n_pairs, n);
}

static uint64_t words_per_sketch(const SketchDevice* sk) { return 1 + sk->f2size + sk->ctr_per_level * 32 / 8; }
uint64_t sketch_state_words(const SketchDevice* sk) { return words_per_sketch(sk) * sk->n_lanes * sk->n_pairs; }

void sketch_state_export(SketchDevice* sk, uint64_t* dst, hipStream_t s)
{
    const uint64_t n_ctr = sk->ctr_per_level * 32, wps = words_per_sketch(sk);
    const uint32_t blocks = (uint32_t)((std::max<uint64_t>(n_ctr / 8, sk->f2size) + 255) / 256);
    for (size_t i = 0; i < sk->ds.size(); ++i)
        hipLaunchKernelGGL(k_sketch_export, dim3(blocks), dim3(256), 0, s, sk->ds[i], dst + i * wps, (uint64_t)sk->f2size, n_ctr);
}
void sketch_state_import(SketchDevice* sk, const uint64_t* src, hipStream_t s)
{
    const uint64_t n_ctr = sk->ctr_per_level * 32, wps = words_per_sketch(sk);
    const uint32_t blocks = (uint32_t)((std::max<uint64_t>(n_ctr / 8, sk->f2size) + 255) / 256);
    for (size_t i = 0; i < sk->ds.size(); ++i)
        hipLaunchKernelGGL(k_sketch_import, dim3(blocks), dim3(256), 0, s, sk->ds[i], src + i * wps, (uint64_t)sk->f2size, n_ctr,
                           (uint64_t)sk->ctr_per_level);
}

// per level: how many counters are 0, how many are 1 (all the estimators F0 / f1 need of the 16 M counters of a sketch)
__global__ __launch_bounds__(256) void k_sketch_zero_one(const uint32_t* __restrict__ counters, uint64_t R, unsigned long long* __restrict__ out /* [32][2] */)
{
    const uint32_t w = blockIdx.y;
    const uint32_t* t = counters + (uint64_t)w * R;
    uint32_t z = 0, o = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < R; j += (uint64_t)gridDim.x * blockDim.x) { const uint32_t v = t[j]; z += v == 0u; o += v == 1u; }
    for (int d = 32; d; d >>= 1) { z += __shfl_down(z, d, 64); o += __shfl_down(o, d, 64); }
    if ((threadIdx.x & 63u) == 0) { if (z) atomicAdd(out + 2 * w, (unsigned long long)z); if (o) atomicAdd(out + 2 * w + 1, (unsigned long long)o); }
}

bool sketch_finalize(SketchDevice* sk, uint32_t lane, std::vector<bqc_sketch_counts>& out, hipStream_t s, std::string& err)
{
    out.clear();
    const size_t R = sk->ctr_per_level;
    std::vector<uint64_t> f2(sk->f2size);
    unsigned long long* d_zo = nullptr;
    if (hipMalloc((void**)&d_zo, 64 * 8) != hipSuccess) { err = "device allocation failed"; return false; }
    struct Free { void* p; ~Free() { (void)hipFree(p); } } free_zo{d_zo};
    for (uint32_t p = 0; p < sk->n_pairs; ++p) {
        const DevSketch& D = sk->ds[(size_t)lane * sk->n_pairs + p];
        uint64_t sum_count = 0;
        unsigned long long zo[64];
        (void)hipMemsetAsync(d_zo, 0, sizeof zo, s);
        hipLaunchKernelGGL(k_sketch_zero_one, dim3(64, 32), dim3(256), 0, s, D.counters, (uint64_t)R, d_zo);
        (void)hipStreamSynchronize(s);
        if (hipMemcpy(zo, d_zo, sizeof zo, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(f2.data(), D.f2, sk->f2size * 8, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(&sum_count, D.misc, 8, hipMemcpyDeviceToHost) != hipSuccess) { err = "device copy failed"; return false; }
        bqc_sketch_counts c{};
        c.q = sk->pp[p].q; c.k = sk->pp[p].k; c.sumCount = sum_count;
        // per level: number of counters > 0, == 0, == 1 (value = min(15, raw))
        std::vector<size_t> nz(32, 0), r0(32, 0), r1(32, 0);
        for (size_t w = 0; w < 32; ++w) { r0[w] = (size_t)zo[2 * w]; r1[w] = (size_t)zo[2 * w + 1]; nz[w] = R - r0[w]; }
        { // F0, StreamCounter.hpp:114-140
            double sum = 0; int n = 0; double limit = 0.2;
            while (n == 0 && limit > 1e-8) {
                for (size_t i = 0; i < 32; i++) {
                    const size_t ts = nz[i];
                    if (ts <= (1 - limit) * R && ts >= limit * R) {
                        double est = (log(1.0 - ts / ((double)R)) / log(1.0 - 1.0 / R)) * pow(2.0, i + 1);
                        sum += est; n++;
                        break;
                    }
                }
                limit = limit / 1.5;
            }
            c.F0 = n ? (size_t)(sum / n) : 9223372036854775808ull; // (size_t)NaN as the reference's g++ build yields it
        }
        { // f1, :142-172
            double sum = 0; int n = 0; double limit = 0.2;
            while (n == 0 && limit > 1e-8) {
                for (size_t i = 0; i < 32; i++) {
                    if ((r0[i] <= (1 - limit) * R) && (r0[i] >= limit * R)) {
                        sum += (R - 1) * (r1[i] / ((double)r0[i])) * pow(2.0, i + 1);
                        n++;
                        break;
                    }
                }
                limit = limit / 1.5;
            }
            c.f1 = n ? (size_t)(sum / n) : 9223372036854775808ull;
        }
        { // F2, :308-317
            double sum = 0, sqsum = 0;
            for (size_t i = 0; i < sk->f2size; i++) { double v = (double)f2[i]; sum += v; sqsum += v * v; }
            c.F2 = (size_t)(sqsum + (sqsum - sum * sum) / sk->f2size);
        }
        out.push_back(c);
    }
    return true;
}
