// k_cov.hip — coverage depth histogram (OverallNumbers.hpp:59-135) and small utility kernels.
#include "kernels_common.h"

// ---------------------------------------------------------------------------------------------------
// k_cov — coverage depth histogram
// ---------------------------------------------------------------------------------------------------
// Virtual coordinates: the host runs the order-dependent anchor recurrence (OverallNumbers.hpp:84-110) and numbers every
// 1000-position window in flush order; k_prep_reads turns every read into its covered interval [win*1000 + off, + len), already
// truncated at (win+2)*1000 (cov[i], 8 B per read; len = 0 for reads that do not enter coverage()).  Depth is then order-free:
// +1 / -1 into an LDS difference array per interval, prefix scan, clamp-100 histogram.  One workgroup per tile of 4 windows; a
// tile names the range of reads whose first live window can touch it (windows never decrease along the stream of a read group);
// the entries of the NEXT tile are loaded (two per thread, coalesced) while the current tile is scanned.
#define KC_AT(p) ((p) + (((p) >> 6) << 2))
__global__ __launch_bounds__(256) void k_cov(DevBatch b, StateLayout sl, uint64_t* __restrict__ state,
                                                uint32_t* __restrict__ carry /* [lane][2][2000] */, uint32_t* parity /* [n_lanes], then a counter */,
                                                const uint8_t* __restrict__ lane_mask, uint8_t* __restrict__ started,
                                                const uint8_t* __restrict__ started_after, uint32_t n_lanes)
{
    // 256 threads x 16 positions >= BQC_COV_TILE + 1; position p lives at KC_AT(p): 4 words of padding after every 64, so that the
    // 16-byte accesses of the scan (thread t: words 16 t ..) of 16 neighbouring threads fall on 64 different banks
    __shared__ __attribute__((aligned(16))) int32_t diff[4096 + 256];
    __shared__ uint32_t hist[BQC_COVSIZE + 1];
    __shared__ uint32_t wsum[4];
    for (uint32_t i = threadIdx.x; i <= BQC_COVSIZE; i += blockDim.x) hist[i] = 0;
    for (uint32_t i = threadIdx.x; i < 4096 + 256; i += blockDim.x) diff[i] = 0;
    uint32_t cur_lane = 0xFFFFFFFFu;
    CovTile t{};
    t.lane = 0xFFFFFFFFu;
    CovEntry e0{0, 0}, e1{0, 0}; // len = 0: nothing
    auto load = [&](const CovTile& tt, uint32_t i) { // entry i of the tile's read range (another read group's: nothing)
        CovEntry e = b.cov[i];
        if (tt.mixed && b.lane[i] != tt.lane) e.off_len = 0;
        return e;
    };
    if (blockIdx.x < b.n_cov_tiles) {
        t = b.cov_tiles[blockIdx.x];
        const uint32_t i0 = t.list_begin + threadIdx.x, i1 = i0 + blockDim.x;
        if (i0 < t.list_end) e0 = load(t, i0);
        if (i1 < t.list_end) e1 = load(t, i1);
    }
    const uint32_t n_extra = b.desc->n_cov_extra;
    // persistent workgroups: the depth histogram stays in LDS across tiles and is flushed once per lane
    for (uint32_t ti = blockIdx.x;; ti += gridDim.x) {
        const bool done = ti >= b.n_cov_tiles;
        if (done) t.lane = 0xFFFFFFFFu;
        if (t.lane != cur_lane) { // block-uniform
            block_sync();
            if (cur_lane != 0xFFFFFFFFu)
                for (uint32_t i = threadIdx.x; i <= BQC_COVSIZE; i += blockDim.x)
                    if (hist[i]) { gadd(state + sl.lane_base(cur_lane) + sl.o_poscov + i, hist[i]); hist[i] = 0; }
            cur_lane = t.lane;
        }
        if (done) break;
        block_sync(); // diff is zero (initially / re-zeroed by the second pass of the previous tile)
        const int64_t lo = (int64_t)t.win_lo * BQC_VSIZE;
        auto add = [&](const CovEntry& e) {
            const uint32_t len = e.off_len >> 16;
            if (!len) return;
            int64_t a = (int64_t)e.win * BQC_VSIZE + (e.off_len & 0xFFFFu) - lo, z = a + len;
            if (a < 0) a = 0;
            if (z > BQC_COV_TILE) z = BQC_COV_TILE;
            if (a < z) { atomicAdd(&diff[KC_AT((uint32_t)a)], 1); atomicAdd(&diff[KC_AT((uint32_t)z)], -1); }
        };
        add(e0); add(e1);
        for (uint32_t e = t.list_begin + 2 * blockDim.x + threadIdx.x; e < t.list_end; e += blockDim.x) add(load(t, e)); // (rare: > 512 reads)
        for (uint32_t e = threadIdx.x; e < n_extra; e += blockDim.x) { // further intervals of reads with a clip between matches (rare)
            const CovExtra x = b.cov_extra[e];
            if (x.lane == t.lane) add(CovEntry{x.win, x.off_len});
        }
        const CovTile tc = t;
        { // next tile of this workgroup: descriptor and first intervals
            const uint32_t tn = ti + gridDim.x;
            e0 = CovEntry{0, 0}; e1 = CovEntry{0, 0};
            if (tn < b.n_cov_tiles) {
                t = b.cov_tiles[tn];
                const uint32_t i0 = t.list_begin + threadIdx.x, i1 = i0 + blockDim.x;
                if (i0 < t.list_end) e0 = load(t, i0);
                if (i1 < t.list_end) e1 = load(t, i1);
            }
        }
        block_sync();
        // block scan of diff: 16 consecutive entries per thread (4000 <= 256 * 16), read as 4 x int4
        const uint32_t s0 = threadIdx.x * 16u;
        int32_t d[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int4 v = *(const int4*)&diff[KC_AT(s0) + 4 * q];
            d[4 * q] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w;
        }
        int32_t loc = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) loc += (s0 + j < BQC_COV_TILE) ? d[j] : 0;
        const int32_t inc = (int32_t)wave_scan_incl((uint32_t)loc); // inclusive wave scan of the thread totals (wrapping arithmetic)
        if (lane_id() == WAVE - 1) wsum[threadIdx.x >> 6] = (uint32_t)inc;
        block_sync();
        // this thread's 16 entries are in registers: zero them for the next tile (its atomics come after the barrier on top)
#pragma unroll
        for (int q = 0; q < 4; ++q) *(int4*)&diff[KC_AT(s0) + 4 * q] = make_int4(0, 0, 0, 0);
        int32_t off = inc - loc;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) off += (int32_t)wsum[w];
        const uint32_t par = parity[tc.lane] & 1u; // flipped by the kernel's epilogue after every batch that owns tiles of this lane
        const uint32_t* cin = carry + ((uint64_t)tc.lane * 2 + par) * 2000;
        uint32_t* cout = carry + ((uint64_t)tc.lane * 2 + (par ^ 1u)) * 2000;
        int32_t run = off;
        uint32_t run_bin = 0xFFFFFFFFu, run_n = 0; // consecutive positions mostly share a depth: one LDS atomic per run
        if (lo >= 2 * BQC_VSIZE && tc.win_lo + BQC_COV_TILE_WINDOWS <= tc.win_final) {
            // (block-uniform) the common tile: nothing carried in, every window complete — no per-position window arithmetic
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (s0 + j >= BQC_COV_TILE) continue;
                run += d[j];
                const uint32_t bin = min((uint32_t)run, (uint32_t)BQC_COVSIZE);
                if (bin == run_bin) ++run_n;
                else { if (run_n) atomicAdd(&hist[run_bin], run_n); run_bin = bin; run_n = 1; }
            }
        } else
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t p = s0 + j;
            if (p >= BQC_COV_TILE) continue;
            run += d[j];
            const int64_t vp = lo + p;
            uint32_t depth = (uint32_t)run;
            if (vp < 2 * BQC_VSIZE) depth += cin[vp]; // partial windows carried over from the previous batch
            const uint32_t win = tc.win_lo + p / BQC_VSIZE;
            if (win < tc.win_final) { // update_coverage :66-77
                const uint32_t bin = depth > BQC_COVSIZE ? BQC_COVSIZE : depth;
                if (bin == run_bin) ++run_n;
                else { if (run_n) atomicAdd(&hist[run_bin], run_n); run_bin = bin; run_n = 1; }
            } else if (win < tc.win_final + 2) cout[(win - tc.win_final) * BQC_VSIZE + p % BQC_VSIZE] = depth;
        }
        if (run_n) atomicAdd(&hist[run_bin], run_n);
    }
    // Batch epilogue by the workgroup that finishes last (every other one has read its parities by then): the lanes that own
    // tiles of this batch switch to the other half of the carry array, and the lanes that have seen a coverage read so far
    // are remembered for the final flush — two tiny kernels less per batch.  (No fence: a workgroup's parity loads have returned
    // their values before it gets here, and what it wrote is for the next kernel.)
    block_sync();
    if (threadIdx.x == 0) wsum[0] = atomicAdd(parity + n_lanes, 1u) == gridDim.x - 1u ? 1u : 0u;
    block_sync();
    if (wsum[0]) {
        for (uint32_t l = threadIdx.x; l < n_lanes; l += blockDim.x) {
            if (lane_mask[l]) parity[l] ^= 1u;
            if (started_after[l]) started[l] = 1;
        }
        if (threadIdx.x == 0) parity[n_lanes] = 0u;
    }
}

// end of stream: histogram the two live windows of every started lane (bamqualcheck.cpp:447-453)
// (sel != nullptr: only the lanes with sel[lane] != 0 — the hand-over of a shard's set-aside reads, where the flush is the one a
// reset performs in the middle of the stream: count_start = 0)
__global__ __launch_bounds__(256) void k_cov_final(StateLayout sl, uint64_t* __restrict__ state, const uint32_t* __restrict__ carry,
                                                      const uint32_t* __restrict__ parity, const uint8_t* __restrict__ started, const uint8_t* __restrict__ sel,
                                                      uint32_t count_start)
{
    __shared__ uint32_t hist[BQC_COVSIZE + 1];
    const uint32_t lane = blockIdx.x;
    if (!started[lane] || (sel && !sel[lane])) return;
    for (uint32_t i = threadIdx.x; i <= BQC_COVSIZE; i += blockDim.x) hist[i] = 0;
    block_sync();
    const uint32_t* c = carry + ((uint64_t)lane * 2 + (parity[lane] & 1u)) * 2000;
    for (uint32_t i = threadIdx.x; i < 2000; i += blockDim.x) {
        const uint32_t d = c[i];
        atomicAdd(&hist[d > BQC_COVSIZE ? BQC_COVSIZE : d], 1u);
    }
    block_sync();
    for (uint32_t i = threadIdx.x; i <= BQC_COVSIZE; i += blockDim.x)
        if (hist[i]) gadd(state + sl.lane_base(lane) + sl.o_poscov + i, hist[i]);
    if (threadIdx.x == 0 && count_start) gadd(state + sl.lane_base(lane) + sl.o_covstart, 1);
}

__global__ void k_or_bytes(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && src[i]) dst[i] = 1;
}

__global__ void k_add_words(uint64_t* __restrict__ state, const uint64_t* __restrict__ idx, const uint64_t* __restrict__ val, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) gadd(state + idx[i], val[i]);
}

// the stream's error record takes over the first batch record that holds an error (batches run in order on one stream)
__global__ void k_err_merge(ErrRec* __restrict__ dst, const ErrRec* __restrict__ src)
{
    if (dst->first_key == BQC_ERRKEY_NONE && dst->flags == 0 && (src->first_key != BQC_ERRKEY_NONE || src->flags != 0)) *dst = *src;
}
extern "C" void bqc_launch_err_merge(ErrRec* dst, const ErrRec* src, hipStream_t s) { hipLaunchKernelGGL(k_err_merge, dim3(1), dim3(1), 0, s, dst, src); }

extern "C" void bqc_launch_cov(const DevBatch& b, const StateLayout& sl, uint64_t* state, uint32_t* carry, uint32_t* parity, const uint8_t* lane_mask,
                               uint8_t* started, const uint8_t* started_after, uint32_t n_lanes, hipStream_t s)
{
    if (b.n_cov_tiles == 0) return;
    const uint32_t grid = b.n_cov_tiles < 2048u ? b.n_cov_tiles : 2048u; // persistent workgroups, ~8 per CU
    hipLaunchKernelGGL(k_cov, dim3(grid), dim3(256), 0, s, b, sl, state, carry, parity, lane_mask, started, started_after, n_lanes);
}

extern "C" void bqc_launch_cov_final(const StateLayout& sl, uint64_t* state, const uint32_t* carry, const uint32_t* parity,
                                     const uint8_t* started, const uint8_t* sel, uint32_t count_start, hipStream_t s)
{
    hipLaunchKernelGGL(k_cov_final, dim3(sl.n_lanes), dim3(256), 0, s, sl, state, carry, parity, started, sel, count_start);
}

extern "C" void bqc_launch_add_words(uint64_t* state, const uint64_t* idx, const uint64_t* val, uint32_t n, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_add_words, dim3((n + 255) / 256), dim3(256), 0, s, state, idx, val, n);
}

extern "C" void bqc_launch_or_bytes(uint8_t* dst, const uint8_t* src, uint32_t n, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_or_bytes, dim3((n + 255) / 256), dim3(256), 0, s, dst, src, n);
}

// ---- profiling aid: streams `n_dwords` with 4-byte-per-lane loads (and 8-byte-per-lane in k_calib_read8) so that
// the rocprofv3 FETCH_SIZE counter can be calibrated on a known byte count (MI355X_MICROARCH.md, HBM section)
__global__ void k_calib_read4(const uint32_t* __restrict__ p, uint64_t n_dwords, uint32_t* __restrict__ out)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_dwords; i += (uint64_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_calib_read8(const uint2* __restrict__ p, uint64_t n, uint32_t* __restrict__ out)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { const uint2 v = p[i]; acc ^= v.x ^ v.y; }
    if (acc == 0x12345678u) out[0] = acc;
}
extern "C" int bqc_calib_read4(uint64_t bytes, int repeat)
{
    uint32_t* p = nullptr;
    uint32_t* o = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&o, 64) != hipSuccess) return 2;
    (void)hipMemset(p, 1, bytes);
    for (int r = 0; r < repeat; ++r) hipLaunchKernelGGL(k_calib_read4, dim3(256 * 8), dim3(256), 0, 0, p, bytes / 4, o);
    for (int r = 0; r < repeat; ++r) hipLaunchKernelGGL(k_calib_read8, dim3(256 * 8), dim3(256), 0, 0, (const uint2*)p, bytes / 8, o);
    (void)hipDeviceSynchronize();
    (void)hipFree(p); (void)hipFree(o);
    return 0;
}
