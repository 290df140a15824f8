// k_trip.hip — triplet counting (reference src/TripletCounting.hpp:195-236) for the few eligible short reads whose
// CIGAR has more than one operation (indels): the fast path (k_short.hip) assumes chromPos = pos + i.
// Wave per read, lane per base; the CIGAR walk is evaluated per lane in closed form (first operation assumed
// match-like, :203).  Small workgroups (4 waves, 4 KiB LDS) so that many are resident and hide the gather latency.
#include "kernels_common.h"

__global__ __launch_bounds__(256) void k_trip_list(DevBatch b, StateLayout sl, uint64_t* __restrict__ state, DevRefs refs)
{
    __shared__ uint32_t trip[1024];
    for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) trip[i] = 0;
    __syncthreads();
    const uint32_t ln = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    uint32_t cur_lane = 0xFFFFFFFFu;
    for (uint32_t ci = blockIdx.x;; ci += gridDim.x) {
        const bool done = ci >= b.n_trip_chunks;
        Chunk ch{0, 0, 0xFFFFFFFFu, 0, 0, 0, 0, 0};
        if (!done) ch = b.trip_chunks[ci];
        if (ch.lane != cur_lane) { // block-uniform: flush
            __syncthreads();
            if (cur_lane != 0xFFFFFFFFu)
                for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) {
                    const uint32_t v = trip[i];
                    if (v) { gadd(state + sl.lane_base(cur_lane) + sl.o_triplet + i, v); trip[i] = 0; }
                }
            __syncthreads();
            cur_lane = ch.lane;
        }
        if (done) break;
        for (uint32_t k = wave; k < ch.count; k += nwaves) {
            const uint32_t r = b.trip_list[ch.first + k];
            const uint32_t flag = b.flag[r], L = b.l_seq[r], ncig = b.n_cigar[r];
            const int32_t rid = b.rid[r];
            if (!(flag & BQC_FLAG_TRIPLET) || (flag & BQC_FLAG_NO_QUAL) || L < 3 || ncig == 0 || rid < 0 || (uint32_t)rid >= refs.n_refs ||
                refs.ref[rid] == nullptr)
                continue;
            const uint8_t* __restrict__ sq = b.seq + b.seq_off[r];
            const uint8_t* __restrict__ ql = b.qual + b.qual_off[r];
            const uint32_t* __restrict__ cg = b.cigar + b.cigar_off[r];
            const uint8_t* __restrict__ ref = refs.ref[rid];
            const int64_t reflen = (int64_t)refs.len[rid], pos = b.pos[r];
            const uint32_t grp = ((flag & 0x10u) ? 2u : 0u) + ((flag & 0x40u) ? 0u : 1u);
            const uint32_t n0 = cg[0] >> 4;
            for (uint32_t t0 = 0; t0 < L; t0 += 64) {
                const uint32_t i = t0 + ln;
                if (i < 1 || i + 1 >= L) continue;
                const uint32_t q = ql[i];
                if (q < 20u || q > 94u) continue; // (signed char)(q+33) >= '5'
                const uint32_t b0 = sq[(i - 1) >> 1], b1 = sq[i >> 1], b2 = sq[(i + 1) >> 1];
                const uint32_t np = ((i - 1) & 1u) ? (b0 & 15u) : (b0 >> 4), nb = (i & 1u) ? (b1 & 15u) : (b1 >> 4),
                               nn = ((i + 1) & 1u) ? (b2 & 15u) : (b2 >> 4);
                const uint32_t base = lut5(LUT5_FWD, nb);
                if (base == 4u || np == 15u || nn == 15u) continue;
                int64_t cp = -1;
                bool found = false;
                if (n0 == 0 || i < n0) { cp = pos + (int64_t)i; found = true; }
                if (n0 != 0) {
                    uint64_t rp = n0;
                    int64_t c = pos + (int64_t)n0;
                    for (uint32_t kk = 1; kk < ncig && rp <= i; ++kk) {
                        const uint32_t w = cg[kk], op = w & 15u, n = w >> 4;
                        if (op == 2u || op == 3u || op == 5u || op == 6u) c += n;   // D N H P
                        else if (op == 4u || op == 1u) rp += n;                        // S I
                        else {                                                         // M = X
                            if ((uint64_t)i >= rp && (uint64_t)i < rp + n) { cp = c + (int64_t)((uint64_t)i - rp); found = true; }
                            rp += n; c += n;
                        }
                    }
                }
                if (!found || cp < 1 || cp + 1 >= reflen) continue;
                const uint32_t r0 = ref[cp - 1] & 3u, r1 = ref[cp] & 3u, r2 = ref[cp + 1] & 3u; // Dna5 -> Dna: N -> A
                if ((lut5(LUT5_FWD, np) & 3u) == r0 && (lut5(LUT5_FWD, nn) & 3u) == r2)
                    atomicAdd(&trip[((r0 << 4) | (r1 << 2) | r2) * 16 + grp * 4 + base], 1u);
            }
        }
    }
}

extern "C" void bqc_launch_trip_list(const DevBatch& b, const StateLayout& sl, uint64_t* state, const DevRefs& refs, uint32_t n_cu, hipStream_t s)
{
    if (b.n_trip_chunks == 0) return;
    const uint32_t grid = b.n_trip_chunks < n_cu * 8 ? b.n_trip_chunks : n_cu * 8;
    hipLaunchKernelGGL(k_trip_list, dim3(grid), dim3(256), 0, s, b, sl, state, refs);
}
