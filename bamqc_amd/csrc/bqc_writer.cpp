// bqc_writer.cpp — `.bamqc` text emission (drop-in for writeOutput, reference
// src/bamqualcheck.cpp:156-233; printString :130-139; ten_most_abundant_kmers
// OverallNumbers.hpp:170-216; avgQualPerPos QualityCheck.hpp:273-279; writeTripletCounts
// TripletCounting.hpp:271-301).  Host-only code, part of libbamqc_gpu.so.
#include <algorithm>
#include <charconv>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bamqc.h"

namespace {
struct Out {
    std::string s;
    void str(const char* p) { s.append(p); }
    void u64(uint64_t v)
    {
        char b[24];
        auto r = std::to_chars(b, b + sizeof b, v);
        s.append(b, r.ptr);
    }
    void line_u64(const char* key, uint64_t v) { str(key); s.push_back(' '); u64(v); s.push_back('\n'); }
    void array(const char* key, const uint64_t* v, size_t n) // `key` + " " + value ... ; empty array prints the bare key
    {
        str(key);
        for (size_t i = 0; i < n; ++i) { s.push_back(' '); u64(v[i]); }
        s.push_back('\n');
    }
    void dbl(double d) // default ostream formatting: 6 significant digits, %g style
    {
        char b[40];
        int n = snprintf(b, sizeof b, "%g", d);
        s.append(b, (size_t)n);
    }
};

// Ten largest counts, then for each the lowest index holding it that has not been used yet.
// The reference selects with std::greater<int> (values truncated to int, OverallNumbers.hpp:177).
void top_kmers(Out& o, const uint64_t* em)
{
    std::vector<uint64_t> v(em, em + BQC_N_8MER);
    std::partial_sort(v.begin(), v.begin() + 10, v.end(), [](uint64_t a, uint64_t b) { return (int)a > (int)b; });
    uint64_t top[10];
    std::copy(v.begin(), v.begin() + 10, top);
    std::sort(top, top + 10, [](uint64_t a, uint64_t b) { return a > b; });
    int used[10];
    for (int i = 0; i < 10; ++i) {
        int pos = 0;
        for (; pos < BQC_N_8MER; ++pos) {
            if (em[pos] != top[i]) continue;
            bool seen = false;
            for (int k = 0; k < i; ++k) seen |= used[k] == pos;
            if (!seen) break;
        }
        used[i] = pos;
        char kmer[9];
        for (int k = 0; k < 8; ++k) kmer[k] = "ACGT"[(pos >> (2 * (7 - k))) & 3];
        kmer[8] = 0;
        o.str("nr_"); o.u64((uint64_t)i + 1); o.str("_most_abundant_8mer "); o.str(kmer); o.s.push_back(' '); o.u64(top[i]); o.s.push_back('\n');
    }
}

void avgqual(Out& o, const char* key, const bqc_mate_counts& m)
{
    o.str(key);
    const double nr = (double)(uint32_t)m.qualcount_readnr;
    for (uint32_t i = 0; i < m.n_cycles; ++i) { o.s.push_back(' '); o.dbl((double)m.qualcount[i] / nr); }
    o.s.push_back('\n');
}
} // namespace

extern "C" int bqc_write_bamqc(const bqc_counts* counts, const bqc_header_info* hdr, const char* path)
{
    if (!counts || !hdr || !path) return BQC_ERR_ARG;
    Out o;
    o.s.reserve(1 << 20);
    for (uint32_t n = 0; n < hdr->n_names; ++n) {
        if (hdr->lane_index[n] >= counts->n_lanes) return BQC_ERR_ARG;
        const bqc_lane_counts& L = counts->lanes[hdr->lane_index[n]];
        const bqc_mate_counts &a = L.mate[0], &b = L.mate[1];
        o.str("sample_id "); o.str(hdr->sample_id ? hdr->sample_id : ""); o.s.push_back('\n');
        o.str("lane "); o.str(hdr->lane_names[n]); o.s.push_back('\n');
        o.line_u64("total_read_pairs", (uint32_t)L.scalars[BQC_S_READCOUNT] / 2);
        o.line_u64("total_bps", L.scalars[BQC_S_TOTALBPS]);
        o.line_u64("supplementary_alignments", L.scalars[BQC_S_SUPPLEMENTARY]);
        o.line_u64("marked_duplicate", L.scalars[BQC_S_DUPLICATES]);
        o.line_u64("QC_failed", L.scalars[BQC_S_QCFAILED]);
        o.line_u64("not_primary_alignment", L.scalars[BQC_S_NOT_PRIMARY]);
        o.line_u64("both_reads_unmapped", L.scalars[BQC_S_BOTHUNMAPPED]);
        o.line_u64("first_read_unmapped", L.scalars[BQC_S_FIRSTUNMAPPED]);
        o.line_u64("second_read_unmapped", L.scalars[BQC_S_SECONDUNMAPPED]);
        o.line_u64("first_and_or_second_read_mapped", L.scalars[BQC_S_FIRST_AND_OR_SECOND_MAPPED]);
        o.line_u64("FF_RR_oriented_pairs", L.scalars[BQC_S_FF_RR]);
        o.line_u64("total_proper_pairs", L.scalars[BQC_S_PROPERPAIR]);
        o.line_u64("total_proper_pairs_autosome", L.scalars[BQC_S_AUTO_PROPERPAIR]);
        o.array("genome_coverage_histogram", L.poscov, BQC_COVSIZE + 1);
        o.array("insert_size_histogram", a.insertSize, a.n_insertSize);
        struct { const char* key; const uint64_t* bqc_mate_counts::*arr; uint32_t bqc_mate_counts::*len; } hists[] = {
            {"read_length_histogram", &bqc_mate_counts::readLength, &bqc_mate_counts::n_readLength},
            {"N_count_histogram", &bqc_mate_counts::Ncount, &bqc_mate_counts::n_Ncount},
            {"GC_content_histogram", &bqc_mate_counts::GCcount, &bqc_mate_counts::n_GCcount},
            {"average_base_qual_histogram", &bqc_mate_counts::averageQual, &bqc_mate_counts::n_averageQual},
            {"mapping_qual_histogram", &bqc_mate_counts::mapQ, &bqc_mate_counts::n_mapQ},
            {"mismatch_count_histogram", &bqc_mate_counts::mismatch, &bqc_mate_counts::n_mismatch},
            {"deletion_count_histogram", &bqc_mate_counts::delhist, &bqc_mate_counts::n_delhist},
            {"insertion_count_histogram", &bqc_mate_counts::inshist, &bqc_mate_counts::n_inshist},
        };
        for (auto& h : hists) {
            std::string k1 = std::string(h.key) + "_first", k2 = std::string(h.key) + "_second";
            o.array(k1.c_str(), a.*(h.arr), a.*(h.len));
            o.array(k2.c_str(), b.*(h.arr), b.*(h.len));
        }
        static const int order[5] = {4, 0, 1, 2, 3}; // N A C G T
        static const char* bn[5] = {"Ns", "As", "Cs", "Gs", "Ts"};
        for (int k = 0; k < 5; ++k) {
            std::string k1 = std::string(bn[k]) + "_by_position_first", k2 = std::string(bn[k]) + "_by_position_second";
            o.array(k1.c_str(), a.dnacount[order[k]], a.n_cycles);
            o.array(k2.c_str(), b.dnacount[order[k]], b.n_cycles);
        }
        avgqual(o, "average_base_qual_by_position_first", a);
        avgqual(o, "average_base_qual_by_position_second", b);
        o.array("soft_clipping_5_prime_by_position_first", a.sc5, a.n_cycles);
        o.array("soft_clipping_3_prime_by_position_first", a.sc3, a.n_cycles);
        o.array("soft_clipping_5_prime_by_position_second", b.sc5, b.n_cycles);
        o.array("soft_clipping_3_prime_by_position_second", b.sc3, b.n_cycles);
        top_kmers(o, L.eightmer);
        o.array("8mer_count", L.eightmer, BQC_N_8MER);
        for (uint32_t s = 0; s < L.n_sketch; ++s) { // bamqualcheck.cpp:221-230
            const bqc_sketch_counts& k = L.sketch[s];
            std::string ks = std::to_string(k.k), qs = std::to_string(k.q);
            o.line_u64((ks + "mer_count_after_qual_clipping_" + qs).c_str(), k.sumCount);
            o.line_u64(("distinct_" + ks + "mer_count_after_qual_clipping_" + qs).c_str(), k.F0);
            o.line_u64(("unique_" + ks + "mer_count_after_qual_clipping_" + qs).c_str(), k.f1);
            o.line_u64((ks + "mer_F2_after_qual_clipping_" + qs).c_str(), k.F2);
        }
        static const char* grp[4] = {"1st_FW", "1st_RC", "2nd_FW", "2nd_RC"};
        static const int gidx[4] = {0, 2, 1, 3}; // state order: fwd1st, fwd2nd, rev1st, rev2nd
        std::vector<uint64_t> row(64);
        for (int bi = 0; bi < 4; ++bi)
            for (int g = 0; g < 4; ++g) {
                for (int x = 0; x < 64; ++x) row[x] = L.triplet[x * 16 + gidx[g] * 4 + bi];
                std::string key = std::string("triplet_counts_") + "ACGT"[bi] + "_" + grp[g];
                o.array(key.c_str(), row.data(), 64);
            }
    }
    FILE* f = fopen(path, "wb");
    if (!f) return BQC_ERR_IO;
    size_t w = fwrite(o.s.data(), 1, o.s.size(), f);
    int rc = fclose(f);
    return (w == o.s.size() && rc == 0) ? 0 : BQC_ERR_IO;
}
