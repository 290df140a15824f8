/*
 * bamqc_oracle.c — CPU restatement of BamQC's per-read aggregation.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (`bamqc_amd/`, the C-ABI
 * library, the `bamqualcheck` CLI) links, loads or executes this file; only
 * `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg do,
 * and only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (DecodeGenetics/BamQC) ships no tests, golden
 * vectors or fixtures for this path and cannot be compiled here (it needs
 * SeqAn 1.4.2, which is neither vendored nor installed).  This restatement is
 * therefore pinned only by (a) the two structural pins the reference tree holds
 * (8-mer index orientation and triplet context index, both via
 * bamqc_summary.py:347,369-380) and (b) hand-derived known-answer tests from the
 * cited source lines (tests/test_oracle_kats.py).  The k-mer sketch part
 * (oracle/sketch_oracle.c) IS pinned against the reference's own
 * kmerstream sources compiled into oracle/_ref.
 *
 * Structure: record-at-a-time, single thread, same pass structure as the
 * reference.  Each function cites the reference lines it follows.  Input is the
 * public bqc_batch (include/bamqc.h); every record is first decoded into the
 * form SeqAn's readRecord would have produced (char sequence, ASCII qualities,
 * CIGAR elements) and then pushed through the loop body of
 * src/bamqualcheck.cpp:303-444.
 *
 * Cases where the reference has undefined behaviour are marked DEFINED: with
 * the behaviour both this oracle and the HIP path implement.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/bamqc.h"
#include "bamqc_oracle.h"

/* ------------------------------------------------------------------------- */
/* growable arrays (seqan::String<T> with resize(n, 0))                        */
/* ------------------------------------------------------------------------- */
typedef struct { uint64_t* d; size_t n, cap; } vec;

static void vec_resize(vec* v, size_t n) /* grow only, zero fill: resize(str, n, 0) */
{
    if (n <= v->n) return;
    if (n > v->cap) {
        size_t c = v->cap ? v->cap : 16;
        while (c < n) c *= 2;
        v->d = (uint64_t*)realloc(v->d, c * sizeof(uint64_t));
        v->cap = c;
    }
    memset(v->d + v->n, 0, (n - v->n) * sizeof(uint64_t));
    v->n = n;
}
static void vec_free(vec* v) { free(v->d); v->d = NULL; v->n = v->cap = 0; }
#define INC32(x) ((x) = (uint32_t)((x) + 1)) /* `unsigned` counter */

/* ------------------------------------------------------------------------- */
/* state (struct Counts, bamqualcheck.cpp:14-38)                               */
/* ------------------------------------------------------------------------- */
typedef struct { /* QualityCheck.hpp:8-56 */
    vec dnacount[5], qualcount, sc5, sc3;
    vec averageQual, Ncount, GCcount, insertSize, mapQ, readLength, mismatch, delhist, inshist;
    uint32_t delcount, inscount, qualcount_readnr;
} qcheck;

typedef struct { /* OverallNumbers.hpp:8-57 */
    uint32_t supplementary, duplicates, QCfailed, not_primary_alignment, readcount;
    uint64_t totalbps;
    uint32_t bothunmapped, firstunmapped, secondunmapped, first_and_or_second_mapped, FF_RR_orientation,
        properpair_count, auto_properpair_count;
    uint32_t poscov[BQC_COVSIZE + 1];
    uint64_t eightmercount[BQC_N_8MER];
    int first;
    uint32_t vsize, covsize;
    int32_t shift, id;
    uint32_t v1[BQC_VSIZE], v2[BQC_VSIZE];
} overall;

typedef struct { /* TripletCounting.hpp:29-46 */
    uint64_t forwardFirst[4], forwardSecond[4], reverseFirst[4], reverseSecond[4];
} tripletcounts;

typedef struct {
    overall all;
    qcheck r1, r2;
    tripletcounts triplet[64];
    void* sketch; /* N1, see sketch_oracle.c */
} counts_t;

struct orc_ctx {
    bqc_options opt;
    uint8_t* main_chrom;
    int32_t* fasta_index;
    counts_t* counts; /* [n_lanes] */
    const uint8_t** ref;
    uint64_t* ref_len;
    int32_t fasta_cursor; /* Genome: index (FASTA order) of the chromosome currently loaded, -1 = none */
    char err[256];
    /* finalize output */
    bqc_counts out;
    bqc_lane_counts* out_lanes;
    uint64_t* out_triplet;
    uint64_t* out_scratch;
    int flushed;
};

/* decoded record (seqan::BamAlignmentRecord after readRecord) */
typedef struct {
    uint32_t flag;
    uint32_t mapq;
    int32_t rid, pos, tlen;
    int mate_main;
    uint32_t L;
    char* seq;   /* [L]            */
    char* qual;  /* [qlen] ASCII   */
    uint32_t qlen;
    uint32_t ncig;
    char* cop;      /* operation chars */
    uint32_t* ccnt; /* counts          */
    int n_nm;
    int32_t nm[16];
    int32_t as;
} rec;

/* ------------------------------------------------------------------------- */
/* alphabets                                            */
/* ------------------------------------------------------------------------- */
static int dna5_of_char(char c) /* char -> Dna5 ordinal */
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 4;
    }
}
static int dna_of_char(char c) /* char -> Dna ordinal: everything else -> A */
{
    int v = dna5_of_char(c);
    return v == 4 ? 0 : v;
}
static char complement_char(char c) /* FunctorComplement<char>: IUPAC aware, case preserving */
{
    switch (c) {
    case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
    case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
    case 'U': return 'A'; case 'u': return 'a';
    case 'M': return 'K'; case 'K': return 'M'; case 'R': return 'Y'; case 'Y': return 'R';
    case 'V': return 'B'; case 'B': return 'V'; case 'H': return 'D'; case 'D': return 'H';
    case 'm': return 'k'; case 'k': return 'm'; case 'r': return 'y'; case 'y': return 'r';
    case 'v': return 'b'; case 'b': return 'v'; case 'h': return 'd'; case 'd': return 'h';
    default: return c; /* W S N = ... */
    }
}

/* ------------------------------------------------------------------------- */
/* OverallNumbers                                                              */
/* ------------------------------------------------------------------------- */
static void update_vectors(overall* o) /* OverallNumbers.hpp:59-64 */
{
    memcpy(o->v1, o->v2, sizeof o->v1); /* clear(v1); resize(v1,vsize,0); swap(v1,v2) */
    memset(o->v2, 0, sizeof o->v2);
}
static void update_coverage(overall* o) /* OverallNumbers.hpp:66-77 */
{
    for (uint32_t i = 0; i < o->vsize; ++i) {
        if (o->v1[i] > o->covsize) INC32(o->poscov[o->covsize]);
        else INC32(o->poscov[o->v1[i]]);
    }
}
static void coverage(overall* o, const rec* r) /* OverallNumbers.hpp:79-135 */
{
    uint32_t beginpos = (uint32_t)r->pos;
    if (o->first) { /* :84-89 */
        o->first = 0;
        o->id = r->rid;
        o->shift = (int32_t)beginpos;
    }
    if (o->id != r->rid || (uint32_t)(beginpos - (uint32_t)o->shift) > 2 * o->vsize) { /* :91-100 */
        o->id = r->rid;
        update_coverage(o);
        update_vectors(o);
        update_coverage(o);
        memset(o->v1, 0, sizeof o->v1);
        o->shift = (int32_t)beginpos;
    }
    uint32_t pos = beginpos - (uint32_t)o->shift;
    if (pos > o->vsize && pos < 2 * o->vsize) { /* :104-110 */
        update_coverage(o);
        update_vectors(o);
        o->shift += (int32_t)o->vsize;
        pos = beginpos - (uint32_t)o->shift;
    }
    int32_t c = 0;
    for (uint32_t i = 0; i < r->ncig; ++i) { /* :112-134 */
        if (r->cop[i] == 'S') c += (int32_t)r->ccnt[i];
        if (r->cop[i] == 'M' || r->cop[i] == 'D') {
            for (uint32_t j = (uint32_t)c; j < r->ccnt[i] + (uint32_t)c; ++j) {
                uint32_t off = pos + j;
                if (off < o->vsize) o->v1[off] += 1;
                else if (off - o->vsize < o->vsize) o->v2[off - o->vsize] += 1;
                /* DEFINED: the reference writes v2[pos - vsize + j] with no upper bound
                   (:127-130); offsets >= 2*vsize are dropped (SeqAn's over-allocated String
                   silently absorbs small overruns, SURVEY §7 hard part 2). */
            }
            c += (int32_t)r->ccnt[i];
        }
    }
}
static void count8mers(overall* o, const rec* r) /* OverallNumbers.hpp:137-168 */
{
    /* DEFINED: L < 8 counts nothing (the reference over-reads for L < 7). */
    if (r->L < 8) return;
    uint32_t skip = 0;
    for (uint32_t i = 0; i < 7; ++i) { /* :147-151 */
        if (r->seq[i] == 'N') skip = 8;
        if (skip > 0) --skip;
    }
    uint32_t h = 0;
    for (uint32_t i = 0; i < 7; ++i) h = (h << 2) | (uint32_t)dna_of_char(r->seq[i]); /* hashInit */
    for (uint32_t p = 0; p + 8 <= r->L; ++p) { /* :153-167, itSeq = p + 7 */
        h = ((h << 2) | (uint32_t)dna_of_char(r->seq[p + 7])) & 0xFFFFu; /* hashNext: big-endian base 4 */
        if (r->seq[p + 7] == 'N') skip = 8;
        if (skip > 0) --skip;
        else ++o->eightmercount[h];
    }
}

/* ------------------------------------------------------------------------- */
/* QualityCheck                                                                */
/* ------------------------------------------------------------------------- */
static void resize_strings(qcheck* q, uint32_t L) /* QualityCheck.hpp:85-105 */
{
    if (q->qualcount.n < L) {
        vec_resize(&q->qualcount, L);
        vec_resize(&q->Ncount, L + 1);
        vec_resize(&q->GCcount, L + 1);
        vec_resize(&q->sc5, L);
        vec_resize(&q->sc3, L);
        for (int j = 0; j < 5; ++j) vec_resize(&q->dnacount[j], L);
    }
}
static void read_counts(qcheck* q, const rec* r) /* QualityCheck.hpp:122-166 */
{
    uint32_t cntN = 0, cntGC = 0, avgQual = 0;
    q->qualcount_readnr += 1;
    for (uint32_t j = 0; j < r->L; ++j) {
        q->dnacount[dna5_of_char(r->seq[j])].d[j] += 1;
        if (r->seq[j] == 'N') cntN += 1;
        if (r->seq[j] == 'C' || r->seq[j] == 'G') cntGC += 1;
    }
    for (uint32_t j = 0; j < r->qlen; ++j) {
        int v = (int)(unsigned char)r->qual[j] - 33; /* ordValue(char) is unsigned */
        q->qualcount.d[j] += (uint64_t)(int64_t)v;
        avgQual += (uint32_t)v;
    }
    /* DEFINED: for L == 0 the reference indexes empty arrays / divides by zero; here the
       histograms below behave as if resize_strings had produced length >= 1 and the
       averageQual update is skipped. */
    vec_resize(&q->Ncount, 1);
    vec_resize(&q->GCcount, 1);
    INC32(q->Ncount.d[cntN]);
    q->GCcount.d[cntGC] += 1;
    if (r->L == 0) return;
    double m = (double)avgQual / (double)r->L;
    if ((double)q->averageQual.n <= ceil(m)) vec_resize(&q->averageQual, (size_t)(ceil(m) + 1));
    INC32(q->averageQual.d[(int)round(m)]);
}
static void read_length(qcheck* q, const rec* r) /* QualityCheck.hpp:168-176 */
{
    if (q->readLength.n <= r->L) vec_resize(&q->readLength, r->L + 1);
    INC32(q->readLength.d[r->L]);
}
static void get_count(qcheck* q, const rec* r) /* QualityCheck.hpp:111-116 */
{
    resize_strings(q, r->L);
    read_counts(q, r);
    read_length(q, r);
}
static void map_Q(qcheck* q, uint32_t mapq) /* QualityCheck.hpp:178-185 */
{
    if (q->mapQ.n <= mapq) vec_resize(&q->mapQ, mapq + 1);
    INC32(q->mapQ.d[mapq]);
}
static void insert_size(qcheck* q, int32_t tlen) /* QualityCheck.hpp:187-196 */
{
    uint32_t index = (tlen == INT32_MIN) ? 0x80000000u : (uint32_t)abs(tlen);
    if (index >= q->insertSize.n) index = (uint32_t)q->insertSize.n - 1;
    INC32(q->insertSize.d[index]);
}
static int mis_match(qcheck* q, const rec* r, uint32_t hist_cap) /* QualityCheck.hpp:198-220 */
{
    for (int t = 0; t < r->n_nm; ++t) {
        uint32_t x = (uint32_t)r->nm[t];
        uint32_t mmcount = x - q->delcount - q->inscount; /* unsigned arithmetic, :210 */
        /* DEFINED: NM < D+I wraps to ~2^32 and the reference dies in resize(); any value
           >= hist_cap is a fatal input error here. */
        if (mmcount >= hist_cap) return BQC_ERR_RANGE;
        if (q->mismatch.n <= mmcount) vec_resize(&q->mismatch, (size_t)mmcount + 1);
        INC32(q->mismatch.d[mmcount]);
    }
    return 0;
}
static int cigar_count(qcheck* q, const rec* r, uint32_t hist_cap) /* QualityCheck.hpp:222-271 */
{
    q->delcount = 0;
    q->inscount = 0;
    int cigarlength = (int)r->ncig;
    /* DEFINED: an empty CIGAR on a mapped read (cigar[0] out of bounds in the reference)
       is treated as "no operations". */
    if (cigarlength > 0) {
        if (r->cop[0] == 'S') {
            /* DEFINED: a clip longer than the read itself (inconsistent record; the reference
               writes past the per-cycle arrays) only marks the read's own cycles. */
            for (uint32_t j = 0; j < r->ccnt[0] && j < r->L; ++j) INC32(q->sc5.d[j]);
        } else if (r->cop[cigarlength - 1] == 'S') {
            uint32_t n = r->ccnt[cigarlength - 1];
            for (uint32_t j = r->L - n; j < r->L; ++j) /* unsigned: n > L wraps and the loop is empty */
                INC32(q->sc3.d[j]);
        }
    }
    for (int i = 0; i < cigarlength; ++i) {
        if (r->cop[i] == 'D') q->delcount += r->ccnt[i];
        else if (r->cop[i] == 'I') q->inscount += r->ccnt[i];
    }
    if (q->delcount >= hist_cap || q->inscount >= hist_cap) return BQC_ERR_RANGE;
    if (q->delhist.n <= q->delcount) vec_resize(&q->delhist, (size_t)q->delcount + 1);
    INC32(q->delhist.d[q->delcount]);
    if (q->inshist.n <= q->inscount) vec_resize(&q->inshist, (size_t)q->inscount + 1);
    INC32(q->inshist.d[q->inscount]);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* TripletCounting                                                             */
/* ------------------------------------------------------------------------- */
/* returns 1 eligible, 0 not, -1 fatal (TripletCounting.hpp:136-168) */
static int checkFlagsAndQuality(const rec* r)
{
    /* flags "1100xxxx000x" (:21): multiple=1, proper=1, unmapped -> 0, next-unmapped=0, secondary=0 */
    if (!(r->flag & 0x1)) return 0;
    if (!(r->flag & 0x2)) return 0;
    if (r->flag & 0x4) return 0;
    if (r->flag & 0x8) return 0;
    if (r->flag & 0x100) return 0;
    if (r->mapq < 60) return 0;                  /* :153 */
    if (r->as == BQC_AS_ABSENT) return -1;       /* alignmentScore() :116-120 */
    if (r->as < 0) return -1;                    /* :155 `if (as < 0) return -1` also hits negative scores */
    if (r->as < 50) return 0;                    /* :156 */
    uint32_t clipped = 0;                        /* :159-165 */
    for (uint32_t i = 0; i < r->ncig; ++i)
        if (r->cop[i] == 'S' || r->cop[i] == 'H') clipped += r->ccnt[i];
    if (clipped > 0) return 0;
    return 1;
}
static void countPosition(tripletcounts* c, int base, const rec* r) /* TripletCounting.hpp:174-189 */
{
    if (r->flag & 0x10) {
        if (r->flag & 0x40) c->reverseFirst[base] += 1;
        else c->reverseSecond[base] += 1;
    } else {
        if (r->flag & 0x40) c->forwardFirst[base] += 1;
        else c->forwardSecond[base] += 1;
    }
}
static void countBasesInTriplets(tripletcounts* counts, const rec* r, const uint8_t* chrom, uint64_t chromLen)
{ /* TripletCounting.hpp:195-236 */
    /* DEFINED: empty CIGAR, L < 3 and missing qualities count nothing (the reference
       dereferences begin(cigar) / computes length-1 on size_t / indexes an empty qual). */
    if (r->ncig == 0 || r->L < 3 || r->qlen != r->L) return;
    uint32_t it = 0;
    uint64_t cigarCount = (uint64_t)r->ccnt[0] - 1; /* first op assumed match-like; size_t wrap if count==0 */
    uint64_t chromPos = (uint64_t)(int64_t)r->pos + 1;
    for (uint64_t readPos = 1; readPos < (uint64_t)r->L - 1; ++readPos, ++chromPos, --cigarCount) {
        int ran_off = 0;
        while (cigarCount == 0) {
            ++it;
            if (it >= r->ncig) { ran_off = 1; break; } /* DEFINED: SEQAN_ASSERT in the reference; stop */
            char op = r->cop[it];
            if (op == 'D' || op == 'N' || op == 'H' || op == 'P') chromPos += r->ccnt[it];
            else if (op == 'S' || op == 'I') readPos += r->ccnt[it];
            else cigarCount = r->ccnt[it];
        }
        if (ran_off) break;
        if (readPos >= (uint64_t)r->L - 1) break;
        if ((signed char)r->qual[readPos] < 53) continue; /* char compare, minBaseQAscii = 53 */
        int base = dna5_of_char(r->seq[readPos]);
        if (base == 4 || r->seq[readPos - 1] == 'N' || r->seq[readPos + 1] == 'N') continue;
        /* DEFINED: context outside the chromosome is skipped (infix() reads out of bounds). */
        if (chromPos < 1 || chromLen < 2 || chromPos > chromLen - 2) continue; /* (no chromPos + 1: it wraps for beginPos <= -2) */
        int c0 = chrom[chromPos - 1] & 3, c1 = chrom[chromPos] & 3, c2 = chrom[chromPos + 1] & 3; /* Dna5->Dna: N->A */
        if (dna_of_char(r->seq[readPos - 1]) != c0) continue; /* char compared as Dna */
        if (dna_of_char(r->seq[readPos + 1]) != c2) continue;
        countPosition(&counts[(c0 << 4) + (c1 << 2) + c2], base, r); /* contextToIndex :50-54 */
    }
}
static int tripletCounting(struct orc_ctx* o, tripletcounts* counts, const rec* r) /* TripletCounting.hpp:242-265 */
{
    int res = checkFlagsAndQuality(r);
    if (res == -1) return BQC_ERR_AS_TAG;
    if (res == 0) return 0;
    /* Genome stream: forward-only scan until the names match (:254-259) */
    if (r->rid < 0 || (uint32_t)r->rid >= o->opt.n_refs) return BQC_ERR_FASTA;
    int32_t target = o->fasta_index ? o->fasta_index[r->rid] : r->rid;
    if (target < 0) return BQC_ERR_FASTA;           /* reads to EOF */
    if (target < o->fasta_cursor) return BQC_ERR_FASTA; /* contig lies before the cursor */
    o->fasta_cursor = target;
    if (!o->ref[r->rid]) return BQC_ERR_FASTA;
    countBasesInTriplets(counts, r, o->ref[r->rid], o->ref_len[r->rid]);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* the loop body, bamqualcheck.cpp:303-444                                     */
/* ------------------------------------------------------------------------- */

static int process_record(struct orc_ctx* o, rec* r, uint32_t lane)
{
    counts_t* c = &o->counts[lane];
    if (r->flag & 0x800) { INC32(c->all.supplementary); return 0; }        /* :318-322 */
    if (r->flag & 0x100) { INC32(c->all.not_primary_alignment); return 0; } /* :323-327 */
    int dup = (r->flag & 0x400) != 0, qcfail = (r->flag & 0x200) != 0;
    if (dup) INC32(c->all.duplicates);
    if (qcfail) INC32(c->all.QCfailed);

    if (!dup && !qcfail) { /* :338-342 */
        int e = tripletCounting(o, c->triplet, r);
        if (e) return e;
    }
    if (r->flag & 0x10) { /* :345-350 reverseComplement(seq); reverse(qual); reverse(cigar) */
        for (uint32_t i = 0; i < r->L / 2; ++i) { char t = r->seq[i]; r->seq[i] = r->seq[r->L - 1 - i]; r->seq[r->L - 1 - i] = t; }
        for (uint32_t i = 0; i < r->L; ++i) r->seq[i] = complement_char(r->seq[i]);
        for (uint32_t i = 0; i < r->qlen / 2; ++i) { char t = r->qual[i]; r->qual[i] = r->qual[r->qlen - 1 - i]; r->qual[r->qlen - 1 - i] = t; }
        for (uint32_t i = 0; i < r->ncig / 2; ++i) {
            uint32_t j = r->ncig - 1 - i;
            char t = r->cop[i]; r->cop[i] = r->cop[j]; r->cop[j] = t;
            uint32_t u = r->ccnt[i]; r->ccnt[i] = r->ccnt[j]; r->ccnt[j] = u;
        }
    }
    INC32(c->all.readcount);   /* :353 */
    c->all.totalbps += r->L;   /* :354 */
    int first = (r->flag & 0x40) != 0, last = (r->flag & 0x80) != 0;
    int unmapped = (r->flag & 0x4) != 0, next_unmapped = (r->flag & 0x8) != 0;
    int proper = (r->flag & 0x2) != 0, rc = (r->flag & 0x10) != 0, next_rc = (r->flag & 0x20) != 0;
    if (first) { /* :355-375 */
        get_count(&c->r1, r);
        if (unmapped) {
            INC32(c->all.firstunmapped);
            if (next_unmapped) INC32(c->all.bothunmapped);
        }
        if (proper) {
            INC32(c->all.properpair_count);
            if ((!rc && !next_rc) || (rc && next_rc)) INC32(c->all.FF_RR_orientation);
        }
    } else if (last) { /* :376-384 */
        get_count(&c->r2, r);
        if (unmapped) INC32(c->all.secondunmapped);
    } else {
        return BQC_ERR_NO_MATE_FLAG; /* :385-389 */
    }
    int in_main = r->rid >= 0 && (uint32_t)r->rid < o->opt.n_refs && o->main_chrom[r->rid];
    if (in_main) { /* :392-434 */
        int e;
        if (first) {
            if (!unmapped) {
                if ((e = cigar_count(&c->r1, r, o->opt.hist_cap))) return e;
                map_Q(&c->r1, r->mapq);
                if ((e = mis_match(&c->r1, r, o->opt.hist_cap))) return e;
                if (!next_unmapped && r->mate_main) insert_size(&c->r1, r->tlen);
            }
            if ((!unmapped || !next_unmapped) && !dup) INC32(c->all.first_and_or_second_mapped);
            if (proper && !dup) INC32(c->all.auto_properpair_count);
        } else if (last) {
            if (!unmapped) {
                if ((e = cigar_count(&c->r2, r, o->opt.hist_cap))) return e;
                map_Q(&c->r2, r->mapq);
                if ((e = mis_match(&c->r2, r, o->opt.hist_cap))) return e;
            }
        }
        if (!unmapped && !dup) coverage(&c->all, r);
    }
    count8mers(&c->all, r); /* :437 */
    if (!qcfail && !dup && c->sketch) /* :439-442 */
        orc_sketch_run(c->sketch, r->seq, r->L, r->qual, r->qlen);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* public API                                                                  */
/* ------------------------------------------------------------------------- */

int orc_create(const bqc_options* opt, struct orc_ctx** out)
{
    if (!opt || !out || opt->n_lanes == 0 || opt->isize < 0) return BQC_ERR_ARG;
    struct orc_ctx* o = (struct orc_ctx*)calloc(1, sizeof *o);
    o->opt = *opt;
    o->main_chrom = (uint8_t*)calloc(opt->n_refs ? opt->n_refs : 1, 1);
    if (opt->main_chrom) memcpy(o->main_chrom, opt->main_chrom, opt->n_refs);
    if (opt->fasta_index) {
        o->fasta_index = (int32_t*)malloc(sizeof(int32_t) * (opt->n_refs ? opt->n_refs : 1));
        memcpy(o->fasta_index, opt->fasta_index, sizeof(int32_t) * opt->n_refs);
    }
    o->ref = (const uint8_t**)calloc(opt->n_refs ? opt->n_refs : 1, sizeof(uint8_t*));
    o->ref_len = (uint64_t*)calloc(opt->n_refs ? opt->n_refs : 1, sizeof(uint64_t));
    o->counts = (counts_t*)calloc(opt->n_lanes, sizeof(counts_t));
    for (uint32_t l = 0; l < opt->n_lanes; ++l) {
        counts_t* c = &o->counts[l];
        c->all.first = 1; c->all.vsize = BQC_VSIZE; c->all.covsize = BQC_COVSIZE; /* OverallNumbers.hpp:50-57 */
        vec_resize(&c->r1.insertSize, (size_t)opt->isize + 1); /* QualityCheck.hpp:60-64 */
        vec_resize(&c->r2.insertSize, (size_t)opt->isize + 1);
        c->sketch = opt->sketch.n_k && opt->sketch.n_q ? orc_sketch_create(&opt->sketch) : NULL;
    }
    o->fasta_cursor = -1;
    *out = o;
    return 0;
}

int orc_set_reference(struct orc_ctx* o, int32_t rid, const uint8_t* dna5, uint64_t len)
{
    if (rid < 0 || (uint32_t)rid >= o->opt.n_refs) return BQC_ERR_ARG;
    uint8_t* p = (uint8_t*)malloc(len ? len : 1);
    memcpy(p, dna5, len);
    free((void*)o->ref[rid]);
    o->ref[rid] = p;
    o->ref_len[rid] = len;
    return 0;
}

static const char SEQ_DECODE[] = "=ACMGRSVTWYHKDBN"; /* SURVEY U8 */
static const char CIG_DECODE[] = "MIDNSHP=X";

int orc_process_batch(struct orc_ctx* o, const bqc_batch* b)
{
    uint64_t so = 0, qo = 0, co = 0;
    uint32_t xe = 0;
    size_t capL = 0, capC = 0;
    rec r;
    memset(&r, 0, sizeof r);
    int rc = 0;
    for (uint32_t i = 0; i < b->n_reads && !rc; ++i) {
        uint32_t L = b->l_seq[i], nc = b->n_cigar[i];
        if (L > o->opt.max_read_len) { rc = BQC_ERR_RANGE; break; }
        if (L + 1 > capL) { capL = (size_t)L * 2 + 16; r.seq = (char*)realloc(r.seq, capL); r.qual = (char*)realloc(r.qual, capL); }
        if (nc + 1 > capC) { capC = (size_t)nc * 2 + 16; r.cop = (char*)realloc(r.cop, capC); r.ccnt = (uint32_t*)realloc(r.ccnt, capC * sizeof(uint32_t)); }
        r.flag = b->flag[i] & 0x0FFFu;
        r.mate_main = (b->flag[i] & BQC_FLAG_MATE_MAIN) != 0;
        r.mapq = b->mapq[i];
        r.rid = b->rid[i]; r.pos = b->pos[i]; r.tlen = b->tlen[i];
        r.L = L;
        for (uint32_t j = 0; j < L; ++j) {
            uint8_t by = b->seq[so + (j >> 1)];
            r.seq[j] = SEQ_DECODE[(j & 1) ? (by & 15) : (by >> 4)];
        }
        /* quality block starting with 0xFF => empty qual; else +33 per byte */
        if ((b->flag[i] & BQC_FLAG_NO_QUAL) || (L > 0 && b->qual[qo] == 0xFF)) r.qlen = 0;
        else { r.qlen = L; for (uint32_t j = 0; j < L; ++j) r.qual[j] = (char)(uint8_t)(b->qual[qo + j] + 33); }
        r.ncig = nc;
        for (uint32_t j = 0; j < nc; ++j) {
            uint32_t v = b->cigar[co + j];
            r.cop[j] = (v & 15) < 9 ? CIG_DECODE[v & 15] : '?';
            r.ccnt[j] = v >> 4;
        }
        r.n_nm = 0;
        if (b->nm[i] != BQC_NM_ABSENT) r.nm[r.n_nm++] = b->nm[i];
        while (xe < b->n_nm_extra && b->nm_extra_read[xe] == i) {
            if (r.n_nm < 16) r.nm[r.n_nm++] = b->nm_extra_val[xe];
            ++xe;
        }
        r.as = b->as[i];
        if (b->lane[i] >= o->opt.n_lanes) rc = BQC_ERR_ARG;
        else rc = process_record(o, &r, b->lane[i]);
        so += (L + 1) / 2; qo += L; co += nc;
    }
    free(r.seq); free(r.qual); free(r.cop); free(r.ccnt);
    return rc;
}

/* final flush, bamqualcheck.cpp:447-453 */
static void final_flush(struct orc_ctx* o)
{
    if (o->flushed) return;
    for (uint32_t l = 0; l < o->opt.n_lanes; ++l) {
        update_coverage(&o->counts[l].all);
        update_vectors(&o->counts[l].all);
        update_coverage(&o->counts[l].all);
    }
    o->flushed = 1;
}

static void fill_mate(bqc_mate_counts* m, qcheck* q)
{
    m->n_cycles = (uint32_t)q->qualcount.n;
    for (int j = 0; j < 5; ++j) m->dnacount[j] = q->dnacount[j].d;
    m->qualcount = q->qualcount.d;
    m->qualcount_readnr = q->qualcount_readnr;
    m->sc5 = q->sc5.d; m->sc3 = q->sc3.d;
    m->n_Ncount = (uint32_t)q->Ncount.n; m->Ncount = q->Ncount.d;
    m->n_GCcount = (uint32_t)q->GCcount.n; m->GCcount = q->GCcount.d;
    m->n_averageQual = (uint32_t)q->averageQual.n; m->averageQual = q->averageQual.d;
    m->n_insertSize = (uint32_t)q->insertSize.n; m->insertSize = q->insertSize.d;
    m->n_mapQ = (uint32_t)q->mapQ.n; m->mapQ = q->mapQ.d;
    m->n_readLength = (uint32_t)q->readLength.n; m->readLength = q->readLength.d;
    m->n_mismatch = (uint32_t)q->mismatch.n; m->mismatch = q->mismatch.d;
    m->n_delhist = (uint32_t)q->delhist.n; m->delhist = q->delhist.d;
    m->n_inshist = (uint32_t)q->inshist.n; m->inshist = q->inshist.d;
}

int orc_finalize(struct orc_ctx* o, const bqc_counts** out)
{
    final_flush(o);
    uint32_t nl = o->opt.n_lanes;
    if (!o->out_lanes) {
        o->out_lanes = (bqc_lane_counts*)calloc(nl, sizeof(bqc_lane_counts));
        o->out_triplet = (uint64_t*)calloc((size_t)nl * BQC_N_TRIPLET, sizeof(uint64_t));
    }
    for (uint32_t l = 0; l < nl; ++l) {
        counts_t* c = &o->counts[l];
        bqc_lane_counts* L = &o->out_lanes[l];
        uint64_t* s = L->scalars;
        s[BQC_S_SUPPLEMENTARY] = c->all.supplementary; s[BQC_S_DUPLICATES] = c->all.duplicates;
        s[BQC_S_QCFAILED] = c->all.QCfailed; s[BQC_S_NOT_PRIMARY] = c->all.not_primary_alignment;
        s[BQC_S_READCOUNT] = c->all.readcount; s[BQC_S_TOTALBPS] = c->all.totalbps;
        s[BQC_S_BOTHUNMAPPED] = c->all.bothunmapped; s[BQC_S_FIRSTUNMAPPED] = c->all.firstunmapped;
        s[BQC_S_SECONDUNMAPPED] = c->all.secondunmapped;
        s[BQC_S_FIRST_AND_OR_SECOND_MAPPED] = c->all.first_and_or_second_mapped;
        s[BQC_S_FF_RR] = c->all.FF_RR_orientation; s[BQC_S_PROPERPAIR] = c->all.properpair_count;
        s[BQC_S_AUTO_PROPERPAIR] = c->all.auto_properpair_count;
        for (int i = 0; i <= BQC_COVSIZE; ++i) L->poscov[i] = c->all.poscov[i];
        L->eightmer = c->all.eightmercount;
        fill_mate(&L->mate[0], &c->r1);
        fill_mate(&L->mate[1], &c->r2);
        uint64_t* t = o->out_triplet + (size_t)l * BQC_N_TRIPLET;
        for (int x = 0; x < 64; ++x)
            for (int b = 0; b < 4; ++b) {
                t[x * 16 + 0 * 4 + b] = c->triplet[x].forwardFirst[b];
                t[x * 16 + 1 * 4 + b] = c->triplet[x].forwardSecond[b];
                t[x * 16 + 2 * 4 + b] = c->triplet[x].reverseFirst[b];
                t[x * 16 + 3 * 4 + b] = c->triplet[x].reverseSecond[b];
            }
        L->triplet = t;
        if (c->sketch) {
            uint32_t n = o->opt.sketch.n_k * o->opt.sketch.n_q;
            bqc_sketch_counts* sk = (bqc_sketch_counts*)calloc(n, sizeof *sk);
            L->n_sketch = orc_sketch_results(c->sketch, sk);
            free((void*)L->sketch);
            L->sketch = sk;
        }
    }
    o->out.n_lanes = nl;
    o->out.lanes = o->out_lanes;
    *out = &o->out;
    return 0;
}

void orc_destroy(struct orc_ctx* o)
{
    if (!o) return;
    for (uint32_t l = 0; l < o->opt.n_lanes; ++l) {
        qcheck* qs[2] = { &o->counts[l].r1, &o->counts[l].r2 };
        for (int m = 0; m < 2; ++m) {
            qcheck* q = qs[m];
            for (int j = 0; j < 5; ++j) vec_free(&q->dnacount[j]);
            vec_free(&q->qualcount); vec_free(&q->sc5); vec_free(&q->sc3); vec_free(&q->averageQual);
            vec_free(&q->Ncount); vec_free(&q->GCcount); vec_free(&q->insertSize); vec_free(&q->mapQ);
            vec_free(&q->readLength); vec_free(&q->mismatch); vec_free(&q->delhist); vec_free(&q->inshist);
        }
        if (o->counts[l].sketch) orc_sketch_destroy(o->counts[l].sketch);
        if (o->out_lanes) free((void*)o->out_lanes[l].sketch);
    }
    for (uint32_t r = 0; r < o->opt.n_refs; ++r) free((void*)o->ref[r]);
    free(o->ref); free(o->ref_len); free(o->main_chrom); free(o->fasta_index);
    free(o->counts); free(o->out_lanes); free(o->out_triplet);
    free(o);
}

/* ------------------------------------------------------------------------- */
/* `.bamqc` writer: writeOutput (bamqualcheck.cpp:156-233), printString        */
/* (:130-139), ten_most_abundant_kmers (OverallNumbers.hpp:170-216),           */
/* avgQualPerPos (QualityCheck.hpp:273-279), writeTripletCounts                */
/* (TripletCounting.hpp:271-301)                                               */
/* ------------------------------------------------------------------------- */
static void print_u64(FILE* f, const char* key, const uint64_t* v, size_t n)
{
    fputs(key, f);
    for (size_t i = 0; i < n; ++i) fprintf(f, " %llu", (unsigned long long)v[i]);
    fputc('\n', f);
}
static int cmp_greater_int(const void* a, const void* b) /* std::greater<int> on uint64 values (:177) */
{
    int x = (int)*(const uint64_t*)a, y = (int)*(const uint64_t*)b;
    return (x > y) ? -1 : (x < y) ? 1 : 0;
}
static int cmp_greater_u64(const void* a, const void* b)
{
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return (x > y) ? -1 : (x < y) ? 1 : 0;
}
static void ten_most_abundant_kmers(FILE* f, const uint64_t* eightmer)
{
    uint64_t* copy = (uint64_t*)malloc(sizeof(uint64_t) * BQC_N_8MER);
    memcpy(copy, eightmer, sizeof(uint64_t) * BQC_N_8MER);
    qsort(copy, BQC_N_8MER, sizeof(uint64_t), cmp_greater_int); /* nth_element(it, it+10, end, greater<int>) */
    uint64_t top[10];
    memcpy(top, copy, sizeof top);
    free(copy);
    qsort(top, 10, sizeof(uint64_t), cmp_greater_u64); /* :187 */
    int used[10], nused = 0;
    for (int i = 0; i < 10; ++i) {
        int pos = 0;
        for (;; ++pos) { /* lowest index holding the value that is not yet used (:197-212) */
            if (pos >= BQC_N_8MER) break;
            if (eightmer[pos] != top[i]) continue;
            int seen = 0;
            for (int k = 0; k < nused; ++k) if (used[k] == pos) seen = 1;
            if (!seen) break;
        }
        used[nused++] = pos;
        char s[9];
        for (int k = 0; k < 8; ++k) s[k] = "ACGT"[(pos >> (2 * (7 - k))) & 3]; /* unhash: big-endian base 4 */
        s[8] = 0;
        fprintf(f, "nr_%d_most_abundant_8mer %s %llu\n", i + 1, s, (unsigned long long)top[i]);
    }
}
static void print_avgqual(FILE* f, const char* key, const bqc_mate_counts* m)
{
    fputs(key, f);
    for (uint32_t i = 0; i < m->n_cycles; ++i) /* default ostream formatting of double == %g */
        fprintf(f, " %g", (double)m->qualcount[i] / (double)(uint32_t)m->qualcount_readnr);
    fputc('\n', f);
}

int orc_write_bamqc(const bqc_counts* counts, const bqc_header_info* hdr, const char* path)
{
    FILE* f = fopen(path, "wb");
    if (!f) return BQC_ERR_IO;
    for (uint32_t n = 0; n < hdr->n_names; ++n) {
        const bqc_lane_counts* L = &counts->lanes[hdr->lane_index[n]];
        const bqc_mate_counts* a = &L->mate[0];
        const bqc_mate_counts* b = &L->mate[1];
        fprintf(f, "sample_id %s\n", hdr->sample_id);
        fprintf(f, "lane %s\n", hdr->lane_names[n]);
        fprintf(f, "total_read_pairs %llu\n", (unsigned long long)((uint32_t)L->scalars[BQC_S_READCOUNT] / 2));
        fprintf(f, "total_bps %llu\n", (unsigned long long)L->scalars[BQC_S_TOTALBPS]);
        fprintf(f, "supplementary_alignments %llu\n", (unsigned long long)L->scalars[BQC_S_SUPPLEMENTARY]);
        fprintf(f, "marked_duplicate %llu\n", (unsigned long long)L->scalars[BQC_S_DUPLICATES]);
        fprintf(f, "QC_failed %llu\n", (unsigned long long)L->scalars[BQC_S_QCFAILED]);
        fprintf(f, "not_primary_alignment %llu\n", (unsigned long long)L->scalars[BQC_S_NOT_PRIMARY]);
        fprintf(f, "both_reads_unmapped %llu\n", (unsigned long long)L->scalars[BQC_S_BOTHUNMAPPED]);
        fprintf(f, "first_read_unmapped %llu\n", (unsigned long long)L->scalars[BQC_S_FIRSTUNMAPPED]);
        fprintf(f, "second_read_unmapped %llu\n", (unsigned long long)L->scalars[BQC_S_SECONDUNMAPPED]);
        fprintf(f, "first_and_or_second_read_mapped %llu\n", (unsigned long long)L->scalars[BQC_S_FIRST_AND_OR_SECOND_MAPPED]);
        fprintf(f, "FF_RR_oriented_pairs %llu\n", (unsigned long long)L->scalars[BQC_S_FF_RR]);
        fprintf(f, "total_proper_pairs %llu\n", (unsigned long long)L->scalars[BQC_S_PROPERPAIR]);
        fprintf(f, "total_proper_pairs_autosome %llu\n", (unsigned long long)L->scalars[BQC_S_AUTO_PROPERPAIR]);
        print_u64(f, "genome_coverage_histogram", L->poscov, BQC_COVSIZE + 1);
        print_u64(f, "insert_size_histogram", a->insertSize, a->n_insertSize);
        print_u64(f, "read_length_histogram_first", a->readLength, a->n_readLength);
        print_u64(f, "read_length_histogram_second", b->readLength, b->n_readLength);
        print_u64(f, "N_count_histogram_first", a->Ncount, a->n_Ncount);
        print_u64(f, "N_count_histogram_second", b->Ncount, b->n_Ncount);
        print_u64(f, "GC_content_histogram_first", a->GCcount, a->n_GCcount);
        print_u64(f, "GC_content_histogram_second", b->GCcount, b->n_GCcount);
        print_u64(f, "average_base_qual_histogram_first", a->averageQual, a->n_averageQual);
        print_u64(f, "average_base_qual_histogram_second", b->averageQual, b->n_averageQual);
        print_u64(f, "mapping_qual_histogram_first", a->mapQ, a->n_mapQ);
        print_u64(f, "mapping_qual_histogram_second", b->mapQ, b->n_mapQ);
        print_u64(f, "mismatch_count_histogram_first", a->mismatch, a->n_mismatch);
        print_u64(f, "mismatch_count_histogram_second", b->mismatch, b->n_mismatch);
        print_u64(f, "deletion_count_histogram_first", a->delhist, a->n_delhist);
        print_u64(f, "deletion_count_histogram_second", b->delhist, b->n_delhist);
        print_u64(f, "insertion_count_histogram_first", a->inshist, a->n_inshist);
        print_u64(f, "insertion_count_histogram_second", b->inshist, b->n_inshist);
        static const int order[5] = { 4, 0, 1, 2, 3 }; /* N A C G T (:203-212) */
        static const char* nm1[5] = { "Ns_by_position_first", "As_by_position_first", "Cs_by_position_first",
                                      "Gs_by_position_first", "Ts_by_position_first" };
        static const char* nm2[5] = { "Ns_by_position_second", "As_by_position_second", "Cs_by_position_second",
                                      "Gs_by_position_second", "Ts_by_position_second" };
        for (int k = 0; k < 5; ++k) {
            print_u64(f, nm1[k], a->dnacount[order[k]], a->n_cycles);
            print_u64(f, nm2[k], b->dnacount[order[k]], b->n_cycles);
        }
        print_avgqual(f, "average_base_qual_by_position_first", a);
        print_avgqual(f, "average_base_qual_by_position_second", b);
        print_u64(f, "soft_clipping_5_prime_by_position_first", a->sc5, a->n_cycles);
        print_u64(f, "soft_clipping_3_prime_by_position_first", a->sc3, a->n_cycles);
        print_u64(f, "soft_clipping_5_prime_by_position_second", b->sc5, b->n_cycles);
        print_u64(f, "soft_clipping_3_prime_by_position_second", b->sc3, b->n_cycles);
        ten_most_abundant_kmers(f, L->eightmer);
        print_u64(f, "8mer_count", L->eightmer, BQC_N_8MER);
        for (uint32_t s = 0; s < L->n_sketch; ++s) { /* :221-230 */
            const bqc_sketch_counts* k = &L->sketch[s];
            fprintf(f, "%umer_count_after_qual_clipping_%u %llu\n", k->k, k->q, (unsigned long long)k->sumCount);
            fprintf(f, "distinct_%umer_count_after_qual_clipping_%u %llu\n", k->k, k->q, (unsigned long long)k->F0);
            fprintf(f, "unique_%umer_count_after_qual_clipping_%u %llu\n", k->k, k->q, (unsigned long long)k->f1);
            fprintf(f, "%umer_F2_after_qual_clipping_%u %llu\n", k->k, k->q, (unsigned long long)k->F2);
        }
        static const char bases[4] = { 'A', 'C', 'G', 'T' };
        static const char* grp[4] = { "1st_FW", "1st_RC", "2nd_FW", "2nd_RC" };
        static const int gidx[4] = { 0, 2, 1, 3 }; /* forwardFirst, reverseFirst, forwardSecond, reverseSecond */
        for (int bi = 0; bi < 4; ++bi)
            for (int g = 0; g < 4; ++g) {
                fprintf(f, "triplet_counts_%c_%s", bases[bi], grp[g]);
                for (int x = 0; x < 64; ++x)
                    fprintf(f, " %llu", (unsigned long long)L->triplet[x * 16 + gidx[g] * 4 + bi]);
                fputc('\n', f);
            }
    }
    fclose(f);
    return 0;
}
