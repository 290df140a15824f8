/*
 * sketch_oracle.c — CPU restatement of BamQC's k-mer sketch.
 *
 * TEST INFRASTRUCTURE ONLY (see the header of bamqc_oracle.c).
 *
 * Pinning: RepHash, the MT19937 seeding of its character table and
 * StreamCounter are checked against the reference's own sources compiled into
 * oracle/_ref (oracle/ref_kmerstream_shim.cpp, built by oracle/Makefile when
 * /root/reference is present) and against tests/golden/kmerstream_*.json that
 * were generated from that build.  The 40-line clipping loop
 * ReadQualityHasher::operator() cannot be compiled (it includes SeqAn via
 * CommandLineParser.hpp) and is restated from src/ReadQualityHasher.hpp:30-68.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/bamqc.h"
#include "bamqc_oracle.h"

/* ---- MT19937 as used by src/kmerstream/mersennetwister.h:194-208,298-329 ---- */
typedef struct { uint32_t state[624]; int next, left; } mt_t;
static void mt_seed(mt_t* m, uint32_t seed)
{
    m->state[0] = seed;
    for (int i = 1; i < 624; ++i)
        m->state[i] = 1812433253u * (m->state[i - 1] ^ (m->state[i - 1] >> 30)) + (uint32_t)i;
    m->left = 0; /* MTRand::seed() calls reload() right away; equivalent */
    m->next = 0;
}
static uint32_t mt_twist(uint32_t m, uint32_t s0, uint32_t s1)
{
    return m ^ (((s0 & 0x80000000u) | (s1 & 0x7fffffffu)) >> 1) ^ ((uint32_t)(-(int32_t)(s1 & 1u)) & 0x9908b0dfu);
}
static void mt_reload(mt_t* m)
{
    uint32_t* p = m->state;
    int i;
    for (i = 624 - 397; i--; ++p) *p = mt_twist(p[397], p[0], p[1]);
    for (i = 397; --i; ++p) *p = mt_twist(p[397 - 624], p[0], p[1]);
    *p = mt_twist(p[397 - 624], p[0], m->state[0]);
    m->left = 624; m->next = 0;
}
static uint32_t mt_randInt(mt_t* m)
{
    if (m->left == 0) mt_reload(m);
    --m->left;
    uint32_t s1 = m->state[m->next++];
    s1 ^= (s1 >> 11);
    s1 ^= (s1 << 7) & 0x9d2c5680u;
    s1 ^= (s1 << 15) & 0xefc60000u;
    return s1 ^ (s1 >> 18);
}

/* ---- RepHash (src/kmerstream/RepHash.hpp:28-122, RepHash.cpp:4-17) ---------- */
typedef struct { uint64_t hi, lo; } state_t;
static const unsigned char twin[32] = { /* RepHash.hpp:8-13: A(1)<->T(20), C(3)<->G(7) */
    0, 20, 2, 7, 4, 5, 6, 3, 8, 9, 10, 11, 12, 13, 14, 15,
    16, 17, 18, 19, 1, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31 };
typedef struct {
    size_t k;
    uint64_t lastkmask, firstkmask;
    state_t h, ht;
    state_t hvals[32];
} rephash;

void orc_rephash_table(int seed, uint64_t out[64]) /* hvals[i].hi, hvals[i].lo for i<32 */
{
    mt_t m;
    mt_seed(&m, (uint32_t)seed);
    for (int i = 0; i < 32; ++i) {
        /* RepHash.cpp:11-12: (randInt()<<32) | randInt(); g++ evaluates the left operand first
           (checked against oracle/_ref, tests/test_sketch_oracle.py) */
        uint64_t a = mt_randInt(&m), b = mt_randInt(&m);
        out[2 * i] = (a << 32) | b;
        uint64_t c = mt_randInt(&m), d = mt_randInt(&m);
        out[2 * i + 1] = (c << 32) | d;
    }
}
static void rh_seed(rephash* r, int seed)
{
    uint64_t t[64];
    orc_rephash_table(seed, t);
    for (int i = 0; i < 32; ++i) { r->hvals[i].hi = t[2 * i]; r->hvals[i].lo = t[2 * i + 1]; }
    r->h.hi = r->h.lo = r->ht.hi = r->ht.lo = 0;
}
static void rh_setk(rephash* r, size_t k) /* RepHash::init(int) :45-51 */
{
    r->k = k;
    r->lastkmask = ((1ULL << k) - 1) << (64 - k);
    r->firstkmask = (1ULL << k) - 1;
}
static void rotl1(state_t* x) /* fastleftshift1 :68-72 */
{
    uint64_t last1 = x->hi & (1ULL << 63);
    x->hi = (x->hi << 1) | ((x->lo & (1ULL << 63)) >> 63);
    x->lo = (x->lo << 1) | (last1 >> 63);
}
static void rotr1(state_t* x) /* fastrightshift1 :74-78 */
{
    uint64_t first1 = x->hi & 1ULL;
    x->hi = (x->hi >> 1) | ((x->lo & 1ULL) << 63);
    x->lo = (x->lo >> 1) | (first1 << 63);
}
static void rotlk(const rephash* r, state_t* x) /* fastleftshiftk :56-60 */
{
    size_t k = r->k;
    uint64_t upper = x->hi & r->lastkmask;
    x->hi = (x->hi << k) | ((x->lo & r->lastkmask) >> (64 - k));
    x->lo = (x->lo << k) | (upper >> (64 - k));
}
static void rh_init(rephash* r, const char* s_) /* :85-97 */
{
    const unsigned char* s = (const unsigned char*)s_;
    r->h.hi = r->h.lo = r->ht.hi = r->ht.lo = 0;
    for (size_t i = 0; i < r->k; ++i) {
        rotl1(&r->h);
        r->h.hi ^= r->hvals[s[i] & 31].hi; r->h.lo ^= r->hvals[s[i] & 31].lo;
        rotl1(&r->ht);
        unsigned t = twin[s[r->k - 1 - i] & 31];
        r->ht.hi ^= r->hvals[t].hi; r->ht.lo ^= r->hvals[t].lo;
    }
}
static void rh_update(rephash* r, unsigned char out, unsigned char in) /* :99-113 */
{
    state_t z = r->hvals[out & 31];
    rotlk(r, &z);
    rotl1(&r->h);
    r->h.hi ^= z.hi; r->h.lo ^= z.lo;
    r->h.hi ^= r->hvals[in & 31].hi; r->h.lo ^= r->hvals[in & 31].lo;
    state_t zt = r->hvals[twin[in & 31]];
    rotlk(r, &zt);
    r->ht.hi ^= r->hvals[twin[out & 31]].hi; r->ht.lo ^= r->hvals[twin[out & 31]].lo;
    r->ht.hi ^= zt.hi; r->ht.lo ^= zt.lo;
    rotr1(&r->ht);
}
static uint64_t rh_hash(const rephash* r) { return r->h.lo ^ r->ht.lo; } /* :81-83 */

/* ---- StreamCounter (src/kmerstream/StreamCounter.hpp:23-356) ---------------- */
static size_t roundUpPowerOfTwo(size_t size) /* :11-21 */
{
    size--;
    size |= size >> 1; size |= size >> 2; size |= size >> 4; size |= size >> 8; size |= size >> 16; size |= size >> 32;
    size++;
    return size;
}
typedef struct {
    size_t MAX_TABLE, countWidth, countsPerLong;
    uint64_t maxVal;
    size_t size, F2size;
    uint64_t mask;
    uint64_t* table;
    uint64_t* F2table;
    size_t* M;
    size_t sumCount;
} streamcounter;

static void sc_init(streamcounter* s, double e) /* ctor :25-46 */
{
    s->MAX_TABLE = 32; s->maxVal = 15; s->countWidth = 4; s->countsPerLong = 16; s->sumCount = 0;
    size_t numcounts = (size_t)(48.0 / (e * e) + 1);
    s->F2size = roundUpPowerOfTwo((size_t)(2.0 / (e * e) + 1));
    s->F2table = (uint64_t*)calloc(s->F2size, sizeof(uint64_t));
    if (numcounts < 8192) numcounts = 8192;
    s->size = (numcounts + s->countsPerLong - 1) / s->countsPerLong;
    s->size = roundUpPowerOfTwo(s->size);
    s->mask = (s->size * s->countsPerLong) - 1;
    s->M = (size_t*)calloc(s->MAX_TABLE, sizeof(size_t));
    s->table = (uint64_t*)calloc(s->size * s->MAX_TABLE, sizeof(uint64_t));
}
static void sc_free(streamcounter* s) { free(s->table); free(s->F2table); free(s->M); }
static uint64_t sc_getVal(const streamcounter* s, size_t index, size_t w) /* :325-330 */
{
    size_t wordindex = w * s->size + (index / s->countsPerLong);
    size_t bitindex = index & (s->countsPerLong - 1);
    uint64_t bitmask = s->maxVal << (s->countWidth * bitindex);
    return (s->table[wordindex] & bitmask) >> (s->countWidth * bitindex);
}
static void sc_setVal(streamcounter* s, size_t index, size_t w, uint64_t val) /* :332-340 */
{
    if (val > s->maxVal) val = s->maxVal;
    size_t wordindex = w * s->size + (index / s->countsPerLong);
    size_t bitindex = index & (s->countsPerLong - 1);
    uint64_t bitmask = s->maxVal << (s->countWidth * bitindex);
    s->table[wordindex] = (((val & s->maxVal) << (s->countWidth * bitindex)) & bitmask) | (s->table[wordindex] & ~bitmask);
}
static uint64_t bitScanForward(uint64_t bb) /* lsb.cpp:4-29 */
{
    static const uint64_t index64[64] = {
        63, 0, 58, 1, 59, 47, 53, 2, 60, 39, 48, 27, 54, 33, 42, 3, 61, 51, 37, 40, 49, 18, 28, 20,
        55, 30, 34, 11, 43, 14, 22, 4, 62, 57, 46, 52, 38, 26, 32, 41, 50, 36, 17, 19, 29, 10, 13, 21,
        56, 45, 25, 31, 35, 16, 9, 12, 44, 24, 15, 8, 23, 7, 6, 5 };
    const uint64_t debruijn64 = 0x07EDD5E59A4E28C2ULL;
    return index64[((bb & (0 - bb)) * debruijn64) >> 58];
}
static void sc_add(streamcounter* s, uint64_t hashval) /* operator() :68-93 */
{
    s->sumCount++;
    ++s->F2table[hashval & (s->F2size - 1)];
    size_t w = bitScanForward(hashval);
    if (w >= s->MAX_TABLE) w = s->MAX_TABLE - 1;
    if (s->M[w] == s->size * s->countsPerLong * s->maxVal) return;
    uint64_t hval = hashval >> (w + 1);
    uint64_t index = hval & s->mask;
    uint64_t val = sc_getVal(s, index, w);
    if (val != s->maxVal) {
        sc_setVal(s, index, w, val + 1);
        s->M[w]++;
    }
}
static size_t sc_F0(const streamcounter* s) /* :114-140 */
{
    size_t R = s->size * s->countsPerLong;
    double sum = 0;
    int n = 0;
    double limit = 0.2;
    while (n == 0 && limit > 1e-8) {
        for (size_t i = 0; i < s->MAX_TABLE; i++) {
            size_t ts = 0;
            for (size_t j = 0; j < R; j++)
                if (sc_getVal(s, j, i) > 0) ts++;
            if (ts <= (1 - limit) * R && ts >= limit * R) {
                double est = (log(1.0 - ts / ((double)R)) / log(1.0 - 1.0 / R)) * pow(2.0, i + 1);
                sum += est;
                n++;
                break;
            }
        }
        limit = limit / 1.5;
    }
    return (size_t)(sum / n);
}
static size_t sc_f1(const streamcounter* s) /* :142-172 */
{
    size_t R = s->size * s->countsPerLong;
    double sum = 0;
    int n = 0;
    double limit = 0.2;
    while (n == 0 && limit > 1e-8) {
        for (size_t i = 0; i < s->MAX_TABLE; i++) {
            size_t r1 = 0, r0 = 0;
            for (size_t j = 0; j < R; j++) {
                uint64_t val = sc_getVal(s, j, i);
                if (val == 0) r0++;
                if (val == 1) r1++;
            }
            if ((r0 <= (1 - limit) * R) && (r0 >= limit * R)) {
                sum += (R - 1) * (r1 / ((double)r0)) * pow(2.0, i + 1);
                n++;
                break;
            }
        }
        limit = limit / 1.5;
    }
    return (size_t)(sum / n);
}
static size_t sc_F2(const streamcounter* s) /* :308-317 */
{
    double sum = 0, sqsum = 0;
    for (size_t i = 0; i < s->F2size; i++) {
        double c = (double)s->F2table[i];
        sum += c;
        sqsum += c * c;
    }
    return (size_t)(sqsum + (sqsum - sum * sum) / s->F2size);
}

/* ---- ReadQualityHasher (src/ReadQualityHasher.hpp:13-111) -------------------- */
typedef struct {
    size_t q_cutoff, q_base, k;
    rephash hf;
    streamcounter sc;
} rqh;

static void rqh_run(rqh* r, const char* s, size_t l, const char* q, size_t ql) /* operator() :30-68 */
{
    size_t i = 0, j = 0, k = r->k;
    int last_valid = 0;
    if (l < k) return;
    /* DEFINED: a record without qualities (ql != l) is skipped; the reference indexes q[j]
       out of bounds. */
    if (ql != l) return;
    while (j < l) {
        char c = s[j];
        if (c != 'N' && c != 'n' && (q[j] >= (char)(r->q_base + r->q_cutoff))) {
            if (last_valid) {
                rh_update(&r->hf, (unsigned char)s[i], (unsigned char)s[j]);
                i++; j++;
            } else {
                if (i + k - 1 == j) {
                    rh_init(&r->hf, s + i);
                    last_valid = 1;
                    j++;
                } else {
                    j++;
                }
            }
        } else {
            j++;
            i = j;
            last_valid = 0;
        }
        if (last_valid) sc_add(&r->sc, rh_hash(&r->hf));
    }
}

typedef struct { uint32_t n_q, n_k; rqh* h; uint32_t* qs; int32_t* ks; } sketchset;

void* orc_sketch_create(const bqc_sketch_options* so) /* Counts ctor, bamqualcheck.cpp:21-37 */
{
    sketchset* ss = (sketchset*)calloc(1, sizeof *ss);
    ss->n_q = so->n_q; ss->n_k = so->n_k;
    ss->h = (rqh*)calloc((size_t)so->n_q * so->n_k, sizeof(rqh));
    ss->qs = (uint32_t*)malloc(sizeof(uint32_t) * so->n_q);
    ss->ks = (int32_t*)malloc(sizeof(int32_t) * so->n_k);
    memcpy(ss->qs, so->qlist, sizeof(uint32_t) * so->n_q);
    memcpy(ss->ks, so->klist, sizeof(int32_t) * so->n_k);
    for (uint32_t i = 0; i < so->n_q; ++i)
        for (uint32_t j = 0; j < so->n_k; ++j) {
            rqh* r = &ss->h[i * so->n_k + j];
            r->q_base = 33; /* ProgramOptions: q_base(33) */
            r->q_cutoff = so->qlist[i];
            rh_seed(&r->hf, so->seed); /* ReadQualityHasher ctor :15-19 (seed != 0) */
            r->k = (size_t)so->klist[j];
            rh_setk(&r->hf, r->k);
            sc_init(&r->sc, so->e);
        }
    return ss;
}
void orc_sketch_destroy(void* p)
{
    sketchset* ss = (sketchset*)p;
    for (uint32_t i = 0; i < ss->n_q * ss->n_k; ++i) sc_free(&ss->h[i].sc);
    free(ss->h); free(ss->qs); free(ss->ks); free(ss);
}
void orc_sketch_run(void* p, const char* seq, size_t l, const char* qual, size_t ql) /* RunBamStream :113-122 */
{
    sketchset* ss = (sketchset*)p;
    for (uint32_t i = 0; i < ss->n_q * ss->n_k; ++i) rqh_run(&ss->h[i], seq, l, qual, ql);
}
uint32_t orc_sketch_results(void* p, bqc_sketch_counts* out)
{
    sketchset* ss = (sketchset*)p;
    for (uint32_t i = 0; i < ss->n_q; ++i)
        for (uint32_t j = 0; j < ss->n_k; ++j) {
            rqh* r = &ss->h[i * ss->n_k + j];
            bqc_sketch_counts* o = &out[i * ss->n_k + j];
            o->q = ss->qs[i]; o->k = (uint32_t)ss->ks[j];
            o->sumCount = r->sc.sumCount;
            o->F0 = sc_F0(&r->sc); o->f1 = sc_f1(&r->sc); o->F2 = sc_F2(&r->sc);
        }
    return ss->n_q * ss->n_k;
}

/* ---- standalone entry points used by the pinning tests ----------------------- */
/* hashes of every k-mer of s (no quality clipping): init at 0, then update */
uint32_t orc_rephash_sequence(int seed, int k, const char* s, uint32_t l, uint64_t* out)
{
    rephash r;
    memset(&r, 0, sizeof r);
    rh_seed(&r, seed);
    rh_setk(&r, (size_t)k);
    if (l < (uint32_t)k) return 0;
    rh_init(&r, s);
    uint32_t n = 0;
    out[n++] = rh_hash(&r);
    for (uint32_t j = (uint32_t)k; j < l; ++j) {
        rh_update(&r, (unsigned char)s[j - k], (unsigned char)s[j]);
        out[n++] = rh_hash(&r);
    }
    return n;
}
/* feed hash values to a fresh StreamCounter; res = {sumCount, F0, f1, F2} */
void orc_streamcounter_run(double e, const uint64_t* hashes, uint64_t n, uint64_t res[4])
{
    streamcounter s;
    sc_init(&s, e);
    for (uint64_t i = 0; i < n; ++i) sc_add(&s, hashes[i]);
    res[0] = s.sumCount; res[1] = sc_F0(&s); res[2] = sc_f1(&s); res[3] = sc_F2(&s);
    sc_free(&s);
}
