/*
 * bamqc_oracle.h — entry points of the CPU restatement (TEST INFRASTRUCTURE
 * ONLY; see bamqc_oracle.c).  Same batch / counts structures as the product
 * ABI (include/bamqc.h) so tests feed both sides identical inputs.
 */
#ifndef BAMQC_ORACLE_H_
#define BAMQC_ORACLE_H_
#include "../include/bamqc.h"
#ifdef __cplusplus
extern "C" {
#endif
struct orc_ctx;
int orc_create(const bqc_options* opt, struct orc_ctx** out);
int orc_set_reference(struct orc_ctx* o, int32_t rid, const uint8_t* dna5, uint64_t len);
int orc_process_batch(struct orc_ctx* o, const bqc_batch* b); /* returns a BQC_ERR_* code */
int orc_finalize(struct orc_ctx* o, const bqc_counts** out);
void orc_destroy(struct orc_ctx* o);
int orc_write_bamqc(const bqc_counts* counts, const bqc_header_info* hdr, const char* path);

/* sketch_oracle.c */
void* orc_sketch_create(const bqc_sketch_options* so);
void orc_sketch_destroy(void* sk);
void orc_sketch_run(void* sk, const char* seq, size_t l, const char* qual, size_t ql);
uint32_t orc_sketch_results(void* sk, bqc_sketch_counts* out);
void orc_rephash_table(int seed, uint64_t out[64]);
uint32_t orc_rephash_sequence(int seed, int k, const char* s, uint32_t l, uint64_t* out);
void orc_streamcounter_run(double e, const uint64_t* hashes, uint64_t n, uint64_t res[4]);
#ifdef __cplusplus
}
#endif
#endif
