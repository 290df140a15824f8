// sketch.h — device side of the k-mer sketch (ReadQualityHasher / StreamCounter).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/bamqc.h"
#include "device_types.h"

struct SketchDevice;
SketchDevice* sketch_create(const bqc_sketch_options& so, uint32_t n_lanes, hipStream_t s, std::string& err);
void sketch_destroy(SketchDevice* sk);
void sketch_reset(SketchDevice* sk, hipStream_t s);
void sketch_process(SketchDevice* sk, const DevBatch& b, hipStream_t s);
uint64_t sketch_state_words(const SketchDevice* sk);
void sketch_state_export(SketchDevice* sk, uint64_t* dev_dst, hipStream_t s);
void sketch_state_import(SketchDevice* sk, const uint64_t* dev_src, hipStream_t s);
bool sketch_finalize(SketchDevice* sk, uint32_t lane, std::vector<bqc_sketch_counts>& out, hipStream_t s, std::string& err);
