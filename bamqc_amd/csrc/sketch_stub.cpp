// Placeholder until the HIP sketch (N1) lands: creating a context with sketch options fails loudly.
#include "sketch.h"
SketchDevice* sketch_create(const bqc_sketch_options&, uint32_t, hipStream_t, std::string& err) { err = "k-mer sketch not built into this library yet"; return nullptr; }
void sketch_destroy(SketchDevice*) {}
void sketch_reset(SketchDevice*, hipStream_t) {}
void sketch_process(SketchDevice*, const DevBatch&, hipStream_t) {}
uint64_t sketch_state_words(const SketchDevice*) { return 0; }
void sketch_state_export(SketchDevice*, uint64_t*, hipStream_t) {}
void sketch_state_import(SketchDevice*, const uint64_t*, hipStream_t) {}
bool sketch_finalize(SketchDevice*, uint32_t, std::vector<bqc_sketch_counts>&, hipStream_t, std::string&) { return true; }
