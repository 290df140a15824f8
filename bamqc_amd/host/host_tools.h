// host_tools.h — internal declarations shared by the host-side tooling sources.
#pragma once
#include "../../include/bamqc_host.h"

struct bqc_synth_owned { // batch first: bqc_synth_batch_free casts back
    bqc_batch batch;
    uint16_t* flag; uint8_t* mapq; uint8_t* lane; int32_t* rid; int32_t* pos; int32_t* tlen; int32_t* nm; int32_t* as;
    uint32_t* l_seq; uint16_t* n_cigar; uint8_t* seq; uint8_t* qual; uint32_t* cigar;
};
