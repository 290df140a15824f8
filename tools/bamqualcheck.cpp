// bamqualcheck — command-line front end; everything lives in libbamqc_gpu.so (bqc_main).
#include <stdlib.h>
#include "../include/bamqc_host.h"
int main(int argc, const char** argv)
{
    setenv("BQC_FAST_EXIT", "1", 0); // a finished run exits without tearing down the GPU context (see driver.cpp)
    return bqc_main(argc, argv);
}
