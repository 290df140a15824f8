// bamqualcheck — command-line front end; everything lives in libbamqc_gpu.so (bqc_main).
#include "../include/bamqc_host.h"
int main(int argc, const char** argv) { return bqc_main(argc, argv); }
