// The 256 x 16 tile shape of the resident CG kernel with 512 threads per block (2 columns per thread, two waves per SIMD; sf 1
// and 2 only: a thread's columns hold whole sf x sf blocks): same source, see kernels_resident.hip
#define SRPS_RES_NT 512
#define SRPS_RES_CPT 2
#define SRPS_RES_TAG 512c2
#include "kernels_resident.hip"
