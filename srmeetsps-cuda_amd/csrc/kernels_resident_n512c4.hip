// The 256 x 32 tile shape of the resident CG kernel with 512 threads per block (4 columns per thread, two waves per SIMD):
// same source, see kernels_resident.hip
#define SRPS_RES_NT 512
#define SRPS_RES_CPT 4
#define SRPS_RES_TAG 512c4
#include "kernels_resident.hip"
