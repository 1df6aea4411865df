// The 256 x 32 tile shape of the resident CG kernel (256 threads per block): same source, see kernels_resident.hip
#define SRPS_RES_NT 256
#include "kernels_resident.hip"
