// The 256 x 16 tile shape of the resident CG kernel (256 threads per block, 4 columns per thread): same source, see
// kernels_resident.hip
#define SRPS_RES_NT 256
#define SRPS_RES_CPT 4
#define SRPS_RES_TAG 256c4
#include "kernels_resident.hip"
