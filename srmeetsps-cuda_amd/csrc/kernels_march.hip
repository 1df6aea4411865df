// kernels_march.hip -- placeholder, filled in below
#include "srps_internal.h"

