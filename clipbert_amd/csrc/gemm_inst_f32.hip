// cb_gemm kernels, tile instantiation <float, 64, 64, 2> (see gemm.hip / gemm_impl.h)
#include "gemm_impl.h"

namespace cbgemm {
template int launch_gemm<float, 64, 64, 2>(const GP&, bool, hipStream_t);
template int launch_gemm_group<float, 64, 64, 2, 1>(const GroupArgs&, int, hipStream_t);
}
