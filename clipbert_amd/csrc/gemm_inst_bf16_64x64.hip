// cb_gemm kernels, tile instantiation <bf16, 64, 64, 3> (see gemm.hip / gemm_impl.h)
#include "gemm_impl.h"

namespace cbgemm {
template int launch_gemm<bf16, 64, 64, 3>(const GP&, bool, hipStream_t);
template int launch_gemm_group<bf16, 64, 64, 3, 1>(const GroupArgs&, int, hipStream_t);
}
