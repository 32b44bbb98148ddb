// cb_gemm kernels, tile instantiation <bf16, 256, 128, 2> (see gemm.hip / gemm_impl.h)
#include "gemm_impl.h"

namespace cbgemm {
template int launch_gemm<bf16, 256, 128, 2>(const GP&, bool, hipStream_t);
}
