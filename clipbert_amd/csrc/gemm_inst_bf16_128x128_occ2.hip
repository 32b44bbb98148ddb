// cb_gemm kernels, tile instantiation <bf16, 128, 128, PF = 1, 2 blocks per CU> (see gemm.hip / gemm_impl.h): the 128x128 tile
// with its registers capped at 256 per lane and one register stage instead of two, so that TWO blocks share a CU
// (the PF = 2 instantiation needs 260-310 registers for every form but the plain forward one: one block per CU)
#include "gemm_impl.h"

namespace cbgemm {
template int launch_gemm<bf16, 128, 128, 1, 2>(const GP&, bool, hipStream_t);
template int launch_gemm_group<bf16, 128, 128, 1, 2>(const GroupArgs&, int, hipStream_t);
}
