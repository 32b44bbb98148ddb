// cb_gemm, 8-wave structure (gemm8_impl.h): tile 128x256, dgrad forms
#include "gemm8_impl.h"

namespace cbgemm {
template int launch_gemm8_dgrad<128, 256, 2, 4, 3>(const GP&, int, float*, hipStream_t);
}
