// cb_gemm, 8-wave structure (gemm8_impl.h): tile 256x256, dgrad forms
#include "gemm8_impl.h"

namespace cbgemm {
template int launch_gemm8_dgrad<256, 256, 2, 4, 2>(const GP&, int, float*, hipStream_t);
}
