// cb_gemm, 8-wave structure (gemm8_impl.h): tile 256x128, dgrad forms
#include "gemm8_impl.h"

namespace cbgemm {
template int launch_gemm8_dgrad<256, 128, 4, 2, 3>(const GP&, int, float*, hipStream_t);
}
