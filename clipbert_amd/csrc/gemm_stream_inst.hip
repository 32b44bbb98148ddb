// cb_gemm, streaming structure (gemm_stream_impl.h): the instantiations and their launcher
#include "gemm_stream_impl.h"

namespace cbgemm {
int launch_gemm_stream(const GP& p, int variant, hipStream_t st) {
    const int w = stream_workgroups_per_cu(variant);
    switch (variant) {
        case 0: return launch_gemm_stream_one<64, 256, 1, 2>(p, w, st);
        case 1: return launch_gemm_stream_one<64, 128, 2, 2>(p, w, st);
    }
    return cb_fail("cb_gemm (stream): bad variant %d", variant);
}
}
