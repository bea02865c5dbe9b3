// Instantiations of the generic implicit-GEMM kernel, group 2 of 3 (tile ids 11, 12, 13, 18, 19, 20, 21 of the table in igemm.hip).
#include "igemm_kernel.h"

namespace sdmi {

int launch_generic_tile_g2(int tile, const IGemmParams& p, bool dma, int splitk, hipStream_t stream) {
  switch (tile) {
    case 11: return launch_cfg<128, 256, 2, 4, 2>(p, dma, splitk, stream);
    case 12: return launch_cfg<64, 256, 1, 4, 3>(p, dma, splitk, stream);
    case 13: return launch_cfg<256, 64, 4, 1, 3>(p, dma, splitk, stream);
    case 18: return launch_cfg<64, 64, 2, 2, 8>(p, dma, splitk, stream);
    case 19: return launch_cfg<64, 128, 2, 2, 6>(p, dma, splitk, stream);
    case 20: return launch_cfg<128, 64, 2, 2, 6>(p, dma, splitk, stream);
    case 21: return launch_cfg<128, 128, 4, 2, 4>(p, dma, splitk, stream);
    default: return fail("not a tile of this group");
  }
}

}  // namespace sdmi
