// Instantiations of the generic implicit-GEMM kernel, group 1 of 3 (tile ids 6, 7, 8, 9, 10 of the table in igemm.hip).
#include "igemm_kernel.h"

namespace sdmi {

int launch_generic_tile_g1(int tile, const IGemmParams& p, bool dma, int splitk, hipStream_t stream) {
  switch (tile) {
    case 6: return launch_cfg<256, 128, 4, 2, 3>(p, dma, splitk, stream);
    case 7: return launch_cfg<128, 128, 2, 2, 3>(p, dma, splitk, stream);
    case 8: return launch_cfg<64, 128, 2, 2, 3>(p, dma, splitk, stream);
    case 9: return launch_cfg<128, 128, 4, 2, 3>(p, dma, splitk, stream);
    case 10: return launch_cfg<64, 64, 2, 2, 4>(p, dma, splitk, stream);
    default: return fail("not a tile of this group");
  }
}

}  // namespace sdmi
