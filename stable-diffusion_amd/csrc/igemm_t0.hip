// Instantiations of the generic implicit-GEMM kernel, group 0 of 3 (tile ids 0, 1, 2, 3, 4, 5 of the table in igemm.hip).
#include "igemm_kernel.h"

namespace sdmi {

int launch_generic_tile_g0(int tile, const IGemmParams& p, bool dma, int splitk, hipStream_t stream) {
  switch (tile) {
    case 0: return launch_cfg<128, 128, 2, 2, 2>(p, dma, splitk, stream);
    case 1: return launch_cfg<128, 64, 2, 2, 2>(p, dma, splitk, stream);
    case 2: return launch_cfg<64, 64, 2, 2, 2>(p, dma, splitk, stream);
    case 3: return launch_cfg<256, 128, 4, 2, 2>(p, dma, splitk, stream);
    case 4: return launch_cfg<128, 64, 2, 2, 3>(p, dma, splitk, stream);
    case 5: return launch_cfg<64, 64, 2, 2, 3>(p, dma, splitk, stream);
    default: return fail("not a tile of this group");
  }
}

}  // namespace sdmi
