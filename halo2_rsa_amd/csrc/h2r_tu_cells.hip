// libh2r.so, translation unit "cells": every instantiation of cells_kernel (h2r_cells.hpp) -- the advice image written
// directly from the operands, one wave per mul_mod -- and its launcher.
#include "h2r_internal.hpp"

namespace h2r {

hipError_t launch_cells_shape(u32 w, bool mont, u32 nwv, u32 lds, const CellsArgs &ca, hipStream_t st, hipEvent_t ea, hipEvent_t eb) {
    auto go = [&](auto kernel, u32 threads) {
        if (lds > 48 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipGetLastError();
        }
        hipExtLaunchKernelGGL(kernel, dim3((unsigned)ca.n_items), dim3(threads), lds, st, ea, eb, 0, ca);
    };
    const bool w64 = w == 64;
    if (!mont) { if (w64) go(&cells_kernel<64>, 64); else go(&cells_kernel<32>, 64); }
    else if (nwv == 1) { if (w64) go(&cells_kernel<64, 0, true>, 64); else go(&cells_kernel<32, 0, true>, 64); }
    else { if (w64) go(&cells_kernel<64, 0, true, 8>, 512); else go(&cells_kernel<32, 0, true, 8>, 512); }
    return hipGetLastError();
}

}  // namespace h2r
