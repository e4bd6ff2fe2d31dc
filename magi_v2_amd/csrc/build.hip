// placeholder until the Matern / Cholesky kernels land (next milestone)
#include "magi_internal.h"

int magi_build_matrices_device(magi_handle* h, const double*, int, int, const double*, const double*, double, int,
                               double*, double*, double*) {
    return magi_fail(h, MAGI_E_STATE, "magi_build_matrices: not implemented yet");
}

int magi_matern_blocks_device(magi_handle* h, const double*, int, double, double, double, double*, double*, double*) {
    return magi_fail(h, MAGI_E_STATE, "magi_matern_blocks: not implemented yet");
}
