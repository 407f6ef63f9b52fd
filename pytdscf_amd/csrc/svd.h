// svd.h -- one-sided Jacobi complex SVD on the GPU (see svd.hip)
#pragma once
#include "common.h"

namespace mitdvp {

size_t svd_work_elems(int r, int c);
// A (r x c, row-major, untouched) = U (r x k) diag(S) Vh (k x c), k = min(r, c);
// S (host, k values) in descending order.  `work` needs svd_work_elems(r, c) elements.
void svd_jacobi(hipStream_t st, const zc* A, int r, int c, zc* U, double* S_host, zc* Vh, zc* work, int* sweeps_out);

// rows of M (nr x nc, in place) -> orthogonal rows s_i q_i; S_host descending, idx_dev the
// permutation; `work` needs nr + 8 elements.
void svd_rows_us(hipStream_t st, zc* M, int nr, int nc, double* S_host, int* idx_dev, zc* work, int* sweeps_out);

}  // namespace mitdvp
