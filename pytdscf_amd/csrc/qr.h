// qr.h -- GPU Householder QR (see qr.hip)
#pragma once
#include "common.h"

namespace mitdvp {

constexpr int QR_NB = 32;     // panel width

// workspace size in complex elements for an (m x n) factorisation
size_t qr_work_elems(int m, int n);
// A (m x n, row-major, ld = n, m >= n) is overwritten by the reflectors;
// Q (m x n, ld = n) and R (n x n, ld = n, zero below the diagonal) are written.
void qr_householder(hipStream_t st, zc* A, int m, int n, zc* Q, zc* R, zc* work, long* nlaunch);

}  // namespace mitdvp
