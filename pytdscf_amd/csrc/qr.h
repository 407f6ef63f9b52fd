// qr.h -- GPU Householder QR (see qr.hip)
#pragma once
#include "common.h"
#include "small_site.h"

namespace mitdvp {

constexpr int QR_NB = 32;     // panel width

// workspace size in complex elements for an (m x n) factorisation
size_t qr_work_elems(int m, int n, int next = 0);
// A (m x n, row-major, ld = n, m >= n) is overwritten by the reflectors;
// Q (m x (n+next), ld = n+next) and R (n x n, ld = n, zero below the diagonal; may be
// null) are written.  next > 0 appends the first `next` columns of the orthogonal
// complement exactly as LAPACK's full-mode Q orders them (H_1..H_n applied to e_{n+1}..).
// sy: the engine's exchange state (granule buffer, abort / error words); with it every 32-column panel is
// factored by ONE persistent launch instead of 37 (nullptr: the per-column launches).
// hist: where the fast panels remember which shapes keep failing their conditioning checks (nullptr: every call tries
// them).  It belongs to the CALLER -- an engine -- and not to the process: two engines that must take identical
// decisions on identical data (the two ranks of a bond-sharded junction update) see identical histories only then.
struct QrHistory;
QrHistory* qr_history_new();
void qr_history_free(QrHistory* h);
void qr_householder(hipStream_t st, zc* A, int m, int n, zc* Q, zc* R, zc* work, long* nlaunch, int next = 0,
                    SmallSync* sy = nullptr, QrHistory* hist = nullptr);

// panel factorisation: 1 = CholeskyQR2 + Householder reconstruction with the per-column kernels as the fallback
// (default), 0 = per-column kernels only
void qr_set_fast(int on);
int qr_get_fast();

}  // namespace mitdvp
