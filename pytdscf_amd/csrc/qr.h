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

// The thin factorisation (next = 0) for a caller that does not need LAPACK's signs on diag(R) -- the sweep's gauge moves
// (a gauge freedom of the bond): gauge_free = true takes the block Gram-Schmidt / Cholesky path of
// qr_gram.hip (R with a positive diagonal, A left intact) where the shape qualifies and its conditioning checks pass, and
// qr_householder otherwise; gauge_free = false IS qr_householder.  MITDVP_QR_GRAM=0 switches the new path off.
void qr_thin(hipStream_t st, zc* A, int m, int n, zc* Q, zc* R, zc* work, long* nlaunch, SmallSync* sy, QrHistory* hist,
             bool gauge_free, bool* used_gauge_free = nullptr);
size_t qr_gram_work_elems(int m, int n);
int qr_gram(hipStream_t st, const zc* A, int m, int n, zc* Q, zc* R, zc* work, int* pub = nullptr, int pub_tag = 0);
int* qr_gram_flag(zc* work, int m, int n);

// panel factorisation: 1 = CholeskyQR2 + Householder reconstruction with the per-column kernels as the fallback
// (default), 0 = per-column kernels only
void qr_set_fast(int on);
int qr_get_fast();

}  // namespace mitdvp
