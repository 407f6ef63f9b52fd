// common.h -- shared types for the MI355X TDVP engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <complex>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

namespace mitdvp {

typedef double2 zc;  // complex128, (x, y) = (re, im), interleaved like NumPy
typedef std::complex<double> hzc;

struct HipError : std::runtime_error {
  explicit HipError(const std::string& s) : std::runtime_error(s) {}
};
struct ArgError : std::runtime_error {
  explicit ArgError(const std::string& s) : std::runtime_error(s) {}
};
struct NotConverged : std::runtime_error {
  explicit NotConverged(const std::string& s) : std::runtime_error(s) {}
};

#define HIP_CHECK(expr)                                                                   \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      char _b[512];                                                                       \
      snprintf(_b, sizeof(_b), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,              \
               hipGetErrorString(_e));                                                    \
      throw ::mitdvp::HipError(_b);                                                       \
    }                                                                                     \
  } while (0)

__host__ __device__ inline zc zmake(double re, double im) { return make_double2(re, im); }
__host__ __device__ inline zc zmul(zc a, zc b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__host__ __device__ inline zc zconj(zc a) { return make_double2(a.x, -a.y); }
__host__ __device__ inline zc zadd(zc a, zc b) { return make_double2(a.x + b.x, a.y + b.y); }
__host__ __device__ inline zc zsub(zc a, zc b) { return make_double2(a.x - b.x, a.y - b.y); }

// ---------------------------------------------------------------- zgemm.hip
// 1/sqrt(x) and 1/x for positive normal x: hardware seed (v_rsq_f64 / v_rcp_f64) + two Newton steps.  The library
// forms (special cases, denormal scaling) cost ~400 cycles of dependent latency each; where a chain of them sits on
// the critical path of a single-workgroup kernel (Jacobi rotations, Householder reflectors) these are used instead.
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  y = y * (1.5 - h * y * y);
  y = y * (1.5 - h * y * y);
  return y;
}
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * (2.0 - x * y);
  y = y * (2.0 - x * y);
  return y;
}

struct ZgemmDesc {
  const zc* A;
  const zc* B;
  zc* C;
  int M, N, K;
  long lda, ldb, ldc;              // leading dimensions in complex elements
  long strideA, strideB, strideC;  // batch strides (0 = shared operand)
  int batch;
  int transA;  // 0: A stored [M][K]; 1: A stored [K][M]
  int conjA;
  int transB;  // 0: B stored [K][N]; 1: B stored [N][K]
  int conjB;
  zc alpha, beta;
  int tile_cfg;  // -1 auto; 0: 128x128 (4M) / 128x64 (3M), 1: 64x64, 2: 32x32
  int mode3m;    // -1 library default; 0: 4M product; 1: 3M (Karatsuba) product
  int ksplit;    // internal: K range per workgroup row in split-K launches (0 = off)
  // block-sparse A (the W stage of an apply with a finite-state-machine MPO): for row tile tm of the 64-row
  // tile grid, klist[tm * klist_stride] = number of K tiles (16 wide) holding a non-zero of A, followed by their
  // indices; nullptr = dense.  Requires tile_cfg 1, no transposes, K % 16 == 0.
  const int* klist;
  int klist_stride;
  // row address map of C: row r is stored at (r % rowmap_p) * rowmap_s1 + (r / rowmap_p) * rowmap_s2 instead of
  // r * ldc (rowmap_p = 0: off) -- lets A's rows be ordered so that its zero blocks line up with the tile grid
  int rowmap_p;
  long rowmap_s1, rowmap_s2;
  int rowmap_r0;  // added to the row index before the map (a GEMM over a row range of a larger mapped matrix)
  // A not transposed: logical row r of A is stored at row r + r / (arow_skip - 1) + 1, i.e. every arow_skip-th stored
  // row (rows 0, arow_skip, ...) is left out of the product (0 = off).  Stage S1 of an apply whose left environment
  // has an identity block in MPO-bond state 0: those rows of X are copies of psi.
  int arow_skip;
  // scheduling experiments of the MFMA kernel (MITDVP_ZGEMM_TUNE; 0 = the tuned default), see zgemm.hip
  int tune;
  // Reducing epilogue (zgemm_reduce; the "edge" form of an H_eff apply, engine.hip::heff_apply_edge): the product is
  // never stored.  Rows are pairs m = (u, x), x in [0, epi_xm), columns pairs n = (v, y), y in [0, epi_yn); a 64 x 64
  // tile holds epi_tu = 64 / epi_xm values of u and epi_tv = 64 / epi_yn values of v whole, and the kernel stores
  //   C[u * epi_su + v * epi_sv + i * epi_si] (+)= sum_{x, y} epi_w[i * epi_ldw + x * epi_yn + y] * T[(u, x)][(v, y)],  i < epi_di
  // (a 16 x (tu * tv) x (xm * yn) product on the matrix cores out of the tile kept in LDS).  M = nu * xm, N = nv * yn.
  const zc* epi_w;
  long epi_ldw, epi_su, epi_sv, epi_si;
  int epi_xm, epi_yn, epi_di, epi_acc;
  int epi_b4;  // internal: the 4 x 4 x 4 form of the reducing epilogue is available (set by zgemm_reduce)
  const zc* epi_wf;  // the same core in the fragment order of the unguarded 4 x 4 x 4 epilogue (zgemm_reduce_pack_core), or nullptr
  int epi_full;  // internal: its unguarded variant may run when the shapes are whole (MITDVP_EPI_FULL=0: never)
};
// C[b] = alpha * op(A[b]) * op(B[b]) + beta * C[b]   (row-major, complex128)
void zgemm(hipStream_t st, const ZgemmDesc& d);
inline ZgemmDesc zgemm_desc(const zc* A, const zc* B, zc* C, int M, int N, int K) {
  ZgemmDesc d{};
  d.A = A; d.B = B; d.C = C; d.M = M; d.N = N; d.K = K;
  d.lda = K; d.ldb = N; d.ldc = N; d.batch = 1;
  d.alpha = make_double2(1.0, 0.0); d.beta = make_double2(0.0, 0.0);
  d.tile_cfg = -1;
  d.mode3m = -1;
  d.ksplit = 0;
  d.klist = nullptr; d.klist_stride = 0; d.rowmap_p = 0; d.rowmap_s1 = 0; d.rowmap_s2 = 0; d.rowmap_r0 = 0;
  d.arow_skip = 0;
  d.tune = -1;
  d.epi_w = nullptr; d.epi_ldw = d.epi_su = d.epi_sv = d.epi_si = 0; d.epi_xm = d.epi_yn = d.epi_di = d.epi_acc = 0; d.epi_b4 = 0; d.epi_full = 0; d.epi_wf = nullptr;
  return d;
}
// the product with the reducing epilogue described at ZgemmDesc::epi_w (NN or NT operands, no batch); false when the
// shape does not qualify (the caller takes another path)
bool zgemm_reduce_ok(int xm, int yn, int di);
// 1 when the 4 x 4 x 4 form of the reducing epilogue is in use (lane maps verified on this device, MITDVP_EPI_B4 != 0)
int zgemm_reduce_b4_available(hipStream_t st);
// the core of a reducing product in the fragment order of the unguarded 4 x 4 x 4 epilogue (ZgemmDesc::epi_wf); false: shape not served
bool zgemm_reduce_full_ok(hipStream_t st, int xm, int yn, int di);
bool zgemm_reduce_pack_core(hipStream_t st, const zc* w, long ldw, int di, int kp, zc* wf);
void zgemm_reduce(hipStream_t st, const ZgemmDesc& d);
int zgemm_default_mode();
void zgemm_set_default_mode(int m);
double mfma_peak_probe(hipStream_t st);
// frees the split-K workspace that belongs to a stream (call before destroying the stream)
void zgemm_release_stream(hipStream_t st);
void mfma_layout_probe(hipStream_t st, int* host_out);
int zgemm_cd_mode(hipStream_t st);  // C/D lane map of v_mfma_f64_16x16x4_f64 (probed once): 0: row = (lane>>4) + 4*reg, 1: 4*(lane>>4) + reg

}  // namespace mitdvp
