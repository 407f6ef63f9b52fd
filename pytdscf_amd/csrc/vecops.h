// vecops.h -- Krylov vector kernels, transposes, RNG (see vecops.hip)
#pragma once
#include "common.h"

namespace mitdvp {

constexpr int NPART = 256;  // partial sums written by every reduction kernel
constexpr int MAXK = 21;    // Krylov vectors kept: ndim (<= 20) + 1

constexpr int SMALL_VEC_EPT = 16;                 // elements per thread of the single-workgroup kernels
constexpr long SMALL_VEC_N = 1024L * SMALL_VEC_EPT;  // longest vector they take

struct Coefs {
  zc c[MAXK];
};

int vec_blocks(long n);
void vec_dot(hipStream_t st, const zc* x, const zc* y, long n, bool conj_x, zc* out_p /*[NPART]*/);
void vec_sumsq(hipStream_t st, const zc* x, long n, double* out_p /*[NPART]*/);
void vec_lanczos_update(hipStream_t st, zc* v, const zc* vm1, const zc* vm2 /*nullable*/, long n,
                        const zc* alpha_p, const double* betaprev_p, double* out_p);
// the three-term update on an UNNORMALISED basis (v: H u_l -> u_{l+1}); alpha_raw_p: partials of the raw dot <x | H u_l>,
// nrm_all: [MAXK][NPART] partials of |u_{j+1}|^2; no separate normalisation launch follows (vecops.hip)
void vec_lanczos_update_deferred(hipStream_t st, zc* v, const zc* ul, const zc* ulm1 /*nullable*/, long n, const zc* alpha_raw_p,
                                 const double* nrm_all, int l, bool orthodox, double eps, double* out_p);
// one Lanczos vector step (dot with x, three-term update, norm, normalisation) in one
// single-workgroup launch; n <= SMALL_VEC_N; same partial layout as the three separate kernels
void vec_lanczos_step_small(hipStream_t st, zc* w, const zc* x, const zc* vl, const zc* vm2 /*nullable*/, long n,
                            zc* alpha_p, const double* betaprev_p, double* nrm_p, double eps);
void vec_scale_inv_norm(hipStream_t st, zc* v, long n, const double* nrm_p, double eps);
void vec_multi_dot(hipStream_t st, const zc* V, long ldv, int k, const zc* v, long n, zc* out_p /*[k][NPART]*/);
void vec_arnoldi_update(hipStream_t st, zc* v, const zc* V, long ldv, int k, long n, const zc* h_p, double* out_p);
void vec_lincomb(hipStream_t st, zc* out /*nullable*/, const zc* V, long ldv, int k, const Coefs& c, long n,
                 double* nrm_p /*nullable*/);
void vec_axpby(hipStream_t st, zc* y, const zc* x, long n, zc a, zc b);
void vec_scale(hipStream_t st, zc* y, long n, zc a);
void vec_randn(hipStream_t st, zc* out, long n, uint64_t seed);
void set_identity(hipStream_t st, zc* out, int rows, int cols, long ld);
// dst[r][c] = a * src[r][c] (+ dst[r][c]) for c < cols; columns cols..zero_to-1 of dst are zeroed
// dst[0..n) = src[0..n): a kernel, not hipMemcpy -- the source or destination may belong to ANOTHER HIP runtime
// instance in this process (torch's), which this one cannot look up but whose addresses are valid on the device
void vec_copy_raw(hipStream_t st, zc* dst, const zc* src, size_t n);
// max |blk[r][s] - delta_rs| over an n x n block with leading dimension ld -> *out_dev (one double, overwritten)
void ident_deviation(hipStream_t st, const zc* blk, long ld, int n, double* out_dev);
// nblk blocks at once (block c at base + c * blk_stride, same ld and n), against multiples of the identity:
// lam_dev[c] = the block's first diagonal element, out_dev[c] = max |block - lam 1|
// mask: bit c set = block c is looked at (the others' outputs are meaningless)
void ident_deviation_multi(hipStream_t st, const zc* base, int nblk, long blk_stride, long ld, int n, double* out_dev,
                           zc* lam_dev, unsigned long long mask = ~0ull, bool clear = true);
void copy2d(hipStream_t st, zc* dst, long ldd, const zc* src, long lds, long rows, int cols, int zero_to, zc a,
            bool accumulate);
// Block lists of the K_eff apply with identity states skipped (Engine::keff_prepare): up to 64 blocks, scalars by value.
struct BlockList { int n; int idx[64]; zc f[64]; };
// dst[a][k][:] = src[a][idx[k]][:]  for k < bl.n  (blocks of `cols` elements; src has m_src, dst n_dst blocks per row a)
void gather_blocks(hipStream_t st, zc* dst, int n_dst, const zc* src, int m_src, long rows, int cols, const BlockList& bl);
// X[a][pos0 + k][:] = f[k] * sig[a][:]  for k < bl.n  (row stride ldx of X, rows x cols matrix sig)
void fill_scaled_blocks(hipStream_t st, zc* X, long ldx, int pos0, const zc* sig, long rows, int cols, const BlockList& bl);
// out[a][:] += sum_k f[k] * X[a][idx[k]][:] + both * sig[a][:]
void accum_scaled_blocks(hipStream_t st, zc* out, const zc* X, long ldx, const zc* sig, long rows, int cols, const BlockList& bl,
                         zc both);
// norm profiles for the adaptive-rank functional (plain device arrays, no partials)
void col_sumsq(hipStream_t st, const zc* x, long rows, int cols, double* out /*[cols]*/);
void row_sumsq(hipStream_t st, const zc* x, int rows, long cols, double* out /*[rows]*/);
void shell_sumsq(hipStream_t st, const zc* x, int n, double* out /*[n]*/);
void transpose_batched(hipStream_t st, const zc* in, zc* out, int rows, int cols, long ldi, long ldo, int batch,
                       long in_bs, long out_bs);
// Liouville space: diagonal (trace = false) or trace (true) over the n x n physical index
void phys_diag(hipStream_t st, const zc* C, zc* out, int dl, int n, int dr, bool trace);
// out[i0][i2][i1][i3] = in[i0][i1][i2][i3]
void permute_0213(hipStream_t st, const zc* in, zc* out, long n0, int n1, int n2, int n3);
void clock_probe(hipStream_t st, long iters, double* host_out3 /* shader cycles, 100 MHz ticks, dummy */);
// XCC / CU ids of the workgroups of one launch on a stream created with this CU mask (nullptr: an ordinary stream)
void where_probe(const uint32_t* mask, int nwords, int nblocks, size_t lds_bytes, int spin_us, int* host_out);
// out (C order, dims[0..4]) = in gathered with in_strides; map2 (device, nullable) replaces index 2
void permute5(hipStream_t st, const zc* in, zc* out, const int dims[5], const long in_strides[5], const int* map2);
void scale_cols(hipStream_t st, zc* x, long rows, int cols, long ld, const double* sc_dev);
void transpose_rev3(hipStream_t st, const zc* in, zc* out, int na, int nj, int ns);

}  // namespace mitdvp
