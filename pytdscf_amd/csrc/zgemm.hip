// zgemm.hip -- complex128 GEMM on the CDNA4 FP64 matrix cores.
//
// Every contraction on the TDVP hot path (H_eff / K_eff applies, environment
// updates, bond absorption, the block-reflector updates inside QR) is a
// row-major complex GEMM, so this kernel is the dominant kernel of the engine.
//
// Design (gfx950):
//   * v_mfma_f64_16x16x4_f64.  Complex product either "4M" (Cre += Are*Bre - Aim*Bim ; Cim += Are*Bim + Aim*Bre:
//     4 real MFMAs per tile step = the 8 flop per complex MAC SURVEY 8(d) counts) or, by default, "3M" (Karatsuba:
//     P1 = Ar Br, P2 = Ai Bi, P3 = (Ar + Ai)(Br + Bi); Re = P1 - P2, Im = P3 - P1 - P2: three accumulator sets,
//     25 % less matrix-core work).
//   * operands stay interleaved (re,im) in HBM and LDS: one lane's A (or B) operand of a tile step is ONE 16-byte
//     element = one ds_read_b128, so transposed / conjugated operand forms cost nothing extra.
//   * 256-thread workgroups, 2x2 waves, each wave owns WMxWN 16x16 blocks.  Default tile 64x64 (WM = WN = 2, 96
//     accumulator registers in 3M) with TWO workgroups per CU -- two waves per SIMD fill each other's barrier and
//     latency bubbles; a 128-row tile (one wave per SIMD) and a 32x32 tile for small outputs exist beside it.
//   * the K loop holds nothing but MFMAs, the 3M sums and memory instructions: FP64 MFMA and FP64 VALU share the
//     datapath on this chip, so every other vector instruction between two MFMAs is matrix time lost.  Staging loads
//     are BUFFER loads (scalar descriptor = tile origin, advanced per K tile on the scalar unit; constant 32-bit
//     per-thread offsets; out-of-range offsets return zero, which fills the K tail), LDS offsets are one base +
//     compile-time strides, conjugation rides on the fused multiply-add of the 3M sums.
//   * register-staged global->LDS prefetch, two staging register sets alternating by tile parity: the loads of tile
//     k+2 are issued first in tile k, the LDS stores of tile k+1 after them, one barrier per tile, two LDS stages.
//   * XCD-aware block remap + grouped tile order so that the blocks sharing an L2 walk neighbouring tiles.
#include <algorithm>
#include <cstdlib>

#include <type_traits>

#include "common.h"

#include <map>
#include <mutex>

namespace mitdvp {

extern __shared__ zc smem_dyn[];

typedef double d4 __attribute__((ext_vector_type(4)));

// lane maps of v_mfma_f64_4x4x4_4b_f64, measured with one-hot operands (tools/probes/mfma_4x4x4_layout.hip, MI355X):
// block = (lane / 4) % 4 for both operands and the result; A operand: row i = lane % 4, k = lane / 16; B operand: column
// j = lane % 4, k = lane / 16; result: row i = lane / 16, column j = lane % 4.  Verified again at first use (b4_layout_ok).
#define MITDVP_B4_BLK(lane) (((lane) >> 2) & 3)
#define MITDVP_B4_Q(lane) ((lane) & 3)
#define MITDVP_B4_K(lane) ((lane) >> 4)
#define MITDVP_B4_D_ROW(lane) ((lane) >> 4)
#define MITDVP_B4_D_COL(lane) ((lane) & 3)

// C/D lane map of v_mfma_f64_16x16x4_f64, detected once by mfma_layout_probe():
//   mode 0: row = (lane>>4) + 4*reg     (cdna_hip_programming.md section 3)
//   mode 1: row = 4*(lane>>4) + reg     (the f32 16x16x4 map)
static int g_cd_mode = -1;

__device__ __forceinline__ double flip_sign(double x, unsigned mask) {
  return __hiloint2double(__double2hiint(x) ^ (int)mask, __double2loint(x));
}

__device__ __forceinline__ int cd_row(int mode, int lk, int r) {
  return mode == 0 ? (lk + 4 * r) : (4 * lk + r);
}

// The reducing epilogue of zgemm_reduce for RB x CB blocks of 16 x 16 outputs (declared here, defined below the kernel)
template <int RB, int CB>
__device__ __forceinline__ void reduce_epilogue(const ZgemmDesc& d, zc* smem, int tm, int tn, int cd_mode);
template <int RB, int CB>
__device__ __forceinline__ void reduce_epilogue_blocks(const ZgemmDesc& d, zc* smem, int tm, int tn, int cd_mode);
template <int NIS, int NPG, bool M3, bool FULL, class ST>
__device__ __forceinline__ void reduce_epilogue_b4(const ZgemmDesc& d, zc* smem, int tm, int tn, ST&& store_tile);
template <bool M3>
__device__ __forceinline__ void reduce_epilogue_b4w(const ZgemmDesc& d, zc* smem, int tm, int tn);

template <int WM, int WN, int BK, bool TA, bool TB, bool M3, bool SP = false, bool EPI = false>
__global__ __launch_bounds__(256, (WM * WN <= 4 ? 2 : 1)) void zgemm_kernel(ZgemmDesc d, int ntm, int ntn, int cd_mode) {
  constexpr int BM = 2 * WM * 16, BN = 2 * WN * 16;
  constexpr int LDAS = TA ? BM : (BK + 1);  // LDS row stride (complex elements)
  constexpr int LDBS = TB ? (BK + 1) : BN;
  constexpr int A_SZ = TA ? BK * BM : BM * (BK + 1);
  constexpr int B_SZ = TB ? BN * (BK + 1) : BK * BN;
  constexpr int A_PT = (BM * BK) / 256, B_PT = (BN * BK) / 256;
  static_assert((BM * BK) % 256 == 0 && (BN * BK) % 256 == 0, "tile/threads");
  // two LDS stages (dynamic LDS: cfg 0 needs > 64 KiB), one barrier per K tile
  constexpr int STAGE = A_SZ + B_SZ;
  zc* smem = smem_dyn;

  // ---- block -> tile (XCD remap, then grouped order) ----------------------
  const int nwg = ntm * ntn;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  constexpr int G = 8;
  const int per_group = G * ntn;
  const int group = bid / per_group;
  const int first_m = group * G;
  const int gsz = min(ntm - first_m, G);
  const int rem = bid - group * per_group;
  const int tm = first_m + rem % gsz;
  const int tn = rem / gsz;
  // EPI: a tile holds whole (u, x) / (v, y) groups only, i.e. 64 / xm * xm rows and 64 / yn * yn columns of it are used
  const int m0 = EPI ? tm * ((BM / d.epi_xm) * d.epi_xm) : tm * BM, n0 = EPI ? tn * ((BN / d.epi_yn) * d.epi_yn) : tn * BN;
  const int b = blockIdx.y;

  const long lda = d.lda, ldb = d.ldb;
  const zc* __restrict__ A = d.A + (long)b * d.strideA;
  const zc* __restrict__ B = d.B + (long)b * d.strideB;
  zc* __restrict__ C = d.C + (long)b * d.strideC;
  const int M = d.M, N = d.N;
  int K = d.K;
  if (d.ksplit > 0) {  // split-K: "batch" b owns k in [b*ksplit, (b+1)*ksplit)
    const long kb = (long)b * d.ksplit;
    A = d.A + (TA ? kb * lda : kb);
    B = d.B + (TB ? kb : kb * ldb);
    K = min(d.ksplit, d.K - (int)kb);
  }

  // Two workgroups share a CU, i.e. two waves running this same program share every SIMD; left alone they tend to
  // meet in the same phase (matrix work beside matrix work, then both at their loads and barrier).
  //   tune bit 0: static priority for the second-dispatched workgroup of a CU
  //   tune bit 1: start it half a K tile late
  //   tune bit 2: pair by block parity instead of dispatch round
  if (d.tune > 0) {
    const bool second = (d.tune & 4) ? (blockIdx.x & 1) : ((blockIdx.x >> 8) & 1);
    if (second) {
      if (d.tune & 1) __builtin_amdgcn_s_setprio(1);
      if (d.tune & 2) {
        __builtin_amdgcn_s_sleep(24);
        __builtin_amdgcn_s_sleep(24);
      }
    }
  }
  const int t = threadIdx.x;
  const int lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int li = lane & 15, lk = lane >> 4;

  // MITDVP_ZG_SETS (compile time): 1 = one register set per operand (the loads of tile k+2 are issued after the LDS
  // stores of tile k+1 freed it: ~0.7 tile between a load and its use), 2 = two sets alternating by tile parity (loads of
  // tile k+2 go out FIRST in tile k, into the set tile k's data left; 1.25 tiles between a load and its use)
#ifndef MITDVP_ZG_SETS
#define MITDVP_ZG_SETS 2
#endif
  constexpr int NSET = MITDVP_ZG_SETS;
#ifndef MITDVP_ZG_STSLOT
#define MITDVP_ZG_STSLOT 2
#endif
  typedef double v2d __attribute__((ext_vector_type(2)));  // one complex element as a register pair (plain loads / stores)
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  v2d ra[NSET][A_PT], rb[NSET][B_PT];

  // Load addresses = uniform base of the tile (scalar registers; one K tile further per tile) + a per-thread byte
  // offset that never changes (32 bits: a tile's rows span BM * lda elements, checked on the host).  Rows / columns
  // beyond M / N are clamped to the last valid one (their products land in output elements that are never stored);
  // elements beyond K are buffer loads with an out-of-range offset, which return zero: LDS stores need no mask.
  unsigned oa[A_PT], ob[B_PT];
  // k index inside the tile: base (per thread) + compile-time offset per element
  const int kbA = TA ? t / BM : t % BK;
  const int kbB = TB ? t % BK : t / BN;
  constexpr int KSA = TA ? 256 / BM : 0;
  constexpr int KSB = TB ? 0 : 256 / BN;
  static_assert(256 % BK == 0 && 256 % BM == 0 && 256 % BN == 0, "thread/tile mapping");
  // LDS element offsets inside a stage: affine in the element number p
  const int la0 = TA ? (t / BM) * LDAS + t % BM : (t / BK) * LDAS + t % BK;
  const int lb0 = A_SZ + (TB ? (t / BK) * LDBS + t % BK : (t / BN) * LDBS + t % BN);
  constexpr int LSA = TA ? (256 / BM) * LDAS : (256 / BK) * LDAS;
  constexpr int LSB = TB ? (256 / BK) * LDBS : (256 / BN) * LDBS;
  // stored row of A's row m (the identity-block shortcut skips one stored row in every arow_skip)
  auto arow = [&](int gm) { return (!TA && d.arow_skip > 1) ? (long)gm + gm / (d.arow_skip - 1) + 1 : (long)gm; };
  const char* const baseA = reinterpret_cast<const char*>(A + (TA ? (long)m0 : arow(m0) * lda));
  const char* const baseB = reinterpret_cast<const char*>(B + (TB ? (long)n0 * ldb : (long)n0));
#pragma unroll
  for (int p = 0; p < A_PT; ++p) {
    const int e = t + p * 256;
    int m, k;
    if (TA) { k = e / BM; m = e % BM; } else { m = e / BK; k = e % BK; }
    const int gm = min(m0 + m, M - 1);
    oa[p] = (unsigned)((TA ? (long)k * lda + (gm - m0) : (arow(gm) - arow(m0)) * lda + k) * (long)sizeof(zc));
  }
#pragma unroll
  for (int p = 0; p < B_PT; ++p) {
    const int e = t + p * 256;
    int n, k;
    if (TB) { n = e / BK; k = e % BK; } else { k = e / BN; n = e % BN; }
    const int gn = min(n0 + n, N - 1);
    ob[p] = (unsigned)((TB ? (long)(gn - n0) * ldb + k : (long)k * ldb + (gn - n0)) * (long)sizeof(zc));
  }
  const long stepA = (TA ? (long)BK * lda : (long)BK) * (long)sizeof(zc);  // bytes per K tile
  const long stepB = (TB ? (long)BK : (long)BK * ldb) * (long)sizeof(zc);

  // side work of the pipeline, one element at a time so that it can be spread
  // between the MFMAs of a tile: item < NP stores element `item` of the tile held
  // in registers into LDS stage `stage` (kv = number of valid k in that tile);
  // item >= NP loads element item-NP of the following tile (kv2 valid k).
  constexpr int NP = A_PT + B_PT;
  // SP: the K tiles to visit come from a list (block-sparse A); `ord` = position in that list of the tile being
  // loaded.  K % BK == 0 there, so a listed tile is valid as a whole and an unlisted position loads nothing.
  // (read through the constant address space: the list is written by the host before the launch, and a uniform load
  // from there is a SCALAR load -- the tile number lands in a scalar register and the buffer descriptor built from it
  // stays uniform; through a plain global pointer hipcc loads it per lane and wraps every staging load in a waterfall loop)
  typedef __attribute__((address_space(4))) const int cint4;
  const cint4* kl = nullptr;
  int nlist = 0;
  if (SP) {
    kl = (const cint4*)(unsigned long)(d.klist + (long)tm * d.klist_stride);
    nlist = kl[0];
    kl += 1;
  }
  // MITDVP_ABLATE (compile time, TIMING EXPERIMENTS ONLY -- the product is wrong with any bit set): 8 no barrier in the
  // K loop, 16 no global loads in the K loop, 32 no LDS stores in the K loop, 64 no 3M sums (P3 = Ar Br).  A second library built with
  // -DMITDVP_ABLATE=n (make ABLATE=n) and loaded through MITDVP_LIB; never the shipped one.
#ifndef MITDVP_ABLATE
#define MITDVP_ABLATE 0
#endif
  constexpr bool abl_nobar = (MITDVP_ABLATE & 8) != 0, abl_noload = (MITDVP_ABLATE & 16) != 0, abl_nostore = (MITDVP_ABLATE & 32) != 0;
  // ss = register set stored from, sl = register set loaded into
  auto side = [&](int item, zc* stage, int kv2, int ord, auto SS, auto SL, auto FULL) __attribute__((always_inline)) {
    constexpr int ss = decltype(SS)::value, sl = decltype(SL)::value;
    constexpr bool full = decltype(FULL)::value;
    if (item < NP && abl_nostore && stage != smem) return;
    if (item >= NP && abl_noload && ord >= 2) return;
    if (item < A_PT) {
      *reinterpret_cast<v2d*>(&stage[la0 + item * LSA]) = ra[ss][item];
    } else if (item < NP) {
      *reinterpret_cast<v2d*>(&stage[lb0 + (item - A_PT) * LSB]) = rb[ss][item - A_PT];
    } else if (item < 2 * NP) {
      // buffer loads: scalar descriptor of the tile (base = uniform tile origin, rebuilt per tile in scalar registers)
      // + the thread's constant 32-bit offset; an offset beyond the descriptor's range returns ZERO without touching
      // memory, which is how elements beyond K are filled (general form: one select per load; FULL: none); the
      // out-of-range offset equals the record count, so that offset + 15 cannot wrap around 32 bits
      const bool isA = item < NP + A_PT;
      const int p = isA ? item - NP : item - NP - A_PT;
      const bool any = full || (SP ? ord < nlist : kv2 > 0);
      const long tk = SP ? (long)kl[any ? ord : 0] : (long)ord;
      const bool ok = full || (SP ? any : (isA ? kbA + p * KSA : kbB + p * KSB) < kv2);
      const char* base = isA ? baseA + tk * stepA : baseB + tk * stepB;
      const unsigned off = isA ? oa[p] : ob[p];
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0xFFFFFFF0u, 0x00020000);
      const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : 0xFFFFFFF0u, 0, 0);
      if (isA) ra[sl][p] = __builtin_bit_cast(v2d, v);
      else rb[sl][p] = __builtin_bit_cast(v2d, v);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, NSET - 1>;
  using GEN = std::false_type;  // loads that may reach beyond K (prologue, last tiles)

  // Accumulators are born as MFMA results (0*0 + 0) so that they live in the
  // AGPR half of the register file for the whole K loop: with a plain constant
  // initialiser hipcc (ROCm 7.2) routes the loop-carried values through VGPR
  // phis and copies all of them AGPR<->VGPR on every K tile.
  double zin = 0.0;
  asm volatile("" : "+v"(zin));
  // 4M: acc0 = Re, acc1 = Im.   3M (Karatsuba): acc0 = sum Ar*Br, acc1 = sum Ai*Bi,
  // acc2 = sum (Ar+Ai)*(Br+Bi);  Re = acc0 - acc1, Im = acc2 - acc0 - acc1.
  constexpr int NACC = M3 ? 3 : 2;
  d4 acc[NACC][WM][WN];
#pragma unroll
  for (int q = 0; q < NACC; ++q)
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
        acc[q][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(zin, zin, (d4){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);

  const unsigned sa_mask = d.conjA ? 0x80000000u : 0u;
  const unsigned sb_mask = d.conjB ? 0x80000000u : 0u;
  const double sgnA = d.conjA ? -1.0 : 1.0, sgnB = d.conjB ? -1.0 : 1.0;
  constexpr int NK4 = BK / 4;
  constexpr int SLOTS = NK4 * WM;                       // MFMA groups per tile
  // side items after each MFMA group: front-loaded (twice the even share) so that
  // the global loads of tile k+2 are in flight for most of tile k
  constexpr int PER_SLOT = 2 * ((2 * NP + SLOTS - 1) / SLOTS);
  // two register sets: first MFMA group after which the LDS stores start (the loads take NP / PER_SLOT groups)
  constexpr int ST_SLOT = (MITDVP_ZG_STSLOT * PER_SLOT >= NP && MITDVP_ZG_STSLOT * PER_SLOT + NP <= SLOTS * PER_SLOT)
                              ? MITDVP_ZG_STSLOT : (NP + PER_SLOT - 1) / PER_SLOT;

  // LDS -> register fragments of k-step k4 of a stage
  auto ldfrag = [&](const zc* st, int k4, zc (&a)[WM], zc (&bb)[WN]) __attribute__((always_inline)) {
    const int kk = k4 * 4 + lk;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int row = (wm * WM + i) * 16 + li;
      a[i] = st[TA ? kk * LDAS + row : row * LDAS + kk];
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int col = (wn * WN + j) * 16 + li;
      bb[j] = st[A_SZ + (TB ? col * LDBS + kk : kk * LDBS + col)];
    }
  };

  // (MITDVP_ZGEMM_TUNE bits 4 / 5 -- TIMING EXPERIMENTS ONLY, the product is wrong: 16 = the reducing epilogue's products
  // skipped, 32 = the K loop skipped)
  const int nkt = (EPI && (d.tune & 32)) ? 0 : (SP ? nlist : (K + BK - 1) / BK);
  // prologue: tile 0 -> LDS stage 0, tile 1 -> registers (set 1 of two)
#pragma unroll
  for (int it = NP; it < 2 * NP; ++it) side(it, nullptr, K, 0, I0{}, I0{}, GEN{});
#pragma unroll
  for (int it = 0; it < NP; ++it) side(it, smem, 0, 0, I0{}, I0{}, GEN{});
#pragma unroll
  for (int it = NP; it < 2 * NP; ++it) side(it, nullptr, K - BK, 1, I0{}, I1{}, GEN{});
  __syncthreads();

  // one K tile; PAR = parity of kt when two register sets alternate (compile-time register indices)
  auto tile = [&](int kt, auto PAR, auto FULL) __attribute__((always_inline)) {
    constexpr int par = decltype(PAR)::value;
    using SL = std::integral_constant<int, NSET == 2 ? par : 0>;        // free set: tile kt+2 is loaded into it
    using SS = std::integral_constant<int, NSET == 2 ? 1 - par : 0>;    // holds tile kt+1: stored to the other stage
    const zc* st = smem + (kt & 1) * STAGE;
    zc* nst = smem + ((kt + 1) & 1) * STAGE;
    // valid k of the tile to load now (<= 0: no such tile)
    const int kv2 = SP ? 0 : K - (kt + 2) * BK;
    zc fa[2][WM], fb[2][WN];
    ldfrag(st, 0, fa[0], fb[0]);
#pragma unroll
    for (int k4 = 0; k4 < NK4; ++k4) {
      if (k4 + 1 < NK4) ldfrag(st, k4 + 1, fa[(k4 + 1) & 1], fb[(k4 + 1) & 1]);
      zc a[WM], bb[WN];
      // conjugation = flip the sign bit of the imaginary part (one 32-bit xor
      // instead of an f64 multiply on the VALU port the MFMAs share)
      // (every vector instruction between the MFMAs takes matrix-pipe issue time on this chip: 4M flips the sign
      // bit of the imaginary parts -- one 32-bit xor each; 3M needs no flip at all: with sA, sB = -1 for a conjugated
      // operand the products are P1 = Ar Br, P2 = Ai Bi (unsigned), P3 = (Ar + sA Ai)(Br + sB Bi), and
      // Re = P1 - sA sB P2, Im = P3 - P1 - sA sB P2 -- the signs ride on the fused multiply-add that forms the sums)
#pragma unroll
      for (int i = 0; i < WM; ++i) { a[i] = fa[k4 & 1][i]; if (!M3) a[i].y = flip_sign(a[i].y, sa_mask); }
#pragma unroll
      for (int j = 0; j < WN; ++j) { bb[j] = fb[k4 & 1][j]; if (!M3) bb[j].y = flip_sign(bb[j].y, sb_mask); }
      double as[WM], bs[WN];
      if (M3) {
#pragma unroll
        for (int i = 0; i < WM; ++i) as[i] = (MITDVP_ABLATE & 64) ? a[i].x : __builtin_fma(sgnA, a[i].y, a[i].x);
#pragma unroll
        for (int j = 0; j < WN; ++j) bs[j] = (MITDVP_ABLATE & 64) ? bb[j].x : __builtin_fma(sgnB, bb[j].y, bb[j].x);
      }
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        if (M3) {
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, bb[j].x, acc[0][i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, bb[j].y, acc[1][i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[NACC - 1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[i], bs[j], acc[NACC - 1][i][j], 0, 0, 0);
        } else {
          const double nai = -a[i].y;
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, bb[j].x, acc[0][i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, bb[j].y, acc[1][i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(nai, bb[j].y, acc[0][i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, bb[j].x, acc[1][i][j], 0, 0, 0);
        }
        // side work spread over the MFMA groups (sched_barrier pins the order: left alone, hipcc clusters all
        // memory operations at the top of the tile, where nothing overlaps them).  One register set: first all
        // LDS stores of tile kt+1 (the other stage), then the global loads of tile kt+2 into the registers they
        // freed.  Two sets: the loads of tile kt+2 first, into the free set, then the stores of tile kt+1.
        const int slot = k4 * WM + i;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < PER_SLOT; ++q) {
          int item = slot * PER_SLOT + q;
          if (NSET == 2) {  // loads in the first slots, stores from slot ST_SLOT on
            constexpr int S0 = ST_SLOT * PER_SLOT;
            item = item < NP ? item + NP : (item >= S0 && item < S0 + NP ? item - S0 : 2 * NP);
          }
          side(item, nst, kv2, kt + 2, SS{}, SL{}, FULL);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!abl_nobar) __syncthreads();
  };
  // tiles kt whose prefetch target kt+2 lies inside K as a whole run the FULL form, the last ones the general one
  const int nfull = (EPI && (d.tune & 32)) ? 0 : max((SP ? nlist : K / BK) - 2, 0);
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  int kt = 0;
  if (NSET == 2) {
    for (; kt + 1 < nfull; kt += 2) {
      tile(kt, P0{}, std::true_type{});
      tile(kt + 1, P1{}, std::true_type{});
    }
    for (; kt < nkt; kt += 2) {  // kt is even here: register-set parity stays a compile-time constant
      tile(kt, P0{}, GEN{});
      if (kt + 1 < nkt) tile(kt + 1, P1{}, GEN{});
    }
  } else {
    for (; kt < nfull; ++kt) tile(kt, P0{}, std::true_type{});
    for (; kt < nkt; ++kt) tile(kt, P0{}, GEN{});
  }

  if constexpr (EPI) {
    // ---- reducing epilogue: the tile goes to LDS (the K loop's stages are free: its last tile ended with a barrier),
    // then out[u][v][i] = sum_{x,y} w[i][(x,y)] T[(u,x)][(v,y)] on the matrix cores (reduce_epilogue)
    static_assert(BM == 64 && BN == 64, "the reducing epilogue is written for 64 x 64 tiles");
    constexpr int LDT = BN + 1;
    static_assert((size_t)BM * LDT <= 2 * (size_t)STAGE, "tile does not fit the K loop's LDS");
    // PM (pair-major, the unguarded 4 x 4 x 4 epilogue; xm, yn powers of two): the xm x yn block of pair (u, v) is stored as
    // one run of xm yn elements in the epilogue's contraction order, runs 4 elements apart modulo 16 -- the four pairs a
    // 16-lane group of the epilogue reads at once then sit in different LDS banks (row-major 64 x 65: pair offsets are
    // multiples of 256 bytes at xm = 16, every fragment read was a 4-way bank conflict and the epilogue LDS-bound)
    auto store_tile = [&](auto PM) __attribute__((always_inline)) {
      constexpr bool pm = decltype(PM)::value;
      const int xs = pm ? __builtin_ctz(d.epi_xm) : 0, ys = pm ? __builtin_ctz(d.epi_yn) : 0;
      const int prs = d.epi_xm * d.epi_yn + 4;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = (wm * WM + i) * 16 + cd_row(cd_mode, lk, r);
            const int col = (wn * WN + j) * 16 + li;
            zc v;
            if (M3) {
              const double p1 = acc[0][i][j][r], p2 = sgnA * sgnB * acc[1][i][j][r], p3 = acc[NACC - 1][i][j][r];
              v = make_double2(p1 - p2, p3 - p1 - p2);
            } else {
              v = make_double2(acc[0][i][j][r], acc[1][i][j][r]);
            }
            if (pm) smem[(((row >> xs) << (6 - ys)) + (col >> ys)) * prs + ((row & (d.epi_xm - 1)) << ys) + (col & (d.epi_yn - 1))] = v;
            else smem[row * LDT + col] = v;
          }
      __syncthreads();
    };
    const int rbn = (d.epi_di + 15) / 16;
    const int cbn = ((BM / d.epi_xm) * (BN / d.epi_yn) + 15) / 16;
    // four blocks of outputs: one block per wave over the whole contraction (no partials to exchange); fewer: the
    // contraction split over the waves
    const int npair_ = (BM / d.epi_xm) * (BN / d.epi_yn);
    if (d.epi_b4 && !(d.epi_di <= 4 && npair_ <= 64) && npair_ <= 8 && d.epi_di <= 32) {
      // few (u, v) pairs per tile (d M = 512: eight): 4 x 4 x 4 products in four independent blocks per instruction
      // instead of 16 x 16 x 4 products half of whose columns are padding.  (It stores the tile itself, after its first
      // loads of the core have gone out.)  FULL: every set of output rows, group of pairs and k-step is whole.
      const int kp_ = d.epi_xm * d.epi_yn;
      // 4 waves x whole rings of 8 k-steps of 4; powers of two (pair-major tile: 8 (xm yn + 4) <= 2 STAGE elements)
      const bool full = (d.epi_full != 0) && d.epi_wf != nullptr && d.epi_yn % 4 == 0 && kp_ % 128 == 0 && !(d.epi_xm & (d.epi_xm - 1)) &&
                        !(d.epi_yn & (d.epi_yn - 1)) && 8 * (kp_ + 4) <= 2 * STAGE;
      if (d.epi_di == 16 && npair_ == 8 && full) reduce_epilogue_b4<1, 2, M3, true>(d, smem, tm, tn, store_tile);
      else if (d.epi_di == 32 && npair_ == 8 && full) reduce_epilogue_b4<2, 2, M3, true>(d, smem, tm, tn, store_tile);
      else if (d.epi_di <= 16) { if (npair_ <= 4) reduce_epilogue_b4<1, 1, M3, false>(d, smem, tm, tn, store_tile); else reduce_epilogue_b4<1, 2, M3, false>(d, smem, tm, tn, store_tile); }
      else { if (npair_ <= 4) reduce_epilogue_b4<2, 1, M3, false>(d, smem, tm, tn, store_tile); else reduce_epilogue_b4<2, 2, M3, false>(d, smem, tm, tn, store_tile); }
      return;
    }
    store_tile(std::false_type{});
    if (d.epi_b4 && d.epi_di <= 4 && npair_ <= 64) {
      // few output rows, many pairs (d = 4: C5): four groups of 4 pairs per instruction against the same 4 rows of the
      // core, one wave per 16 pairs over the whole contraction -- the 16 x 16 form used 4 of its 16 rows
      reduce_epilogue_b4w<M3>(d, smem, tm, tn);
      return;
    }
    if (cbn == 1) {
      if (rbn == 1) reduce_epilogue<1, 1>(d, smem, tm, tn, cd_mode);
      else if (rbn == 2) reduce_epilogue<2, 1>(d, smem, tm, tn, cd_mode);
      else reduce_epilogue_blocks<4, 1>(d, smem, tm, tn, cd_mode);
    } else if (cbn == 2) {
      if (rbn == 1) reduce_epilogue<1, 2>(d, smem, tm, tn, cd_mode);
      else reduce_epilogue_blocks<2, 2>(d, smem, tm, tn, cd_mode);
    } else {
      reduce_epilogue_blocks<1, 4>(d, smem, tm, tn, cd_mode);
    }
    return;
  }

  // ---- epilogue: C = alpha*acc + beta*C -----------------------------------
  const zc alpha = d.alpha, beta = d.beta;
  const bool has_beta = (beta.x != 0.0 || beta.y != 0.0);
  const long ldc = d.ldc;
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + (wm * WM + i) * 16 + cd_row(cd_mode, lk, r);
        const int col = n0 + (wn * WN + j) * 16 + li;
        if (row < M && col < N) {
          zc v;
          if (M3) {
            const double p1 = acc[0][i][j][r], p2 = sgnA * sgnB * acc[1][i][j][r], p3 = acc[NACC - 1][i][j][r];
            v = make_double2(p1 - p2, p3 - p1 - p2);
          } else {
            v = make_double2(acc[0][i][j][r], acc[1][i][j][r]);
          }
          zc o = zmul(alpha, v);
          const int rr = row + d.rowmap_r0;
          zc* p = C + (d.rowmap_p > 0 ? (long)(rr % d.rowmap_p) * d.rowmap_s1 + (long)(rr / d.rowmap_p) * d.rowmap_s2
                                      : (long)row * ldc) + col;
          if (has_beta) o = zadd(o, zmul(beta, *p));
          *p = o;
        }
      }
}

// The reducing epilogue (ZgemmDesc::epi_w).  T = the 64 x 64 tile in LDS (row stride 65).  The contraction index
// k' = (x, y) is split over the four waves; every wave forms RB x CB partial blocks of 16 (i) x 16 (pairs (u, v)) with
// the 4M complex product; the partials meet in LDS (over T, once every wave has read it) and are summed in a fixed
// order.  The fragments of w come from global memory (L2: the matrix is d x (xm yn), shared by all tiles) four k-steps
// ahead of their use.
template <int RB, int CB>
__device__ __forceinline__ void reduce_epilogue(const ZgemmDesc& d, zc* smem, int tm, int tn, int cd_mode) {
  constexpr int LDT = 65;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 15, lk = lane >> 4;
  const int XM = d.epi_xm, YN = d.epi_yn, KP = XM * YN, DI = d.epi_di;
  const int TU = 64 / XM, TV = 64 / YN, npair = TU * TV;
  const int nk4 = (KP + 3) / 4, per = (nk4 + 3) / 4;
  const int k4a = w * per, k4b = min(nk4, k4a + per);
  const zc* __restrict__ Wm = d.epi_w;
  const long ldw = d.epi_ldw;
  double zin = 0.0;
  asm volatile("" : "+v"(zin));
  d4 zr[RB * CB], zi[RB * CB];
#pragma unroll
  for (int q = 0; q < RB * CB; ++q) {
    zr[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(zin, zin, (d4){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
    zi[q] = zr[q];
  }
  // LDS offsets of this lane's pairs (one per column block)
  int toff[CB];
  bool tval[CB];
#pragma unroll
  for (int jb = 0; jb < CB; ++jb) {
    const int pr = jb * 16 + li;
    tval[jb] = pr < npair;
    const int prc = tval[jb] ? pr : 0;
    const int ul = prc / TV, vl = prc - ul * TV;
    toff[jb] = ul * XM * LDT + vl * YN;
  }
  constexpr int CH = 4;  // k-steps per chunk of w fragments
  zc wv[2][CH][RB];
  auto load_w = [&](int buf, int k4s) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int kk = (k4s + c) * 4 + lk;
#pragma unroll
      for (int ib = 0; ib < RB; ++ib) {
        const int i = ib * 16 + li;
        const bool ok = (k4s + c) < k4b && kk < KP && i < DI;
        // (the index is clamped and the VALUE selected: a select between the load and a constant makes hipcc select
        // between two addresses, the constant's in scratch)
        zc v = Wm[ok ? (long)i * ldw + kk : 0];
        v.x = ok ? v.x : 0.0;
        v.y = ok ? v.y : 0.0;
        wv[buf][c][ib] = v;
      }
    }
  };
  auto chunk = [&](int buf, int k4s) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int k4 = k4s + c;
      if (k4 < k4b) {
        const int kk = k4 * 4 + lk;
        const bool kv = kk < KP;
        const int kc = kv ? kk : 0;
        const int x = kc / YN, y = kc - x * YN;
        zc tv[CB];
#pragma unroll
        for (int jb = 0; jb < CB; ++jb) {
          const bool ok = kv && tval[jb];
          zc v = smem[ok ? toff[jb] + x * LDT + y : 0];
          v.x = ok ? v.x : 0.0;
          v.y = ok ? v.y : 0.0;
          tv[jb] = v;
        }
#pragma unroll
        for (int ib = 0; ib < RB; ++ib) {
          const zc a = wv[buf][c][ib];
          const double nai = -a.y;
#pragma unroll
          for (int jb = 0; jb < CB; ++jb) {
            const int q = ib * CB + jb;
            zr[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, tv[jb].x, zr[q], 0, 0, 0);
            zi[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, tv[jb].y, zi[q], 0, 0, 0);
            zr[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(nai, tv[jb].y, zr[q], 0, 0, 0);
            zi[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, tv[jb].x, zi[q], 0, 0, 0);
          }
        }
      }
    }
  };
  load_w(0, k4a);
  for (int k4s = k4a; k4s < k4b; k4s += 2 * CH) {
    load_w(1, k4s + CH);
    chunk(0, k4s);
    load_w(0, k4s + 2 * CH);
    chunk(1, k4s + CH);
  }
  __syncthreads();  // every wave has read its part of T
  // partials: [wave][block][reg * 64 + lane]
#pragma unroll
  for (int q = 0; q < RB * CB; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) smem[(w * (RB * CB) + q) * 256 + r * 64 + lane] = make_double2(zr[q][r], zi[q][r]);
  __syncthreads();
  const int el = t & 63, er = t >> 6;  // thread t owns (lane el, register er) of every block
  const int row_b = cd_row(cd_mode, el >> 4, er), col_b = el & 15;
#pragma unroll
  for (int q = 0; q < RB * CB; ++q) {
    const int ib = q / CB, jb = q - ib * CB;
    double re = 0.0, im = 0.0;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      const zc v = smem[(ww * (RB * CB) + q) * 256 + t];
      re += v.x;
      im += v.y;
    }
    const int i = ib * 16 + row_b, pr = jb * 16 + col_b;
    if (i < DI && pr < npair) {
      const int ul = pr / TV, vl = pr - ul * TV;
      const long u = (long)tm * TU + ul, v = (long)tn * TV + vl;
      if (u * XM < d.M && v * YN < d.N) {
        zc* p = d.C + u * d.epi_su + v * d.epi_sv + (long)i * d.epi_si;
        if (d.epi_acc) { const zc o = *p; re += o.x; im += o.y; }
        *p = make_double2(re, im);
      }
    }
  }
}

// The same contraction when there are FOUR blocks of 16 x 16 outputs (RB * CB = 4: d = 4, M = 16 gives 64 (u, v) pairs per
// tile = four column blocks): wave w owns block w over the whole contraction index, so nothing is exchanged between the
// waves and each writes its own outputs -- two barriers and a 64 KB LDS round trip less than the split form.
template <int RB, int CB>
__device__ __forceinline__ void reduce_epilogue_blocks(const ZgemmDesc& d, zc* smem, int tm, int tn, int cd_mode) {
  static_assert(RB * CB == 4, "one block per wave");
  constexpr int LDT = 65;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 15, lk = lane >> 4;
  const int ib = w / CB, jb = w - ib * CB;
  const int XM = d.epi_xm, YN = d.epi_yn, KP = XM * YN, DI = d.epi_di;
  const int TU = 64 / XM, TV = 64 / YN, npair = TU * TV;
  const int nk4 = (KP + 3) / 4;
  const zc* __restrict__ Wm = d.epi_w;
  const long ldw = d.epi_ldw;
  double zin = 0.0;
  asm volatile("" : "+v"(zin));
  d4 zr = __builtin_amdgcn_mfma_f64_16x16x4f64(zin, zin, (d4){0.0, 0.0, 0.0, 0.0}, 0, 0, 0), zi = zr;
  const int pr = jb * 16 + li;
  const bool pval = pr < npair;
  const int prc = pval ? pr : 0;
  const int ul = prc / TV, vl = prc - ul * TV;
  const int toff = ul * XM * LDT + vl * YN;
  const int i = ib * 16 + li;
  constexpr int CH = 4;
  zc wv[2][CH];
  auto load_w = [&](int buf, int k4s) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int kk = (k4s + c) * 4 + lk;
      const bool ok = (k4s + c) < nk4 && kk < KP && i < DI;
      zc v = Wm[ok ? (long)i * ldw + kk : 0];
      v.x = ok ? v.x : 0.0;
      v.y = ok ? v.y : 0.0;
      wv[buf][c] = v;
    }
  };
  auto chunk = [&](int buf, int k4s) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int k4 = k4s + c;
      if (k4 < nk4) {
        const int kk = k4 * 4 + lk;
        const bool ok = kk < KP && pval;
        const int kc = kk < KP ? kk : 0;
        const int x = kc / YN, y = kc - x * YN;
        zc tv = smem[ok ? toff + x * LDT + y : 0];
        tv.x = ok ? tv.x : 0.0;
        tv.y = ok ? tv.y : 0.0;
        const zc a = wv[buf][c];
        const double nai = -a.y;
        zr = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, tv.x, zr, 0, 0, 0);
        zi = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, tv.y, zi, 0, 0, 0);
        zr = __builtin_amdgcn_mfma_f64_16x16x4f64(nai, tv.y, zr, 0, 0, 0);
        zi = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, tv.x, zi, 0, 0, 0);
      }
    }
  };
  load_w(0, 0);
  for (int k4s = 0; k4s < nk4; k4s += 2 * CH) {
    load_w(1, k4s + CH);
    chunk(0, k4s);
    load_w(0, k4s + 2 * CH);
    chunk(1, k4s + CH);
  }
  // lane (li, lk) holds outputs (row cd_row(lk, r), column li) of its block, r = 0..3
  if (pval) {
    const long u = (long)tm * TU + ul, v = (long)tn * TV + vl;
    if (u * XM < d.M && v * YN < d.N) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int io = ib * 16 + cd_row(cd_mode, lk, r);
        if (io < DI) {
          zc* p = d.C + u * d.epi_su + v * d.epi_sv + (long)io * d.epi_si;
          double re = zr[r], im = zi[r];
          if (d.epi_acc) { const zc o = *p; re += o.x; im += o.y; }
          *p = make_double2(re, im);
        }
      }
    }
  }
}

// The same contraction with v_mfma_f64_4x4x4_4b_f64: FOUR independent 4 x 4 x 4 products per instruction (same multiply-add
// rate as the 16 x 16 x 4 form).  A tile with few (u, v) pairs -- d M = 512 leaves eight -- fills only half of the sixteen
// columns of a 16 x 16 product (profiles/r04_edge_apply_ab.txt: the reason the edge form lost at C3 / C4); here the four
// blocks of an instruction are four groups of 4 output rows i against ONE group of 4 pairs, so nothing is padding while
// DI is a multiple of 4 and the pairs come in fours.  NIS = sets of 16 output rows, NPG = groups of 4 pairs.
// Lane maps of the instruction: the MITDVP_B4_* macros at the top of this file.
template <int NIS, int NPG, bool M3, bool FULL, class ST>
__device__ __forceinline__ void reduce_epilogue_b4(const ZgemmDesc& d, zc* smem, int tm, int tn, ST&& store_tile) {
  constexpr int LDT = 65;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);  // wave-uniform: the k ranges below live in scalar registers
  const int blk = MITDVP_B4_BLK(lane), q4 = MITDVP_B4_Q(lane), k4l = MITDVP_B4_K(lane);  // operand maps: (block, row / column, k)
  const int XM = d.epi_xm, YN = d.epi_yn, KP = XM * YN, DI = d.epi_di;
  const int TU = 64 / XM, TV = 64 / YN, npair = TU * TV;
  const int nk4 = (KP + 3) / 4, per = (nk4 + 3) / 4;
  const int k4a = w * per, k4b = (d.tune & 16) ? k4a : min(nk4, k4a + per);
  const zc* __restrict__ Wm = d.epi_w;
  const long ldw = d.epi_ldw;
  double p1[NIS][NPG], p2[NIS][NPG], p3[NIS][NPG];
#pragma unroll
  for (int s = 0; s < NIS; ++s)
#pragma unroll
    for (int g = 0; g < NPG; ++g) p1[s][g] = p2[s][g] = p3[s][g] = 0.0;
  // A operand of i-set s: w[(4 s + blk) 4 + q4][4 k4 + k4l];  B operand of pair group g: T of pair 4 g + q4 at k = 4 k4 + k4l
  long woff[NIS];
  bool wval[NIS];
#pragma unroll
  for (int s = 0; s < NIS; ++s) {
    const int i = (4 * s + blk) * 4 + q4;
    wval[s] = FULL || i < DI;
    woff[s] = (long)(wval[s] ? i : 0) * ldw + k4l;
  }
  int toff[NPG];
  bool tval[NPG];
#pragma unroll
  for (int g = 0; g < NPG; ++g) {
    const int pr = 4 * g + q4;
    tval[g] = FULL || pr < npair;
    const int prc = tval[g] ? pr : 0;
    const int ul = prc / TV, vl = prc - ul * TV;
    toff[g] = FULL ? prc * (KP + 4) + k4l : ul * XM * LDT + vl * YN;
  }
  constexpr int CH = 4;  // k-steps per chunk of w fragments, loaded one chunk ahead
  zc wv[2][CH][NIS];
  auto load_w = [&](int buf, int k4s) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if constexpr (FULL) {
#pragma unroll
        for (int s = 0; s < NIS; ++s) wv[buf][c][s] = Wm[woff[s] + (long)(k4s + c) * 4];
      } else {
        const int kk = (k4s + c) * 4 + k4l;
#pragma unroll
        for (int s = 0; s < NIS; ++s) {
          const bool ok = (k4s + c) < k4b && kk < KP && wval[s];
          zc v = Wm[ok ? woff[s] - k4l + kk : 0];
          v.x = ok ? v.x : 0.0;
          v.y = ok ? v.y : 0.0;
          wv[buf][c][s] = v;
        }
      }
    }
  };
  auto products = [&](const zc (&av)[NIS], const zc (&tv)[NPG]) __attribute__((always_inline)) {
    // the complex product as in the K loop: 4M (p1 = Re, p2 = Im), or 3M (Karatsuba): P1 = sum ar tr, P2 = sum ai ti,
    // P3 = sum (ar + ai)(tr + ti) -- the sums cost NIS + NPG additions per k-step and save NIS NPG products
    double ts[NPG];
#pragma unroll
    for (int g = 0; g < NPG; ++g) ts[g] = tv[g].x + tv[g].y;
#pragma unroll
    for (int s = 0; s < NIS; ++s) {
      const zc a = av[s];
      const double as = a.x + a.y, nai = -a.y;
#pragma unroll
      for (int g = 0; g < NPG; ++g) {
        if constexpr (M3) {
          p1[s][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x, tv[g].x, p1[s][g], 0, 0, 0);
          p2[s][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.y, tv[g].y, p2[s][g], 0, 0, 0);
          p3[s][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(as, ts[g], p3[s][g], 0, 0, 0);
        } else {
          p1[s][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x, tv[g].x, p1[s][g], 0, 0, 0);
          p2[s][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x, tv[g].y, p2[s][g], 0, 0, 0);
          p1[s][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(nai, tv[g].y, p1[s][g], 0, 0, 0);
          p2[s][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.y, tv[g].x, p2[s][g], 0, 0, 0);
        }
      }
    }
  };
  int tk = 4 * k4a;  // FULL (pair-major tile): a pair's fragments are read in the order they are stored in
  auto chunk = [&](int buf, int k4s) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if constexpr (FULL) {
        zc tv[NPG];
#pragma unroll
        for (int g = 0; g < NPG; ++g) tv[g] = smem[toff[g] + tk];
        tk += 4;
        products(wv[buf][c], tv);
      } else {
        const int k4 = k4s + c;
        if (k4 < k4b) {
          const int kk = k4 * 4 + k4l;
          const bool kv = kk < KP;
          const int kc = kv ? kk : 0;
          const int x = kc / YN, y = kc - x * YN;
          zc tv[NPG];
#pragma unroll
          for (int g = 0; g < NPG; ++g) {
            const bool ok = kv && tval[g];
            zc v = smem[ok ? toff[g] + x * LDT + y : 0];
            v.x = ok ? v.x : 0.0;
            v.y = ok ? v.y : 0.0;
            tv[g] = v;
          }
          products(wv[buf][c], tv);
        }
      }
    }
  };
  if constexpr (FULL) {
    // a ring of eight k-steps of core fragments, requested RD steps ahead of their use (the core comes from L2: ~0.6 us under
    // load, three k-steps of products), and the tile's fragments read one k-step ahead
    constexpr int RING = 2 * CH, RD = RING - 2;
    zc (&ring)[RING][NIS] = reinterpret_cast<zc (&)[RING][NIS]>(wv);
    // (the core in fragment order: the 64 lanes of one load read 1 KB in a row -- eight whole cache lines; from the
    // row-major core a load touched sixteen half lines 8 KB apart, and the core loads, not the products, set the
    // epilogue's pace: 52 us of a 155 us stage at C3 against 27 us without them)
    const zc* __restrict__ Wf = d.epi_wf + lane;
    auto ld = [&](int slot, int k4) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < NIS; ++s) ring[slot][s] = Wf[(long)(k4 * NIS + s) * 64];
    };
#pragma unroll
    for (int q = 0; q < RD; ++q) ld(q, k4a + q);  // on their way while the tile goes to LDS (k4b - k4a >= RING)
    store_tile(std::true_type{});
    zc tv[2][NPG];
    auto ldt = [&](int b) __attribute__((always_inline)) {
#pragma unroll
      for (int g = 0; g < NPG; ++g) tv[b][g] = smem[toff[g] + tk];
      tk += 4;
    };
    ldt(0);
    for (int k4s = k4a; k4s < k4b; k4s += RING) {
#pragma unroll
      for (int c = 0; c < RING; ++c) {
        if (k4s + c + RD < k4b && !(d.tune & 64)) ld((c + RD) % RING, k4s + c + RD);  // (tune 64 / 128: timing experiments)
        if (k4s + c + 1 < k4b && !(d.tune & 128)) ldt((c + 1) & 1);
        products(ring[c], tv[c & 1]);
      }
    }
  } else {
    load_w(0, k4a);  // the first fragments of the core are on their way while the tile goes to LDS
    store_tile(std::false_type{});
    for (int k4s = k4a; k4s < k4b; k4s += 2 * CH) {
      load_w(1, k4s + CH);
      chunk(0, k4s);
      load_w(0, k4s + 2 * CH);
      chunk(1, k4s + CH);
    }
  }
  // where this thread's result goes (thread t owns result lane t % 64 of set t / 64), and -- accumulating stage -- what
  // is there now: asked for before the exchange of the partials, not after it
  constexpr int NQ = NIS * NPG;
  zc* outp = nullptr;
  zc old = make_double2(0.0, 0.0);
  if (t < NQ * 64) {
    const int q = t >> 6, el = t & 63;
    const int s = q / NPG, g = q - s * NPG;
    const int i = (4 * s + MITDVP_B4_BLK(el)) * 4 + MITDVP_B4_D_ROW(el), pr = 4 * g + MITDVP_B4_D_COL(el);
    if (i < DI && pr < npair) {
      const int ul = pr / TV, vl = pr - ul * TV;
      const long u = (long)tm * TU + ul, v = (long)tn * TV + vl;
      if (u * XM < d.M && v * YN < d.N) {
        outp = d.C + u * d.epi_su + v * d.epi_sv + (long)i * d.epi_si;
        if (d.epi_acc) old = *outp;
      }
    }
  }
  __syncthreads();  // every wave has read its part of T
  // partials of the four waves (the contraction index was split over them): [wave][set][lane]
#pragma unroll
  for (int s = 0; s < NIS; ++s)
#pragma unroll
    for (int g = 0; g < NPG; ++g)  // 3M: Re = P1 - P2, Im = P3 - P1 - P2
      smem[(w * NQ + s * NPG + g) * 64 + lane] =
          M3 ? make_double2(p1[s][g] - p2[s][g], p3[s][g] - p1[s][g] - p2[s][g]) : make_double2(p1[s][g], p2[s][g]);
  __syncthreads();
  if (outp) {
    const int q = t >> 6, el = t & 63;
    double re = 0.0, im = 0.0;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      const zc v = smem[(ww * NQ + q) * 64 + el];
      re += v.x;
      im += v.y;
    }
    *outp = make_double2(re + old.x, im + old.y);
  }
}

// DI <= 4 output rows and up to 64 pairs per tile (d = 4, M = 16): wave w owns pairs 16 w .. 16 w + 15 over the whole
// contraction index -- the four blocks of an instruction are four groups of 4 pairs against the same 4 rows of the core;
// nothing is exchanged between the waves.
template <bool M3>
__device__ __forceinline__ void reduce_epilogue_b4w(const ZgemmDesc& d, zc* smem, int tm, int tn) {
  constexpr int LDT = 65;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int blk = MITDVP_B4_BLK(lane), q4 = MITDVP_B4_Q(lane), k4l = MITDVP_B4_K(lane);
  const int XM = d.epi_xm, YN = d.epi_yn, KP = XM * YN, DI = d.epi_di;
  const int TU = 64 / XM, TV = 64 / YN, npair = TU * TV;
  const int nk4 = (KP + 3) / 4;
  const zc* __restrict__ Wm = d.epi_w;
  const long ldw = d.epi_ldw;
  const bool wval = q4 < DI;
  const long woff = (long)(wval ? q4 : 0) * ldw;
  const int pr = 16 * w + 4 * blk + q4;
  const bool pval = pr < npair;
  const int prc = pval ? pr : 0;
  const int ul = prc / TV, vl = prc - ul * TV;
  const int toff = ul * XM * LDT + vl * YN;
  double p1 = 0.0, p2 = 0.0, p3 = 0.0;
  constexpr int CH = 4;
  zc wv[2][CH];
  auto load_w = [&](int buf, int k4s) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int kk = (k4s + c) * 4 + k4l;
      const bool ok = (k4s + c) < nk4 && kk < KP && wval;
      zc v = Wm[ok ? woff + kk : 0];
      v.x = ok ? v.x : 0.0;
      v.y = ok ? v.y : 0.0;
      wv[buf][c] = v;
    }
  };
  auto chunk = [&](int buf, int k4s) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int k4 = k4s + c;
      if (k4 < nk4) {
        const int kk = k4 * 4 + k4l;
        const bool ok = kk < KP && pval;
        const int kc = kk < KP ? kk : 0;
        const int x = kc / YN, y = kc - x * YN;
        zc tv = smem[ok ? toff + x * LDT + y : 0];
        tv.x = ok ? tv.x : 0.0;
        tv.y = ok ? tv.y : 0.0;
        const zc a = wv[buf][c];
        if constexpr (M3) {
          p1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x, tv.x, p1, 0, 0, 0);
          p2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a.y, tv.y, p2, 0, 0, 0);
          p3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x + a.y, tv.x + tv.y, p3, 0, 0, 0);
        } else {
          p1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x, tv.x, p1, 0, 0, 0);
          p2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x, tv.y, p2, 0, 0, 0);
          p1 = __builtin_amdgcn_mfma_f64_4x4x4f64(-a.y, tv.y, p1, 0, 0, 0);
          p2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a.y, tv.x, p2, 0, 0, 0);
        }
      }
    }
  };
  load_w(0, 0);
  for (int k4s = 0; k4s < nk4; k4s += 2 * CH) {
    load_w(1, k4s + CH);
    chunk(0, k4s);
    load_w(0, k4s + 2 * CH);
    chunk(1, k4s + CH);
  }
  // result lane: row i = lane / 16, pair 16 w + 4 block + lane % 4
  const int io = MITDVP_B4_D_ROW(lane), po = 16 * w + 4 * MITDVP_B4_BLK(lane) + MITDVP_B4_D_COL(lane);
  if (io < DI && po < npair) {
    const int uo = po / TV, vo = po - uo * TV;
    const long u = (long)tm * TU + uo, v = (long)tn * TV + vo;
    if (u * XM < d.M && v * YN < d.N) {
      zc* p = d.C + u * d.epi_su + v * d.epi_sv + (long)io * d.epi_si;
      double re = M3 ? p1 - p2 : p1, im = M3 ? p3 - p1 - p2 : p2;
      if (d.epi_acc) { const zc o = *p; re += o.x; im += o.y; }
      *p = make_double2(re, im);
    }
  }
}

// ---------------------------------------------------------------------------
// MFMA probes
// ---------------------------------------------------------------------------
__global__ void mfma_layout_kernel(const double* A /*16x4*/, const double* B /*4x16*/, double* out /*64*4*/) {
  const int lane = threadIdx.x;
  const int li = lane & 15, lk = lane >> 4;
  d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[li * 4 + lk], B[lk * 16 + li], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}

void mfma_layout_probe(hipStream_t st, int* host_out) {
  double hA[64], hB[64], hD[256], ref[256];
  // D[i][j] = (i+1) + 1000 (j+1) + 2: every entry distinct, and a k-slot mismatch
  // between the A and B lane maps would pair the wrong factors
  for (int i = 0; i < 16; ++i) {
    hA[i * 4 + 0] = i + 1.0; hA[i * 4 + 1] = 1.0; hA[i * 4 + 2] = 0.5; hA[i * 4 + 3] = 0.25;
  }
  for (int j = 0; j < 16; ++j) {
    hB[0 * 16 + j] = 1.0; hB[1 * 16 + j] = 1000.0 * (j + 1); hB[2 * 16 + j] = 2.0; hB[3 * 16 + j] = 4.0;
  }
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j];
      ref[i * 16 + j] = s;
    }
  double *dA, *dB, *dD;
  HIP_CHECK(hipMalloc(&dA, sizeof(hA)));
  HIP_CHECK(hipMalloc(&dB, sizeof(hB)));
  HIP_CHECK(hipMalloc(&dD, sizeof(hD)));
  HIP_CHECK(hipMemcpyAsync(dA, hA, sizeof(hA), hipMemcpyHostToDevice, st));
  HIP_CHECK(hipMemcpyAsync(dB, hB, sizeof(hB), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(mfma_layout_kernel, dim3(1), dim3(64), 0, st, dA, dB, dD);
  HIP_CHECK(hipMemcpyAsync(hD, dD, sizeof(hD), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  HIP_CHECK(hipFree(dA));
  HIP_CHECK(hipFree(dB));
  HIP_CHECK(hipFree(dD));
  // locate every (lane, reg) value in the reference product (all values distinct)
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      int fr = -1, fc = -1;
      for (int i = 0; i < 16 && fr < 0; ++i)
        for (int j = 0; j < 16; ++j)
          if (ref[i * 16 + j] == hD[l * 4 + r]) { fr = i; fc = j; break; }
      if (host_out) { host_out[(l * 4 + r) * 2] = fr; host_out[(l * 4 + r) * 2 + 1] = fc; }
    }
  int mode = -1;
  for (int m = 0; m < 2 && mode < 0; ++m) {
    bool ok = true;
    for (int l = 0; l < 64 && ok; ++l)
      for (int r = 0; r < 4; ++r) {
        const int lk = l >> 4, li = l & 15;
        const int row = m == 0 ? lk + 4 * r : 4 * lk + r;
        if (hD[l * 4 + r] != ref[row * 16 + li]) { ok = false; break; }
      }
    if (ok) mode = m;
  }
  if (mode < 0) throw HipError("v_mfma_f64_16x16x4_f64: unknown operand/result lane map");
  g_cd_mode = mode;
}

__global__ __launch_bounds__(256) void mfma_peak_kernel(double* out, int iters, unsigned long long* stamps) {
  d4 acc[8];
  double zin = 0.0;
  asm volatile("" : "+v"(zin));
  for (int i = 0; i < 8; ++i)  // born in AGPRs (see zgemm_kernel)
    acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(zin, zin, (d4){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
  double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (s == 123.456) out[0] = s;  // keep the chain live
  if (stamps && blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

// the same chain with v_mfma_f64_4x4x4_4b_f64 (four 4 x 4 x 4 products per instruction: 256 multiply-adds)
__global__ __launch_bounds__(256) void mfma_peak_kernel_b4(double* out, int iters, unsigned long long* stamps) {
  double acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = 0.0;
  double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (s == 123.456) out[0] = s;
  if (stamps && blockIdx.x == 0 && threadIdx.x == 0) stamps[0] = t1 - t0;
}

// Raw v_mfma_f64_16x16x4_f64 issue rate with `waves_per_simd` resident waves on
// every SIMD.  Prints cycles per MFMA and the in-kernel clock when
// MITDVP_VERBOSE is set (MI355X_MICROARCH.md, DVFS give-back item 6).
double mfma_peak_probe(hipStream_t st) {
  double* dout;
  unsigned long long* dst;
  HIP_CHECK(hipMalloc(&dout, 8));
  HIP_CHECK(hipMalloc(&dst, 16));
  double best = 0.0;
  const bool verbose = getenv("MITDVP_VERBOSE") != nullptr;
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = 256 * wps, iters = 4000;
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, st, dout, 10, nullptr);
    HIP_CHECK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, st, dout, iters, dst);
    HIP_CHECK(hipEventRecord(e1, st));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    HIP_CHECK(hipEventDestroy(e0));
    HIP_CHECK(hipEventDestroy(e1));
    unsigned long long hs[2];
    HIP_CHECK(hipMemcpy(hs, dst, 16, hipMemcpyDeviceToHost));
    const double flops = (double)blocks * 4 /*waves*/ * iters * 8.0 * (2.0 * 16 * 16 * 4);
    const double tf = flops / (ms * 1e-3) / 1e12;
    if (verbose)
      fprintf(stderr, "[mitdvp] f64 mfma probe: %d wave/SIMD: %.1f TFLOP/s, %.1f cycles per MFMA per wave, clock %.2f GHz\n",
              wps, tf, (double)hs[0] / (iters * 8.0), (double)hs[0] / (double)hs[1] * 0.1);
    best = std::max(best, tf);
    if (verbose) {  // the 4 x 4 x 4 form's rate beside it
      HIP_CHECK(hipEventCreate(&e0));
      HIP_CHECK(hipEventCreate(&e1));
      hipLaunchKernelGGL(mfma_peak_kernel_b4, dim3(blocks), dim3(256), 0, st, dout, 10, nullptr);
      HIP_CHECK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(mfma_peak_kernel_b4, dim3(blocks), dim3(256), 0, st, dout, iters, dst);
      HIP_CHECK(hipEventRecord(e1, st));
      HIP_CHECK(hipEventSynchronize(e1));
      HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      HIP_CHECK(hipEventDestroy(e0));
      HIP_CHECK(hipEventDestroy(e1));
      HIP_CHECK(hipMemcpy(hs, dst, 8, hipMemcpyDeviceToHost));
      const double f4 = (double)blocks * 4 * iters * 8.0 * (2.0 * 4 * 4 * 4 * 4);
      fprintf(stderr, "[mitdvp] f64 mfma 4x4x4_4b probe: %d wave/SIMD: %.1f TFLOP/s, %.1f cycles per MFMA per wave\n", wps,
              f4 / (ms * 1e-3) / 1e12, (double)hs[0] / (iters * 8.0));
    }
  }
  HIP_CHECK(hipFree(dout));
  HIP_CHECK(hipFree(dst));
  return best;
}

// ---------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------
template <int WM, int WN, int BK, bool TA, bool TB, bool M3>
static void launch_one(hipStream_t st, const ZgemmDesc& d, dim3 grid, int ntm, int ntn) {
  constexpr int BM = 2 * WM * 16, BN = 2 * WN * 16;
  constexpr int A_SZ = TA ? BK * BM : BM * (BK + 1);
  constexpr int B_SZ = TB ? BN * (BK + 1) : BK * BN;
  constexpr size_t lds = 2 * (size_t)(A_SZ + B_SZ) * sizeof(zc);
  auto kern = zgemm_kernel<WM, WN, BK, TA, TB, M3>;
  if (lds > 65536) {
    // the attribute is per DEVICE (a process may hold engines on several GPUs): one flag per
    // instantiation and device ordinal
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 0 || dev >= 64 || !done[dev]) {
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      if (dev >= 0 && dev < 64) done[dev] = true;
    }
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, d, ntm, ntn, g_cd_mode);
}

template <int WM, int WN, int BK, bool M3>
static void launch_cfg(hipStream_t st, const ZgemmDesc& d) {
  constexpr int BM = 2 * WM * 16, BN = 2 * WN * 16;
  const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
  dim3 grid(ntm * ntn, d.batch);
  if (!d.transA && !d.transB) launch_one<WM, WN, BK, false, false, M3>(st, d, grid, ntm, ntn);
  else if (!d.transA && d.transB) launch_one<WM, WN, BK, false, true, M3>(st, d, grid, ntm, ntn);
  else if (d.transA && !d.transB) launch_one<WM, WN, BK, true, false, M3>(st, d, grid, ntm, ntn);
  else launch_one<WM, WN, BK, true, true, M3>(st, d, grid, ntm, ntn);
  HIP_CHECK(hipGetLastError());
}

// Complex-product form: 0 = "4M" (4 real MFMAs per complex tile step, the
// textbook product), 1 = "3M" (Karatsuba, 3 real MFMAs: 25 % less matrix-core
// work, normwise -- not componentwise -- stable).  MITDVP_ZGEMM=4m|3m overrides.
static int g_gemm_mode = -1;
int zgemm_default_mode() {
  if (g_gemm_mode < 0) {
    const char* e = getenv("MITDVP_ZGEMM");
    g_gemm_mode = (e && (e[0] == '4')) ? 0 : 1;
  }
  return g_gemm_mode;
}
void zgemm_set_default_mode(int m) { g_gemm_mode = m ? 1 : 0; }

// C = alpha * sum_z ws[z] + beta * C   (deterministic split-K combine)
__global__ __launch_bounds__(256) void zgemm_splitk_reduce(const zc* __restrict__ ws, int splits, int M, int N,
                                                           zc* __restrict__ C, long ldc, zc alpha, zc beta) {
  const long tot = (long)M * N;
  const bool has_beta = (beta.x != 0.0 || beta.y != 0.0);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long)gridDim.x * 256) {
    double re = 0.0, im = 0.0;
    for (int z = 0; z < splits; ++z) {
      const zc v = ws[(long)z * tot + e];
      re += v.x;
      im += v.y;
    }
    zc o = zmul(alpha, make_double2(re, im));
    zc* p = C + (e / N) * ldc + (e % N);
    if (has_beta) o = zadd(o, zmul(beta, *p));
    *p = o;
  }
}

// split-K partial slabs: one workspace PER STREAM (several engines, each with its own stream, may
// run concurrently from different host threads), grown on demand, released with the stream
struct SplitKWs { zc* p = nullptr; size_t n = 0; };
static std::mutex g_splitk_mu;
static std::map<hipStream_t, SplitKWs> g_splitk;

static zc* splitk_workspace(hipStream_t st, size_t need) {
  std::lock_guard<std::mutex> lk(g_splitk_mu);
  SplitKWs& w = g_splitk[st];
  if (need > w.n) {
    if (w.p) { HIP_CHECK(hipStreamSynchronize(st)); HIP_CHECK(hipFree(w.p)); w.p = nullptr; w.n = 0; }
    HIP_CHECK(hipMalloc(&w.p, need * sizeof(zc)));
    w.n = need;
  }
  return w.p;
}
void zgemm_release_stream(hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_splitk_mu);
  auto it = g_splitk.find(st);
  if (it == g_splitk.end()) return;
  if (it->second.p) (void)hipFree(it->second.p);
  g_splitk.erase(it);
}

// The staging loads address a tile as uniform base + a 32-bit per-thread byte offset: the rows a tile spans times the row
// stride must stay below 2^32 bytes (an untransposed A spans BM rows -- up to 2 BM + 2 stored rows with the identity-block
// row skip --, a transposed one BK = 16; B likewise with BN).  64 x 64 tiles: lda, ldb < 2^22 elements untransposed.
static void check_tile_offsets(const ZgemmDesc& d, int bm, int bn) {
  constexpr long BKMAX = 16, LIM = 0xFFFFFFF0L / (long)sizeof(zc);
  const long rowsA = d.transA ? BKMAX : (d.arow_skip > 1 ? 2L * bm + 2 : (long)bm), colsA = d.transA ? bm : BKMAX;
  const long rowsB = d.transB ? (long)bn : BKMAX, colsB = d.transB ? BKMAX : bn;
  if ((rowsA - 1) * d.lda + colsA >= LIM || (rowsB - 1) * d.ldb + colsB >= LIM)
    throw ArgError("zgemm: row stride too wide for the 32-bit per-thread tile offsets (rows of a tile x stride x 16 B must stay below 2^32)");
}

static void launch_tiles(hipStream_t st, const ZgemmDesc& d, int cfg, int m3) {
  check_tile_offsets(d, cfg == 0 ? 128 : (cfg == 1 ? 64 : 32), cfg == 0 ? (m3 ? 64 : 128) : (cfg == 1 ? 64 : 32));
  if (m3) {
    switch (cfg) {
      case 0: launch_cfg<4, 2, 16, true>(st, d); break;
      case 1: launch_cfg<2, 2, 16, true>(st, d); break;
      case 2: launch_cfg<1, 1, 16, true>(st, d); break;
      default: throw ArgError("zgemm: bad tile_cfg");
    }
  } else {
    switch (cfg) {
      case 0: launch_cfg<4, 4, 8, false>(st, d); break;
      case 1: launch_cfg<2, 2, 16, false>(st, d); break;
      case 2: launch_cfg<1, 1, 16, false>(st, d); break;
      default: throw ArgError("zgemm: bad tile_cfg");
    }
  }
}

// block-sparse A: 64x64 tiles, NN, K tiles from the list
static void launch_sparse(hipStream_t st, const ZgemmDesc& d, int m3) {
  if (d.transA || d.transB || d.K % 16 != 0 || d.ksplit) throw ArgError("zgemm: the block-sparse form needs NN operands and K % 16 == 0");
  constexpr int BM = 64, BN = 64;
  check_tile_offsets(d, BM, BN);
  const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
  dim3 grid(ntm * ntn, d.batch);
  constexpr size_t lds = 2 * (size_t)(BM * 17 + 16 * BN) * sizeof(zc);
  if (m3) hipLaunchKernelGGL((zgemm_kernel<2, 2, 16, false, false, true, true>), grid, dim3(256), lds, st, d, ntm, ntn, g_cd_mode);
  else hipLaunchKernelGGL((zgemm_kernel<2, 2, 16, false, false, false, true>), grid, dim3(256), lds, st, d, ntm, ntn, g_cd_mode);
  HIP_CHECK(hipGetLastError());
}

static int zgemm_tune_default() {
  // bits 0-2 change scheduling only.  Bits 4-7 switch parts of the reducing epilogue OFF for timing (the product is
  // wrong): they are honoured only together with MITDVP_TIMING_ABLATION=1, so that a stray value cannot corrupt a run.
  static const int v = [] {
    const char* e = std::getenv("MITDVP_ZGEMM_TUNE");
    const int t = e ? std::atoi(e) : 0;
    const char* a = std::getenv("MITDVP_TIMING_ABLATION");
    return (a && a[0] == '1') ? t : (t & 7);
  }();
  return v;
}

int zgemm_cd_mode(hipStream_t st);

bool zgemm_reduce_ok(int xm, int yn, int di) {
  if (xm < 1 || yn < 1 || di < 1 || xm > 64 || yn > 64 || di > 64) return false;
  const int rb = (di + 15) / 16, cb = ((64 / xm) * (64 / yn) + 15) / 16;
  // the block shapes reduce_epilogue is instantiated for (3 row blocks run as 4, 3 column blocks as 4)
  return (cb == 1) || (cb == 2 && rb <= 2) || (cb <= 4 && rb == 1);
}

// v_mfma_f64_4x4x4_4b_f64 with the lane maps reduce_epilogue_b4 assumes, against the products formed on the host (small
// integers: exact).  Once per process; a mismatch only switches the 4 x 4 x 4 epilogue off (the 16 x 16 form stays).
__global__ void k_b4_layout(double* out) {
  const int l = threadIdx.x, blk = MITDVP_B4_BLK(l), q = MITDVP_B4_Q(l), k = MITDVP_B4_K(l);
  const double a = 1.0 + blk * 16 + q * 4 + k;            // A_blk[i = q][k]
  const double b = 100.0 * (1.0 + blk * 16 + k * 4 + q);  // B_blk[k][j = q]
  out[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
}
static int g_b4_ok = -1;
static int b4_layout_ok(hipStream_t st) {
  static std::once_flag once;
  std::call_once(once, [&] {
    g_b4_ok = 0;
    if (const char* e = std::getenv("MITDVP_EPI_B4")) { if (std::atoi(e) == 0) return; }
    double* dout = nullptr;
    double h[64];
    HIP_CHECK(hipMalloc(&dout, sizeof(h)));
    hipLaunchKernelGGL(k_b4_layout, dim3(1), dim3(64), 0, st, dout);
    HIP_CHECK(hipMemcpyAsync(h, dout, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipFree(dout));
    bool ok = true;
    for (int l = 0; l < 64 && ok; ++l) {
      const int blk = MITDVP_B4_BLK(l), i = MITDVP_B4_D_ROW(l), j = MITDVP_B4_D_COL(l);
      double want = 0.0;
      for (int k = 0; k < 4; ++k) want += (1.0 + blk * 16 + i * 4 + k) * (100.0 * (1.0 + blk * 16 + k * 4 + j));
      ok = h[l] == want;
    }
    g_b4_ok = ok ? 1 : 0;
    if (!ok && std::getenv("MITDVP_VERBOSE")) fprintf(stderr, "[mitdvp] v_mfma_f64_4x4x4_4b_f64: unexpected lane map, 4 x 4 x 4 epilogue off\n");
  });
  return g_b4_ok;
}
int zgemm_reduce_b4_available(hipStream_t st) { return b4_layout_ok(st); }

// the reduced core w[di][kp] (row stride ldw) in the order the unguarded 4 x 4 x 4 epilogue reads it:
// wf[(k4 * (di / 16) + s) * 64 + lane] = w[(4 s + blk(lane)) * 4 + q(lane)][4 k4 + k(lane)]
__global__ __launch_bounds__(256) void k_pack_core(const zc* __restrict__ w, long ldw, int nis, long total, zc* __restrict__ wf) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int lane = (int)(e & 63);
    const long f = e >> 6;
    const int s = (int)(f % nis);
    const long k4 = f / nis;
    wf[e] = w[(long)((4 * s + MITDVP_B4_BLK(lane)) * 4 + MITDVP_B4_Q(lane)) * ldw + 4 * k4 + MITDVP_B4_K(lane)];
  }
}
// whether a reducing product of this shape runs the unguarded 4 x 4 x 4 epilogue (the condition of zgemm_kernel's EPI branch)
bool zgemm_reduce_full_ok(hipStream_t st, int xm, int yn, int di) {
  if (!zgemm_reduce_b4_available(st) || xm < 1 || yn < 1 || xm > 64 || yn > 64) return false;
  static const bool on = !(std::getenv("MITDVP_EPI_FULL") && std::getenv("MITDVP_EPI_FULL")[0] == '0');
  const int kp = xm * yn, npair = (64 / xm) * (64 / yn);
  return on && (di == 16 || di == 32) && npair == 8 && yn % 4 == 0 && kp % 128 == 0 && !(xm & (xm - 1)) && !(yn & (yn - 1)) &&
         8 * (kp + 4) <= 2 * (64 * 17 + 16 * 64);
}
bool zgemm_reduce_pack_core(hipStream_t st, const zc* w, long ldw, int di, int kp, zc* wf) {
  if (di % 16 || kp % 4 || di < 16 || !zgemm_reduce_b4_available(st)) return false;
  const long total = (long)di * kp;
  hipLaunchKernelGGL(k_pack_core, dim3((unsigned)std::min<long>((total + 255) / 256, 1024)), dim3(256), 0, st, w, ldw, di / 16, total, wf);
  HIP_CHECK(hipGetLastError());
  return true;
}

void zgemm_reduce(hipStream_t st, const ZgemmDesc& d0) {
  ZgemmDesc d = d0;
  d.epi_b4 = b4_layout_ok(st);
  static const int epi_full = [] { const char* e = std::getenv("MITDVP_EPI_FULL"); return (e && e[0] == '0') ? 0 : 1; }();
  d.epi_full = epi_full;
  if (d.tune < 0) d.tune = zgemm_tune_default();
  if (d.M <= 0 || d.N <= 0) return;
  if (!zgemm_reduce_ok(d.epi_xm, d.epi_yn, d.epi_di) || !d.epi_w) throw ArgError("zgemm_reduce: shape outside the reducing epilogue's range");
  if (d.transA || d.batch != 1 || d.ksplit || d.klist || d.arow_skip || d.rowmap_p || d.conjA || d.conjB)
    throw ArgError("zgemm_reduce: plain NN / NT operands only");
  if (d.M % d.epi_xm || d.N % d.epi_yn) throw ArgError("zgemm_reduce: M, N must be whole groups");
  check_tile_offsets(d, 64, 64);
  (void)zgemm_cd_mode(st);
  const int m3 = d.mode3m < 0 ? zgemm_default_mode() : d.mode3m;
  const int tu = 64 / d.epi_xm, tv = 64 / d.epi_yn;
  const int nu = d.M / d.epi_xm, nv = d.N / d.epi_yn;
  const int ntm = (nu + tu - 1) / tu, ntn = (nv + tv - 1) / tv;
  dim3 grid(ntm * ntn, 1);
  auto launch = [&](auto kern, size_t lds) {
    if (lds > 65536) {  // per device and instantiation, as in launch_one
      static std::mutex mu;
      static std::map<std::pair<const void*, int>, bool> done;
      int dev = 0;
      HIP_CHECK(hipGetDevice(&dev));
      std::lock_guard<std::mutex> lk(mu);
      bool& f = done[{reinterpret_cast<const void*>(kern), dev}];
      if (!f) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        f = true;
      }
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, d, ntm, ntn, g_cd_mode);
  };
  const size_t lds_nn = 2 * (size_t)(64 * 17 + 16 * 64) * sizeof(zc), lds_nt = 2 * (size_t)(64 * 17 + 64 * 17) * sizeof(zc);
  if (!d.transB) {
    if (m3) launch(zgemm_kernel<2, 2, 16, false, false, true, false, true>, lds_nn);
    else launch(zgemm_kernel<2, 2, 16, false, false, false, false, true>, lds_nn);
  } else {
    if (m3) launch(zgemm_kernel<2, 2, 16, false, true, true, false, true>, lds_nt);
    else launch(zgemm_kernel<2, 2, 16, false, true, false, false, true>, lds_nt);
  }
  HIP_CHECK(hipGetLastError());
}

int zgemm_cd_mode(hipStream_t st) {
  // the accumulator lane map is a property of the gfx950 ISA, not of a device: once per process
  static std::once_flag probe_once;  // several engines may issue their first GEMM concurrently
  std::call_once(probe_once, [&] { if (g_cd_mode < 0) mfma_layout_probe(st, nullptr); });
  return g_cd_mode;
}

void zgemm(hipStream_t st, const ZgemmDesc& d0) {
  ZgemmDesc d = d0;
  if (d.tune < 0) d.tune = zgemm_tune_default();
  if (d.M <= 0 || d.N <= 0 || d.batch <= 0) return;
  if (d.K < 0) throw ArgError("zgemm: negative K");
  if (d.arow_skip && (d.transA || d.arow_skip < 2 || d.klist)) throw ArgError("zgemm: arow_skip needs a plain, untransposed A");
  if (d.batch > 65535) throw ArgError("zgemm: batch > 65535");
  (void)zgemm_cd_mode(st);
  int cfg = d.tile_cfg;
  const int m3 = d.mode3m < 0 ? zgemm_default_mode() : d.mode3m;
  if (d.klist) { launch_sparse(st, d, m3); return; }
  auto tiles = [&](int bm, int bn) { return (long)((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn) * d.batch; };
  if (cfg < 0) {
    // 64x64 tiles (two resident workgroups per CU fill each other's barrier
    // bubbles) beat the 128-wide tile at every size measured on MI355X; the
    // 32x32 tile is for outputs too small to give every CU a 64x64 tile.
    // Exception: a mid-size output (16 .. 255 tiles of 64x64) with a long contraction takes the
    // 64x64 tile WITH split-K (below): 512x512x8192 runs at 53 vs 42 TFLOP/s, 256x256x4096 at 47 vs 35.
    const long t64 = tiles(64, 64);
    cfg = (t64 >= 256 || (t64 >= 16 && d.K >= 1024 && d.batch == 1)) ? 1 : 2;
    // MITDVP_ZGEMM_BIGTILE=1 (experiments): the 128-row tile, one workgroup per CU, for outputs of >= 1024 such tiles
    static const int big_env = [] { const char* e = std::getenv("MITDVP_ZGEMM_BIGTILE"); return e ? std::atoi(e) : 0; }();
    if (big_env && cfg == 1 && tiles(128, 64) >= 1024) cfg = 0;
  }
  // split-K for skinny outputs with a long contraction (QR block reflectors,
  // K_eff second stage): too few tiles to fill 256 CUs otherwise
  const int tb = cfg == 2 ? 32 : 64;
  const long nt = cfg == 0 ? tiles(128, m3 ? 64 : 128) : tiles(tb, tb);
  // ... and for outputs with MANY tiles when the contraction is very long (stage S3 of an apply at D = 1024:
  // K = M D = 32768, i.e. 2048 K tiles per workgroup; the third stage of an environment update: K = 16384).
  // Workgroups that live that long drift apart and stop sharing their operand strips in the 4 MiB L2 of their XCD:
  // FETCH_SIZE of S3 is 180 GB unsplit, 123 / 94 / 68 / 77 GB with 4 / 8 / 16 / 32 splits (tools/longk_probe.sh),
  // for 2.1 GB of partial slabs at 8 splits and a kernel that gets 2 % faster (1.2 ms of 54; the ordered combine
  // takes 0.4 ms of that back).  ~4096 columns of K per workgroup.  Same-box A/B: C4 +0.5 % sweeps/s (8 splits), C5
  // +1.7 % (its S3: 32 x 8 tiles, K = 8192, 2 splits: 0.857 -> 0.812 ms).  The third stage of an environment update
  // (A transposed, K = 16384, 16 x 512 tiles) does not gain (+0.3 % time) and stays unsplit.
  // MITDVP_LONGK_SPLITS: 0 = this rule (default), 1 = off, n = force n splits.
  static const int longk_env = [] { const char* e = std::getenv("MITDVP_LONGK_SPLITS"); return e ? std::atoi(e) : 0; }();
  const int longk = longk_env > 0 ? longk_env : (int)std::min<long>(8, d.K / 4096);
  const bool long_k = d.batch == 1 && longk > 1 && nt >= 192 && d.K >= 8192 && cfg == 1 && !d.rowmap_p &&
                      (longk_env > 0 || !d.transA);
  // (a row map scatters C rows into the real output buffer: the partial slabs of a split are plain M x N arrays,
  // so mapped outputs are never split)
  if (d.batch == 1 && !d.rowmap_p && ((nt < 192 && d.K >= 1024) || long_k)) {
    // short outputs: enough slabs to give every CU its two workgroups (MITDVP_SPLITK_TARGET workgroups, default 512 = two per CU; 384 before round 3: the C5 K_eff second stage 512 x 512 x 8192 ran 333 us, now 273)
    static const int sk_target = [] { const char* e = std::getenv("MITDVP_SPLITK_TARGET"); return e ? std::max(64, std::atoi(e)) : 512; }();
    int splits = long_k ? longk : (int)std::min<long>((sk_target + nt - 1) / nt, d.K / 256);
    if (splits >= 2) {
      int kc = (d.K + splits - 1) / splits;
      kc = (kc + 15) / 16 * 16;
      splits = (d.K + kc - 1) / kc;
      const size_t need = (size_t)splits * d.M * d.N;
      zc* ws = splitk_workspace(st, need);
      ZgemmDesc p = d;
      p.C = ws;
      p.ldc = d.N;
      p.strideA = p.strideB = 0;
      p.strideC = (long)d.M * d.N;
      p.batch = splits;
      p.ksplit = kc;
      p.alpha = make_double2(1.0, 0.0);
      p.beta = make_double2(0.0, 0.0);
      launch_tiles(st, p, cfg, m3);
      const long tot = (long)d.M * d.N;
      const int nb = (int)std::min<long>(1024, (tot + 255) / 256);
      hipLaunchKernelGGL(zgemm_splitk_reduce, dim3(nb), dim3(256), 0, st, ws, splits, d.M, d.N, d.C, d.ldc,
                         d.alpha, d.beta);
      HIP_CHECK(hipGetLastError());
      return;
    }
  }
  launch_tiles(st, d, cfg, m3);
}

}  // namespace mitdvp
