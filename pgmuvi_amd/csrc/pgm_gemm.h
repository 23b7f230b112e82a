// pgm_gemm.h -- the one dense contraction of the path, on CDNA4 fp64 MFMA.
//
//   acc[m][n] += sum_k A[k][m] * B[k][n]          ("TN": both operands k-major)
//
// Every GEMM-shaped step of the blocked factorisation is this product on 128-row
// k-blocks of row-major storage (see DESIGN.md "Data layout"): factoring the UPPER
// triangle (A = U^T U) makes the panel operands U[k][:] contiguous along the output
// index, so global loads are coalesced 16-B/lane and the LDS image [k][m] is exactly
// what v_mfma_f64_16x16x4_f64 wants (lane l supplies A[i = l&15][k = l>>4] and
// B[k = l>>4][j = l&15]; D: col = l&15, row = (l>>4) + 4*reg).
//
// 256 threads = 4 wavefronts (64 lanes), one per SIMD; each wave owns a WM x WN
// sub-tile as TM x TN MFMA tiles with the accumulators in registers.
//
// Two forms of the loop:
//  * staged (DIRECT = false): k advances in chunks of KB=16 rows, double-buffered in LDS with the next chunk's global
//    loads in flight behind the current chunk's MFMAs (one barrier per chunk).
//  * direct (DIRECT = true, round 3): NO LDS.  The rows of a wave's sub-tile are dealt to its MFMA tiles round-robin --
//    MFMA tile ti holds rows m0 + TM i + ti, i = 0..15 -- so lane (k = l>>4, i = l&15) needs A[k][m0 + TM i .. + TM-1]:
//    TM contiguous doubles, and one k-step (4 rows) of ALL the wave's A fragments is one or two 16-byte loads per lane,
//    4 x (8 TM 16)-byte contiguous runs per wavefront, straight from memory into the MFMA operand registers; the same for
//    B.  No staging registers -> LDS stores -> barrier -> LDS reads: 4 loads and 16 MFMAs per k-step, wavefronts never wait
//    for one another, PF k-steps of fragments in flight.  Measured (tools/lab/directlab, 128x128 tiles, 2 workgroups per
//    CU): 0.95 of the fp64 MFMA peak at k-depth 2048 on L2-hot operands against 0.85 staged, 0.875 / 0.80 at k-depth 512,
//    0.82 / 0.75 at 256; on operands streamed from HBM 0.88 / 0.87, 0.84 / 0.82, 0.85 / 0.80.  Each element still
//    receives the k-steps in ascending order from the same start value: same bits as the staged loop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int NB = 128;   // block size of the blocked algorithms (rows per k-block)
constexpr int KB = 16;    // k rows staged per LDS chunk
constexpr int NTHREADS = 256;

template <int BM_, int BN_, int WM_, int WN_, int PF_ = 1, int NT_ = 256, int KB_ = KB, bool DIRECT_ = false>
struct TileCfg {
  static constexpr bool DIRECT = DIRECT_;   // fragments straight from memory (PF = k-steps in flight), no LDS staging
  static constexpr int KC = KB_;          // k rows staged per LDS chunk (one barrier per chunk)
  static constexpr int NT = NT_;          // threads per workgroup: 4 wavefronts, or 16 for the filler tiles of k_diag
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
  // chunks kept in flight in registers ahead of the one being multiplied: a wave of a small tile
  // issues only 16 MFMAs (~0.4 us) per chunk, less than one L2/Infinity-Cache round trip, so a single
  // prefetched chunk leaves the loop latency-bound; PF > 1 hides it
  static constexpr int PF = PF_;
  static_assert(PF == 1 || PF == 2 || PF == 4 || PF == 8 || PF == 16 || PF == 32, "prefetch depth must divide the chunks per k-block");
  static_assert(DIRECT_ ? (NB / 4) % PF_ == 0 : ((NB / KB_) % PF_ == 0 && KB_ % 4 == 0), "chunking");
  static constexpr int TM = WM / 16, TN = WN / 16;
  static constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == NT / 64, "one WM x WN sub-tile per wavefront");
  // LDS row pitch (doubles) == 16 (mod 32): the two k rows a 32-lane group of
  // ds_read_b64 touches land on disjoint halves of the 64 banks.
  static constexpr int PA = BM + 16, PB = BN + 16;
  static constexpr int STAGE = KC * (PA + PB);
  static constexpr int LDS_DOUBLES = 2 * STAGE;
  // 16-B staging loads per thread and chunk; a narrow operand (fewer 16-B pieces than threads) is loaded
  // by the first threads only
  static constexpr int EA = KC * BM / 2, EB = KC * BN / 2;
  static constexpr int VA = (EA + NT - 1) / NT;
  static constexpr int VB = (EB + NT - 1) / NT;
  static_assert(EA % NT == 0 || EA < NT, "operand A: whole rounds of 16-B loads, or a single partial one");
  static_assert(EB % NT == 0 || EB < NT, "operand B: whole rounds of 16-B loads, or a single partial one");
};

struct WavePos {
  int lane, wave, m0, n0;
};

template <class C>
__device__ __forceinline__ WavePos wave_pos() {
  WavePos p;
  p.lane = threadIdx.x & 63;
  // (wave-uniform: the sub-tile's place stays in scalar registers.  Modulo the configuration's wavefronts: a workgroup of 16
  //  wavefronts may run four 4-wavefront tiles of a barrier-free configuration side by side, k_trsm16)
  p.wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) % (C::NT / 64);
  p.m0 = (p.wave / C::WAVES_N) * C::WM;
  p.n0 = (p.wave % C::WAVES_N) * C::WN;
  return p;
}
// element (ti, tj, r) of a wave's accumulator sits at tile-local (row, col):
template <class C>
__device__ __forceinline__ int acc_row(const WavePos& p, int ti, int r) {
  if constexpr (C::DIRECT) return p.m0 + C::TM * ((p.lane >> 4) + 4 * r) + ti;        // (rows dealt round-robin to the MFMA tiles)
  else return p.m0 + ti * 16 + (p.lane >> 4) + 4 * r;
}
template <class C>
__device__ __forceinline__ int acc_col(const WavePos& p, int tj) {
  if constexpr (C::DIRECT) return p.n0 + C::TN * (p.lane & 15) + tj;
  else return p.n0 + tj * 16 + (p.lane & 15);
}

// n contiguous doubles (n = 1, 2, 4) at an address that is a multiple of 8 n bytes: one or two 16-byte accesses
// NT: the access carries the non-temporal hint -- a C tile that one workgroup reads once and writes once should not push the
// operand panels that many workgroups share out of the L2 (the filler tiles of the fused sweep: -1.6 % at N=4096; the batched
// deep updates are better off without it: 512 x N=2048 +0.7 %)
template <int N, bool NT = false>
__device__ __forceinline__ void load_run(const double* __restrict__ p, double (&x)[N]) {
  static_assert(N == 1 || N == 2 || N == 4, "run length");
  if constexpr (N == 1) x[0] = NT ? __builtin_nontemporal_load(p) : p[0];
  else {
#pragma unroll
    for (int h = 0; h < N / 2; ++h) {
      const v2d* q = reinterpret_cast<const v2d*>(p + 2 * h);
      const v2d v = NT ? __builtin_nontemporal_load(q) : *q;
      x[2 * h] = v[0]; x[2 * h + 1] = v[1];
    }
  }
}
template <int N, bool NT = false>
__device__ __forceinline__ void store_run(double* __restrict__ p, const double (&x)[N]) {
  if constexpr (N == 1) { if (NT) __builtin_nontemporal_store(x[0], p); else p[0] = x[0]; }
  else {
#pragma unroll
    for (int h = 0; h < N / 2; ++h) {
      v2d* q = reinterpret_cast<v2d*>(p + 2 * h);
      const v2d v = v2d{x[2 * h], x[2 * h + 1]};
      if (NT) __builtin_nontemporal_store(v, q); else *q = v;
    }
  }
}

// the direct loop (see the head of this file).  Addressing is kept off the vector unit: per k-block one buffer descriptor
// per operand (scalar registers) and one 32-bit lane offset, per k-step a scalar row offset -- a load is
// `buffer_load_dwordx4 v, voff, s[desc], soff offen`, no 64-bit address arithmetic per lane (with plain global loads the
// address math of every k-step cost more than the LDS staging it replaced).
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
template <int N>
__device__ __forceinline__ void load_frag(const __amdgpu_buffer_rsrc_t rs, int voff, int soff, double (&x)[N]) {
  static_assert(N == 1 || N == 2 || N == 4, "fragment run");
  if constexpr (N == 1) x[0] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
  else {
#pragma unroll
    for (int h = 0; h < N / 2; ++h) {
      const v2d v = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16 * h, soff, 0));
      x[2 * h] = v[0]; x[2 * h + 1] = v[1];
    }
  }
}
struct OperandBlock { __amdgpu_buffer_rsrc_t ra, rb; int voa, vob, sa, sb; };   // descriptors, lane offsets, bytes per k-step
template <class C, class PtrFn>
__device__ __forceinline__ void gemm_direct(int nkb, PtrFn&& ptrs, v4d (&acc)[C::TM][C::TN], bool negate_late, int skip_ks = 0) {
  const WavePos wp = wave_pos<C>();
  constexpr int PD = C::PF, TM = C::TM, TN = C::TN, KS = NB / 4;      // k-steps per k-block
  static_assert(KS % PD == 0 && PD <= KS, "prefetch distance");
  const int g = wp.lane >> 4, i = wp.lane & 15;
  const int ca = (wp.m0 + TM * i) * 8, cb = (wp.n0 + TN * i) * 8;
  auto block = [&](int kb) {
    const double* pa; const double* pb; int64_t lda, ldb;
    ptrs(kb, pa, lda, pb, ldb);                                        // (uniform: scalar registers)
    OperandBlock o;
    o.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pa), 0, 0x7fffffff, 0x00027000);
    o.rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pb), 0, 0x7fffffff, 0x00027000);
    o.voa = g * (int)lda * 8 + ca; o.vob = g * (int)ldb * 8 + cb;
    o.sa = 4 * (int)lda * 8; o.sb = 4 * (int)ldb * 8;
    return o;
  };
  double a[PD][TM], b[PD][TN];
  auto load = [&](int u, int ks) {                                     // k-step ks -> stage u
    const OperandBlock o = block(ks / KS);                             // (scalar work: it rides in the shadow of the MFMAs)
    const int kr = ks % KS;
    load_frag<TM>(o.ra, o.voa, kr * o.sa, a[u]); load_frag<TN>(o.rb, o.vob, kr * o.sb, b[u]);
  };
  // (skip_ks: k-steps left out at the END of the last k-block -- rows the caller knows to be zero, a multiple of PD)
  const int nks = nkb * KS - skip_ks;
#pragma unroll
  for (int u = 0; u < PD; ++u) load(u, u);
  if (negate_late) {                                                   // acc holds +C from acc_load_raw: its loads and the first
#pragma unroll                                                         // fragments' are one memory round trip instead of two
    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
      for (int tj = 0; tj < TN; ++tj) acc[ti][tj] = -acc[ti][tj];
  }
  for (int ks0 = 0; ks0 < nks; ks0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
#pragma unroll
      for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int tj = 0; tj < TN; ++tj)
          acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][ti], b[u][tj], acc[ti][tj], 0, 0, 0);
      if (ks0 + u + PD < nks) load(u, ks0 + u + PD);                   // the stage is free again
    }
  }
}

// Symmetric tiles (round 4).  A diagonal tile of A^-1 = V^T V -- operands A and B are the same block column -- equals its own
// transpose, and with the rows of a sub-tile dealt round-robin to the MFMA tiles (above) MFMA tile (tj, ti) of a diagonal
// 64x64 quadrant is exactly the transpose of MFMA tile (ti, tj).  So the four wavefronts of a 128x128 tile share the work as
//   wave 0: quadrant (0, 0), MFMA tiles ti <= tj (10 of 16)      wave 1: quadrant (0, 64), MFMA tile rows ti = 0, 1 (8)
//   wave 3: quadrant (64, 64), the same                          wave 2: quadrant (0, 64), MFMA tile rows ti = 2, 3 (8)
// and quadrant (64, 0) is nobody's: 36 MFMAs per k-step instead of 64, at most 10 on any SIMD (0.625 of the time of a full tile).
// The caller weights the tiles in whatever it contracts them with (2 for a tile that stands for its mirror image, 1 on the
// diagonal).  SH: which MFMA tiles of the wavefront's 4x4 are computed; the others are left as they are.
enum TileShape { SH_FULL = 0, SH_UPPER = 1, SH_ROWS_LO = 2, SH_ROWS_HI = 3 };
template <int SH> __host__ __device__ constexpr bool shape_has(int ti, int tj) {
  return SH == SH_FULL ? true : SH == SH_UPPER ? ti <= tj : SH == SH_ROWS_LO ? ti < 2 : ti >= 2;
}
__device__ __forceinline__ int shape_tj_lo(int shape, int ti) {       // the computed MFMA tiles of row ti are tj = shape_tj_lo .. 3
  return shape == SH_FULL ? 0 : shape == SH_UPPER ? ti : shape == SH_ROWS_LO ? (ti < 2 ? 0 : 4) : (ti >= 2 ? 0 : 4);
}
template <class C, int SH, class PtrFn>
__device__ __forceinline__ void gemm_direct_shaped(const WavePos wp, int nkb, PtrFn&& ptrs, v4d (&acc)[C::TM][C::TN], bool negate_late, int skip_ks = 0) {
  static_assert(C::DIRECT && C::TM == 4 && C::TN == 4, "shaped tiles: the direct 64x64-per-wavefront form");
  constexpr int PD = C::PF, TM = C::TM, TN = C::TN, KS = NB / 4;
  constexpr int TA = (SH == SH_ROWS_LO || SH == SH_ROWS_HI) ? 2 : TM, A0 = SH == SH_ROWS_HI ? 2 : 0;   // A fragments: rows A0 .. A0 + TA - 1
  const int g = wp.lane >> 4, i = wp.lane & 15;
  const int ca = (wp.m0 + TM * i + A0) * 8, cb = (wp.n0 + TN * i) * 8;
  auto block = [&](int kb) {
    const double* pa; const double* pb; int64_t lda, ldb;
    ptrs(kb, pa, lda, pb, ldb);
    OperandBlock o;
    o.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pa), 0, 0x7fffffff, 0x00027000);
    o.rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pb), 0, 0x7fffffff, 0x00027000);
    o.voa = g * (int)lda * 8 + ca; o.vob = g * (int)ldb * 8 + cb;
    o.sa = 4 * (int)lda * 8; o.sb = 4 * (int)ldb * 8;
    return o;
  };
  double a[PD][TA], b[PD][TN];
  auto load = [&](int u, int ks) {
    const OperandBlock o = block(ks / KS);
    const int kr = ks % KS;
    load_frag<TA>(o.ra, o.voa, kr * o.sa, a[u]); load_frag<TN>(o.rb, o.vob, kr * o.sb, b[u]);
  };
  const int nks = nkb * KS - skip_ks;
#pragma unroll
  for (int u = 0; u < PD; ++u) load(u, u);
  if (negate_late) {
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
      for (int tj = 0; tj < TN; ++tj) if (shape_has<SH>(ti, tj)) acc[ti][tj] = -acc[ti][tj];
  }
  for (int ks0 = 0; ks0 < nks; ks0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
#pragma unroll
      for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int tj = 0; tj < TN; ++tj)
          if (shape_has<SH>(ti, tj))                           // (folded when the loops are unrolled)
            acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][(ti - A0) & (TA - 1)], b[u][tj], acc[ti][tj], 0, 0, 0);
      if (ks0 + u + PD < nks) load(u, ks0 + u + PD);
    }
  }
}

// the accumulators of a shaped wavefront: its computed MFMA tiles start from +C (c != null; negated inside the loop) or from
// zero; the others are cleared AFTER the loop by acc_clear_outside (so that they are dead registers while it runs)
template <class C, int SH>
__device__ __forceinline__ void acc_init_shaped(const double* __restrict__ c, int64_t ldc, v4d (&acc)[C::TM][C::TN], const WavePos wp) {
#pragma unroll
  for (int ti = 0; ti < C::TM; ++ti) {
    bool any = false;
#pragma unroll
    for (int tj = 0; tj < C::TN; ++tj) any = any || shape_has<SH>(ti, tj);
    if (!any) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double x[C::TN];
      if (c) load_run<C::TN>(c + (int64_t)acc_row<C>(wp, ti, r) * ldc + acc_col<C>(wp, 0), x);
#pragma unroll
      for (int tj = 0; tj < C::TN; ++tj) if (shape_has<SH>(ti, tj)) acc[ti][tj][r] = c ? x[tj] : 0.0;
    }
  }
}
template <class C, int SH>
__device__ __forceinline__ void acc_clear_outside(v4d (&acc)[C::TM][C::TN]) {
#pragma unroll
  for (int ti = 0; ti < C::TM; ++ti)
#pragma unroll
    for (int tj = 0; tj < C::TN; ++tj) if (!shape_has<SH>(ti, tj)) acc[ti][tj] = v4d{0.0, 0.0, 0.0, 0.0};
}

// ptrs(kb, pa, lda, pb, ldb): operand tile pointers of k-block kb: A tile is
// NB x BM at pa (row pitch lda), B tile NB x BN at pb.
// bsplit != 0: the right half of the B tile (columns BN/2..BN-1) sits bsplit columns further along in memory (two
// separate column slabs multiplied as one operand).
template <class C, class PtrFn>
__device__ __forceinline__ void gemm_tn(double* __restrict__ lds, int nkb, PtrFn&& ptrs,
                                        v4d (&acc)[C::TM][C::TN], int bsplit = 0, bool negate_late = false, int skip_ks = 0) {
  if constexpr (C::DIRECT) { gemm_direct<C>(nkb, ptrs, acc, negate_late, skip_ks); return; }   // (bsplit: staged form only; skip_ks: direct form only)
  const int t = threadIdx.x;
  const WavePos wp = wave_pos<C>();
  constexpr int KC = C::KC;
  const int nchunks = nkb * (NB / KC);
  constexpr int D = C::PF;
  v2d ra[D][C::VA], rb[D][C::VB];

  auto gload = [&](int c, v2d (&xa)[C::VA], v2d (&xb)[C::VB]) {
    const int kb = c / (NB / KC), kr = (c % (NB / KC)) * KC;
    const double* pa; const double* pb; int64_t lda, ldb;
    ptrs(kb, pa, lda, pb, ldb);
#pragma unroll
    for (int s = 0; s < C::VA; ++s) {
      const int e = t + C::NT * s, row = e / (C::BM / 2), c2 = e % (C::BM / 2);
      if (C::EA >= C::NT || e < C::EA) xa[s] = *reinterpret_cast<const v2d*>(pa + (int64_t)(kr + row) * lda + 2 * c2);
    }
#pragma unroll
    for (int s = 0; s < C::VB; ++s) {
      const int e = t + C::NT * s, row = e / (C::BN / 2), c2 = e % (C::BN / 2);
      if (C::EB >= C::NT || e < C::EB) xb[s] = *reinterpret_cast<const v2d*>(pb + (int64_t)(kr + row) * ldb + 2 * c2 + (2 * c2 >= C::BN / 2 ? bsplit : 0));
    }
  };
  auto sstore = [&](int buf, const v2d (&xa)[C::VA], const v2d (&xb)[C::VB]) {
    double* As = lds + buf * C::STAGE;
    double* Bs = As + KC * C::PA;
#pragma unroll
    for (int s = 0; s < C::VA; ++s) {
      const int e = t + C::NT * s, row = e / (C::BM / 2), c2 = e % (C::BM / 2);
      if (C::EA >= C::NT || e < C::EA) *reinterpret_cast<v2d*>(As + row * C::PA + 2 * c2) = xa[s];
    }
#pragma unroll
    for (int s = 0; s < C::VB; ++s) {
      const int e = t + C::NT * s, row = e / (C::BN / 2), c2 = e % (C::BN / 2);
      if (C::EB >= C::NT || e < C::EB) *reinterpret_cast<v2d*>(Bs + row * C::PB + 2 * c2) = xb[s];
    }
  };

  // register set u holds chunk c with c % D == u; LDS is double-buffered
#pragma unroll
  for (int u = 0; u < D; ++u) gload(u, ra[u], rb[u]);       // nchunks >= 8 >= D
  if (negate_late) {                                        // acc holds +C from acc_load_raw: its loads and the first chunks' are
#pragma unroll                                              // one memory round trip instead of two
    for (int ti = 0; ti < C::TM; ++ti)
#pragma unroll
      for (int tj = 0; tj < C::TN; ++tj) acc[ti][tj] = -acc[ti][tj];
  }
  sstore(0, ra[0], rb[0]);
  __syncthreads();
  for (int c0 = 0; c0 < nchunks; c0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int c = c0 + u;
      if (c + D < nchunks) gload(c + D, ra[u], rb[u]);       // set u is free: chunk c already sits in LDS
      const double* As = lds + (c & 1) * C::STAGE;
      const double* Bs = As + KC * C::PA;
#pragma unroll
      for (int kk = 0; kk < KC / 4; ++kk) {
        const int krow = kk * 4 + (wp.lane >> 4);
        double a[C::TM], b[C::TN];
#pragma unroll
        for (int ti = 0; ti < C::TM; ++ti) a[ti] = As[krow * C::PA + wp.m0 + ti * 16 + (wp.lane & 15)];
#pragma unroll
        for (int tj = 0; tj < C::TN; ++tj) b[tj] = Bs[krow * C::PB + wp.n0 + tj * 16 + (wp.lane & 15)];
#pragma unroll
        for (int ti = 0; ti < C::TM; ++ti)
#pragma unroll
          for (int tj = 0; tj < C::TN; ++tj)
            acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
      }
      if (c + 1 < nchunks) sstore((c + 1) & 1, ra[(u + 1) % D], rb[(u + 1) % D]);
      __syncthreads();
    }
  }
}

// acc = scale * C; C tile at c, pitch ldc
template <class C, bool NT = false>
__device__ __forceinline__ void acc_load_scaled(const double* __restrict__ c, int64_t ldc, v4d (&acc)[C::TM][C::TN], double scale,
                                                const WavePos wp = wave_pos<C>()) {
  if constexpr (C::DIRECT) {       // a lane's TN columns of one row are contiguous: 16-byte accesses, 16 TN 8-byte runs per row
#pragma unroll
    for (int ti = 0; ti < C::TM; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double x[C::TN];
        load_run<C::TN, NT>(c + (int64_t)acc_row<C>(wp, ti, r) * ldc + acc_col<C>(wp, 0), x);
#pragma unroll
        for (int tj = 0; tj < C::TN; ++tj) acc[ti][tj][r] = scale * x[tj];
      }
  } else {
#pragma unroll
    for (int ti = 0; ti < C::TM; ++ti)
#pragma unroll
      for (int tj = 0; tj < C::TN; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[ti][tj][r] = scale * c[(int64_t)acc_row<C>(wp, ti, r) * ldc + acc_col<C>(wp, tj)];
  }
}
// acc = -C (so that the accumulated result is -(C - A^T B))
template <class C>
__device__ __forceinline__ void acc_load_neg(const double* __restrict__ c, int64_t ldc, v4d (&acc)[C::TM][C::TN]) { acc_load_scaled<C>(c, ldc, acc, -1.0); }
// acc = +C, to be negated by gemm_tn(..., negate_late = true) once the first operand chunks are on their way
template <class C, bool NT = false>
__device__ __forceinline__ void acc_load_raw(const double* __restrict__ c, int64_t ldc, v4d (&acc)[C::TM][C::TN], const WavePos wp = wave_pos<C>()) {
  acc_load_scaled<C, NT>(c, ldc, acc, 1.0, wp);
}
template <class C>
__device__ __forceinline__ void acc_negate(v4d (&acc)[C::TM][C::TN]) {
#pragma unroll
  for (int ti = 0; ti < C::TM; ++ti)
#pragma unroll
    for (int tj = 0; tj < C::TN; ++tj) acc[ti][tj] = -acc[ti][tj];
}
template <class C>
__device__ __forceinline__ void acc_zero(v4d (&acc)[C::TM][C::TN]) {
#pragma unroll
  for (int ti = 0; ti < C::TM; ++ti)
#pragma unroll
    for (int tj = 0; tj < C::TN; ++tj) acc[ti][tj] = v4d{0.0, 0.0, 0.0, 0.0};
}
// C = sign * acc
template <class C, bool NT = false>
__device__ __forceinline__ void acc_store(double* __restrict__ c, int64_t ldc, const v4d (&acc)[C::TM][C::TN], double sign) {
  const WavePos wp = wave_pos<C>();
  if constexpr (C::DIRECT) {
#pragma unroll
    for (int ti = 0; ti < C::TM; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double x[C::TN];
#pragma unroll
        for (int tj = 0; tj < C::TN; ++tj) x[tj] = sign * acc[ti][tj][r];
        store_run<C::TN, NT>(c + (int64_t)acc_row<C>(wp, ti, r) * ldc + acc_col<C>(wp, 0), x);
      }
  } else {
#pragma unroll
    for (int ti = 0; ti < C::TM; ++ti)
#pragma unroll
      for (int tj = 0; tj < C::TN; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          c[(int64_t)acc_row<C>(wp, ti, r) * ldc + acc_col<C>(wp, tj)] = sign * acc[ti][tj][r];
  }
}
