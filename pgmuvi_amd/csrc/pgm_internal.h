// pgm_internal.h -- workspace layout shared by the kernels and the host driver.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "pgm_gemm.h"

constexpr int PGM_MAX_QD = 16;      // Q*d the LDS staging is sized for
constexpr int PGM_MAX_D = 2;
constexpr int64_t PGM_MAX_N = 16384;   // points per light curve (128 block rows): the largest size held to an oracle fixture (include/pgmuvi_hip.h)
constexpr int CRIT_MAX = 48;      // light curves per call the fused sweep is ever used for (batch x block rows <= 128, >= 3 block rows)

// A composed stationary kernel as a sum of products of leaf kernels (pgm_generic.inc); mirrors pgm_kernel_program of the
// C ABI.  Travels by value inside PgmDev (kernel arguments: decoding it touches no memory).
constexpr int KP_MAXL = 6, KP_MAXT = 4, KP_MAXP = 12;
// Trimmed ragged launch sets (PgmDev::trim): the members are sorted by block rows, longest first, so the light curves of one
// block-row count ("class") are neighbours: class c = members z0[c] .. z0[c+1]-1, nb[c] block rows, x[c] workgroups each in this
// launch -- exactly the tiles such a member has.  The trailing update and the inverse/gradient pass run as ONE 1-D grid per
// launch instead of a (tiles of the longest) x (members) box whose workgroups leave at once where a member has no such tile
// (the dispatcher still spends ~20 ns on each).  The dispatcher deals workgroups round-robin to the 8 XCDs; XCD g (workgroups
// w = 8 s + g) works through the members m = g, g + 8, g + 16 ... one after the other, all workgroups of one before the next --
// the placement of xcd_batch_remap, whatever the class sizes.  Travels as a kernel argument (decoding touches no memory beyond
// the kernarg segment).  n = 0: the box.
constexpr int RAG_MAX_CLASS = 64;
struct RagClasses {
  int n;
  int x[RAG_MAX_CLASS];
  unsigned short z0[RAG_MAX_CLASS + 1];
  unsigned char nb[RAG_MAX_CLASS];
};
struct KProg {
  int nleaf, nterm, nparam;
  unsigned char kind[KP_MAXL];       // leaf kind (KLeaf)
  unsigned char dims[KP_MAXL];       // bit mask of the input dimensions the leaf sees
  unsigned char par[KP_MAXL];        // index of the leaf's first parameter in theta
  unsigned char tmask[KP_MAXT];      // leaves of term t (bit mask)
  unsigned char tnscale[KP_MAXT];    // scale parameters of term t ...
  unsigned char tscale[KP_MAXT][3];  // ... and their indices in theta
};

// Device view of one call (passed by value to every kernel).  All problems of a
// batch share sizes; buffer b of a batch sits at base + b * stride.
struct PgmDev {
  int n, np, nb, q, d, qd, dim_order, need_grad, batch;
  int nslot, ntiles, pre_slots, nitems;
  int64_t ld;                                   // == np
  int64_t sA, sDinv, sPre, sVec, sPart, sLogdet, sDpart, sOut;
  double* A;          // [batch][np*np]  upper blocks: K+noise -> U ; strictly lower blocks: V = U^-T
  double* Dinv;       // [batch][nb][2][NB*NB]  0: Uinv_kk ([p][m])   1: Uinv_kk^T = V_kk ([k][n])
  double* crit;       // [min(batch, CRIT_MAX)][NB*NB]  look-ahead of the fused sweep: copy of tile (k, k+1) before its row solve
  double* pre;        // [batch][3*qd + d][np]  cos, sin, x*v per (q,d); raw x per d
  double* r;          // [batch][np]  residual y - mean, consumed by the forward substitution
  double* z;          // [batch][np]  U^-T r
  double* alpha;      // [batch][np]  A^-1 r
  double* logdet;     // [batch][nb]
  double* partials;   // [batch][nitems][nslot]
  double* dpart;      // [batch][AINV_SPLITS][np]  partial column sums of squares of V
  const int4* items;  // [nitems] (i, j, first k-block, k-blocks | flags << 16) of the inverse/gradient pass
  double* R;          // [np*np]  early part of the A^-1 tiles (negated), formed by filler workgroups of the late
                      //          diagonal-block launches (one light curve, fused sweep), or null
  const int4* tasks;  // (i, j, block row p, flags) of those filler workgroups, in launch order
  double* hyp;        // [batch][3*PGM_MAX_QD] copy of (w, mu, v): the in-graph kernels and prediction read this
  double* diagadd;    // [batch][np]  noise_i + scalar noise + jitter
  double* out_small;  // [batch][1 + q + 2*q*d (+pad)]  mll, g_w, g_mu, g_v
  double* out_gnoise; // [batch][np]
  double* out_gmean;  // [batch][np]
  int* info;          // [batch]
  // ragged batches (pgm_mll_value_grad_ragged_f64): light curves of different lengths in one launch set.  All of them share
  // np / nb (the set's block-row count); the ones that are shorter end in identity padding, whole block rows of it if need be.
  const int* nvec;    // [batch] points of light curve b, or null: every light curve has P.n
  const int* cmap;    // [batch] the caller's index of workspace slot b (the sets are formed from a sorted order), or null: b
  int trim_tri;       // 1 (trimmed sets of 64 members and more, whose inverse/gradient pass has one work item per tile): the work items
                      //   of a member are the tiles of ITS block rows in triangular order, (i, j) <-> j (j + 1) / 2 + i, no table
  int trim;           // 1 (ragged launch sets of the panel sweep): nothing is padded beyond a light curve's own block rows -- the
                      //   workgroups of the build, the row solve, the updates and the inverse/gradient pass whose tile lies in a
                      //   block row or column the light curve does not have leave at once, the others stop at its last block row
  int64_t cstride;    // points per light curve slot in the caller's arrays (= n unless ragged); k_finalize, replayed from a graph,
                      //   reads it from outp[8]
  unsigned long long* outp;   // [16] the caller's output pointers of THIS evaluation (mll, g_w, g_mu, g_v, g_noise, g_mean, info), left
                      //     in device memory by k_precompute: k_finalize, replayed from a graph, writes the results there
  int* info_host;     // host-mapped copy of `info` (device address) that the LAST diagonal-block launch fills in, or null
  long long* seq_host; // host-mapped [batch]: the number of the evaluation whose status info_host[b] holds (written after it, system scope)
  long long seq;      // number of THIS evaluation (k_precompute leaves it in outp[7]; the graph's kernels read it from there)
  double jitter, noise_scalar;
  const double *x, *y, *mean, *noise, *noise_scalar_dev, *w, *mu, *v;
  double *mll, *g_w, *g_mu, *g_v, *g_noise, *g_mean;
  int* info_out;
  int items_kc;       // k-blocks per work item of `items` (the split length its table was made with)
  int ainv_from_tiles; // k-blocks per work item (0 = off): diag(A^-1) is taken from the accumulators of the (j, j) tiles' work items
                      //    in the inverse/gradient launch (item number s of the tile -> dpart row s) and the separate column-sum
                      //    pass over V is skipped
  int prebuilt;       // 1: the kernel matrix was built in front of the graph, by k_prebuild together with the per-point factors (short light curves)
  int build_beside;   // 1: k_build builds block row 0 only, the rest of the matrix is built by the spare workgroups of diagonal block 0's launch
  int lauum_sub;      // 1: the inverse/gradient launch runs four quarter-tile workgroups per work item (nitems counts workgroups); 2: sixteen sixteenth-tile ones
  int small_eval;     // 1: k_small evaluated this call (the inverse images' identity padding is completed on demand: k_small_pad)
  int generic;        // 1: the kernel is `prog` (q = its parameter count, qd = 0, theta travels through `w` / `hyp`)
  KProg prog;
};

enum PgmPhase { PH_PRE = 0, PH_BUILD, PH_DIAG, PH_TRSM, PH_UPDATE, PH_LAUUM, PH_FINAL, PH_FUSED, PH_COUNT };

struct pgm_ws {
  int device;
  int64_t max_n, max_np;
  int max_q, max_d, max_batch, max_nb;
  size_t bytes;
  double *A, *Dinv, *pre, *r, *z, *alpha, *logdet, *partials, *hyp, *dpart, *diagadd, *out_small, *out_gnoise, *out_gmean;
  unsigned long long* outp;
  int* info;
  int* ragged_tab;       // device [2 * ragged_cap]: nvec | cmap of the last ragged call, in its sorted order (grown on demand)
  int ragged_cap;
  std::vector<int> ragged_host;   // what ragged_tab holds (a repeated call with the same lengths uploads nothing)
  // work-item tables of the inverse/gradient pass, one per (block rows, batch class, epilogue weight) ever asked for: built once,
  // never rewritten (captured graphs and the launch sets of a ragged batch keep pointing at theirs)
  struct ItemsEntry { int nb, sim_batch, kc, count; double epi; int4* dev; };
  std::vector<ItemsEntry> items_cache;
  int4* items;           // the table of the current call (one of items_cache)
  int items_nb, items_batch, items_count, items_cap, items_kc;
  int64_t part_rows;     // rows of `partials` per problem
  double items_epi, early_epi;  // epilogue weight the two work lists were split for (spectral mixture 3, generic kernels 15)
  // state of the last need_grad evaluation (for pgm_predict_f64)
  PgmDev last;
  bool last_valid;
  double* pred_buf;      // right-hand sides of pgm_predict_f64 (grown on demand)
  size_t pred_bytes;
  int panel;             // block rows per delayed trailing update (k-depth = panel*128); 0 = fused sweep
  int inleft;            // batches: left-looking inside a panel (run_sweep)
  int strips_min;        // batches: k_trsm_strips from this many block rows x light curves on
  int strips;            // batches: row solve by k_trsm_strips
  std::vector<int> rag_nb, rag_z0;   // the current trimmed set's classes: block rows, first member (+ one past the last), longest first
  int ragged_trim;       // ragged batches: the launch sets of the panel sweep are merged, nothing padded beyond a light curve's own block rows (PgmDev::trim)
  int prebuild;          // short light curves: per-point factors and kernel matrix in one launch (k_prebuild)
  int small;             // light curves of at most 128 points (1-D spectral mixture): the whole evaluation in ONE launch (k_small; PGM_SMALL=0: the launch sequence of every other size)
  int trsm16;            // fused sweep: the chain's row solve by k_trsm16 (16 wavefronts, one memory round trip)
  int trsm64;            // fused sweep, a handful of light curves: the row solve by k_trsm64 (128 x 64 slabs, 6 look-ahead workgroups per light curve)
  int upd_big_min;       // k_update: 128x128 tiles from this many tiles x light curves on
  int pairs;             // fused sweep: two-source filler passes allowed (run_sweep)
  int build_beside;      // 1-D spectral mixture, fused sweep: most light curves per call whose matrix below block row 0 is built beside
                         //   diagonal block 0 (PGM_BUILD_BESIDE; 0: never, 1: one light curve only, default 8)
  int lazy, lazy_end;    // fused sweep: lazy plan (run_sweep), and the tile count from which it turns eager
  int lookahead;         // fused sweep: first block row whose successor's diagonal tile is formed inside the row-solve launch
                         //   (no head launch on the chain from there on); >= 64: never
  double* crit;          // (inside the Dinv allocation)
  int bh, bt;            // fused sweep: update-tile budgets of the head and row-solve launches (128x128 tiles)
  int batch_window;      // small batches in the fused sweep: rows per window (-1 auto, 0 never)
  int window;            // big single light curves: rows per window of the windowed fused sweep (0 = plain panels)
  // early inverse pass (fused sweep, one light curve): the late diagonal-block launches have fewer update tiles than
  // CUs; their spare workgroups form  R_ij = sum_p V_pi^T V_pj  over block rows p that are already final
  int lauum_sub_max;     // inverse/gradient pass: quarter-tile workgroups when a call has at most this many work items in all (0: never)
  int lauum_sub16_max;   // ... sixteenth-tile workgroups when it has at most this many (0: never)
  int early;             // 0 = off
  int early_t;           // early inverse-pass tasks also in the row-solve launches' idle CUs: most per launch (PGM_EARLY_T, 0: off)
  int early_nb;          // block rows the tables below were made for (-1: none)
  std::vector<int4> early_host;          // [final work items | filler tasks]
  std::vector<int> early_lo, early_n;    // filler tasks of diagonal-block launch k: [early_lo[k], +early_n[k]) of the task part
  std::vector<int> early_lo_t, early_n_t; // the same for row-solve launch k (64x64 sub-tiles, four workgroups per task)
  int4* early_items;     // device copy
  int early_cap, early_final_n;
  double* Rbuf; size_t R_bytes;
  // hipGraph replay of the launch sequence behind k_precompute
  bool use_graph;
  hipStream_t cap_stream;
  struct GraphEntry { int n, d, q, dim_order, need_grad, batch, panel; int early; uint64_t prog_hash; int parts; const int* nvec; hipGraphExec_t exec; };
  // factorisation status for the host, final as soon as the sweep is (pgm_factorisation_status)
  long long* seq_host;   // host-mapped pinned [max_batch]: evaluation number behind each info_host entry (the host polls it: no event, one graph)
  long long* seq_host_dev;
  int* info_host;        // host-mapped pinned copy of `info`, written by the last diagonal-block launch of the sweep
  int* info_host_dev;    // its device address
  int status_batch;      // problems of the last evaluation that published a status (0: none, or inside a caller's capture)
  int64_t eval_seq;      // evaluations run on this workspace so far: the status belongs to evaluation number eval_seq (pgm_last_evaluation)
  std::vector<GraphEntry> graphs;
  // profiling
  bool prof_on;
  std::vector<hipEvent_t> ev_pool;
  std::vector<int> ev_phase;      // phase of event pair i (events 2i, 2i+1)
  size_t ev_used;
  hipStream_t prof_stream;
  double prof_ms[PH_COUNT];
  int64_t prof_launches[PH_COUNT];
};
