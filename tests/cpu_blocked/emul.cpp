// TEST INFRASTRUCTURE ONLY -- never linked into libgpflowslim_hip.so.
// Instantiates gpflow-slim_amd/csrc/blocked.hpp (the recursion used by the product) with naive
// host loops in place of the HIP kernels, so that the block/index arithmetic of potrf_rec /
// trsm_rec / trsm_rn_rec / trsv_rec can be checked against scipy on a machine without a GPU.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../gpflow-slim_amd/csrc/blocked.hpp"

typedef int64_t i64;
static const i64 T = GPS_TILE;

static i64 g_rl_group = 1;
static int g_lookahead = 0;
static int g_tall = 0;        // emul_set_tall: solves of more than 512 columns panel by panel, left-looking (blocked.hpp::tall_panels)
static int g_leaf512 = 0;     // emul_set_leaf512: 512-column nodes of the triangular solves as one operation (Ops::trsm_leaf512)
static i64 g_rl_max = 0;      // emul_set_rl_max: size up to which potrf_rec takes the right-looking sweep
static int g_side_bad = 0;    // operations of the chain that touched what a side section had written and the chain had not joined yet
static int g_forget_join = 0; // emul_set_forget_join: self-test of the race detector -- an Ops whose chain_join does nothing

struct CpuOps {
  std::vector<double> linv, linvT;
  int info = 0;
  int n_gemm = 0, n_base = 0;
  explicit CpuOps(i64 nblk) : linv(nblk * T * T, 0.0), linvT(nblk * T * T, 0.0) {}

  int potrf_base(double* A, i64 lda, i64 blk, i64 row0) {
    ++n_base;
    touch(A, lda, T, T, true);
    for (i64 j = 0; j < T; ++j) {
      double d = A[j * lda + j];
      for (i64 k = 0; k < j; ++k) d -= A[j * lda + k] * A[j * lda + k];
      if (!(d > 0.0) && info == 0) info = (int)(row0 + j + 1);
      d = std::sqrt(d);
      A[j * lda + j] = d;
      for (i64 i = j + 1; i < T; ++i) {
        double s = A[i * lda + j];
        for (i64 k = 0; k < j; ++k) s -= A[i * lda + k] * A[j * lda + k];
        A[i * lda + j] = s / d;
      }
    }
    double* X = &linv[blk * T * T];
    double* XT = &linvT[blk * T * T];
    for (i64 c = 0; c < T; ++c) {
      for (i64 i = 0; i < T; ++i) {
        double s = (i == c) ? 1.0 : 0.0;
        for (i64 k = c; k < i; ++k) s -= A[i * lda + k] * X[k * T + c];
        X[i * T + c] = (i < c) ? 0.0 : s / A[i * lda + i];
      }
    }
    for (i64 i = 0; i < T; ++i) for (i64 c = 0; c < T; ++c) XT[c * T + i] = X[i * T + c];
    return 0;
  }
  int gemm(int op, int lower, i64 M, i64 N, i64 K, const double* A, i64 lda, const double* B, i64 ldb,
           double* C, i64 ldc) {
    ++n_gemm;
    if ((M % T && !(M % 64 == 0 && lower == 0)) || N % T || K % 16) return -1;      // (half a tile row of right-hand sides: as the device launcher)
    touch(A, lda, M, K, false); touch(B, ldb, N, K, false); touch(C, ldc, M, N, true);
    if (op != 1 && op != 3) touch(C, ldc, M, N, false);
    std::vector<double> out((size_t)M * N, 0.0);
    std::vector<char> done((size_t)M * N, 0);
    for (i64 i = 0; i < M; ++i)
      for (i64 j = 0; j < N; ++j) {
        if (lower == 1 && (j / T) > (i / T)) continue;
        double s = 0.0;
        // lower == 2: A is upper triangular; the device kernel never reads A's blocks left of the diagonal block
        for (i64 k = (lower == 2) ? (i / T) * T : 0; k < K; ++k) s += A[i * lda + k] * B[j * ldb + k];
        out[i * N + j] = s; done[i * N + j] = 1;
      }
    for (i64 i = 0; i < M; ++i)
      for (i64 j = 0; j < N; ++j)
        if (done[i * N + j])
          C[i * ldc + j] = (op == 0) ? C[i * ldc + j] - out[i * N + j] : (op == 2 ? C[i * ldc + j] + out[i * N + j] : (op == 3 ? -out[i * N + j] : out[i * N + j]));
    return 0;
  }
  // a product with a triangular operand and / or a batch of them (blocked.hpp: wide_inverse, trsm_wide_rec).  As the device kernel:
  // the zero part of a triangular operand beyond its diagonal 128-blocks is never READ (the tests poison it with NaN).
  int n_gemm_ex = 0;
  int gemm_ex(int op, int tri, i64 M, i64 N, i64 K, const double* A, i64 lda, const double* B, i64 ldb, double* C, i64 ldc, const GemmBatch* bt) {
    ++n_gemm_ex;
    if (M % 64 || N % T || K % 16 || (tri && (tri == 3 ? N : M) != K)) return -61;
    if (tri < 0 || tri > 3 || (op != 1 && op != 3 && op != 0)) return -62;
    const i64 nb = bt ? bt->batch : 1;
    for (i64 p = 0; p < nb; ++p) {
      const double* a = A; const double* b = B; double* c = C;
      if (bt) {
        a += p * bt->a_rs * lda + (bt->a_cm ? (p * bt->a_cs) % bt->a_cm : p * bt->a_cs);
        b += p * bt->b_rs * ldb + (bt->b_cm ? (p * bt->b_cs) % bt->b_cm : p * bt->b_cs);
        c += p * bt->c_rs * ldc + (bt->c_cm ? (p * bt->c_cs) % bt->c_cm : p * bt->c_cs);
      }
      std::vector<double> out((size_t)M * N);
      for (i64 i = 0; i < M; ++i)
        for (i64 j = 0; j < N; ++j) {
          i64 k0 = 0, k1 = K;                                   // K range at the granularity of the 128-blocks
          if (tri == 1) k0 = (i / T) * T;
          if (tri == 2) k1 = (i / T + 1) * T;
          if (tri == 3) k1 = (j / T + 1) * T;
          double s = 0.0;
          for (i64 k = k0; k < k1; ++k) s += a[i * lda + k] * b[j * ldb + k];
          out[i * N + j] = s;
        }
      for (i64 i = 0; i < M; ++i)
        for (i64 j = 0; j < N; ++j)
          c[i * ldc + j] = (op == 0) ? c[i * ldc + j] - out[i * N + j] : (op == 3 ? -out[i * N + j] : out[i * N + j]);
    }
    return 0;
  }
  int blocks_to_diag(const double* src, double* dst, i64 nblk, i64 wb) {
    for (i64 b = 0; b < nblk; ++b)
      for (i64 i = 0; i < T; ++i)
        for (i64 j = 0; j < T; ++j) dst[(b * T + i) * wb + (b * T) % wb + j] = src[b * T * T + i * T + j];
    return 0;
  }
  int trsv_t_base(i64 blk, double* y, i64 ldy, i64 r, const double* = nullptr, i64 = 0) {
    const double* W = linv.data() + blk * T * T;
    for (i64 q = 0; q < r; ++q) {
      double tmp[GPS_TILE];
      for (i64 i = 0; i < T; ++i) { double s = 0; for (i64 c = 0; c < T; ++c) s += W[c * T + i] * y[q * ldy + c]; tmp[i] = s; }
      for (i64 i = 0; i < T; ++i) y[q * ldy + i] = tmp[i];
    }
    return 0;
  }
  int gemv_t_sub(const double* L21, i64 ldl, i64 n2, i64 n1, const double* y2, double* y1, i64 ldy, i64 r) {
    for (i64 q = 0; q < r; ++q)
      for (i64 k = 0; k < n1; ++k) {
        double s = 0; for (i64 i = 0; i < n2; ++i) s += L21[i * ldl + k] * y2[q * ldy + i];
        y1[q * ldy + k] -= s;
      }
    return 0;
  }
  int copy_linvT(i64 blk, double* Y, i64 ldy) {
    const double* W = linvT.data() + blk * T * T;
    for (i64 i = 0; i < T; ++i) for (i64 c = 0; c < T; ++c) Y[i * ldy + c] = W[i * T + c];
    return 0;
  }
  i64 rl_max() const { return g_rl_max; }
  i64 rl_group() const { return g_rl_group; }
  // look-ahead hooks: the host emulation is sequential; the hooks check that forks, side sections and joins pair up
  unsigned long long ticket = 0, open_side = 0, unjoined = 0;
  bool lookahead() { return g_lookahead != 0; }
  i64 lookahead_min_rows() const { return 128; }
  unsigned long long la_fork() { return ++ticket; }
  bool follower() { return g_lookahead != 0; }
  int def_open = 0, def_unjoined = 0;
  bool deferred() { return g_lookahead != 0; }
  int deferred_open() { if (def_open || def_unjoined || open_side) return -13; def_open = 1; return 0; }
  int deferred_close() { if (!def_open) return -14; def_open = 0; def_unjoined = 1; return 0; }
  int deferred_join() { if (!def_unjoined) return -15; def_unjoined = 0; return 0; }
  i64 follower_cols() const { return g_lookahead == 2 ? 512 : 256; }
  unsigned long long fol_pub = 0;
  // side sections: what they write stays "in flight" until the chain has joined the join value published after it (values
  // grow: a section may publish several, e.g. after the first block column of a remainder update and after the rest); until
  // then no operation of the chain may touch it (touch(), g_side_bad)
  struct Rect { i64 r0, c0, nr, nc; };
  unsigned long long published = 0, joined = 0;
  struct SideRect { Rect r; unsigned long long v; bool rd; };       // v = 0: touched, not yet published; rd: only read
  std::vector<SideRect> side_pending;
  int side_open(unsigned long long t, bool) { if (open_side || t != ticket) return -7; open_side = t; return 0; }
  int side_publish_join(unsigned long long v) {
    if (!open_side || v <= published) return -8;
    published = v;
    for (auto& q : side_pending) if (q.v == 0) q.v = v;
    return 0;
  }
  int side_close() {
    if (!open_side) return -10;
    for (auto& q : side_pending) if (q.v == 0) return -18;           // (something written in the section was never published)
    open_side = 0; return 0;
  }
  int fol_open = 0;
  int follower_open(unsigned long long t, bool) { if (fol_open || def_open || t != ticket) return -16; fol_open = 1; return 0; }
  int follower_close() { if (!fol_open) return -17; fol_open = 0; return 0; }
  int follower_publish() { if (!fol_open) return -11; ++fol_pub; return 0; }
  int follower_join() { if (open_side || fol_open || fol_pub == 0) return -12; return 0; }
  const double* base = nullptr; i64 base_ld = 0, base_rows = 0;        // the matrix being factored (set by the entry points below)
  static bool overlap(const Rect& a, const Rect& b) { return a.r0 < b.r0 + b.nr && b.r0 < a.r0 + a.nr && a.c0 < b.c0 + b.nc && b.c0 < a.c0 + a.nc; }
  void touch(const double* p, i64 ld, i64 nr, i64 nc, bool write) {
    if (!base || ld != base_ld || p < base || p >= base + base_rows * base_ld) return;      // (block inverses etc.: not in the matrix)
    const Rect r{(i64)((p - base) / base_ld), (i64)((p - base) % base_ld), nr, nc};
    if (open_side) { side_pending.push_back(SideRect{r, 0, !write}); return; }
    if (!fol_open && !def_open)                                // an operation of the chain
      for (const SideRect& q : side_pending) if ((write || !q.rd) && overlap(r, q.r)) ++g_side_bad;
  }
  int chain_join(unsigned long long v) {
    if (g_forget_join) return 0;                                // (self-test of the race detector: an Ops that forgets to join)
    if (v > published || v <= joined) return -9;              // a join for something that was never published / joined twice
    joined = v;
    std::vector<SideRect> keep;
    for (auto& q : side_pending) if (q.v > v) keep.push_back(q);
    side_pending.swap(keep);
    return 0;
  }
  bool fill_zeros() const { return true; }
  int zero_block(double* Y, i64 ldy, i64 rows, i64 cols) {
    for (i64 i = 0; i < rows; ++i) for (i64 c = 0; c < cols; ++c) Y[i * ldy + c] = 0.0;
    return 0;
  }
  int trsm_base(i64 blk, int transposed, double* B, i64 ldb, i64 m, const double* = nullptr, i64 = 0) {
    const double* W = (transposed ? linvT.data() : linv.data()) + blk * T * T;
    return gemm(1, 0, m, T, T, B, ldb, W, T, B, ldb);
  }
  // the 512-column node as the device does it (csrc/trsm_panel.hip): block substitution with the four block inverses and
  // the off-diagonal blocks of D (forward: the lower block L; backward: U = L^T, blocks above the diagonal)
  int n_leaf512 = 0;
  bool leaf512(i64 m, int, i64) const { return g_leaf512 != 0 && m % 64 == 0; }
  bool trsm_left_looking(i64, i64) const { return g_tall != 0; }
  int trsm_leaf512(i64 blk, int transposed, double* B, i64 ldb, i64 m, const double* D, i64 ldd) {
    ++n_leaf512;
    for (int jj = 0; jj < 4; ++jj) {
      const int j = transposed ? 3 - jj : jj;
      int rc = trsm_base(blk + j, transposed, B + j * T, ldb, m);
      if (rc) return rc;
      for (int i = 0; i < 4; ++i) {
        if (transposed ? (i >= j) : (i <= j)) continue;
        rc = gemm(0, 0, m, T, T, B + j * T, ldb, D + i * T * ldd + j * T, ldd, B + i * T, ldb);      // B_i -= X_j D_ij^T
        if (rc) return rc;
      }
    }
    return 0;
  }
  int trsv_base(i64 blk, double* y, i64 ldy, i64 r, const double* = nullptr, i64 = 0) {
    const double* W = linv.data() + blk * T * T;
    for (i64 q = 0; q < r; ++q) {
      double tmp[GPS_TILE];
      for (i64 i = 0; i < T; ++i) { double s = 0; for (i64 c = 0; c < T; ++c) s += W[i * T + c] * y[q * ldy + c]; tmp[i] = s; }
      for (i64 i = 0; i < T; ++i) y[q * ldy + i] = tmp[i];
    }
    return 0;
  }
  int gemv_sub(const double* L21, i64 ldl, i64 n2, i64 n1, const double* y1, double* y2, i64 ldy, i64 r) {
    for (i64 q = 0; q < r; ++q)
      for (i64 i = 0; i < n2; ++i) {
        double s = 0; for (i64 k = 0; k < n1; ++k) s += L21[i * ldl + k] * y1[q * ldy + k];
        y2[q * ldy + i] -= s;
      }
    return 0;
  }
};

extern "C" {
void emul_set_rl_max(i64 v) { g_rl_max = v; }
void emul_set_leaf512(int v) { g_leaf512 = v; }
void emul_set_tall(int v) { g_tall = v; }
void emul_set_rl_group(i64 v) { g_rl_group = v; }
void emul_set_lookahead(int v) { g_lookahead = v; }
void emul_set_forget_join(int v) { g_forget_join = v; }
int emul_side_bad(int reset) { const int v = g_side_bad; if (reset) g_side_bad = 0; return v; }
// the general panel sweep (panels of nb columns factored by potrf_rec, solved by trsm_rec): index check only
int emul_potrf_rl(double* A, i64 n, i64 nb, int* info) {
  CpuOps ops(n / T);
  Blocked<CpuOps> bl(ops);
  int rc = bl.potrf_rl(A, n, n, nb, 0, 0);
  *info = ops.info;
  return rc;
}
// A [n,n] in place -> L (lower valid); B [m,n]: X L^T = B ; B2 [m,n]: X L = B ; y [r][n]: L a = y
int emul_all(double* A, i64 n, double* B, double* B2, i64 m, double* y, i64 r, int* info) {
  CpuOps ops(n / T);
  ops.base = A; ops.base_ld = n; ops.base_rows = n;
  Blocked<CpuOps> bl(ops);
  int rc = bl.potrf_rec(A, n, n, 0, 0);
  if (rc) return rc;
  *info = ops.info;
  rc = bl.trsm_rec(A, n, n, 0, B, n, m);
  if (rc) return rc;
  std::vector<double> U((size_t)n * n, 0.0);
  for (i64 i = 0; i < n; ++i) for (i64 j = 0; j <= i; ++j) U[j * n + i] = A[i * n + j];
  rc = bl.trsm_rn_rec(U.data(), n, n, 0, B2, n, m);
  if (rc) return rc;
  return bl.trsv_rec(A, n, n, 0, y, n, r);
}
// A [n, n] SPD in place -> L; then the wide inverse blocks of its first nf = (n / wb) wb columns (W, Wt: [nf, wb], NaN where
// nothing may be read or is written; T: scratch) and X L^T = B for those columns against them (B [m, n] destroyed, X [m, n] out)
int emul_wide(double* A, i64 n, i64 wb, double* W, double* Wt, double* Tm, double* B, double* X, i64 m, int* info) {
  CpuOps ops(n / T);
  Blocked<CpuOps> bl(ops);
  int rc = bl.potrf_rec(A, n, n, 0, 0);
  *info = ops.info;
  if (rc) return rc;
  const i64 nf = (n / wb) * wb;
  rc = bl.wide_inverse(A, n, nf, wb, ops.linv.data(), ops.linvT.data(), W, Wt, Tm);
  if (rc) return rc;
  return bl.trsm_wide_rec(A, n, 0, nf, wb, W, B, X, n, m);
}
// A [(n + e), n] in place: rows 0..n-1 SPD -> L, rows n..n+e-1 (e a multiple of 128) -> E L^-T  (augmented rows)
int emul_potrf_aug(double* A, i64 n, i64 e, int* info) {
  CpuOps ops(n / T);
  ops.base = A; ops.base_ld = n; ops.base_rows = n + e;
  Blocked<CpuOps> bl(ops);
  int rc = bl.potrf_rec(A, n, n, 0, 0, nullptr, e);
  *info = ops.info;
  return rc;
}
// A [n,n] SPD in place -> L ; yt [r][n]: L^T a = yt ; Kinv [n,n] (lower valid) = A^-1 ; Yout = L^-T
int emul_grad_pieces(double* A, i64 n, double* yt, i64 r, double* Yout, double* Kinv) {
  CpuOps ops(n / T);
  Blocked<CpuOps> bl(ops);
  int rc = bl.potrf_rec(A, n, n, 0, 0);
  if (rc) return rc;
  rc = bl.trsv_t_rec(A, n, n, 0, yt, n, r);
  if (rc) return rc;
  for (i64 i = 0; i < n * n; ++i) { Yout[i] = NAN; Kinv[i] = NAN; }
  rc = bl.inv_t_rec(A, n, n, 0, Yout, n);
  if (rc) return rc;
  return bl.lauum_rec(Yout, n, n, Kinv, n);
}
}
