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
static int g_follower_tail = 2;   // emul_set_follower_tail: the follower's last block right-looking (blocked.hpp::potrf_rl_groups)
static int g_tall = 0;        // emul_set_tall: solves of more than 512 columns panel by panel, left-looking (blocked.hpp::tall_panels)
static int g_leaf512 = 0;     // emul_set_leaf512: 512-column nodes of the triangular solves as one operation (Ops::trsm_leaf512)
static i64 g_rl_max = 0;      // emul_set_rl_max: size up to which potrf_rec takes the right-looking sweep
static int g_fused = 0;       // emul_set_fused: the sweep's one-launch steps (Ops::step)
static int g_bulk = 0;        // emul_set_bulk: cross-level look-ahead -- bit 0: piece (a), bit 1: piece (b); rows of a (b) piece below
static i64 g_bulk_rows = 256;
static int g_side_bad = 0;    // operations of the chain that touched what a side section had written and the chain had not joined yet
static int g_bulk_pieces = 0, g_bulk_bad = 0;      // pieces issued / pairing or race violations seen (read by the tests)

struct CpuOps {
  std::vector<double> linv, linvT;
  int info = 0;
  int n_gemm = 0, n_base = 0;
  explicit CpuOps(i64 nblk) : linv(nblk * T * T, 0.0), linvT(nblk * T * T, 0.0) {}

  int potrf_base(double* A, i64 lda, i64 blk, i64 row0) {
    if (join_armed) return -31;                 // a join handed to "the next GEMM" must be followed by one
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
    join_armed = false;                         // (a pending fused join is consumed by this launch)
    if (M % T || N % T || K % 16) return -1;
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
  // one launch per 128 columns (blocked.hpp: Ops::step): here the three operations it stands for, in order
  int n_step = 0;
  bool fused_step() { return (g_fused & 1) != 0; }
  bool two_stage_join() const { return (g_fused & 2) != 0; }
  int step_join(unsigned long long t) { return chain_join(t); }
  // a join value the NEXT step needs, awaited by this step before it ends (after its own work): it takes effect when the
  // step has run AND the value has been published, whichever comes last
  unsigned long long carry = 0;
  bool carry_armed = false;                                     // step_carry_join must be followed by the step that carries it
  bool step_exit_join() const { return (g_fused & 8) != 0; }
  int step_carry_join(unsigned long long v) { if (carry || v <= joined) return -52; carry = v; carry_armed = true; return 0; }
  void apply_carry() { if (carry && !carry_armed && published >= carry) { (void)chain_join(carry); carry = 0; } }
  int step(i64 blk, double* B, i64 ldb, i64 m, i64 kprev, i64 row0_next) {
    if (m < T || open_side || fol_open || def_open) return -51;
    ++n_step;
    int rc = trsm_base(blk, 0, B, ldb, m);
    if (!rc) rc = gemm(0, 0, m, T, kprev + T, B - kprev, ldb, B - kprev, ldb, B + T, ldb);
    if (!rc) rc = potrf_base(B + T, ldb, blk + 1, row0_next);
    carry_armed = false;
    apply_carry();
    return rc;
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
  int follower_tail() const { return g_follower_tail; }
  bool trail_follows() const { return g_lookahead == 2; }        // (both forms of the trailing update are emulated)
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
    apply_carry();
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
  // cross-level look-ahead (blocked.hpp: pieces (a) and (b)).  Sequential here -- a piece is executed when it is issued --, so
  // what the hooks check is (i) the pairing: one piece in flight at a time, every piece joined, and (ii) a race detector by
  // regions: every rectangle of the matrix a piece reads or writes is recorded, and until the join no operation of the
  // calling stream may write a rectangle the piece touches or read one it writes (touch()).
  int bulk_state = 0;            // 0 idle, 1 open (launches go to the bulk stream), 2 in flight (closed, not joined)
  std::vector<Rect> bulk_reads, bulk_writes;
  const double* base = nullptr; i64 base_ld = 0, base_rows = 0;        // the matrix being factored (set by the entry points below)
  static bool overlap(const Rect& a, const Rect& b) { return a.r0 < b.r0 + b.nr && b.r0 < a.r0 + a.nr && a.c0 < b.c0 + b.nc && b.c0 < a.c0 + a.nc; }
  void touch(const double* p, i64 ld, i64 nr, i64 nc, bool write) {
    if (!base || ld != base_ld || p < base || p >= base + base_rows * base_ld) return;      // (block inverses etc.: not in the matrix)
    const Rect r{(i64)((p - base) / base_ld), (i64)((p - base) % base_ld), nr, nc};
    if (open_side) { side_pending.push_back(SideRect{r, 0, !write}); return; }
    if (!fol_open && !def_open && bulk_state != 1)             // an operation of the chain
      for (const SideRect& q : side_pending) if ((write || !q.rd) && overlap(r, q.r)) ++g_side_bad;
    if (bulk_state == 1) { (write ? bulk_writes : bulk_reads).push_back(r); return; }
    if (bulk_state != 2) return;
    for (const Rect& w : bulk_writes) if (overlap(r, w)) ++g_bulk_bad;
    if (write) for (const Rect& q : bulk_reads) if (overlap(r, q)) ++g_bulk_bad;
  }
  bool bulk() { return g_bulk != 0 && bulk_state == 0; }
  bool bulk_rest() { return (g_bulk & 1) && bulk(); }
  i64 bulk_chunk(i64 k) const { return k > 256 ? 256 : k; }
  i64 bulk_rows(i64, i64 m) { if (!(g_bulk & 2) || !bulk()) return 0; return g_bulk_rows < m ? g_bulk_rows : m; }
  int n_bulk_open = 0;
  int bulk_open() { if (bulk_state != 0 || open_side || def_open || fol_open) { ++g_bulk_bad; return -41; } bulk_state = 1; ++n_bulk_open; return 0; }
  int bulk_close() { if (bulk_state != 1) { ++g_bulk_bad; return -42; } bulk_state = 2; return 0; }
  int bulk_join() {
    if (bulk_state != 2) { ++g_bulk_bad; return -43; }
    bulk_reads.clear(); bulk_writes.clear();
    bulk_state = 0;
    return 0;
  }
  int chain_join(unsigned long long v) {
    if (g_fused & 4) return 0;                                  // (self-test of the race detector: an Ops that forgets to join)
    if (v > published || v <= joined) return -9;              // a join for something that was never published / joined twice
    joined = v;
    std::vector<SideRect> keep;
    for (auto& q : side_pending) if (q.v > v) keep.push_back(q);
    side_pending.swap(keep);
    return 0;
  }
  // (the join carried by the next GEMM: that launch must follow at once)
  bool join_armed = false;
  int chain_join_next_gemm(unsigned long long t) { int rc = chain_join(t); join_armed = (rc == 0); return rc; }
  // forward substitution following the factorisation: sequential here; the hooks check pairing
  int y_opened = 0, y_sections = 0;
  int y_open() { if (y_opened || open_side || def_open || fol_open) return -21; y_opened = 1; ++y_sections; return 0; }
  int y_close() { if (!y_opened) return -22; y_opened = 0; return 0; }
  int y_prepare(i64, i64) { return y_opened ? 0 : -23; }
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
void emul_set_follower_tail(int v) { g_follower_tail = v; }
void emul_set_rl_group(i64 v) { g_rl_group = v; }
void emul_set_lookahead(int v) { g_lookahead = v; }
void emul_set_fused(int v) { g_fused = v; }
int emul_side_bad(int reset) { const int v = g_side_bad; if (reset) g_side_bad = 0; return v; }
void emul_set_bulk(int v, i64 rows) { g_bulk = v; g_bulk_rows = rows; g_bulk_pieces = g_bulk_bad = 0; }
void emul_bulk_counts(int* pieces, int* bad) { *pieces = g_bulk_pieces; *bad = g_bulk_bad; }
// the race detector itself: a piece that writes a block the calling stream then reads before the join must be flagged (1),
// the same with the join in between must not (0)
int emul_bulk_selftest(int with_join) {
  const i64 n = 3 * T;
  std::vector<double> A((size_t)n * n, 1.0);
  CpuOps ops(n / T);
  ops.base = A.data(); ops.base_ld = n; ops.base_rows = n;
  g_bulk_bad = 0;
  const int saved = g_bulk; g_bulk = 3;
  int rc = ops.bulk_open();
  if (!rc) rc = ops.gemm(0, 0, T, T, T, A.data() + T * n, n, A.data() + T * n, n, A.data() + 2 * T * n + 2 * T, n);     // piece writes block (2, 2)
  if (!rc) rc = ops.bulk_close();
  if (!rc && with_join) rc = ops.bulk_join();
  if (!rc) rc = ops.gemm(0, 0, T, T, T, A.data() + 2 * T * n + 2 * T, n, A.data(), n, A.data() + T * n, n);               // calling stream reads block (2, 2)
  g_bulk = saved;
  return rc ? -1 : (g_bulk_bad > 0 ? 1 : 0);
}
// A [(n + e), n] in place with the cross-level look-ahead hooks armed: factor (+ augmented rows), count the bulk pieces and
// the violations of the race detector; a piece still in flight at the end is a violation
int emul_potrf_bulk(double* A, i64 n, i64 e, int* info) {
  CpuOps ops(n / T);
  ops.base = A; ops.base_ld = n; ops.base_rows = n + e;
  Blocked<CpuOps> bl(ops);
  int rc = bl.potrf_rec(A, n, n, 0, 0, nullptr, e);
  *info = ops.info;
  g_bulk_pieces += ops.n_bulk_open;
  if (ops.bulk_state != 0) ++g_bulk_bad;
  return rc;
}
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
// A [(n + e), n] in place: rows 0..n-1 SPD -> L, rows n..n+e-1 (e a multiple of 128) -> E L^-T  (augmented rows)
int emul_potrf_aug(double* A, i64 n, i64 e, int* info) {
  CpuOps ops(n / T);
  ops.base = A; ops.base_ld = n; ops.base_rows = n + e;
  Blocked<CpuOps> bl(ops);
  int rc = bl.potrf_rec(A, n, n, 0, 0, nullptr, e);
  *info = ops.info;
  return rc;
}
// A [n, n] in place -> L, with y [r][n] -> L^-1 y issued block by block behind the factorisation (YFollow)
int emul_potrf_yfollow(double* A, i64 n, double* y, i64 r, int* info, int* sections) {
  CpuOps ops(n / T);
  Blocked<CpuOps> bl(ops);
  Blocked<CpuOps>::YFollow yf{y, n, r};
  int rc = bl.potrf_rec(A, n, n, 0, 0, nullptr, 0, &yf);
  *info = ops.info; *sections = ops.y_sections;
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
