// emu_consumers.cpp — TEST INFRASTRUCTURE: the output consumers of SURVEY section 8 f4 with their kernel BODIES executed on the host (tests/cpp/hip_cpu_emu.hpp)
// and the library's own host arithmetic (quad_plan.hpp, consumer_kernels.hpp) in front of them — what nnhip_cumtrapz_fn_batch_f64_dev,
// nnhip_cumsimpson_fn_batch_f64_dev, nnhip_cumtrapz_batch_f64_dev, nnhip_cumsimpson_batch_f64_dev, nnhip_hermite_spline_eval_batch_f64_dev and
// nnhip_hermite_spline_slopes_f64_dev do, minus the uploads.  tests/test_kernel_bodies_on_cpu.py compares the output with the vectors an execution of the
// reference's text produced (tests/golden/reference_text_quad_vectors.json), bit for bit.  One request per line on stdin (hex floats):
//   fn <rule 0 trapz | 1 simpson> <dim 1|3> <layout 0 SoA | 1 AoS> <N> <dx> <p0> <p1> <p2> <n_x> X...      -> "rc <code> rows <r>" + one line per row: N*dim values
//   trapz <n> <M> X... Y[n][M]...        X in the CALLER's order (sortAndTrimDataset runs here as in the entries)     -> one line of M values per result row
//   simpson <n> <M> X... Y[n][M]...                                                                          -> one line per result row (the caller's abscissae, the caller's order)
//   slopes <n> <M> X... Y[n][M]...                                                                           -> one line per sorted, trimmed knot
//   (a refusal — NaN in X, impure duplicates, too few distinct abscissae — prints one line "error <why>")
//   eval <n> <M> <n_q> <deriv> <extrap> <extrap_value> X... Y[n][M]... dY[n][M]... xq...                     -> n_q lines of M values
// The integrand of `fn` is the one the GPU test compiles at run time: f_c(x) = ((p0 x + p1) x) (1 + c) + p2.
#include "consumer_kernels.hpp"
#include "dataset_plan.hpp"
#include "quad_plan.hpp"

#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

using namespace nnhip;

template <int DIM>
struct RhsPoly {
  static constexpr int dim = DIM;
  NNHIP_DEV static void eval(double t, const double (&)[DIM], double (&dy)[DIM], const Params& P) {
    for (int c = 0; c < DIM; ++c) dy[c] = ((P.p[0] * t + P.p[1]) * t) * (1.0 + (double)c) + P.p[2];
  }
};

static double rd(std::istringstream& in) {
  std::string w;
  in >> w;
  return std::strtod(w.c_str(), nullptr);  // hex floats
}
static void rdv(std::istringstream& in, std::vector<double>& v, size_t n) {
  v.resize(n);
  for (auto& x : v) x = rd(in);
}
static void print_rows(const std::vector<double>& out, size_t rows, size_t width) {
  for (size_t r = 0; r < rows; ++r) {
    for (size_t k = 0; k < width; ++k) std::printf("%s%a", k ? " " : "", out[r * width + k]);
    std::printf("\n");
  }
}

template <int DIM>
static void launch_fn(int rule, const QuadArgs& a) {
  const dim3 grid((unsigned)((a.N + kBlock - 1) / kBlock)), block(kBlock);
  if (rule == 0) hipemu::launch(cumtrapz_fn_kernel<RhsPoly<DIM>>, grid, block, a);
  else hipemu::launch(cumsimpson_fn_kernel<RhsPoly<DIM>>, grid, block, a);
}

static int do_fn(std::istringstream& in) {
  int rule, dim, layout, n_x;
  int64_t N;
  in >> rule >> dim >> layout >> N;
  const double dx = rd(in);
  double p[3] = {rd(in), rd(in), rd(in)};
  in >> n_x;
  std::vector<double> X;
  rdv(in, X, (size_t)n_x);
  CumquadPlan pl;
  std::string why;
  const int rc = plan_cumquad(rule, X.data(), n_x, dx, pl, why);
  if (rc != NNHIP_OK) { std::printf("rc %d rows 0\n", rc); return 0; }
  QuadArgs& a = pl.a;
  std::vector<double> out((size_t)pl.rows.nRows * (size_t)N * (size_t)dim, -7.0);
  a.out = out.data(); a.N = N;
  a.ivpStride = layout == 0 ? 1 : dim; a.compStride = layout == 0 ? N : 1; a.rowStride = N * dim;
  for (int k = 0; k < 3; ++k) a.P.p[k] = p[k];
  a.emits = pl.rows.emits.data(); a.lastRows = pl.rows.lastRows.data();
  a.pairs = pl.pairs.data(); a.pts = pl.pts.data();
  if (N > 0 && pl.rows.nRows > 0) { if (dim == 1) launch_fn<1>(rule, a); else launch_fn<3>(rule, a); }
  std::printf("rc 0 rows %d\n", pl.rows.nRows);
  print_rows(out, (size_t)pl.rows.nRows, (size_t)N * (size_t)dim);
  return 0;
}

// sorted_dataset of ode_capi_aux.hip minus the uploads: the host's plan, the purity check and the gather, kernel bodies on the host.  -> false: refused
static bool sort_and_trim(std::vector<double>& X, std::vector<std::vector<double>*> Ys, int& n, int64_t M, DatasetPlan& pl, const std::vector<double>& callerX) {
  std::string why;
  if (dataset_plan(callerX.data(), n, pl, why) != 0) { std::printf("error %s\n", why.c_str()); return false; }
  if (pl.identity) return true;
  const unsigned gx = (unsigned)((M + kBlock - 1) / kBlock);
  if (!pl.dupKeep.empty()) {
    unsigned int flag = 0;
    for (auto* Y : Ys)
      hipemu::launch(dup_rows_differ_kernel, dim3(gx, (unsigned)pl.dupKeep.size()), dim3(kBlock), (const int32_t*)pl.dupKeep.data(), (const int32_t*)pl.dupDrop.data(),
                     (int)pl.dupKeep.size(), (const double*)Y->data(), M, &flag);
    if (flag) { std::printf("error impure y-duplicates\n"); return false; }
  }
  for (auto* Y : Ys) {
    std::vector<double> g(pl.src.size() * (size_t)M, -7.0);
    hipemu::launch(gather_rows_kernel, dim3(gx, (unsigned)pl.src.size()), dim3(kBlock), (const int32_t*)pl.src.data(), (int)pl.src.size(), (const double*)Y->data(), g.data(), M);
    Y->swap(g);
  }
  X = pl.x;
  n = (int)pl.x.size();
  return true;
}

static int do_discrete(const std::string& what, std::istringstream& in) {
  int n;
  int64_t M;
  in >> n >> M;
  std::vector<double> X, Y;
  rdv(in, X, (size_t)n);
  rdv(in, Y, (size_t)n * (size_t)M);
  const std::vector<double> callerX = X;   // X in the CALLER's order: sortAndTrimDataset runs here, as in the entries
  const int nCaller = n;
  DatasetPlan pl;
  if (!sort_and_trim(X, {&Y}, n, M, pl, callerX)) return 0;
  if ((what == "simpson" && n < 3) || (what == "slopes" && n < 2)) { std::printf("error too few distinct abscissae\n"); return 0; }
  std::vector<double> out((size_t)n * (size_t)M, -7.0);
  const dim3 grid((unsigned)((M + kBlock - 1) / kBlock)), block(kBlock);
  if (what == "trapz") {  // nnhip_cumtrapz_batch_f64_dev
    int first = 0;
    do {
      TrapzWeights W;
      const int nw = trapz_weights_fill(X.data(), n, first, W);
      hipemu::launch(cumtrapz_kernel, grid, block, W, nw, first, (const double*)Y.data(), out.data(), M);
      first += nw;
    } while (first < n - 1);
  } else if (what == "simpson") {  // nnhip_cumsimpson_batch_f64_dev
    std::vector<SimpsonPair> pairs;
    std::vector<SimpsonPoint> pts;
    int64_t nPairs = 0;
    bool evenN = false;
    simpson_tables(X.data(), n, pairs, pts, nPairs, evenN);
    hipemu::launch(cumsimpson_kernel, grid, block, (const SimpsonPair*)pairs.data(), (int)nPairs, evenN ? 1 : 0, (const SimpsonPoint*)pts.data(), (const double*)Y.data(),
                   out.data(), M, n);
    if (!pl.identity) {  // back to the caller's abscissae (hermiteInterpolate, integrate.nim:375)
      std::vector<int32_t> rows;
      simpson_result_rows(pl, callerX.data(), nCaller, rows);
      std::vector<double> back(rows.size() * (size_t)M, -7.0);
      hipemu::launch(gather_rows_kernel, dim3(grid.x, (unsigned)rows.size()), block, (const int32_t*)rows.data(), (int)rows.size(), (const double*)out.data(), back.data(), M);
      out.swap(back);
      n = (int)rows.size();
    }
  } else {  // nnhip_hermite_spline_slopes_f64_dev
    hipemu::launch(hermite_slopes_kernel, dim3(grid.x, (unsigned)n), block, (const double*)X.data(), n, (const double*)Y.data(), M, out.data());
  }
  print_rows(out, (size_t)n, (size_t)M);
  return 0;
}

static int do_eval(std::istringstream& in) {
  int n, n_q, deriv, extrap;
  int64_t M;
  in >> n >> M >> n_q >> deriv >> extrap;
  const double val = rd(in);
  std::vector<double> X, Y, dY, xq;
  rdv(in, X, (size_t)n); rdv(in, Y, (size_t)n * (size_t)M); rdv(in, dY, (size_t)n * (size_t)M); rdv(in, xq, (size_t)n_q);
  const std::vector<double> callerX = X;   // knots in the CALLER's order: the constructor's sortAndTrimDataset(@X, @[@Y, @dY]) (interpolate.nim:231) runs here, as in the entry
  DatasetPlan pl;
  if (!sort_and_trim(X, {&Y, &dY}, n, M, pl, callerX)) return 0;
  if (n < 2) { std::printf("error too few distinct abscissae\n"); return 0; }
  std::vector<double> out((size_t)n_q * (size_t)M, -7.0);
  for (int q0 = 0; q0 < n_q; q0 += kHermChunk) {  // nnhip_hermite_spline_eval_batch_f64_dev
    HermChunk c;
    const int nq = std::min(kHermChunk, n_q - q0);
    herm_chunk_fill(X.data(), n, xq.data() + q0, nq, deriv != 0, extrap, val, c);
    hipemu::launch(hermite_interp_kernel, dim3((unsigned)((M + kBlock - 1) / kBlock), (unsigned)nq), dim3(kBlock), c, nq, (const double*)Y.data(), (const double*)dY.data(), M,
                   out.data() + (int64_t)q0 * M);
  }
  print_rows(out, (size_t)n_q, (size_t)M);
  return 0;
}

int main() {
  std::string line;
  while (std::getline(std::cin, line)) {
    if (line.empty()) continue;
    std::istringstream in(line);
    std::string what;
    in >> what;
    int rc;
    if (what == "fn") rc = do_fn(in);
    else if (what == "trapz" || what == "simpson" || what == "slopes") rc = do_discrete(what, in);
    else if (what == "eval") rc = do_eval(in);
    else return 2;
    if (rc) return rc;
    std::printf("end\n");
  }
  return 0;
}
