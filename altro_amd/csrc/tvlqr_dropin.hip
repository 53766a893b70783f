// tvlqr_dropin.hip -- the reference's three tvlqr_* entry points (include/tvlqr/tvlqr.h) on the GPU.
// Pure plumbing around kernels/tvlqr_generic.hip: gather host blocks -> one H2D copy -> one launch ->
// one D2H copy -> scatter.  One problem = one wavefront; this path exists for drop-in fidelity
// (callers that hold a single problem as arrays of per-knot-point pointers), not for throughput --
// batches go through include/altro_hip/altro_hip.h.
#include "tvlqr/tvlqr.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>

#include <cstring>
#include <vector>

#include "kernels/tvlqr_generic.hip"
#include "kernels/tvlqr_lane.hip"   // plan LANE's sweeps: a single small problem rides one lane (four for (4, 2) and (2, 1))
#include "capi_tile32.h"            // plan MFMA32's backward kernel: a single larger problem rides the matrix cores

using namespace altro_hip;

int tvlqr_TotalMemSize(const int* nx, const int* nu, int num_horizon, bool is_diag) {
  // src/tvlqr/tvlqr.cpp:18-63: bytes of a flat buffer holding every per-knot-point block
  if (!nx) return 0;
  if (!nu) return 0;
  long long count = 0;
  for (int k = 0; k <= num_horizon; ++k) {
    const long long n = nx[k];
    count += (is_diag ? n : n * n) + n;   // Q q
    count += n * n + n;                   // P p
    count += n + n;                       // x y
    if (k < num_horizon) {
      const long long m = nu[k];
      count += n * n + n * m + n;                              // A B f
      count += (is_diag ? m : m * m) + (is_diag ? 0 : m * n) + m;   // R H r
      count += m * n + m;                                      // K d
      count += 2 * (n * n + m * m + m * n + n + m);            // Q-blocks and their scratch twins
      count += m;                                              // u
    }
  }
  count += 2;   // delta_V
  return (int)(count * (long long)sizeof(lqr_float));
}

namespace {

// Cached per thread: no allocation after the first call of a given size, and -- a solver calls these entries with the same
// dimensions iteration after iteration -- no table upload after the first call of a given layout.  One call is ONE
// asynchronous host-to-device copy of the INPUT blocks (the arena's prefix, from pinned memory), one single-wave kernel
// that writes its OUTPUT blocks, the status word and delta_V straight into the pinned arena (mapped into the device's
// address space: posted PCIe writes, nobody waits on them) and one wait on a private stream -- round 2 copied the whole
// arena both ways with two blocking copies.
struct Workspace {
  double* dev = nullptr;
  double* host_dev = nullptr;     // the pinned arena as the device sees it
  hipStream_t stream = nullptr;
  size_t dev_elems = 0;
  int64_t* dev_off = nullptr;
  int* dev_dims = nullptr;
  size_t table_k = 0;
  double* big = nullptr;          // the backward kernel's work block when a knot point does not fit 64 KB of LDS
  size_t big_bytes = 0;
  double* host = nullptr;         // pinned staging arena (hipHostMalloc), same size as dev
  std::vector<int64_t> off;       // the tables the device holds right now
  std::vector<int> dims;
  ~Workspace() {
    int c = 0;   // thread / process teardown: the HIP runtime may already be gone -- then there is nothing left to free
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) return;
    if (dev) (void)hipFree(dev);
    if (big) (void)hipFree(big);
    if (dev_off) (void)hipFree(dev_off);
    if (dev_dims) (void)hipFree(dev_dims);
    if (host) (void)hipHostFree(host);
    if (stream) (void)hipStreamDestroy(stream);
  }
};
thread_local Workspace g_ws;

// ALTRO_TVLQR_DROPIN_STATS=1: where a tvlqr_BackwardPass call spends its time -- layout + workspace, staging + launches + the wait,
// the scatter into the caller's blocks -- summed over the process and printed at exit (tests/test_gpu_scotty.py shows it).
struct SeamStats {
  bool on = std::getenv("ALTRO_TVLQR_DROPIN_STATS") != nullptr;
  double us[3] = {0, 0, 0};
  long calls = 0;
  ~SeamStats() {
    if (on && calls)
      std::fprintf(stderr, "tvlqr_BackwardPass seam: %ld calls, per call %.1f us layout + workspace, %.1f us staging + launches + wait, "
                           "%.1f us scatter\n", calls, us[0] / calls, us[1] / calls, us[2] / calls);
  }
};
SeamStats g_seam_stats;
struct SeamClock {
  std::chrono::steady_clock::time_point t;
  SeamClock() { if (g_seam_stats.on) t = std::chrono::steady_clock::now(); }
  void lap(int i) {
    if (!g_seam_stats.on) return;
    const auto now = std::chrono::steady_clock::now();
    g_seam_stats.us[i] += std::chrono::duration<double, std::micro>(now - t).count();
    t = now;
    if (i == 2) ++g_seam_stats.calls;
  }
};

bool have_device() {
  int c = 0;
  return hipGetDeviceCount(&c) == hipSuccess && c > 0;
}

// any dimensions, like the reference (tvlqr.cpp:92-121): blocks that do not fit 64 KB of LDS are worked on in global memory
// (generic_backward_kernel<T, true>); kGenericMaxDim only bounds that work block
bool dims_supported(const int* nx, const int* nu, int N) {
  for (int k = 0; k <= N; ++k)
    if (nx[k] < 0 || nx[k] > kGenericMaxDim || (k < N && (nu[k] < 0 || nu[k] > kGenericMaxDim))) return false;
  return true;
}

// Lays every array's knot-point blocks out in one arena; returns the arena size in elements.
struct Layout {
  int N;
  std::vector<int64_t> off;   // [(N+1) * G_NUM]
  int64_t total;
  int64_t group_end[4];       // arena offset where each group of arrays ends
  int nmax, mmax;
  int xstage;                 // elements reserved for the forward pass's x0 in front of delta_V at the arena's end
};

int64_t block_size(int arr, int n, int m, int n2, bool is_diag) {
  switch (arr) {
    case G_A: return (int64_t)n2 * n;
    case G_B: return (int64_t)n2 * m;
    case G_f: return n2;
    case G_Q: return is_diag ? n : (int64_t)n * n;
    case G_R: return is_diag ? m : (int64_t)m * m;
    case G_H: case G_K: case G_Qux: case G_Qux_tmp: return (int64_t)m * n;
    case G_q: case G_p: case G_Qx: case G_Qx_tmp: case G_x: case G_y: return n;
    case G_r: case G_d: case G_Qu: case G_Qu_tmp: case G_u: return m;
    case G_P: case G_Qxx: case G_Qxx_tmp: return (int64_t)n * n;
    case G_Quu: case G_Quu_tmp: return (int64_t)m * m;
    default: return 0;
  }
}

// Arena order: [A B f of every k] [Q R H q r of every k] [K d P p of every k] [everything else] [x0 | delta_V | status].
// The backward pass's inputs are the first two groups, the forward pass's the first three: one contiguous copy each.
Layout build_layout(const int* nx, const int* nu, int N, bool is_diag) {
  Layout L;
  L.N = N;
  L.off.assign((size_t)(N + 1) * G_NUM, 0);
  L.nmax = 0; L.mmax = 0;
  int64_t cur = 0;
  auto group_of = [](int a) { return (a == G_A || a == G_B || a == G_f) ? 0 : (a == G_Q || a == G_R || a == G_H || a == G_q || a == G_r) ? 1
                                     : (a == G_K || a == G_d || a == G_P || a == G_p) ? 2 : 3; };
  for (int g = 0; g < 4; ++g) {
    for (int k = 0; k <= N; ++k) {
      const int n = nx[k], m = (k < N) ? nu[k] : 0, n2 = (k < N) ? nx[k + 1] : 0;
      if (n > L.nmax) L.nmax = n;
      if (m > L.mmax) L.mmax = m;
      for (int a = 0; a < G_NUM; ++a) {
        if (group_of(a) != g) continue;
        const bool has_terminal = (a == G_Q || a == G_q || a == G_P || a == G_p || a == G_x || a == G_y);
        L.off[(size_t)k * G_NUM + a] = cur;
        if (k < N || has_terminal) cur += (block_size(a, n, m, n2, is_diag) + 1) & ~int64_t(1);   // keep 16-B alignment
      }
    }
    L.group_end[g] = cur;
  }
  L.xstage = std::max(32, (L.nmax + 3) & ~3);
  L.total = cur + L.xstage + 4;   // + x0 staging + delta_V[2] (+pad) at the end
  return L;
}

// A solver calls the seam with the same dimensions sweep after sweep: the layout of the last call is kept per thread and found
// again by comparing the dimension arrays (N = 30, (4, 2): building it took 48 of a call's 140 us -- the clock of
// ALTRO_TVLQR_DROPIN_STATS through tests/cpp/bicycle_mpc_test.cpp).
struct LayoutCache {
  Layout L;
  std::vector<int> nx, nu;
  bool is_diag = false, valid = false;
};
thread_local LayoutCache g_layout[2];   // [0]: the backward pass's, [1]: the forward pass's (always dense)
const Layout& make_layout(const int* nx, const int* nu, int N, bool is_diag, int slot = 0) {
  LayoutCache& c = g_layout[slot];
  if (c.valid && c.L.N == N && c.is_diag == is_diag && std::memcmp(c.nx.data(), nx, sizeof(int) * (size_t)(N + 1)) == 0 &&
      (N == 0 || std::memcmp(c.nu.data(), nu, sizeof(int) * (size_t)N) == 0))
    return c.L;
  c.valid = false;
  c.L = build_layout(nx, nu, N, is_diag);
  c.nx.assign(nx, nx + N + 1);
  c.nu.assign(nu, nu + N);
  c.is_diag = is_diag;
  c.valid = true;
  return c.L;
}

int prepare(Workspace& w, const Layout& L, const int* nx, const int* nu) {
  const int N = L.N;
  if (w.dev_elems < (size_t)L.total) {
    if (w.dev) (void)hipFree(w.dev);
    w.dev = nullptr;
    w.dev_elems = 0;
    if (w.host) (void)hipHostFree(w.host);
    w.host = nullptr;
    if (hipMalloc(&w.dev, (size_t)L.total * sizeof(double)) != hipSuccess) { w.dev = nullptr; return -1; }
    if (hipHostMalloc((void**)&w.host, (size_t)L.total * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void**)&w.host_dev, w.host, 0) != hipSuccess) {
      (void)hipFree(w.dev); w.dev = nullptr;
      if (w.host) (void)hipHostFree(w.host);
      w.host = nullptr; return -1;
    }
    w.dev_elems = (size_t)L.total;
  }
  if (!w.stream && hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking) != hipSuccess) { w.stream = nullptr; return -1; }
  if (w.table_k < (size_t)(N + 1)) {
    if (w.dev_off) (void)hipFree(w.dev_off);
    if (w.dev_dims) (void)hipFree(w.dev_dims);
    w.dev_off = nullptr; w.dev_dims = nullptr; w.table_k = 0;   // nothing dangling if an allocation below fails
    if (hipMalloc(&w.dev_off, (size_t)(N + 1) * G_NUM * sizeof(int64_t)) != hipSuccess) { w.dev_off = nullptr; return -1; }
    if (hipMalloc(&w.dev_dims, (size_t)(N + 1) * 2 * sizeof(int)) != hipSuccess) { w.dev_dims = nullptr; return -1; }
    w.table_k = (size_t)(N + 1);
    w.off.clear(); w.dims.clear();
  }
  bool same = w.dims.size() == (size_t)(N + 1) * 2 && w.off.size() == L.off.size();
  for (int k = 0; k <= N && same; ++k) same = w.dims[k] == nx[k] && w.dims[(size_t)(N + 1) + k] == ((k < N) ? nu[k] : 0);
  if (same) same = std::memcmp(w.off.data(), L.off.data(), L.off.size() * sizeof(int64_t)) == 0;
  if (!same) {
    std::vector<int> dims((size_t)(N + 1) * 2, 0);
    for (int k = 0; k <= N; ++k) {
      dims[k] = nx[k];
      dims[(size_t)(N + 1) + k] = (k < N) ? nu[k] : 0;
    }   // a new layout: upload its tables (the usual call finds them in place)
    w.off.clear(); w.dims.clear();
    if (hipMemcpy(w.dev_off, L.off.data(), L.off.size() * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess) return -1;
    if (hipMemcpy(w.dev_dims, dims.data(), dims.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return -1;
    w.off = L.off;
    w.dims = dims;
  }
  return 0;
}

// `inputs_end`: arrays laid out below this arena offset are read from the device copy, everything else -- outputs, scratch
// twins, delta_V, the status word, the forward pass's x0 -- lives in the pinned arena itself
GenericArgs<double> make_args(Workspace& w, const Layout& L, double reg, bool is_diag, int want_y, int64_t inputs_end) {
  GenericArgs<double> a;
  for (int i = 0; i < G_NUM; ++i) { a.base[i] = (L.off[i] < inputs_end) ? w.dev : w.host_dev; a.bstride[i] = 0; }
  a.off = w.dev_off;
  a.nx = w.dev_dims;
  a.nu = w.dev_dims + (L.N + 1);
  a.x0 = w.host_dev + L.total - 4 - L.xstage;
  a.x0_stride = 0;
  a.delta_V = w.host_dev + L.total - 4;
  a.status = reinterpret_cast<int*>(w.host_dev + L.total - 2);   // the arena's last two elements are padding: the status word lives there
  a.N = L.N; a.batch = 1; a.nmax = L.nmax; a.mmax = L.mmax > 0 ? L.mmax : 1;
  a.reg = reg; a.is_diag = is_diag ? 1 : 0; a.store_q = 2; a.want_y = want_y;
  return a;
}


// ---- fast path for ONE small problem with uniform dimensions (n <= 6, m <= 3) ----------------------------------------------------
// The single-wave GENERIC kernel stages every block through LDS and takes 7-10 us per knot point; the lane-per-problem
// sweeps of the batched ABI keep a knot point in registers: 3 us per knot point in one lane, 1.6 us with four lanes per
// problem.  So: pack the inputs as a batch of ONE in plan LANE's record layout, run its backward sweep (K, d, P, p, delta_V,
// status into device memory), then ONE more launch -- a thread per knot point, nothing serial -- recomputes what the seam's
// signature carries besides (Qxx, Quu, Qux, Qx, Qu and the final contents of their scratch twins, tvlqr.cpp:125-191) from the
// inputs and the sweep's results with the sweep's own expressions, and writes everything into the mapped pinned arena at the
// GENERIC layout's offsets, where the scatter code below picks it up.  A failed factorisation (rare) repeats the call on the
// GENERIC path, whose partial outputs follow the reference's early return to the letter.
ALTRO_FP_REGION_OFF
template <int n, int m>
__global__ void dropin_qblocks_kernel(const double* __restrict__ in, const double* __restrict__ term, const double* __restrict__ out,
                                      const double* __restrict__ outn, const double* __restrict__ dV, const int* __restrict__ status,
                                      int N, double reg, double* __restrict__ host, const int64_t* __restrict__ off, int64_t total) {
  using D = LaneDims<n, m>;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > N) return;
  auto at = [&](int arr) -> double* { return host + off[(size_t)k * G_NUM + arr]; };
  if (k == N) {
    double* Ph = at(G_P); double* ph = at(G_p);
    for (int e = 0; e < n * n; ++e) Ph[e] = outn[e];
    for (int e = 0; e < n; ++e) ph[e] = outn[n * n + e];
    host[total - 4] = dV[0]; host[total - 3] = dV[1];
    *reinterpret_cast<int*>(host + total - 2) = status[0];
    (void)term;
    return;
  }
  const double* rec = in + (size_t)k * D::E_IN;
  const double* o = out + (size_t)k * D::E_OUT;
  const double* Pn = (k + 1 < N) ? out + (size_t)(k + 1) * D::E_OUT + D::O_P : outn;
  const double* pn = (k + 1 < N) ? out + (size_t)(k + 1) * D::E_OUT + D::O_p : outn + n * n;
  double A[n * n], Bm[n * m], f[n], Qxx[n * n], Quu[m * m], Qux[m * n], Qx[n], Qu[m], P[n * n], pp[n], K[m * n], d[m];
  for (int e = 0; e < n * n; ++e) { A[e] = rec[D::O_A + e]; Qxx[e] = rec[D::O_Q + e]; P[e] = Pn[e]; }
  for (int e = 0; e < n * m; ++e) { Bm[e] = rec[D::O_B + e]; Qux[e] = rec[D::O_H + e]; K[e] = o[D::O_K + e]; }
  for (int e = 0; e < n; ++e) { f[e] = rec[D::O_f + e]; Qx[e] = rec[D::O_q + e]; pp[e] = pn[e]; }
  for (int e = 0; e < m * m; ++e) Quu[e] = rec[D::O_R + e];
  for (int e = 0; e < m; ++e) { Qu[e] = rec[D::O_r + e]; d[e] = o[D::O_d + e]; }
  // the expressions of lane_backward_step (kernels/tvlqr_lane_body.inc), index-ordered dot products, no contraction
  double T1[n * n], T2[m * n], t[n];
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) { double s = 0.0; for (int kk = 0; kk < n; ++kk) s += A[kk + i * n] * P[kk + j * n]; T1[i + j * n] = 0.0 + s; }
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < m; ++i) { double s = 0.0; for (int kk = 0; kk < n; ++kk) s += Bm[kk + i * n] * P[kk + j * n]; T2[i + j * m] = 0.0 + s; }
  for (int i = 0; i < n; ++i) { double s = 0.0; for (int kk = 0; kk < n; ++kk) s += P[i + kk * n] * f[kk]; t[i] = pp[i] + s; }
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) { double s = 0.0; for (int kk = 0; kk < n; ++kk) s += T1[i + kk * n] * A[kk + j * n]; Qxx[i + j * n] = Qxx[i + j * n] + s; }
  for (int j = 0; j < m; ++j)
    for (int i = 0; i < m; ++i) { double s = 0.0; for (int kk = 0; kk < n; ++kk) s += T2[i + kk * m] * Bm[kk + j * n]; Quu[i + j * m] = Quu[i + j * m] + s; }
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < m; ++i) { double s = 0.0; for (int kk = 0; kk < n; ++kk) s += T2[i + kk * m] * A[kk + j * n]; Qux[i + j * m] = Qux[i + j * m] + s; }
  for (int i = 0; i < n; ++i) { double s = 0.0; for (int kk = 0; kk < n; ++kk) s += A[kk + i * n] * t[kk]; Qx[i] = Qx[i] + s; }
  for (int i = 0; i < m; ++i) { double s = 0.0; for (int kk = 0; kk < n; ++kk) s += Bm[kk + i * n] * t[kk]; Qu[i] = Qu[i] + s; }
  // the scratch twins as the reference leaves them: chol(Quu + reg I) in place, Quu K, K^T Qux, K^T Qu, Quu d
  double L[m * m], QuuK[m * n], KtQux[n * n], KtQu[n], Quud[m];
  for (int e = 0; e < m * m; ++e) L[e] = Quu[e] + ((e % m == e / m) ? reg : 0.0);
  for (int kk = 0; kk < m; ++kk) {
    double x = L[kk + kk * m];
    for (int j = 0; j < kk; ++j) x -= L[kk + j * m] * L[kk + j * m];
    x = sqrt(x);
    L[kk + kk * m] = x;
    for (int i = kk + 1; i < m; ++i) {
      double s = L[i + kk * m];
      for (int j = 0; j < kk; ++j) s -= L[i + j * m] * L[kk + j * m];
      L[i + kk * m] = s / x;
    }
  }
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < m; ++i) { double s = 0.0; for (int kk = 0; kk < m; ++kk) s += Quu[i + kk * m] * K[kk + j * m]; QuuK[i + j * m] = 0.0 + s; }
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) { double s = 0.0; for (int kk = 0; kk < m; ++kk) s += K[kk + i * m] * Qux[kk + j * m]; KtQux[i + j * n] = 0.0 + s; }
  for (int i = 0; i < n; ++i) { double s = 0.0; for (int kk = 0; kk < m; ++kk) s += K[kk + i * m] * Qu[kk]; KtQu[i] = 0.0 + s; }
  for (int i = 0; i < m; ++i) { double s = 0.0; for (int kk = 0; kk < m; ++kk) s += Quu[i + kk * m] * d[kk]; Quud[i] = 0.0 + s; }
  double* h;
  h = at(G_K); for (int e = 0; e < m * n; ++e) h[e] = K[e];
  h = at(G_d); for (int e = 0; e < m; ++e) h[e] = d[e];
  h = at(G_P); for (int e = 0; e < n * n; ++e) h[e] = o[D::O_P + e];
  h = at(G_p); for (int e = 0; e < n; ++e) h[e] = o[D::O_p + e];
  h = at(G_Qxx); for (int e = 0; e < n * n; ++e) h[e] = Qxx[e];
  h = at(G_Quu); for (int e = 0; e < m * m; ++e) h[e] = Quu[e];
  h = at(G_Qux); for (int e = 0; e < m * n; ++e) h[e] = Qux[e];
  h = at(G_Qx); for (int e = 0; e < n; ++e) h[e] = Qx[e];
  h = at(G_Qu); for (int e = 0; e < m; ++e) h[e] = Qu[e];
  h = at(G_Qxx_tmp); for (int e = 0; e < n * n; ++e) h[e] = KtQux[e];
  h = at(G_Quu_tmp); for (int e = 0; e < m * m; ++e) h[e] = L[e];
  h = at(G_Qux_tmp); for (int e = 0; e < m * n; ++e) h[e] = QuuK[e];
  h = at(G_Qx_tmp); for (int e = 0; e < n; ++e) h[e] = KtQu[e];
  h = at(G_Qu_tmp); for (int e = 0; e < m; ++e) h[e] = Quud[e];
}
ALTRO_FP_REGION_END

#define DROPIN_SHAPES(X)                                                                                 \
  X(1, 1) X(2, 1) X(3, 1) X(4, 1) X(5, 1) X(6, 1) X(1, 2) X(2, 2) X(3, 2) X(4, 2) X(5, 2) X(6, 2) X(1, 3) \
  X(2, 3) X(3, 3) X(4, 3) X(5, 3) X(6, 3)

struct FastWs {           // device + pinned staging of the batch-of-one LANE records (per thread, grown on demand)
  double* dev = nullptr;
  double* host = nullptr;
  size_t elems = 0;
  ~FastWs() {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) return;
    if (dev) (void)hipFree(dev);
    if (host) (void)hipHostFree(host);
  }
};
thread_local FastWs g_fast;

bool fast_shape(const int* nx, const int* nu, int N, int* n_out, int* m_out) {
  if (N < 1) return false;
  const int n = nx[0], m = nu[0];
  if (n < 1 || n > 6 || m < 1 || m > 3) return false;
  for (int k = 0; k <= N; ++k)
    if (nx[k] != n || (k < N && nu[k] != m)) return false;
  *n_out = n; *m_out = m;
  return true;
}

// returns 0 when the fast path ran to completion (outputs and status in the pinned arena w.host), 1 to fall back
int fast_backward(Workspace& w, const Layout& L, int n, int m, int N, const double* const* A, const double* const* B,
                  const double* const* f, const double* const* Q, const double* const* R, const double* const* H,
                  const double* const* q, const double* const* r, double reg, bool is_diag) {
  if (std::getenv("ALTRO_TVLQR_DROPIN_GENERIC") != nullptr) return 1;   // A/B hook: always the GENERIC kernel
  const int e_in = 2 * n * n + 2 * n * m + m * m + 2 * n + m, e_term = n * n + n, e_out = m * n + m + n * n + n;
  const size_t in_elems = (size_t)N * e_in + e_term;
  const size_t total = in_elems + (size_t)N * e_out + e_term + 4;
  FastWs& fw = g_fast;
  if (fw.elems < total) {
    if (fw.dev) (void)hipFree(fw.dev);
    if (fw.host) (void)hipHostFree(fw.host);
    fw.dev = nullptr; fw.host = nullptr; fw.elems = 0;
    if (hipMalloc(&fw.dev, total * sizeof(double)) != hipSuccess) { fw.dev = nullptr; return 1; }
    if (hipHostMalloc((void**)&fw.host, in_elems * sizeof(double) + 64, hipHostMallocDefault) != hipSuccess) {
      (void)hipFree(fw.dev); fw.dev = nullptr; fw.host = nullptr; return 1;
    }
    fw.elems = total;
  }
  // plan LANE's record of one knot point: A | B | f | Q | R | H | q | r (kernels/tvlqr_lane.hip, LaneDims)
  const int oB = n * n, of = oB + n * m, oQ = of + n, oR = oQ + n * n, oH = oR + m * m, oq = oH + m * n, orr = oq + n;
  auto put = [](double* dst, const double* src, int cnt) {
    if (src) memcpy(dst, src, sizeof(double) * cnt);
    else memset(dst, 0, sizeof(double) * cnt);
  };
  auto put_sym = [&](double* dst, const double* src, int dim) {   // a diagonal block (is_diag) expanded to the dense one
    if (!is_diag) { put(dst, src, dim * dim); return; }
    memset(dst, 0, sizeof(double) * dim * dim);
    for (int i = 0; i < dim; ++i) dst[i + i * dim] = src[i];
  };
  for (int k = 0; k < N; ++k) {
    double* rec = fw.host + (size_t)k * e_in;
    put(rec, A[k], n * n); put(rec + oB, B[k], n * m); put(rec + of, f[k], n);
    put_sym(rec + oQ, Q[k], n); put_sym(rec + oR, R[k], m);
    put(rec + oH, is_diag ? nullptr : H[k], m * n);
    put(rec + oq, q[k], n); put(rec + orr, r[k], m);
  }
  {
    double* tr = fw.host + (size_t)N * e_in;
    put_sym(tr, Q[N], n); put(tr + n * n, q[N], n);
  }
  double* d_in = fw.dev;
  double* d_term = d_in + (size_t)N * e_in;
  double* d_out = d_term + e_term;
  double* d_outn = d_out + (size_t)N * e_out;
  double* d_dv = d_outn + e_term;
  int* d_status = reinterpret_cast<int*>(d_dv + 2);
  // (Measured and not kept: the kernel reading its records from the pinned staging itself, no copy enqueued -- the single wave then
  //  waits on the bus at every knot point: 85 -> 101-107 us per call at N = 30, (4, 2); profiles/r06g_scotty_seam.txt.)
  if (hipMemcpyAsync(d_in, fw.host, in_elems * sizeof(double), hipMemcpyHostToDevice, w.stream) != hipSuccess) return 1;
  LaneArgs<double> a{d_in, d_term, d_out, d_outn, nullptr, nullptr, d_dv, d_status, N, 1, reg, nullptr, nullptr};
  bool done = false;
  if (n == 4 && m == 2) { hipLaunchKernelGGL((quad_backward_kernel<2, double>), dim3(8), dim3(64), 0, w.stream, a); done = true; }
  else if (n == 2 && m == 1) { hipLaunchKernelGGL((quad2_backward_kernel<double>), dim3(8), dim3(64), 0, w.stream, a); done = true; }
#define X(N_, M_)                                                                                                     \
  if (!done && n == N_ && m == M_) { hipLaunchKernelGGL((lane_backward_kernel<N_, M_, double>), dim3(8), dim3(64), 0, w.stream, a); done = true; }
  DROPIN_SHAPES(X)
#undef X
  if (!done || hipGetLastError() != hipSuccess) return 1;
  const dim3 qgrid((N + 1 + 63) / 64), qblock(64);
  done = false;
#define X(N_, M_)                                                                                                     \
  if (!done && n == N_ && m == M_) {                                                                                  \
    hipLaunchKernelGGL((dropin_qblocks_kernel<N_, M_>), qgrid, qblock, 0, w.stream, (const double*)d_in, (const double*)d_term,     \
                       (const double*)d_out, (const double*)d_outn, (const double*)d_dv, (const int*)d_status, N, reg, w.host_dev,  \
                       (const int64_t*)w.dev_off, (int64_t)L.total);                                                  \
    done = true;                                                                                                      \
  }
  DROPIN_SHAPES(X)
#undef X
  if (!done || hipGetLastError() != hipSuccess) return 1;
  if (hipStreamSynchronize(w.stream) != hipSuccess) return 1;
  int status;
  memcpy(&status, w.host + L.total - 2, sizeof(int));
  return status == TVLQR_SUCCESS ? 0 : 1;   // a failed factorisation: the GENERIC path reproduces the reference's early return
}

// ---- fast path for ONE problem with uniform dimensions on the matrix cores (7 <= n <= 31, m <= 8, n + m <= 32) ---------------------
// The single-wave GENERIC kernel takes 14 us per knot point at (12, 4) and 40 at (20, 8); plan MFMA32's backward kernel
// (kernels/tvlqr_tile32.hip: 2 x 2 tiles of v_mfma_f64_16x16x4, one wave per problem) takes 2 - 4.  So: the inputs staged as a batch
// of ONE in plan GENERIC's array layout ([k][block]), that kernel (K, d, P, p, delta_V, status into device memory), then ONE more
// launch -- a wave per knot point, nothing serial -- forms what the seam's signature carries besides (Qxx, Quu, Qux, Qx, Qu and
// the final contents of their scratch twins, tvlqr.cpp:125-191) from the inputs and the sweep's P, p, K, d in the index order of the
// GENERIC kernel, and writes everything into the mapped pinned arena at the GENERIC layout's offsets.  The sweep's sums run in the
// matrix cores' order, not the oracle's: K, d, P, p agree with the GENERIC path to ~1e-13 relative (tests/cpp/tvlqr_dropin_tile_test.cpp
// holds 1e-9), not bit for bit as the two other paths do.  A failed factorisation repeats the call on the GENERIC path.
ALTRO_FP_REGION_OFF
__global__ __launch_bounds__(64) void dropin_qblocks_wave_kernel(Tile32Args a, double reg, double* __restrict__ host,
                                                                const int64_t* __restrict__ off, int64_t total) {
  extern __shared__ double sm[];
  const int n = a.n, m = a.m, N = a.N, k = (int)blockIdx.x, t = (int)threadIdx.x;
  auto at = [&](int arr) -> double* { return host + off[(size_t)k * G_NUM + arr]; };
  if (k == N) {
    double* Ph = at(G_P); double* ph = at(G_p);
    for (int e = t; e < n * n; e += 64) Ph[e] = a.P[(size_t)N * n * n + e];
    for (int e = t; e < n; e += 64) ph[e] = a.p[(size_t)N * n + e];
    if (t == 0) {
      host[total - 4] = a.delta_V[0]; host[total - 3] = a.delta_V[1];
      *reinterpret_cast<int*>(host + total - 2) = a.status[0];
    }
    return;
  }
  double* A = sm;               // n x n
  double* Bm = A + n * n;       // n x m
  double* Pn = Bm + n * m;      // n x n   P_{k+1}
  double* T1 = Pn + n * n;      // n x n   A^T P'
  double* T2 = T1 + n * n;      // m x n   B^T P'
  double* Qxx = T2 + m * n;     // n x n
  double* Qux = Qxx + n * n;    // m x n
  double* Quu = Qux + m * n;    // m x m
  double* Kk = Quu + m * m;     // m x n
  double* L = Kk + m * n;       // m x m
  double* f = L + m * m;        // n
  double* tt = f + n;           // n       p' + P' f
  double* Qx = tt + n;          // n
  double* Qu = Qx + n;          // m
  double* dk = Qu + m;          // m
  for (int e = t; e < n * n; e += 64) {
    A[e] = a.A[(size_t)k * n * n + e];
    Pn[e] = a.P[(size_t)(k + 1) * n * n + e];
    Qxx[e] = a.Q[(size_t)k * n * n + e];
  }
  for (int e = t; e < n * m; e += 64) {
    Bm[e] = a.B[(size_t)k * n * m + e];
    Qux[e] = a.H[(size_t)k * n * m + e];
    Kk[e] = a.K[(size_t)k * n * m + e];
  }
  for (int e = t; e < m * m; e += 64) Quu[e] = a.R[(size_t)k * m * m + e];
  for (int e = t; e < n; e += 64) { f[e] = a.f[(size_t)k * n + e]; Qx[e] = a.q[(size_t)k * n + e]; }
  for (int e = t; e < m; e += 64) { Qu[e] = a.r[(size_t)k * m + e]; dk[e] = a.d[(size_t)k * m + e]; }
  __syncthreads();
  // the expressions of the GENERIC kernel / lane_backward_step: index-ordered dot products, no contraction
  for (int e = t; e < n * n; e += 64) { const int i = e % n, j = e / n; double s = 0.0; for (int c = 0; c < n; ++c) s += A[c + i * n] * Pn[c + j * n]; T1[e] = 0.0 + s; }
  for (int e = t; e < m * n; e += 64) { const int i = e % m, j = e / m; double s = 0.0; for (int c = 0; c < n; ++c) s += Bm[c + i * n] * Pn[c + j * n]; T2[e] = 0.0 + s; }
  for (int i = t; i < n; i += 64) { double s = 0.0; for (int c = 0; c < n; ++c) s += Pn[i + c * n] * f[c]; tt[i] = a.p[(size_t)(k + 1) * n + i] + s; }
  __syncthreads();
  for (int e = t; e < n * n; e += 64) { const int i = e % n, j = e / n; double s = 0.0; for (int c = 0; c < n; ++c) s += T1[i + c * n] * A[c + j * n]; Qxx[e] = Qxx[e] + s; }
  for (int e = t; e < m * m; e += 64) { const int i = e % m, j = e / m; double s = 0.0; for (int c = 0; c < n; ++c) s += T2[i + c * m] * Bm[c + j * n]; Quu[e] = Quu[e] + s; }
  for (int e = t; e < m * n; e += 64) { const int i = e % m, j = e / m; double s = 0.0; for (int c = 0; c < n; ++c) s += T2[i + c * m] * A[c + j * n]; Qux[e] = Qux[e] + s; }
  for (int i = t; i < n; i += 64) { double s = 0.0; for (int c = 0; c < n; ++c) s += A[c + i * n] * tt[c]; Qx[i] = Qx[i] + s; }
  for (int i = t; i < m; i += 64) { double s = 0.0; for (int c = 0; c < n; ++c) s += Bm[c + i * n] * tt[c]; Qu[i] = Qu[i] + s; }
  __syncthreads();
  // the scratch twins as the reference leaves them: chol(Quu + reg I) in place, Quu K, K^T Qux, K^T Qu, Quu d
  for (int e = t; e < m * m; e += 64) L[e] = Quu[e] + ((e % m == e / m) ? reg : 0.0);
  __syncthreads();
  if (t == 0)
    for (int c = 0; c < m; ++c) {
      double x = L[c + c * m];
      for (int j = 0; j < c; ++j) x -= L[c + j * m] * L[c + j * m];
      x = sqrt(x);
      L[c + c * m] = x;
      for (int i = c + 1; i < m; ++i) {
        double s = L[i + c * m];
        for (int j = 0; j < c; ++j) s -= L[i + j * m] * L[c + j * m];
        L[i + c * m] = s / x;
      }
    }
  double* h;
  h = at(G_Qux_tmp); for (int e = t; e < m * n; e += 64) { const int i = e % m, j = e / m; double s = 0.0; for (int c = 0; c < m; ++c) s += Quu[i + c * m] * Kk[c + j * m]; h[e] = 0.0 + s; }
  h = at(G_Qxx_tmp); for (int e = t; e < n * n; e += 64) { const int i = e % n, j = e / n; double s = 0.0; for (int c = 0; c < m; ++c) s += Kk[c + i * m] * Qux[c + j * m]; h[e] = 0.0 + s; }
  h = at(G_Qx_tmp); for (int i = t; i < n; i += 64) { double s = 0.0; for (int c = 0; c < m; ++c) s += Kk[c + i * m] * Qu[c]; h[i] = 0.0 + s; }
  h = at(G_Qu_tmp); for (int i = t; i < m; i += 64) { double s = 0.0; for (int c = 0; c < m; ++c) s += Quu[i + c * m] * dk[c]; h[i] = 0.0 + s; }
  h = at(G_K); for (int e = t; e < m * n; e += 64) h[e] = Kk[e];
  h = at(G_d); for (int e = t; e < m; e += 64) h[e] = dk[e];
  h = at(G_P); for (int e = t; e < n * n; e += 64) h[e] = a.P[(size_t)k * n * n + e];
  h = at(G_p); for (int e = t; e < n; e += 64) h[e] = a.p[(size_t)k * n + e];
  h = at(G_Qxx); for (int e = t; e < n * n; e += 64) h[e] = Qxx[e];
  h = at(G_Quu); for (int e = t; e < m * m; e += 64) h[e] = Quu[e];
  h = at(G_Qux); for (int e = t; e < m * n; e += 64) h[e] = Qux[e];
  h = at(G_Qx); for (int e = t; e < n; e += 64) h[e] = Qx[e];
  h = at(G_Qu); for (int e = t; e < m; e += 64) h[e] = Qu[e];
  __syncthreads();
  h = at(G_Quu_tmp); for (int e = t; e < m * m; e += 64) h[e] = L[e];
}
ALTRO_FP_REGION_END

bool tile_shape(const int* nx, const int* nu, int N, int* n_out, int* m_out) {
  if (N < 1) return false;
  const int n = nx[0], m = nu[0];
  if (!capi::tile32_kernel_ok(n, m)) return false;
  for (int k = 0; k <= N; ++k)
    if (nx[k] != n || (k < N && nu[k] != m)) return false;
  *n_out = n; *m_out = m;
  return true;
}

// returns 0 when the path ran to completion (outputs and status in the pinned arena w.host), 1 to fall back
int tile_backward(Workspace& w, const Layout& L, int n, int m, int N, const double* const* A, const double* const* B,
                  const double* const* f, const double* const* Q, const double* const* R, const double* const* H,
                  const double* const* q, const double* const* r, double reg, bool is_diag) {
  if (std::getenv("ALTRO_TVLQR_DROPIN_GENERIC") != nullptr) return 1;   // A/B hook: always the GENERIC kernel
  const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
  // inputs, one copy: A | B | f | Q | R | H | q | r ; outputs on the device behind them: K | d | P | p | delta_V, status
  const size_t oA = 0, oB = oA + N * nn, of = oB + N * nm, oQ = of + (size_t)N * n, oR = oQ + (N + 1) * nn, oH = oR + N * mm,
               oq = oH + N * nm, orr = oq + (size_t)(N + 1) * n, in_elems = orr + (size_t)N * m;
  const size_t oK = (in_elems + 1) & ~size_t(1), od = oK + N * nm, oP = od + (size_t)N * m, op = oP + (N + 1) * nn,
               odv = op + (size_t)(N + 1) * n, total = odv + 4;
  FastWs& fw = g_fast;
  if (fw.elems < total) {
    if (fw.dev) (void)hipFree(fw.dev);
    if (fw.host) (void)hipHostFree(fw.host);
    fw.dev = nullptr; fw.host = nullptr; fw.elems = 0;
    if (hipMalloc(&fw.dev, total * sizeof(double)) != hipSuccess) { fw.dev = nullptr; return 1; }
    if (hipHostMalloc((void**)&fw.host, total * sizeof(double) + 64, hipHostMallocDefault) != hipSuccess) {
      (void)hipFree(fw.dev); fw.dev = nullptr; fw.host = nullptr; return 1;
    }
    fw.elems = total;
  }
  auto put = [](double* dst, const double* src, size_t cnt) {
    if (src) memcpy(dst, src, sizeof(double) * cnt);
    else memset(dst, 0, sizeof(double) * cnt);
  };
  auto put_sym = [&](double* dst, const double* src, int dim) {   // a diagonal block (is_diag) expanded to the dense one
    if (!is_diag) { put(dst, src, (size_t)dim * dim); return; }
    memset(dst, 0, sizeof(double) * dim * dim);
    for (int i = 0; i < dim; ++i) dst[i + (size_t)i * dim] = src[i];
  };
  double* hs = fw.host;
  for (int k = 0; k < N; ++k) {
    put(hs + oA + k * nn, A[k], nn); put(hs + oB + k * nm, B[k], nm); put(hs + of + (size_t)k * n, f[k], n);
    put_sym(hs + oQ + k * nn, Q[k], n); put_sym(hs + oR + k * mm, R[k], m);
    put(hs + oH + k * nm, is_diag ? nullptr : H[k], nm);
    put(hs + oq + (size_t)k * n, q[k], n); put(hs + orr + (size_t)k * m, r[k], m);
  }
  put_sym(hs + oQ + N * nn, Q[N], n); put(hs + oq + (size_t)N * n, q[N], n);
  double* dv = fw.dev;
  if (hipMemcpyAsync(dv, hs, in_elems * sizeof(double), hipMemcpyHostToDevice, w.stream) != hipSuccess) return 1;
  Tile32Args a{};
  a.A = dv + oA; a.B = dv + oB; a.f = dv + of; a.Q = dv + oQ; a.R = dv + oR; a.H = dv + oH; a.q = dv + oq; a.r = dv + orr;
  a.K = dv + oK; a.d = dv + od; a.P = dv + oP; a.p = dv + op;
  a.delta_V = dv + odv; a.status = reinterpret_cast<int*>(dv + odv + 2);
  a.N = N; a.batch = 1; a.n = n; a.m = m; a.reg = reg; a.no_f = 0;
  a.L = tile32_lds_layout(n, m);
  if (capi::tile32_backward_dispatch(capi::Tile32Launch{w.stream, nullptr, nullptr}, a) || hipGetLastError() != hipSuccess) return 1;
  const size_t lds = (4 * nn + 4 * nm + 2 * mm + 4 * (size_t)n + 2 * (size_t)m) * sizeof(double);
  hipLaunchKernelGGL(dropin_qblocks_wave_kernel, dim3(N + 1), dim3(64), lds, w.stream, a, reg, w.host_dev, (const int64_t*)w.dev_off,
                     (int64_t)L.total);
  if (hipGetLastError() != hipSuccess) return 1;
  if (hipStreamSynchronize(w.stream) != hipSuccess) return 1;
  int status;
  memcpy(&status, w.host + L.total - 2, sizeof(int));
  return status == TVLQR_SUCCESS ? 0 : 1;   // a failed factorisation: the GENERIC path reproduces the reference's early return
}

}  // namespace

int tvlqr_BackwardPass(const int* nx, const int* nu, int num_horizon, const lqr_float* const* A,
                       const lqr_float* const* B, const lqr_float* const* f, const lqr_float* const* Q,
                       const lqr_float* const* R, const lqr_float* const* H, const lqr_float* const* q,
                       const lqr_float* const* r, lqr_float reg, lqr_float** K, lqr_float** d,
                       lqr_float** P, lqr_float** p, lqr_float* delta_V, lqr_float** Qxx, lqr_float** Quu,
                       lqr_float** Qux, lqr_float** Qx, lqr_float** Qu, lqr_float** Qxx_tmp,
                       lqr_float** Quu_tmp, lqr_float** Qux_tmp, lqr_float** Qx_tmp, lqr_float** Qu_tmp,
                       bool linear_only_update, bool is_diag) {
  (void)linear_only_update;   // accepted and ignored, like tvlqr.cpp:78
  if (!have_device()) return TVLQR_NO_DEVICE;
  const int N = num_horizon;
  if (!dims_supported(nx, nu, N)) return TVLQR_UNSUPPORTED_SIZE;
  SeamClock clk;
  Workspace& w = g_ws;
  const Layout& L = make_layout(nx, nu, N, is_diag, 0);
  if (prepare(w, L, nx, nu)) return TVLQR_NO_DEVICE;
  double* hs = w.host;
  clk.lap(0);
  int fn = 0, fm = 0;
  const bool fast = (fast_shape(nx, nu, N, &fn, &fm) && fast_backward(w, L, fn, fm, N, A, B, f, Q, R, H, q, r, reg, is_diag) == 0) ||
                    (tile_shape(nx, nu, N, &fn, &fm) && tile_backward(w, L, fn, fm, N, A, B, f, Q, R, H, q, r, reg, is_diag) == 0);
  auto put = [&](int arr, int k, const double* src, int64_t cnt) {
    if (!cnt) return;
    if (src) memcpy(hs + L.off[(size_t)k * G_NUM + arr], src, sizeof(double) * cnt);
    else memset(hs + L.off[(size_t)k * G_NUM + arr], 0, sizeof(double) * cnt);   // an absent block is a zero block
  };
  if (!fast) {
  for (int k = 0; k <= N; ++k) {
    const int n = nx[k];
    put(G_Q, k, Q[k], is_diag ? n : (int64_t)n * n);
    put(G_q, k, q[k], n);
    if (k < N) {
      const int m = nu[k], n2 = nx[k + 1];
      put(G_A, k, A[k], (int64_t)n2 * n);
      put(G_B, k, B[k], (int64_t)n2 * m);
      put(G_f, k, f[k], n2);
      put(G_R, k, R[k], is_diag ? m : (int64_t)m * m);
      if (!is_diag) put(G_H, k, H[k], (int64_t)m * n);
      put(G_r, k, r[k], m);
    }
  }
  if (hipMemcpyAsync(w.dev, hs, (size_t)L.group_end[1] * sizeof(double), hipMemcpyHostToDevice, w.stream) != hipSuccess) return TVLQR_NO_DEVICE;
  GenericArgs<double> a = make_args(w, L, reg, is_diag, 0, L.group_end[1]);
  const size_t lds4 = generic_backward_lds_bytes<double>(a.nmax, a.mmax), lds3 = generic_backward_lds_bytes<double>(a.nmax, a.mmax, true);
  const bool late_q = lds4 > kGenericLdsLimit && lds3 <= kGenericLdsLimit;   // (the form without a block for Qxx keeps n up to ~37 in LDS)
  const size_t lds = late_q ? lds3 : lds4;
  if (late_q) {
    hipLaunchKernelGGL((generic_backward_kernel<double, false, false, true>), dim3(1), dim3(64), lds, w.stream, a);
  } else if (lds > kGenericLdsLimit) {
    if (w.big_bytes < lds) {
      if (w.big) (void)hipFree(w.big);
      w.big = nullptr; w.big_bytes = 0;
      if (hipMalloc((void**)&w.big, lds) != hipSuccess) { w.big = nullptr; return TVLQR_NO_DEVICE; }
      w.big_bytes = lds;
    }
    a.ws = w.big; a.ws_stride = 0;
    hipLaunchKernelGGL((generic_backward_kernel<double, true>), dim3(1), dim3(64), 0, w.stream, a);
  } else {
    hipLaunchKernelGGL((generic_backward_kernel<double, false>), dim3(1), dim3(64), lds, w.stream, a);
  }
  if (hipGetLastError() != hipSuccess) return TVLQR_NO_DEVICE;
  if (hipStreamSynchronize(w.stream) != hipSuccess) return TVLQR_NO_DEVICE;
  }
  clk.lap(1);
  int status;
  memcpy(&status, hs + L.total - 2, sizeof(int));
  auto get = [&](int arr, int k, double* dst, int64_t cnt) {
    if (dst && cnt) memcpy(dst, hs + L.off[(size_t)k * G_NUM + arr], sizeof(double) * cnt);
  };
  // Knot points the recursion reached: all of them on success, k >= status on failure
  const int k_lo = (status == TVLQR_SUCCESS) ? 0 : status;
  get(G_P, N, P[N], (int64_t)nx[N] * nx[N]);
  get(G_p, N, p[N], nx[N]);
  for (int k = N - 1; k >= k_lo; --k) {
    const int n = nx[k], m = nu[k];
    const bool failed_here = (k == status);
    get(G_K, k, K[k], (int64_t)m * n);
    get(G_d, k, d[k], m);
    if (!failed_here) {
      get(G_P, k, P[k], (int64_t)n * n);
      get(G_p, k, p[k], n);
    }
    get(G_Qxx, k, Qxx[k], (int64_t)n * n);
    get(G_Quu, k, Quu[k], (int64_t)m * m);
    get(G_Qux, k, Qux[k], (int64_t)m * n);
    get(G_Qx, k, Qx[k], n);
    get(G_Qu, k, Qu[k], m);
    get(G_Qxx_tmp, k, Qxx_tmp[k], (int64_t)n * n);
    get(G_Quu_tmp, k, Quu_tmp[k], (int64_t)m * m);
    get(G_Qux_tmp, k, Qux_tmp[k], (int64_t)m * n);
    get(G_Qx_tmp, k, Qx_tmp[k], n);
    if (!failed_here) get(G_Qu_tmp, k, Qu_tmp[k], m);
  }
  delta_V[0] = hs[L.total - 4];
  delta_V[1] = hs[L.total - 3];
  clk.lap(2);
  return status;
}

// One throw-away backward pass on blocks of this shape (A = B = 0, Q = R = I): device initialisation, the code object's load, the
// workspace, the pinned arenas and the layout tables happen here instead of inside the caller's first timed sweep
// (SolverImpl::Initialize calls it: a 200-step MPC run of 0.17 s spent a third of its time there).  Uniform dimensions run the dense
// form a solver uses; per-knot-point dimensions the diagonal form (one buffer of ones serves every block size).
int tvlqr_hip_warmup(const int* nx, const int* nu, int num_horizon) {
  if (!have_device()) return TVLQR_NO_DEVICE;
  const int N = num_horizon;
  if (N < 1 || !dims_supported(nx, nu, N)) return TVLQR_UNSUPPORTED_SIZE;
  int nmax = 0, mmax = 0;
  for (int k = 0; k <= N; ++k) { nmax = std::max(nmax, nx[k]); if (k < N) mmax = std::max(mmax, nu[k]); }
  bool uniform = true;
  for (int k = 0; k <= N && uniform; ++k) uniform = nx[k] == nmax && (k == N || nu[k] == mmax);
  const size_t nn = (size_t)nmax * nmax, nm = (size_t)nmax * mmax, mm = (size_t)mmax * mmax;
  std::vector<double> zero(std::max({nn, nm, mm, (size_t)1}), 0.0), In(nn, 0.0), Im(mm, 0.0), ones((size_t)std::max(nmax, mmax) + 1, 1.0);
  for (int i = 0; i < nmax; ++i) In[(size_t)i * nmax + i] = 1.0;
  for (int i = 0; i < mmax; ++i) Im[(size_t)i * mmax + i] = 1.0;
  std::vector<const double*> pz(N + 1, zero.data()), pQ(N + 1, uniform ? In.data() : ones.data()), pR(N + 1, uniform ? Im.data() : ones.data());
  const size_t each[14] = {nm, (size_t)mmax, nn, (size_t)nmax, nn, mm, nm, (size_t)nmax, (size_t)mmax, nn, mm, nm, (size_t)nmax, (size_t)mmax};
  size_t tot = 0;
  for (size_t e : each) tot += e;
  std::vector<double> out((size_t)(N + 1) * tot + 1, 0.0);
  std::vector<double*> slots[14];
  double* cur = out.data();
  for (int i = 0; i < 14; ++i) {
    slots[i].assign(N + 1, nullptr);
    for (int k = 0; k <= N; ++k) { slots[i][k] = cur; cur += each[i]; }
  }
  double dv[2];
  return tvlqr_BackwardPass(nx, nu, N, pz.data(), pz.data(), pz.data(), pQ.data(), pR.data(), pz.data(), pz.data(), pz.data(), 0.0,
                            slots[0].data(), slots[1].data(), slots[2].data(), slots[3].data(), dv, slots[4].data(), slots[5].data(),
                            slots[6].data(), slots[7].data(), slots[8].data(), slots[9].data(), slots[10].data(), slots[11].data(),
                            slots[12].data(), slots[13].data(), false, !uniform);
}

int tvlqr_ForwardPass(const int* nx, const int* nu, int num_horizon, const lqr_float* const* A,
                      const lqr_float* const* B, const lqr_float* const* f, const lqr_float* const* K,
                      const lqr_float* const* d, const lqr_float* const* P, const lqr_float* const* p,
                      const lqr_float* x0, lqr_float** x, lqr_float** u, lqr_float** y) {
  if (!have_device()) return TVLQR_NO_DEVICE;
  const int N = num_horizon;
  if (!dims_supported(nx, nu, N)) return TVLQR_UNSUPPORTED_SIZE;
  Workspace& w = g_ws;
  const Layout& L = make_layout(nx, nu, N, false, 1);
  if (prepare(w, L, nx, nu)) return TVLQR_NO_DEVICE;
  double* hs = w.host;
  auto put = [&](int arr, int k, const double* src, int64_t cnt) {
    if (!cnt) return;
    if (src) memcpy(hs + L.off[(size_t)k * G_NUM + arr], src, sizeof(double) * cnt);
    else memset(hs + L.off[(size_t)k * G_NUM + arr], 0, sizeof(double) * cnt);   // an absent block is a zero block
  };
  for (int k = 0; k <= N; ++k) {
    const int n = nx[k];
    if (y) {
      put(G_P, k, P[k], (int64_t)n * n);
      put(G_p, k, p[k], n);
    }
    if (k < N) {
      const int m = nu[k], n2 = nx[k + 1];
      put(G_A, k, A[k], (int64_t)n2 * n);
      put(G_B, k, B[k], (int64_t)n2 * m);
      put(G_f, k, f[k], n2);
      put(G_K, k, K[k], (int64_t)m * n);
      put(G_d, k, d[k], m);
    }
  }
  memcpy(hs + L.total - 4 - L.xstage, x0, sizeof(double) * nx[0]);
  if (hipMemcpyAsync(w.dev, hs, (size_t)L.group_end[2] * sizeof(double), hipMemcpyHostToDevice, w.stream) != hipSuccess) return TVLQR_NO_DEVICE;
  GenericArgs<double> a = make_args(w, L, 0.0, false, y ? 1 : 0, L.group_end[2]);
  const size_t staged = generic_forward_lds_bytes(a.nmax, a.mmax, a.want_y, sizeof(double));
  if (staged <= kGenericLdsLimit) hipLaunchKernelGGL((generic_forward_kernel<double, true>), dim3(1), dim3(64), staged, w.stream, a);
  else hipLaunchKernelGGL((generic_forward_kernel<double, false>), dim3(1), dim3(64), (size_t)(2 * a.nmax + a.mmax) * sizeof(double) + 64, w.stream, a);
  if (hipGetLastError() != hipSuccess) return TVLQR_NO_DEVICE;
  if (hipStreamSynchronize(w.stream) != hipSuccess) return TVLQR_NO_DEVICE;
  for (int k = 0; k <= N; ++k) {
    memcpy(x[k], hs + L.off[(size_t)k * G_NUM + G_x], sizeof(double) * nx[k]);
    if (y) memcpy(y[k], hs + L.off[(size_t)k * G_NUM + G_y], sizeof(double) * nx[k]);
    if (k < N) memcpy(u[k], hs + L.off[(size_t)k * G_NUM + G_u], sizeof(double) * nu[k]);
  }
  return TVLQR_SUCCESS;
}
