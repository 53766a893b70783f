// capi_stats.hip -- C ABI: solver statistics, reduced on the device and (across GPUs) by RCCL over xGMI.
//
// problem instances are sharded over the GPUs of a node and never talk during sweeps; the
// only thing that crosses GPUs is what SolverImpl::Solve reports (solver.cpp:464-469, :492-509; AltroStats,
// solver_stats.hpp:14-25), summed / maximised over the problems:
//     ncclSum, ncclInt64  over {problems, Cholesky failures, converged, iterations, problems with a non-finite result}
//     ncclSum, ncclDouble over {cost, delta_V0, delta_V1}
//     ncclMax, ncclDouble over {stationarity, feasibility, |x_N|}
// (the counts travel as the 64-bit integers SURVEY 8(e) names: the reduced vector holds their bit patterns in its first five slots)
// Per GPU the reduction is a deterministic two-stage kernel pair (fixed tree order: the result does not depend on
// scheduling), so nothing but 16 doubles ever goes to the host.  RCCL is resolved with dlopen at first use -- the
// copy the process already has (torch's, an MPI program's) if there is one -- so libaltro_hip.so itself links only
// against the HIP runtime.
#include "capi_internal.h"

#include <dlfcn.h>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <rccl/rccl.h>   // types and prototypes only; every call goes through the table below

using namespace altro_hip;
using namespace altro_hip::capi;

namespace {

// positions inside the reduced vector (device, kStatsStride doubles)
enum { ST_PROBLEMS = 0, ST_CHOL, ST_CONVERGED, ST_ITERATIONS, ST_NONFINITE, ST_NINT /* the counts */,
       ST_COST = 5, ST_DV0, ST_DV1, ST_NSUM /* end of the sums */,
       ST_MAX0 = 8, ST_MAX_STAT = 8, ST_MAX_FEAS, ST_MAX_XN, ST_NMAX = 3 };

struct StatsIn {
  const int* status;        // [batch] -1 or the failing knot point (valid when have_bwd)
  const void* delta_V;      // [batch][2], element type T
  const IlqrProb* prob;     // per-problem control blocks (valid when have_solve)
  const void* xN;           // x_N of problem 0; element (b, i) at xN[b * x_bs + i * x_is]
  int64_t x_bs, x_is;
  int n, batch;
  int have_bwd, have_fwd, have_solve;
};

// max that keeps a NaN (fmax drops it): a diverged problem must show in the batch's maxima, as it does in the
// per-problem results and in sum_cost
__device__ __forceinline__ double nanmax(double a, double b) { return (a != a || b != b) ? (a + b) : fmax(a, b); }

template <typename T>
__global__ __launch_bounds__(256) void stats_partial_kernel(StatsIn a, double* __restrict__ partial) {
  __shared__ double sh[256][11];
  double acc[11];
#pragma unroll
  for (int i = 0; i < 11; ++i) acc[i] = 0.0;
  // fixed problem -> thread assignment and in-order accumulation: the sums are reproducible bit for bit
  for (int b = blockIdx.x * 256 + threadIdx.x; b < a.batch; b += gridDim.x * 256) {
    acc[ST_PROBLEMS] += 1.0;
    if (a.have_bwd) {
      if (a.status[b] != ALTRO_HIP_TVLQR_SUCCESS) acc[ST_CHOL] += 1.0;
      else {
        acc[ST_DV0] += (double)((const T*)a.delta_V)[2 * (size_t)b];
        acc[ST_DV1] += (double)((const T*)a.delta_V)[2 * (size_t)b + 1];
      }
    }
    bool bad = false;   // any non-finite number among this problem's reported quantities
    if (a.have_solve) {
      const IlqrProb& p = a.prob[b];
      if (p.status == 0) acc[ST_CONVERGED] += 1.0;
      acc[ST_ITERATIONS] += (double)p.iterations;
      const double cost = p.ls_iters > 0 ? p.ls.phi : p.phi0;
      acc[ST_COST] += cost;
      acc[ST_MAX_STAT] = nanmax(acc[ST_MAX_STAT], fabs(p.stationarity));
      acc[ST_MAX_FEAS] = nanmax(acc[ST_MAX_FEAS], p.feasibility);
      bad = !isfinite(cost) || !isfinite(p.stationarity) || !isfinite(p.feasibility);
    }
    if (a.have_fwd) {
      const T* x = (const T*)a.xN + (size_t)b * a.x_bs;
      double mx = 0.0;
      for (int i = 0; i < a.n; ++i) mx = nanmax(mx, fabs((double)x[(size_t)i * a.x_is]));
      acc[ST_MAX_XN] = nanmax(acc[ST_MAX_XN], mx);
      bad = bad || !isfinite(mx);
    }
    if (bad) acc[ST_NONFINITE] += 1.0;
  }
#pragma unroll
  for (int i = 0; i < 11; ++i) sh[threadIdx.x][i] = acc[i];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
#pragma unroll
      for (int i = 0; i < ST_NSUM; ++i) sh[threadIdx.x][i] += sh[threadIdx.x + s][i];
#pragma unroll
      for (int i = ST_MAX0; i < ST_MAX0 + ST_NMAX; ++i) sh[threadIdx.x][i] = nanmax(sh[threadIdx.x][i], sh[threadIdx.x + s][i]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 11) partial[(size_t)blockIdx.x * kStatsStride + threadIdx.x] = sh[0][threadIdx.x];
}

__global__ __launch_bounds__(64) void stats_final_kernel(const double* __restrict__ partial, int nblk,
                                                         double* __restrict__ red) {
  const int i = threadIdx.x;
  if (i >= kStatsStride) return;
  double v = 0.0;
  if (i < ST_NSUM) for (int k = 0; k < nblk; ++k) v += partial[(size_t)k * kStatsStride + i];    // in block order
  else if (i >= ST_MAX0 && i < ST_MAX0 + ST_NMAX)
    for (int k = 0; k < nblk; ++k) v = nanmax(v, partial[(size_t)k * kStatsStride + i]);
  // the counts leave as 64-bit integers (exact: they were whole numbers below 2^53 all along)
  if (i < ST_NINT) reinterpret_cast<long long*>(red)[i] = (long long)llrint(v);
  else red[i] = v;
}

// enqueue the local reduction of h on its stream; h->st_red then holds the vector
int stats_local_launch(altro_hip_batch* h) {
  StatsIn a{};
  a.status = h->status; a.delta_V = h->delta_V; a.prob = h->i_prob; a.n = h->n; a.batch = h->batch;
  a.have_bwd = h->backward_done ? 1 : 0;
  a.have_fwd = h->forward_done ? 1 : 0;
  a.have_solve = (h->solve_done && h->i_prob) ? 1 : 0;
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {          // candidate records [k][b][28] = x 12 | y 12 | u 4
    a.xN = (const char*)h->m_xuy + (size_t)h->N * h->m_st.xuy_ks * h->esz; a.x_bs = h->m_st.xuy_bs; a.x_is = 1;
  } else if (h->plan == ALTRO_HIP_PLAN_LANE) {     // [k][element][batch], elements x n | y n | u m
    a.xN = (const char*)h->l_xuy + (size_t)h->N * (2 * h->n + h->m) * h->batch * h->esz; a.x_bs = 1; a.x_is = h->batch;
  } else {                                         // reference layout [b][N+1][n]
    const int64_t xN_at = h->ragged ? h->g_bstride[G_x] - h->nxv[h->N] : (int64_t)h->N * h->n;
    if (h->ragged) a.n = h->nxv[h->N];
    a.xN = (const char*)h->g_arr[G_x] + (size_t)xN_at * h->esz; a.x_bs = h->g_bstride[G_x]; a.x_is = 1;
  }
  const int nblk = std::min(kStatsBlocks, (h->batch + 255) / 256);
  if (h->dtype == ALTRO_HIP_F64)
    hipLaunchKernelGGL(stats_partial_kernel<double>, dim3(nblk), dim3(256), 0, h->stream, a, h->st_partial);
  else
    hipLaunchKernelGGL(stats_partial_kernel<float>, dim3(nblk), dim3(256), 0, h->stream, a, h->st_partial);
  hipLaunchKernelGGL(stats_final_kernel, dim3(1), dim3(64), 0, h->stream, h->st_partial, nblk, h->st_red);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "statistics kernel launch: %s", hipGetErrorString(e));
  return 0;
}

int stats_read(altro_hip_batch* h, altro_hip_stats* out) {
  double v[kStatsStride];
  HIP_TRY(hipMemcpyAsync(v, h->st_red, sizeof(v), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  int64_t cnt[ST_NINT];
  std::memcpy(cnt, v, sizeof(cnt));
  out->problems = cnt[ST_PROBLEMS];
  out->cholesky_failures = cnt[ST_CHOL];
  out->converged = cnt[ST_CONVERGED];
  out->iterations = cnt[ST_ITERATIONS];
  out->sum_cost = v[ST_COST];
  out->sum_delta_V0 = v[ST_DV0];
  out->sum_delta_V1 = v[ST_DV1];
  out->max_stationarity = v[ST_MAX_STAT];
  out->max_feasibility = v[ST_MAX_FEAS];
  out->max_abs_xN = v[ST_MAX_XN];
  out->non_finite = cnt[ST_NONFINITE];
  return 0;
}

// ---- RCCL, resolved at run time -----------------------------------------------------------------------------
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

int rccl(const Rccl** out) {
  // resolved once per process, whichever host thread gets here first (one thread per GPU may call comm_create at once)
  static Rccl r;
  static std::string why = "symbols missing";
  static std::once_flag once;
  std::call_once(once, [] {
    const char* env = std::getenv("ALTRO_HIP_RCCL");
    void* lib = env ? dlopen(env, RTLD_NOW | RTLD_GLOBAL) : nullptr;
    // the copy this process already uses (a communicator must be driven by the library that made it) ...
    for (const char* name : {"librccl.so.1", "librccl.so"})
      if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    // ... else the system one
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if (!lib) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (!lib) {
          const char* err = dlerror();   // read ONCE: the call clears the message
          if (err) why = err;
        }
      }
    if (!lib) return;
    const char* missing = nullptr;
#define SYM(field, name) \
  do { r.field = (decltype(r.field))dlsym(lib, name); if (!r.field && !missing) missing = name; } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommInitAll, "ncclCommInitAll");
    SYM(CommDestroy, "ncclCommDestroy"); SYM(AllReduce, "ncclAllReduce"); SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    if (missing) why = std::string("symbol ") + missing + " not found in the RCCL library that was loaded";
    else r.lib = lib;
  });
  if (!r.lib) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "RCCL (librccl.so.1) could not be loaded: %s", why.c_str());
  *out = &r;
  return 0;
}

#define RCCL_TRY(R, expr)                                                                             \
  do {                                                                                                \
    ncclResult_t r_ = (expr);                                                                         \
    if (r_ != ncclSuccess) return fail(ALTRO_HIP_ERR_HIP, "%s failed: %s", #expr, (R)->GetErrorString(r_)); \
  } while (0)

}  // namespace

struct altro_hip_comm {
  ncclComm_t comm = nullptr;
  int device = 0, rank = 0, world = 1;
};

namespace {
int enqueue_allreduce(const Rccl* R, altro_hip_batch* h, altro_hip_comm* c) {
  RCCL_TRY(R, R->AllReduce(h->st_red, h->st_red, ST_NINT, ncclInt64, ncclSum, c->comm, h->stream));
  RCCL_TRY(R, R->AllReduce(h->st_red + ST_NINT, h->st_red + ST_NINT, ST_NSUM - ST_NINT, ncclDouble, ncclSum, c->comm, h->stream));
  RCCL_TRY(R, R->AllReduce(h->st_red + ST_MAX0, h->st_red + ST_MAX0, ST_NMAX, ncclDouble, ncclMax, c->comm, h->stream));
  return 0;
}
}  // namespace

extern "C" {

int altro_hip_stats_reduce(altro_hip_batch* h, altro_hip_stats* out) {
  int rc = check(h);
  if (rc) return rc;
  if (!out) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "out == NULL");
  if (!h->backward_done && !h->solve_done) return fail(ALTRO_HIP_ERR_NOT_SET, "nothing computed yet: no statistics");
  if ((rc = stats_local_launch(h))) return rc;
  return stats_read(h, out);
}

int altro_hip_comm_unique_id(void* id) {
  const Rccl* R;
  int rc = rccl(&R);
  if (rc) return rc;
  if (!id) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "id == NULL");
  static_assert(sizeof(ncclUniqueId) == ALTRO_HIP_COMM_ID_BYTES, "ALTRO_HIP_COMM_ID_BYTES");
  ncclUniqueId u;
  RCCL_TRY(R, R->GetUniqueId(&u));
  std::memcpy(id, &u, sizeof(u));
  return 0;
}

int altro_hip_comm_create(altro_hip_comm** out, int device, int rank, int world, const void* id) {
  if (!out || !id) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "out / id == NULL");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "bad rank %d of %d", rank, world);
  if (device < 0 || device >= altro_hip_device_count())
    return fail(ALTRO_HIP_ERR_NO_DEVICE, "no HIP device %d", device);
  const Rccl* R;
  int rc = rccl(&R);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(device));
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  // ncclCommInitRank blocks until every rank of the world has called it: a rank that never arrives (died at start-up, bound to
  // the wrong device, a unique id that was not the one rank 0 made) would hang the others for good.  It runs on a helper thread
  // and this call gives up after ALTRO_HIP_COMM_TIMEOUT_S seconds (default 120) with a message that says what was waited for.
  struct InitState { std::mutex mu; std::condition_variable cv; bool done = false; ncclResult_t res = ncclSuccess; ncclComm_t comm = nullptr; };
  auto st = std::make_shared<InitState>();
  double timeout_s = 120.0;
  if (const char* e = std::getenv("ALTRO_HIP_COMM_TIMEOUT_S")) { const double v = std::atof(e); if (v > 0.0) timeout_s = v; }
  std::thread([st, R, world, u, rank, device]() {
    ncclComm_t cm = nullptr;
    ncclResult_t r = (hipSetDevice(device) == hipSuccess) ? R->CommInitRank(&cm, world, u, rank) : ncclUnhandledCudaError;
    std::lock_guard<std::mutex> lock(st->mu);
    st->res = r; st->comm = cm; st->done = true;
    st->cv.notify_all();
  }).detach();
  {
    std::unique_lock<std::mutex> lock(st->mu);
    if (!st->cv.wait_for(lock, std::chrono::duration<double>(timeout_s), [&] { return st->done; }))
      return fail(ALTRO_HIP_ERR_HIP, "ncclCommInitRank(rank %d of %d, device %d) did not return within %.0f s: not every rank of the world "
                                     "joined (a rank that died, a wrong device binding or unique id); ALTRO_HIP_COMM_TIMEOUT_S changes the limit",
                  rank, world, device, timeout_s);
  }
  if (st->res != ncclSuccess)
    return fail(ALTRO_HIP_ERR_HIP, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device, R->GetErrorString(st->res));
  altro_hip_comm* c = new altro_hip_comm();
  c->device = device; c->rank = rank; c->world = world; c->comm = st->comm;
  *out = c;
  return 0;
}

int altro_hip_comm_create_all(altro_hip_comm** out, int ndev, const int* devices) {
  if (!out || ndev < 1) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "out == NULL or ndev < 1");
  for (int i = 0; i < ndev; ++i) out[i] = nullptr;
  const int count = altro_hip_device_count();
  std::vector<int> dev(ndev);
  for (int i = 0; i < ndev; ++i) {
    dev[i] = devices ? devices[i] : i;
    if (dev[i] < 0 || dev[i] >= count) return fail(ALTRO_HIP_ERR_NO_DEVICE, "no HIP device %d (count = %d)", dev[i], count);
  }
  const Rccl* R;
  int rc = rccl(&R);
  if (rc) return rc;
  std::vector<ncclComm_t> comms(ndev, nullptr);
  RCCL_TRY(R, R->CommInitAll(comms.data(), ndev, dev.data()));
  for (int i = 0; i < ndev; ++i) {
    out[i] = new altro_hip_comm();
    out[i]->comm = comms[i]; out[i]->device = dev[i]; out[i]->rank = i; out[i]->world = ndev;
  }
  return 0;
}

int altro_hip_comm_rank(const altro_hip_comm* c) { return c ? c->rank : ALTRO_HIP_ERR_BAD_ARGUMENT; }
int altro_hip_comm_world(const altro_hip_comm* c) { return c ? c->world : ALTRO_HIP_ERR_BAD_ARGUMENT; }
int altro_hip_comm_device(const altro_hip_comm* c) { return c ? c->device : ALTRO_HIP_ERR_BAD_ARGUMENT; }

void altro_hip_comm_destroy(altro_hip_comm* c) {
  if (!c) return;
  const Rccl* R;
  if (c->comm && rccl(&R) == 0) {
    (void)hipSetDevice(c->device);
    (void)R->CommDestroy(c->comm);
  }
  delete c;
}

int altro_hip_stats_allreduce(altro_hip_batch* h, altro_hip_comm* comm, altro_hip_stats* out) {
  int rc = check(h);
  if (rc) return rc;
  if (!comm || !out) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "comm / out == NULL");
  if (comm->device != h->device) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "communicator on device %d, handle on device %d", comm->device, h->device);
  if (!h->backward_done && !h->solve_done) return fail(ALTRO_HIP_ERR_NOT_SET, "nothing computed yet: no statistics");
  const Rccl* R;
  if ((rc = rccl(&R))) return rc;
  if ((rc = stats_local_launch(h))) return rc;
  if ((rc = enqueue_allreduce(R, h, comm))) return rc;
  return stats_read(h, out);
}

int altro_hip_stats_allreduce_multi(altro_hip_batch* const* handles, altro_hip_comm* const* comms, int n,
                                    altro_hip_stats* out) {
  if (!handles || !comms || n < 1 || !out) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "handles / comms / out == NULL or n < 1");
  const Rccl* R;
  int rc = rccl(&R);
  if (rc) return rc;
  for (int i = 0; i < n; ++i) {
    if ((rc = check(handles[i]))) return rc;
    if (!comms[i] || comms[i]->device != handles[i]->device)
      return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "handle %d and communicator %d are not on the same device", i, i);
    if (!handles[i]->backward_done && !handles[i]->solve_done) return fail(ALTRO_HIP_ERR_NOT_SET, "handle %d has computed nothing yet", i);
    if ((rc = stats_local_launch(handles[i]))) return rc;
  }
  RCCL_TRY(R, R->GroupStart());
  for (int i = 0; i < n && !rc; ++i) {
    (void)hipSetDevice(handles[i]->device);
    rc = enqueue_allreduce(R, handles[i], comms[i]);
  }
  ncclResult_t ge = R->GroupEnd();
  if (rc) return rc;
  if (ge != ncclSuccess) return fail(ALTRO_HIP_ERR_HIP, "ncclGroupEnd failed: %s", R->GetErrorString(ge));
  for (int i = 1; i < n; ++i) {   // every rank holds the same vector; wait for all, report rank 0's
    if ((rc = check(handles[i]))) return rc;
    HIP_TRY(hipStreamSynchronize(handles[i]->stream));
  }
  if ((rc = check(handles[0]))) return rc;
  return stats_read(handles[0], out);
}

}  // extern "C"
