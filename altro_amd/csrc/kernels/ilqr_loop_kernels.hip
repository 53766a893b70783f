// ilqr_loop_kernels.hip -- per-problem bookkeeping kernels of the batched AL-iLQR loop (one thread per problem):
// the resumable line search, convergence tests, outer (dual / penalty) updates, regularisation retry.
// Non-template kernels: included by exactly one translation unit (ilqr_launch_f64.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "ilqr_loop_logic.h"
#include "ilqr_types.h"

namespace altro_hip {

// AltroStats of every problem (solver.cpp:503-509) in the layout the C ABI hands out
__global__ void ilqr_results_kernel(const IlqrProb* __restrict__ prob, IlqrResult* __restrict__ out, int batch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  out[b] = ilqr_result_of(prob[b]);
}

// Straggler compaction of the one-launch solve (capi_solve.hip, run_fused): the still-running problems of a launch, listed in
// order and dealt round-robin to the workgroups of the next one (ilqr_list_layout).  One workgroup: every thread counts the set
// flags of its contiguous run of slots, an LDS scan places the runs.
__global__ __launch_bounds__(1024) void ilqr_list_running_kernel(const int* __restrict__ flags, const int* __restrict__ list_in,
                                                                 int count_in, int* __restrict__ list_out, int resident) {
  __shared__ int s_sum[1024];
  const int tid = (int)threadIdx.x;
  const int per = (count_in + 1023) / 1024;
  const int lo = tid * per < count_in ? tid * per : count_in;
  const int hi = lo + per < count_in ? lo + per : count_in;
  int c = 0;
  for (int i = lo; i < hi; ++i) c += flags[i] != 0 ? 1 : 0;
  s_sum[tid] = c;
  __syncthreads();
  for (int step = 1; step < 1024; step <<= 1) {   // inclusive scan
    const int v = tid >= step ? s_sum[tid - step] : 0;
    __syncthreads();
    s_sum[tid] += v;
    __syncthreads();
  }
  const int total = s_sum[1023];
  if (total == 0) return;
  int group, groups;
  ilqr_list_layout(total, resident, &group, &groups);
  for (int i = tid; i < group * groups; i += 1024) list_out[i] = -1;
  __syncthreads();
  int pos = s_sum[tid] - c;
  for (int i = lo; i < hi; ++i)
    if (flags[i] != 0) {
      list_out[(pos % groups) * group + pos / groups] = list_in ? list_in[i] : i;
      ++pos;
    }
}

// The counters of a counting kernel, handed to the host without a copy on the stream: the block that finishes last (a ticket in
// counters[7]) stores counters[0..6] into the launch's slot of host-mapped memory; the host reads it after the event it recorded
// behind the launch.  Called by every thread of the kernel (no early exits before it).
__device__ __forceinline__ void ilqr_publish_counters(const IlqrLoopArgs& a) {
  if (!a.counters_pub) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const int ticket = atomicAdd(&a.counters[7], 1);
    if (ticket == (int)gridDim.x - 1) {
      __threadfence();
#pragma unroll
      for (int i = 0; i < 7; ++i)
        __hip_atomic_store(&a.counters_pub[i], __hip_atomic_load(&a.counters[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---- batched line search + sweep bookkeeping (one thread per problem) ---------------------------------
__global__ void ilqr_loop_init_kernel(IlqrLoopArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  IlqrProb& p = a.prob[b];
  ilqr_prob_init(p);
  a.reg[b] = a.reg_initial;
  a.active[b] = 1;
  a.alpha[b] = 0.0;
}

// after merit(alpha = 0): ForwardPass's head (solver.cpp:241-249)
__global__ void ilqr_ls_begin_kernel(IlqrLoopArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < a.batch) {
    const bool was_running = a.prob[b].running != 0;
    if (ilqr_ls_begin_body(a, b)) atomicAdd(&a.counters[0], 1);
    if (a.spec_flip && was_running) {
      if (a.spec_refresh[b]) atomicAdd(&a.counters[3], 1);
      if (!a.stat_done[b]) atomicAdd(&a.counters[4], 1);
    }
  }
  ilqr_publish_counters(a);
}

// after merit(alpha[b]): advance every searching problem's state machine
__global__ void ilqr_ls_feed_kernel(IlqrLoopArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < a.batch && ilqr_ls_feed_body<true>(a, b)) atomicAdd(&a.counters[0], 1);
  ilqr_publish_counters(a);
}

// end of one sweep (solver.cpp:459-502): convergence test, bookkeeping; `active` := still running
__global__ void ilqr_finish_iter_kernel(IlqrLoopArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < a.batch && ilqr_finish_iter_body(a, b)) atomicAdd(&a.counters[1], 1);
  ilqr_publish_counters(a);
}

// Regularisation retry -- an EXTENSION: the reference passes reg = 0 and ignores a failed factorisation
// (tvlqr.cpp:159-164, solver.cpp:363, :449).  After a backward pass, every running problem whose Cholesky
// failed gets reg <- max(reg * scale, reg_min) and is marked for another backward pass (counters[2] counts
// them); a problem that succeeded relaxes reg <- max(reg / scale, reg_initial) for its next sweep.
__global__ void ilqr_reg_retry_kernel(IlqrLoopArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < a.batch && ilqr_reg_retry_body(a, b)) atomicAdd(&a.counters[2], 1);
  ilqr_publish_counters(a);
}

// SetPenalty (solver.cpp:429) after the initial gradient
__global__ void ilqr_set_penalty_kernel(IlqrLoopArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  a.prob[b].rho = a.penalty_initial;
}

// PenaltyUpdate (solver.cpp:396-409) after the duals of every knot point have been updated with the old
// penalty; then the projected duals are refreshed with the new one (solver.cpp:483-486).  `active` :=
// problems whose cost gradients have to be recomputed.
__global__ void ilqr_penalty_update_kernel(IlqrLoopArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  ilqr_penalty_update_body(a, b);
}

// set `active` := running (used before the per-sweep kernels)
__global__ void ilqr_mark_running_kernel(IlqrLoopArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  ilqr_mark_running_body(a, b);
}

}  // namespace altro_hip
