// tools/valubench.hip -- per-instruction issue cost probes (cycles per wave-instruction per SIMD) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(64) void probe(double* out, long long* cyc, int iters) {
  double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 1.0000001, c = 1e-9;
  f64x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 8 independent f64 fma
      a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
      a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
    } else if (MODE == 1) {  // 8 dependent f64 fma
      a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c);
      a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c);
    } else if (MODE == 2) {  // 8 independent f64 add
      a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c;
    } else if (MODE == 3) {  // 2 independent MFMA f64 16x16x4 chains
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a1, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, a3, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a1, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, a3, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a1, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, a3, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a1, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, a3, acc1, 0, 0, 0);
    } else if (MODE == 4) {  // 8 dependent MFMA (same accumulator)
#pragma unroll
      for (int q = 0; q < 8; ++q) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a1, acc0, 0, 0, 0);
    } else if (MODE == 5) {  // 8 f32 fma independent
      float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3;
#pragma unroll
      for (int q = 0; q < 2; ++q) { f0 = fmaf(f0, 1.0001f, 1e-3f); f1 = fmaf(f1, 1.0001f, 1e-3f); f2 = fmaf(f2, 1.0001f, 1e-3f); f3 = fmaf(f3, 1.0001f, 1e-3f); }
      a0 = f0; a1 = f1; a2 = f2; a3 = f3;
    } else if (MODE == 6) {  // 4 MFMA + 8 independent fma interleaved (same wave)
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a1, acc0, 0, 0, 0);
      a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, a3, acc1, 0, 0, 0);
      a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a1, acc0, 0, 0, 0);
      a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, a3, acc1, 0, 0, 0);
      a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc0[0] + acc1[1] + acc0[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, int blocks, int ops_per_iter) {
  double* out; long long* cyc; hipMalloc(&out, blocks * 64 * 8); hipMalloc(&cyc, blocks * 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<blocks, 64>>>(out, cyc, 100);
  hipEventRecord(e0); probe<MODE><<<blocks, 64>>>(out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  // waves per SIMD = blocks / 1024 (rounded up); wall cycles at ~2.4 GHz from event time
  double wps = blocks / 1024.0; if (wps < 1) wps = 1;
  printf("%-44s blocks=%5d  s_memtime/iter=%8.1f  wall_ns/iter=%8.2f  ns per instr per SIMD=%6.2f\n", name, blocks, (double)h / iters,
         ms * 1e6 / iters, ms * 1e6 / iters / ops_per_iter / wps);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int blocks : {1024, 4096}) {
    run<0>("8 indep v_fma_f64", blocks, 8);
    run<1>("8 dep   v_fma_f64", blocks, 8);
    run<2>("8 indep v_add_f64", blocks, 8);
    run<3>("8 MFMA f64 16x16x4 (2 chains)", blocks, 8);
    run<4>("8 MFMA f64 16x16x4 (1 chain)", blocks, 8);
    run<5>("8 indep v_fma_f32 (+cvt)", blocks, 8);
    run<6>("4 MFMA + 8 fma interleaved", blocks, 12);
  }
  return 0;
}
