// tools/membench_f32.hip -- HBM streaming probes for the fp32 (12, 4) records of BASELINE.json configs[4]
// (N = 512, batch = 16384, 364 elements read + 144 written per knot point, 4 B each): what does the backward sweep's
// access pattern reach with NO arithmetic, by bytes per lane, prefetch depth and problems per wave?
//   hipcc --offload-arch=gfx950 -O3 tools/membench_f32.hip -o /tmp/membench_f32 && /tmp/membench_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int RIN = 364, ROUT = 144;

__device__ __forceinline__ int xcd_problem(int block, int nblk) { return (block & 7) * (nblk / 8) + (block >> 3); }

// one wave per problem, 4 B per lane, DEPTH records ahead (register ring, unrolled)
template <int DEPTH>
__global__ __launch_bounds__(64) void rec4(const float* __restrict__ in, float* __restrict__ out, int N, int batch) {
  constexpr int LI = (RIN + 63) / 64, LO = (ROUT + 63) / 64;
  const int lane = threadIdx.x, b = xcd_problem(blockIdx.x, batch);
  float ring[DEPTH][LI];
  auto load = [&](float* r, int k) {
    const float* rec = in + ((size_t)k * batch + b) * RIN;
#pragma unroll
    for (int c = 0; c < LI; ++c) { int e = c * 64 + lane; r[c] = rec[e < RIN ? e : RIN - 1]; }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(ring[d], N - 1 - d);
  float acc = 0.f;
  for (int s0 = 0; s0 < N; s0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int k = N - 1 - s0 - d;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < LI; ++c) s += ring[d][c];
      acc += s;
      load(ring[d], k - DEPTH > 0 ? k - DEPTH : 0);
      float* o = out + ((size_t)(k > 0 ? k : 0) * batch + b) * ROUT;
#pragma unroll
      for (int c = 0; c < LO; ++c) { int e = c * 64 + lane; o[e < ROUT ? e : ROUT - 1] = acc + c; }
    }
  }
}

// one wave per problem, 16 B per lane (float4): 91 + 36 quads per record
template <int DEPTH>
__global__ __launch_bounds__(64) void rec16(const float* __restrict__ in, float* __restrict__ out, int N, int batch) {
  constexpr int PI = RIN / 4, PO = ROUT / 4;
  constexpr int LI = (PI + 63) / 64;
  const int lane = threadIdx.x, b = xcd_problem(blockIdx.x, batch);
  float4 ring[DEPTH][LI];
  auto load = [&](float4* r, int k) {
    const float4* rec = (const float4*)(in + ((size_t)k * batch + b) * RIN);
#pragma unroll
    for (int c = 0; c < LI; ++c) { int e = c * 64 + lane; r[c] = rec[e < PI ? e : PI - 1]; }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(ring[d], N - 1 - d);
  float acc = 0.f;
  for (int s0 = 0; s0 < N; s0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int k = N - 1 - s0 - d;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < LI; ++c) s += ring[d][c].x + ring[d][c].y + ring[d][c].z + ring[d][c].w;
      acc += s;
      load(ring[d], k - DEPTH > 0 ? k - DEPTH : 0);
      float4* o = (float4*)(out + ((size_t)(k > 0 ? k : 0) * batch + b) * ROUT);
      o[lane < PO ? lane : PO - 1] = make_float4(acc, s, acc, s);
    }
  }
}

// PPW problems per wave (adjacent problems: their records are contiguous in a [k][b] slab), 4 B or 16 B per lane
template <int PPW, int VEC, int DEPTH>
__global__ __launch_bounds__(64) void recN(const float* __restrict__ in, float* __restrict__ out, int N, int batch) {
  constexpr int EI = RIN * PPW / VEC, EO = ROUT * PPW / VEC;   // VEC-wide units per wave and knot point
  constexpr int LI = (EI + 63) / 64, LO = (EO + 63) / 64;
  const int lane = threadIdx.x, b0 = xcd_problem(blockIdx.x, batch / PPW) * PPW;
  typedef float vec __attribute__((ext_vector_type(VEC)));
  vec ring[DEPTH][LI];
  auto load = [&](vec* r, int k) {
    const vec* rec = (const vec*)(in + ((size_t)k * batch + b0) * RIN);
#pragma unroll
    for (int c = 0; c < LI; ++c) { int e = c * 64 + lane; r[c] = rec[e < EI ? e : EI - 1]; }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(ring[d], N - 1 - d);
  float acc = 0.f;
  for (int s0 = 0; s0 < N; s0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int k = N - 1 - s0 - d;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < LI; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) s += ring[d][c][v];
      acc += s;
      load(ring[d], k - DEPTH > 0 ? k - DEPTH : 0);
      vec* o = (vec*)(out + ((size_t)(k > 0 ? k : 0) * batch + b0) * ROUT);
      vec val;
#pragma unroll
      for (int v = 0; v < VEC; ++v) val[v] = acc + v;
#pragma unroll
      for (int c = 0; c < LO; ++c) { int e = c * 64 + lane; o[e < EO ? e : EO - 1] = val; }
    }
  }
}

int main() {
  const int N = 512, batch = 16384;
  const size_t in_n = (size_t)batch * N * RIN, out_n = (size_t)batch * N * ROUT;
  float *in, *out;
  CK(hipMalloc(&in, in_n * 4)); CK(hipMalloc(&out, out_n * 4));
  CK(hipMemset(in, 0, in_n * 4)); CK(hipMemset(out, 0, out_n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = ((double)in_n + (double)out_n) * 4;
  auto timeit = [&](const char* name, auto launch) {
    launch();
    hipEventRecord(e0); for (int i = 0; i < 3; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("%-52s %8.3f ms  %8.1f GB/s physical  %8.1f GB/s algorithmic (636 el)\n", name, ms, bytes / ms / 1e6,
           (double)batch * N * 636 * 4 / ms / 1e6);
  };
  printf("fp32 records, N=%d batch=%d: %.2f GB read + %.2f GB written per sweep\n", N, batch, in_n * 4 / 1e9, out_n * 4 / 1e9);
  timeit("1 problem/wave, 4 B/lane, depth 1", [&] { rec4<1><<<batch, 64>>>(in, out, N, batch); });
  timeit("1 problem/wave, 4 B/lane, depth 2", [&] { rec4<2><<<batch, 64>>>(in, out, N, batch); });
  timeit("1 problem/wave, 4 B/lane, depth 4", [&] { rec4<4><<<batch, 64>>>(in, out, N, batch); });
  timeit("1 problem/wave, 4 B/lane, depth 8", [&] { rec4<8><<<batch, 64>>>(in, out, N, batch); });
  timeit("1 problem/wave, 16 B/lane, depth 1", [&] { rec16<1><<<batch, 64>>>(in, out, N, batch); });
  timeit("1 problem/wave, 16 B/lane, depth 2", [&] { rec16<2><<<batch, 64>>>(in, out, N, batch); });
  timeit("1 problem/wave, 16 B/lane, depth 4", [&] { rec16<4><<<batch, 64>>>(in, out, N, batch); });
  timeit("1 problem/wave, 16 B/lane, depth 8", [&] { rec16<8><<<batch, 64>>>(in, out, N, batch); });
  timeit("2 problems/wave, 4 B/lane, depth 2", [&] { recN<2, 1, 2><<<batch / 2, 64>>>(in, out, N, batch); });
  timeit("2 problems/wave, 16 B/lane, depth 2", [&] { recN<2, 4, 2><<<batch / 2, 64>>>(in, out, N, batch); });
  timeit("4 problems/wave, 4 B/lane, depth 1", [&] { recN<4, 1, 1><<<batch / 4, 64>>>(in, out, N, batch); });
  timeit("4 problems/wave, 4 B/lane, depth 2", [&] { recN<4, 1, 2><<<batch / 4, 64>>>(in, out, N, batch); });
  timeit("4 problems/wave, 16 B/lane, depth 1", [&] { recN<4, 4, 1><<<batch / 4, 64>>>(in, out, N, batch); });
  timeit("4 problems/wave, 16 B/lane, depth 2", [&] { recN<4, 4, 2><<<batch / 4, 64>>>(in, out, N, batch); });
  timeit("4 problems/wave, 16 B/lane, depth 4", [&] { recN<4, 4, 4><<<batch / 4, 64>>>(in, out, N, batch); });
  timeit("4 problems/wave, 8 B/lane, depth 2", [&] { recN<4, 2, 2><<<batch / 4, 64>>>(in, out, N, batch); });
  return 0;
}
