// tools/membench.hip -- HBM streaming probes on MI355X: what bandwidth can the backward sweep's access
// pattern reach with NO arithmetic?  (Practical roof next to the 8 TB/s datasheet number.)
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o /tmp/membench && /tmp/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void copy16(const double2* __restrict__ src, double2* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void read16(const double2* __restrict__ src, double* sink, size_t n) {
  double acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = src[i]; acc += v.x + v.y; }
  if (acc == 1.2345) sink[0] = acc;
}
__global__ void write16nt(double2* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    __builtin_nontemporal_store(1.0, &dst[i].x); __builtin_nontemporal_store(2.0, &dst[i].y);
  }
}
__global__ void write8(double* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = 1.0;
}
__global__ void write8nt(double* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(1.0, &dst[i]);
}
__global__ void read8(const double* __restrict__ src, double* sink, size_t n) {
  double acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
  if (acc == 1.2345) sink[0] = acc;
}
__global__ void write16(double2* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_double2(1.0, 2.0);
}

// wave-per-problem record streaming: per step read RIN doubles, write ROUT doubles, 8 B per lane,
// software prefetch DEPTH steps ahead.  LAY = 0: [b][k], 1: [k][b].
template <int DEPTH, int LAY, int W>
__global__ __launch_bounds__(64) void stream_records(const double* __restrict__ in, double* __restrict__ out, int N, int batch) {
  constexpr int RIN = 428, ROUT = 208;
  const int lane = threadIdx.x, b = blockIdx.x;
  const size_t in_bs = LAY ? RIN : (size_t)N * RIN, in_ks = LAY ? (size_t)batch * RIN : RIN;
  const size_t out_bs = LAY ? ROUT : (size_t)N * ROUT, out_ks = LAY ? (size_t)batch * ROUT : ROUT;
  const double* ip = in + b * in_bs;
  double* op = out + b * out_bs;
  double buf[DEPTH + 1][7];
  auto load = [&](double* r, int k) {
    const double* rec = ip + (size_t)k * in_ks;
#pragma unroll
    for (int c = 0; c < 6; ++c) r[c] = rec[c * 64 + lane];
    r[6] = rec[384 + (lane < 44 ? lane : 43)];
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(buf[d], N - 1 - d);
  double acc = 0.0;
  for (int k = N - 1; k >= 0; --k) {
    int kp = k - DEPTH; if (kp < 0) kp = 0;
    load(buf[DEPTH], kp);
    double s = 0;
#pragma unroll
    for (int c = 0; c < 7; ++c) s += buf[0][c];
    acc += s;
    double* o = op + (size_t)k * out_ks;
#pragma unroll
    for (int c = 0; c < 3; ++c) { if (W) __builtin_nontemporal_store(s + c, &o[c * 64 + lane]); else o[c * 64 + lane] = s + c; }
    if (W) __builtin_nontemporal_store(acc, &o[192 + (lane & 15)]); else o[192 + (lane & 15)] = acc;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int c = 0; c < 7; ++c) buf[d][c] = buf[d + 1][c];
  }
}

// forward-sweep pattern: per step read two fully-used records (204 + 208 doubles), write 28 doubles
template <int DEPTH>
__global__ __launch_bounds__(64) void stream_fwd(const double* __restrict__ dyn, const double* __restrict__ outr, double* __restrict__ xuy, int N, int batch) {
  const int lane = threadIdx.x, b = blockIdx.x;
  double buf[DEPTH + 1][7];
  auto load = [&](double* r, int k) {
    const double* d = dyn + ((size_t)k * batch + b) * 204;
    const double* o = outr + ((size_t)k * batch + b) * 208;
#pragma unroll
    for (int c = 0; c < 3; ++c) r[c] = d[c * 64 + lane];
#pragma unroll
    for (int c = 0; c < 3; ++c) r[3 + c] = o[c * 64 + lane];
    r[6] = o[192 + (lane & 15)] + d[192 + (lane < 12 ? lane : 11)];
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(buf[d], d);
  double acc = 0.0;
  for (int k = 0; k < N; ++k) {
    int kp = k + DEPTH; if (kp >= N) kp = N - 1;
    load(buf[DEPTH], kp);
    double s = 0;
#pragma unroll
    for (int c = 0; c < 7; ++c) s += buf[0][c];
    acc += s;
    xuy[((size_t)k * batch + b) * 28 + (lane % 28)] = acc;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int c = 0; c < 7; ++c) buf[d][c] = buf[d + 1][c];
  }
}


// generic record streaming, [k][b] layout: per step read RIN doubles and write ROUT doubles (8 B per lane, unit
// stride), prefetch one step ahead.  Used to price symmetric-packed record layouts before building them.
template <int RIN, int ROUT>
__global__ __launch_bounds__(64) void stream_generic(const double* __restrict__ in, double* __restrict__ out, int N, int batch) {
  constexpr int LI = (RIN + 63) / 64, LO = (ROUT + 63) / 64;
  const int lane = threadIdx.x, b = blockIdx.x;
  double cur[LI], nxt[LI];
  auto load = [&](double* r, int k) {
    const double* rec = in + ((size_t)k * batch + b) * RIN;
#pragma unroll
    for (int c = 0; c < LI; ++c) { int e = c * 64 + lane; r[c] = rec[e < RIN ? e : RIN - 1]; }
  };
  load(cur, N - 1);
  double acc = 0.0;
  for (int k = N - 1; k >= 0; --k) {
    load(nxt, k > 0 ? k - 1 : 0);
    double s = 0;
#pragma unroll
    for (int c = 0; c < LI; ++c) s += cur[c];
    acc += s;
    double* o = out + ((size_t)k * batch + b) * ROUT;
#pragma unroll
    for (int c = 0; c < LO; ++c) { int e = c * 64 + lane; o[e < ROUT ? e : ROUT - 1] = acc + c; }
#pragma unroll
    for (int c = 0; c < LI; ++c) cur[c] = nxt[c];
  }
}

// the same with 16 bytes per lane (double2 loads and stores; RIN, ROUT multiples of 2)
template <int RIN, int ROUT>
__global__ __launch_bounds__(64) void stream_generic16(const double* __restrict__ in, double* __restrict__ out, int N, int batch) {
  constexpr int PI = RIN / 2, PO = ROUT / 2;
  constexpr int LI = (PI + 63) / 64, LO = (PO + 63) / 64;
  const int lane = threadIdx.x, b = blockIdx.x;
  double2 cur[LI], nxt[LI];
  auto load = [&](double2* r, int k) {
    const double2* rec = (const double2*)(in + ((size_t)k * batch + b) * RIN);
#pragma unroll
    for (int c = 0; c < LI; ++c) { int e = c * 64 + lane; r[c] = rec[e < PI ? e : PI - 1]; }
  };
  load(cur, N - 1);
  double acc = 0.0;
  for (int k = N - 1; k >= 0; --k) {
    load(nxt, k > 0 ? k - 1 : 0);
    double s = 0;
#pragma unroll
    for (int c = 0; c < LI; ++c) s += cur[c].x + cur[c].y;
    acc += s;
    double2* o = (double2*)(out + ((size_t)k * batch + b) * ROUT);
#pragma unroll
    for (int c = 0; c < LO; ++c) { int e = c * 64 + lane; o[e < PO ? e : PO - 1] = make_double2(acc + c, acc); }
#pragma unroll
    for (int c = 0; c < LI; ++c) cur[c] = nxt[c];
  }
}

// generic record streaming with (MAP) the XCD-aware block -> problem mapping and WPB waves (adjacent problems) per block
template <int RIN, int ROUT, int MAP, int WPB>
__global__ __launch_bounds__(64 * WPB) void stream_mapped(const double* __restrict__ in, double* __restrict__ out, int N, int batch) {
  constexpr int LI = (RIN + 63) / 64, LO = (ROUT + 63) / 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nblk = batch / WPB, chunk = nblk / 8;
  const int blk = MAP ? (int)((blockIdx.x & 7) * chunk + (blockIdx.x >> 3)) : (int)blockIdx.x;
  const int b = blk * WPB + wv;
  double cur[LI], nxt[LI];
  auto load = [&](double* r, int k) {
    const double* rec = in + ((size_t)k * batch + b) * RIN;
#pragma unroll
    for (int c = 0; c < LI; ++c) { int e = c * 64 + lane; r[c] = rec[e < RIN ? e : RIN - 1]; }
  };
  load(cur, N - 1);
  double acc = 0.0;
  for (int k = N - 1; k >= 0; --k) {
    load(nxt, k > 0 ? k - 1 : 0);
    double s = 0;
#pragma unroll
    for (int c = 0; c < LI; ++c) s += cur[c];
    acc += s;
    double* o = out + ((size_t)k * batch + b) * ROUT;
#pragma unroll
    for (int c = 0; c < LO; ++c) { int e = c * 64 + lane; o[e < ROUT ? e : ROUT - 1] = acc + c; }
#pragma unroll
    for (int c = 0; c < LI; ++c) cur[c] = nxt[c];
  }
}

// P adjacent problems per wave ([k][b] slabs: their records are one contiguous run of P * RIN doubles), V doubles per lane
// and access (8 or 16 bytes), records requested DEPTH knot points ahead, XCD-aware block -> problem mapping.  Prices the
// "two problems per wave, two doubles per lane" form of the fp64 backward sweep before it is built.
template <int RIN, int ROUT, int P, int V, int DEPTH>
__global__ __launch_bounds__(64) void stream_multi(const double* __restrict__ in, double* __restrict__ out, int N, int batch) {
  constexpr int EI = P * RIN / V, EO = P * ROUT / V;            // V-wide elements per knot point
  constexpr int LI = (EI + 63) / 64, LO = (EO + 63) / 64;
  typedef double vec __attribute__((ext_vector_type(V)));
  const int lane = threadIdx.x;
  const int nblk = batch / P, chunk = nblk / 8;
  const int blk = (int)((blockIdx.x & 7) * chunk + (blockIdx.x >> 3));
  const size_t b0 = (size_t)blk * P;
  vec ring[DEPTH][LI];
  auto load = [&](vec* r, int k) {
    const vec* rec = (const vec*)(in + ((size_t)k * batch + b0) * RIN);
#pragma unroll
    for (int c = 0; c < LI; ++c) { int e = c * 64 + lane; r[c] = rec[e < EI ? e : EI - 1]; }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(ring[d], N - 1 - d > 0 ? N - 1 - d : 0);
  double acc = 0.0;
  for (int k0 = N - 1; k0 >= 0; k0 -= DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int k = k0 - d;
      if (k < 0) break;
      double s = 0;
#pragma unroll
      for (int c = 0; c < LI; ++c)
#pragma unroll
        for (int v = 0; v < V; ++v) s += ring[d][c][v];
      acc += s;
      load(ring[d], k - DEPTH > 0 ? k - DEPTH : 0);
      vec* o = (vec*)(out + ((size_t)k * batch + b0) * ROUT);
#pragma unroll
      for (int c = 0; c < LO; ++c) {
        int e = c * 64 + lane;
        vec w;
#pragma unroll
        for (int v = 0; v < V; ++v) w[v] = acc + c + v;
        o[e < EO ? e : EO - 1] = w;
      }
    }
  }
}

int main(int argc, char** argv) {
  const bool only_multi = argc > 1;   // any argument: just the problems-per-wave table
  const int N = 256, batch = 4096;
  const size_t in_n = (size_t)batch * N * 428, out_n = (size_t)batch * N * 208;
  double *in, *out, *sink;
  CK(hipMalloc(&in, in_n * 8)); CK(hipMalloc(&out, out_n * 8)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(in, 0, in_n * 8)); CK(hipMemset(out, 0, out_n * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double bytes, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0); for (int i = 0; i < 10; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-44s %8.3f ms  %8.1f GB/s\n", name, ms, bytes / ms / 1e6);
  };
  if (only_multi) {
    auto gen = [&](const char* name, auto launch) { timeit(name, (double)batch * N * (364 + 144) * 8, launch); };
    printf("# backward-sweep record set (364 r + 144 w doubles per knot point and problem, 4.26 GB), [k][b] slabs, XCD-aware mapping:\n"
           "# problems per wave x bytes per lane x prefetch depth (no arithmetic)\n");
    for (int rep = 0; rep < 2; ++rep) {
      gen("1 problem/wave,  8 B/lane, depth 1", [&] { stream_multi<364, 144, 1, 1, 1><<<batch, 64>>>(in, out, N, batch); });
      gen("1 problem/wave,  8 B/lane, depth 2", [&] { stream_multi<364, 144, 1, 1, 2><<<batch, 64>>>(in, out, N, batch); });
      gen("1 problem/wave, 16 B/lane, depth 1", [&] { stream_multi<364, 144, 1, 2, 1><<<batch, 64>>>(in, out, N, batch); });
      gen("1 problem/wave, 16 B/lane, depth 2", [&] { stream_multi<364, 144, 1, 2, 2><<<batch, 64>>>(in, out, N, batch); });
      gen("2 problems/wave,  8 B/lane, depth 1", [&] { stream_multi<364, 144, 2, 1, 1><<<batch / 2, 64>>>(in, out, N, batch); });
      gen("2 problems/wave,  8 B/lane, depth 2", [&] { stream_multi<364, 144, 2, 1, 2><<<batch / 2, 64>>>(in, out, N, batch); });
      gen("2 problems/wave, 16 B/lane, depth 1", [&] { stream_multi<364, 144, 2, 2, 1><<<batch / 2, 64>>>(in, out, N, batch); });
      gen("2 problems/wave, 16 B/lane, depth 2", [&] { stream_multi<364, 144, 2, 2, 2><<<batch / 2, 64>>>(in, out, N, batch); });
      gen("2 problems/wave, 16 B/lane, depth 3", [&] { stream_multi<364, 144, 2, 2, 3><<<batch / 2, 64>>>(in, out, N, batch); });
      gen("4 problems/wave, 16 B/lane, depth 1", [&] { stream_multi<364, 144, 4, 2, 1><<<batch / 4, 64>>>(in, out, N, batch); });
      gen("4 problems/wave, 16 B/lane, depth 2", [&] { stream_multi<364, 144, 4, 2, 2><<<batch / 4, 64>>>(in, out, N, batch); });
      gen("today: stream_mapped 1/wave 8 B depth 1", [&] { stream_mapped<364, 144, 1, 1><<<batch, 64>>>(in, out, N, batch); });
    }
    return 0;
  }
  const size_t n16 = in_n / 2;
  timeit("copy16 in->in2 (full 3.59 GB r + w)", 2.0 * out_n * 8, [&] { copy16<<<256 * 8, 256>>>((const double2*)in, (double2*)out, out_n / 2); });
  timeit("read-only 16B/lane (3.59 GB)", 1.0 * in_n * 8, [&] { read16<<<256 * 8, 256>>>((const double2*)in, sink, n16); });
  timeit("write-only 16B/lane (1.74 GB)", 1.0 * out_n * 8, [&] { write16<<<256 * 8, 256>>>((double2*)out, out_n / 2); });
  timeit("write-only 16B/lane nontemporal", 1.0 * out_n * 8, [&] { write16nt<<<256 * 8, 256>>>((double2*)out, out_n / 2); });
  timeit("write-only 8B/lane", 1.0 * out_n * 8, [&] { write8<<<256 * 8, 256>>>(out, out_n); });
  timeit("write-only 8B/lane nontemporal", 1.0 * out_n * 8, [&] { write8nt<<<256 * 8, 256>>>(out, out_n); });
  timeit("write-only 16B/lane, 64K blocks", 1.0 * out_n * 8, [&] { write16<<<65536, 256>>>((double2*)out, out_n / 2); });
  timeit("read-only 8B/lane (3.59 GB)", 1.0 * in_n * 8, [&] { read8<<<256 * 8, 256>>>(in, sink, in_n); });
  timeit("read-only 16B/lane, 64K blocks", 1.0 * in_n * 8, [&] { read16<<<65536, 256>>>((const double2*)in, sink, n16); });
  const double rb = (double)in_n * 8 + (double)out_n * 8;
  timeit("records [b][k] depth1", rb, [&] { stream_records<1, 0, 0><<<batch, 64>>>(in, out, N, batch); });
  timeit("records [k][b] depth1", rb, [&] { stream_records<1, 1, 0><<<batch, 64>>>(in, out, N, batch); });
  timeit("records [k][b] depth2", rb, [&] { stream_records<2, 1, 0><<<batch, 64>>>(in, out, N, batch); });
  timeit("records [k][b] depth3", rb, [&] { stream_records<3, 1, 0><<<batch, 64>>>(in, out, N, batch); });
  timeit("records [k][b] depth1 nt stores", rb, [&] { stream_records<1, 1, 1><<<batch, 64>>>(in, out, N, batch); });
  timeit("records [b][k] depth1 nt stores", rb, [&] { stream_records<1, 0, 1><<<batch, 64>>>(in, out, N, batch); });
  timeit("records [b][k] depth3", rb, [&] { stream_records<3, 0, 0><<<batch, 64>>>(in, out, N, batch); });
  {
    const double fb = (double)batch * N * (204 + 208 + 28) * 8;
    timeit("forward pattern (412 r + 28 w) depth1", fb, [&] { stream_fwd<1><<<batch, 64>>>(in, in + (size_t)batch * N * 204, out, N, batch); });
    timeit("forward pattern (412 r + 28 w) depth3", fb, [&] { stream_fwd<3><<<batch, 64>>>(in, in + (size_t)batch * N * 204, out, N, batch); });
  }
  {
    auto gen = [&](const char* name, int rin, int rout, auto launch) {
      timeit(name, (double)batch * N * (rin + rout) * 8, launch);
    };
    gen("generic 428 r + 208 w (today's backward)", 428, 208, [&] { stream_generic<428, 208><<<batch, 64>>>(in, out, N, batch); });
    gen("generic 428 r + 144 w (P packed)", 428, 144, [&] { stream_generic<428, 144><<<batch, 64>>>(in, out, N, batch); });
    gen("generic 364 r + 144 w (Q and P packed)", 364, 144, [&] { stream_generic<364, 144><<<batch, 64>>>(in, out, N, batch); });
    gen("generic 360 r + 144 w (Q, R and P packed)", 360, 144, [&] { stream_generic<360, 144><<<batch, 64>>>(in, out, N, batch); });
    gen("generic 412 r + 28 w (today's forward)", 412, 28, [&] { stream_generic<412, 28><<<batch, 64>>>(in, out, N, batch); });
    gen("generic 348 r + 28 w (forward, P packed)", 348, 28, [&] { stream_generic<348, 28><<<batch, 64>>>(in, out, N, batch); });
    gen("16 B/lane: 364 r + 144 w", 364, 144, [&] { stream_generic16<364, 144><<<batch, 64>>>(in, out, N, batch); });
    gen("16 B/lane: 348 r + 28 w", 348, 28, [&] { stream_generic16<348, 28><<<batch, 64>>>(in, out, N, batch); });
    gen("8 B/lane again: 364 r + 144 w", 364, 144, [&] { stream_generic<364, 144><<<batch, 64>>>(in, out, N, batch); });
    gen("8 B/lane again: 348 r + 28 w", 348, 28, [&] { stream_generic<348, 28><<<batch, 64>>>(in, out, N, batch); });
    gen("364 r + 144 w, plain, 1 wave/block", 364, 144, [&] { stream_mapped<364, 144, 0, 1><<<batch, 64>>>(in, out, N, batch); });
    gen("364 r + 144 w, XCD map, 1 wave/block", 364, 144, [&] { stream_mapped<364, 144, 1, 1><<<batch, 64>>>(in, out, N, batch); });
    gen("364 r + 144 w, XCD map, 2 waves/block", 364, 144, [&] { stream_mapped<364, 144, 1, 2><<<batch / 2, 128>>>(in, out, N, batch); });
    gen("364 r + 144 w, XCD map, 4 waves/block", 364, 144, [&] { stream_mapped<364, 144, 1, 4><<<batch / 4, 256>>>(in, out, N, batch); });
    gen("364 r + 144 w, plain, 4 waves/block", 364, 144, [&] { stream_mapped<364, 144, 0, 4><<<batch / 4, 256>>>(in, out, N, batch); });
    gen("364 r + 144 w, XCD map, 8 waves/block", 364, 144, [&] { stream_mapped<364, 144, 1, 8><<<batch / 8, 512>>>(in, out, N, batch); });
  }
  return 0;
}
