// tools/ldsbench.hip -- what a wave instruction costs on the LDS pipe of a gfx950 CU (shared by the CU's four SIMDs), and
// what the alternatives to an LDS operand broadcast cost on the VALU: DPP row_newbcast (64-bit, gfx90a+), v_readlane.
// Every block is one wave; 4096 blocks = 4 waves per SIMD on 256 CUs.  "ns per instr per CU" = wall time / (instructions
// issued by the 16 waves of a CU) -- for an LDS-bound loop that is the pipe's time per wave instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ double bcast16(double v) {   // lane (CTRL & 15) of this lane's row of 16 lanes: two 32-bit DPP moves
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x150 + CTRL, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x150 + CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// acc += bcast(v) * w in ONE instruction: v_fmac_f64 with a DPP source (64-bit DPP knows row_newbcast only)
#define FMAC_BCAST(acc, v, w, N) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(w))
#define MOV_BCAST(dst, v, N) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "=v"(dst) : "v"(v))

// eight plain ds_read_b64 (inline asm: the compiler would pair them into ds_read2*_b64), then the eight adds
#define RD64(a0, a1, a2, a3, a4, a5, a6, a7, byteaddr) do { \
  double t0, t1, t2, t3, t4, t5, t6, t7; const unsigned ad = (unsigned)(byteaddr) + lds_base; \
  asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:1032\n ds_read_b64 %2, %8 offset:2064\n ds_read_b64 %3, %8 offset:3096\n" \
               "ds_read_b64 %4, %8 offset:4128\n ds_read_b64 %5, %8 offset:5160\n ds_read_b64 %6, %8 offset:6192\n ds_read_b64 %7, %8 offset:7224\n" \
               "s_waitcnt lgkmcnt(0)" : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(ad) : "memory"); \
  a0 += t0; a1 += t1; a2 += t2; a3 += t3; a4 += t4; a5 += t5; a6 += t6; a7 += t7; } while (0)
template <int MODE>
__global__ __launch_bounds__(64) void probe(double* out, int iters, int stride) {
  __shared__ double lds[1152];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) double*)lds;
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = i * 1e-3;
  __syncthreads();
  double a0 = lane * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const int base = (MODE == 1 || MODE == 3) ? 0 : lane * stride;   // broadcast: every lane the same address
  double* const L = lds;   // (indexing the __shared__ array itself keeps the accesses ds_* -- a volatile generic pointer turns them into flat loads)
  for (int i = 0; i < iters; ++i) {
    const int o = (i & 15) * 2;
    asm volatile("" ::: "memory");
    if (MODE == 0 || MODE == 1) {           // 8 ds_read_b64
      RD64(a0, a1, a2, a3, a4, a5, a6, a7, (base + o) * 8);
    } else if (MODE == 2 || MODE == 3) {    // 8 ds_read_b128
      const f64x2* V = (const f64x2*)lds;
      const int b2 = (MODE == 3) ? 0 : lane * stride;
      f64x2 v0 = V[b2 + o], v1 = V[b2 + o + 48], v2 = V[b2 + o + 96], v3 = V[b2 + o + 144], v4 = V[b2 + o + 192], v5 = V[b2 + o + 240],
            v6 = V[b2 + o + 288], v7 = V[b2 + o + 336];
      a0 += v0[0] + v0[1]; a1 += v1[0] + v1[1]; a2 += v2[0] + v2[1]; a3 += v3[0] + v3[1];
      a4 += v4[0] + v4[1]; a5 += v5[0] + v5[1]; a6 += v6[0] + v6[1]; a7 += v7[0] + v7[1];
    } else if (MODE == 4) {                 // 8 ds_read_b64, 16 lanes active
      if (lane >= 48) {
        RD64(a0, a1, a2, a3, a4, a5, a6, a7, (base + o) * 8);
      }
    } else if (MODE == 5) {                 // 8 ds_read_b64, 32 lanes active
      if (lane >= 32) {
        RD64(a0, a1, a2, a3, a4, a5, a6, a7, (base + o) * 8);
      }
    } else if (MODE == 6) {                 // 8 ds_write_b64
      L[base + o] = a0; L[base + o + 129] = a1; L[base + o + 258] = a2; L[base + o + 387] = a3;
      L[base + o + 516] = a4; L[base + o + 645] = a5; L[base + o + 774] = a6; L[base + o + 903] = a7;
    } else if (MODE == 7) {                 // 8 x (row_newbcast operand + fma): the DPP broadcast
      a0 = __builtin_fma(bcast16<0>(a7), 1.0000001, a0); a1 = __builtin_fma(bcast16<1>(a7), 1.0000001, a1);
      a2 = __builtin_fma(bcast16<2>(a7), 1.0000001, a2); a3 = __builtin_fma(bcast16<3>(a7), 1.0000001, a3);
      a4 = __builtin_fma(bcast16<4>(a7), 1.0000001, a4); a5 = __builtin_fma(bcast16<5>(a7), 1.0000001, a5);
      a6 = __builtin_fma(bcast16<6>(a7), 1.0000001, a6); a0 = __builtin_fma(bcast16<7>(a7), 1.0000001, a0);
    } else if (MODE == 10) {                // 8 x v_fmac_f64_dpp row_newbcast
      const double w = 1.0000001;
      FMAC_BCAST(a0, a7, w, 0); FMAC_BCAST(a1, a7, w, 1); FMAC_BCAST(a2, a7, w, 2); FMAC_BCAST(a3, a7, w, 3);
      FMAC_BCAST(a4, a7, w, 4); FMAC_BCAST(a5, a7, w, 5); FMAC_BCAST(a6, a7, w, 6); FMAC_BCAST(a0, a7, w, 7);
    } else if (MODE == 11) {                // 8 x (v_mov_b64_dpp row_newbcast + fma)
      double t0, t1, t2, t3, t4, t5, t6, t7;
      MOV_BCAST(t0, a7, 0); MOV_BCAST(t1, a7, 1); MOV_BCAST(t2, a7, 2); MOV_BCAST(t3, a7, 3);
      MOV_BCAST(t4, a7, 4); MOV_BCAST(t5, a7, 5); MOV_BCAST(t6, a7, 6); MOV_BCAST(t7, a7, 7);
      a0 = __builtin_fma(t0, 1.0000001, a0); a1 = __builtin_fma(t1, 1.0000001, a1); a2 = __builtin_fma(t2, 1.0000001, a2);
      a3 = __builtin_fma(t3, 1.0000001, a3); a4 = __builtin_fma(t4, 1.0000001, a4); a5 = __builtin_fma(t5, 1.0000001, a5);
      a6 = __builtin_fma(t6, 1.0000001, a6); a0 = __builtin_fma(t7, 1.0000001, a0);
    } else if (MODE == 8) {                 // 8 x (readlane pair + fma with the scalar)
      a0 = __builtin_fma(__shfl(a7, 0, 64), 1.0000001, a0); a1 = __builtin_fma(__shfl(a7, 1, 64), 1.0000001, a1);
      a2 = __builtin_fma(__shfl(a7, 2, 64), 1.0000001, a2); a3 = __builtin_fma(__shfl(a7, 3, 64), 1.0000001, a3);
      a4 = __builtin_fma(__shfl(a7, 4, 64), 1.0000001, a4); a5 = __builtin_fma(__shfl(a7, 5, 64), 1.0000001, a5);
      a6 = __builtin_fma(__shfl(a7, 6, 64), 1.0000001, a6); a0 = __builtin_fma(__shfl(a7, 7, 64), 1.0000001, a0);
    } else if (MODE == 9) {                 // 8 x (readlane pair via the builtin + fma with the scalar)
#define RL(v, l) __builtin_bit_cast(double, ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(__builtin_bit_cast(unsigned long long, v) >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)__builtin_bit_cast(unsigned long long, v), l))
      a0 = __builtin_fma(RL(a7, 0), 1.0000001, a0); a1 = __builtin_fma(RL(a7, 1), 1.0000001, a1);
      a2 = __builtin_fma(RL(a7, 2), 1.0000001, a2); a3 = __builtin_fma(RL(a7, 3), 1.0000001, a3);
      a4 = __builtin_fma(RL(a7, 4), 1.0000001, a4); a5 = __builtin_fma(RL(a7, 5), 1.0000001, a5);
      a6 = __builtin_fma(RL(a7, 6), 1.0000001, a6); a0 = __builtin_fma(RL(a7, 7), 1.0000001, a0);
    }
  }
  out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void check_dpp(double* out) {   // acc = 100 + v[lane 5 of my row of 16] * w, v = lane, w = 2
  double acc = 100.0, v = (double)threadIdx.x, w = 2.0;
  FMAC_BCAST(acc, v, w, 5);
  out[threadIdx.x] = acc;
}
template <int MODE>
void run(const char* name, int blocks, int stride = 1) {
  double* out; hipMalloc(&out, (size_t)blocks * 64 * 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<blocks, 64>>>(out, 100, stride);
  hipEventRecord(e0); probe<MODE><<<blocks, 64>>>(out, iters, stride); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_cu = blocks / 256.0;
  printf("%-52s blocks=%5d  wall ns/iter=%8.2f  ns per wave-instr per CU=%6.3f\n", name, blocks, ms * 1e6 / iters,
         ms * 1e6 / iters / 8.0 / waves_per_cu);
  hipFree(out);
}
int main() {
  {
    double* o; hipMalloc(&o, 64 * 8); check_dpp<<<1, 64>>>(o); double h[64]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0; for (int l = 0; l < 64; ++l) bad += h[l] != 100.0 + 2.0 * ((l & ~15) + 5);
    printf("v_fmac_f64_dpp row_newbcast:5 -> lane 0 %.1f lane 17 %.1f lane 63 %.1f : %s\n", h[0], h[17], h[63], bad ? "WRONG" : "as expected");
  }
  for (int blocks : {1024, 4096}) {
    run<0>("8 ds_read_b64, lane-consecutive", blocks);
    run<1>("8 ds_read_b64, one address (broadcast)", blocks);
    run<2>("8 ds_read_b128, lane-consecutive", blocks);
    run<3>("8 ds_read_b128, one address (broadcast)", blocks);
    run<0>("8 ds_read_b64, lane stride 2 doubles", blocks, 2);
    run<4>("8 ds_read_b64, 16 lanes active", blocks);
    run<5>("8 ds_read_b64, 32 lanes active", blocks);
    run<6>("4 ds_write2_b64 (8 doubles per lane)", blocks);
    run<7>("8 x (2 v_mov_b32_dpp row_newbcast + fma)", blocks);
    run<10>("8 x v_fmac_f64_dpp row_newbcast", blocks);
    run<11>("8 x (v_mov_b64_dpp row_newbcast + fma)", blocks);
    run<8>("8 x (__shfl const lane + fma)", blocks);
    run<9>("8 x (2 v_readlane + fma)", blocks);
  }
  return 0;
}
