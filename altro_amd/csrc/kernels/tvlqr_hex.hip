// tvlqr_hex.hip -- plan LANE's backward sweep with SIXTEEN (n <= 2: EIGHT) lanes per problem, fp64, the exact (uncontracted)
// flavour.  Included by tvlqr_lane.hip inside its contract(off) region.
//
// Why (profiles/r04zf_quad_pmc.txt, r05a_hex_ab.txt).  At the per-GPU batch of BASELINE.json configs[2] / [3]
// (8192 problems) the four-lanes-per-problem sweep is bound twice over:
//   * by its chain -- 668 instructions per knot point issued in order by ONE wave per SIMD on half the SIMDs (4700 cycles);
//   * by the vector-memory path -- a lane of a quad re-loads what its neighbours load (45 loads of 8 bytes per lane and knot point
//     for 62 distinct elements: 2.9 x the bytes), and a CU moves 4 lane-dwords per clock through that path whatever the lanes hold
//     (measured here: a wave-wide 8-byte load costs ~34 CU cycles), i.e. 3.4 TB/s at 2.9 x replication -- what the kernel ran at.
// Here one wave owns 4 (8) problems:
//   * every input element is loaded ONCE, all 64 lanes useful (62 rows x 4 problems = 4 loads per lane and knot point), into a
//     wave-private LDS stage two knot points ahead; operands are read from LDS (128 B/clk per CU), replicated there for free;
//   * one OUTPUT ELEMENT per lane in every product of tvlqr.cpp:125-191: ~220 VALU instructions per knot point and wave instead
//     of ~560, four times the waves (two per SIMD at 8192 problems);
//   * what a later product needs from another lane goes through the problem's LDS scratch record (four exchanges per knot point),
//     a workgroup is one wave, so a "barrier" is a wait on the LDS counter and nothing else;
//   * the OUT record K | d | P | p is assembled in LDS and stored once, whole rows.
// Every element is still its own index-ordered dot product, no FMA contraction, IEEE division / square root: results are
// bit-identical to lane_backward_kernel and to the CPU path (tests/test_gpu_hex.py).  n <= 4, m <= 3.
//
// Measured and dropped (profiles/r05a_hex_ab.txt): the FORWARD sweep in the same shape with per-lane loads straight from the SoA
// records (x broadcast by v_mov_b64_dpp row_newbcast) was bit-identical and twice as SLOW as quad_forward_kernel (0.084 against
// 0.046 ms at 8192 bicycles): 14 vector-memory instructions per wave of 4 problems, 3.3 x the bytes through the 4-lane-dwords-per-
// clock path.  quad_forward_kernel already streams 4.8 TB/s.
#pragma once

namespace altro_hip {

constexpr uint32_t HEX_OOB = 0x80000000u;   // beyond the 2 GiB buffer window (no row offset added to it wraps): loads give 0, stores are dropped

template <int n, int m>
struct HexDims {
  using D = LaneDims<n, m>;
  static constexpr int LPP = n <= 2 ? 8 : 16;   // lanes per problem
  static constexpr int PPW = 64 / LPP;          // problems per wave
  static constexpr int RPI = 64 / PPW;          // record rows one wave-wide load / store instruction covers (= LPP)
  static constexpr int NLD = (D::E_IN + RPI - 1) / RPI, NST = (D::E_OUT + RPI - 1) / RPI;
  // the problem's scratch record in LDS (doubles).  [0, E_OUT) is the OUT record K | d | P | p in the reference order: what the
  // sweep stores for this knot point and, for P and p, what the next one reads as P', p'.
  static constexpr int oK = D::O_K, od = D::O_d, oP = D::O_P, op = D::O_p;
  static constexpr int oPT = D::E_OUT;             // P' again, row-major (row i contiguous)
  static constexpr int oT1 = oPT + n * n;          // Qxx_tmp = A^T P', row-major
  static constexpr int oT2 = oT1 + n * n;          // Qux_tmp = B^T P', row-major (m rows of n)
  static constexpr int ot = oT2 + m * n;           // Qx_tmp = p' + P' f
  static constexpr int oQuu = ot + n;              // column-major m x m
  static constexpr int oQux = oQuu + m * m;        // column-major m x n
  static constexpr int oQx = oQux + m * n;
  static constexpr int oQu = oQx + n;
  static constexpr int oZero = oQu + m;            // constant 0.0
  static constexpr int oTrash = oZero + 1;         // where lanes that own no element of a product write
  // record strides: == 8 (mod 32) doubles, so the problems of a wave start 16 banks apart (a stride of 64 or 96 doubles -- what
  // rounding up to even gave for (4, 2) -- put all four on the same banks: 0.126 -> ... ms at 8192 bicycles)
  static constexpr int pad8(int x) { return ((x + 31 - 8) / 32) * 32 + 8; }
  static constexpr int SC = pad8(oTrash + 1);
  static constexpr int SI = pad8(D::E_IN);                        // one problem's input record in a stage
  static constexpr int STAGES = 2;
  static constexpr int LDS_DOUBLES = PPW * (STAGES * SI + SC);
  // S2-B: lanes [0, n): Qux(:, j) ; lane n: Qu ; lanes [n+1, n+1+m): Quu(:, c) ; then ceil(n / m) lanes: Qx, m entries each
  static constexpr int NQX = (n + m - 1) / m;
  static_assert(n * n <= LPP && m * n + n <= LPP && n + 1 + m + NQX <= LPP && n <= 4 && m <= 3, "shape does not fit the lane roles");
};

// add + sum_{k < L} a[k] b[k], the sum taken as the lane kernel takes it: s = 0; s += a_k b_k in index order; add + s
template <int L>
__device__ __forceinline__ double hex_dot(const double* a, const double* b, double add) {
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < L; ++k) s += a[k] * b[k];
  return add + s;
}

template <int n, int m>
__global__ __launch_bounds__(64) void hex_backward_kernel(LaneArgs<double> a) {
  using D = LaneDims<n, m>;
  using H = HexDims<n, m>;
  constexpr int LPP = H::LPP, PPW = H::PPW, RPI = H::RPI, SC = H::SC, SI = H::SI;
  __shared__ double lds[H::LDS_DOUBLES];
  __shared__ int state_s[PPW];   // 2: alive ; 1: failed at this knot point (K, d stay unsolved and are stored, tvlqr.cpp:162-164) ; 0: stopped
  const int nwv = (a.batch + PPW - 1) / PPW, chk = (nwv + 7) / 8;
  const int wv = (int)((blockIdx.x & 7) * chk + (blockIdx.x >> 3));   // contiguous runs of waves per XCD, like the lane kernels
  if (wv >= nwv) return;
  const int tid = threadIdx.x;
  const int64_t B = a.batch;
  const int N = a.N;
  const int64_t b0 = (int64_t)wv * PPW;
  const uint32_t rowB = (uint32_t)B * 8u;
  // ---- compute roles: problem cq of the wave, lane r of its LPP
  const int cq = tid / LPP, r = tid % LPP;
  const int64_t cb = b0 + cq;
  const bool cvalid = cb < B && !(a.active && !a.active[cb < B ? cb : 0]);
  const double reg = (a.reg_pp && cb < B) ? a.reg_pp[cb] : a.reg;
  double* const scr = lds + PPW * H::STAGES * SI + cq * SC;
  const int ei = r % n, ej = (r / n) % n;            // element (i, j) of the n x n objects (lanes r < n n)
  const bool isE = r < n * n;
  // S1-A: T1(i, j) = 0 + sum_k A(k, i) P'(k, j)                                                    (tvlqr.cpp:135)
  const int s1a_a = D::O_A + n * ei, s1a_b = H::oP + n * ej, s1a_o = isE ? H::oT1 + ei * n + ej : H::oTrash;
  // S1-B: lanes < m n: T2(i, j) = 0 + sum_k B(k, i) P'(k, j) ; the next n lanes: t(i) = p'(i) + sum_k P'(i, k) f(k)   (:139, :147-148)
  const bool isT2 = r < m * n, isT = r >= m * n && r < m * n + n;
  const int t2i = r % m, t2j = (r / m) % n, ti = (r - m * n) % n;
  // (operand a is read from the input stage for T2 lanes and from the scratch record for t lanes: one LDS address either way)
  const int s1b_a = isT ? (int)(scr - lds) + H::oPT + n * ti : -1 - (D::O_B + n * t2i);   // >= 0: absolute ; < 0: stage-relative, -1 - off
  const int s1b_b = isT ? -1 - D::O_f : (int)(scr - lds) + H::oP + n * t2j;
  const int s1b_add = isT ? H::op + ti : H::oZero;
  const int s1b_o = isT2 ? H::oT2 + t2i * n + t2j : isT ? H::ot + ti : H::oTrash;
  // S2-A: Qxx(i, j) = Q(i, j) + sum_k T1(i, k) A(k, j)                                              (:136)
  const int s2a_a = H::oT1 + ei * n, s2a_b = D::O_A + n * ej, s2a_add = D::O_Q + ei + n * ej;
  // S2-B, m dots per lane (e = 0 .. m-1)                                                             (:140, :143, :149-152)
  int s2b_a[m], s2b_b[m], s2b_add[m], s2b_o[m];     // a: >= 0 scratch-relative, < 0 stage-relative ; b likewise ; add: stage-relative ; o: scratch
  const bool isQux = r < n, isQu = r == n, isQuu = r > n && r < n + 1 + m, isQx = r >= n + 1 + m && r < n + 1 + m + H::NQX;
#pragma unroll
  for (int e = 0; e < m; ++e) {
    if (isQux) {            // Qux(e, j) = H(e, j) + sum_k T2(e, k) A(k, j),  j = r
      s2b_a[e] = H::oT2 + e * n; s2b_b[e] = -1 - (D::O_A + n * r); s2b_add[e] = D::O_H + e + m * r; s2b_o[e] = H::oQux + e + m * r;
    } else if (isQu) {      // Qu(e) = r(e) + sum_k B(k, e) t(k)
      s2b_a[e] = -1 - (D::O_B + n * e); s2b_b[e] = H::ot; s2b_add[e] = D::O_r + e; s2b_o[e] = H::oQu + e;
    } else if (isQuu) {     // Quu(e, c) = R(e, c) + sum_k T2(e, k) B(k, c),  c = r - n - 1
      const int c = r - n - 1;
      s2b_a[e] = H::oT2 + e * n; s2b_b[e] = -1 - (D::O_B + n * c); s2b_add[e] = D::O_R + e + m * c; s2b_o[e] = H::oQuu + e + m * c;
    } else {                // Qx(i) = q(i) + sum_k A(k, i) t(k),  i = m (r - n - 1 - m) + e
      const int i = m * (r - n - 1 - m) + e;
      const bool ok = isQx && i < n;
      const int ic = ok ? i : 0;
      s2b_a[e] = -1 - (D::O_A + n * ic); s2b_b[e] = H::ot; s2b_add[e] = D::O_q + ic; s2b_o[e] = ok ? H::oQx + i : H::oTrash;
    }
  }
  const uint32_t negrhs = isQu ? 0x80000000u : 0u;   // d = -Qu (:156)
  // gains: lanes < n solve column r of K, lane n solves d; where they go in the OUT record
  const int s3_o = isQux ? H::oK + m * r : isQu ? H::od : H::oTrash;
  const bool s3_many = isQux || isQu;
  // ---- cooperative rows: lane tid handles problem lq of the wave, rows lr + RPI t
  const int lq = tid % PPW, lr = tid / PPW;
  const int64_t lb = b0 + lq;
  const bool lvalid = lb < B;
  uint32_t ld_off[H::NLD];
  int ld_lds[H::NLD];
#pragma unroll
  for (int t = 0; t < H::NLD; ++t) {
    const int e = lr + RPI * t;
    const int ec = e < D::E_IN ? e : D::E_IN - 1;                 // (clamped lanes re-write the last element with its own value)
    ld_off[t] = lvalid ? (uint32_t)lq * 8u + (uint32_t)ec * rowB : HEX_OOB;
    ld_lds[t] = lq * SI + ec;
  }
  const double* __restrict__ pin = a.in + b0;
  double* __restrict__ pout = a.out + b0;
  double pre[H::NLD];
  auto load_rec = [&](int k) {
    k = __builtin_amdgcn_readfirstlane(k);      // (wave-uniform by construction; said so, or hipcc builds the buffer resource in VGPRs
                                                //  and wraps every load in a readfirstlane loop)
    const LaneBuf bi(pin + (int64_t)k * D::E_IN * B);
#pragma unroll
    for (int t = 0; t < H::NLD; ++t) pre[t] = lane_ld<double>(bi, ld_off[t], 0u);
  };
  auto put_rec = [&](int stage) {
    double* st = lds + stage * (PPW * SI);
#pragma unroll
    for (int t = 0; t < H::NLD; ++t) st[ld_lds[t]] = pre[t];
  };
  // ---- terminal cost-to-go: P_N, p_N -> scratch (both layouts) and OUTN
  if (tid < PPW) state_s[tid] = 0;
  {
    const LaneBuf bt(a.term + b0), bn(a.outn + b0);
    const uint32_t co = (cb < B) ? (uint32_t)cq * 8u : HEX_OOB;
    double Pe = 0.0, pe = 0.0;
    if (isE) Pe = lane_ld<double>(bt, co, (uint32_t)(ei + n * ej) * rowB);
    if (r < n) pe = lane_ld<double>(bt, co, (uint32_t)(n * n + r) * rowB);
    if (isE) { scr[H::oP + ei + n * ej] = Pe; scr[H::oPT + ej + n * ei] = Pe; }
    if (r < n) scr[H::op + r] = pe;
    if (r == 0) { scr[H::oZero] = 0.0; state_s[cq] = cvalid ? 2 : 0; }
    const uint32_t so = cvalid ? co : HEX_OOB;
    if (isE) lane_st<double>(bn, so, (uint32_t)(ei + n * ej) * rowB, Pe);
    if (r < n) lane_st<double>(bn, so, (uint32_t)(n * n + r) * rowB, pe);
  }
  double dv0 = 0.0, dv1 = 0.0;
  int fail_k = -1;
  double outv[H::NST];
  uint32_t outo[H::NST];
  int out_k = 0;
#pragma unroll
  for (int t = 0; t < H::NST; ++t) { outv[t] = 0.0; outo[t] = HEX_OOB; }
  auto store_out = [&]() {
    const LaneBuf bo(pout + (int64_t)__builtin_amdgcn_readfirstlane(out_k) * D::E_OUT * B);
#pragma unroll
    for (int t = 0; t < H::NST; ++t) lane_st<double>(bo, outo[t], 0u, outv[t]);
  };
  // records: knot point k in LDS stage k & 1, k - 1 in `pre` (written to the other stage at the top of the step), k - 2 requested
  load_rec(N - 1);
  put_rec((N - 1) & 1);
  load_rec(N >= 2 ? N - 2 : 0);
  for (int k = N - 1; k >= 0; --k) {
    const double* st = lds + (k & 1) * (PPW * SI) + cq * SI;
    put_rec((k & 1) ^ 1);                       // record k - 1 (harmless repeat of record 0 at k = 0)
    store_out();                                // knot point k + 1's OUT record (nothing before the first step)
    load_rec(k >= 2 ? k - 2 : 0);
    __syncthreads();                            // stage k and the scratch record (P', p' of step k + 1) are visible
    // S1
    {
      const double t1 = hex_dot<n>(st + s1a_a, scr + s1a_b, 0.0);
      const double* pa = s1b_a >= 0 ? lds + s1b_a : st + (-1 - s1b_a);
      const double* pb = s1b_b >= 0 ? lds + s1b_b : st + (-1 - s1b_b);
      const double t2 = hex_dot<n>(pa, pb, scr[s1b_add]);
      scr[s1a_o] = t1;
      scr[s1b_o] = t2;
    }
    __syncthreads();
    // S2
    const double Qxx = hex_dot<n>(scr + s2a_a, st + s2a_b, st[s2a_add]);
    double rhs[m];
#pragma unroll
    for (int e = 0; e < m; ++e) {
      const double* pa = s2b_a[e] >= 0 ? scr + s2b_a[e] : st + (-1 - s2b_a[e]);
      const double* pb = s2b_b[e] >= 0 ? scr + s2b_b[e] : st + (-1 - s2b_b[e]);
      const double v = hex_dot<n>(pa, pb, st[s2b_add[e]]);
      scr[s2b_o[e]] = v;
      lane_v2u w = __builtin_bit_cast(lane_v2u, v);
      w.y ^= negrhs;
      rhs[e] = __builtin_bit_cast(double, w);
    }
    __syncthreads();
    // S3: LL^T = Quu + reg I in every lane; one right-hand side per lane                         (tvlqr.cpp:155-166)
    double Quu[m * m], L[m * m];
#pragma unroll
    for (int e = 0; e < m * m; ++e) Quu[e] = scr[H::oQuu + e];
#pragma unroll
    for (int e = 0; e < m * m; ++e) L[e] = Quu[e] + ((e % m == e / m) ? reg : 0.0);
    bool fail = false;
#pragma unroll
    for (int kk = 0; kk < m; ++kk) {
      double x = L[kk + kk * m];
#pragma unroll
      for (int j = 0; j < kk; ++j) x -= L[kk + j * m] * L[kk + j * m];
      if (x <= 0.0) fail = true;
      x = sqrt(x);
      L[kk + kk * m] = x;
#pragma unroll
      for (int i = kk + 1; i < m; ++i) {
        double s = L[i + kk * m];
#pragma unroll
        for (int j = 0; j < kk; ++j) s -= L[i + j * m] * L[kk + j * m];
        L[i + kk * m] = s / x;
      }
    }
    const bool was_alive = fail_k < 0;
    if (was_alive && fail) fail_k = k;
    const bool alive = fail_k < 0;
    if (was_alive && !alive) {                  // K_k = Qux, d_k = -Qu stay unsolved and are stored; the problem stops here
      if (s3_many) {
#pragma unroll
        for (int e = 0; e < m; ++e) scr[s3_o + e] = rhs[e];
      }
      if (r == 0 && cvalid) state_s[cq] = 1;
    }
    if (alive) {
#pragma unroll
      for (int i = 0; i < m; ++i) {
        double s = rhs[i];
#pragma unroll
        for (int j = 0; j < i; ++j) s -= L[i + j * m] * rhs[j];
        rhs[i] = s / L[i + i * m];
      }
#pragma unroll
      for (int i = m - 1; i >= 0; --i) {
        double s = rhs[i];
#pragma unroll
        for (int j = i + 1; j < m; ++j) s -= L[j + i * m] * rhs[j];
        rhs[i] = s / L[i + i * m];
      }
#pragma unroll
      for (int e = 0; e < m; ++e) scr[(s3_many ? s3_o : H::oTrash) + (s3_many ? e : 0)] = rhs[e];
    }
    __syncthreads();
    // S4 / S5: cost-to-go                                                                          (tvlqr.cpp:173-191)
    if (alive) {
      double Ki[m], Kj[m], Xi[m], Xj[m], dd[m], Qu[m];
#pragma unroll
      for (int e = 0; e < m; ++e) {
        Ki[e] = scr[H::oK + e + m * ei]; Kj[e] = scr[H::oK + e + m * ej];
        Xi[e] = scr[H::oQux + e + m * ei]; Xj[e] = scr[H::oQux + e + m * ej];
        dd[e] = scr[H::od + e]; Qu[e] = scr[H::oQu + e];
      }
      const double Qxi = scr[H::oQx + ei];
      double Ui[m], w[m];
#pragma unroll
      for (int e = 0; e < m; ++e) {
        double s = 0.0;
#pragma unroll
        for (int kk = 0; kk < m; ++kk) s += Quu[e + kk * m] * Ki[kk];
        Ui[e] = 0.0 + s;                        // (Quu K)(e, i)
        double s2 = 0.0;
#pragma unroll
        for (int kk = 0; kk < m; ++kk) s2 += Quu[e + kk * m] * dd[kk];
        w[e] = 0.0 + s2;                        // (Quu d)(e)
      }
      double sv = 0.0, svt = 0.0, su = 0.0;
#pragma unroll
      for (int kk = 0; kk < m; ++kk) sv += Ki[kk] * Xj[kk];      // (K^T Qux)(i, j)
#pragma unroll
      for (int kk = 0; kk < m; ++kk) svt += Kj[kk] * Xi[kk];     // (K^T Qux)(j, i)
#pragma unroll
      for (int kk = 0; kk < m; ++kk) su += Ui[kk] * Kj[kk];
      double Pk = Qxx + su;
      Pk -= 0.0 + sv;
      Pk -= 0.0 + svt;
      double s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int kk = 0; kk < m; ++kk) s1 += Ui[kk] * dd[kk];
      double v = Qxi + (-1.0) * s1;
#pragma unroll
      for (int kk = 0; kk < m; ++kk) s2 += Ki[kk] * Qu[kk];
      v = v + (-1.0) * s2;
#pragma unroll
      for (int kk = 0; kk < m; ++kk) s3 += Xi[kk] * dd[kk];
      const double pk = v + s3;
      double q0 = 0.0, q1 = 0.0;
#pragma unroll
      for (int i = 0; i < m; ++i) q0 += dd[i] * Qu[i];
#pragma unroll
      for (int i = 0; i < m; ++i) q1 += dd[i] * w[i];
      dv0 += q0;
      dv1 += 0.5 * q1;
      if (isE) { scr[H::oP + ei + n * ej] = Pk; scr[H::oPT + ej + n * ei] = Pk; }
      if (r < n) scr[H::op + r] = pk;           // (lanes r < n are elements (r, 0): their i is r)
    }
    __syncthreads();
    // the OUT record of this knot point: read whole rows out of LDS now, STORE them at the top of the next step, ahead of its loads --
    // loads and stores share one in-order counter on gfx9 (vmcnt), so a store issued after the loads of record k - 2 would make the
    // wait for those loads a wait for the store's acknowledgement as well (measured: the top-of-step wait was the longest of the step)
    {
      const int stt = state_s[lq];
#pragma unroll
      for (int t = 0; t < H::NST; ++t) {
        const int e = lr + RPI * t;
        const bool ok = lvalid && e < D::E_OUT && (stt == 2 || (stt == 1 && e < D::O_P));
        outv[t] = lds[PPW * H::STAGES * SI + lq * SC + (e < D::E_OUT ? e : 0)];
        outo[t] = ok ? (uint32_t)lq * 8u + (uint32_t)e * rowB : HEX_OOB;
      }
      out_k = k;
    }
    __syncthreads();                            // (state 1 -> 0 only after every lane has read it)
    if (!alive && r == 0 && cvalid && state_s[cq] == 1) state_s[cq] = 0;
  }
  store_out();
  if (r == 0 && cvalid) {
    a.status[cb] = fail_k;
    a.delta_V[2 * cb + 0] = dv0;
    a.delta_V[2 * cb + 1] = dv1;
  }
}

}  // namespace altro_hip
