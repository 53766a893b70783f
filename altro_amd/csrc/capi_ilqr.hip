// capi_ilqr.hip -- C ABI: the batched AL-iLQR loop around the sweep (SolverImpl::Solve per problem), the device models' entry
// points, augmented-Lagrangian constraint blocks and the MPC receding-horizon operations.  The host sequences launches;
// every per-problem decision is taken on the device (kernels/ilqr_loop_kernels.hip).
#include "capi_internal.h"

#include <cstring>

using namespace altro_hip;
using namespace altro_hip::capi;

namespace altro_hip {
namespace capi {   // (external linkage: capi_solve.hip drives these)

// ---- iLQR loop (plan LANE) ---------------------------------------------------------------------------
// (re)build the device tables of the constraint blocks; duals restart from zero when the structure changes
template <typename T>
int al_upload_typed(altro_hip_batch* h) {
  const int64_t B = h->batch;
  h->al_Gpad_count = 0;
  for (void** p : {(void**)&h->al_d_knots, (void**)&h->al_d_big, (void**)&h->al_d_gsel, &h->al_d_G, &h->al_d_Gpad, &h->al_d_g, &h->al_d_z})
    if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (h->al_defs.empty()) { h->al_rows = 0; return 0; }
  // G on the device: p x (n + m) column-major as given on plan LANE; on plan MFMA16 p x 16 in the tile's own column order
  // (states in columns 0..11, inputs in 12..15), so that a padded shape's blocks address the padded [x; u] correctly
  const bool tile = h->plan == ALTRO_HIP_PLAN_MFMA16;
  const int w_log = h->n + h->m, w_dev = tile ? MF_N + MF_M : w_log;
  auto dev_col = [&](int e) { return (tile && e >= h->n) ? MF_N + (e - h->n) : e; };
  // Plan MFMA16 lays a block out over SLOTS of at most AL_MAXP rows (al_types.h: AL_TILE_MAXC): rows 8 s .. 8 s + 7 of a block in the
  // zero / identity / orthant cones are a block of their own (those cones project row by row, cones.cpp:13-38) with the same duals
  // in the same place; a second-order-cone block (p <= AL_MAXSOC) is one slot.  slot_base[i]: first slot definition of block i.
  auto nslots = [&](const AlDef& d) { return (tile && d.cone != CONE_SOC) ? (d.p + AL_MAXP - 1) / AL_MAXP : 1; };
  std::vector<int> slot_base(h->al_defs.size() + 1, 0);
  for (size_t i = 0; i < h->al_defs.size(); ++i) slot_base[i + 1] = slot_base[i] + nslots(h->al_defs[i]);
  std::vector<T> G;
  std::vector<int> G_off_dev(tile ? (size_t)slot_base.back() : h->al_defs.size(), 0);   // plan MFMA16: per slot definition
  for (size_t i = 0; i < h->al_defs.size(); ++i) {
    const AlDef& d0 = h->al_defs[i];
    if (tile) {   // every slot its own ps x 16 column-major matrix (the LDS-form kernels address a slot like a block)
      for (int sl = 0; sl < nslots(d0); ++sl) {
        const int r0 = sl * AL_MAXP, ps = std::min(d0.p - r0, AL_MAXP);
        const size_t off = G.size();
        G_off_dev[(size_t)slot_base[i] + sl] = (int)off;
        G.resize(off + (size_t)ps * w_dev, (T)0);
        for (int e = 0; e < w_log; ++e)
          for (int r = 0; r < ps; ++r) G[off + r + (size_t)dev_col(e) * ps] = (T)h->al_G[(size_t)d0.G_off + (r0 + r) + (size_t)e * d0.p];
      }
      continue;
    }
    G_off_dev[i] = (int)G.size();
    if (h->ragged) {   // per-knot-point dimensions (plan GENERIC): the block as given, p x (nx[k] + nu[k]) of its knot points
      G.resize(G.size() + (size_t)d0.p * d0.w, (T)0);
      for (size_t e = 0; e < (size_t)d0.p * d0.w; ++e) G[(size_t)G_off_dev[i] + e] = (T)h->al_G[(size_t)d0.G_off + e];
      continue;
    }
    G.resize(G.size() + (size_t)d0.p * w_dev, (T)0);
    for (int e = 0; e < w_log; ++e)
      for (int r = 0; r < d0.p; ++r) G[(size_t)G_off_dev[i] + r + (size_t)dev_col(e) * d0.p] = (T)h->al_G[(size_t)d0.G_off + r + (size_t)e * d0.p];
  }
  std::vector<T> Gpad;   // the same slots zero-padded for the row-layout kernels (al_types.h: AL_GP_DEF)
  if (tile) {
    Gpad.assign((size_t)slot_base.back() * (size_t)AL_GP_DEF, (T)0);
    for (size_t i = 0; i < h->al_defs.size(); ++i) {
      const AlDef& d0 = h->al_defs[i];
      if (d0.user) continue;
      for (int e = 0; e < w_log; ++e)
        for (int r = 0; r < d0.p; ++r)
          Gpad[(size_t)(slot_base[i] + r / AL_MAXP) * AL_GP_DEF + (size_t)(r % AL_MAXP) * AL_GP_LD + dev_col(e)] =
              (T)h->al_G[(size_t)d0.G_off + r + (size_t)e * d0.p];
    }
  }
  std::vector<T> g;
  std::vector<AlDef> defs = h->al_defs;
  for (size_t i = 0; i < defs.size(); ++i) {
    defs[i].g_off = (int64_t)g.size();
    const std::vector<double>& src = h->al_g[i];
    const int p = defs[i].p;
    if (defs[i].g_per_problem) {
      const size_t base = g.size();
      g.resize(base + (size_t)p * B);
      for (int64_t b = 0; b < B; ++b)
        for (int r = 0; r < p; ++r) g[base + (size_t)r * B + b] = (T)src[(size_t)b * p + r];
    } else {
      for (int r = 0; r < p; ++r) g.push_back((T)src[r]);
    }
  }
  int rows = 0;
  const bool gen = h->plan == ALTRO_HIP_PLAN_GENERIC;
  std::vector<AlKnotBig> big = h->al_knots;
  std::vector<AlKnot> knots(gen ? 0 : big.size(), AlKnot{});
  int max_ncon = 0;
  for (size_t k = 0; k < big.size(); ++k) {
    AlKnotBig& bk = big[k];
    int ns = 0;   // slots of knot point k so far (plans LANE: one per block)
    for (int j = 0; j < bk.ncon; ++j) {
      const AlDef& d = defs[bk.def[j]];
      bk.z_off[j] = rows; rows += d.p;
      bk.cone[j] = d.cone; bk.p[j] = d.p; bk.g_per_problem[j] = d.g_per_problem; bk.G_off[j] = tile ? G_off_dev[slot_base[bk.def[j]]] : G_off_dev[bk.def[j]];
      bk.g_off[j] = d.g_off;
      if (gen) continue;
      AlKnot& kn = knots[k];
      for (int sl = 0; sl < nslots(d); ++sl, ++ns) {
        if (ns >= (tile ? AL_TILE_MAXC : AL_MAXC))   // (altro_hip_add_linear_constraint counted the slots: not reached)
          return fail(ALTRO_HIP_ERR_UNSUPPORTED, "knot point %d: more constraint rows than the plan's table holds", (int)k);
        const int r0 = sl * AL_MAXP, ps = tile ? std::min(d.p - r0, d.cone == CONE_SOC ? d.p : AL_MAXP) : d.p;
        kn.def[ns] = bk.def[j];
        kn.z_off[ns] = bk.z_off[j] + r0;
        kn.cone[ns] = d.cone; kn.p[ns] = ps; kn.g_per_problem[ns] = d.g_per_problem;
        kn.G_off[ns] = tile ? G_off_dev[(size_t)slot_base[bk.def[j]] + sl] : G_off_dev[bk.def[j]];
        kn.g_off[ns] = d.g_off + (d.g_per_problem ? (int64_t)r0 * B : (int64_t)r0);   // g is [p] or [p][batch]: row r0 on
        kn.user[ns] = d.user;
        kn.Gp_off[ns] = tile ? (slot_base[bk.def[j]] + sl) * AL_GP_DEF : 0;
        // bound-type slot: every row of G is +-e_idx
        const int w = h->n + h->m;
        bool sel = d.cone != CONE_SOC;
        for (int r = 0; r < ps && sel; ++r) {
          int nz = 0, at = -1;
          for (int e = 0; e < w; ++e) {
            const double v = h->al_G[(size_t)d.G_off + (r0 + r) + (size_t)e * d.p];
            if (v != 0.0) { ++nz; at = e; if (v != 1.0 && v != -1.0) sel = false; }
          }
          if (nz != 1) sel = false;
          else kn.sidx[ns][r] = h->al_G[(size_t)d.G_off + (r0 + r) + (size_t)at * d.p] > 0 ? dev_col(at) + 1 : -(dev_col(at) + 1);
        }
        kn.sel[ns] = sel ? 1 : 0;
      }
      kn.ncon = ns;
    }
    max_ncon = std::max(max_ncon, gen ? bk.ncon : ns);
  }
  h->al_max_ncon = max_ncon;
  h->al_all_gsel = gen && !h->ragged && !defs.empty();
  if (gen && !h->ragged) {   // plan GENERIC: which blocks are bound-type (AlTable::gsel; the expansion's Gauss-Newton term is then diagonal)
    std::vector<int> gsel(defs.size() * (size_t)(1 + GEN_MAXP), 0);
    const int wn = h->n + h->m;
    for (size_t i = 0; i < defs.size(); ++i) {
      const AlDef& d = defs[i];
      bool sel = d.cone != CONE_SOC && !d.user && d.p <= GEN_MAXP;
      for (int r = 0; r < d.p && sel; ++r) {
        int nz = 0, at = -1;
        for (int e = 0; e < wn; ++e) {
          const double v = h->al_G[(size_t)d.G_off + r + (size_t)e * d.p];
          if (v != 0.0) { ++nz; at = e; if (v != 1.0 && v != -1.0) sel = false; }
        }
        if (nz != 1) sel = false;
        else gsel[i * (size_t)(1 + GEN_MAXP) + 1 + r] = h->al_G[(size_t)d.G_off + r + (size_t)at * d.p] > 0 ? at + 1 : -(at + 1);
      }
      gsel[i * (size_t)(1 + GEN_MAXP)] = sel ? 1 : 0;
      if (!sel) h->al_all_gsel = false;
    }
    if (!gsel.empty()) {
      int rcg = dmalloc(h, (void**)&h->al_d_gsel, gsel.size() * sizeof(int));
      if (rcg) return rcg;
      HIP_TRY(hipMemcpy(h->al_d_gsel, gsel.data(), gsel.size() * sizeof(int), hipMemcpyHostToDevice));
    }
  }
  h->al_row32_ok = true;   // (kernels/ilqr_row32.hip: a lane position per row, the row-wise cones)
  for (const AlDef& d : defs) if (d.cone == CONE_SOC || d.p > 32 || d.user) h->al_row32_ok = false;
  h->al_rows = rows;
  if (h->plan == ALTRO_HIP_PLAN_LANE && (uint64_t)rows * (uint64_t)B * sizeof(T) >= (1ull << 31))
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "plan LANE: %d dual rows x batch %d exceed the 2 GiB buffer window; split the batch", rows, h->batch);
  {   // uniform running knot points?  (then the dual rows of knot point k start k * rows_per_knot after those of 0)
    const AlKnotBig& k0 = big[0];
    int r0 = 0;
    for (int j = 0; j < k0.ncon; ++j) r0 += defs[k0.def[j]].p;
    bool uni = h->N >= 1 && k0.ncon > 0;
    for (int k = 1; k < h->N && uni; ++k) {
      uni = big[k].ncon == k0.ncon;
      for (int j = 0; j < k0.ncon && uni; ++j) uni = big[k].def[j] == k0.def[j] && big[k].z_off[j] == k0.z_off[j] + k * r0;
    }
    h->al_uniform = uni ? 1 : 0;
    h->al_rows_per_knot = r0;
  }
  int rc = 0;
  if (gen) {
    if ((rc = dmalloc(h, (void**)&h->al_d_big, big.size() * sizeof(AlKnotBig)))) return rc;
    HIP_TRY(hipMemcpy(h->al_d_big, big.data(), big.size() * sizeof(AlKnotBig), hipMemcpyHostToDevice));
  }
  if ((rc = dmalloc(h, (void**)&h->al_d_knots, std::max<size_t>(knots.size(), 1) * sizeof(AlKnot)))) return rc;
  if ((rc = dmalloc(h, &h->al_d_G, G.size() * sizeof(T)))) return rc;
  if ((rc = dmalloc(h, &h->al_d_g, g.size() * sizeof(T)))) return rc;
  if ((rc = dmalloc(h, &h->al_d_z, (size_t)rows * B * sizeof(T)))) return rc;
  if (!knots.empty()) HIP_TRY(hipMemcpy(h->al_d_knots, knots.data(), knots.size() * sizeof(AlKnot), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->al_d_G, G.data(), G.size() * sizeof(T), hipMemcpyHostToDevice));
  if (!Gpad.empty()) {
    if ((rc = dmalloc(h, &h->al_d_Gpad, Gpad.size() * sizeof(T)))) return rc;
    HIP_TRY(hipMemcpy(h->al_d_Gpad, Gpad.data(), Gpad.size() * sizeof(T), hipMemcpyHostToDevice));
    h->al_Gpad_count = (int)Gpad.size();
  }
  HIP_TRY(hipMemcpy(h->al_d_g, g.data(), g.size() * sizeof(T), hipMemcpyHostToDevice));
  // (memsets go on the handle's own stream: it is non-blocking, so a null-stream memset would race the kernels)
  HIP_TRY(hipMemsetAsync(h->al_d_z, 0, (size_t)rows * B * sizeof(T), h->stream));
  h->al_knots = big;
  h->al_G_count = (int)G.size();
  h->al_has_soc = 0;
  for (const AlDef& d : defs) if (d.cone == CONE_SOC) h->al_has_soc = 1;
  h->al_all_sel = gen ? 0 : 1;
  for (const AlKnot& kn : knots)
    for (int j = 0; j < kn.ncon; ++j) if (!kn.sel[j]) h->al_all_sel = 0;
  return 0;
}
int al_upload(altro_hip_batch* h) {
  if (!h->al_dirty) return 0;
  int rc = h->dtype == ALTRO_HIP_F64 ? al_upload_typed<double>(h) : al_upload_typed<float>(h);
  if (!rc) h->al_dirty = false;
  return rc;
}

template <typename T>
IlqrArgs<T> ilqr_args(altro_hip_batch* h, bool use_alpha, bool use_active, int want_deriv, double alpha_const) {
  IlqrArgs<T> a;
  a.al.knots = h->al_d_knots; a.al.G = (const T*)h->al_d_G; a.al.g = (const T*)h->al_d_g;
  a.al.z = (T*)h->al_d_z; a.al.enabled = h->al_defs.empty() ? 0 : 1;
  a.al.uniform = h->al_uniform; a.al.rows_per_knot = h->al_rows_per_knot; a.al.N = h->N; a.al.G_count = h->al_G_count; a.al.has_soc = h->al_has_soc; a.al.all_sel = h->al_all_sel; a.al.Gpad = (decltype(a.al.Gpad))h->al_d_Gpad; a.al.Gpad_count = h->al_Gpad_count; a.al.max_ncon = h->al_max_ncon;
  a.mode = EXPAND_GRADIENT | EXPAND_HESSIAN;
  a.in = (T*)h->l_in; a.term = (T*)h->l_term; a.out = (const T*)h->l_out; a.outn = (const T*)h->l_outn;
  a.nom = (T*)h->l_nom; a.cand = (T*)h->l_xuy; a.x0 = (const T*)h->l_x0;
  a.cost = (const T*)(h->cost_dense ? h->l_costq : h->l_cost); a.cost_kind = h->cost_dense ? 1 : 0;
  a.alpha = use_alpha ? h->i_alpha : nullptr;
  a.active = use_active ? h->i_active : nullptr;
  a.phi = h->i_phi; a.dphi = h->i_dphi; a.prob = h->i_prob;
  a.mp = h->model; a.N = h->N; a.batch = h->batch; a.want_derivative = want_deriv; a.alpha_const = alpha_const;
  a.cand_spec = (T*)h->i_cand_spec; a.spec_trials = h->i_cand_spec ? h->spec_trials : 1; a.spec_sel = h->i_spec_sel;
  a.spec_pre = h->i_cand_spec ? h->spec_pre : 0;
  a.spec_flip = 0;
  a.spec_stride = (int64_t)h->batch * (h->N + 1) * lane_sizes(h->n, h->m).e_xuy;
  a.ls_beta = h->spec_beta; a.ls_max_iters = h->spec_max_iters;
  a.merit_jk = h->merit_split == 1 ? (T*)h->i_merit_jk : nullptr;
  a.spec_jac = (T*)h->i_spec_jac;
  return a;
}
// Most line-search steps one merit launch of this handle can ever evaluate: the sequenced loop speculates 2 / 4 / 8 wide only
// while wavefronts x trials x 2 stays below what the chip holds (plan LANE: 512 wavefronts; the scattered searching lanes
// occupy every wave of the batch), the fused solve kernel never goes beyond 4.  The spare trajectories and the per-trial cost
// array are sized to THIS, not to ILQR_SPEC_TRIALS: at 65536 bicycles that is 3 spares instead of 7 (0.8 GB instead of 1.9).
int spec_trials_cap(const altro_hip_batch* h) {
  if (h->plan != ALTRO_HIP_PLAN_LANE) return ILQR_SPEC_TRIALS;   // MFMA16: the searching problems alone decide
  const int64_t units = (h->batch + 63) / 64;
  int t = ILQR_SPEC_TRIALS;
  while (t > 4 && units * t > 512) t /= 2;   // the solve loop widens while units * (2 * trials) <= 512: the widest launch is units * t
  return t;
}
// `count` spare candidate trajectories (grown on demand, never shrunk); false = no memory for them (nothing is changed)
bool ensure_spares(altro_hip_batch* h, int count, size_t bytes_each) {
  if (h->spare_count >= count) return true;
  if (h->spare_failed > 0 && count >= h->spare_failed) return false;   // this size did not fit before: do not ask again
  void* fresh = nullptr;
  if (hipMalloc(&fresh, (size_t)count * bytes_each) != hipSuccess) { (void)hipGetLastError(); h->spare_failed = count; return false; }
  if (h->i_cand_spec) { (void)hipStreamSynchronize(h->stream); (void)hipFree(h->i_cand_spec); h->device_bytes -= (size_t)h->spare_count * bytes_each; }
  h->i_cand_spec = fresh;
  h->spare_count = count;
  h->device_bytes += (size_t)count * bytes_each;
  return true;
}
// The buffers of the three-launch merit evaluation (first use).  No memory for them: the one-launch kernel, for good.
void merit_split_prepare(altro_hip_batch* h) {
  if (h->merit_split >= 0) return;
  h->merit_split = form(h, ALTRO_HIP_FORM_MERIT_ONE_LAUNCH) ? 0 : 1;
  if (!h->merit_split) return;
  const size_t jk = (size_t)spec_trials_cap(h) * (h->N + 1) * h->batch * h->esz;
  const size_t jac = ((size_t)h->N * (h->n * h->n + h->n * h->m + h->n + h->m) + h->n) * h->batch * h->esz;
  if (dmalloc(h, &h->i_merit_jk, jk) || dmalloc(h, &h->i_spec_jac, jac)) {
    (void)hipGetLastError();
    if (h->i_merit_jk) { (void)hipFree(h->i_merit_jk); h->i_merit_jk = nullptr; }
    h->merit_split = 0;
  }
}

template <typename T>
int ilqr_launch(altro_hip_batch* h, int which, IlqrArgs<T> a) {
  if (h->model.kind == MODEL_USER) return rtc_launch<T>(h, which, a);   // the caller's own dynamics, compiled at run time (capi_rtc.hip)
  const int rc = ilqr_launch_kernel<T>(h->stream, which, h->model.kind, h->n, h->m, a);
  if (rc == 1) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "no device model for (kind, n, m) = (%d, %d, %d)", h->model.kind, h->n, h->m);
  if (rc) return fail(ALTRO_HIP_ERR_HIP, "iLQR kernel launch failed");
  return 0;
}
template <typename S>
int wave_run(altro_hip_batch* h, int which, bool use_alpha, bool use_active, int want_deriv, double alpha_const, int mode) {
  IlqrWaveArgs<S> a;
  a.al.knots = h->al_d_knots; a.al.G = (const S*)h->al_d_G; a.al.g = (const S*)h->al_d_g; a.al.z = (S*)h->al_d_z;
  a.al.enabled = h->al_defs.empty() ? 0 : 1;
  a.al.uniform = h->al_uniform; a.al.rows_per_knot = h->al_rows_per_knot; a.al.N = h->N; a.al.G_count = h->al_G_count; a.al.has_soc = h->al_has_soc; a.al.all_sel = h->al_all_sel; a.al.Gpad = (decltype(a.al.Gpad))h->al_d_Gpad; a.al.Gpad_count = h->al_Gpad_count; a.al.max_ncon = h->al_max_ncon;
  a.mode = mode;
  a.penalty_scaling = h->expand_penalty_scaling; a.penalty_max = h->expand_penalty_max;
  if (which == IK_STATIONARITY || which == IK_DUAL)   // constraint rows in the DPP form unless ALTRO_HIP_FORM_ALROWS_LDS
    a.mode = form(h, ALTRO_HIP_FORM_ALROWS_LDS) ? 0 : STAT_NO_FEAS;
  if (which == IK_EXPAND && form(h, ALTRO_HIP_FORM_EXPAND_LDS) && !h->cost_dense) a.mode |= EXPAND_LDS;   // (the dense cost lives in the row-layout kernels only)
  a.costd = (const S*)h->m_costd; a.costd_term = (const S*)h->m_costd_term; a.cost_dense = h->cost_dense ? 1 : 0;
  if (h->model_set) {   // a device model: the dynamics expansion rides with every gradient expansion of a stored trajectory
    a.mp = h->model;    // (not with the end-of-sweep refresh, EXPAND_NEXT: the merit pass that made the candidate left Z already)
    if (which == IK_EXPAND && (a.mode & EXPAND_GRADIENT) && !(a.mode & EXPAND_NEXT)) a.mode |= EXPAND_DYN;
  }
  if (which == IK_MERIT)   // a line-search round: the DPP form of the merit evaluation unless ALTRO_HIP_FORM_MERIT_LDS keeps the LDS form
    a.mode = form(h, ALTRO_HIP_FORM_MERIT_LDS) ? 0 : form(h, ALTRO_HIP_FORM_MERIT_DPP_ALWAYS) ? 3 : 2;   // (3: the DPP form whatever the launcher's rule)
  a.dyn = (const S*)h->m_in; a.dyn_bs = h->m_st.in_bs; a.dyn_ks = h->m_st.in_ks;
  a.cin = (S*)h->m_cin; a.cin_bs = h->m_st.cin_bs; a.cin_ks = h->m_st.cin_ks;
  a.term = (S*)h->m_term; a.out = (const S*)h->m_out; a.out_bs = h->m_st.out_bs; a.out_ks = h->m_st.out_ks;
  a.outn = (const S*)h->m_outn; a.nom = (S*)h->m_nom; a.cand = (S*)h->m_xuy; a.xuy_bs = h->m_st.xuy_bs;
  a.xuy_ks = h->m_st.xuy_ks; a.costp = (const S*)h->m_costp; a.x0 = (const S*)h->x0;
  a.alpha = use_alpha ? h->i_alpha : nullptr; a.active = use_active ? h->i_active : nullptr;
  a.phi = h->i_phi; a.dphi = h->i_dphi; a.prob = h->i_prob; a.N = h->N; a.batch = h->batch;
  a.want_derivative = want_deriv; a.alpha_const = alpha_const;
  a.cand_spec = (S*)h->i_cand_spec; a.spec_trials = h->i_cand_spec ? h->spec_trials : 1; a.spec_sel = h->i_spec_sel;
  a.spec_pre = h->i_cand_spec ? h->spec_pre : 0;
  a.spec_stride = (int64_t)h->batch * (h->N + 1) * 28;
  a.ls_beta = h->spec_beta; a.ls_max_iters = h->spec_max_iters;
  a.skip = which == IK_STATIONARITY ? h->stat_skip : nullptr;
  if constexpr (sizeof(S) == 8) {   // affine line-search trials (capi_solve.hip switches them on for a solve): dynamics as data only
    if (h->aff_enabled && !h->model_set) {
      if (h->aff_store && (which == IK_MERIT2 || which == IK_MERIT)) {   // the sweep's phi(0) evaluation leaves base + sensitivity behind
        a.sens = (double*)h->i_sens; a.sens_alpha = (double*)h->i_sens_alpha;
      } else if (which == IK_MERIT && h->aff_round && a.mode >= 2 && !a.spec_pre) {
        a.sens = (double*)h->i_sens; a.sens_alpha = (double*)h->i_sens_alpha;
        a.aff = 1; a.aff_part = (double*)h->i_aff_part; a.aff_on = (int*)h->i_aff_on;   // (flags: zero when allocated, cleared again by the reduction)
      }
    }
  }
  if (h->model_set && h->model.kind == MODEL_USER) {   // the caller's own model, compiled at run time (capi_rtc.hip): fp64 handles only
    if constexpr (sizeof(S) == 8) {
      if (which == IK_ROLLOUT || which == IK_MERIT || which == IK_MERIT2) return rtc_tile_launch(h, which, a);
      if (which == IK_EXPAND && (a.mode & EXPAND_DYN)) {
        const int rcm = rtc_tile_launch(h, IK_EXPAND, a);
        if (rcm) return rcm;
      }
    }
    a.mp.kind = MODEL_LINEAR;   // (what follows are the cost kernels: none of them steps the dynamics)
  }
  const int rc = ilqr_wave_launch_kernel<S>(h->stream, which, a);
  if (rc == 1) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "operation %d is not available on plan MFMA16", which);
  if (rc) return fail(ALTRO_HIP_ERR_HIP, "iLQR kernel launch failed");
  return 0;
}
// one problem's length of the dense arrays of plan GENERIC's iLQR loop (nominal trajectory, the cost's own blocks): the sum of the
// knot points' blocks, whatever their dimensions
struct GenSizes { int64_t sx, su, sQ, sR, sH; };
GenSizes gen_sizes(const altro_hip_batch* h) {
  GenSizes z{0, 0, 0, 0, 0};
  for (int k = 0; k <= h->N; ++k) {
    const int64_t nk = h->ragged ? h->nxv[k] : h->n, mk = k < h->N ? (h->ragged ? h->nuv[k] : h->m) : 0;
    z.sx += nk; z.su += mk; z.sQ += nk * nk; z.sR += mk * mk; z.sH += mk * nk;
  }
  return z;
}
// the loop kernels of kernels/ilqr_row32.hip serve this handle: plan MFMA32's shapes (also on a handle created as plan GENERIC), fp64,
// uniform dimensions, dynamics as data, every constraint block in a row-wise cone with at most 32 rows
bool row32_eligible(const altro_hip_batch* h) {
  return h->plan == ALTRO_HIP_PLAN_GENERIC && h->dtype == ALTRO_HIP_F64 && !h->ragged && !h->model_set && tile32_supported(h->n, h->m) &&
         !form(h, ALTRO_HIP_FORM_GENERIC_MERIT_LDS) && (h->al_defs.empty() || h->al_row32_ok);
}
// the same with a compiled-in device model (row32_model.hip has the kernels: ilqr_generic_model_supported's models)
bool row32_model_eligible(const altro_hip_batch* h) {
  return h->plan == ALTRO_HIP_PLAN_GENERIC && h->dtype == ALTRO_HIP_F64 && !h->ragged && h->model_set && tile32_supported(h->n, h->m) &&
         (ilqr_generic_model_supported(h->model.kind, h->n, h->m) || (h->model.kind == MODEL_USER && h->rtc_row32_ok)) &&
         !form(h, ALTRO_HIP_FORM_GENERIC_MERIT_LDS) &&
         (h->al_defs.empty() || h->al_row32_ok);
}
// plan GENERIC: any (n_k, m_k) up to 64, dynamics as data, quadratic cost, linear constraint blocks (kernels/ilqr_generic.hip)
template <typename T>
int gen_run(altro_hip_batch* h, int which, bool use_alpha, bool use_active, int want_deriv, double alpha_const, int mode) {
  const int n = h->n, m = h->m, N = h->N;
  IlqrGenArgs<T> a;
  a.A = (const T*)h->g_arr[G_A]; a.A_bs = h->g_bstride[G_A]; a.B = (const T*)h->g_arr[G_B]; a.B_bs = h->g_bstride[G_B];
  a.f = (const T*)h->g_arr[G_f]; a.f_bs = h->g_bstride[G_f];
  a.Q = (T*)h->g_arr[G_Q]; a.Q_bs = h->g_bstride[G_Q]; a.R = (T*)h->g_arr[G_R]; a.R_bs = h->g_bstride[G_R];
  a.H = (T*)h->g_arr[G_H]; a.H_bs = h->g_bstride[G_H]; a.q = (T*)h->g_arr[G_q]; a.q_bs = h->g_bstride[G_q];
  a.r = (T*)h->g_arr[G_r]; a.r_bs = h->g_bstride[G_r];
  a.K = (const T*)h->g_arr[G_K]; a.K_bs = h->g_bstride[G_K]; a.d = (const T*)h->g_arr[G_d]; a.d_bs = h->g_bstride[G_d];
  a.P = (const T*)h->g_arr[G_P]; a.P_bs = h->g_bstride[G_P]; a.p = (const T*)h->g_arr[G_p]; a.p_bs = h->g_bstride[G_p];
  a.x = (T*)h->g_arr[G_x]; a.x_bs = h->g_bstride[G_x]; a.u = (T*)h->g_arr[G_u]; a.u_bs = h->g_bstride[G_u];
  a.y = (T*)h->g_arr[G_y]; a.y_bs = h->g_bstride[G_y];
  a.xn = (T*)h->g_xn; a.un = (T*)h->g_un;
  a.cQ = (const T*)h->g_cQ; a.cR = (const T*)h->g_cR; a.cH = (const T*)h->g_cH; a.cq = (const T*)h->g_cq; a.cr = (const T*)h->g_cr;
  a.cc = (const T*)h->g_cc;
  a.x0 = (const T*)h->x0; a.x0_stride = h->x0_stride;
  a.alpha = use_alpha ? h->i_alpha : nullptr; a.active = use_active ? h->i_active : nullptr; a.alpha_const = alpha_const;
  a.phi = h->i_phi; a.dphi = h->i_dphi; a.prob = h->i_prob;
  a.N = N; a.n = n; a.m = m; a.batch = h->batch; a.want_derivative = want_deriv; a.mode = mode;
  a.off = h->g_off; a.nx = h->g_nx; a.nu = h->g_nu;
  const GenSizes gs = gen_sizes(h);
  a.sx = gs.sx; a.su = gs.su; a.sQ = gs.sQ; a.sR = gs.sR; a.sH = gs.sH;
  a.al.knots = h->al_d_knots; a.al.G = (const T*)h->al_d_G; a.al.g = (const T*)h->al_d_g; a.al.z = (T*)h->al_d_z;
  a.al.enabled = h->al_defs.empty() ? 0 : 1;
  a.al.uniform = h->al_uniform; a.al.rows_per_knot = h->al_rows_per_knot; a.al.N = h->N; a.al.G_count = h->al_G_count;
  a.al.has_soc = h->al_has_soc; a.al.all_sel = h->al_all_sel; a.al.Gpad = nullptr; a.al.Gpad_count = 0;
  a.al.big = h->al_d_big; a.al.gsel = h->al_d_gsel;
  // MeritFunction in the row layout of kernels/ilqr_row32.hip: plan MFMA32's shapes (also on a handle created as plan GENERIC), fp64,
  // dynamics as data, every constraint block in a row-wise cone with at most 32 rows
  a.row32 = (sizeof(T) == 8 && row32_eligible(h)) ? 1 : 0;
  a.row32m = (sizeof(T) == 8 && row32_model_eligible(h)) ? 1 : 0;
  if ((a.row32 || a.row32m) && which == IK_STATIONARITY && N > 32) {   // the stationarity walk in chunks of 32 knot points: their maxima meet here
    const size_t need = (size_t)((N + 31) / 32) * h->batch * 2 * sizeof(double);
    if (h->g_stat_part_bytes < need) {
      if (h->g_stat_part) { (void)hipFree(h->g_stat_part); h->g_stat_part = nullptr; h->g_stat_part_bytes = 0; }
      if (hipMalloc((void**)&h->g_stat_part, need) == hipSuccess) h->g_stat_part_bytes = need;
      else { (void)hipGetLastError(); h->g_stat_part = nullptr; }   // (an optimisation only: one chunk then)
    }
    a.stat_part = h->g_stat_part;
  }
  if (h->model_set) {   // a device model: the dynamics expansion rides with every gradient expansion of a stored trajectory
    a.mp = h->model;
    if (which == IK_EXPAND && (a.mode & EXPAND_GRADIENT)) a.mode |= EXPAND_DYN;
  }
  if constexpr (sizeof(T) == 8) {
    if (h->model_set && h->model.kind == MODEL_USER) {   // the caller's own model, compiled at run time (capi_rtc.hip)
      if (which == IK_ROLLOUT || which == IK_MERIT || (which == IK_MERIT2 && a.row32m)) return rtc_gen_launch(h, which, a);
      if (which == IK_EXPAND) {   // the cost's expansion from the library's kernel (the model kind it sees is not one it steps), then A_k, B_k
        IlqrGenArgs<T> ac = a;
        ac.mp.kind = MODEL_LINEAR;
        const int rc0 = ilqr_generic_launch<T>(h->stream, which, ac);
        if (rc0) return fail(ALTRO_HIP_ERR_HIP, "iLQR kernel launch failed");
        return (a.mode & EXPAND_DYN) ? rtc_gen_launch(h, which, a) : 0;
      }
    }
  }
  const int rc = ilqr_generic_launch<T>(h->stream, which, a);
  if (rc == 1) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "operation %d is not available on plan GENERIC", which);
  if (rc) return fail(ALTRO_HIP_ERR_HIP, "iLQR kernel launch failed");
  return 0;
}
int ilqr_run(altro_hip_batch* h, int which, bool use_alpha, bool use_active, int want_deriv, double alpha_const, int mode) {
  int rc = al_upload(h);
  if (rc) return rc;
  if (h->plan == ALTRO_HIP_PLAN_GENERIC)
    return h->dtype == ALTRO_HIP_F64 ? gen_run<double>(h, which, use_alpha, use_active, want_deriv, alpha_const, mode)
                                     : gen_run<float>(h, which, use_alpha, use_active, want_deriv, alpha_const, mode);
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {   // linear dynamics: "expand" = cost gradient (+ AL Hessian terms when constrained)
    rc = h->dtype == ALTRO_HIP_F64 ? wave_run<double>(h, which, use_alpha, use_active, want_deriv, alpha_const, mode)
                                   : wave_run<float>(h, which, use_alpha, use_active, want_deriv, alpha_const, mode);
    // (altro_hip_batch::expansion_current) who writes the candidate trajectory, and who leaves its expansion behind
    if (which == IK_MERIT) h->expansion_current = !rc && want_deriv && !use_active && h->spec_trials == 1 && !h->spec_pre && h->al_defs.empty();
    else if (which == IK_ROLLOUT || which == IK_SHIFT || which == IK_SPEC_SELECT || which == IK_MERIT2 || which == IK_DUAL) h->expansion_current = false;
    return rc;
  }
  if (which == IK_MERIT) merit_split_prepare(h);
  if (h->dtype == ALTRO_HIP_F64) {
    auto a = ilqr_args<double>(h, use_alpha, use_active, want_deriv, alpha_const);
    a.mode = mode;
    return ilqr_launch<double>(h, which, a);
  }
  auto a = ilqr_args<float>(h, use_alpha, use_active, want_deriv, alpha_const);
  a.mode = mode;
  return ilqr_launch<float>(h, which, a);
}

// plan GENERIC: the cost's own dense blocks (what the merit function and the expansion evaluate) next to the backward sweep's
// inputs, which altro_hip_set_cost fills; nominal trajectory allocated on first use.  Host arrays: Q [nb][nkx][n n], R [nb][nku][m m],
// H [nb][nku][m n], q [nb][nkx][n], r [nb][nku][m], c [nb][nkx] or NULL.
template <typename T>
int generic_cost_def(altro_hip_batch* h, const double* Q, const double* R, const double* H, const double* q, const double* r,
                     const double* c, int kz, int bz) {
  const int n = h->n, m = h->m, N = h->N;
  const int nkx = kz ? 2 : N + 1, nku = kz ? 1 : N;
  const size_t B = h->batch, E = h->esz;
  const GenSizes gs = gen_sizes(h);
  int rc = 0;
  {   // every array has its own guard: a failed allocation in the middle leaves the later pointers null and the next call retries them
    // (one `if (!h->g_cQ)` around all eight skipped the retry and handed null arrays to the kernels)
    struct { void** p; size_t bytes; bool zero; } want[] = {
        {&h->g_cQ, B * gs.sQ * E, false}, {&h->g_cR, B * gs.sR * E, false}, {&h->g_cH, B * gs.sH * E, false},
        {&h->g_cq, B * gs.sx * E, false}, {&h->g_cr, B * gs.su * E, false}, {&h->g_cc, B * (N + 1) * E, false},
        {&h->g_xn, B * gs.sx * E, true},  {&h->g_un, B * gs.su * E, true}};
    for (auto& w : want) {
      if (*w.p) continue;
      if ((rc = dmalloc(h, w.p, w.bytes))) return rc;
      if (w.zero) HIP_TRY(hipMemsetAsync(*w.p, 0, w.bytes, h->stream));
    }
    // the candidate inputs (g_arr[G_u]) are NOT touched here: altro_hip_set_input_guess may come before the cost, like the reference's
    // SetInput may; they are zeroed once, when the handle is created
  }
  auto put = [&](void* dst, int blk, int nk_total, const double* src, int nk_host, bool with_terminal) -> int {
    if (!src) { HIP_TRY(hipMemsetAsync(dst, 0, B * nk_total * blk * E, h->stream)); return 0; }
    int r_ = aos_set<T>(h, (T*)dst, (int64_t)nk_total * blk, (int64_t)blk, src, blk, N, kz, bz, nk_host, 0);
    if (!r_ && with_terminal)   // knot point N: element N of a full host array, or the second entry of a {running, terminal} pair
      r_ = aos_set<T>(h, (T*)dst + (size_t)N * blk, (int64_t)nk_total * blk, (int64_t)blk, src, blk, 1, 1, bz, nk_host, (kz ? 1 : N) * blk);
    return r_;
  };
  if (h->ragged) {   // per-knot-point dimensions: the caller's packed [b][k][block_k] arrays are the device layout, one block per problem
    if (kz) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "k_stride_zero needs uniform dimensions");
    auto whole = [&](void* dst, int64_t len, const double* src) -> int {
      if (!src) { HIP_TRY(hipMemsetAsync(dst, 0, B * len * E, h->stream)); return 0; }
      return aos_set<T>(h, (T*)dst, len, len, src, (int)len, 1, 0, bz);
    };
    rc = whole(h->g_cQ, gs.sQ, Q);
    if (!rc) rc = whole(h->g_cR, gs.sR, R);
    if (!rc) rc = whole(h->g_cH, gs.sH, H);
    if (!rc) rc = whole(h->g_cq, gs.sx, q);
    if (!rc) rc = whole(h->g_cr, gs.su, r);
    if (!rc) rc = whole(h->g_cc, N + 1, c);
    return rc;
  }
  rc = put(h->g_cQ, n * n, N + 1, Q, nkx, true);
  if (!rc) rc = put(h->g_cR, m * m, N, R, nku, false);
  if (!rc) rc = put(h->g_cH, m * n, N, H, nku, false);
  if (!rc) rc = put(h->g_cq, n, N + 1, q, nkx, true);
  if (!rc) rc = put(h->g_cr, m, N, r, nku, false);
  if (!rc) rc = put(h->g_cc, 1, N + 1, c, nkx, true);
  return rc;
}
int generic_cost(altro_hip_batch* h, const double* Q, const double* R, const double* H, const double* q, const double* r,
                 const double* c, int kz, int bz) {
  int rc = altro_hip_set_cost(h, Q, R, H, q, r, 0, kz, bz);   // the backward sweep's blocks: lxx = Q, luu = R, lux = H (lx, lu refreshed by the loop)
  if (rc) return rc;
  rc = h->dtype == ALTRO_HIP_F64 ? generic_cost_def<double>(h, Q, R, H, q, r, c, kz, bz) : generic_cost_def<float>(h, Q, R, H, q, r, c, kz, bz);
  if (!rc) { h->lqr_cost_set = true; h->ilqr_linear = true; h->cost_dense = true; h->spec_no_memory = true; /* one step per launch on this plan */ }
  return rc;
}
int ilqr_check(altro_hip_batch* h, bool need_guess) {
  int rc = loop_entry(h, true);
  if (rc) return rc;
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {   // dynamics are data (altro_hip_set_dynamics) or a device model of the tile plan
    if (!h->dyn_set && !h->model_set)
      return fail(ALTRO_HIP_ERR_NOT_SET, "neither altro_hip_set_dynamics nor altro_hip_set_model has been called");
  } else if (h->plan == ALTRO_HIP_PLAN_GENERIC) {   // any (n, m) up to 64: dynamics as data, a quadratic cost
    if (h->n > 64 || h->m > 64)
      return fail(ALTRO_HIP_ERR_UNSUPPORTED, "the iLQR loop of plan GENERIC gives one lane to every state row and one to every input row: "
                                             "n, m <= 64 (got %d, %d); the TVLQR sweeps (altro_hip_backward / _forward_ltv / _sweep) take any size", h->n, h->m);
    if (!h->dyn_set) return fail(ALTRO_HIP_ERR_NOT_SET, "altro_hip_set_dynamics has not been called");
    if (h->ragged && h->is_diag)   // (a diagonal altro_hip_set_cost re-lays the sweep's Q / R offsets, which the loop's dense blocks share)
      return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "per-knot-point dimensions: the iLQR loop works on the dense blocks of altro_hip_set_quadratic_cost / "
                                              "_set_tracking_cost; a later altro_hip_set_cost(is_diag = 1) re-laid them -- set the loop's cost again");
  } else if (!h->model_set) {
    return fail(ALTRO_HIP_ERR_NOT_SET, "altro_hip_set_model has not been called (plan LANE runs device models; for dynamics given as "
                                       "data -- altro_hip_set_dynamics -- create the handle with ALTRO_HIP_PLAN_MFMA16)");
  }
  if (!h->lqr_cost_set) return fail(ALTRO_HIP_ERR_NOT_SET, "neither altro_hip_set_tracking_cost nor altro_hip_set_quadratic_cost has been called");
  if (!h->x0_set) return fail(ALTRO_HIP_ERR_NOT_SET, "altro_hip_set_initial_state has not been called");
  if (need_guess && !h->guess_set) return fail(ALTRO_HIP_ERR_NOT_SET, "altro_hip_set_input_guess has not been called");
  return 0;
}

template IlqrArgs<double> ilqr_args<double>(altro_hip_batch*, bool, bool, int, double);
template IlqrArgs<float> ilqr_args<float>(altro_hip_batch*, bool, bool, int, double);
}  // namespace capi
}  // namespace altro_hip

extern "C" {

// ---- iLQR loop entry points -----------------------------------------------------------------------------
int altro_hip_set_model(altro_hip_batch* h, int model, float timestep, int bicycle_frame,
                        double bicycle_length, double bicycle_lr) {
  int rc = loop_entry(h);
  if (rc) return rc;
  h->expansion_current = false;
  if (!(timestep > 0.0f)) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "time step must be positive (ErrorCodes::TimestepNotPositive)");
  // plan AUTO put a small shape on the padded tile (cheaper sweeps below ~6000 problems, capi_core.hip); a compiled-in model that only
  // plan LANE carries moves the still-empty handle there
  if (h->plan == ALTRO_HIP_PLAN_MFMA16 && !ilqr_tile_model_supported(model, h->n, h->m) && ilqr_supported(model, h->n, h->m) &&
      lane_supported(h->n, h->m)) {
    if ((rc = replan_empty_handle(h, ALTRO_HIP_PLAN_LANE))) return rc;
  }
  if (h->plan == ALTRO_HIP_PLAN_GENERIC) {   // plans GENERIC / MFMA32: the model steps inside that plan's loop kernels (kernels/ilqr_generic.hip)
    if (h->ragged) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "device models need uniform dimensions");
    if (!ilqr_generic_model_supported(model, h->n, h->m))
      return fail(ALTRO_HIP_ERR_UNSUPPORTED, "no device model %d for plan %d with (n, m) = (%d, %d): plans GENERIC / MFMA32 carry "
                                             "ALTRO_HIP_MODEL_QUADROTOR13 at (13, 4)", model, altro_hip_batch_plan(h), h->n, h->m);
    if (h->dtype != ALTRO_HIP_F64) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "device models on plans GENERIC / MFMA32 run on fp64 handles");
    h->model = ModelParams{model, timestep, 0, 2.7, 1.5};
    h->model_set = true;
    // A_k, B_k are the EXPANSION's from now on (written by the expansion / merit kernels), f = 0
    HIP_TRY(hipMemsetAsync(h->g_arr[G_f], 0, (size_t)h->batch * h->g_bstride[G_f] * h->esz, h->stream));
    HIP_TRY(hipMemsetAsync(h->g_arr[G_A], 0, (size_t)h->batch * h->g_bstride[G_A] * h->esz, h->stream));
    HIP_TRY(hipMemsetAsync(h->g_arr[G_B], 0, (size_t)h->batch * h->g_bstride[G_B] * h->esz, h->stream));
    h->dyn_set = true; h->has_f = 0;
    return 0;
  }
  const bool tile = h->plan == ALTRO_HIP_PLAN_MFMA16 && ilqr_tile_model_supported(model, h->n, h->m);   // kernels/ilqr_tile_model.hip
  if (tile && h->dtype != ALTRO_HIP_F64)
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "device models on plan MFMA16 run on fp64 records (create the handle with ALTRO_HIP_F64)");
  if (!tile && (h->plan != ALTRO_HIP_PLAN_LANE || !ilqr_supported(model, h->n, h->m)))
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "no device model %d for plan %d with (n, m) = (%d, %d)", model, h->plan, h->n, h->m);
  h->model = ModelParams{model, timestep, bicycle_frame, bicycle_length > 0 ? bicycle_length : 2.7,
                         bicycle_lr > 0 ? bicycle_lr : 1.5};
  h->model_set = true;
  if (tile) {   // the DYN records are the EXPANSION's from now on (Z = [A B] written by the expansion / merit kernels, f = 0)
    HIP_TRY(hipMemsetAsync(h->m_in, 0, (size_t)h->batch * h->N * MF_DYN * h->esz, h->stream));
    h->dyn_set = true; h->has_f = 0;
  }
  return 0;
}

int altro_hip_set_tracking_cost(altro_hip_batch* h, const double* Qd, const double* Rd, const double* xref,
                                const double* uref, int kz, int bz) {
  // ALTROSolver::SetLQRCost (altro_solver.cpp:138-172): q = -Q xref, r = -R uref,
  // c = 1/2 xref'Q xref (+ 1/2 uref'R uref for k < N) -> KnotPointData::SetDiagonalCost
  int rc = loop_entry(h, true);
  if (rc) return rc;
  h->expansion_current = false;
  if (h->plan != ALTRO_HIP_PLAN_LANE && h->plan != ALTRO_HIP_PLAN_MFMA16 && h->plan != ALTRO_HIP_PLAN_GENERIC)
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "unknown plan");
  if (!Qd || !Rd || !xref || !uref) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "Qd, Rd, xref, uref are required");
  if (h->dev_ptrs)
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "altro_hip_set_tracking_cost forms q = -Q xref on the host: pass host arrays "
                                           "(altro_hip_set_pointer_mode(h, 0))");
  if (h->ragged) {   // per-knot-point dimensions: Qd, xref [b][sum nx], Rd, uref [b][sum nu] -> the dense blocks the loop of plan GENERIC evaluates
    if (kz) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "k_stride_zero needs uniform dimensions");
    const int N = h->N, nb = bz ? 1 : h->batch;
    const GenSizes gs = gen_sizes(h);
    std::vector<double> Q((size_t)nb * gs.sQ, 0.0), R((size_t)nb * gs.sR, 0.0), H((size_t)nb * gs.sH, 0.0), q((size_t)nb * gs.sx),
        r((size_t)nb * gs.su), c((size_t)nb * (N + 1));
    for (int b = 0; b < nb; ++b) {
      int64_t ox = 0, ou = 0, oQ = 0, oR = 0;
      for (int k = 0; k <= N; ++k) {
        const int nk = h->nxv[k], mk = k < N ? h->nuv[k] : 0;
        const double *Q_ = Qd + (size_t)b * gs.sx + ox, *x_ = xref + (size_t)b * gs.sx + ox;
        double cc = 0.0;
        for (int i = 0; i < nk; ++i) {
          Q[(size_t)b * gs.sQ + oQ + i + (size_t)i * nk] = Q_[i];
          q[(size_t)b * gs.sx + ox + i] = -(Q_[i] * x_[i]);
          cc += x_[i] * Q_[i] * x_[i];
        }
        cc *= 0.5;
        if (k < N) {
          const double *R_ = Rd + (size_t)b * gs.su + ou, *u_ = uref + (size_t)b * gs.su + ou;
          double cu = 0.0;
          for (int i = 0; i < mk; ++i) {
            R[(size_t)b * gs.sR + oR + i + (size_t)i * mk] = R_[i];
            r[(size_t)b * gs.su + ou + i] = -(R_[i] * u_[i]);
            cu += u_[i] * R_[i] * u_[i];
          }
          cc += 0.5 * cu;
        }
        c[(size_t)b * (N + 1) + k] = cc;
        ox += nk; ou += mk; oQ += (int64_t)nk * nk; oR += (int64_t)mk * mk;
      }
    }
    return generic_cost(h, Q.data(), R.data(), H.data(), q.data(), r.data(), c.data(), 0, bz);
  }
  const int n = h->n, m = h->m, N = h->N;
  const int nb = bz ? 1 : h->batch, nkx = kz ? 2 : N + 1, nku = kz ? 1 : N;
  const int E = 2 * n + 2 * m + 1;
  // host-side arithmetic exactly as the reference's setter does it, then one upload per field
  std::vector<double> q((size_t)nb * nkx * n), r((size_t)nb * nku * m), c((size_t)nb * nkx);
  for (int b = 0; b < nb; ++b)
    for (int k = 0; k < nkx; ++k) {
      const double* Q_ = Qd + ((size_t)b * nkx + k) * n;
      const double* x_ = xref + ((size_t)b * nkx + k) * n;
      double cc = 0.0;
      for (int i = 0; i < n; ++i) { q[((size_t)b * nkx + k) * n + i] = -(Q_[i] * x_[i]); cc += x_[i] * Q_[i] * x_[i]; }
      cc *= 0.5;
      const bool terminal = kz ? (k == 1) : (k == N);
      if (!terminal) {
        const int ku = kz ? 0 : k;
        const double* R_ = Rd + ((size_t)b * nku + ku) * m;
        const double* u_ = uref + ((size_t)b * nku + ku) * m;
        double cu = 0.0;
        for (int i = 0; i < m; ++i) cu += u_[i] * R_[i] * u_[i];
        cc += 0.5 * cu;
      }
      c[(size_t)b * nkx + k] = cc;
    }
  for (int b = 0; b < nb; ++b)
    for (int k = 0; k < nku; ++k)
      for (int i = 0; i < m; ++i)
        r[((size_t)b * nku + k) * m + i] = -(Rd[((size_t)b * nku + k) * m + i] * uref[((size_t)b * nku + k) * m + i]);
  if (h->plan == ALTRO_HIP_PLAN_GENERIC) {   // the diagonal cost as dense blocks (H = 0): what kernels/ilqr_generic.hip evaluates
    std::vector<double> Qf((size_t)nb * nkx * n * n, 0.0), Rf((size_t)nb * nku * m * m, 0.0), Hf((size_t)nb * nku * m * n, 0.0);
    for (size_t t = 0; t < (size_t)nb * nkx; ++t)
      for (int i = 0; i < n; ++i) Qf[t * n * n + i + (size_t)n * i] = Qd[t * n + i];
    for (size_t t = 0; t < (size_t)nb * nku; ++t)
      for (int i = 0; i < m; ++i) Rf[t * m * m + i + (size_t)m * i] = Rd[t * m + i];
    return generic_cost(h, Qf.data(), Rf.data(), Hf.data(), q.data(), r.data(), c.data(), kz, bz);
  }
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {
    // (a) the backward sweep's blocks: lxx = diag(Qd), luu = diag(Rd), lux = 0; lx, lu are refreshed by the loop
    rc = altro_hip_set_cost(h, Qd, Rd, nullptr, q.data(), r.data(), 1, kz, bz);
    if (rc) return rc;
    // (b) the cost parameters the merit function needs: [k][b][36] = Qd | Rd | q | r | c
    const size_t Ez = h->dtype == ALTRO_HIP_F64 ? 8 : 4;
    const int64_t B = h->batch;
    if (!h->m_costp) {
      if ((rc = dmalloc(h, &h->m_costp, (size_t)B * (N + 1) * MF_COSTP * Ez))) return rc;
      if ((rc = dmalloc(h, &h->m_nom, (size_t)B * (N + 1) * MF_NOM * Ez))) return rc;
      HIP_TRY(hipMemsetAsync(h->m_costp, 0, (size_t)B * (N + 1) * MF_COSTP * Ez, h->stream));
      HIP_TRY(hipMemsetAsync(h->m_nom, 0, (size_t)B * (N + 1) * MF_NOM * Ez, h->stream));
    }
    auto put = [&](const double* src, int len, int off, int k0, int nk, int nk_host, int src_off) -> int {
      // records k0..k0+nk-1; with kz the host holds {running, terminal} and src_off selects which
      if (h->dtype == ALTRO_HIP_F64)
        return aos_set<double>(h, (double*)h->m_costp + (size_t)k0 * B * MF_COSTP + off, MF_COSTP, B * MF_COSTP, src, len, nk,
                               kz, bz, nk_host, src_off);
      return aos_set<float>(h, (float*)h->m_costp + (size_t)k0 * B * MF_COSTP + off, MF_COSTP, B * MF_COSTP, src, len, nk, kz,
                            bz, nk_host, src_off);
    };
    rc = put(Qd, n, 0, 0, N, nkx, 0);
    if (!rc) rc = put(Rd, m, 12, 0, N, nku, 0);
    if (!rc) rc = put(q.data(), n, 16, 0, N, nkx, 0);
    if (!rc) rc = put(r.data(), m, 28, 0, N, nku, 0);
    if (!rc) rc = put(c.data(), 1, 32, 0, N, nkx, 0);
    if (!rc && m < MF_M) {   // padded inputs: Rd = 1 keeps their diagonal of Quu positive (they stay at exactly 0: K, d rows are 0)
      const double ones[MF_M] = {1.0, 1.0, 1.0, 1.0};
      rc = h->dtype == ALTRO_HIP_F64
               ? aos_set<double>(h, (double*)h->m_costp + 12 + m, MF_COSTP, B * MF_COSTP, ones, MF_M - m, N, 1, 1, 1, 0)
               : aos_set<float>(h, (float*)h->m_costp + 12 + m, MF_COSTP, B * MF_COSTP, ones, MF_M - m, N, 1, 1, 1, 0);
    }
    // terminal record N: element N of a full host array, or the second entry of a {running, terminal} pair
    auto put_term = [&](const double* src, int len, int off) -> int {
      const double* base = kz ? src : src;   // per problem the host holds nkx knot points
      if (h->dtype == ALTRO_HIP_F64)
        return aos_set<double>(h, (double*)h->m_costp + (size_t)N * B * MF_COSTP + off, MF_COSTP, B * MF_COSTP, base, len, 1, 1,
                               bz, nkx, (kz ? 1 : N) * len);
      return aos_set<float>(h, (float*)h->m_costp + (size_t)N * B * MF_COSTP + off, MF_COSTP, B * MF_COSTP, base, len, 1, 1, bz,
                            nkx, (kz ? 1 : N) * len);
    };
    if (!rc) rc = put_term(Qd, n, 0);
    if (!rc) rc = put_term(q.data(), n, 16);
    if (!rc) rc = put_term(c.data(), 1, 32);
    if (!rc) { h->lqr_cost_set = true; h->ilqr_linear = true; h->cost_dense = false; }
    return rc;
  }
  auto pk = [&](const double* src, int len, int off, int nk, int k_src0, int nk_host, int src_off) -> int {
    return h->dtype == ALTRO_HIP_F64
               ? lane_pack<double>(h, (double*)h->l_cost + (size_t)0, E, src, len, off, 0, nk, k_src0, nk_host, kz, bz, src_off)
               : lane_pack<float>(h, (float*)h->l_cost + (size_t)0, E, src, len, off, 0, nk, k_src0, nk_host, kz, bz, src_off);
  };
  auto pk_term = [&](const double* src, int len, int off, int nk_host) -> int {   // record N of l_cost
    const size_t base = (size_t)N * E * h->batch;
    return h->dtype == ALTRO_HIP_F64
               ? lane_pack<double>(h, (double*)h->l_cost + base, E, src, len, off, 0, 1, kz ? 0 : N, nk_host, kz, bz, kz ? len : 0)
               : lane_pack<float>(h, (float*)h->l_cost + base, E, src, len, off, 0, 1, kz ? 0 : N, nk_host, kz, bz, kz ? len : 0);
  };
  rc = pk(Qd, n, 0, N, 0, nkx, 0);
  if (!rc) rc = pk(Rd, m, n, N, 0, nku, 0);
  if (!rc) rc = pk(q.data(), n, n + m, N, 0, nkx, 0);
  if (!rc) rc = pk(r.data(), m, 2 * n + m, N, 0, nku, 0);
  if (!rc) rc = pk(c.data(), 1, 2 * n + 2 * m, N, 0, nkx, 0);
  if (!rc) rc = pk_term(Qd, n, 0, nkx);
  if (!rc) rc = pk_term(q.data(), n, n + m, nkx);
  if (!rc) rc = pk_term(c.data(), 1, 2 * n + 2 * m, nkx);
  if (!rc) { h->lqr_cost_set = true; h->cost_set = true; h->dyn_set = true; h->is_diag = 0; h->has_f = 0; h->cost_dense = false; }
  return rc;
}

int altro_hip_set_quadratic_cost(altro_hip_batch* h, const double* Q, const double* R, const double* H, const double* q,
                                 const double* r, const double* c, int kz, int bz) {
  // ALTROSolver::SetQuadraticCost (altro_solver.cpp:118-136) -> KnotPointData::SetQuadraticCost (knotpoint_data.cpp:64-85) for the
  // device iLQR loop: the cost 1/2 x'Qx + 1/2 u'Ru + u'Hx + q'x + r'u + c per knot point, stored dense and evaluated as
  // CalcOriginalCost / Gradient / Hessian do (knotpoint_data.cpp:624-634, :659-668, :691-698).
  int rc = loop_entry(h, true);
  if (rc) return rc;
  h->expansion_current = false;
  if (!Q || !R || !H || !q || !r) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "Q, R, H, q, r are required (c may be NULL: zero)");
  if (h->plan == ALTRO_HIP_PLAN_GENERIC) return generic_cost(h, Q, R, H, q, r, c, kz, bz);   // (per-knot-point dimensions: packed arrays)
  const int n = h->n, m = h->m, N = h->N;
  const int nkx = kz ? 2 : N + 1, nku = kz ? 1 : N;
  const size_t Ez = h->esz;
  const int64_t B = h->batch;
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {
    // (a) the backward sweep's blocks: lxx = Q, luu = R, lux = H; lx, lu are refreshed by the loop
    rc = altro_hip_set_cost(h, Q, R, H, q, r, 0, kz, bz);
    if (rc) return rc;
    // (b) the cost's own record, in the COST / TERM layout, for the merit function and the expansion
    if (!h->m_costd) {
      if ((rc = dmalloc(h, &h->m_costd, (size_t)B * (N + 1) * MF_COST * Ez))) return rc;
      if ((rc = dmalloc(h, &h->m_costd_term, (size_t)B * MF_TERM * Ez))) return rc;
      HIP_TRY(hipMemsetAsync(h->m_costd, 0, (size_t)B * (N + 1) * MF_COST * Ez, h->stream));
      HIP_TRY(hipMemsetAsync(h->m_costd_term, 0, (size_t)B * MF_TERM * Ez, h->stream));
    }
    if (!h->m_nom) {
      if ((rc = dmalloc(h, &h->m_nom, (size_t)B * (N + 1) * MF_NOM * Ez))) return rc;
      HIP_TRY(hipMemsetAsync(h->m_nom, 0, (size_t)B * (N + 1) * MF_NOM * Ez, h->stream));
    }
    {
      Dims d{n, m};
      DevSrc dQ, dR, dH, dq, dr;
      rc = put_src(h, Q, d.Q(0), nkx, kz, bz, &dQ);
      if (!rc) rc = put_src(h, R, d.R(0), nku, kz, bz, &dR);
      if (!rc) rc = put_src(h, H, d.H(), nku, kz, bz, &dH);
      if (!rc) rc = put_src(h, q, n, nkx, kz, bz, &dq);
      if (!rc) rc = put_src(h, r, m, nku, kz, bz, &dr);
      SrcArr none{nullptr, 0, 0, 0};
      SrcArr tQ = dQ.s, tq = dq.s;   // terminal views: knot point N, or block 1 of the {running, terminal} pair
      if (kz) { tQ.p += d.Q(0); tq.p += n; }
      if (!rc) rc = mfma16_pack_launch(h, MSEG_Q, dQ.s, none, h->m_costd, h->m_costd_term, 0);
      if (!rc) rc = mfma16_pack_launch(h, MSEG_TERM_Q, tQ, none, h->m_costd, h->m_costd_term, 0);
      if (!rc) rc = mfma16_pack_launch(h, MSEG_HR, dH.s, dR.s, h->m_costd, h->m_costd_term, 0);
      if (!rc) rc = mfma16_pack_launch(h, MSEG_QR, dq.s, dr.s, h->m_costd, h->m_costd_term, 0);
      if (!rc) rc = mfma16_pack_launch(h, MSEG_TERM_q, tq, none, h->m_costd, h->m_costd_term, 0);
      if (!rc) HIP_TRY(hipStreamSynchronize(h->stream));   // (the sources are freed on leaving this block)
    }
    if (!rc && c) {   // c_k into the record's spare slot, k = 0 .. N (record N holds nothing else)
      auto putc = [&](int k0, int nk, int src_off) -> int {
        if (h->dtype == ALTRO_HIP_F64)
          return aos_set<double>(h, (double*)h->m_costd + (size_t)k0 * B * MF_COST + MF_COSTD_C, MF_COST, B * MF_COST, c, 1, nk, kz, bz, nkx, src_off);
        return aos_set<float>(h, (float*)h->m_costd + (size_t)k0 * B * MF_COST + MF_COSTD_C, MF_COST, B * MF_COST, c, 1, nk, kz, bz, nkx, src_off);
      };
      rc = putc(0, N, 0);
      if (!rc) {   // record N: element N of a full host array, or the second entry of a {running, terminal} pair
        if (h->dtype == ALTRO_HIP_F64)
          rc = aos_set<double>(h, (double*)h->m_costd + (size_t)N * B * MF_COST + MF_COSTD_C, MF_COST, B * MF_COST, c, 1, 1, 1, bz, nkx, kz ? 1 : N);
        else
          rc = aos_set<float>(h, (float*)h->m_costd + (size_t)N * B * MF_COST + MF_COSTD_C, MF_COST, B * MF_COST, c, 1, 1, 1, bz, nkx, kz ? 1 : N);
      }
    } else if (!rc) {   // c == NULL: zero (a handle that carried another cost before)
      const double zero = 0.0;
      struct HostPtrs { altro_hip_batch* h; bool keep; ~HostPtrs() { h->dev_ptrs = keep; } } host_ptrs{h, h->dev_ptrs};
      h->dev_ptrs = false;   // (`zero` lives on the host whatever the caller's pointer mode)
      rc = h->dtype == ALTRO_HIP_F64
               ? aos_set<double>(h, (double*)h->m_costd + MF_COSTD_C, MF_COST, B * MF_COST, &zero, 1, N + 1, 1, 1, 1, 0)
               : aos_set<float>(h, (float*)h->m_costd + MF_COSTD_C, MF_COST, B * MF_COST, &zero, 1, N + 1, 1, 1, 1, 0);
    }
    if (!rc) { h->lqr_cost_set = true; h->ilqr_linear = true; h->cost_dense = true; }
    return rc;
  }
  // plan LANE: [k][Q n n | R m m | H m n | q n | r m | c][batch], column-major blocks as given (IlqrDims<n, m, 1>)
  const int E = n * n + m * m + m * n + n + m + 1;
  const int oR = n * n, oH = oR + m * m, oq = oH + m * n, or_ = oq + n, oc = or_ + m;
  if ((uint64_t)B * (uint64_t)E * Ez >= (1ull << 31))
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "plan LANE: batch %d too large for one handle (dense cost rows exceed 2 GiB)", h->batch);
  if (!h->l_costq) {
    if ((rc = dmalloc(h, &h->l_costq, (size_t)B * (N + 1) * E * Ez))) return rc;
    HIP_TRY(hipMemsetAsync(h->l_costq, 0, (size_t)B * (N + 1) * E * Ez, h->stream));
  }
  auto pk = [&](const double* src, int len, int off, int nk_host) -> int {   // knot points 0 .. N - 1  (src == NULL: zeros)
    return h->dtype == ALTRO_HIP_F64 ? lane_pack<double>(h, (double*)h->l_costq, E, src, len, off, 0, N, 0, nk_host, kz, bz, 0)
                                     : lane_pack<float>(h, (float*)h->l_costq, E, src, len, off, 0, N, 0, nk_host, kz, bz, 0);
  };
  auto pk_term = [&](const double* src, int len, int off) -> int {           // record N
    const size_t base = (size_t)N * E * h->batch;
    return h->dtype == ALTRO_HIP_F64
               ? lane_pack<double>(h, (double*)h->l_costq + base, E, src, len, off, 0, 1, kz ? 0 : N, nkx, kz, bz, kz ? len : 0)
               : lane_pack<float>(h, (float*)h->l_costq + base, E, src, len, off, 0, 1, kz ? 0 : N, nkx, kz, bz, kz ? len : 0);
  };
  rc = pk(Q, n * n, 0, nkx);
  if (!rc) rc = pk(R, m * m, oR, nku);
  if (!rc) rc = pk(H, m * n, oH, nku);
  if (!rc) rc = pk(q, n, oq, nkx);
  if (!rc) rc = pk(r, m, or_, nku);
  if (!rc) rc = pk(c, 1, oc, nkx);
  if (!rc) rc = pk_term(Q, n * n, 0);
  if (!rc) rc = pk_term(q, n, oq);
  if (!rc) rc = pk_term(c, 1, oc);
  if (!rc) { h->lqr_cost_set = true; h->cost_set = true; h->dyn_set = true; h->is_diag = 0; h->has_f = 0; h->cost_dense = true; }
  return rc;
}

int altro_hip_set_input_guess(altro_hip_batch* h, const double* u, int kz, int bz) {
  // ALTROSolver::SetInput (altro_solver.cpp:242-251): writes the CANDIDATE inputs u_
  int rc = loop_entry(h, true);
  if (rc) return rc;
  h->expansion_current = false;
  if (!u) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "u == NULL");
  const int n = h->n, m = h->m, N = h->N;
  if (h->plan == ALTRO_HIP_PLAN_GENERIC) {   // candidate inputs: the forward sweep's u array, reference layout
    if (h->ragged) {   // one packed block per problem
      if (kz) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "k_stride_zero needs uniform dimensions");
      const int len = (int)h->g_bstride[G_u];
      rc = h->dtype == ALTRO_HIP_F64 ? generic_set<double>(h, G_u, u, len, 1, 0, bz) : generic_set<float>(h, G_u, u, len, 1, 0, bz);
    } else
    rc = h->dtype == ALTRO_HIP_F64 ? generic_set<double>(h, G_u, u, m, N, kz, bz) : generic_set<float>(h, G_u, u, m, N, kz, bz);
    if (!rc) h->guess_set = true;
    return rc;
  }
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {   // candidate records [k][b][28] = x | y | u
    rc = h->dtype == ALTRO_HIP_F64
             ? aos_set<double>(h, (double*)h->m_xuy + 24, h->m_st.xuy_bs, h->m_st.xuy_ks, u, m, N, kz, bz)
             : aos_set<float>(h, (float*)h->m_xuy + 24, h->m_st.xuy_bs, h->m_st.xuy_ks, u, m, N, kz, bz);
    if (!rc) h->guess_set = true;
    return rc;
  }
  rc = h->dtype == ALTRO_HIP_F64
           ? lane_pack<double>(h, (double*)h->l_xuy, 2 * n + m, u, m, 2 * n, 0, N, 0, kz ? 1 : N, kz, bz)
           : lane_pack<float>(h, (float*)h->l_xuy, 2 * n + m, u, m, 2 * n, 0, N, 0, kz ? 1 : N, kz, bz);
  if (!rc) h->guess_set = true;
  return rc;
}

int altro_hip_set_state_guess(altro_hip_batch* h, const double* x, int kz, int bz) {
  // ALTROSolver::SetState (altro_solver.cpp:229-240): writes the CANDIDATE states x_
  int rc = loop_entry(h, true);
  if (rc) return rc;
  h->expansion_current = false;
  if (!x) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "x == NULL");
  const int n = h->n, m = h->m, N = h->N;
  const bool f64 = h->dtype == ALTRO_HIP_F64;
  if (h->plan == ALTRO_HIP_PLAN_GENERIC && h->ragged) {   // one packed block per problem
    if (kz) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "k_stride_zero needs uniform dimensions");
    const int len = (int)h->g_bstride[G_x];
    return f64 ? generic_set<double>(h, G_x, x, len, 1, 0, bz) : generic_set<float>(h, G_x, x, len, 1, 0, bz);
  }
  if (h->plan == ALTRO_HIP_PLAN_GENERIC)
    return f64 ? generic_set<double>(h, G_x, x, n, N + 1, kz, bz) : generic_set<float>(h, G_x, x, n, N + 1, kz, bz);
  if (h->plan == ALTRO_HIP_PLAN_MFMA16)
    return f64 ? aos_set<double>(h, (double*)h->m_xuy, h->m_st.xuy_bs, h->m_st.xuy_ks, x, n, N + 1, kz, bz)
               : aos_set<float>(h, (float*)h->m_xuy, h->m_st.xuy_bs, h->m_st.xuy_ks, x, n, N + 1, kz, bz);
  return f64 ? lane_pack<double>(h, (double*)h->l_xuy, 2 * n + m, x, n, 0, 0, N + 1, 0, kz ? 1 : N + 1, kz, bz)
             : lane_pack<float>(h, (float*)h->l_xuy, 2 * n + m, x, n, 0, 0, N + 1, 0, kz ? 1 : N + 1, kz, bz);
}

int altro_hip_open_loop_rollout(altro_hip_batch* h) {
  int rc = ilqr_check(h, true);
  if (!rc) rc = ilqr_run(h, IK_ROLLOUT, false, false, 0, 0.0);
  if (!rc) h->forward_done = true;
  return rc;
}
int altro_hip_accept(altro_hip_batch* h) {
  int rc = ilqr_check(h, false);
  if (!rc) rc = ilqr_run(h, IK_ACCEPT, false, false, 0, 0.0);
  return rc;
}
int altro_hip_expand(altro_hip_batch* h) {
  int rc = ilqr_check(h, false);
  if (rc) return rc;
  // CalcExpansions (solver.cpp:189-201) right after a merit evaluation with derivative of an unconstrained problem: the pass left
  // lx, lu (A, B for a device model) of this very candidate, and the cost's Hessian blocks are the constants the setter stored
  // (knotpoint_data.cpp:691-705 copies them afresh each sweep) -- nothing to launch
  if (h->plan == ALTRO_HIP_PLAN_MFMA16 && h->expansion_current && h->al_defs.empty()) return 0;
  return ilqr_run(h, IK_EXPAND, false, false, 0, 0.0);
}
int altro_hip_merit(altro_hip_batch* h, const double* alpha, int alpha_is_uniform, int want_derivative,
                    double* phi, double* dphi) {
  int rc = ilqr_check(h, false);
  if (rc) return rc;
  if (!h->backward_done) return fail(ALTRO_HIP_ERR_NOT_SET, "backward must precede merit");
  if (!alpha || !phi) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "alpha and phi are required");
  if (!alpha_is_uniform)
    HIP_TRY(hipMemcpyAsync(h->i_alpha, alpha, (size_t)h->batch * 8, hipMemcpyHostToDevice, h->stream));
  rc = ilqr_run(h, IK_MERIT, !alpha_is_uniform, false, want_derivative, alpha[0]);
  if (rc) return rc;
  h->forward_done = true;
  HIP_TRY(hipMemcpyAsync(phi, h->i_phi, (size_t)h->batch * 8, hipMemcpyDeviceToHost, h->stream));
  if (want_derivative && dphi)
    HIP_TRY(hipMemcpyAsync(dphi, h->i_dphi, (size_t)h->batch * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}
int altro_hip_stationarity(altro_hip_batch* h, double* out) {
  int rc = ilqr_check(h, false);
  if (rc) return rc;
  rc = ilqr_run(h, IK_STATIONARITY, false, false, 0, 0.0);
  if (rc) return rc;
  std::vector<IlqrProb> pr(h->batch);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(pr.data(), h->i_prob, pr.size() * sizeof(IlqrProb), hipMemcpyDeviceToHost));
  for (int b = 0; b < h->batch; ++b) out[b] = pr[b].stationarity;
  return 0;
}
int altro_hip_feasibility(altro_hip_batch* h, double* out) {
  // SolverImpl::Feasibility (solver.cpp:224-231) of the candidate trajectory
  int rc = ilqr_check(h, false);
  if (rc) return rc;
  if (!out) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "out == NULL");
  rc = ilqr_run(h, IK_STATIONARITY, false, false, 0, 0.0);
  if (rc) return rc;
  std::vector<IlqrProb> pr(h->batch);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(pr.data(), h->i_prob, pr.size() * sizeof(IlqrProb), hipMemcpyDeviceToHost));
  for (int b = 0; b < h->batch; ++b) out[b] = pr[b].feasibility;
  return 0;
}

// ---- MPC receding-horizon operations on the resident batch -----------------
int altro_hip_shift_trajectory(altro_hip_batch* h) {
  int rc = ilqr_check(h, true);
  if (!rc && h->ragged)   // (the reference's ShiftTrajectory copies knot point k + 1 into k: meaningful between equal dimensions only)
    rc = fail(ALTRO_HIP_ERR_UNSUPPORTED, "altro_hip_shift_trajectory needs uniform dimensions");
  if (!rc) rc = ilqr_run(h, IK_SHIFT, false, false, 0, 0.0);
  return rc;
}
int altro_hip_update_linear_costs(altro_hip_batch* h, const double* q, const double* r, const double* c,
                                  int k_first, int k_last, int kz, int bz) {
  // ALTROSolver::UpdateLinearCosts (altro_solver.cpp:266-281) -> KnotPointData::UpdateLinearCosts
  // (knotpoint_data.cpp:193-226) for knot points k_first..k_last (inclusive) of every problem
  int rc = loop_entry(h, true);
  if (rc) return rc;
  h->expansion_current = false;
  if (!h->lqr_cost_set) return fail(ALTRO_HIP_ERR_NOT_SET, "no quadratic cost to update (ErrorCodes::CostNotQuadratic)");
  const int n = h->n, m = h->m, N = h->N;
  if (k_first < 0 || k_last > N || k_first > k_last)
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "knot point range [%d, %d] outside [0, %d] (ErrorCodes::BadIndex)", k_first, k_last, N);
  if (r && k_last == N)
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "cannot update linear input costs at the terminal knot point "
                                            "(ErrorCodes::InvalidOptAtTerminalKnotPoint)");
  const int nk = k_last - k_first + 1;
  if (h->plan == ALTRO_HIP_PLAN_GENERIC && h->ragged) {   // per-knot-point dimensions: q [b][sum over the range of nx[k]], r likewise, c [b][nk]
    if (kz) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "k_stride_zero needs uniform dimensions");
    const GenSizes gs = gen_sizes(h);
    int64_t ox = 0, ou = 0, lx = 0, lu = 0;
    for (int k = 0; k < k_first; ++k) { ox += h->nxv[k]; ou += h->nuv[k]; }
    for (int k = k_first; k <= k_last; ++k) { lx += h->nxv[k]; if (k < N) lu += h->nuv[k]; }
    auto put = [&](void* base, int64_t at, int64_t bs, const double* src, int64_t len) -> int {
      if (!src || len <= 0) return 0;
      char* dst = (char*)base + (size_t)at * h->esz;
      return h->dtype == ALTRO_HIP_F64 ? aos_set<double>(h, (double*)dst, bs, bs, src, (int)len, 1, 0, bz)
                                       : aos_set<float>(h, (float*)dst, bs, bs, src, (int)len, 1, 0, bz);
    };
    rc = put(h->g_cq, ox, gs.sx, q, lx);
    if (!rc) rc = put(h->g_cr, ou, gs.su, r, lu);
    if (!rc) rc = put(h->g_cc, k_first, N + 1, c, nk);
    return rc;
  }
  if (h->plan == ALTRO_HIP_PLAN_GENERIC) {   // the cost's own q, r, c blocks (dense [b][k][block])
    auto put = [&](void* base, int blk, int nk_total, const double* src, int nkp) -> int {
      if (!src || nkp <= 0) return 0;
      char* dst = (char*)base + (size_t)k_first * blk * h->esz;
      return h->dtype == ALTRO_HIP_F64
                 ? aos_set<double>(h, (double*)dst, (int64_t)nk_total * blk, (int64_t)blk, src, blk, nkp, kz, bz, kz ? 1 : nk, 0)
                 : aos_set<float>(h, (float*)dst, (int64_t)nk_total * blk, (int64_t)blk, src, blk, nkp, kz, bz, kz ? 1 : nk, 0);
    };
    rc = put(h->g_cq, n, N + 1, q, nk);
    if (!rc) rc = put(h->g_cr, m, N, r, nk);
    if (!rc) rc = put(h->g_cc, 1, N + 1, c, nk);
    return rc;
  }
  if (h->plan == ALTRO_HIP_PLAN_MFMA16 && h->cost_dense) {
    // the dense cost's record: [q r] at MF_OFF_QR, c in the spare slot; the terminal knot point's q_N lives in costd_term
    const int64_t B = h->batch;
    const int nkr = k_last == N ? nk - 1 : nk;    // running knot points in the range
    auto put = [&](void* base, int64_t bs, int64_t ks, const double* src, int len, int nkp, int src_off) -> int {
      if (!src || nkp <= 0) return 0;
      if (h->dtype == ALTRO_HIP_F64) return aos_set<double>(h, (double*)base, bs, ks, src, len, nkp, kz, bz, kz ? 1 : nk, src_off);
      return aos_set<float>(h, (float*)base, bs, ks, src, len, nkp, kz, bz, kz ? 1 : nk, src_off);
    };
    char* cd = (char*)h->m_costd + (size_t)k_first * B * MF_COST * h->esz;
    rc = put(cd + (size_t)MF_OFF_QR * h->esz, MF_COST, B * MF_COST, q, n, nkr, 0);
    if (!rc) rc = put(cd + (size_t)(MF_OFF_QR + 12) * h->esz, MF_COST, B * MF_COST, r, m, nkr, 0);
    if (!rc) rc = put(cd + (size_t)MF_COSTD_C * h->esz, MF_COST, B * MF_COST, c, 1, nk, 0);
    if (!rc && k_last == N)   // q_N: the last knot point of the host range (or the one shared entry)
      rc = put((char*)h->m_costd_term + (size_t)144 * h->esz, MF_TERM, 0, q, n, 1, kz ? 0 : (nk - 1) * n);
    return rc;
  }
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {
    const int64_t B = h->batch;
    auto put = [&](const double* src, int len, int off) -> int {
      if (!src) return 0;
      if (h->dtype == ALTRO_HIP_F64)
        return aos_set<double>(h, (double*)h->m_costp + (size_t)k_first * B * MF_COSTP + off, MF_COSTP, B * MF_COSTP, src, len,
                               nk, kz, bz);
      return aos_set<float>(h, (float*)h->m_costp + (size_t)k_first * B * MF_COSTP + off, MF_COSTP, B * MF_COSTP, src, len, nk,
                            kz, bz);
    };
    rc = put(q, n, 16);
    if (!rc) rc = put(r, m, 28);
    if (!rc) rc = put(c, 1, 32);
    return rc;
  }
  const bool dense = h->cost_dense;
  const int E = dense ? n * n + m * m + m * n + n + m + 1 : 2 * n + 2 * m + 1;
  const int oq = dense ? n * n + m * m + m * n : n + m;
  void* cost = dense ? h->l_costq : h->l_cost;
  const size_t base = (size_t)k_first * E * h->batch;
  auto pk = [&](const double* src, int len, int off) -> int {
    if (!src) return 0;
    return h->dtype == ALTRO_HIP_F64
               ? lane_pack<double>(h, (double*)cost + base, E, src, len, off, 0, nk, 0, kz ? 1 : nk, kz, bz)
               : lane_pack<float>(h, (float*)cost + base, E, src, len, off, 0, nk, 0, kz ? 1 : nk, kz, bz);
  };
  rc = pk(q, n, oq);
  if (!rc) rc = pk(r, m, oq + n);
  if (!rc) rc = pk(c, 1, oq + n + m);
  return rc;
}
int altro_hip_get_knot(altro_hip_batch* h, int k, double* x, double* u) {
  // ALTROSolver::GetState / GetInput (altro_solver.cpp:323-347) of one knot point for the whole batch:
  // x [batch][n], u [batch][m] (u must be NULL at k = N); per-knot-point dimensions: [batch][nx[k]], [batch][nu[k]]
  int rc = loop_entry(h, true);
  if (rc) return rc;
  const int n = h->n, m = h->m, N = h->N;
  if (k < 0 || k > N) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "knot point %d outside [0, %d] (ErrorCodes::BadIndex)", k, N);
  if (u && k == N) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "no input at the terminal knot point");
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {
    if (!h->m_nom) return fail(ALTRO_HIP_ERR_NOT_SET, "altro_hip_set_tracking_cost has not been called");
    const int64_t B = h->batch;
    if (x) {
      rc = h->dtype == ALTRO_HIP_F64
               ? aos_get<double>(h, x, (const double*)h->m_nom + (size_t)k * B * MF_NOM, MF_NOM, B * MF_NOM, n, 1)
               : aos_get<float>(h, x, (const float*)h->m_nom + (size_t)k * B * MF_NOM, MF_NOM, B * MF_NOM, n, 1);
      if (rc) return rc;
    }
    if (u)
      rc = h->dtype == ALTRO_HIP_F64
               ? aos_get<double>(h, u, (const double*)h->m_nom + (size_t)k * B * MF_NOM + 12, MF_NOM, B * MF_NOM, m, 1)
               : aos_get<float>(h, u, (const float*)h->m_nom + (size_t)k * B * MF_NOM + 12, MF_NOM, B * MF_NOM, m, 1);
    return rc;
  }
  if (h->plan == ALTRO_HIP_PLAN_GENERIC && h->ragged) {
    if (!h->g_xn) return fail(ALTRO_HIP_ERR_NOT_SET, "no cost has been set for the iLQR loop");
    const GenSizes gs = gen_sizes(h);
    int64_t ox = 0, ou = 0;
    for (int j = 0; j < k; ++j) { ox += h->nxv[j]; ou += h->nuv[j]; }
    const int nk = h->nxv[k], mk = k < N ? h->nuv[k] : 0;
    if (x) {
      rc = h->dtype == ALTRO_HIP_F64 ? aos_get<double>(h, x, (const double*)h->g_xn + ox, gs.sx, nk, nk, 1)
                                     : aos_get<float>(h, x, (const float*)h->g_xn + ox, gs.sx, nk, nk, 1);
      if (rc) return rc;
    }
    if (u)
      rc = h->dtype == ALTRO_HIP_F64 ? aos_get<double>(h, u, (const double*)h->g_un + ou, gs.su, mk, mk, 1)
                                     : aos_get<float>(h, u, (const float*)h->g_un + ou, gs.su, mk, mk, 1);
    return rc;
  }
  if (h->plan == ALTRO_HIP_PLAN_GENERIC) {
    if (!h->g_xn) return fail(ALTRO_HIP_ERR_NOT_SET, "no cost has been set for the iLQR loop");
    if (x) {
      rc = h->dtype == ALTRO_HIP_F64 ? aos_get<double>(h, x, (const double*)h->g_xn + (size_t)k * n, (int64_t)(N + 1) * n, n, n, 1)
                                     : aos_get<float>(h, x, (const float*)h->g_xn + (size_t)k * n, (int64_t)(N + 1) * n, n, n, 1);
      if (rc) return rc;
    }
    if (u)
      rc = h->dtype == ALTRO_HIP_F64 ? aos_get<double>(h, u, (const double*)h->g_un + (size_t)k * m, (int64_t)N * m, m, m, 1)
                                     : aos_get<float>(h, u, (const float*)h->g_un + (size_t)k * m, (int64_t)N * m, m, m, 1);
    return rc;
  }
  const size_t E = h->dtype == ALTRO_HIP_F64 ? 8 : 4;
  const char* rec = (const char*)h->l_nom + (size_t)k * (n + m) * h->batch * E;
  if (x) {
    rc = h->dtype == ALTRO_HIP_F64 ? lane_get<double>(h, x, rec, nullptr, n + m, 0, 0, n, 1, 1)
                                   : lane_get<float>(h, x, rec, nullptr, n + m, 0, 0, n, 1, 1);
    if (rc) return rc;
  }
  if (u)
    rc = h->dtype == ALTRO_HIP_F64 ? lane_get<double>(h, u, rec, nullptr, n + m, n, 0, m, 1, 1)
                                   : lane_get<float>(h, u, rec, nullptr, n + m, n, 0, m, 1, 1);
  return rc;
}

int altro_hip_add_linear_constraint(altro_hip_batch* h, int k_first, int k_last, int cone, int p, const double* G,
                                    const double* g, int g_per_problem) {
  // ALTROSolver::SetConstraint (altro_solver.cpp:175-215) for c(x,u) = G [x;u] - g
  int rc = loop_entry(h, true);
  if (rc) return rc;
  h->expansion_current = false;
  if (!G || !g) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "G and g are required");
  if (cone < CONE_EQUALITY || cone > CONE_SOC) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "unknown cone %d", cone);
  // capacities (stated in altro_hip.h): plan GENERIC loops over the blocks with one lane per row, plan LANE unrolls two blocks, plan
  // MFMA16 lays the rows out over slots of eight (kernels/al_types.h: AL_TILE_MAXC)
  const bool gen = h->plan == ALTRO_HIP_PLAN_GENERIC, tile = h->plan == ALTRO_HIP_PLAN_MFMA16;
  const int tile_slots = h->dtype == ALTRO_HIP_F64 ? AL_TILE_MAXC : AL_MAXC;   // (fp32 records: the two-slot kernels only)
  const int pmax = cone == CONE_SOC ? (gen ? GEN_MAXSOC : AL_MAXSOC) : (gen ? GEN_MAXP : tile ? tile_slots * AL_MAXP : AL_MAXP);
  const int cmax = gen ? GEN_MAXC : AL_MAXC, dmax = gen ? GEN_MAXDEF : tile ? AL_TILE_MAXSLOTDEF : AL_MAXDEF;
  if (p < 1 || p > pmax)
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "constraint dimension %d outside [1, %d]%s", p, pmax,
                gen ? "" : cone == CONE_SOC ? " on this plan (ALTRO_HIP_PLAN_GENERIC takes second-order cones of up to 32 rows)"
                                            : " on this plan (ALTRO_HIP_PLAN_GENERIC takes up to 64 rows per block and 8 blocks per knot point)");
  if (k_first < 0 || k_last > h->N || k_first > k_last)
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "knot point range [%d, %d] outside [0, %d] (ErrorCodes::BadIndex)", k_first, k_last, h->N);
  auto slots_of = [&](int cn, int rows) { return cn == CONE_SOC ? 1 : (rows + AL_MAXP - 1) / AL_MAXP; };
  if (tile) {   // slots: per knot point and per handle
    int defslots = slots_of(cone, p);
    for (const AlDef& d0 : h->al_defs) defslots += slots_of(d0.cone, d0.p);
    if (defslots > AL_TILE_MAXSLOTDEF)
      return fail(ALTRO_HIP_ERR_UNSUPPORTED, "at most %d constraint slots (blocks, counted in units of %d rows) per handle on this plan: "
                                             "ALTRO_HIP_PLAN_GENERIC takes 64 blocks", AL_TILE_MAXSLOTDEF, AL_MAXP);
    for (int k = k_first; k <= k_last; ++k) {
      int used = slots_of(cone, p);
      for (int j = 0; j < h->al_knots[k].ncon; ++j) { const AlDef& d0 = h->al_defs[h->al_knots[k].def[j]]; used += slots_of(d0.cone, d0.p); }
      if (used > tile_slots)
        return fail(ALTRO_HIP_ERR_UNSUPPORTED, "at most %d constraint slots of %d rows per knot point on this plan (k = %d would need %d; a block "
                                               "takes ceil(p / %d) of them, a second-order cone one): ALTRO_HIP_PLAN_GENERIC takes 8 blocks of 64 rows",
                    tile_slots, AL_MAXP, k, used, AL_MAXP);
    }
  } else {
    if ((int)h->al_defs.size() >= dmax) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "at most %d constraint blocks", dmax);
    for (int k = k_first; k <= k_last; ++k)
      if (h->al_knots[k].ncon >= cmax)
        return fail(ALTRO_HIP_ERR_UNSUPPORTED, "at most %d constraint blocks per knot point on this plan (k = %d)%s", cmax, k,
                    gen ? "" : ": ALTRO_HIP_PLAN_GENERIC takes 8");
  }
  int w = h->n + h->m;
  if (h->ragged) {   // G is p x (nx[k] + nu[k]): every knot point of the range must have the dimensions of the first (the terminal one: its nx)
    const int nk = h->nxv[k_first], mk = k_first < h->N ? h->nuv[k_first] : 0;   // (a block of the terminal knot point alone: p x nx[N])
    for (int k = k_first; k <= k_last; ++k)
      if (h->nxv[k] != nk || (k < h->N && h->nuv[k] != mk))
        return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "knot points %d and %d differ in dimension: one constraint block takes one [x; u] size "
                                                "(register it per range, like ALTROSolver::SetConstraint per index)", k_first, k);
    w = nk + mk;
  }
  AlDef d{cone, p, g_per_problem ? 1 : 0, (int)h->al_G.size(), 0, 0, w};
  h->al_G.insert(h->al_G.end(), G, G + (size_t)p * w);
  h->al_g.emplace_back(g, g + (size_t)p * (g_per_problem ? h->batch : 1));
  const int id = (int)h->al_defs.size();
  h->al_defs.push_back(d);
  for (int k = k_first; k <= k_last; ++k) {
    AlKnotBig& kn = h->al_knots[k];
    kn.def[kn.ncon++] = id;
  }
  h->al_dirty = true;
  return id;
}
int altro_hip_add_user_constraint(altro_hip_batch* h, int k_first, int k_last, int cone, int p, int id) {
  // ALTROSolver::SetConstraint with a general callback pair (altro_solver.cpp:192-223): value and Jacobian of block `id` come
  // from the source given to altro_hip_set_model_source (altro_user_constraint / altro_user_constraint_jacobian)
  int rc = loop_entry(h);
  if (rc) return rc;
  if (h->plan != ALTRO_HIP_PLAN_LANE || h->model.kind != MODEL_USER || !h->rtc_has_constraints)
    return fail(ALTRO_HIP_ERR_NOT_SET, "altro_hip_set_model_source must come first, with a source that defines "
                                       "altro_user_constraint and altro_user_constraint_jacobian (plan LANE)");
  if (id < 0) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "constraint id %d", id);
  std::vector<double> G((size_t)std::max(p, 0) * (h->n + h->m), 0.0), g((size_t)std::max(p, 0), 0.0);   // placeholders: never read
  rc = altro_hip_add_linear_constraint(h, k_first, k_last, cone, p, G.data(), g.data(), 0);
  if (rc < 0) return rc;
  h->al_defs[rc].user = id + 1;
  h->al_dirty = true;
  return rc;
}
int altro_hip_clear_constraints(altro_hip_batch* h) {
  int rc = loop_entry(h, true);
  if (rc) return rc;
  h->expansion_current = false;
  h->al_defs.clear(); h->al_G.clear(); h->al_g.clear();
  h->al_knots.assign((size_t)h->N + 1, AlKnotBig{});
  h->al_dirty = true;
  return al_upload(h);
}
int altro_hip_reset_duals(altro_hip_batch* h, double penalty) {
  // duals back to zero and every constraint's penalty to `penalty` (what a fresh Initialize leaves: 1)
  int rc = loop_entry(h, true);
  if (rc) return rc;
  h->expansion_current = false;
  if (!(penalty > 0.0)) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "penalty must be positive");
  if ((rc = al_upload(h))) return rc;
  const size_t E = h->dtype == ALTRO_HIP_F64 ? 8 : 4;
  if (h->al_d_z) HIP_TRY(hipMemsetAsync(h->al_d_z, 0, (size_t)h->al_rows * h->batch * E, h->stream));
  std::vector<IlqrProb> pr(h->batch);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(pr.data(), h->i_prob, pr.size() * sizeof(IlqrProb), hipMemcpyDeviceToHost));
  for (auto& q : pr) { q.rho = penalty; q.rho_est = penalty; }
  HIP_TRY(hipMemcpy(h->i_prob, pr.data(), pr.size() * sizeof(IlqrProb), hipMemcpyHostToDevice));
  return 0;
}
int altro_hip_get_duals(altro_hip_batch* h, int k, int slot, double* z) {
  // duals of constraint block `slot` of knot point k, [batch][p]
  int rc = loop_entry(h, true);
  if (rc) return rc;
  if ((rc = al_upload(h))) return rc;
  if (k < 0 || k > h->N || slot < 0 || slot >= h->al_knots[k].ncon || !z)
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "no constraint block %d at knot point %d", slot, k);
  const AlKnotBig& kn = h->al_knots[k];
  const int p = h->al_defs[kn.def[slot]].p;
  const int64_t B = h->batch;
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->dtype == ALTRO_HIP_F64) {
    std::vector<double> t((size_t)p * B);
    HIP_TRY(hipMemcpy(t.data(), (const double*)h->al_d_z + (size_t)kn.z_off[slot] * B, t.size() * 8, hipMemcpyDeviceToHost));
    for (int64_t b = 0; b < B; ++b)
      for (int r = 0; r < p; ++r) z[(size_t)b * p + r] = t[(size_t)r * B + b];
  } else {
    std::vector<float> t((size_t)p * B);
    HIP_TRY(hipMemcpy(t.data(), (const float*)h->al_d_z + (size_t)kn.z_off[slot] * B, t.size() * 4, hipMemcpyDeviceToHost));
    for (int64_t b = 0; b < B; ++b)
      for (int r = 0; r < p; ++r) z[(size_t)b * p + r] = t[(size_t)r * B + b];
  }
  return 0;
}

// per-problem results gathered on the device into the struct's own layout: 56 bytes per problem cross PCIe, through pinned memory
}  // extern "C"
int altro_hip::capi::ilqr_gather_results(altro_hip_batch* h, altro_hip_solve_result* results) {
  static_assert(sizeof(IlqrResult) == sizeof(altro_hip_solve_result), "IlqrResult mirrors altro_hip_solve_result");
  static_assert(sizeof(IlqrPollRec) == sizeof(altro_hip_poll_record), "IlqrPollRec mirrors altro_hip_poll_record");
  int rc;
  const size_t bytes = (size_t)h->batch * sizeof(IlqrResult);
  if (!h->i_results) {
    if ((rc = dmalloc(h, &h->i_results, bytes))) return rc;
    if (hipHostMalloc(&h->i_results_host, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); h->i_results_host = nullptr; }
  }
  if (ilqr_launch_results(h->stream, h->i_prob, (IlqrResult*)h->i_results, h->batch)) return fail(ALTRO_HIP_ERR_HIP, "results kernel launch failed");
  void* stage = h->i_results_host ? h->i_results_host : (void*)results;
  HIP_TRY(hipMemcpyAsync(stage, h->i_results, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (stage != (void*)results) std::memcpy(results, stage, bytes);
  return 0;
}
extern "C" {

int altro_hip_ilqr_solve_async(altro_hip_batch* h, const altro_hip_solve_options* opts) {
  if (!h) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "null handle");
  h->async_request = true;
  const int rc = altro_hip_ilqr_solve(h, opts, nullptr);
  h->async_request = false;
  return rc;
}
int altro_hip_ilqr_poll(altro_hip_batch* h, int* n_done, const altro_hip_poll_record** records) {
  if (!h) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "null handle");
  if (!h->poll_host) return fail(ALTRO_HIP_ERR_NOT_SET, "altro_hip_ilqr_solve_async has not been called");
  if (n_done) *n_done = __atomic_load_n(h->poll_count_host, __ATOMIC_ACQUIRE);
  if (records) *records = (const altro_hip_poll_record*)h->poll_host;
  return 0;
}
int altro_hip_ilqr_wait(altro_hip_batch* h, altro_hip_solve_result* results) {
  int rc = loop_entry(h, true);
  if (rc) return rc;
  if (!h->poll_host) return fail(ALTRO_HIP_ERR_NOT_SET, "altro_hip_ilqr_solve_async has not been called");
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->async_pending) {   // the books of THAT solve: a later synchronous solve, or a second wait, keeps its own counts
    h->async_pending = false;
    int c4[4];
    HIP_TRY(hipMemcpy(c4, h->i_counters, sizeof(c4), hipMemcpyDeviceToHost));
    h->last_sweeps = c4[3];
    h->last_merit_launches = 0;
  }
  if (results) return ilqr_gather_results(h, results);
  return 0;
}

// altro_hip_ilqr_solve: capi_solve.hip

void altro_hip_default_solve_options(altro_hip_solve_options* o) {
  if (!o) return;
  o->iterations_max = 200;            // solver_options.hpp:16-39
  o->tol_stationarity = 1e-4;
  o->tol_primal_feasibility = 1e-4;
  o->tol_meritfun_gradient = 1e-8;
  o->use_backtracking_linesearch = 0;
  o->penalty_initial = 1.0;
  o->penalty_scaling = 10.0;
  o->penalty_max = 1e8;
  o->reg_initial = 0.0;     // the reference: reg = 0, failures ignored (solver.cpp:363, :449)
  o->reg_retry_max = 0;
  o->reg_scale = 10.0;
  o->reg_min = 1e-6;
  o->reg_max = 1e8;
  o->stop_when_running_at_most = 0;   // (extension: every problem to its own end)
  o->forms = 0;
  o->fused_sweeps = 0;
  o->decision_margin = 0.0;   // (the guard of the affine rounds is opt-in: it costs more than the rounds save -- altro_hip.h)
}
int altro_hip_set_forms(altro_hip_batch* h, unsigned forms) {
  int rc = check(h);
  if (rc) return rc;
  if ((forms & ALTRO_HIP_FORM_LANE_QUAD_OFF) && (forms & ALTRO_HIP_FORM_LANE_QUAD_ON))
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "ALTRO_HIP_FORM_LANE_QUAD_OFF and _ON exclude each other");
  if ((forms & ALTRO_HIP_FORM_GENERIC_LATE_Q_OFF) && (forms & ALTRO_HIP_FORM_GENERIC_LATE_Q_ON))
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "ALTRO_HIP_FORM_GENERIC_LATE_Q_OFF and _ON exclude each other");
  if (((h->forms ^ forms) & ALTRO_HIP_FORM_MERIT_ONE_LAUNCH) && h->merit_split >= 0) {   // decided at the first evaluation: decide again
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->i_merit_jk) { (void)hipFree(h->i_merit_jk); h->i_merit_jk = nullptr; }
    if (h->i_spec_jac) { (void)hipFree(h->i_spec_jac); h->i_spec_jac = nullptr; }
    h->merit_split = -1;
  }
  h->forms = forms;
  return 0;
}
unsigned altro_hip_get_forms(const altro_hip_batch* h) { return h ? h->forms : 0u; }
int altro_hip_last_solve_counts(const altro_hip_batch* h, int* sweeps, int* merit_launches) {
  if (!h) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "null handle");
  if (sweeps) *sweeps = h->last_sweeps;
  if (merit_launches) *merit_launches = h->last_merit_launches;
  return 0;
}
int altro_hip_get_nominal(altro_hip_batch* h, double* x, double* u) {
  int rc = loop_entry(h, true);
  if (rc) return rc;
  const int n = h->n, m = h->m, N = h->N;
  if (h->plan == ALTRO_HIP_PLAN_GENERIC && h->ragged) {   // per-knot-point dimensions: packed [batch][sum nx], [batch][sum nu]
    if (!h->g_xn) return fail(ALTRO_HIP_ERR_NOT_SET, "no cost has been set for the iLQR loop");
    const GenSizes gs = gen_sizes(h);
    if (x) {
      rc = h->dtype == ALTRO_HIP_F64 ? aos_get<double>(h, x, (const double*)h->g_xn, gs.sx, gs.sx, (int)gs.sx, 1)
                                     : aos_get<float>(h, x, (const float*)h->g_xn, gs.sx, gs.sx, (int)gs.sx, 1);
      if (rc) return rc;
    }
    if (u)
      rc = h->dtype == ALTRO_HIP_F64 ? aos_get<double>(h, u, (const double*)h->g_un, gs.su, gs.su, (int)gs.su, 1)
                                     : aos_get<float>(h, u, (const float*)h->g_un, gs.su, gs.su, (int)gs.su, 1);
    return rc;
  }
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {
    if (!h->m_nom) return fail(ALTRO_HIP_ERR_NOT_SET, "altro_hip_set_tracking_cost has not been called");
    const int64_t B = h->batch;
    if (x) {
      rc = h->dtype == ALTRO_HIP_F64 ? aos_get<double>(h, x, (const double*)h->m_nom, MF_NOM, B * MF_NOM, n, N + 1)
                                     : aos_get<float>(h, x, (const float*)h->m_nom, MF_NOM, B * MF_NOM, n, N + 1);
      if (rc) return rc;
    }
    if (u)
      rc = h->dtype == ALTRO_HIP_F64 ? aos_get<double>(h, u, (const double*)h->m_nom + 12, MF_NOM, B * MF_NOM, m, N)
                                     : aos_get<float>(h, u, (const float*)h->m_nom + 12, MF_NOM, B * MF_NOM, m, N);
    return rc;
  }
  if (h->plan == ALTRO_HIP_PLAN_GENERIC) {
    if (!h->g_xn) return fail(ALTRO_HIP_ERR_NOT_SET, "no cost has been set for the iLQR loop");
    if (x) {
      rc = h->dtype == ALTRO_HIP_F64 ? aos_get<double>(h, x, (const double*)h->g_xn, (int64_t)(N + 1) * n, n, n, N + 1)
                                     : aos_get<float>(h, x, (const float*)h->g_xn, (int64_t)(N + 1) * n, n, n, N + 1);
      if (rc) return rc;
    }
    if (u)
      rc = h->dtype == ALTRO_HIP_F64 ? aos_get<double>(h, u, (const double*)h->g_un, (int64_t)N * m, m, m, N)
                                     : aos_get<float>(h, u, (const float*)h->g_un, (int64_t)N * m, m, m, N);
    return rc;
  }
  if (x) {
    rc = h->dtype == ALTRO_HIP_F64 ? lane_get<double>(h, x, h->l_nom, nullptr, n + m, 0, 0, n, N + 1, N + 1)
                                   : lane_get<float>(h, x, h->l_nom, nullptr, n + m, 0, 0, n, N + 1, N + 1);
    if (rc) return rc;
  }
  if (u)
    rc = h->dtype == ALTRO_HIP_F64 ? lane_get<double>(h, u, h->l_nom, nullptr, n + m, n, 0, m, N, N)
                                   : lane_get<float>(h, u, h->l_nom, nullptr, n + m, n, 0, m, N, N);
  return rc;
}
// Expansion the backward pass will consume: A | B | lx | lu (reference layout), for parity tests.
int altro_hip_get_expansion(altro_hip_batch* h, double* A, double* B, double* lx, double* lu) {
  int rc = loop_entry(h);
  if (rc) return rc;
  const int n = h->n, m = h->m, N = h->N;
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {   // Z = [A B] rows of the DYN records, [lx lu] of the COST records, lx_N of TERM
    if (h->dev_ptrs) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "altro_hip_get_expansion on plan MFMA16 writes host arrays");
    const int64_t B_ = h->batch;
    std::vector<double> raw;
    auto fetch = [&](const void* src, int64_t bs, int64_t ks, int off, int block, int nk) -> int {
      raw.assign((size_t)B_ * nk * block, 0.0);
      return h->dtype == ALTRO_HIP_F64 ? aos_get<double>(h, raw.data(), (const double*)src + off, bs, ks, block, nk)
                                       : aos_get<float>(h, raw.data(), (const float*)src + off, bs, ks, block, nk);
    };
    if (A || B) {
      if ((rc = fetch(h->m_in, h->m_st.in_bs, h->m_st.in_ks, MF_OFF_Z, 192, N))) return rc;
      for (int64_t b = 0; b < B_; ++b)
        for (int k = 0; k < N; ++k) {
          const double* zr = raw.data() + ((size_t)b * N + k) * 192;
          if (A) for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) A[((size_t)b * N + k) * n * n + i + (size_t)n * j] = zr[i * 16 + j];
          if (B) for (int j = 0; j < m; ++j) for (int i = 0; i < n; ++i) B[((size_t)b * N + k) * n * m + i + (size_t)n * j] = zr[i * 16 + 12 + j];
        }
    }
    if (lx || lu) {
      if ((rc = fetch(h->m_cin, h->m_st.cin_bs, h->m_st.cin_ks, MF_OFF_QR, 16, N))) return rc;
      for (int64_t b = 0; b < B_; ++b)
        for (int k = 0; k < N; ++k) {
          const double* qr = raw.data() + ((size_t)b * N + k) * 16;
          if (lx) for (int i = 0; i < n; ++i) lx[((size_t)b * (N + 1) + k) * n + i] = qr[i];
          if (lu) for (int i = 0; i < m; ++i) lu[((size_t)b * N + k) * m + i] = qr[12 + i];
        }
      if (lx) {
        if ((rc = fetch(h->m_term, MF_TERM, 0, 144, 12, 1))) return rc;
        for (int64_t b = 0; b < B_; ++b)
          for (int i = 0; i < n; ++i) lx[((size_t)b * (N + 1) + N) * n + i] = raw[(size_t)b * 12 + i];
      }
    }
    return 0;
  }
  if (h->plan == ALTRO_HIP_PLAN_GENERIC) {   // reference layout already: A, B as given, lx / lu the backward sweep's q / r
    auto get = [&](double* dst, int arr, int blk, int nk) -> int {
      if (!dst) return 0;
      return h->dtype == ALTRO_HIP_F64 ? generic_get<double>(h, arr, dst, blk, nk) : generic_get<float>(h, arr, dst, blk, nk);
    };
    rc = get(A, G_A, n * n, N);
    if (!rc) rc = get(B, G_B, n * m, N);
    if (!rc) rc = get(lx, G_q, n, N + 1);
    if (!rc) rc = get(lu, G_r, m, N);
    return rc;
  }
  const LaneSizes z = lane_sizes(n, m);
  const int oq = 2 * n * n + 2 * n * m + m * m + n, orr = oq + n;
  auto get = [&](double* dst, int off, int len, int nk, bool with_term, int off_term) -> int {
    if (!dst) return 0;
    return h->dtype == ALTRO_HIP_F64
               ? lane_get<double>(h, dst, h->l_in, with_term ? h->l_term : nullptr, z.e_in, off, off_term, len, nk, N)
               : lane_get<float>(h, dst, h->l_in, with_term ? h->l_term : nullptr, z.e_in, off, off_term, len, nk, N);
  };
  rc = get(A, 0, n * n, N, false, 0);
  if (!rc) rc = get(B, n * n, n * m, N, false, 0);
  if (!rc) rc = get(lx, oq, n, N + 1, true, n * n);
  if (!rc) rc = get(lu, orr, m, N, false, 0);
  return rc;
}

// The line-search state machine driven on the host by a callback: lets CPU-only tests pin it against
// the real reference line search (oracle/_ref) without a GPU.
double altro_hip_linesearch_host(altro_hip_merit_fn f, void* ctx, double alpha0, double phi0, double dphi0,
                                 int try_cubic_first, int use_backtracking, double c1, double c2,
                                 int* status, int* iters, double* phi, double* dphi) {
  LsOptions o = ls_default_options();
  o.try_cubic_first = try_cubic_first; o.use_backtracking = use_backtracking; o.c1 = c1; o.c2 = c2;
  LsState s;
  bool need = ls_begin(s, o, alpha0, phi0, dphi0);
  while (need) {
    double p = 0.0, dp = 0.0;
    f(s.alpha, &p, s.want_derivative ? &dp : nullptr, ctx);
    need = ls_feed(s, o, p, dp);
  }
  if (status) *status = s.status;
  if (iters) *iters = s.n_iters;
  if (phi) *phi = s.phi;
  if (dphi) *dphi = s.dphi;
  return s.alpha;
}

}  // extern "C"
