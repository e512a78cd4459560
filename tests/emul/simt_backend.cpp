// tests/emul/simt_backend.cpp — TEST-ONLY: the DEVICE sections of solver_core.h / marg_core.h (matrix-core products,
// DPP reductions, v_readlane broadcasts, s_barrier phases) executed on the host by the wave64 SIMT emulator of
// tests/emul/simt.h, one fiber per work-item, through the same pack / make_view / carve / unpack code and the same
// kernel body as vio_backend.hip's vio_window_kernel. Not part of the product library and not a CPU fallback.
#define SIMT_IMPLEMENTATION
#include "simt.h"

#include <algorithm>
#include <limits>
#include <vector>

#include "batch.h"
#include "marg_core.h"

using namespace vio;

// variant: 1 = matrix in "LDS" (vio_window_kernel<true, true>), 2 = the same with the IMU coupling in global scratch
// (<true, false>: the layout of windows with many landmarks), 0 = matrix in global scratch (<false, false>), -1 = what the
// launcher would pick. Returns VIO_ECAP when the requested variant does not fit the CU's LDS.
// how many cooperative shares the band's trailing product is dealt out to (run one after the other by the one emulated workgroup)
extern "C" void simt_set_syrk_shares(int n) { vio::simt_syrk_shares() = n < 1 ? 1 : n; }

extern "C" int simt_solve_window(const VioConfig *cfg, VioWindow *win, VioSolveStats *stats, int nthreads, int variant,
                                 int order) {
  if (nthreads != 256 && nthreads != 512) return VIO_EINVAL;
  bool any_loop = false;
  for (int k = 0; k < win->n_factors; k++)
    if (win->factor_target[k] == win->window_size + 1) any_loop = true;
  HostBatch hb;
  hb.resize(make_dims(*cfg, win->window_size, win->n_features, win->n_factors, any_loop), 1);
  hb.d.lds_asp = variant == 2 ? 0 : 1;
  const bool lds_shape = pose_jp(hb.d) <= 16 * kPanelTiles && variant != 0;
  int rc = pack_window(hb, 0, *win, false, order % 2 ? 0 : stage_chunk_slots(hb.d, lds_shape, nthreads));  // (with and without bucket alignment)
  if (rc != VIO_OK) return rc;
  const BatchStrides &s = hb.s;
  const double kNaN = std::numeric_limits<double>::quiet_NaN();
  std::vector<double> scratch(s.scratch, kNaN), hm(s.hm, kNaN), out_pose(s.out_pose), out_sb(s.out_sb), out_feat(s.out_feat),
      raw_pose(s.out_pose), raw_sb(s.out_sb), raw_feat(s.out_feat), out_loop(7), stats_d(s.stats_d);
  std::vector<int> stats_i(s.stats_i);
  std::vector<double> m_scratch(marg_scratch_doubles(hb.d.Wcap), kNaN), m_x0(9 * kMaxPriorBlocks), m_J((size_t)hb.d.Ncap * hb.d.Ncap),
      m_r(hb.d.Ncap);
  std::vector<int> m_int(4 + 3 * kMaxPriorBlocks);
  BatchPtrs B;
  B.n = 1, B.d = hb.d, B.s = s, B.order = nullptr, B.ptab = nullptr;
  B.hdr = hb.hdr.data(), B.hdr_d = hb.hdr_d.data();
  B.pose = hb.pose.data(), B.sb = hb.sb.data(), B.ex = hb.ex.data(), B.feat = hb.feat.data();
  B.fhost = hb.fhost.data(), B.ftarget = hb.ftarget.data(), B.ffeat = hb.ffeat.data();
  B.fslot = hb.fslot.data(), B.fstart = hb.fstart.data();
  B.pair_h = hb.pair_h.data(), B.pair_t = hb.pair_t.data(), B.pair_s0 = hb.pair_s0.data(), B.pair_s1 = hb.pair_s1.data();
  B.pts_i = hb.pts_i.data(), B.pts_j = hb.pts_j.data(), B.preint = hb.preint.data();
  B.pr_kind = hb.pr_kind.data(), B.pr_index = hb.pr_index.data(), B.pr_offset = hb.pr_offset.data();
  B.pr_x0 = hb.pr_x0.data(), B.pr_J = hb.pr_J.data(), B.pr_r = hb.pr_r.data();
  B.scratch = scratch.data(), B.hm = hm.data();
  B.out_pose = out_pose.data(), B.out_sb = out_sb.data(), B.out_feat = out_feat.data();
  B.raw_pose = raw_pose.data(), B.raw_sb = raw_sb.data(), B.raw_feat = raw_feat.data(), B.out_loop = out_loop.data();
  B.stats_d = stats_d.data(), B.stats_i = stats_i.data();
  B.d.Flds = std::max(1, win->n_features);
  B.d.lds_asp = variant == 2 ? 0 : 1;  // variant 2: the LDS matrix with the IMU coupling in global scratch

  // LDS or global matrix: the launcher's rule (vio_backend.hip backend_upload_impl)
  auto lds_need = [&](bool lds_matrix) {
    size_t se = 0, tail = 0;
    const size_t bs = carve_work<double *>(B.d, lds_matrix, nthreads, nullptr, nullptr, nullptr, nullptr, &se, &tail);
    const size_t bm = (se + tail) * sizeof(double) + carve_marg<double *>(B.d, lds_matrix, nullptr, nullptr, nullptr, 0);
    return std::max(bs, bm + 64 * kMargSlot * sizeof(double));
  };
  bool lds_matrix = pose_jp(B.d) <= 16 * kPanelTiles && lds_need(true) <= kLdsBytes;
  if ((variant == 1 || variant == 2) && !lds_matrix) return VIO_ECAP;
  if (variant == 0) lds_matrix = false;
  if (!lds_matrix && lds_need(false) > kLdsBytes) return VIO_ECAP;
  // (the launcher gives an LDS-variant workgroup half a CU whenever its layout fits there: two windows per CU)
  const size_t lds_bytes = lds_matrix ? (lds_need(true) <= kLdsBytes / 2 ? kLdsBytes / 2 : kLdsBytes) : lds_need(false);
  const size_t lds_doubles = lds_bytes / sizeof(double);
  std::vector<double> lds(lds_doubles + 2, kNaN);

  MargOut mo;
  mo.n = m_int.data(), mo.kind = m_int.data() + 4, mo.index = mo.kind + kMaxPriorBlocks, mo.offset = mo.index + kMaxPriorBlocks;
  mo.x0 = m_x0.data(), mo.J = m_J.data(), mo.r = m_r.data(), mo.scratch = lds_matrix ? nullptr : m_scratch.data(), mo.ncap = hb.d.Ncap;

  // the body of vio_window_kernel (vio_backend.hip), one fiber per work-item
  simt::launch(nthreads, [&](int tid) {
    WinView v = make_view(B, 0);
    const Carved<double *> cw = carve_all<double *>(B.d, lds_matrix, nthreads, lds.data(), hm.data(), v.AspG, lds_doubles);
    WorkT<double *> w = cw.w;
    Ctx cx;
    cx.tid = tid, cx.nt = nthreads, cx.prof = nullptr;
    cx.wrot = order % (nthreads / 64);  // (the device takes it from the hardware wave slot: every rotation must work)
    cx.red = cw.red, cx.lprof = cw.lprof;
    const size_t state_end = cw.state_end_doubles;
    if (lds_matrix && nthreads == 256) solve_window<true, 4>(cx, v, w, SameView{v}, SameWork<decltype(w)>{w});
    else if (lds_matrix) solve_window<true, 8>(cx, v, w, SameView{v}, SameWork<decltype(w)>{w});
    else if (nthreads == 256) solve_window<false, 4>(cx, v, w, SameView{v}, SameWork<decltype(w)>{w});
    else solve_window<false, 8>(cx, v, w, SameView{v}, SameWork<decltype(w)>{w});
    MargWorkT<double *> mw = carve_marg_all<double *>(B.d, lds_matrix, lds.data() + state_end, mo.scratch, lds_doubles - state_end - cw.tail_doubles).m;
    __syncthreads();
    marginalize_window_impl(cx, v, w.xpose, w.xsb, w.xfeat, w.ex, mw, mo);
  }, order);

  unpack_window(s, 0, out_pose.data(), out_sb.data(), out_feat.data(), raw_pose.data(), raw_sb.data(),
                raw_feat.data(), out_loop.data(), stats_d.data(), stats_i.data(), *win, stats);
  if (win->next_prior) unpack_prior(mo, *win->next_prior);
  return VIO_OK;
}
