// tests/emul/simt_store.cpp — TEST-ONLY: the device passes of the landmark store (vins-mobile_amd/csrc/store_core.h) executed
// on the host by the SIMT emulator, frame by frame against the host-side list (vio_window.cpp, the restatement of
// FeatureManager the estimator uses) and against pack_window (batch.h) on a seeded stream of frames: every list entry,
// every observation, every depth bit, the keyframe decisions, the factor arrays and the bucket layout must be identical.
// Not part of the product library.
#define SIMT_IMPLEMENTATION
#include "simt.h"

#include <stdarg.h>

#include <random>
#include <string>
#include <vector>

#include "batch.h"
#include "store_core.h"

#include "../../vins-mobile_amd/csrc/vio_window.cpp"

using namespace vio;
namespace st = vio::store;

static_assert(st::PH_F == H_F && st::PH_M == H_M && st::PH_HAS_LOOP == H_HAS_LOOP && st::PH_LOOP_FRAME == H_LOOP_FRAME &&
                  st::PH_MARG == H_MARG && st::PH_NPAIRS == H_NPAIRS && st::PH_NSLOTS == H_NSLOTS && st::PH_NREV == H_NREV,
              "store_core.h writes the header slots of batch.h");

namespace {

struct Msg {
  char *buf;
  int cap, n = 0, fails = 0;
  void add(const char *fmt, ...) {
    fails++;
    if (n >= cap - 1) return;
    va_list ap;
    va_start(ap, fmt);
    n += vsnprintf(buf + n, cap - n, fmt, ap);
    va_end(ap);
    if (n < cap - 1) buf[n++] = '\n', buf[n] = 0;
  }
};

struct SimStore {
  st::Dims d;
  std::vector<int> fid[2], start[2], nobs[2], flag[2], ctl;
  std::vector<double> depth[2], obs[2], ctld;
  std::vector<int> lds_i;
  SimStore(int W, int Lcap, int Ocap) {
    d.W = W, d.Lcap = Lcap, d.Ocap = Ocap;
    for (int b = 0; b < 2; b++) {
      fid[b].assign(Lcap, -7), start[b].assign(Lcap, -7), nobs[b].assign(Lcap, -7), flag[b].assign(Lcap, -7);
      depth[b].assign(Lcap, std::numeric_limits<double>::quiet_NaN());
      obs[b].assign((size_t)Lcap * (W + 1) * 3, std::numeric_limits<double>::quiet_NaN());
    }
    ctl.assign(st::C_COUNT, 0), ctld.assign(st::kCtlDoubles, 0);
    lds_i.assign(st::lds_bytes(d) / sizeof(int) + 16, -1);
  }
  st::Bank bank(int b) { return st::Bank{fid[b].data(), start[b].data(), nobs[b].data(), flag[b].data(), depth[b].data(), obs[b].data()}; }
  st::Lds lds() {
    double *dbase;
    std::fill(lds_i.begin(), lds_i.end(), -1);  // nothing survives in LDS from one launch to the next
    return st::carve_lds<int *, double *>(d, lds_i.data(), &dbase);
  }
};

void rot_xyz(double rx, double ry, double rz, double R[9]) {
  const double ypr[3] = {rz, ry, rx};
  ypr2R(ypr, R);
}

}  // namespace

// Returns the number of mismatches (0 = identical); the first ones are described in msg.
extern "C" int simt_store_fuzz(int seed, int frames, int W, int order, int n_landmarks, char *msg_buf, int msg_cap,
                               long long *stats /* [8] frames, sum F, sum M, keyframes, failures, max list, removed by depth, relocalization factors */) {
  Msg msg{msg_buf, msg_cap};
  if (msg_cap > 0) msg_buf[0] = 0;
  const int P = W + 1;
  std::mt19937_64 rng(seed);
  auto uni = [&](double a, double b) { return a + (b - a) * (double)(rng() >> 11) / 9007199254740992.0; };
  // scene: landmarks in front of a camera that drifts sideways; every landmark is visible during an interval of frames
  struct Lm {
    double X[3];
    int first, last;
  };
  std::vector<Lm> lms(n_landmarks);
  const int total = W + frames + 2;
  for (auto &m : lms) {
    m.X[0] = uni(-6, 12), m.X[1] = uni(-4, 4), m.X[2] = uni(2, 12);
    m.first = (int)uni(-3, total), m.last = m.first + (int)uni(0, 2.2 * W);
  }
  double tic[3] = {0.02, -0.04, 0.01}, ric[9];
  rot_xyz(1.5, -2.0, 88.0, ric);
  auto pose_at = [&](int k, double Pk[3], double Rk[9]) {
    const double speed = seed % 4 == 1 ? 0.012 : 0.11;  // slow streams: most frames are not keyframes (MARGIN_SECOND_NEW)
    Pk[0] = speed * k + 0.03 * speed * sin(0.7 * k), Pk[1] = 0.05 * sin(0.3 * k), Pk[2] = 0.02 * cos(0.5 * k);
    rot_xyz(2.0 * sin(0.21 * k), 1.5 * cos(0.17 * k), 3.0 * sin(0.1 * k), Rk);
  };
  auto observe = [&](int k, std::vector<VioObs> &out) {
    out.clear();
    double Pk[3], Rk[9], Rc[9], tc[3], tmp[3];
    pose_at(k, Pk, Rk);
    mat3mul(Rk, ric, Rc);
    mat3vec(Rk, tic, tmp);
    for (int c = 0; c < 3; c++) tc[c] = Pk[c] + tmp[c];
    for (int i = 0; i < n_landmarks; i++) {
      const Lm &m = lms[i];
      if (k < m.first || k > m.last) continue;
      if ((rng() & 31) == 0) continue;  // a lost track: the id never comes back in this window... it may, later: a new list entry
      double dX[3] = {m.X[0] - tc[0], m.X[1] - tc[1], m.X[2] - tc[2]}, RcT[9], Xc[3];
      mat3T(Rc, RcT);
      mat3vec(RcT, dX, Xc);
      if (Xc[2] < 0.3) continue;
      VioObs o;
      o.id = i, o.x = Xc[0] / Xc[2] + uni(-1e-3, 1e-3), o.y = Xc[1] / Xc[2] + uni(-1e-3, 1e-3), o.z = 1.0;
      out.push_back(o);
    }
    // image_msg reaches the estimator in the tracker's order, not sorted: shuffle
    for (size_t i = out.size(); i > 1; i--) std::swap(out[i - 1], out[rng() % i]);
  };

  vio_features_t *fm = nullptr;
  vio_features_create(W, &fm);
  std::vector<double> Ps(3 * P), Rs(9 * P);
  std::vector<int> frame_of(P);  // scene frame of every window slot
  std::vector<VioObs> ob;
  int k = 0;
  // the window fills on the host list (frame_count 0 .. W-1), sliding is not needed yet
  for (int fc = 0; fc < W; fc++, k++) {
    observe(k, ob);
    int enough = 0, pn = 0, tr = 0;
    if (vio_features_add_check_parallax(fm, fc, ob.data(), (int)ob.size(), &enough, &pn, &tr) != VIO_OK) msg.add("fill: add failed");
    frame_of[fc] = k;
  }
  // promotion: the list goes to the store as it stands
  const int Lcap = 1024, Ocap = 512;
  SimStore S(W, Lcap, Ocap);
  auto load_store = [&]() {
    int n = 0, np = 0;
    vio_features_dump(fm, nullptr, 0, &n, nullptr, 0, &np);
    std::vector<VioFeatureInfo> info(n + 1);
    std::vector<double> pts(3 * (size_t)np + 3);
    vio_features_dump(fm, info.data(), n, &n, pts.data(), np, &np);
    const int b = S.ctl[st::C_BANK];
    int p = 0;
    for (int i = 0; i < n; i++) {
      S.fid[b][i] = info[i].id, S.start[b][i] = info[i].start_frame, S.nobs[b][i] = info[i].n_obs, S.flag[b][i] = info[i].solve_flag;
      S.depth[b][i] = info[i].estimated_depth;
      for (int j = 0; j < info[i].n_obs; j++, p++)
        for (int c = 0; c < 3; c++) S.obs[b][((size_t)i * P + j) * 3 + c] = pts[3 * (size_t)p + c];
    }
    S.ctl[st::C_N] = n;
  };
  load_store();
  auto compare_lists = [&](const char *where, int frame) {
    int n = 0, np = 0;
    vio_features_dump(fm, nullptr, 0, &n, nullptr, 0, &np);
    std::vector<VioFeatureInfo> info(n + 1);
    std::vector<double> pts(3 * (size_t)np + 3);
    vio_features_dump(fm, info.data(), n, &n, pts.data(), np, &np);
    const int b = S.ctl[st::C_BANK];
    if (S.ctl[st::C_N] != n) {
      msg.add("%s frame %d: list length %d, store %d", where, frame, n, S.ctl[st::C_N]);
      return;
    }
    int p = 0;
    for (int i = 0; i < n; i++) {
      if (S.fid[b][i] != info[i].id || S.start[b][i] != info[i].start_frame || S.nobs[b][i] != info[i].n_obs)
        msg.add("%s frame %d entry %d: id %d/%d start %d/%d nobs %d/%d", where, frame, i, info[i].id, S.fid[b][i], info[i].start_frame,
                S.start[b][i], info[i].n_obs, S.nobs[b][i]);
      if (memcmp(&S.depth[b][i], &info[i].estimated_depth, 8) != 0)
        msg.add("%s frame %d entry %d (id %d): depth %.17g, store %.17g", where, frame, i, info[i].id, info[i].estimated_depth, S.depth[b][i]);
      if (S.flag[b][i] != info[i].solve_flag) msg.add("%s frame %d entry %d: solve_flag %d, store %d", where, frame, i, info[i].solve_flag, S.flag[b][i]);
      for (int j = 0; j < info[i].n_obs; j++, p++)
        if (S.nobs[b][i] == info[i].n_obs && memcmp(&S.obs[b][((size_t)i * P + j) * 3], &pts[3 * (size_t)p], 24) != 0)
          msg.add("%s frame %d entry %d obs %d differs", where, frame, i, j);
    }
  };
  compare_lists("load", -1);
  double last_P[3], last_R[9];
  pose_at(k - 1, last_P, last_R);
  memcpy(S.ctld.data(), last_P, 24), memcpy(S.ctld.data() + 3, last_R, 72);

  VioConfig cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.window_size = W, cfg.max_features = 1000, cfg.max_factors = 8192, cfg.max_iterations = 10;
  cfg.fx = 526.6, cfg.gravity = 9.805, cfg.cauchy_a = 1.0;

  for (int f = 0; f < frames && msg.fails < 20; f++, k++) {
    frame_of[W] = k;
    for (int i = 0; i < P; i++) pose_at(frame_of[i], &Ps[3 * i], &Rs[9 * i]);
    observe(k, ob);
    if (f % 7 == 3 && ob.size() > 30) ob.resize(12);  // few tracked landmarks: the keyframe rule's other branch
    // ---- host list
    int enough = 0, pn = 0, tr = 0;
    int rc = vio_features_add_check_parallax(fm, W, ob.data(), (int)ob.size(), &enough, &pn, &tr);
    if (rc != VIO_OK) msg.add("frame %d: host add rc %d", f, rc);
    rc = vio_features_triangulate(fm, Ps.data(), Rs.data(), tic, ric);
    if (rc != VIO_OK) msg.add("frame %d: host triangulate rc %d", f, rc);
    std::vector<double> inv(cfg.max_features);
    int nf = 0, nf2 = 0, m = 0;
    vio_features_get_depth_vector(fm, inv.data(), cfg.max_features, &nf);
    std::vector<int> fh(cfg.max_factors), ft(cfg.max_factors), ff(cfg.max_factors);
    std::vector<double> pi(3 * (size_t)cfg.max_factors), pj(3 * (size_t)cfg.max_factors);
    // every fifth frame carries a relocalization frame: an old keyframe matched to window frame lf saw some of the landmarks
    // (and some ids the window does not know)
    int lf = -1, n_loopf = 0;
    std::vector<int> lids;
    std::vector<double> lxy;
    if (f % 5 == 2) {
      lf = (int)(rng() % W);
      for (int id = 0; id < n_landmarks + 40; id++)
        if (rng() % 3 == 0) lids.push_back(id), lxy.push_back(uni(-0.5, 0.5)), lxy.push_back(uni(-0.5, 0.5));
    }
    vio_features_export_factors_loop(fm, cfg.max_factors, lf, lids.data(), lxy.data(), (int)lids.size(), fh.data(), ft.data(), ff.data(),
                                     pi.data(), pj.data(), &m, &nf2, &n_loopf);
    if (stats) stats[7] += n_loopf;
    // ---- store: pass 1
    {
      st::Bank bk = S.bank(S.ctl[st::C_BANK]);
      st::Lds l = S.lds();
      simt::launch(st::kThreads, [&](int tid) {
        st::Cx cx{tid, st::kThreads};
        st::store_ingest(cx, S.d, bk, S.ctl.data(), l, ob.data(), (int)ob.size(), Ps.data(), Rs.data(), tic, ric);
      }, order);
    }
    if (S.ctl[st::C_STATUS] != VIO_OK) msg.add("frame %d: store status %d", f, S.ctl[st::C_STATUS]);
    if (S.ctl[st::C_MARG] != (enough ? VIO_MARGIN_OLD : VIO_MARGIN_SECOND_NEW) || S.ctl[st::C_TRACK] != tr || S.ctl[st::C_PNUM] != pn)
      msg.add("frame %d: marg %d/%d track %d/%d parallax_num %d/%d", f, enough ? VIO_MARGIN_OLD : VIO_MARGIN_SECOND_NEW, S.ctl[st::C_MARG], tr,
              S.ctl[st::C_TRACK], pn, S.ctl[st::C_PNUM]);
    if (S.ctl[st::C_F] != nf || S.ctl[st::C_M] != m - n_loopf)
      msg.add("frame %d: F %d/%d M %d/%d (without %d relocalization factors)", f, nf, S.ctl[st::C_F], m - n_loopf, S.ctl[st::C_M], n_loopf);
    compare_lists("ingest", f);
    if (stats) stats[0]++, stats[1] += nf, stats[2] += m, stats[3] += enough ? 1 : 0, stats[5] = std::max<long long>(stats[5], S.ctl[st::C_N]);
    // ---- pass 2 against pack_window
    {
      BatchDims bd = make_dims(cfg, W, std::max(nf, 1) + 5, std::max(m, 1) + 9, n_loopf > 0);
      HostBatch hb;
      hb.resize(bd, 1);
      const int chunk = (f % 3 == 0) ? 0 : (f % 3 == 1 ? 96 : 40);
      VioWindow w;
      memset(&w, 0, sizeof(w));
      std::vector<double> pose(7 * P, 0.0), sbv(9 * P, 0.0);
      std::vector<VioPreintegration> pre(W);
      memset(pre.data(), 0, sizeof(VioPreintegration) * W);
      double ex[7] = {0, 0, 0, 0, 0, 0, 1};
      w.window_size = W, w.n_features = nf, w.n_factors = m, w.marginalization_flag = enough ? VIO_MARGIN_OLD : VIO_MARGIN_SECOND_NEW;
      w.pose = pose.data(), w.speed_bias = sbv.data(), w.ex_pose = ex, w.inv_depth = inv.data();
      w.factor_host = fh.data(), w.factor_target = ft.data(), w.factor_feature = ff.data(), w.factor_pts_i = pi.data(), w.factor_pts_j = pj.data();
      w.preint = pre.data(), w.loop_frame = n_loopf > 0 ? lf : -1;
      rc = pack_window(hb, 0, w, false, chunk);
      if (rc != VIO_OK) msg.add("frame %d: pack_window rc %d", f, rc);
      const BatchStrides &s = hb.s;
      std::vector<int> hdr(kHdrInts, -9), o_fhost(s.fint, -9), o_ftarget(s.fint, -9), o_ffeat(s.fint, -9), o_fslot(s.fint, -9), o_fstart(s.fstart, -9),
          o_ph(s.pair, -9), o_pt(s.pair, -9), o_s0(s.pair, -9), o_s1(s.pair, -9);
      std::vector<double> o_feat(s.feat, -9.0), o_pi(s.pts, -9.0), o_pj(s.pts, -9.0);
      st::PackOut o;
      o.hdr = hdr.data(), o.feat = o_feat.data(), o.fhost = o_fhost.data(), o.ftarget = o_ftarget.data(), o.ffeat = o_ffeat.data();
      o.fslot = o_fslot.data(), o.fstart = o_fstart.data(), o.pair_h = o_ph.data(), o.pair_t = o_pt.data(), o.pair_s0 = o_s0.data(), o.pair_s1 = o_s1.data();
      o.pts_i = o_pi.data(), o.pts_j = o_pj.data(), o.Fcap = bd.Fcap, o.Mcap = bd.Mcap, o.pair_cap = bd.pair_cap, o.slot_cap = slot_capacity(bd);
      std::vector<unsigned short> keys(bd.Mcap + 8, 0xffff), own(bd.Mcap + 8, 0xffff);
      std::vector<int> bins(3 * (P + 1) * (P + 1), -1);
      st::Bank bk = S.bank(S.ctl[st::C_BANK]);
      st::Lds l = S.lds();
      const st::LoopIn lp{lf, (int)lids.size(), lids.data(), lxy.data()};
      simt::launch(st::kThreads, [&](int tid) {
        st::Cx cx{tid, st::kThreads};
        st::store_pack(cx, S.d, bk, S.ctl.data(), l, o, chunk, keys.data(), own.data(), bins.data(), lp);
      }, order);
      if (S.ctl[st::C_STATUS] != VIO_OK) msg.add("frame %d: store_pack status %d", f, S.ctl[st::C_STATUS]);
      if (S.ctl[st::C_NLOOP] != n_loopf || S.ctl[st::C_M] != m) msg.add("frame %d: relocalization factors %d/%d, M %d/%d", f, n_loopf, S.ctl[st::C_NLOOP], m, S.ctl[st::C_M]);
      const int *hh = hb.hdr.data();
      const int idx[] = {H_F, H_M, H_HAS_LOOP, H_LOOP_FRAME, H_MARG, H_NPAIRS, H_NSLOTS, H_NREV};
      for (int q : idx)
        if (hdr[q] != hh[q]) msg.add("frame %d: hdr[%d] %d, store %d", f, q, hh[q], hdr[q]);
      auto cmp_i = [&](const char *nm, const int *a, const int *b, int cnt) {
        for (int i = 0; i < cnt; i++)
          if (a[i] != b[i]) {
            msg.add("frame %d: %s[%d] %d, store %d (chunk %d)", f, nm, i, a[i], b[i], chunk);
            return;
          }
      };
      auto cmp_d = [&](const char *nm, const double *a, const double *b, int cnt) {
        if (memcmp(a, b, sizeof(double) * cnt) != 0) msg.add("frame %d: %s differs", f, nm);
      };
      cmp_i("fhost", hb.fhost.data(), o_fhost.data(), m), cmp_i("ftarget", hb.ftarget.data(), o_ftarget.data(), m);
      cmp_i("ffeat", hb.ffeat.data(), o_ffeat.data(), m), cmp_i("fslot", hb.fslot.data(), o_fslot.data(), m);
      cmp_i("fstart", hb.fstart.data(), o_fstart.data(), nf + 1);
      const int np = hh[H_NPAIRS];
      cmp_i("pair_h", hb.pair_h.data(), o_ph.data(), np), cmp_i("pair_t", hb.pair_t.data(), o_pt.data(), np);
      cmp_i("pair_s0", hb.pair_s0.data(), o_s0.data(), np), cmp_i("pair_s1", hb.pair_s1.data(), o_s1.data(), np);
      cmp_d("feat", hb.feat.data(), o_feat.data(), nf), cmp_d("pts_i", hb.pts_i.data(), o_pi.data(), 3 * m), cmp_d("pts_j", hb.pts_j.data(), o_pj.data(), 3 * m);
    }
    // ---- a stand-in for the solve: depths near the list's, a few of them negative; the window states as they are
    std::vector<double> x(std::max(nf, 1)), pose(7 * P), sbv(9 * P, 0.0);
    for (int i = 0; i < nf; i++) x[i] = inv[i] * uni(0.8, 1.25) * ((rng() % 29) == 0 ? -1.0 : 1.0);
    for (int i = 0; i < P; i++) {
      const Quat q = RtoQ(&Rs[9 * i]);
      double *p = &pose[7 * i];
      p[0] = Ps[3 * i], p[1] = Ps[3 * i + 1], p[2] = Ps[3 * i + 2], p[3] = q.x, p[4] = q.y, p[5] = q.z, p[6] = q.w;
    }
    const bool break_it = f == frames / 2 && (seed % 3) == 0;  // a frame the failure detection rejects
    sbv[9 * W + 6] = break_it ? 1.5 : 0.01;
    // ---- host list: setDepth, failureDetection, slide, removeFailures
    vio_features_set_depth(fm, x.data(), nf);
    double Rn[9], R0[9], R1[9];
    qtoR(qfrom_pose(&pose[7 * W]), Rn), qtoR(qfrom_pose(&pose[0]), R0), qtoR(qfrom_pose(&pose[7]), R1);
    int reasons = 0;
    vio_failure_detection(tr, &sbv[9 * W + 6], &pose[7 * W], Rn, last_P, last_R, &reasons);
    if (stats && reasons) stats[4]++;
    if (stats)
      for (int i = 0; i < nf; i++) stats[6] += x[i] < 0;
    if (reasons) {
      vio_features_clear(fm);
    } else {
      if (enough) {
        double mR[9], mP[3], nR[9], nP[3], tt[3];
        mat3mul(R0, ric, mR), mat3mul(R1, ric, nR);
        mat3vec(R0, tic, tt);
        for (int c = 0; c < 3; c++) mP[c] = pose[c] + tt[c];
        mat3vec(R1, tic, tt);
        for (int c = 0; c < 3; c++) nP[c] = pose[7 + c] + tt[c];
        vio_features_remove_back_shift_depth(fm, mR, mP, nR, nP);
      } else {
        vio_features_remove_front(fm, W);
      }
      vio_features_remove_failures(fm);
      memcpy(last_P, &pose[7 * W], 24), memcpy(last_R, Rn, 72);
    }
    // ---- store: pass 3
    {
      const int b = S.ctl[st::C_BANK];
      st::Bank bk = S.bank(b), nb = S.bank(1 - b);
      st::Lds l = S.lds();
      simt::launch(st::kThreads, [&](int tid) {
        st::Cx cx{tid, st::kThreads};
        st::store_finish(cx, S.d, bk, nb, S.ctl.data(), S.ctld.data(), l, x.data(), pose.data(), sbv.data(), tic, ric);
      }, order);
    }
    if (S.ctl[st::C_FAIL] != reasons) msg.add("frame %d: failure reasons %d, store %d", f, reasons, S.ctl[st::C_FAIL]);
    if (reasons) {
      // both sides restart: the window fills again on the host list, then moves to the store
      if (S.ctl[st::C_N] != 0) msg.add("frame %d: store not cleared after a failure", f);
      for (int fc = 0; fc < W; fc++) {
        k++;
        observe(k, ob);
        vio_features_add_check_parallax(fm, fc, ob.data(), (int)ob.size(), &enough, &pn, &tr);
        frame_of[fc] = k;
      }
      load_store();
      pose_at(k, last_P, last_R);
      memcpy(S.ctld.data(), last_P, 24), memcpy(S.ctld.data() + 3, last_R, 72);
      continue;
    }
    compare_lists("finish", f);
    if (memcmp(S.ctld.data(), last_P, 24) != 0 || memcmp(S.ctld.data() + 3, last_R, 72) != 0) msg.add("frame %d: last_P / last_R differ", f);
    // the window slides
    if (enough) {
      for (int i = 0; i < W; i++) frame_of[i] = frame_of[i + 1];
    } else {
      frame_of[W - 1] = frame_of[W];
    }
  }
  vio_features_destroy(fm);
  return msg.fails;
}

// ---- IMU pre-integration on a wave (preint_core.h) against the host restatement (vio_preint.h) -----------------------------
#include "preint_core.h"
#include "vio_preint.h"

// n intervals of random length; every second one is integrated in two pieces (store, load, continue: the non-keyframe merge).
// Returns the number of doubles that differ in any bit.
extern "C" int simt_preint_fuzz(int seed, int n_intervals, int order) {
  std::mt19937_64 rng(seed);
  auto uni = [&](double a, double b) { return a + (b - a) * (double)(rng() >> 11) / 9007199254740992.0; };
  VioConfig cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.acc_n = 0.5, cfg.acc_w = 0.002, cfg.gyr_n = 0.2, cfg.gyr_w = 4.0e-5;
  int bad = 0;
  for (int it = 0; it < n_intervals; it++) {
    const int n = 1 + (int)(rng() % 24), split = (it & 1) ? 1 + (int)(rng() % n) : n;
    double acc0[3], gyr0[3], ba[3], bg[3];
    for (int k = 0; k < 3; k++) acc0[k] = uni(-2, 2) + (k == 2 ? 9.8 : 0), gyr0[k] = uni(-0.5, 0.5), ba[k] = uni(-0.1, 0.1), bg[k] = uni(-0.01, 0.01);
    std::vector<double> dt(n), acc(3 * n), gyr(3 * n);
    for (int i = 0; i < n; i++) {
      dt[i] = uni(0.004, 0.012);
      for (int k = 0; k < 3; k++) acc[3 * i + k] = uni(-2, 2) + (k == 2 ? 9.8 : 0), gyr[3 * i + k] = uni(-0.5, 0.5);
    }
    host::Preint hp;
    host::preint_init(hp, &cfg, acc0, gyr0, ba, bg);
    for (int i = 0; i < n; i++) host::propagate(hp, dt[i], &acc[3 * i], &gyr[3 * i]);
    VioPreintegration want;
    host::preint_export(hp, &want);
    std::vector<double> blk(467, -1.0), side(preint::kSide, -1.0), lds(preint::kLdsDoubles, std::numeric_limits<double>::quiet_NaN());
    auto run = [&](int i0, int i1, bool init) {
      std::fill(lds.begin(), lds.end(), std::numeric_limits<double>::quiet_NaN());  // nothing survives in LDS between launches
      simt::launch(64, [&](int lane) {
        preint::State s;
        if (init) preint::init_wave(s, lds.data(), lane, acc0, gyr0, ba, bg);
        else preint::load_wave(s, lds.data(), lane, blk.data(), side.data());
        for (int i = i0; i < i1; i++) preint::propagate_wave(s, lds.data(), lane, dt[i], &acc[3 * i], &gyr[3 * i], hp.noise);
        preint::store_wave(s, lds.data(), lane, blk.data(), side.data());
      }, order);
    };
    run(0, split, true);
    if (split < n) run(split, n, false);
    static_assert(sizeof(VioPreintegration) == 467 * sizeof(double), "block layout");
    const double *w = reinterpret_cast<const double *>(&want);
    for (int k = 0; k < 467; k++) bad += memcmp(&w[k], &blk[k], 8) != 0;
    for (int k = 0; k < 3; k++) bad += memcmp(&side[k], &hp.acc_0[k], 8) != 0, bad += memcmp(&side[3 + k], &hp.gyr_0[k], 8) != 0;
  }
  return bad;
}
