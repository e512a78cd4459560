// preint_core.h — IMU pre-integration of one interval by one wave64: IntegrationBase::propagate / midPointIntegration
// (VINS_ios/integration_base.h:63-169) with the 15 x 15 matrix work spread over the lanes.
//
// The host restatement (vio_preint.h, host::propagate) is what the estimator runs per IMU sample and sequence; for
// sequences whose windows are assembled on the device (store_core.h) the samples travel instead of the integrated blocks
// and this file integrates them there. Same operations, same order per element -- the dense k-ascending sums of the
// reference; the host code skips structural zeros of F and V, which leaves every sum's bits (a zero term adds nothing) --
// and the same -ffp-contract=off build: the blocks are the host's bit for bit (tests/test_simt_store.py runs this file on
// the SIMT emulator against host::propagate).
#pragma once

#include "solver_core.h"
#include "vio_amd.h"

namespace vio {
namespace preint {

constexpr int kLdsDoubles = 225 + 270 + 225 + 225 + 225 + 225;  // F | V | J | C | FC | Tn per wave
constexpr int kSide = 8;  // doubles kept per interval beside its block: acc_0[3], gyr_0[3] (the last sample), 2 spare

struct State {  // the wave-uniform part of an interval's integration state (every lane holds the same values)
  double acc_0[3], gyr_0[3], ba[3], bg[3];
  double sum_dt, dp[3], dv[3];
  Quat dq;
};

VIO_DEV void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// IntegrationBase(acc_0, gyr_0, ba, bg) (integration_base.h:20-37): J = I, C = 0 in LDS.
VIO_DEV void init_wave(State &s, ldsd lds, int lane, const double acc_0[3], const double gyr_0[3], const double ba[3], const double bg[3]) {
  for (int k = 0; k < 3; k++) s.acc_0[k] = acc_0[k], s.gyr_0[k] = gyr_0[k], s.ba[k] = ba[k], s.bg[k] = bg[k];
  s.sum_dt = 0;
  for (int k = 0; k < 3; k++) s.dp[k] = s.dv[k] = 0;
  s.dq = Quat{0, 0, 0, 1};
  ldsd J = lds + 495, C = lds + 720;
  for (int e = lane; e < 225; e += 64) J[e] = (e % 16 == 0) ? 1.0 : 0.0, C[e] = 0.0;
  wave_sync();
}

// One sample (dt, acc_1, gyr_1). noise: the 18 diagonal entries of integration_base.h:30-36.
VIO_DEV void propagate_wave(State &ib, ldsd lds, int lane, double dt, const double acc_1[3], const double gyr_1[3], const double *noise) {
  ldsd F = lds, V = lds + 225, J = lds + 495, C = lds + 720, FC = lds + 945, Tn = lds + 1170;
  // midpoint rule (integration_base.h:71-81): every lane computes the same scalars
  double a0[3], a1[3], w[3];
  for (int k = 0; k < 3; k++) {
    a0[k] = ib.acc_0[k] - ib.ba[k];
    a1[k] = acc_1[k] - ib.ba[k];
    w[k] = 0.5 * (ib.gyr_0[k] + gyr_1[k]) - ib.bg[k];
  }
  double ua0[3], ua1[3];
  qrot(ib.dq, a0, ua0);
  Quat nq = qmul(ib.dq, Quat{w[0] * dt / 2, w[1] * dt / 2, w[2] * dt / 2, 1.0});
  qrot(nq, a1, ua1);
  double ua[3], np[3], nv[3];
  for (int k = 0; k < 3; k++) {
    ua[k] = 0.5 * (ua0[k] + ua1[k]);
    np[k] = ib.dp[k] + ib.dv[k] * dt + 0.5 * ua[k] * dt * dt;
    nv[k] = ib.dv[k] + ua[k] * dt;
  }
  double Wx[9], A0x[9], A1x[9], R0[9], R1[9], IW[9], R0A0[9], R1A1[9], R1A1IW[9];
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  skew3(w, Wx), skew3(a0, A0x), skew3(a1, A1x);
  qtoR(ib.dq, R0), qtoR(nq, R1);
  for (int i = 0; i < 9; i++) IW[i] = I3[i] - Wx[i] * dt;
  mat3mul(R0, A0x, R0A0), mat3mul(R1, A1x, R1A1), mat3mul(R1A1, IW, R1A1IW);
  // F (15 x 15) and V (15 x 18), integration_base.h:84-131: zero fill, then the 25 non-zero 3 x 3 blocks, one per lane
  for (int e = lane; e < 495; e += 64) lds[e] = 0.0;
  wave_sync();
  {
    double T[9];
    int r0 = -1, c0 = 0, ld = 15;
    double sc = 1.0;
    const double *B = T;
    ldsd M = F;
    switch (lane) {
      case 0: r0 = 0, c0 = 0, B = I3; break;
      case 1:
        for (int i = 0; i < 9; i++) T[i] = -0.25 * R0A0[i] * dt * dt + -0.25 * R1A1IW[i] * dt * dt;
        r0 = 0, c0 = 3;
        break;
      case 2: r0 = 0, c0 = 6, B = I3, sc = dt; break;
      case 3:
        for (int i = 0; i < 9; i++) T[i] = -0.25 * (R0[i] + R1[i]) * dt * dt;
        r0 = 0, c0 = 9;
        break;
      case 4:
        for (int i = 0; i < 9; i++) T[i] = -0.25 * R1A1[i] * dt * dt * -dt;
        r0 = 0, c0 = 12;
        break;
      case 5: r0 = 3, c0 = 3, B = IW; break;
      case 6: r0 = 3, c0 = 12, B = I3, sc = -1.0 * dt; break;
      case 7:
        for (int i = 0; i < 9; i++) T[i] = -0.5 * R0A0[i] * dt + -0.5 * R1A1IW[i] * dt;
        r0 = 6, c0 = 3;
        break;
      case 8: r0 = 6, c0 = 6, B = I3; break;
      case 9:
        for (int i = 0; i < 9; i++) T[i] = -0.5 * (R0[i] + R1[i]) * dt;
        r0 = 6, c0 = 9;
        break;
      case 10:
        for (int i = 0; i < 9; i++) T[i] = -0.5 * R1A1[i] * dt * -dt;
        r0 = 6, c0 = 12;
        break;
      case 11: r0 = 9, c0 = 9, B = I3; break;
      case 12: r0 = 12, c0 = 12, B = I3; break;
      // V
      case 13: M = V, ld = 18, r0 = 0, c0 = 0, B = R0, sc = 0.25 * dt * dt; break;
      case 14:
      case 15:
        for (int i = 0; i < 9; i++) T[i] = 0.25 * -R1A1[i] * dt * dt * 0.5 * dt;
        M = V, ld = 18, r0 = 0, c0 = lane == 14 ? 3 : 9;
        break;
      case 16: M = V, ld = 18, r0 = 0, c0 = 6, B = R1, sc = 0.25 * dt * dt; break;
      case 17: M = V, ld = 18, r0 = 3, c0 = 3, B = I3, sc = 0.5 * dt; break;
      case 18: M = V, ld = 18, r0 = 3, c0 = 9, B = I3, sc = 0.5 * dt; break;
      case 19: M = V, ld = 18, r0 = 6, c0 = 0, B = R0, sc = 0.5 * dt; break;
      case 20:
      case 21:
        for (int i = 0; i < 9; i++) T[i] = 0.5 * -R1A1[i] * dt * 0.5 * dt;
        M = V, ld = 18, r0 = 6, c0 = lane == 20 ? 3 : 9;
        break;
      case 22: M = V, ld = 18, r0 = 6, c0 = 6, B = R1, sc = 0.5 * dt; break;
      case 23: M = V, ld = 18, r0 = 9, c0 = 12, B = I3, sc = dt; break;
      case 24: M = V, ld = 18, r0 = 12, c0 = 15, B = I3, sc = dt; break;
      default: break;
    }
    if (r0 >= 0)
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) M[(r0 + i) * ld + c0 + j] = sc * B[i * 3 + j];  // put33
  }
  wave_sync();
  // jacobian = F jacobian ; covariance = F covariance F^T + V noise V^T (:135-136): lane e handles elements e, e + 64, ...
  double jn[4], cn[4];
  for (int q = 0; q < 4; q++) {
    const int e = lane + 64 * q;
    if (e >= 225) break;
    const int i = e / 15, j = e - 15 * i;
    double sj = 0.0, sc = 0.0, st = 0.0;
    for (int k = 0; k < 15; k++) {
      const double f = F[i * 15 + k];
      sj += f * J[k * 15 + j], sc += f * C[k * 15 + j];
    }
    for (int k = 0; k < 18; k++) {
      const double a = V[i * 18 + k] * noise[k];
      st += a * V[j * 18 + k];
    }
    jn[q] = sj;
    FC[e] = sc, Tn[e] = st;
  }
  wave_sync();
  for (int q = 0; q < 4; q++) {
    const int e = lane + 64 * q;
    if (e >= 225) break;
    const int i = e / 15, j = e - 15 * i;
    double s = 0;
    for (int k = 0; k < 15; k++) s += FC[i * 15 + k] * F[j * 15 + k];
    cn[q] = s + Tn[e];
  }
  wave_sync();  // (every lane has read the old J and C)
  for (int q = 0; q < 4; q++) {
    const int e = lane + 64 * q;
    if (e >= 225) break;
    J[e] = jn[q], C[e] = cn[q];
  }
  wave_sync();
  for (int k = 0; k < 3; k++) ib.dp[k] = np[k], ib.dv[k] = nv[k];
  ib.dq = qnormalized(nq);  // delta_q.normalize() (:164)
  ib.sum_dt += dt;
  for (int k = 0; k < 3; k++) ib.acc_0[k] = acc_1[k], ib.gyr_0[k] = gyr_1[k];
}

// The interval's block as the solver reads it (VioPreintegration, 467 doubles) + the side state.
VIO_DEV void store_wave(const State &s, cldsd lds, int lane, double *blk, double *side) {
  cldsd J = lds + 495, C = lds + 720;
  if (lane == 0) {
    blk[0] = s.sum_dt;
    for (int k = 0; k < 3; k++) blk[1 + k] = s.dp[k], blk[8 + k] = s.dv[k], blk[11 + k] = s.ba[k], blk[14 + k] = s.bg[k];
    blk[4] = s.dq.x, blk[5] = s.dq.y, blk[6] = s.dq.z, blk[7] = s.dq.w;
    for (int k = 0; k < 3; k++) side[k] = s.acc_0[k], side[3 + k] = s.gyr_0[k];
  }
  for (int e = lane; e < 225; e += 64) blk[17 + e] = J[e], blk[242 + e] = C[e];
}
VIO_DEV void load_wave(State &s, ldsd lds, int lane, const double *blk, const double *side) {
  s.sum_dt = blk[0];
  for (int k = 0; k < 3; k++) s.dp[k] = blk[1 + k], s.dv[k] = blk[8 + k], s.ba[k] = blk[11 + k], s.bg[k] = blk[14 + k];
  s.dq = Quat{blk[4], blk[5], blk[6], blk[7]};
  for (int k = 0; k < 3; k++) s.acc_0[k] = side[k], s.gyr_0[k] = side[3 + k];
  ldsd J = lds + 495, C = lds + 720;
  for (int e = lane; e < 225; e += 64) J[e] = blk[17 + e], C[e] = blk[242 + e];
  wave_sync();
}

}  // namespace preint
}  // namespace vio
