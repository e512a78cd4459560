// tests/emul/simt.h — TEST-ONLY wave64 SIMT emulator for the device sections of the kernel sources.
//
// The -DVIO_EMUL build (emul_backend.cpp) runs scalar stand-ins of the wave-level sections (matrix cores, DPP,
// v_readlane): index maths and control flow of the phase code, but not one of the sections that actually run on the
// GPU. This header closes that gap: with -DVIO_SIMT the DEVICE sections themselves are compiled for the host and every
// work-item of a workgroup runs as a fiber. A lane runs until it reaches a wave-level operation (v_mfma_f64_16x16x4,
// v_readlane, DPP moves, ballot, wave barrier) or s_barrier; when all 64 lanes of its wave (all waves of the workgroup)
// have arrived the operation is evaluated with the hardware's lane semantics and the lanes continue. A wave whose lanes
// arrive at DIFFERENT operations (divergent control flow around a cross-lane instruction, which the hardware would
// execute with a partial EXEC mask) is reported and aborts; so is a workgroup that can make no progress.
//
// Lanes of a wave run one after the other between two rendezvous points, in an order chosen by SIMT_ORDER
// (forward / reverse / a seeded shuffle per segment): code that relies on the hardware's lockstep without saying so
// (an LDS hand-over between lanes with no wave barrier in between, a missing s_barrier) gives different results under
// different orders, which the tests compare.
//
// Not part of the product; nothing under vins-mobile_amd/ includes it.
#pragma once

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <type_traits>
#include <vector>

namespace simt {

typedef double v4d __attribute__((vector_size(32)));

enum Op { OP_NONE = 0, OP_READLANE, OP_READFIRST, OP_DPP, OP_BALLOT, OP_MFMA, OP_WAVE_BARRIER, OP_SYNC, OP_DONE };

struct Lane {
  void *sp = nullptr;
  char *stack = nullptr;
  int op = OP_NONE;       // what the lane waits for
  int site = 0;           // __LINE__ of the call (diagnostics)
  uint64_t a[4] = {0, 0, 0, 0};
  double d[6] = {0, 0, 0, 0, 0, 0};
  uint64_t r = 0;
  double rd[4] = {0, 0, 0, 0};
};

struct Group {
  int nt = 0;
  std::vector<Lane> lanes;
  void *sched_sp = nullptr;
  int cur = -1;
  std::function<void(int)> body;
  uint64_t n_wave_ops = 0, n_barriers = 0, n_mfma = 0;
};

inline Group *&group() {
  static Group *g = nullptr;
  return g;
}

extern "C" void simt_switch(void **save_sp, void *load_sp);
#ifdef SIMT_IMPLEMENTATION
asm(R"(
.text
.globl simt_switch
.type simt_switch,@function
simt_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size simt_switch,.-simt_switch
)");
#endif

inline int lane_id() { return group()->cur; }

inline void yield_to_scheduler() {
  Group *g = group();
  Lane &l = g->lanes[g->cur];
  simt_switch(&l.sp, g->sched_sp);
}

inline void trampoline() {
  Group *g = group();
  const int me = g->cur;
  g->body(me);
  g->lanes[me].op = OP_DONE;
  yield_to_scheduler();
  abort();  // a finished lane is never resumed
}

inline void die(const char *msg) {
  fprintf(stderr, "simt: %s\n", msg);
  fflush(stderr);
  abort();
}

// ---- operation semantics (wave64, gfx950) ----------------------------------------------------------------------
inline void resolve_wave(Group *g, int w0) {
  Lane *L = &g->lanes[w0];
  const int op = L[0].op;
  for (int i = 1; i < 64; i++)
    if (L[i].op != op || L[i].site != L[0].site) {
      fprintf(stderr, "simt: divergent wave-level operation in wave %d: lane 0 at op %d line %d, lane %d at op %d line %d\n",
              w0 / 64, op, L[0].site, i, L[i].op, L[i].site);
      abort();
    }
  g->n_wave_ops++;
  switch (op) {
    case OP_READLANE: {
      const int src = (int)L[0].a[1] & 63;
      for (int i = 0; i < 64; i++)
        if ((int)(L[i].a[1] & 63) != src) die("v_readlane with a non-uniform lane select");
      const uint64_t v = L[src].a[0];
      for (int i = 0; i < 64; i++) L[i].r = v;
      break;
    }
    case OP_READFIRST: {
      const uint64_t v = L[0].a[0];
      for (int i = 0; i < 64; i++) L[i].r = v;
      break;
    }
    case OP_BALLOT: {
      uint64_t m = 0;
      for (int i = 0; i < 64; i++) m |= (uint64_t)(L[i].a[0] != 0) << i;
      for (int i = 0; i < 64; i++) L[i].r = m;
      break;
    }
    case OP_DPP: {
      // a[0] = old, a[1] = src, a[2] = ctrl, a[3] = row_mask | bank_mask << 4 | bound_ctrl << 8
      uint32_t out[64];
      for (int i = 0; i < 64; i++) {
        const int ctrl = (int)L[i].a[2], row_mask = (int)L[i].a[3] & 15, bank_mask = ((int)L[i].a[3] >> 4) & 15;
        const bool bound = ((int)L[i].a[3] >> 8) & 1;
        const int row = i >> 4, bank = (i >> 2) & 3, in_row = i & 15;
        int src = -1;
        bool valid = true;
        if (ctrl >= 0 && ctrl <= 0xff) {  // quad_perm
          src = (i & ~3) + ((ctrl >> (2 * (i & 3))) & 3);
        } else if (ctrl >= 0x101 && ctrl <= 0x10f) {  // row_shl:n
          const int n = ctrl & 15;
          valid = in_row + n < 16, src = i + n;
        } else if (ctrl >= 0x111 && ctrl <= 0x11f) {  // row_shr:n
          const int n = ctrl & 15;
          valid = in_row >= n, src = i - n;
        } else if (ctrl >= 0x121 && ctrl <= 0x12f) {  // row_ror:n
          const int n = ctrl & 15;
          src = (i & ~15) + ((in_row - n) & 15);
        } else if (ctrl == 0x130) {  // wave_shl:1
          valid = i + 1 < 64, src = i + 1;
        } else if (ctrl == 0x138) {  // wave_shr:1
          valid = i >= 1, src = i - 1;
        } else if (ctrl == 0x140) {  // row_mirror
          src = (i & ~15) + 15 - in_row;
        } else if (ctrl == 0x141) {  // row_half_mirror
          src = (i & ~7) + 7 - (i & 7);
        } else if (ctrl == 0x142) {  // row_bcast:15 -> lane 15 of the previous row
          valid = row >= 1, src = 16 * row - 1;
        } else if (ctrl == 0x143) {  // row_bcast:31 -> lane 31 for rows 2, 3
          valid = row >= 2, src = 31;
        } else {
          die("unsupported DPP control");
        }
        const bool enabled = ((row_mask >> row) & 1) && ((bank_mask >> bank) & 1);
        uint32_t v = (uint32_t)L[i].a[0];
        if (enabled) {
          if (valid) v = (uint32_t)L[src].a[1];
          else if (bound) v = 0;
        }
        out[i] = v;
      }
      for (int i = 0; i < 64; i++) L[i].r = out[i];
      break;
    }
    case OP_MFMA: {
      // v_mfma_f64_16x16x4_f64: lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]; C/D element r of lane l is
      // [(l >> 4) + 4 r][l & 15]
      g->n_mfma++;
      double A[16][4], B[4][16];
      for (int l = 0; l < 64; l++) A[l & 15][l >> 4] = L[l].d[0], B[l >> 4][l & 15] = L[l].d[1];
      for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) {
          const int i = (l >> 4) + 4 * r, j = l & 15;
          double acc = L[l].d[2 + r];
          for (int k = 0; k < 4; k++) acc = fma(A[i][k], B[k][j], acc);
          L[l].rd[r] = acc;
        }
      break;
    }
    case OP_WAVE_BARRIER:
      break;
    default:
      die("unknown wave-level operation");
  }
  for (int i = 0; i < 64; i++) L[i].op = OP_NONE;
}

// Runs body(tid) for tid in [0, nt) as one workgroup. order: 0 forward, 1 reverse, >= 2 seeded shuffle per segment.
inline void launch(int nt, std::function<void(int)> body, int order = -1, size_t stack_bytes = 512 * 1024) {
  if (nt % 64) die("workgroup size must be a multiple of 64");
  if (order < 0) {
    const char *e = getenv("SIMT_ORDER");
    order = e ? atoi(e) : 0;
  }
  Group g;
  g.nt = nt, g.body = body;
  g.lanes.resize(nt);
  Group *prev = group();
  group() = &g;
  for (int i = 0; i < nt; i++) {
    Lane &l = g.lanes[i];
    l.stack = (char *)aligned_alloc(64, stack_bytes);
    if (!l.stack) die("out of memory for fiber stacks");
    uintptr_t top = ((uintptr_t)l.stack + stack_bytes) & ~(uintptr_t)15;
    void **s = (void **)top;
    *--s = nullptr;               // fake return address of the trampoline
    *--s = (void *)&trampoline;   // `ret` of the first switch lands here
    for (int k = 0; k < 6; k++) *--s = nullptr;  // rbp rbx r12-r15
    l.sp = (void *)s;
  }
  const int nw = nt / 64;
  uint64_t rng = 0x9e3779b97f4a7c15ULL * (uint64_t)(order + 1);
  std::vector<int> perm(64);
  int done = 0;
  while (done < nt) {
    bool progress = false;
    for (int w = 0; w < nw; w++) {
      for (int i = 0; i < 64; i++) perm[i] = order == 1 ? 63 - i : i;
      if (order >= 2)
        for (int i = 63; i > 0; i--) {
          rng ^= rng << 13, rng ^= rng >> 7, rng ^= rng << 17;
          std::swap(perm[i], perm[(int)(rng % (uint64_t)(i + 1))]);
        }
      for (int ii = 0; ii < 64; ii++) {
        const int id = 64 * w + perm[ii];
        Lane &l = g.lanes[id];
        if (l.op != OP_NONE) continue;  // waiting or finished
        g.cur = id;
        simt_switch(&g.sched_sp, l.sp);
        progress = true;
        if (l.op == OP_DONE) done++;
      }
      // every lane of the wave now waits: wave-level operation?
      Lane *L = &g.lanes[64 * w];
      bool all_wave = true;
      for (int i = 0; i < 64; i++)
        if (L[i].op == OP_NONE || L[i].op == OP_SYNC || L[i].op == OP_DONE) all_wave = false;
      if (all_wave) resolve_wave(&g, 64 * w), progress = true;
      else {
        // a mix of lanes at a wave-level operation and lanes at s_barrier / finished can never resolve
        bool any_wave = false, any_other = false;
        for (int i = 0; i < 64; i++) {
          if (L[i].op == OP_SYNC || L[i].op == OP_DONE) any_other = true;
          else if (L[i].op != OP_NONE) any_wave = true;
        }
        if (any_wave && any_other) {
          fprintf(stderr, "simt: wave %d is split between a wave-level operation and s_barrier / exit:", w);
          for (int i = 0; i < 64; i++) fprintf(stderr, " %d@%d", L[i].op, L[i].site);
          fprintf(stderr, "\n");
          abort();
        }
      }
    }
    // s_barrier: every unfinished lane waits at it
    int n_sync = 0, n_done = 0;
    for (int i = 0; i < nt; i++) n_sync += g.lanes[i].op == OP_SYNC, n_done += g.lanes[i].op == OP_DONE;
    if (n_sync > 0 && n_sync + n_done == nt) {
      if (n_done) die("s_barrier reached by part of the workgroup while other lanes have exited");
      for (int i = 0; i < nt; i++) g.lanes[i].op = OP_NONE;
      g.n_barriers++;
      progress = true;
    }
    if (!progress) {
      fprintf(stderr, "simt: deadlock; lane states (op@line):");
      for (int i = 0; i < nt; i += 64) fprintf(stderr, " w%d:%d@%d", i / 64, g.lanes[i].op, g.lanes[i].site);
      fprintf(stderr, "\n");
      abort();
    }
  }
  if (getenv("SIMT_STATS"))
    fprintf(stderr, "simt: %llu wave-level ops (%llu mfma), %llu barriers\n", (unsigned long long)g.n_wave_ops,
            (unsigned long long)g.n_mfma, (unsigned long long)g.n_barriers);
  for (int i = 0; i < nt; i++) free(g.lanes[i].stack);
  group() = prev;
}

inline Lane &me() { return group()->lanes[group()->cur]; }

inline uint32_t readlane(uint32_t v, int lane, int site) {
  Lane &l = me();
  l.a[0] = v, l.a[1] = (uint64_t)lane, l.op = OP_READLANE, l.site = site;
  yield_to_scheduler();
  return (uint32_t)l.r;
}
inline uint32_t readfirstlane(uint32_t v, int site) {
  Lane &l = me();
  l.a[0] = v, l.op = OP_READFIRST, l.site = site;
  yield_to_scheduler();
  return (uint32_t)l.r;
}
inline uint32_t dpp(uint32_t old, uint32_t src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl, int site) {
  Lane &l = me();
  l.a[0] = old, l.a[1] = src, l.a[2] = (uint64_t)ctrl, l.a[3] = (uint64_t)(row_mask | bank_mask << 4 | (bound_ctrl ? 256 : 0));
  l.op = OP_DPP, l.site = site;
  yield_to_scheduler();
  return (uint32_t)l.r;
}
inline uint64_t ballot(bool p, int site) {
  Lane &l = me();
  l.a[0] = p, l.op = OP_BALLOT, l.site = site;
  yield_to_scheduler();
  return l.r;
}
inline v4d mfma(double a, double b, v4d c, int site) {
  Lane &l = me();
  l.d[0] = a, l.d[1] = b, l.d[2] = c[0], l.d[3] = c[1], l.d[4] = c[2], l.d[5] = c[3];
  l.op = OP_MFMA, l.site = site;
  yield_to_scheduler();
  v4d r = {l.rd[0], l.rd[1], l.rd[2], l.rd[3]};
  return r;
}
inline void wave_barrier(int site) {
  Lane &l = me();
  l.op = OP_WAVE_BARRIER, l.site = site;
  yield_to_scheduler();
}
inline void syncthreads(int site) {
  Lane &l = me();
  l.op = OP_SYNC, l.site = site;
  yield_to_scheduler();
}
template <class P, class T>
inline auto fetch_add(P p, T v) -> typename std::remove_reference<decltype(*p)>::type {
  auto old = *p;
  *p = old + v;
  return old;
}
inline int d2lo(double v) {
  uint64_t u;
  memcpy(&u, &v, 8);
  return (int)(uint32_t)u;
}
inline int d2hi(double v) {
  uint64_t u;
  memcpy(&u, &v, 8);
  return (int)(uint32_t)(u >> 32);
}
inline double hilo2d(int hi, int lo) {
  uint64_t u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
  double v;
  memcpy(&v, &u, 8);
  return v;
}

}  // namespace simt

// ---- the spellings the kernel sources use -----------------------------------------------------------------------
#define __device__
#define __host__
#define __forceinline__ inline
#define __global__
#define __launch_bounds__(...)
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __HIP_MEMORY_SCOPE_AGENT 1
#define __builtin_amdgcn_readlane(v, lane) ((int)::simt::readlane((uint32_t)(v), (lane), __LINE__))
#define __builtin_amdgcn_readfirstlane(v) ((int)::simt::readfirstlane((uint32_t)(v), __LINE__))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) \
  ((int)::simt::dpp((uint32_t)(old), (uint32_t)(src), (ctrl), (rm), (bm), (bc), __LINE__))
#define __builtin_amdgcn_ballot_w64(p) (::simt::ballot((p), __LINE__))
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) (::simt::mfma((a), (b), (c), __LINE__))
#define __builtin_amdgcn_wave_barrier() (::simt::wave_barrier(__LINE__))
#define __builtin_amdgcn_fence(order, ...) ((void)0)
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_rsq(x) (1.0 / sqrt((double)(x)))
#define __builtin_amdgcn_rcp(x) (1.0 / (double)(x))
#define __hip_atomic_fetch_add(p, v, order, scope) (::simt::fetch_add((p), (v)))
#define __double2loint(v) (::simt::d2lo(v))
#define __double2hiint(v) (::simt::d2hi(v))
#define __hiloint2double(hi, lo) (::simt::hilo2d((hi), (lo)))
#define __syncthreads() (::simt::syncthreads(__LINE__))
#define clock64() (0LL)
