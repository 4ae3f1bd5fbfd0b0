// Deterministic MPM forward (ud_mpm_conf.deterministic): the per-element arithmetic, plain C++ shared by the kernels of
// mpm_det.hip and -- compiled by the host compiler, UD_HOST_BUILD -- by the same-order CPU restatement the tests compare them
// with bit for bit (oracle/csrc/mpm_det_host.cpp).
//
// What "deterministic" fixes: the reference's p2g is a scatter-add that XLA's CPU backend applies update by update, in the order of the
// flattened update array (mpm_simulator.py:178-194, :259-274) -- and that array is [27 offsets][N particles] (`offset = idx[:, None, :]
// .repeat(n_particles, axis=1)`, `pos_in_grid.reshape(-1, 3)`): OFFSET-major, so a cell receives its contributions ordered by (offset in
// (i, j, k) order, then particle index).  The fast kernels sum each cell in the arrival order of LDS / memory atomics instead, so two runs
// differ in the last bits.  Here every touched cell is summed by ONE thread in that (offset, particle) order, in f32, and g2p adds its 27
// cells in (i, j, k) order in one lane.  Nothing is accumulated through atomics.  (Rounds 2-3 summed particle-major: also deterministic,
// but not the order of the flattened array; no reference data can tell the two apart -- both sit 1e-6 from the recording.)
// Cost: the particles are bucketed by base cell once per substep (sorted by (base cell, index): det_sort_kernel), so a cell walks the 27
// buckets that can reach it -- for offset (i, j, k) the bucket of base cell (cell - (i, j, k)), its particles in ascending index -- instead
// of all N particles (round 3: 332x the default forward at N = 798).  Particles whose stencil wraps or is cut at the domain edge (negative
// base index, Q9; base + 2 outside `res`) are kept apart ("irregular", bucket key -1) and merged into every offset's walk by index.
// Scope: the forward, position control or soft contact (collide_batch of up to four box / container primitives: mpm_collide.h compiled into
// both builds; its exp is ud_expf's plain-IEEE form there).  The rotation of a primitive goes through the platform's sinf / cosf (ocml vs glibc):
// equal for the non-rotating primitives tested.
// Reference lines as in mpm.hip / mpm_device.h: particle pre-pass :233-258, p2g :259-274, grid op :283-313, g2p :196-221, :318-328,
// forward kinematics primitives.py:185-194, position control primitives.py:232-239.
#pragma once
#include "mpm_device.h"
#include "mpm_collide.h"

namespace ud {

#define UD_DET_PRE 27   // floats per particle handed from the pre-pass to the cell sums: base[3] (int bits), fx[3], w[9], affine[9], v[3]

// forward kinematics of the whole step, in place on the [S][3] / [S][4] rows the caller filled with the input arrays
// (the recurrence of lg_fk_all, mpm_cluster.h)
__host__ __device__ inline void det_fk_rows(int S, const float* action6, float* pp, float* pr) {
  for (int d = 0; d < 3; ++d) {
    const float pva = clipf(action6[d], -1.f, 1.f) * 1.f / (float)S;
    float prev = pp[d];
    pp[d] = clipf(prev, -2.f, 2.f);
    for (int f = 0; f + 1 < S; ++f) {
      const float nxt = clipf(prev + pva, -2.f, 2.f);
      pp[(f + 1) * 3 + d] = nxt;
      prev = nxt;
    }
  }
  float pw[3];
  for (int d = 0; d < 3; ++d) pw[d] = clipf(action6[3 + d], -1.f, 1.f) * 1.f / (float)S;
  const float ang = sqrtf(pw[0] * pw[0] + pw[1] * pw[1] + pw[2] * pw[2]) + 1e-12f;
  const float sn = sinf(ang / 2.f);
  const float q[4] = {cosf(ang / 2.f), pw[0] / ang * sn, pw[1] / ang * sn, pw[2] / ang * sn};
  float r[4] = {pr[0], pr[1], pr[2], pr[3]};
  for (int f = 0; f + 1 < S; ++f) {
    const float o0 = r[0] * q[0] - r[1] * q[1] - r[2] * q[2] - r[3] * q[3];
    const float o1 = r[0] * q[1] + r[1] * q[0] - r[2] * q[3] + r[3] * q[2];
    const float o2 = r[0] * q[2] + r[1] * q[3] + r[2] * q[0] - r[3] * q[1];
    const float o3 = r[0] * q[3] - r[1] * q[2] + r[2] * q[1] + r[3] * q[0];
    const float nn = clipf(sqrtf(o0 * o0 + o1 * o1 + o2 * o2 + o3 * o3), 1e-12f, INFINITY);
    r[0] = o0 / nn; r[1] = o1 / nn; r[2] = o2 / nn; r[3] = o3 / nn;
    float* w = pr + (f + 1) * 4;
    w[0] = r[0]; w[1] = r[1]; w[2] = r[2]; w[3] = r[3];
  }
}

// primitive 0 as the grid op of substep f sees it (load_prim_f, mpm_large.hip)
__host__ __device__ inline void det_prim(int S, int f, const float* pp, const float* pr, const float* psize3, const float* action6,
                                         float friction, PrimF& pf) {
  const int fc = min(max(f, 0), S - 1);
  for (int d = 0; d < 3; ++d) {
    pf.pv[d] = clipf(action6[d], -1.f, 1.f) * 1.f / (float)S;
    pf.pos[d] = pp[fc * 3 + d]; pf.size[d] = psize3[d];
  }
  const float* r = pr + fc * 4;
  const float r0 = r[0], r1 = -r[1], r2 = -r[2], r3 = -r[3];
  const float n = sqrtf(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3) + 1e-12f;
  pf.iq[0] = r0 / n; pf.iq[1] = r1 / n; pf.iq[2] = r2 / n; pf.iq[3] = r3 / n;
  pf.friction = friction;
}

// primitive ip as collide_batch sees it in substep f (load_primc_f, mpm_large.hip): rows f and f + 1 of the FK arrays, clamped (Q5)
__host__ __device__ inline void det_primc(const MpmConst& c, int f, const float* pp, const float* pr, const float* psize3, int ip, PrimC& pc) {
  const int S = c.steps, f0 = min(max(f, 0), S - 1), f1 = min(max(f + 1, 0), S - 1);
  for (int d = 0; d < 3; ++d) { pc.p0[d] = pp[f0 * 3 + d]; pc.p1[d] = pp[f1 * 3 + d]; pc.size[d] = psize3[d]; }
  for (int d = 0; d < 4; ++d) { pc.r0[d] = pr[f0 * 4 + d]; pc.r1[d] = pr[f1 * 4 + d]; }
  pc.soft = c.prim_softness_each[ip]; pc.mu = c.prim_friction_each[ip]; pc.kind = c.sdf_kind;
  primc_finish(pc);
}
// the grid op of one cell (mpm_simulator.py:283-313) from its summed (m, mv): position control (grid_op) or collide_batch of each primitive
// in turn, then ground friction and the boundary (lg_grid_cell, mpm_large.hip).  pp / pr / psize / action: the env's rows, [P][S*3] / [P][S*4] /
// [P][3] / [6 P]
__host__ __device__ inline void det_grid_cell(const MpmConst& c, int f, const float* pp, const float* pr, const float* psize, const float* action,
                                              float friction, int ci, int cj, int ck, float m, const float* mv, float* vo) {
  const int S = c.steps;
  if (c.position_control) {
    PrimF pf;
    det_prim(S, f, pp, pr, psize, action, friction, pf);
    grid_op<false>(c, pf, ci, cj, ck, m, mv, vo, nullptr);
    return;
  }
  float v0[3], v1[3] = {0.f, 0.f, 0.f};
  for (int d = 0; d < 3; ++d) v0[d] = ((m > 0.f) ? mv[d] / m : mv[d]) + c.dtg[d];
  const float gp[3] = {(float)ci * c.dx, (float)cj * c.dx, (float)ck * c.dx};
  for (int ip = 0; ip < c.n_prim; ++ip) {                       // primitive after primitive (mpm_simulator.py:292-294)
    PrimC pc;
    det_primc(c, f, pp + (long)ip * S * 3, pr + (long)ip * S * 4, psize + ip * 3, ip, pc);
    CollideRec cr;
    collide_cell(pc, c.dt, gp, v0, v1, cr);
    for (int d = 0; d < 3; ++d) v0[d] = v1[d];
  }
  grid_tail<false>(c, friction, ci, cj, ck, v1, vo, nullptr);
}

__host__ __device__ inline long det_lin(const MpmConst& c, int key) {
  int ci, cj, ck;
  decode_cell(c, key, ci, cj, ck);
  return ((long)ci * c.res[1] + cj) * c.res[2] + ck;
}

__host__ __device__ inline bool det_regular(const MpmConst& c, int b0, int b1, int b2) {
  return b0 >= 0 && b1 >= 0 && b2 >= 0 && b0 + 2 < c.res[0] && b1 + 2 < c.res[1] && b2 + 2 < c.res[2];
}
__host__ __device__ inline long det_lin3(const MpmConst& c, int ci, int cj, int ck) { return ((long)ci * c.res[1] + cj) * c.res[2] + ck; }

// particle p of one env: state at substep f (SoA record `h`: x, v, C, F rows of Np floats) -> pre[UD_DET_PRE][Np], F of substep
// f + 1 into `hn`.  Returns the particle's bucket key: the linear index of its base cell, or -1 for an irregular particle.
// (`store`: the device runs it in all 32 lanes of a particle's group -- same inputs, same bits -- and lets lane 0 write; q, v: the lanes'
// own copies of what the contributions below are made of)
__host__ __device__ inline int det_pre_particle(const MpmConst& c, const float* h, float* hn, int p, float mu, float la, int material,
                                                float hard, float* pre, bool store, Pre& q, float* v) {
  const int Np = c.Np;
  float x[3], Cm[9], F[9];
  for (int d = 0; d < 3; ++d) { x[d] = h[d * Np + p]; v[d] = h[(3 + d) * Np + p]; }
  for (int d = 0; d < 9; ++d) { Cm[d] = h[(6 + d) * Np + p]; F[d] = h[(15 + d) * Np + p]; }
  particle_pre<false>(c, x, Cm, F, mu, la, material, hard, q, nullptr);
  if (store) {
    if (hn)        // (null: the deterministic backward re-runs the pre-pass on the caller's checkpoint, which it does not write)
      for (int d = 0; d < 9; ++d) hn[(15 + d) * Np + p] = q.Fn[d];
    for (int d = 0; d < 3; ++d) { pre[d * Np + p] = __builtin_bit_cast(float, q.base[d]); pre[(3 + d) * Np + p] = q.fx[d]; pre[(24 + d) * Np + p] = v[d]; }
    for (int d = 0; d < 9; ++d) { pre[(6 + d) * Np + p] = q.w[d]; pre[(15 + d) * Np + p] = q.affine[d]; }
  }
  return det_regular(c, q.base[0], q.base[1], q.base[2]) ? (int)det_lin3(c, q.base[0], q.base[1], q.base[2]) : -1;
}
// what particle p adds to the cell its offset (i, j, k) reaches: (m, mv xyz) -- one update of the reference's scatter-add (:178-194)
__host__ __device__ inline void det_contrib(const MpmConst& c, const Pre& q, const float* v, int i, int j, int k, float* out4) {
  const float weight = q.w[i * 3 + 0] * q.w[j * 3 + 1] * q.w[k * 3 + 2];
  const float dpos[3] = {((float)i - q.fx[0]) * c.dx, ((float)j - q.fx[1]) * c.dx, ((float)k - q.fx[2]) * c.dx};
  out4[0] = weight * c.p_mass;
  for (int r = 0; r < 3; ++r)
    out4[1 + r] = weight * (c.p_mass * v[r] + (q.affine[r * 3] * dpos[0] + q.affine[r * 3 + 1] * dpos[1] + q.affine[r * 3 + 2] * dpos[2]));
}
// the cells particle p's scatter or gather will touch, offset cidx = 0 .. 26: the scatter cell (or -1: dropped) and the gather cell (a clamped
// gather may read a cell nobody scatters to) -- the callers stamp both into the env's touched-cell set
__host__ __device__ inline void det_touch_base(const MpmConst& c, int b0, int b1, int b2, int cidx, long& sc_lin, long& gc_lin) {
  const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
  const int sc = cell_scatter(c, b0 + i, b1 + j, b2 + k);
  sc_lin = sc >= 0 ? det_lin(c, sc) : -1;
  gc_lin = det_lin(c, cell_gather(c, b0 + i, b1 + j, b2 + k));
}
__host__ __device__ inline void det_touch(const MpmConst& c, const float* pre, int p, int cidx, long& sc_lin, long& gc_lin) {
  const int Np = c.Np;
  const int b0 = __builtin_bit_cast(int, pre[p]), b1 = __builtin_bit_cast(int, pre[Np + p]), b2 = __builtin_bit_cast(int, pre[2 * Np + p]);
  const int i = cidx / 9, j = (cidx / 3) % 3, k = cidx % 3;
  const int sc = cell_scatter(c, b0 + i, b1 + j, b2 + k);
  sc_lin = sc >= 0 ? det_lin(c, sc) : -1;
  gc_lin = det_lin(c, cell_gather(c, b0 + i, b1 + j, b2 + k));
}

struct DetRange { int s, e; };
// one env's particle buckets of one substep: `order` = the particle indices sorted by (bucket key, index) -- the n_irr irregular ones (key -1)
// first; brange[base cell] = [s, e) of that cell's bucket in `order`, valid where bflag[base cell] == epoch
struct DetBuckets { const int* order; const DetRange* brange; const int* bflag; int n_irr, epoch; };

// one touched cell: (m, mv) summed in the order of the reference's flattened update array -- offsets in (i, j, k) order, the particles of an
// offset in ascending index -- then the grid op.  contrib: [27][Np][4], det_contrib of every (offset, particle).
struct DetPrimRows { int f; const float *pp, *pr, *psize, *action; float friction; };   // the env's primitive rows for det_grid_cell
__host__ __device__ inline void det_cell(const MpmConst& c, int ci, int cj, int ck, const float* pre, const float* contrib, const DetPrimRows& pw,
                                         const DetBuckets& bk, float* vo) {
  const int Np = c.Np, key = ci | (cj << 10) | (ck << 20);
  float m = 0.f, mv[3] = {0.f, 0.f, 0.f};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) {
        int bs = 0, be = 0;                          // the regular particles whose base cell is cell - (i, j, k)
        const int c0 = ci - i, c1 = cj - j, c2 = ck - k;
        if (det_regular(c, c0, c1, c2)) {
          const long bl = det_lin3(c, c0, c1, c2);
          if (bk.bflag[bl] == bk.epoch) { bs = bk.brange[bl].s; be = bk.brange[bl].e; }
        }
        int ir = 0;                                  // merged, by index, with the irregular particles that reach this cell with this offset
        while (bs < be || ir < bk.n_irr) {
          const int pb = bs < be ? bk.order[bs] : 0x7fffffff, pi = ir < bk.n_irr ? bk.order[ir] : 0x7fffffff;
          int p;
          if (pb < pi) { p = pb; ++bs; }
          else {
            p = pi; ++ir;
            const int b0 = __builtin_bit_cast(int, pre[p]), b1 = __builtin_bit_cast(int, pre[Np + p]), b2 = __builtin_bit_cast(int, pre[2 * Np + p]);
            if (cell_scatter(c, b0 + i, b1 + j, b2 + k) != key) continue;
          }
          const float* q4 = contrib + ((long)(i * 9 + j * 3 + k) * Np + p) * 4;
          m += q4[0];
          for (int r = 0; r < 3; ++r) mv[r] += q4[1 + r];
        }
      }
  det_grid_cell(c, pw.f, pw.pp, pw.pr, pw.psize, pw.action, pw.friction, ci, cj, ck, m, mv, vo);
}

// particle p: gather of the 27 cell velocities in (i, j, k) order, advection; x, v, C of substep f + 1 into `hn`.
// vel: [G][4] floats per env (v in .x .y .z).  Returns the particle's Q6 row sum (used for rows 0..2 only).
__host__ __device__ inline float det_g2p_particle(const MpmConst& c, const float* h, float* hn, int p, const float* pre, const float* vel) {
  const int Np = c.Np;
  const int b0 = __builtin_bit_cast(int, pre[p]), b1 = __builtin_bit_cast(int, pre[Np + p]), b2 = __builtin_bit_cast(int, pre[2 * Np + p]);
  float fx[3], w[9];
  for (int d = 0; d < 3; ++d) fx[d] = pre[(3 + d) * Np + p];
  for (int d = 0; d < 9; ++d) w[d] = pre[(6 + d) * Np + p];
  float nv[3] = {0.f, 0.f, 0.f}, nC[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) {
        const float* g = vel + det_lin(c, cell_gather(c, b0 + i, b1 + j, b2 + k)) * 4;
        const float weight = w[i * 3 + 0] * w[j * 3 + 1] * w[k * 3 + 2];
        const float dp[3] = {(float)i - fx[0], (float)j - fx[1], (float)k - fx[2]};
        for (int r = 0; r < 3; ++r) {
          nv[r] += weight * g[r];
          for (int s2 = 0; s2 < 3; ++s2) nC[r * 3 + s2] += 4.f * weight * (g[r] * dp[s2]) * c.inv_dx;
        }
      }
  for (int d = 0; d < 3; ++d) { hn[d * Np + p] = h[d * Np + p] + c.dt * nv[d]; hn[(3 + d) * Np + p] = nv[d]; }
  for (int d = 0; d < 9; ++d) hn[(6 + d) * Np + p] = nC[d];
  return (p == 0) ? nC[0] + nC[1] + nC[2] : ((p == 1) ? nC[3] + nC[4] + nC[5] : nC[6] + nC[7] + nC[8]);
}

}  // namespace ud
