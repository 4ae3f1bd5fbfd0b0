// The cloth substep in the reference's LITERAL f32 operation order (cloth_simulator.py:257-337), shared by the kernels of cloth.hip
// (modes 1 / 3, bodies above 1024 particles on one workgroup) and the mode-3 forward of cloth_ref.hip, whose fast path falls back to it.
// Every including file is compiled with -ffp-contract=off and correctly rounded f32 divide / sqrt (hipcc's default): only
// + - * / sqrt, one rounding each -- bit-identical to oracle/csrc/cloth_oracle.hpp::cloth_substep_fwd.
#pragma once
#include "cloth_common.h"

namespace ud {

// per-particle intermediates of one forward substep that the adjoint needs
struct Inter {
  float F1, cF, muF, xV, yV, sV, dm, Ax, Az, sF, zm, nz, R;
  float v3[3], v4[3], x2[3], v5[3];
  bool m0, m1;
};

__device__ __forceinline__ void macro_action(const float* a8, float* act) {  // :168-169
#pragma unroll
  for (int g = 0; g < 2; ++g) {
#pragma unroll
    for (int c = 0; c < 3; ++c) act[g * 4 + c] = clipf(a8[g * 4 + c], -2.0f, 2.0f) * (1.0f / 50.0f);   // "/ 50." under jit = * (1 / 50) (pinned by the demos)
    act[g * 4 + 3] = a8[g * 4 + 3];
  }
}

// One forward substep for particle `i` (own x,v in registers, neighbours' x in LDS plane X).
// Same operation order as oracle/csrc/cloth_oracle.hpp::cloth_substep_fwd (= the reference's).
template <bool KEEP>
__device__ __forceinline__ void substep_fwd(const ClothConst& c, int i, const int* nb, const float* L0,
                                            const float* X, float k, float mu, const float* x, const float* v,
                                            const float* ps, const float* act, float* xo, float* vo, Inter* in, int stride = 0) {
  const int Pp = stride ? stride : c.Pp;   // stride of the LDS planes x | y | z in X
  const float INF = INFINITY;
  float v1[3] = {v[0], v[1] - c.gdt, v[2]};
  float F[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int l = 0; l < 8; ++l) {
    const int j = nb[l];
    const bool ok = j >= 0;
    const int jj = ok ? j : i;
    float r0 = X[jj] - x[0], r1 = X[Pp + jj] - x[1], r2 = X[2 * Pp + jj] - x[2];
    float s = r0 * r0 + r1 * r1 + r2 * r2;
    float len = sqrtf(clipf(s, 1e-12f, INF));
    float L = L0[l];
    float f0 = k * r0 / len * (len - L) / L;
    float f1 = k * r1 / len * (len - L) / L;
    float f2 = k * r2 / len * (len - L) / L;
    F[0] += ok ? f0 : 0.f;
    F[1] += ok ? f1 : 0.f;
    F[2] += ok ? f2 : 0.f;
  }
  F[1] += -c.g;
  const bool fm = x[1] <= c.eps;
  float cF = clipf(F[1], -INF, 0.f);
  float muF = mu * cF * -1.0f;
  float xV = v1[0], yV = v1[2];
  float sV = sqrtf(xV * xV + yV * yV + c.eps);
  float dm = (fm && sV > c.eps) ? 1.f : 0.f;
  float Ax = F[0] - dm * muF * xV / sV;
  float Az = F[2] - dm * muF * yV / sV;
  const bool st = fm && (sV <= c.eps);
  float sF = sqrtf(Ax * Ax + Az * Az + c.eps);
  float zm = (st && muF > sF) ? 1.f : 0.f;
  float Bx = 0.f + (1.f - zm) * Ax, Bz = 0.f + (1.f - zm) * Az;
  float nz = (st && muF <= sF) ? 1.f : 0.f;
  float R = 1.f - muF / sF;
  float Cx = (R * Ax) * nz + Bx * (1.f - nz);
  float Cz = (R * Az) * nz + Bz * (1.f - nz);
  float Ff[3] = {Cx, F[1], Cz};
  float vv[3], xx[3] = {x[0], x[1], x[2]};
#pragma unroll
  for (int a = 0; a < 3; ++a) vv[a] = (v1[a] + Ff[a] * c.dt) * c.damp;
  if (KEEP) {
    in->F1 = F[1]; in->cF = cF; in->muF = muF; in->xV = xV; in->yV = yV; in->sV = sV; in->dm = dm;
    in->Ax = Ax; in->Az = Az; in->sF = sF; in->zm = zm; in->nz = nz; in->R = R;
#pragma unroll
    for (int a = 0; a < 3; ++a) in->v3[a] = vv[a];
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const float* p = ps + g * 4;
    const float* ac = act + g * 4;
    float d0 = xx[0] - p[0], d1 = xx[1] - p[1], d2 = xx[2] - p[2];
    float dist = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    const bool m = dist <= p[3];
    const float suction = ac[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float vs = suction * vv[a];
      float xs = xx[a] + ac[a] * (1.f - suction);
      vv[a] = m ? vs : vv[a];
      xx[a] = m ? xs : xx[a];
    }
    if (KEEP) {
      if (g == 0) { in->m0 = m;
#pragma unroll
        for (int a = 0; a < 3; ++a) in->v4[a] = vv[a];
      } else { in->m1 = m; }
    } else {
      if (g == 0) in->m0 = m; else in->m1 = m;
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (KEEP) { in->x2[a] = xx[a]; in->v5[a] = vv[a]; }
    float xc = clipf(xx[a], 0.f, 1.f);
    float vc = clipf(vv[a], -c.max_v, c.max_v);
    xo[a] = xc + c.dt * vc;
    vo[a] = vc;
  }
}

__device__ __forceinline__ void prim_update(const float* ps, const float* act, float* po) {  // :322-323
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int a = 0; a < 4; ++a) po[g * 4 + a] = clipf(ps[g * 4 + a] + (a < 3 ? act[g * 4 + a] : 0.f), 0.f, 1.f);
}

}  // namespace ud
