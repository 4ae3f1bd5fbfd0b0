// ORACLE -- test infrastructure only. C ABI over plb_oracle.hpp (TaichiEnv.step in copy mode, mpm_simulator.py:438-449).
#include <omp.h>

#include "plb_oracle.hpp"

using namespace oracle;

extern "C" {

void* oc_plb_create(int N, int n_grid, int substeps, double dt, const double* gravity, double ground_friction, int n_prim,
                    const double* radius) {
  auto* p = new PlbParams;
  p->N = N; p->n_grid = n_grid; p->substeps = substeps; p->n_prim = n_prim; p->dt = dt;
  p->dx = 1.0 / n_grid; p->inv_dx = (double)n_grid; p->p_vol = (p->dx * 0.5) * (p->dx * 0.5); p->p_mass = p->p_vol;
  for (int d = 0; d < 3; ++d) { p->gravity[d] = gravity[d]; p->lo[d] = 0; p->hi[d] = 1; }
  p->ground_friction = ground_friction;
  for (int i = 0; i < n_prim && i < 2; ++i) p->radius[i] = radius[i];
  return p;
}
void oc_plb_destroy(void* h) { delete (PlbParams*)h; }

// One env.step for B envs: x,v [B,N,3], C,F [B,N,3,3], prim_pos [B,n_prim,3], softness [B,n_prim], action [B,3],
// E, nu, yield_stress [B]
void oc_plb_step(void* h, int B, const double* x, const double* v, const double* C, const double* F, const double* prim_pos,
                 const double* softness, const double* action, const double* E, const double* nu, const double* ys, double* xo,
                 double* vo, double* Co, double* Fo, double* prim_o, int nthreads) {
  const PlbParams& pr = *(PlbParams*)h;
  const int N = pr.N, S = pr.substeps, np = pr.n_prim;
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int b = 0; b < B; ++b) {
    std::vector<double> xa(x + (size_t)b * N * 3, x + (size_t)(b + 1) * N * 3), va(v + (size_t)b * N * 3, v + (size_t)(b + 1) * N * 3);
    std::vector<double> Ca(C + (size_t)b * N * 9, C + (size_t)(b + 1) * N * 9), Fa(F + (size_t)b * N * 9, F + (size_t)(b + 1) * N * 9);
    std::vector<double> xb(N * 3), vb(N * 3), Cb(N * 9), Fb(N * 9), gm, gv;
    std::vector<double> pos(prim_pos + (size_t)b * np * 3, prim_pos + (size_t)(b + 1) * np * 3), pos1(np * 3), pv(np * 3, 0.0);
    for (int d = 0; d < 3; ++d) pv[d] = std::min(std::max(action[b * 3 + d], -1.0), 1.0) * 1.0 / S;   // primitive 0 only (Torus)
    for (int s = 0; s < S; ++s) {
      for (int i = 0; i < np * 3; ++i) pos1[i] = std::max(std::min(pos[i] + pv[i], pr.hi[i % 3]), pr.lo[i % 3]);
      plb_substep(pr, xa.data(), va.data(), Ca.data(), Fa.data(), pos.data(), pos1.data(), softness + b * np, E[b], nu[b], ys[b],
                  xb.data(), vb.data(), Cb.data(), Fb.data(), gm, gv);
      xa.swap(xb); va.swap(vb); Ca.swap(Cb); Fa.swap(Fb); pos.swap(pos1);
    }
    std::memcpy(xo + (size_t)b * N * 3, xa.data(), sizeof(double) * N * 3);
    std::memcpy(vo + (size_t)b * N * 3, va.data(), sizeof(double) * N * 3);
    std::memcpy(Co + (size_t)b * N * 9, Ca.data(), sizeof(double) * N * 9);
    std::memcpy(Fo + (size_t)b * N * 9, Fa.data(), sizeof(double) * N * 9);
    std::memcpy(prim_o + (size_t)b * np * 3, pos.data(), sizeof(double) * np * 3);
  }
}

}  // extern "C"
