// TEST INFRASTRUCTURE.  The same-order checker of the deterministic MPM forward (ud_mpm_conf.deterministic): the device's own
// per-element source -- unidom_amd/csrc/mpm_det.h, mpm_device.h and (soft contact) mpm_collide.h -- compiled by the host compiler (UD_HOST_BUILD: no HIP, IEEE
// arithmetic, -ffp-contract=off) and driven by plain loops in the order the kernels of mpm_det.hip define: per substep every
// particle's pre-pass, the particles bucketed by base cell (sorted by (bucket key, index)), every touched cell summed over the 27
// offsets in (i, j, k) order and each offset's bucket in ascending particle index, every particle's gather in (i, j, k) order.
// It pins ORDER and ARITHMETIC of the GPU run bit for bit; that this arithmetic is the reference's algorithm is what the
// independent restatement (mpm_oracle.hpp, mpm_simulator.py:178-328) is for -- tests/test_mpm_det.py holds the two against each
// other on the CPU, within the tolerance the different SVD and summation order leave.
#define UD_HOST_BUILD 1
#include <algorithm>
#include <cstring>
#include <utility>
#include <vector>

#include "../../unidom_amd/csrc/mpm_det.h"

extern "C" {

// x, v [B][N][3], C, F [B][N][3][3] -> the same after `steps` substeps; ppos [B][P][S][3] / prot [B][P][S][4] in: the input arrays,
// out: the rows forward kinematics leaves; psize [B][P][3], action [B][6 P].  position_control != 0: one box primitive (P = 1); else
// collide_batch of P box (sdf_kind 0) or container (1) primitives with their own friction / softness (prim_friction / prim_softness [P]).
// Returns 0.
int oc_mpm_det_forward(int B, int N, int n_grid, const int* res, int steps, float dt, float p_mass, float p_vol, const float* gravity,
                       const int* material, const float* hard, const float* x, const float* v, const float* Cm, const float* F,
                       float* ppos, float* prot, const float* psize, const float* friction, const float* mu, const float* lamda,
                       const float* action, float* xo, float* vo, float* Co, float* Fo, int position_control, int n_prim, int sdf_kind,
                       const float* prim_friction, const float* prim_softness) {
  ud::MpmConst c;
  std::memset(&c, 0, sizeof(c));
  c.N = N; c.Np = (N + 15) / 16 * 16; c.n_grid = n_grid; c.steps = steps;
  for (int d = 0; d < 3; ++d) c.res[d] = res[d];
  const double dx = 1.0 / n_grid;                       // as ud_mpm_create derives them (mpm.hip)
  c.dt = dt; c.dx = (float)dx; c.inv_dx = (float)(double)n_grid;
  c.p_mass = p_mass; c.p_vol = p_vol;
  c.stress_c = (float)(-(double)dt * (double)p_vol * 4.0);
  c.dx2 = (float)(dx * dx);
  for (int d = 0; d < 3; ++d) c.dtg[d] = dt * gravity[d];
  c.position_control = position_control ? 1 : 0; c.n_prim = position_control ? 1 : n_prim; c.sdf_kind = sdf_kind; c.det = 1;
  for (int ip = 0; ip < c.n_prim && ip < 4; ++ip) { c.prim_friction_each[ip] = prim_friction ? prim_friction[ip] : 0.f; c.prim_softness_each[ip] = prim_softness ? prim_softness[ip] : 0.f; }
  const int P = c.n_prim;
  const long G = (long)res[0] * res[1] * res[2];
  const int Np = c.Np, S = steps;
  std::vector<float> h0((size_t)24 * Np), h1((size_t)24 * Np), pre((size_t)UD_DET_PRE * Np), vel((size_t)G * 4), contrib((size_t)27 * Np * 4);
  std::vector<int> flag((size_t)G, 0), bflag((size_t)G, 0), order((size_t)N), bkey((size_t)N);
  std::vector<ud::DetRange> brange((size_t)G);
  std::vector<std::pair<unsigned, int>> keyed((size_t)N);
  for (int b = 0; b < B; ++b) {
    float* pp = ppos + (long)b * P * S * 3;
    float* pr = prot + (long)b * P * S * 4;
    for (int ip = 0; ip < P; ++ip) ud::det_fk_rows(S, action + ((long)b * P + ip) * 6, pp + (long)ip * S * 3, pr + (long)ip * S * 4);
    std::fill(h0.begin(), h0.end(), 0.f); std::fill(h1.begin(), h1.end(), 0.f);
    for (int p = 0; p < N; ++p) {                      // lg_pack: SoA record, nan_to_num on the way in
      for (int d = 0; d < 3; ++d) { h0[d * Np + p] = ud::nan_to_num(x[((long)b * N + p) * 3 + d]); h0[(3 + d) * Np + p] = ud::nan_to_num(v[((long)b * N + p) * 3 + d]); }
      for (int d = 0; d < 9; ++d) { h0[(6 + d) * Np + p] = ud::nan_to_num(Cm[((long)b * N + p) * 9 + d]); h0[(15 + d) * Np + p] = ud::nan_to_num(F[((long)b * N + p) * 9 + d]); }
    }
    std::fill(flag.begin(), flag.end(), 0); std::fill(bflag.begin(), bflag.end(), 0);
    float* h = h0.data();
    float* hn = h1.data();
    for (int f = 0; f < S; ++f) {
      const int epoch = f + 1;
      for (int p = 0; p < N; ++p) {
        ud::Pre q;
        float vp[3];
        bkey[p] = ud::det_pre_particle(c, h, hn, p, mu[b], lamda[b], material[p], hard[p], pre.data(), true, q, vp);
        for (int cidx = 0; cidx < 27; ++cidx) {
          ud::det_contrib(c, q, vp, cidx / 9, (cidx / 3) % 3, cidx % 3, contrib.data() + ((size_t)cidx * Np + p) * 4);
          long sl, gl;
          ud::det_touch(c, pre.data(), p, cidx, sl, gl);
          if (sl >= 0) flag[sl] = epoch;
          flag[gl] = epoch;
        }
        keyed[p] = {(unsigned)(bkey[p] + 1), p};
      }
      std::sort(keyed.begin(), keyed.end());             // by (bucket key, index): irregular particles (key -1) first
      int n_irr = 0;
      for (int i = 0; i < N; ++i) {
        order[i] = keyed[i].second;
        const int key = (int)keyed[i].first - 1;
        if (key < 0) { n_irr = i + 1; continue; }
        if (i == 0 || (int)keyed[i - 1].first - 1 != key) { brange[key].s = i; bflag[key] = epoch; }
        brange[key].e = i + 1;
      }
      const ud::DetBuckets bk{order.data(), brange.data(), bflag.data(), n_irr, epoch};
      const ud::DetPrimRows pw{f, pp, pr, psize + (long)b * P * 3, action + (long)b * P * 6, friction[b]};
      for (long lin = 0; lin < G; ++lin) {
        if (flag[lin] != epoch) continue;
        const int ck = (int)(lin % res[2]), cj = (int)((lin / res[2]) % res[1]), ci = (int)(lin / ((long)res[2] * res[1]));
        float o[3];
        ud::det_cell(c, ci, cj, ck, pre.data(), contrib.data(), pw, bk, o);
        vel[lin * 4] = o[0]; vel[lin * 4 + 1] = o[1]; vel[lin * 4 + 2] = o[2]; vel[lin * 4 + 3] = 0.f;
      }
      for (int p = 0; p < N; ++p) (void)ud::det_g2p_particle(c, h, hn, p, pre.data(), vel.data());
      std::swap(h, hn);
    }
    for (int p = 0; p < N; ++p) {
      for (int d = 0; d < 3; ++d) { xo[((long)b * N + p) * 3 + d] = h[d * Np + p]; vo[((long)b * N + p) * 3 + d] = h[(3 + d) * Np + p]; }
      for (int d = 0; d < 9; ++d) { Co[((long)b * N + p) * 9 + d] = h[(6 + d) * Np + p]; Fo[((long)b * N + p) * 9 + d] = h[(15 + d) * Np + p]; }
    }
  }
  return 0;
}

}  // extern "C"
