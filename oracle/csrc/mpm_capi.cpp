// ORACLE -- test infrastructure only. C ABI over mpm_oracle.hpp for ctypes (tests/, smoke(), bench cpu_baseline).
// `step` structure follows mpm_simulator.py:413-429 (norm_grad_state / norm_grad :375-411, set_action
// primitives.py:212-229, fori_loop of substep, copy_frame :365-373).
#include <omp.h>

#include "mpm_oracle.hpp"

using namespace oracle;

struct OcMpm {
  MpmParams<float> pf;
  MpmParams<double> pd;
};
template <class T> static const MpmParams<T>& P(const OcMpm* h);
template <> const MpmParams<float>& P<float>(const OcMpm* h) { return h->pf; }
template <> const MpmParams<double>& P<double>(const OcMpm* h) { return h->pd; }

template <class T>
static void fill_params(MpmParams<T>& p, int N, int n_grid, const int* res, int steps, double dt, double p_mass,
                        double p_vol, const double* gravity, int position_control, const int* material, const double* hard,
                        double prim_friction, double prim_softness, int n_prim, int sdf_kind) {
  p.N = N; p.n_grid = n_grid; p.steps = steps;
  for (int d = 0; d < 3; ++d) p.res[d] = res[d];
  const double dx = 1.0 / n_grid;
  p.dt = (T)dt; p.dx = (T)dx; p.inv_dx = (T)(double)n_grid; p.p_mass = (T)p_mass; p.p_vol = (T)p_vol;
  p.stress_c = (T)(-dt * p_vol * 4);
  p.dx2 = (T)(dx * dx);
  for (int d = 0; d < 3; ++d) p.dtg[d] = (T)dt * (T)gravity[d];
  p.position_control = position_control;
  p.prim_friction = (T)prim_friction; p.prim_softness = (T)prim_softness; p.n_prim = n_prim; p.sdf_kind = sdf_kind;
  p.material.assign(material, material + N);
  p.h.resize(N);
  for (int i = 0; i < N; ++i) p.h[i] = (T)hard[i];
}

template <class T>
struct StepIO {
  const T *x, *v, *C, *F, *J, *ppos, *prot, *psize, *friction, *mu, *lamda, *action;
};

template <class T>
static void load_state(const MpmParams<T>& pr, const StepIO<T>& io, int b, MpmState<T>& st, T* a_clipped /*[6*n_prim]*/) {
  const int N = pr.N, S = pr.steps, P = pr.n_prim;
  st.alloc(N, S, P);
  for (int i = 0; i < N * 3; ++i) { st.x[i] = nan_to_num(io.x[(size_t)b * N * 3 + i]); st.v[i] = nan_to_num(io.v[(size_t)b * N * 3 + i]); }
  for (int i = 0; i < N * 9; ++i) { st.C[i] = nan_to_num(io.C[(size_t)b * N * 9 + i]); st.F[i] = nan_to_num(io.F[(size_t)b * N * 9 + i]); }
  for (int i = 0; i < N; ++i) st.J[i] = nan_to_num(io.J[(size_t)b * N + i]);
  st.friction = io.friction[b]; st.mu = io.mu[b]; st.lamda = io.lamda[b];
  for (int ip = 0; ip < P; ++ip) {                       // primitive arrays are [B][n_prim][...]
    PrimS<T>& pp = st.prims[ip];
    const size_t bp = (size_t)b * P + ip;
    for (int i = 0; i < S * 3; ++i) pp.ppos[i] = io.ppos[bp * S * 3 + i];
    for (int i = 0; i < S * 4; ++i) pp.prot[i] = io.prot[bp * S * 4 + i];
    for (int a = 0; a < 3; ++a) pp.psize[a] = io.psize[bp * 3 + a];
    T* ac = a_clipped + ip * 6;
    for (int c = 0; c < 6; ++c) ac[c] = clipf(io.action[bp * 6 + c], T(-1), T(1));       // :419
    for (int j = 0; j < S; ++j)                                                            // set_velocity
      for (int c = 0; c < 3; ++c) { pp.pv[j * 3 + c] = ac[c] * T(1) / (T)S; pp.pw[j * 3 + c] = ac[3 + c] * T(1) / (T)S; }
  }
}

template <class T>
static void copy_frame(const MpmParams<T>& pr, MpmState<T>& st) {  // copy_frame(steps, 0): source index clamps (Q5)
  const int src = pr.steps - 1;
  for (auto& pp : st.prims) {
    for (int a = 0; a < 3; ++a) pp.ppos[a] = pp.ppos[src * 3 + a];
    for (int a = 0; a < 4; ++a) pp.prot[a] = pp.prot[src * 4 + a];
  }
}

template <class T>
static void step_fwd(const OcMpm* h, int B, StepIO<T> io, T* xo, T* vo, T* Co, T* Fo, T* Jo, T* ppos_o, T* prot_o,
                     T* pv_o, T* pw_o, int nthreads) {
  const auto& pr = P<T>(h);
  const int N = pr.N, S = pr.steps, P = pr.n_prim;
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int b = 0; b < B; ++b) {
    MpmState<T> a, c;
    T ac[6 * 8];
    load_state(pr, io, b, a, ac);
    std::vector<T> gm, gv;
    for (int f = 0; f < S; ++f) { mpm_substep_fwd(pr, f, a, c, gm, gv); std::swap(a, c); }
    copy_frame(pr, a);
    std::memcpy(xo + (size_t)b * N * 3, a.x.data(), sizeof(T) * N * 3);
    std::memcpy(vo + (size_t)b * N * 3, a.v.data(), sizeof(T) * N * 3);
    std::memcpy(Co + (size_t)b * N * 9, a.C.data(), sizeof(T) * N * 9);
    std::memcpy(Fo + (size_t)b * N * 9, a.F.data(), sizeof(T) * N * 9);
    std::memcpy(Jo + (size_t)b * N, a.J.data(), sizeof(T) * N);
    for (int ip = 0; ip < P; ++ip) {
      const size_t bp = (size_t)b * P + ip;
      std::memcpy(ppos_o + bp * S * 3, a.prims[ip].ppos.data(), sizeof(T) * S * 3);
      std::memcpy(prot_o + bp * S * 4, a.prims[ip].prot.data(), sizeof(T) * S * 4);
      if (pv_o) std::memcpy(pv_o + bp * S * 3, a.prims[ip].pv.data(), sizeof(T) * S * 3);
      if (pw_o) std::memcpy(pw_o + bp * S * 3, a.prims[ip].pw.data(), sizeof(T) * S * 3);
    }
  }
}

template <class T>
static void step_bwd(const OcMpm* h, int B, StepIO<T> io, const T* gx, const T* gv, const T* gC, const T* gF,
                     const T* gppos, const T* gprot, int clip, T* gx0, T* gv0, T* gC0, T* gF0, T* gppos0, T* gprot0, T* gfric,
                     T* gmu, T* glam, T* gaction, int nthreads) {
  const auto& pr = P<T>(h);
  const int N = pr.N, S = pr.steps, P = pr.n_prim;
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int b = 0; b < B; ++b) {
    std::vector<MpmState<T>> states(S + 1);
    T ac[6 * 8];
    load_state(pr, io, b, states[0], ac);
    std::vector<T> gm, gvv;
    for (int f = 0; f < S; ++f) mpm_substep_fwd(pr, f, states[f], states[f + 1], gm, gvv);
    MpmGrad<T> g;
    g.alloc(N, S, P);
    for (int i = 0; i < N * 3; ++i) { g.x[i] = gx[(size_t)b * N * 3 + i]; g.v[i] = gv[(size_t)b * N * 3 + i]; }
    for (int i = 0; i < N * 9; ++i) { g.C[i] = gC[(size_t)b * N * 9 + i]; g.F[i] = gF[(size_t)b * N * 9 + i]; }
    for (int ip = 0; ip < P; ++ip) {
      auto& G = g.prims[ip];
      const size_t bp = (size_t)b * P + ip;
      for (int i = 0; i < S * 3; ++i) G.ppos[i] = gppos[bp * S * 3 + i];
      // position control: nothing downstream reads the rotation array, the boundary's contract is a zero incoming cotangent
      // (include/unidom_hip.h) -- ignored here too, so that it cannot leak into the clip norm
      if (gprot && !pr.position_control) for (int i = 0; i < S * 4; ++i) G.prot[i] = gprot[bp * S * 4 + i];
      // copy_frame adjoint: position[0] <- position[steps-1], rotation likewise
      if (S - 1 != 0) for (int a = 0; a < 3; ++a) { G.ppos[(S - 1) * 3 + a] += G.ppos[a]; G.ppos[a] = 0; }
      if (S - 1 != 0) for (int a = 0; a < 4; ++a) { G.prot[(S - 1) * 4 + a] += G.prot[a]; G.prot[a] = 0; }
    }
    for (int f = S - 1; f >= 0; --f) mpm_substep_bwd(pr, f, states[f], g);
    // set_action adjoint (primitives.py:212-229); action_scale = 1 is a state leaf (its cotangent only enters the norm)
    T ga[6 * 8] = {0}, gscale[6 * 8] = {0};
    for (int ip = 0; ip < P; ++ip)
      for (int j = 0; j < S; ++j)
        for (int c = 0; c < 3; ++c) { ga[ip * 6 + c] += g.prims[ip].pv[j * 3 + c] * T(1) / (T)S; gscale[ip * 6 + c] += g.prims[ip].pv[j * 3 + c] * ac[ip * 6 + c] / (T)S; }
    // rotation path.  Position control: no cotangent reaches rotation; d|w|/dw at w = 0 is NaN in the reference and
    // is zeroed by nan_to_num at this boundary -> reported as 0.  Soft contact: the chain rule as written (NaN at w = 0).
    if (!pr.position_control)
      for (int ip = 0; ip < P; ++ip)
        for (int j = 0; j < S; ++j)
          for (int c = 0; c < 3; ++c) { ga[ip * 6 + 3 + c] += g.prims[ip].pw[j * 3 + c] * T(1) / (T)S; gscale[ip * 6 + 3 + c] += g.prims[ip].pw[j * 3 + c] * ac[ip * 6 + 3 + c] / (T)S; }
    for (int c = 0; c < 6 * P; ++c) ga[c] *= clip_grad(io.action[(size_t)b * P * 6 + c], T(-1), T(1));
    if (clip) {  // norm_grad_bwd / norm_grad_state_bwd (:389-394, :403-408)
      T n2 = 0;
      for (int c = 0; c < 6 * P; ++c) { ga[c] = nan_to_num(ga[c] + T(0)); n2 += ga[c] * ga[c]; }
      T nrm = std::sqrt(n2);
      if (!(nrm < T(1))) for (int c = 0; c < 6 * P; ++c) ga[c] = ga[c] / nrm;
      T s2 = 0;
      auto acc = [&](std::vector<T>& a) { for (auto& q : a) { q = nan_to_num(q + T(0)); s2 += q * q; } };
      acc(g.x); acc(g.v); acc(g.C); acc(g.F);
      for (auto& G : g.prims) {
        acc(G.ppos); acc(G.prot);
        for (int c = 0; c < 3; ++c) { G.psize[c] = nan_to_num(G.psize[c]); s2 += G.psize[c] * G.psize[c]; }
        G.pfriction = nan_to_num(G.pfriction); s2 += G.pfriction * G.pfriction;
      }
      g.friction = nan_to_num(g.friction); g.mu = nan_to_num(g.mu); g.lamda = nan_to_num(g.lamda);
      s2 += g.friction * g.friction + g.mu * g.mu + g.lamda * g.lamda;
      for (int c = 0; c < 6 * P; ++c) { gscale[c] = nan_to_num(gscale[c]); s2 += gscale[c] * gscale[c]; }
      T sn = std::sqrt(s2);
      if (!(sn < T(1))) {
        auto sc = [&](std::vector<T>& a) { for (auto& q : a) q = q / sn; };
        sc(g.x); sc(g.v); sc(g.C); sc(g.F);
        for (auto& G : g.prims) { sc(G.ppos); sc(G.prot); }
        g.friction /= sn; g.mu /= sn; g.lamda /= sn;
      }
    }
    std::memcpy(gx0 + (size_t)b * N * 3, g.x.data(), sizeof(T) * N * 3);
    std::memcpy(gv0 + (size_t)b * N * 3, g.v.data(), sizeof(T) * N * 3);
    std::memcpy(gC0 + (size_t)b * N * 9, g.C.data(), sizeof(T) * N * 9);
    std::memcpy(gF0 + (size_t)b * N * 9, g.F.data(), sizeof(T) * N * 9);
    for (int ip = 0; ip < P; ++ip) {
      const size_t bp = (size_t)b * P + ip;
      std::memcpy(gppos0 + bp * S * 3, g.prims[ip].ppos.data(), sizeof(T) * S * 3);
      if (gprot0) std::memcpy(gprot0 + bp * S * 4, g.prims[ip].prot.data(), sizeof(T) * S * 4);
    }
    gfric[b] = g.friction; gmu[b] = g.mu; glam[b] = g.lamda;
    for (int c = 0; c < 6 * P; ++c) gaction[(size_t)b * P * 6 + c] = ga[c];
  }
}

extern "C" {

void* oc_mpm_create(int N, int n_grid, const int* res, int steps, double dt, double p_mass, double p_vol,
                    const double* gravity, int position_control, const int* material, const double* hardness,
                    double prim_friction, double prim_softness, int n_prim, int sdf_kind) {
  if (n_prim < 1 || n_prim > 8 || (position_control && n_prim != 1)) return nullptr;
  auto* h = new OcMpm;
  fill_params(h->pf, N, n_grid, res, steps, dt, p_mass, p_vol, gravity, position_control, material, hardness, prim_friction, prim_softness, n_prim, sdf_kind);
  fill_params(h->pd, N, n_grid, res, steps, dt, p_mass, p_vol, gravity, position_control, material, hardness, prim_friction, prim_softness, n_prim, sdf_kind);
  return h;
}
void oc_mpm_destroy(void* h) { delete (OcMpm*)h; }
// per-primitive friction / softness (PrimitiveState.friction / .softness as create_primitive sets them, mpm_env.py:201-217)
void oc_mpm_set_prim_each(void* hv, int n, const double* friction, const double* softness) {
  OcMpm* h = (OcMpm*)hv;
  h->pf.prim_friction_each.assign(friction, friction + n); h->pd.prim_friction_each.assign(friction, friction + n);
  h->pf.prim_softness_each.assign(softness, softness + n); h->pd.prim_softness_each.assign(softness, softness + n);
}

void oc_svd3_f32(const float* A, float* U, float* S, float* Vh) {
  M3<float> a = load9(A), u, vh;
  svd3(a, u, S, vh);
  store9(U, u); store9(Vh, vh);
}
void oc_svd3_f64(const double* A, double* U, double* S, double* Vh) {
  M3<double> a = load9(A), u, vh;
  svd3(a, u, S, vh);
  store9(U, u); store9(Vh, vh);
}

#define DEFINE(SUF, T)                                                                                              \
  void oc_mpm_step_fwd_##SUF(void* h, int B, const T* x, const T* v, const T* C, const T* F, const T* J,             \
                             const T* ppos, const T* prot, const T* psize, const T* friction, const T* mu,           \
                             const T* lamda, const T* action, T* xo, T* vo, T* Co, T* Fo, T* Jo, T* ppos_o,         \
                             T* prot_o, T* pv_o, T* pw_o, int nthreads) {                                            \
    StepIO<T> io{x, v, C, F, J, ppos, prot, psize, friction, mu, lamda, action};                                     \
    step_fwd<T>((OcMpm*)h, B, io, xo, vo, Co, Fo, Jo, ppos_o, prot_o, pv_o, pw_o, nthreads);                         \
  }                                                                                                                  \
  void oc_mpm_step_bwd_##SUF(void* h, int B, const T* x, const T* v, const T* C, const T* F, const T* J,             \
                             const T* ppos, const T* prot, const T* psize, const T* friction, const T* mu,           \
                             const T* lamda, const T* action, const T* gx, const T* gv, const T* gC, const T* gF,   \
                             const T* gppos, const T* gprot, int clip, T* gx0, T* gv0, T* gC0, T* gF0, T* gppos0,   \
                             T* gprot0, T* gfric, T* gmu, T* glam, T* gaction, int nthreads) {                        \
    StepIO<T> io{x, v, C, F, J, ppos, prot, psize, friction, mu, lamda, action};                                     \
    step_bwd<T>((OcMpm*)h, B, io, gx, gv, gC, gF, gppos, gprot, clip, gx0, gv0, gC0, gF0, gppos0, gprot0, gfric, gmu, \
                glam, gaction, nthreads);                                                                                \
  }
DEFINE(f32, float)
DEFINE(f64, double)
#undef DEFINE

}  // extern "C"
