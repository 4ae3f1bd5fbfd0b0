// ORACLE -- test infrastructure only. C ABI over cloth_oracle.hpp for ctypes (tests/, smoke(), bench cpu_baseline).
// Rollout structure follows robot_step (cloth_simulator.py:163-180) driven by lax.scan over the macro
// actions (cloth_env.py:211).
#include <cstring>
#include <omp.h>

#include "cloth_oracle.hpp"

using namespace oracle;

struct OcCloth {
  int order = 1;   // 1: reference operation order, 2: re-associated IEEE order ("v2")
  ClothTables<float> tf;
  ClothTables<double> td;
  ClothParams<float> pf;
  ClothParams<double> pd;
};

template <class T> static const ClothTables<T>& tabs(const OcCloth* h);
template <> const ClothTables<float>& tabs<float>(const OcCloth* h) { return h->tf; }
template <> const ClothTables<double>& tabs<double>(const OcCloth* h) { return h->td; }
template <class T> static const ClothParams<T>& pars(const OcCloth* h);
template <> const ClothParams<float>& pars<float>(const OcCloth* h) { return h->pf; }
template <> const ClothParams<double>& pars<double>(const OcCloth* h) { return h->pd; }

template <class T>
static void macro_action(const T* a8, T* act /*[2][4]*/) {  // :168-169
  for (int g = 0; g < 2; ++g) {
    for (int c = 0; c < 3; ++c) act[g * 4 + c] = clipf(a8[g * 4 + c], T(-2), T(2)) * (T(1) / T(50));   // x / 50. as XLA executes it under jit: A / Const => A * (1 / Const) (AlgebraicSimplifier); the recorded demos discriminate
    act[g * 4 + 3] = a8[g * 4 + 3];
  }
}

template <class T>
static void rollout_fwd(const OcCloth* h, int B, int TT, const T* x0, const T* v0, const T* prim0, const T* k,
                        const T* mu, const T* actions, T* x_out, T* v_out, T* prim_out, T* x_list, T* v_list,
                        T* prim_list, uint8_t* grasp, T* ckpt, int nthreads) {
  const auto& tb = tabs<T>(h);
  const auto& pr = pars<T>(h);
  const int P = tb.P, S = pr.substeps;
  const size_t rec = (size_t)P * 6 + 8;
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int b = 0; b < B; ++b) {
    std::vector<T> xa(x0 + (size_t)b * P * 3, x0 + (size_t)(b + 1) * P * 3), va(v0 + (size_t)b * P * 3, v0 + (size_t)(b + 1) * P * 3);
    std::vector<T> xb(P * 3), vb(P * 3);
    T pa[8], pb[8], act[8];
    std::memcpy(pa, prim0 + b * 8, sizeof(pa));
    for (int t = 0; t < TT; ++t) {
      macro_action(actions + ((size_t)t * B + b) * 8, act);
      for (int s = 0; s < S; ++s) {
        if (ckpt) {
          T* c = ckpt + (((size_t)b * TT + t) * S + s) * rec;
          std::memcpy(c, xa.data(), sizeof(T) * P * 3);
          std::memcpy(c + P * 3, va.data(), sizeof(T) * P * 3);
          std::memcpy(c + P * 6, pa, sizeof(pa));
        }
        uint8_t* g0 = grasp ? grasp + ((((size_t)t * S + s) * B + b) * 2 + 0) * P : nullptr;
        uint8_t* g1 = grasp ? g0 + P : nullptr;
        if (h->order == 2)
          cloth_substep_fwd_v2(tb, pr, k[b], mu[b], xa.data(), va.data(), pa, act, xb.data(), vb.data(), pb, g0, g1);
        else
          cloth_substep_fwd(tb, pr, k[b], mu[b], xa.data(), va.data(), pa, act, xb.data(), vb.data(), pb, g0, g1);
        xa.swap(xb); va.swap(vb);
        std::memcpy(pa, pb, sizeof(pa));
      }
      if (x_list) std::memcpy(x_list + ((size_t)t * B + b) * P * 3, xa.data(), sizeof(T) * P * 3);
      if (v_list) std::memcpy(v_list + ((size_t)t * B + b) * P * 3, va.data(), sizeof(T) * P * 3);
      if (prim_list) std::memcpy(prim_list + ((size_t)t * B + b) * 8, pa, sizeof(pa));
    }
    std::memcpy(x_out + (size_t)b * P * 3, xa.data(), sizeof(T) * P * 3);
    std::memcpy(v_out + (size_t)b * P * 3, va.data(), sizeof(T) * P * 3);
    std::memcpy(prim_out + b * 8, pa, sizeof(pa));
  }
}

template <class T>
static void rollout_bwd(const OcCloth* h, int B, int TT, const T* x0, const T* v0, const T* prim0, const T* k,
                        const T* mu, const T* actions, const T* gx, const T* gv, const T* gprim, const T* gx_list,
                        const T* gv_list, const T* gprim_list, int normalize, T* gx0, T* gv0, T* gprim0,
                        T* gactions, T* gk, T* gmu, int nthreads) {
  const auto& tb = tabs<T>(h);
  const auto& pr = pars<T>(h);
  const int P = tb.P, S = pr.substeps;
  const size_t rec = (size_t)P * 6 + 8;
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int b = 0; b < B; ++b) {
    // recompute + store every substep input state of this env
    std::vector<T> ck((size_t)TT * S * rec);
    {
      std::vector<T> xo(P * 3), vo(P * 3), xl, vl;
      T po[8];
      // reuse rollout_fwd on a single env (B=1 view)
      std::vector<T> a1((size_t)TT * 8);
      for (int t = 0; t < TT; ++t) std::memcpy(&a1[(size_t)t * 8], actions + ((size_t)t * B + b) * 8, sizeof(T) * 8);
      rollout_fwd<T>(h, 1, TT, x0 + (size_t)b * P * 3, v0 + (size_t)b * P * 3, prim0 + b * 8, k + b, mu + b, a1.data(),
                     xo.data(), vo.data(), po, nullptr, nullptr, nullptr, nullptr, ck.data(), 1);
    }
    std::vector<T> cgx(gx + (size_t)b * P * 3, gx + (size_t)(b + 1) * P * 3), cgv(gv + (size_t)b * P * 3, gv + (size_t)(b + 1) * P * 3);
    T cgp[8];
    std::memcpy(cgp, gprim + b * 8, sizeof(cgp));
    T ak = 0, amu = 0;
    std::vector<T> scratch;
    for (int t = TT - 1; t >= 0; --t) {
      if (gx_list) for (int i = 0; i < P * 3; ++i) cgx[i] += gx_list[((size_t)t * B + b) * P * 3 + i];
      if (gv_list) for (int i = 0; i < P * 3; ++i) cgv[i] += gv_list[((size_t)t * B + b) * P * 3 + i];
      if (gprim_list) for (int i = 0; i < 8; ++i) cgp[i] += gprim_list[((size_t)t * B + b) * 8 + i];
      T act[8], gact[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const T* a8 = actions + ((size_t)t * B + b) * 8;
      macro_action(a8, act);
      for (int s = S - 1; s >= 0; --s) {
        const T* c = ck.data() + ((size_t)t * S + s) * rec;
        cloth_substep_bwd(tb, pr, normalize != 0, k[b], mu[b], c, c + P * 3, c + P * 6, act, cgx.data(), cgv.data(), cgp,
                          gact, &ak, &amu, scratch);
      }
      T* ga = gactions + ((size_t)t * B + b) * 8;
      for (int g = 0; g < 2; ++g) {
        for (int c3 = 0; c3 < 3; ++c3) ga[g * 4 + c3] = gact[g * 4 + c3] * (T(1) / T(50)) * clip_grad(a8[g * 4 + c3], T(-2), T(2));
        ga[g * 4 + 3] = gact[g * 4 + 3];
      }
    }
    std::memcpy(gx0 + (size_t)b * P * 3, cgx.data(), sizeof(T) * P * 3);
    std::memcpy(gv0 + (size_t)b * P * 3, cgv.data(), sizeof(T) * P * 3);
    std::memcpy(gprim0 + b * 8, cgp, sizeof(cgp));
    gk[b] = ak;
    gmu[b] = amu;
  }
}

extern "C" {

void* oc_cloth_create(int N, const uint8_t* mask, double gravity, double dt, double damp_f32, double damp_f64,
                      double max_v, double small_num, int substeps) {
  auto* h = new OcCloth;
  h->tf = make_tables<float>(N, mask);
  h->td = make_tables<double>(N, mask);
  double nm = 0;
  for (int i = 0; i < N * N; ++i) nm += mask[i] ? 1 : 0;
  h->pf = {(float)(gravity * dt), (float)gravity, (float)dt, (float)damp_f32, (float)max_v, (float)small_num, substeps, (float)nm};
  h->pd = {gravity * dt, gravity, dt, damp_f64, max_v, small_num, substeps, nm};
  return h;
}
void oc_cloth_destroy(void* h) { delete (OcCloth*)h; }
void oc_cloth_set_order(void* h, int order) { ((OcCloth*)h)->order = order; }
int oc_cloth_num_particles(void* h) { return ((OcCloth*)h)->tf.P; }
void oc_cloth_tables(void* h, int* nbr, float* L0) {
  auto* c = (OcCloth*)h;
  std::memcpy(nbr, c->tf.nbr.data(), sizeof(int) * c->tf.nbr.size());
  std::memcpy(L0, c->tf.L0.data(), sizeof(float) * c->tf.L0.size());
}

#define DEFINE(SUF, T)                                                                                               \
  void oc_cloth_rollout_fwd_##SUF(void* h, int B, int TT, const T* x0, const T* v0, const T* prim0, const T* k,       \
                                  const T* mu, const T* actions, T* x_out, T* v_out, T* prim_out, T* x_list,         \
                                  T* v_list, T* prim_list, uint8_t* grasp, T* ckpt, int nthreads) {                  \
    rollout_fwd<T>((OcCloth*)h, B, TT, x0, v0, prim0, k, mu, actions, x_out, v_out, prim_out, x_list, v_list,        \
                   prim_list, grasp, ckpt, nthreads);                                                                 \
  }                                                                                                                   \
  void oc_cloth_rollout_bwd_##SUF(void* h, int B, int TT, const T* x0, const T* v0, const T* prim0, const T* k,       \
                                  const T* mu, const T* actions, const T* gx, const T* gv, const T* gprim,           \
                                  const T* gx_list, const T* gv_list, const T* gprim_list, int normalize, T* gx0,    \
                                  T* gv0, T* gprim0, T* gactions, T* gk, T* gmu, int nthreads) {                     \
    rollout_bwd<T>((OcCloth*)h, B, TT, x0, v0, prim0, k, mu, actions, gx, gv, gprim, gx_list, gv_list, gprim_list,   \
                   normalize, gx0, gv0, gprim0, gactions, gk, gmu, nthreads);                                        \
  }
DEFINE(f32, float)
DEFINE(f64, double)
#undef DEFINE

}  // extern "C"
