// Host interface of the deterministic MPM forward (mpm_det.hip), called by mpm_large.hip when ud_mpm_conf.deterministic is set.
#pragma once
#include "mpm_device.h"

namespace ud {

struct DetArgs {
  MpmConst c;
  int B;                  // envs of this call
  long G;                 // cells per env
  const int* material;    // [N]
  const float* hard;      // [N]
  float* ppos;            // [B][S*3] primitive position rows: input array in, forward kinematics in place
  float* prot;            // [B][S*4]
  const float *psize, *friction, *mu, *lamda, *action;
  float* hist;            // SoA history, env stride stride_b, record stride rec; pingpong: two records, else S + 1
  long rec, stride_b;
  int pingpong;
  float* vel;             // [B][G][4] grid velocity after the grid op (written at the stamped cells only)
  int* flag;              // [B][G]    epoch stamp of the cells a substep touches
  float* pre;             // [B][UD_DET_PRE][Np]
  float* contrib;         // [B][27][Np][4] what every (offset, particle) adds to its cell: (m, mv)
  int* bkey;              // [B][Np]   bucket key of every particle (linear index of its base cell; -1 irregular)
  int* order;             // [B][Np]   particle indices sorted by (bucket key, index)
  void* brange;           // [B][G]    DetRange: the bucket of a base cell in `order`
  int* bflag;             // [B][G]    epoch stamp of the valid brange entries
  int* nirr;              // [B]       irregular particles (the first nirr entries of `order`)
  int* list;              // [B][cap]  touched cells of the substep (any order: every cell is summed on its own)
  int* count;             // [B]
  int cap;
  float* trq3;            // [B][S][3] Q6 row sums of particles 0..2
  float* trq;             // [B][S]    their sum, in a fixed order
  // the deterministic BACKWARD (mpm_large.hip drives it; null / 0 in the forward): the recomputed substep also leaves, for the grid-op adjoint,
  int bwd;                // != 0: the history is the caller's checkpoint -- read only
  float* val_out;         // [B][G][4] (m, mv) of the touched cells
  int* keylist;           // [B][cap]  their keys (ci | cj << 10 | ck << 20), in the list's order
  int* keycount;          // [B]
  float* gacc;            // [B][G][4] cotangent of the grid velocity: the ordered sums of the g2p adjoint's contributions (det_gcells)
  float* cellred;         // [B][capc][K] per listed cell, in the list's (sorted) order: what the grid-op adjoint adds to the env's cotangents --
                          // position control: K = 4 (ground friction, controlled velocity xyz); soft contact: K = 1 + 18 n_prim (ground
                          // friction; per primitive p0[3] r0[4] p1[3] r1[4] size[3] mu, as lg_grid_adj_tile books them)
  int capc, K;
  float *gppos, *grot, *gpsz;   // soft contact: the primitives' cotangent rows [B][P][S][3], [B][P][S][4], [B][P][4]
  int* status;            // [B] or null: 1 is OR-ed in when an env touches more cells than the sorted list holds (its gradients are then incomplete)
  float* pacc;            // [B][2][Np] per-particle mu / lamda cotangents, summed over the substeps by the particle's own thread
  float* acc;             // [B][4]    the env's scalars: friction, mu, lamda (, norm)
  float* gpv;             // [B][S][3] cotangent of the controlled velocity rows
};

int mpm_det_forward(const DetArgs& a, int* epoch, hipStream_t st);
// pieces of the deterministic backward of substep f, in launch order around mpm_large.hip's own kernels:
int mpm_det_bwd_recompute(const DetArgs& a, int f, int* epoch, hipStream_t st);   // pre-pass, buckets, the cell list SORTED, ordered (m, mv) sums + grid op -> vel, val_out, keylist
int mpm_det_bwd_gcells(const DetArgs& a, int f, int epoch, hipStream_t st);       // ordered sums of a.contrib (the g2p adjoint's contributions) -> gacc
int mpm_det_bwd_reduce_cells(const DetArgs& a, int f, hipStream_t st);            // cellred -> the env's cotangents of substep f, in a fixed order
int mpm_det_bwd_clear(const DetArgs& a, hipStream_t st);                          // gacc of the touched cells back to zero, the list retired
int mpm_det_bwd_reduce_particles(const DetArgs& a, hipStream_t st);               // pacc -> acc[.][1], acc[.][2] in a fixed order; pacc zeroed again

}  // namespace ud
