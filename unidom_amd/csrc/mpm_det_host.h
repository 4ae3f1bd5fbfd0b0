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
};

int mpm_det_forward(const DetArgs& a, int* epoch, hipStream_t st);

}  // namespace ud
