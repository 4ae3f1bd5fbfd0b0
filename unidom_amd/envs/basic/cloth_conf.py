"""Constants shared by the cloth task configurations, and the patch-shaped mask most of them use.

Every DaXBench cloth env repeats the same conf block with two or three values changed (fold_cloth1_env.py:15-33,
fold_cloth3_env.py:18-38, unfold_cloth1_env.py:15-35, fold_cloth_tshirt_env.py:19-40); here the common part lives once and the
task confs state only what differs.  The attribute names are the reference's: ClothSimulator / ClothEnv read them by name."""
import os

import numpy as np

ENVS_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class ClothConfBase:
    N = 80                    # lattice points per side; the cloth occupies the points its mask selects
    gravity = 0.5
    damping = 2
    dt = 2e-3
    max_v = 2.
    small_num = 1e-8
    mu = 0.5                  # ground friction
    seed = 1
    mem_saving_level = 2      # accepted for interface parity; the HIP path checkpoints per substep instead
    use_substep_obs = True
    task = None

    @property
    def cell_size(self):
        return 1.0 / self.N

    @property
    def size(self):
        return int(self.N / 5.0)

    @property
    def goal_path(self):
        return f"{ENVS_DIR}/goals/{self.task}/goal.npy"


def patch_mask(conf):
    """The 16 x 32 patch of the 80 x 80 lattice that fold_cloth1/3 and unfold_cloth1/3 share (fold_cloth1_env.py:48-53)."""
    n, s = conf.N, conf.size
    mask = np.zeros((n, n), dtype=np.float32)
    mask[2 * s:3 * s, 2 * s:4 * s] = 1
    return mask
