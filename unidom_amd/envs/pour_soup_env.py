"""pour_soup -- mirrors /root/reference/DaXBench/daxbench/core/envs/pour_soup_env.py:29-183
(DefaultConf :29-58, get_primitive_actions :73-86, auto_reset :92-101, reset :110-183).  pour_water's two container bowls
at n_grid 128 with a mixed cloud: 2877 liquid particles (material 0), two 343-particle tofu blocks and a 4068-point vegetable
(material 1, hardness 0.3) -- 7631 particles, which is what the reference's hard-coded observation_size 45861 = 7631*6 + 25*3
says its own reset produces.

The vegetable is the reference's point-cloud asset (core/engine/pyrender/models/veg/model.pcd, kept here as the xyz
columns in others/veg_points.npy; tests/golden/make_golden.py extracts it) passed through open3d's
voxel_down_sample(0.5); open3d is absent, `voxel_down_sample` below restates it.  The count it yields (4068) is pinned by
that observation_size; open3d returns the voxels in the iteration order of a std::unordered_map, here they come in
lexicographic voxel order -- the particle order inside the vegetable block differs, the set of particles does not.

conf.goal_path names goals/pour_soup2/goal.npy, which the reference does not ship (its goals/pour_soup/goal.npy holds 1102
points of an older particle set): MPMEnv then warns and scores against zeros((1,3)) (mpm_env.py:46-48), and so does this env.
"""
import os
from dataclasses import dataclass

import numpy as np
import torch

from ..engine.primitives.container import _sdf_batch as container_sdf
from ..engine.primitives.primitives import set_sdf
from ..utils import prng
from .basic.mpm_env import MPMEnv
from .pour_water_env import PourWaterEnv

my_path = os.path.dirname(os.path.abspath(__file__))


def voxel_down_sample(points, voxel_size):
    """open3d.geometry.PointCloud.voxel_down_sample: voxel (i,j,k) = floor((p - (min_bound - voxel_size/2)) / voxel_size), one
    output point per occupied voxel = the mean of its points (float64)."""
    pts = np.asarray(points, np.float64)
    origin = pts.min(0) - voxel_size * 0.5
    idx = np.floor((pts - origin) / voxel_size).astype(np.int64)
    _, inv, cnt = np.unique(idx, axis=0, return_inverse=True, return_counts=True)
    out = np.zeros((cnt.shape[0], 3), np.float64)
    np.add.at(out, inv.reshape(-1), pts)
    return out / cnt[:, None]


@dataclass
class DefaultConf:
    seed = 1
    n_primitive = 2
    obs_type = MPMEnv.PARTICLE
    key = prng.PRNGKey(0)

    ground_friction: float = 0.1
    n_grid: int = 128
    dt: float = 4e-4
    primitive_action_steps = 1
    primitive_action_duration = 0.01  # seconds
    steps = int(primitive_action_duration / primitive_action_steps / dt)  # internal steps (= 25)
    E: float = 100
    nu: float = 0.1

    res: tuple = (n_grid, n_grid // 2, n_grid)

    dx, inv_dx = 1 / n_grid, float(n_grid)
    p_vol, p_rho = (dx * 0.5) ** 2, 1
    p_mass = p_vol * p_rho
    gravity = (0, -9.8, 0)

    # kernel options, not reference fields (include/unidom_hip.h; see pour_water_env.py)
    sort_particles = 1
    grid_ckpt_cells = 2

    task = "pour_soup2"
    goal_path = f"{my_path}/goals/{task}/goal.npy"
    veg_path = f"{my_path}/others/veg_points.npy"


PourSoupConfig = DefaultConf


class PourSoupEnv(PourWaterEnv):
    # get_primitive_actions (:73-86), process_pre_step_actions (:88-90) and auto_reset (:92-101) are pour_water's, line for line

    def __init__(self, batch_size, seed, max_steps=120, conf=None, aux_reward=False, device="cuda", **kwargs):
        conf = DefaultConf() if conf is None else conf
        super().__init__(batch_size, seed, max_steps=max_steps, conf=conf, aux_reward=aux_reward, device=device)
        self.observation_size = 45861   # 7631 * 6 + 25 * 3

    def reset(self, key):   # :110-183
        self.clean_up_b4_reset()
        set_sdf(container_sdf)
        conf, size_list = self.conf, []
        state = self.simulator.add_box(conf=conf, state=None, hardness=1, size=[0.07, 0.07, 0.07], init_pos=[0.5, 0.2, 0.5],
                                       z_rotation_angle=0, material=0, density=4)                        # the soup
        size_list.append(state.x.shape[0])
        for pos in ([0.47, 0.2, 0.5], [0.5, 0.2, 0.55]):                                                 # two tofu blocks
            state = self.simulator.add_box(conf=conf, state=state, hardness=0.3, size=[0.03, 0.03, 0.03], init_pos=pos,
                                           z_rotation_angle=0, material=1, density=2)
            size_list.append(state.x.shape[0] - sum(size_list))
        veg_pc = voxel_down_sample(np.load(conf.veg_path), voxel_size=0.5)                               # :152-154
        veg_pc = (veg_pc - veg_pc.mean(0)) / 400.0
        veg_pc = (veg_pc + np.array([0.55, 0.2, 0.5], np.float32)).astype(np.float32)                    # jnp (f32) from here on
        state = self.simulator.add_box_from_points(conf, state, veg_pc, hardness=0.3, material=1)        # :160-168
        size_list.append(state.x.shape[0] - sum(size_list))
        box_size = np.array([[0.09, 0., 0.008], [0.08, 0., 0.008]])
        self.create_primitive(conf, state=state, friction=0.1, softness=666, color=[0.5, 0.5, 0.5],
                              size=box_size[0], init_pos=[0.5, 0.2, 0.5])
        self.create_primitive(conf, state=state, friction=0.1, softness=666, color=[0.5, 0.5, 0.5],
                              size=box_size[1], init_pos=[0.5, 0.06, 0.3])
        self.initialize_after_adding_particle_primitives(state)
        self.state = self.auto_reset(self.init_state, self.init_state, self.init_state.key)
        self.size_list = size_list
        return self.get_obs(self.state), self.state
