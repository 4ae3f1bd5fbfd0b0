"""PrimitiveState / create_primitive -- mirrors the state container of
/root/reference/DaXBench/daxbench/core/engine/primitives/primitives.py:9-60 (torch tensors instead of jnp arrays).
The primitive dynamics themselves (forward_kinematics :185-194, set_action :212-229, position_control_batch
:232-239, collide_batch :154-182, box SDF box.py:6-18, container SDF container.py:8-16) run inside the MPM kernels
(csrc/mpm.hip, mpm_large.hip, mpm_collide.h).  set_sdf records the SDF kind, which each simulator copies into its kernel
handle at reset -- a per-handle constant instead of the reference's process-global `_sdf_batch` function pointer
(:5-6, :26-28), so two envs with different SDFs can live in one process."""
from __future__ import annotations

from typing import NamedTuple

import torch


class PrimitiveState(NamedTuple):
    size: torch.Tensor
    dim: torch.Tensor
    friction: torch.Tensor
    softness: torch.Tensor
    color: torch.Tensor
    position: torch.Tensor
    rotation: torch.Tensor
    v: torch.Tensor
    w: torch.Tensor
    xyz_limit: torch.Tensor
    action_buffer: torch.Tensor
    action_scale: torch.Tensor
    min_dist: torch.Tensor
    dist_norm: torch.Tensor


_SDF_KIND = "box"


def set_sdf(kind):
    """whip_rope_env.py:122 / shape_rope_env.py:158 call set_sdf(box_sdf), pour_water_env.py:119 set_sdf(container_sdf)."""
    global _SDF_KIND
    name = kind if isinstance(kind, str) else getattr(kind, "__name__", "box")
    if "container" in name:
        _SDF_KIND = "container"
    elif "box" in name:
        _SDF_KIND = "box"
    else:
        raise NotImplementedError(f"unknown SDF {name!r}: box (box.py) and container (container.py) are implemented")


def get_sdf_kind():
    return _SDF_KIND


def create_primitive(conf, friction, softness, color, size, init_pos, device="cpu"):   # :31-60
    max_steps = conf.steps
    f32 = lambda a: torch.as_tensor(a, dtype=torch.float32, device=device)
    position = torch.zeros((max_steps, 3), device=device)
    position[0] = f32(init_pos)
    rotation = torch.tensor([[1., 0., 0., 0.]], device=device).repeat(max_steps, 1)
    return PrimitiveState(
        size=f32(size), dim=torch.tensor([3], device=device), friction=f32(friction),
        softness=torch.as_tensor(softness, device=device), color=f32(color), position=position, rotation=rotation,
        v=torch.zeros((max_steps, 3), device=device), w=torch.zeros((max_steps, 3), device=device),
        xyz_limit=torch.tensor([[0., 1.], [0., 1.], [0., 1.]], device=device),
        action_buffer=torch.zeros((6,), device=device), action_scale=torch.ones((6,), device=device),
        min_dist=torch.tensor(0, device=device), dist_norm=torch.tensor(0, device=device))
