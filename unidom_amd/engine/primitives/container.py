"""Container SDF marker (cut hollow sphere, size = (r, h, t); the arithmetic of container.py:8-16 lives in
csrc/mpm_collide.h::container_sdf_x)."""


def _sdf_batch(*args, **kwargs):  # noqa: D401
    raise NotImplementedError("the container SDF is evaluated inside the MPM kernels (csrc/mpm_collide.h::container_sdf_x)")


_sdf_batch.__name__ = "container_sdf"
