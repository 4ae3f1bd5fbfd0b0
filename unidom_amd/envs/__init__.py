from .registration import env_functions  # noqa: F401
