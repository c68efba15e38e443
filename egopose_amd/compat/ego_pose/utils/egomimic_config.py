from egopose_amd.config import Config  # noqa: F401
