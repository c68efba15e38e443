from egopose_amd.logging_utils import Logger  # noqa: F401
