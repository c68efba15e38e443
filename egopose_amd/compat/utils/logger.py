from egopose_amd.logging_utils import create_logger  # noqa: F401
