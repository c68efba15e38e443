from egopose_amd.nets import MLP  # noqa: F401
