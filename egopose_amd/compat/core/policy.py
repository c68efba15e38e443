from egopose_amd.nets import Policy  # noqa: F401
