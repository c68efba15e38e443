from egopose_amd.nets import Value  # noqa: F401
