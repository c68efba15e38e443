from egopose_amd.nets import ResNet  # noqa: F401
