from egopose_amd.nets import DiagGaussian  # noqa: F401
