from egopose_amd.nets import PolicyGaussian  # noqa: F401
