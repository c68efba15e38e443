from egopose_amd.nets import RNN  # noqa: F401
