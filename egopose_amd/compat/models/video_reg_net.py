from egopose_amd.nets import VideoRegNet  # noqa: F401
