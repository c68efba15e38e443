from egopose_amd.nets import VideoStateNet  # noqa: F401
