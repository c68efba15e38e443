from egopose_amd.nets import VideoForecastNet  # noqa: F401
