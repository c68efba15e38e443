from egopose_amd.config import ForecastConfig as Config  # noqa: F401
