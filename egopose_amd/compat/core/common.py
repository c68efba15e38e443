from egopose_amd.advantages import estimate_advantages  # noqa: F401
