from egopose_amd.statereg import Dataset  # noqa: F401
