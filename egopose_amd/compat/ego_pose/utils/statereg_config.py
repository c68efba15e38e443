from egopose_amd.statereg import StateRegConfig as Config  # noqa: F401
