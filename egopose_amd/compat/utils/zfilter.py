from egopose_amd.zfilter import RunningStat, ZFilter  # noqa: F401
