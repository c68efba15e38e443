from egopose_amd.metrics import (get_joint_angles, get_joint_vels, get_joint_accels, get_mean_dist,  # noqa: F401
                                 get_mean_abs)
