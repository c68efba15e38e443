from egopose_amd.reward import reward_func, quat_space_reward_v3, constant_reward  # noqa: F401
