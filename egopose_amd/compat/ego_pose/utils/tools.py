from egopose_amd.metrics import remove_noisy_hands, align_human_state  # noqa: F401
