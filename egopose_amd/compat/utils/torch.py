import numpy as np  # noqa: F401
import torch  # noqa: F401
from egopose_amd.torch_utils import *  # noqa: F401,F403
from egopose_amd.torch_utils import (tensor, DoubleTensor, FloatTensor, LongTensor, ByteTensor, ones, zeros, to_cpu,  # noqa: F401
                                     to_device, to_test, to_train, batch_to, get_flat_params_from, set_flat_params_to,
                                     get_flat_grad_from, compute_flat_grad, set_optimizer_lr, filter_state_dict)
