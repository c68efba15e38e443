import os  # noqa: F401
import numpy as np  # noqa: F401
from egopose_amd.config import recreate_dirs  # noqa: F401
from egopose_amd.logging_utils import get_body_qposaddr  # noqa: F401
