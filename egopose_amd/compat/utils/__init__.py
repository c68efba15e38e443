"""`from utils import *` of the reference (utils/__init__.py:1-7): names the ego_mimic driver relies on."""
from utils.memory import *
from utils.zfilter import *
from utils.torch import *
from utils.tools import *
from utils.logger import *
from utils.tb_logger import *
