from core.common import estimate_advantages  # noqa: F401
from core.logger_rl import LoggerRL  # noqa: F401
from core.trajbatch import TrajBatch  # noqa: F401
