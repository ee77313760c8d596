from .panoptic_seg import *  # noqa: F401,F403
from .utils_seg import *  # noqa: F401,F403
