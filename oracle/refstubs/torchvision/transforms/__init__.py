from . import transforms  # noqa: F401
from .transforms import Compose  # noqa: F401
