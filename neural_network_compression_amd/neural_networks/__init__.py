from .le_net_300_100 import LeNet300100  # noqa: F401
from .le_net_5 import LeNet5  # noqa: F401
