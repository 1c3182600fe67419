"""`from networks import FXencoder, TCNModel` - same star-export surface as the reference's networks package."""
from .architectures import *  # noqa: F401,F403
from .architectures import FXencoder, TCNModel, TCNBlock
from .network_utils import *  # noqa: F401,F403
from .network_utils import Conv1d_layer, Res_ConvBlock, ConvBlock, FiLM
