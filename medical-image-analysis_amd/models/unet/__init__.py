from .unet import UNet, UNetDecoder, UNetEncoder
from .blocks import PlainBlock, ResidualBlock, convert_sync_batchnorm
from .unet_processor import UnetProcessor
