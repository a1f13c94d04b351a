from .unet import UNet, UNetDecoder, UNetEncoder
from .blocks import PlainBlock, ResidualBlock
from .unet_processor import UnetProcessor
