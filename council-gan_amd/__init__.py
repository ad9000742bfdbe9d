"""council_gan_amd -- MI355X-native Council-GAN training step.

Public surface = the reference's: `Council_Trainer` (trainer_council.py) and the network classes
(networks.py), plus the host helpers its constructor uses.  Importing the package needs neither a
GPU nor the built library; running anything on the hot path needs both and fails loudly otherwise."""
from . import hip, ops  # noqa: F401
from .networks import (AdaINGen, AdaptiveInstanceNorm2d, Conv2dBlock, ContentEncoder, Decoder_V2_atten,  # noqa: F401
                       LayerNorm, LinearBlock, MLP, MsImageDis, MsImageDisCouncil, ResBlock, ResBlocks,
                       StyleEncoder)
from .input import DeviceInput  # noqa: F401
from .optim import FlatAdam  # noqa: F401
from .parallel import CouncilShard, init_distributed  # noqa: F401
from .trainer import Council_Trainer  # noqa: F401
from .utils import (get_config, get_model_list, get_scheduler, seed_everything, synthetic_batch,  # noqa: F401
                    weights_init)

__version__ = "0.1.0"
