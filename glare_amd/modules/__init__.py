"""Python mirror of the reference's operator surface (code/models/modules/*): same class names,
constructor arguments, forward signatures and state-dict keys, with every forward running on the
HIP kernels of libglare_hip.so.  There is no CPU path: CPU tensors raise."""
from .encoder_decoder import AttnBlock, Decoder, Downsample, Encoder, ResnetBlock, Upsample  # noqa: F401
from .quantize import VectorQuantizer2  # noqa: F401
from .VQModel_arch import VQModel  # noqa: F401
from .ConditionEncoder import ConEncoder1  # noqa: F401
from .FlowUpsamplerNet import FlowUpsamplerNet  # noqa: F401
from .deformableDecoder_arch import DCNv2Pack, Mix, MultiScaleDecoder2, WarpBlock  # noqa: F401
from .VQLLFLOWDeformable_arch import VQLLFLOWDeformable  # noqa: F401
from .LLFlowVQGAN_arch import LLFlowVQGAN2  # noqa: F401
