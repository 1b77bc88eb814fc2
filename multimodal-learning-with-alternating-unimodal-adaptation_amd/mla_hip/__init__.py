"""mla_hip: MI355X-native MLA alternating-unimodal training step (HIP kernels behind a C ABI).

Importing the package does not touch the GPU; the first op loads libmla_hip.so and raises
MLAHipError if it is missing (there is no CPU or eager fallback).
"""
from ._lib import MLAHipError, LIB_PATH  # noqa: F401
from .data import NpyBatcher, load_fbank, load_token  # noqa: F401
from .dist import Comm  # noqa: F401
from .encoder import ResNet18Encoder  # noqa: F401
from .feed import DeviceFeeder  # noqa: F401
from .model import AVClassifier, ConcatFusion, SharedHead  # noqa: F401
from .m3ae import ConcatFusion3, M3AEClassifier, M3AEEncoder, Modal3Classifier  # noqa: F401
from .modulation import OGM  # noqa: F401
from .optim import FusedSGD  # noqa: F401
from .plugin import GSPlugin  # noqa: F401
from .protocol import CrossEntropyLoss, DataParallel, setup_seed, weight_init  # noqa: F401
from .trainer import Evaluator, MLATrainer  # noqa: F401
from . import torch_ops  # noqa: F401,E402  registers torch.ops.mla_hip.* (dispatch key CUDA; no other implementation)
