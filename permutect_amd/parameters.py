"""Model hyperparameters.

`ModelParameters` has the same constructor signature and attribute names as the reference class
(reference permutect/parameters.py:16-40) because instances are pickled into the .pt checkpoint under the
key "hyperparams" (reference architecture/artifact_model.py:327-337).  `install_pickle_alias()` makes this
class resolvable as `permutect.parameters.ModelParameters` when the reference package is not installed, so
reference-written checkpoints unpickle here and checkpoints written here unpickle in the reference.
"""
from __future__ import annotations

import sys
import types
from typing import List


class ModelParameters:
    def __init__(
        self,
        read_layers: List[int],
        self_attention_hidden_dimension: int,
        num_self_attention_layers: int,
        info_layers: List[int],
        aggregation_layers: List[int],
        num_artifact_clusters: int,
        calibration_layers: List[int],
        ref_seq_layers_strings: List[str],
        dropout_p: float,
        reweighting_range: float,
        batch_normalize: bool = False,
    ):
        self.read_layers = read_layers
        self.info_layers = info_layers
        self.ref_seq_layer_strings = ref_seq_layers_strings
        self.self_attention_hidden_dimension = self_attention_hidden_dimension
        self.num_self_attention_layers = num_self_attention_layers
        self.aggregation_layers = aggregation_layers
        self.num_artifact_clusters = num_artifact_clusters
        self.calibration_layers = calibration_layers
        self.dropout_p = dropout_p
        self.reweighting_range = reweighting_range
        self.batch_normalize = batch_normalize


# Pickle writes the defining module path.  Advertise the reference path so that checkpoints are portable both ways.
ModelParameters.__module__ = "permutect.parameters"


def install_pickle_alias() -> None:
    """Make `permutect.parameters.ModelParameters` importable (no-op if the reference package is installed)."""
    try:
        import permutect.parameters  # noqa: F401  (the real one, if present)

        return
    except Exception:
        pass
    pkg = sys.modules.get("permutect")
    if pkg is None:
        pkg = types.ModuleType("permutect")
        pkg.__path__ = []  # mark as package
        sys.modules["permutect"] = pkg
    mod = types.ModuleType("permutect.parameters")
    mod.ModelParameters = ModelParameters
    sys.modules["permutect.parameters"] = mod
    pkg.parameters = mod


# ---- named configurations used by tests / bench ------------------------------------------
P0_CNN = [
    "convolution/kernel_size=3/out_channels=32",
    "pool/kernel_size=2",
    "leaky_relu",
    "convolution/kernel_size=3/out_channels=32",
    "leaky_relu",
    "flatten",
    "linear/out_features=10",
]
# the 4-convolution stack of the shipped v0.4.0 checkpoint (10 -> 32 (k3) -> 32 (k3) -> 32 (k5) -> 32 (k5) -> linear 10)
P0_CNN_LEGACY = [
    "convolution/kernel_size=3/out_channels=32", "leaky_relu",
    "convolution/kernel_size=3/out_channels=32", "leaky_relu",
    "convolution/kernel_size=5/out_channels=32", "leaky_relu",
    "convolution/kernel_size=5/out_channels=32", "leaky_relu",
    "flatten",
    "linear/out_features=10",
]
# the stack of the reference's docstring (dna_sequence_convolution.py:36-42: dilation, selu) followed by a strided and a padded
# convolution.  The reference tracks lengths WITHOUT the padding (its conv_output_length takes `pad`, the layer string says
# `padding`): a padded convolution only builds a runnable reference model where both lengths agree, as in the last one here
# (2 -> 1 either way).
T0_CNN_OPTIONS = [
    "convolution/kernel_size=3/out_channels=64",
    "pool/kernel_size=2",
    "leaky_relu",
    "convolution/kernel_size=3/dilation=2/out_channels=5",
    "selu",
    "convolution/kernel_size=3/stride=2/out_channels=8", "leaky_relu",
    "convolution/kernel_size=2/stride=3/padding=1/out_channels=6", "leaky_relu",
    "flatten",
    "linear/out_features=10",
]
# the production stack with the reference's `batch_norm` token in its three foldable places: behind a convolution (folds into that
# convolution's output rows), in front of an un-padded convolution (into its input channels) and between flatten and the linear
# (per flattened feature, into the linear's columns): eval mode only (tests/golden/p0_cnn_batchnorm_eval.npz)
P0_CNN_BATCHNORM = [
    "convolution/kernel_size=3/out_channels=32",
    "batch_norm",
    "pool/kernel_size=2",
    "leaky_relu",
    "batch_norm",
    "convolution/kernel_size=3/out_channels=32",
    "leaky_relu",
    "flatten",
    "batch_norm",
    "linear/out_features=10",
]
T0_CNN = [
    "convolution/kernel_size=3/out_channels=64",
    "pool/kernel_size=2",
    "leaky_relu",
    "flatten",
    "linear/out_features=10",
]


class TrainingParameters:
    """reference parameters.py:160-177"""

    def __init__(self, batch_size: int, num_epochs: int, learning_rate: float = 0.001, weight_decay: float = 0.01,
                 num_workers: int = 0, num_calibration_epochs: int = 0, inference_batch_size: int = 8192,
                 fit_downsampler: bool = True):
        """`fit_downsampler` (not in the reference, which always fits): run Downsampler.optimize_downsampling_balance before
        training (reference training/model_training.py:60); False keeps the uniform mixture weights."""
        self.fit_downsampler = fit_downsampler
        self.batch_size, self.num_epochs = batch_size, num_epochs
        self.learning_rate, self.weight_decay = learning_rate, weight_decay
        self.num_workers, self.num_calibration_epochs = num_workers, num_calibration_epochs
        self.inference_batch_size = inference_batch_size


def p0_params() -> ModelParameters:
    """Production-shaped hyperparameters: 59 845 parameters with F=61, I=71, H=42."""
    return ModelParameters([30, -2, -2, -2], 20, 6, [20, -2, -2, -2], [-2, -2, 10], 4, [10, 10], list(P0_CNN), 0.0, 0.3)


def t0_params() -> ModelParameters:
    """The reference's own test configuration (reference test/tools/test_train_permutect_model.py:19-35)."""
    return ModelParameters([10, 10, 10], 20, 2, [10, 10], [20, 20, 20], 4, [10, 10, 10], list(T0_CNN), 0.0, 0.3)


def wide_params() -> ModelParameters:
    """A configuration with layers wider than 64 (read width 48, info width 40: d_model 98; a 98-wide reducer; d_ffn 32): what a
    `--read_layers` / `--info_layers` edit away from the defaults gives.  Runs the wide build of the library (engine/instances.py)."""
    return ModelParameters([48, -2], 32, 2, [40, -1], [-1, 20], 4, [10, 10], list(P0_CNN), 0.0, 0.3)


def wide64_params() -> ModelParameters:
    """wide_params with d_ffn = 64: the gated blocks' hidden halves take two 16-feature tiles (the wide32 build of the library)"""
    p = wide_params()
    p.self_attention_hidden_dimension = 64
    return p


P0_DIMS = dict(num_read_features=61, num_info_features=71, haplotypes_length=42)


# ---- the reference's command-line flags (parameters.py:43-242): same names, same defaults --------------------------------------
def parse_model_params(args) -> ModelParameters:
    from permutect_amd import constants as c
    return ModelParameters(getattr(args, c.READ_LAYERS_NAME), getattr(args, c.SELF_ATTENTION_HIDDEN_DIMENSION_NAME),
                           getattr(args, c.NUM_SELF_ATTENTION_LAYERS_NAME), getattr(args, c.INFO_LAYERS_NAME),
                           getattr(args, c.AGGREGATION_LAYERS_NAME), getattr(args, c.NUM_ARTIFACT_CLUSTERS_NAME),
                           getattr(args, c.CALIBRATION_LAYERS_NAME), getattr(args, c.REF_SEQ_LAYER_STRINGS_NAME),
                           getattr(args, c.DROPOUT_P_NAME), getattr(args, c.REWEIGHTING_RANGE_NAME), getattr(args, c.BATCH_NORMALIZE_NAME))


def add_model_params_to_parser(parser) -> None:
    from permutect_amd import constants as c
    parser.add_argument("--" + c.PRETRAINED_ARTIFACT_MODEL_NAME, type=str, required=False, help="optional pretrained artifact model to start from")
    parser.add_argument("--" + c.READ_LAYERS_NAME, nargs="+", type=int, required=True, help="read embedding layers; -d = a d-layer residual skip block")
    parser.add_argument("--" + c.SELF_ATTENTION_HIDDEN_DIMENSION_NAME, type=int, required=True, help="hidden dimension (d_ffn) of the gated blocks")
    parser.add_argument("--" + c.NUM_SELF_ATTENTION_LAYERS_NAME, type=int, required=True, help="number of gated ref / alt blocks")
    parser.add_argument("--" + c.INFO_LAYERS_NAME, nargs="+", type=int, required=True, help="info embedding layers")
    parser.add_argument("--" + c.AGGREGATION_LAYERS_NAME, nargs="+", type=int, required=True, help="reducer layers behind the gated blocks")
    parser.add_argument("--" + c.NUM_ARTIFACT_CLUSTERS_NAME, type=int, default=4, required=False, help="artifact clusters of the generative head")
    parser.add_argument("--" + c.CALIBRATION_LAYERS_NAME, nargs="+", type=int, required=True, help="calibration layers (kept in the checkpoint's hyperparameters)")
    parser.add_argument("--" + c.REF_SEQ_LAYER_STRINGS_NAME, nargs="+", type=str, required=True, help="haplotype CNN layer strings, e.g. convolution/kernel_size=3/out_channels=64")
    parser.add_argument("--" + c.DROPOUT_P_NAME, type=float, default=0.0, required=False, help="dropout probability")
    parser.add_argument("--" + c.REWEIGHTING_RANGE_NAME, type=float, default=0.3, required=False, help="reweighting range")
    parser.add_argument("--" + c.BATCH_NORMALIZE_NAME, action="store_true", help="BatchNorm1d in front of every Linear (runs in eval mode only here)")


def parse_training_params(args) -> TrainingParameters:
    from permutect_amd import constants as c
    return TrainingParameters(getattr(args, c.BATCH_SIZE_NAME), getattr(args, c.NUM_EPOCHS_NAME), getattr(args, c.LEARNING_RATE_NAME),
                              getattr(args, c.WEIGHT_DECAY_NAME), getattr(args, c.NUM_WORKERS_NAME), getattr(args, c.NUM_CALIBRATION_EPOCHS_NAME),
                              getattr(args, c.INFERENCE_BATCH_SIZE_NAME))


def add_training_params_to_parser(parser) -> None:
    from permutect_amd import constants as c
    parser.add_argument("--" + c.LEARNING_RATE_NAME, type=float, default=0.001, required=False, help="learning rate")
    parser.add_argument("--" + c.WEIGHT_DECAY_NAME, type=float, default=0.0, required=False, help="weight decay")
    parser.add_argument("--" + c.BATCH_SIZE_NAME, type=int, default=64, required=False, help="batch size")
    parser.add_argument("--" + c.NUM_WORKERS_NAME, type=int, default=0, required=False, help="accepted for compatibility (the device chunk loader has its own prefetch threads)")
    parser.add_argument("--" + c.NUM_EPOCHS_NAME, type=int, required=True, help="training epochs")
    parser.add_argument("--" + c.NUM_CALIBRATION_EPOCHS_NAME, type=int, default=0, required=False, help="calibration-only epochs behind them")
    parser.add_argument("--" + c.INFERENCE_BATCH_SIZE_NAME, type=int, default=8192, required=False, help="batch size of the evaluation passes")
