"""Parameter-holding module tree of the artifact model.

These classes exist to (a) own the learnable tensors under exactly the state_dict key names the reference
produces, so `.pt` checkpoints are interchangeable, and (b) describe the layer programs
that `permutect_amd.engine.plan` lowers to the packed device layout consumed by the HIP kernels.

The read-set path (read MLP, gated ref/alt MLP, reducer, rotation, clustering head) is *never* evaluated through
these modules: `ArtifactModel` hands their tensors to the C ABI.  Only the small per-variant branches
(info MLP, haplotype CNN, adversaries) have a torch `forward`, which runs on the ROCm device.

Key-name contract follows: reference architecture/mlp.py:8-76, gated_mlp.py:148-276, feature_clustering.py:49-80,
exponentially_modified_gaussian.py:58-80, euclidean_transformation.py:8-23, adversarial.py:6-27,
dna_sequence_convolution.py:31-111, parameterizations.py:21-112.
"""
from __future__ import annotations

from math import floor
from typing import List

import torch
from torch import Tensor, nn
from torch.nn.utils import parametrize
from torch.nn.utils.parametrizations import orthogonal

from permutect_amd import constants


# ----------------------------------------------------------------------------------------------------------------
# parametrizations (they define the `...parametrizations.<name>.original` state_dict keys)
# ----------------------------------------------------------------------------------------------------------------
class UnitVector(nn.Module):
    def forward(self, x: Tensor) -> Tensor:
        return x / torch.linalg.vector_norm(x, dim=-1, keepdim=True)


class PositiveNumber(nn.Module):
    def forward(self, x: Tensor) -> Tensor:
        return torch.exp(x)

    def right_inverse(self, p: Tensor) -> Tensor:
        return torch.log(p)


class BoundedNumber(nn.Module):
    def __init__(self, min_val: float, max_val: float):
        super().__init__()
        assert min_val <= max_val
        self.min_val, self.max_val, self.size = min_val, max_val, max_val - min_val

    def forward(self, x: Tensor) -> Tensor:
        return self.size * torch.sigmoid(x) + self.min_val

    def right_inverse(self, p: Tensor) -> Tensor:
        return torch.logit((p - self.min_val) / self.size)


class LogWeights(nn.Module):
    def forward(self, x: Tensor) -> Tensor:
        return torch.log_softmax(x, dim=-1)


# ----------------------------------------------------------------------------------------------------------------
# MLP with residual skip blocks
# ----------------------------------------------------------------------------------------------------------------
class DenseSkipBlock(nn.Module):
    """x + alpha * f(x); f = (SELU, Linear) * num_layers, all of width `input_size`."""

    def __init__(self, input_size: int, num_layers: int, batch_normalize: bool = False, dropout_p: float = 0):
        super().__init__()
        self.mlp = MLP((num_layers + 1) * [input_size], batch_normalize, dropout_p, prepend_activation=True)
        self.alpha = nn.Parameter(torch.tensor(0.1))
        self.width, self.num_layers = input_size, num_layers

    def forward(self, x: Tensor) -> Tensor:
        return x + self.alpha * self.mlp(x)


class MLP(nn.Module):
    """layer_sizes[0] is the input width; a negative entry -d is a d-layer DenseSkipBlock at the current width."""

    def __init__(self, layer_sizes: List[int], batch_normalize: bool = False, dropout_p: float = 0,
                 prepend_activation: bool = False):
        super().__init__()
        # batch_normalize (reference mlp.py:52-53; off by default, parameters.py:139-156): an nn.BatchNorm1d in front of every Linear,
        # kept as modules so that a reference checkpoint loads key for key.  EVAL mode (running statistics: what filter_variants and
        # evaluation run) is an affine map per feature and is folded into the Linear behind it when the model is lowered
        # (engine/plan.py); TRAINING with it is refused loudly (ArtifactModel._encode): batch statistics span every read of the
        # batch in the middle of the fused read-set pass.
        self.batch_normalize = bool(batch_normalize)
        # dropout_p > 0 (reference mlp.py:57-58; default 0): the nn.Dropout modules are kept in the Sequential so that the
        # state_dict keys of a reference checkpoint line up; the engine lowers them to a flag of the MLP (engine/plan.py) and the
        # kernels mask every Linear's output in train mode (pmt_dropout.hpp) and run the identity in eval mode.
        self.dropout_p = float(dropout_p)
        layers: List[nn.Module] = [nn.SELU()] if prepend_activation else []
        self._input_dim = layer_sizes[0]
        width = layer_sizes[0]
        last = len(layer_sizes) - 2
        for k, out in enumerate(layer_sizes[1:]):
            if out < 0:
                layers.append(DenseSkipBlock(width, -out, batch_normalize, dropout_p))
                continue
            if batch_normalize:
                layers.append(nn.BatchNorm1d(num_features=width))
            layers.append(nn.Linear(width, out))
            if dropout_p > 0:
                layers.append(nn.Dropout(p=dropout_p))
            if k < last:
                layers.append(nn.SELU())
            width = out
        self._output_dim = width
        self._model = nn.Sequential(*layers)

    def input_dimension(self) -> int:
        return self._input_dim

    def output_dimension(self) -> int:
        return self._output_dim

    def forward(self, x: Tensor) -> Tensor:
        return self._model(x)


# ----------------------------------------------------------------------------------------------------------------
# gated ref/alt MLP (parameters only; evaluated by the HIP kernels)
# ----------------------------------------------------------------------------------------------------------------
class SpatialGatingUnitRefAlt(nn.Module):
    def __init__(self, d_z: int):
        super().__init__()
        self.norm = nn.LayerNorm([d_z // 2])
        self.alpha_ref = nn.Parameter(torch.tensor(0.01))
        self.alpha_alt = nn.Parameter(torch.tensor(0.01))
        self.beta_ref = nn.Parameter(torch.tensor(0.01))
        self.beta_alt = nn.Parameter(torch.tensor(0.01))
        self.gamma = nn.Parameter(torch.tensor(0.01))
        self.ref_regularizer = nn.Parameter(0.1 * torch.ones(d_z // 2))
        self.reg_weight = nn.Parameter(torch.tensor(0.1))
        parametrize.register_parametrization(self, "reg_weight", PositiveNumber())


class GatedRefAltMLPBlock(nn.Module):
    def __init__(self, d_model: int, d_ffn: int):
        super().__init__()
        assert d_ffn % 2 == 0, "the spatial gating unit splits d_ffn in two"
        self.norm = nn.LayerNorm([d_model])
        self.activation = nn.SELU()
        self.proj1_ref = nn.Linear(d_model, d_ffn)
        self.proj1_alt = nn.Linear(d_model, d_ffn)
        self.sgu = SpatialGatingUnitRefAlt(d_ffn)
        self.proj2_ref = nn.Linear(d_ffn // 2, d_model)
        self.proj2_alt = nn.Linear(d_ffn // 2, d_model)
        self.size = d_model


class GatedRefAltMLP(nn.Module):
    def __init__(self, d_model: int, d_ffn: int, num_blocks: int):
        super().__init__()
        self.blocks = nn.ModuleList([GatedRefAltMLPBlock(d_model, d_ffn) for _ in range(num_blocks)])
        self.dimension, self.d_ffn = d_model, d_ffn

    def input_dimension(self) -> int:
        return self.dimension

    def output_dimension(self) -> int:
        return self.dimension


# ----------------------------------------------------------------------------------------------------------------
# rotation + clustering head (parameters only)
# ----------------------------------------------------------------------------------------------------------------
class EuclideanTransformation(nn.Module):
    def __init__(self, dimension: int):
        super().__init__()
        self.translation_e = nn.Parameter(torch.rand(dimension))
        self.rotation_ee = orthogonal(nn.Linear(dimension, dimension, bias=False))


class ExponentiallyModifiedGaussian(nn.Module):
    def __init__(self, num_distributions: int):
        super().__init__()
        self.mu_k = nn.Parameter(2 * torch.ones(num_distributions))
        self.sigma_k = nn.Parameter(torch.ones(num_distributions))
        parametrize.register_parametrization(self, "sigma_k", BoundedNumber(constants.MIN_STDEV, constants.MAX_STDEV))
        self.lambda_k = nn.Parameter(torch.ones(num_distributions))
        parametrize.register_parametrization(self, "lambda_k", BoundedNumber(constants.MIN_LAMBDA, constants.MAX_LAMBDA))


class FeatureClustering(nn.Module):
    def __init__(self, feature_dimension: int, num_artifact_clusters: int):
        super().__init__()
        self.feature_dim, self.num_artifact_clusters = feature_dimension, num_artifact_clusters
        stdev = BoundedNumber(constants.MIN_STDEV, constants.MAX_STDEV)
        self.nonartifact_stdev_e = nn.Parameter(torch.ones(feature_dimension))
        parametrize.register_parametrization(self, "nonartifact_stdev_e", stdev)
        self.artifact_directions_ke = nn.Parameter(torch.rand(num_artifact_clusters, feature_dimension))
        parametrize.register_parametrization(self, "artifact_directions_ke", UnitVector())
        self.artifact_emg = ExponentiallyModifiedGaussian(num_artifact_clusters)
        self.artifact_stdev_k = nn.Parameter(torch.ones(num_artifact_clusters))
        parametrize.register_parametrization(self, "artifact_stdev_k", stdev)
        self.log_cluster_weights_k = nn.Parameter(torch.ones(num_artifact_clusters))
        parametrize.register_parametrization(self, "log_cluster_weights_k", LogWeights())


# ----------------------------------------------------------------------------------------------------------------
# adversarial wrapper with gradient reversal
# ----------------------------------------------------------------------------------------------------------------
class _ReverseGradient(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, alpha: float):
        ctx.alpha = alpha
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad_output: Tensor):
        return -ctx.alpha * grad_output, None


class GradientReversal(nn.Module):
    def __init__(self, alpha: float = 1.0):
        super().__init__()
        self.alpha = alpha

    def set_alpha(self, alpha_new: float):
        self.alpha = alpha_new

    def forward(self, x: Tensor) -> Tensor:
        return _ReverseGradient.apply(x, self.alpha)


class Adversarial(nn.Module):
    def __init__(self, wrapped_module: nn.Module, adversarial_strength: float = 1.0):
        super().__init__()
        self.wrapped_module = wrapped_module
        self.gradient_reversal = GradientReversal(alpha=adversarial_strength)

    def set_adversarial_strength(self, new_alpha: float):
        self.gradient_reversal.set_alpha(new_alpha)

    def forward(self, x: Tensor) -> Tensor:
        return self.wrapped_module(self.gradient_reversal(x))

    def adversarial_forward(self, x: Tensor) -> Tensor:
        return self.forward(x)


# ----------------------------------------------------------------------------------------------------------------
# haplotype CNN, configured by layer strings
# ----------------------------------------------------------------------------------------------------------------
INITIAL_NUM_CHANNELS = 10  # (A, C, G, T, indel) x (ref, alt), interleaved ref/alt per base


def _conv_len(n, kernel_size=1, stride=1, padding=0, dilation=1, **_):
    return floor((n + 2 * padding - dilation * (kernel_size - 1) - 1) / stride + 1)


def _pool_len(n, kernel_size=1, stride=None, padding=0, dilation=1, **_):
    stride = kernel_size if stride is None else stride
    return floor((n + 2 * padding - dilation * (kernel_size - 1) - 1) / stride + 1)


class DNASequenceConvolution(nn.Module):
    """Layer-string grammar: `<type>[/key=int]...`, types convolution | pool | leaky_relu | selu | batch_norm | flatten | linear."""

    def __init__(self, layer_strings: List[str], sequence_length: int):
        super().__init__()
        channels, length = INITIAL_NUM_CHANNELS, sequence_length
        layers: List[nn.Module] = []
        for spec in layer_strings:
            kind, *rest = spec.split("/")
            kwargs = {k: int(v) for k, v in (tok.split("=") for tok in rest)}
            if kind == "convolution":
                layers.append(nn.Conv1d(in_channels=channels, **kwargs))
                channels, length = kwargs["out_channels"], _conv_len(length, **kwargs)
            elif kind == "pool":
                assert length > 1
                layers.append(nn.MaxPool1d(**kwargs))
                length = _pool_len(length, **kwargs)
            elif kind == "leaky_relu":
                layers.append(nn.LeakyReLU())
            elif kind == "selu":
                layers.append(nn.SELU())
            elif kind == "batch_norm":
                # reference dna_sequence_convolution.py:82-83.  The kernels have no such layer: in eval mode the engine folds its
                # per-channel affine map into the convolution / linear next to it (engine/plan.py: _lower_cnn, cnn_bn_folds); training
                # with batch statistics is refused (ArtifactModel.compute_batch_output)
                layers.append(nn.BatchNorm1d(channels))
            elif kind == "flatten":
                layers.append(nn.Flatten())
                channels, length = channels * length, 1
            elif kind == "linear":
                assert length == 1, "linear layer before flatten"
                layers.append(nn.Linear(in_features=channels, **kwargs))
                channels = kwargs["out_features"]
            else:
                raise ValueError("unsupported layer_type: " + kind)
        assert length == 1, "data have not been flattened"
        self._output_dimension = channels
        self._model = nn.Sequential(*layers)

    def output_dimension(self) -> int:
        return self._output_dimension

    def forward(self, x: Tensor) -> Tensor:
        return self._model(x)
