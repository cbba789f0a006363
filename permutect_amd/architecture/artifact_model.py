"""ArtifactModel: drop-in for the reference class of the same name, computed on MI355X by hand-written HIP kernels.

Same constructor, method names, return containers, state_dict keys and `.pt` dictionary as
reference permutect/architecture/artifact_model.py:97-369.  What is different is where the
arithmetic runs: `calculate_features` + `FeatureClustering.calculate_logits` + `RaggedSets.means_over_sets` are ONE fused
HIP launch (permutect_amd/csrc/pmt_forward.hip) and their backward is one fused launch (pmt_backward.hip), behind
`ReadSetFunction`.  The per-variant branches (info MLP, haplotype CNN, adversaries, losses) are O(B) work evaluated with
torch ops on the same ROCm device.  There is no CPU compute path: on a CPU device the compute methods raise.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

from permutect_amd import constants
from permutect_amd.architecture import modules as M
from permutect_amd.data.batch import Batch
from permutect_amd.data.datum import Data
from permutect_amd.engine import lib as L
from permutect_amd.engine.runtime import (HaplotypeCnnFunction, LossesFunction, PhiFunction, ReadSetEngine, ReadSetFunction,  # noqa: F401
                                          RowsMlpFunction, VariantEmbedFunction)
from permutect_amd.enums import Epoch
from permutect_amd.parameters import ModelParameters, install_pickle_alias

_BCE = nn.BCEWithLogitsLoss(reduction="none")


def gpu_if_available() -> torch.device:
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


class BatchOutput:
    """reference artifact_model.py:38-73"""

    def __init__(self, features_be: Tensor, ref_features_be: Tensor, logits_b: Tensor, logits_bk: Tensor,
                 weights: Tensor, source_weights: Tensor):
        self.features_be = features_be
        self.ref_features_be = ref_features_be
        self.logits_b = logits_b
        self.logits_bk = logits_bk
        self.weights = weights
        self.source_weights = source_weights

    # derived on demand (the reference computes both eagerly; the fused loss kernel needs neither)
    @property
    def artifact_probs_b(self) -> Tensor:
        return torch.sigmoid(self.logits_b)

    @property
    def outlier_binary_logits(self) -> Tensor:
        lk = self.logits_bk
        nonoutlier = torch.logsumexp(torch.cat((lk[:, 0:1], lk[:, 2:]), dim=-1), dim=-1)
        return lk[:, 1] - nonoutlier


class BatchLosses:
    """reference artifact_model.py:76-90 (total_loss is a SUM over the batch)"""

    def __init__(self, supervised_losses_b, unsupervised_losses_b, alt_count_losses_b, source_prediction_losses_b,
                 total_losses_b):
        self.supervised_losses_b = supervised_losses_b
        self.unsupervised_losses_b = unsupervised_losses_b
        self.alt_count_losses_b = alt_count_losses_b
        self.source_prediction_losses_b = source_prediction_losses_b
        self.total_losses_b = total_losses_b
        self.total_loss = torch.sum(total_losses_b)


class SetMeansView:
    """What callers of `calculate_features` consume from the reference's RaggedSets: the per-set means."""

    def __init__(self, means_be: Tensor, lengths_b: Tensor):
        self._means, self.lengths_b = means_be, lengths_b

    def means_over_sets(self) -> Tensor:
        return self._means

    def batch_size(self) -> int:
        return len(self.lengths_b)


class ArtifactModel(nn.Module):
    def __init__(self, params: ModelParameters, num_read_features: int, num_info_features: int,
                 haplotypes_length: int, device=None):
        super().__init__()
        if device is None:
            device = gpu_if_available()
        self._device = torch.device(device)
        if self._device.type == "cuda" and self._device.index is None:
            self._device = torch.device("cuda", torch.cuda.current_device())
        self._dtype = torch.float32  # reference data/datum.py:37-38: fp32 on every device
        self._haplotypes_length = haplotypes_length
        self._params = params

        self.read_embedding = M.MLP([num_read_features] + params.read_layers, params.batch_normalize, params.dropout_p)
        self.info_embedding = M.MLP([num_info_features] + params.info_layers, params.batch_normalize, params.dropout_p)
        self.haplotypes_cnn = M.DNASequenceConvolution(params.ref_seq_layer_strings, haplotypes_length // 2)
        embedding_dim = (self.read_embedding.output_dimension() + self.info_embedding.output_dimension()
                         + self.haplotypes_cnn.output_dimension())
        self.ref_alt_reads_encoder = M.GatedRefAltMLP(embedding_dim, params.self_attention_hidden_dimension,
                                                      params.num_self_attention_layers)
        self.reducer = M.MLP([embedding_dim] + params.aggregation_layers, params.batch_normalize, params.dropout_p)
        e = self.reducer.output_dimension()
        self.pre_clustering_transform = M.EuclideanTransformation(e)
        self.feature_clustering = M.FeatureClustering(e, params.num_artifact_clusters)
        self.alt_count_predictor = M.Adversarial(M.MLP([e] + [30, -1, -1, -1, 1]), adversarial_strength=0.01)
        self.alt_count_loss_func = nn.MSELoss(reduction="none")
        self.source_predictor = M.Adversarial(M.MLP([e] + [1], params.batch_normalize, params.dropout_p), 0.01)
        self.num_sources = 1

        self.to(device=self._device, dtype=self._dtype)
        self._engine: Optional[ReadSetEngine] = None

    # ---- engine ---------------------------------------------------------------------------------------------------
    def engine(self) -> ReadSetEngine:
        """Flattens the parameters into one buffer and lowers the model to the device descriptor (once)."""
        if self._engine is None:
            # (built as ordinary tensors also when the first forward happens under torch.inference_mode: the flat buffers are
            #  written in place by later training steps, and their version counters key the packed-weight cache)
            with torch.inference_mode(False):
                self._engine = ReadSetEngine(self, self._device)
        return self._engine

    def _invalidate_engine(self):
        self._engine = None

    def reset_source_predictor(self, num_sources: int = 1):
        hidden = [] if num_sources == 1 else [-1, -1]
        layers = [self.reducer.output_dimension()] + hidden + [num_sources]
        self.source_predictor = M.Adversarial(
            M.MLP(layers, self._params.batch_normalize, self._params.dropout_p), 0.01
        ).to(device=self._device, dtype=self._dtype)
        self.num_sources = num_sources
        self._invalidate_engine()  # new leaves: the flat parameter space is rebuilt on next use

    def ref_alt_seq_embedding_dimension(self) -> int:
        return self.haplotypes_cnn.output_dimension()

    def haplotypes_length(self) -> int:
        return self._haplotypes_length

    def calibration_parameters(self):
        fc = self.feature_clustering
        return [fc.parametrizations.nonartifact_stdev_e.original, fc.parametrizations.artifact_stdev_k.original]

    def set_epoch_type(self, epoch_type: Epoch):
        train = epoch_type == Epoch.TRAIN
        self.train(train)
        for p in self.parameters():
            p.requires_grad_(train)

    def forward(self, batch: Batch):
        pass

    # ---- forward ------------------------------------------------------------------------------------------------------
    def variant_embedding(self, batch: Batch) -> Tensor:
        """[B, E_info + E_hap]: the per-variant part of every read's input (reference artifact_model.py:244-246).
        The info MLP is a HIP row kernel (pmt_rows_forward); requires packed weights to be current (see _encode)."""
        eng = self.engine()
        return VariantEmbedFunction.apply(eng, batch.get_info_be(), batch.get_haplotypes_bs(), eng.trigger)

    def _cnn_has_batchnorm(self) -> bool:
        """(asked on every training forward; the module tree does not change under a model: looked at once)"""
        flag = self.__dict__.get("_cnn_bn_flag")
        if flag is None:
            flag = self.__dict__["_cnn_bn_flag"] = any(isinstance(m, nn.BatchNorm1d) for m in self.haplotypes_cnn.modules())
        return flag

    def _encode(self, batch: Batch):
        if self.training and self._cnn_has_batchnorm():
            # the reference's `batch_norm` token of the haplotype CNN (dna_sequence_convolution.py:82-83): same story as below
            raise NotImplementedError("permutect_amd runs a haplotype CNN with batch_norm tokens in eval mode only (model.eval() under "
                                      "no_grad / inference_mode: filter_variants, evaluation); training with BatchNorm statistics is not built")
        if self.training and self._params.batch_normalize:
            # reference mlp.py:52-53: BatchNorm1d normalises with the statistics of the whole batch in train mode.  The kernels run its
            # eval-mode form (running statistics folded into the Linear behind it, engine/plan.py): refuse, never train on the wrong map.
            raise NotImplementedError("permutect_amd runs a batch_normalize model in eval mode only (model.eval(): filter_variants, "
                                      "evaluation); training with BatchNorm statistics is not built")
        eng = self.engine()
        # reference mlp.py:57-58: nn.Dropout draws new masks on every forward in train mode and is the identity in eval mode
        eng.draw_dropout_seed(self.training)
        if batch.size() == 0:  # nothing to launch
            d, dev = eng.plan.desc, self._device
            z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)  # noqa: E731
            return (z(0), z(0, d.num_clusters + 2), z(0, d.feature_dim), z(0, d.feature_dim)), z(0, d.variant_embed_dim)
        # weights -> parametrizations (phi) -> MFMA fragment order (packed).  While the parameters do not change -- every forward
        # of filter_variants and of an evaluation pass -- both are reused: two launches and their gaps off every step.
        key = None if torch.is_grad_enabled() else eng.params_key()  # (a training forward re-packs anyway: no need to prove anything unchanged)
        if key is not None and eng.packed_for is not None and eng.packed_for[0] == key:
            phi = eng.packed_for[1]
        else:
            prog = eng.plan.phi_program(self)
            phi = eng.plan.materialize_phi(self) if prog is None else PhiFunction.apply(eng, prog, eng.trigger)
            eng.pack(phi.detach().contiguous())  # once per forward when training, before any kernel uses the weights
            eng.packed_for = (key, phi.detach()) if (key is not None and not torch.is_grad_enabled()) else None
        variant_embed = self.variant_embedding(batch)
        outs = ReadSetFunction.apply(eng, batch, phi, variant_embed)
        return outs, variant_embed

    def calculate_features(self, batch: Batch, weight_range: float = 0):
        (logits_b, logits_bk, feats, ref_feats), ve = self._encode(batch)
        e_hap = self.haplotypes_cnn.output_dimension()
        return (SetMeansView(ref_feats, batch.get(Data.REF_COUNT)), SetMeansView(feats, batch.get(Data.ALT_COUNT)),
                ve[:, ve.shape[1] - e_hap:])

    def compute_batch_output(self, batch: Batch, balancer=None) -> BatchOutput:
        (logits_b, logits_bk, feats, ref_feats), _ = self._encode(batch)
        if balancer is None:
            weights_b = source_weights = torch.ones_like(logits_b)  # (1 * 1: one fill instead of two and a product)
        elif logits_b.is_cuda and hasattr(balancer, "weights_from_logits"):
            weights_b, source_weights = balancer.weights_from_logits(batch, logits_b)  # two launches (pmt_balance_step)
        else:
            weights_b, source_weights_b = balancer.process_batch_and_compute_weights(
                batch, artifact_probs_b=torch.sigmoid(logits_b).detach())
            source_weights = weights_b * source_weights_b
        return BatchOutput(features_be=feats, ref_features_be=ref_feats, logits_b=logits_b, logits_bk=logits_bk,
                           weights=weights_b, source_weights=source_weights)

    # ---- losses (reference artifact_model.py:267-325) --------------------------------------------------------------------
    def compute_source_prediction_losses(self, features_be: Tensor, batch: Batch) -> Tensor:
        if self.num_sources > 1:
            eng = self.engine()
            logits = RowsMlpFunction.apply(eng, L.ROWS_SOURCE, features_be, eng.trigger,
                                           float(self.source_predictor.gradient_reversal.alpha))
            probs = torch.softmax(logits, dim=-1)
            targets = torch.nn.functional.one_hot(batch.get(Data.SOURCE).long(), self.num_sources)
            return torch.sum(torch.square(probs - targets), dim=-1)
        return torch.zeros(batch.size(), device=self._device, dtype=self._dtype)

    def compute_alt_count_losses(self, features_be: Tensor, batch: Batch) -> Tensor:
        eng = self.engine()
        raw = RowsMlpFunction.apply(eng, L.ROWS_ALT_COUNT, features_be, eng.trigger,
                                    float(self.alt_count_predictor.gradient_reversal.alpha))
        pred = torch.sigmoid(raw.view(-1))
        target = batch.get(Data.ALT_COUNT).to(dtype=pred.dtype) / constants.MAX_ALT_COUNT
        return self.alt_count_loss_func(pred, target)

    def compute_batch_losses(self, output: BatchOutput, batch: Batch) -> BatchLosses:
        """All five per-variant loss vectors in one launch (pmt_losses_forward); the two adversaries' MLPs are row
        kernels with the gradient reversal folded into their input gradient."""
        eng = self.engine()
        alt_raw = RowsMlpFunction.apply(eng, L.ROWS_ALT_COUNT, output.features_be, eng.trigger,
                                        float(self.alt_count_predictor.gradient_reversal.alpha))
        source_logits, sources = None, None
        if self.num_sources > 1:
            source_logits = RowsMlpFunction.apply(eng, L.ROWS_SOURCE, output.features_be, eng.trigger,
                                                  float(self.source_predictor.gradient_reversal.alpha))
            sources = batch.get(Data.SOURCE).long()
        sup, unsup, alt, src, total = LossesFunction.apply(
            eng, output.logits_b, output.logits_bk, alt_raw, source_logits, batch.get(Data.LABEL).long(),
            batch.get(Data.ALT_COUNT).long(), sources, output.weights, output.source_weights)
        return BatchLosses(sup, unsup, alt, src, total)

    def compute_batch_losses_torch(self, output: BatchOutput, batch: Batch) -> BatchLosses:
        """The same losses as the reference composes them from torch ops (test reference for the fused kernels)."""
        labels_b = batch.get_training_labels()
        is_labeled_b = batch.get_is_labeled_mask()
        supervised = is_labeled_b * _BCE(output.logits_b, labels_b)
        clipped = torch.clip(output.outlier_binary_logits, max=constants.MAX_OUTLIER_LOGIT)
        unsupervised = (1 - is_labeled_b) * _BCE(clipped, torch.zeros_like(clipped))
        alt_count = self.compute_alt_count_losses(output.features_be, batch)
        source = self.compute_source_prediction_losses(output.features_be, batch)
        total = output.weights * (supervised + unsupervised + alt_count) + output.source_weights * source
        return BatchLosses(supervised, unsupervised, alt_count, source, total)

    # ---- checkpoint format (reference artifact_model.py:327-369) ---------------------------------------------------------
    def make_dict_for_saving(self, artifact_log_priors=None, artifact_spectra=None):
        # clone: the live tensors are views into one flat buffer and torch.save would serialise the whole storage
        state = {k: v.detach().clone() for k, v in self.state_dict().items()}
        return {
            constants.STATE_DICT_NAME: state,
            constants.HYPERPARAMS_NAME: self._params,
            constants.NUM_READ_FEATURES_NAME: self.read_embedding.input_dimension(),
            constants.NUM_INFO_FEATURES_NAME: self.info_embedding.input_dimension(),
            constants.REF_SEQUENCE_LENGTH_NAME: self.haplotypes_length(),
            constants.ARTIFACT_LOG_PRIORS_NAME: artifact_log_priors,
            constants.ARTIFACT_SPECTRA_STATE_DICT_NAME: None if artifact_spectra is None else artifact_spectra.state_dict(),
        }

    def save_model(self, path, artifact_log_priors=None, artifact_spectra=None):
        self.reset_source_predictor()
        install_pickle_alias()
        torch.save(self.make_dict_for_saving(artifact_log_priors, artifact_spectra), path)


def load_model(path, device: torch.device = None):
    if device is None:
        device = gpu_if_available()
    install_pickle_alias()
    # the restricted unpickler: tensors, containers and the hyperparameter class only (the reference loads with
    # weights_only=False, architecture/artifact_model.py:352; a checkpoint is untrusted input)
    with torch.serialization.safe_globals([ModelParameters]):
        saved = torch.load(path, map_location=device, weights_only=True)
    model = ArtifactModel(saved[constants.HYPERPARAMS_NAME], num_read_features=saved[constants.NUM_READ_FEATURES_NAME],
                          num_info_features=saved[constants.NUM_INFO_FEATURES_NAME],
                          haplotypes_length=saved[constants.REF_SEQUENCE_LENGTH_NAME], device=device)
    model.load_state_dict(saved[constants.STATE_DICT_NAME])
    model.to(model._dtype)
    return model, saved[constants.ARTIFACT_LOG_PRIORS_NAME], saved[constants.ARTIFACT_SPECTRA_STATE_DICT_NAME]
