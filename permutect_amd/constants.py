"""Checkpoint dictionary keys (mirror of reference permutect/constants.py:1-8)."""

STATE_DICT_NAME = "model_state_dict"
OPTIMIZER_STATE_DICT_NAME = "optimizer_state_dict"
ARTIFACT_LOG_PRIORS_NAME = "artifact_log_priors"
ARTIFACT_SPECTRA_STATE_DICT_NAME = "artifact_spectra_state_dict"
HYPERPARAMS_NAME = "hyperparams"
NUM_READ_FEATURES_NAME = "num_read_features"
NUM_INFO_FEATURES_NAME = "num_info_features"
REF_SEQUENCE_LENGTH_NAME = "ref_sequence_length"

# reference permutect/data/count_binning.py:9-11
MAX_REF_COUNT = 10
MIN_ALT_COUNT = 1
MAX_ALT_COUNT = 15

# reference permutect/architecture/feature_clustering.py:20, artifact_model.py:32
MAX_LOGIT = 20.0
MAX_OUTLIER_LOGIT = 10.0
MIN_STDEV, MAX_STDEV = 0.01, 100.0
MIN_LAMBDA, MAX_LAMBDA = 0.01, 100.0
