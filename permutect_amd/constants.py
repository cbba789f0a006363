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

# command-line flag names (reference permutect/constants.py:15-62): the `train_artifact_model` CLI takes the reference's flags
INPUT_NAME, OUTPUT_NAME = "input", "output"
READ_LAYERS_NAME = "read_layers"
SELF_ATTENTION_HIDDEN_DIMENSION_NAME = "self_attention_hidden_dimension"
NUM_SELF_ATTENTION_LAYERS_NAME = "num_self_attention_layers"
INFO_LAYERS_NAME = "info_layers"
AGGREGATION_LAYERS_NAME = "aggregation_layers"
NUM_ARTIFACT_CLUSTERS_NAME = "num_artifact_clusters"
CALIBRATION_LAYERS_NAME = "calibration_layers"
REF_SEQ_LAYER_STRINGS_NAME = "ref_seq_layer_strings"
DROPOUT_P_NAME = "dropout_p"
LEARNING_RATE_NAME = "learning_rate"
WEIGHT_DECAY_NAME = "weight_decay"
BATCH_NORMALIZE_NAME = "batch_normalize"
TRAIN_TAR_NAME = "train_tar"
REWEIGHTING_RANGE_NAME = "reweighting_range"
BATCH_SIZE_NAME = "batch_size"
NUM_EPOCHS_NAME = "num_epochs"
NUM_CALIBRATION_EPOCHS_NAME = "num_calibration_epochs"
INFERENCE_BATCH_SIZE_NAME = "inference_batch_size"
NUM_WORKERS_NAME = "num_workers"
TENSORBOARD_DIR_NAME = "tensorboard_dir"
PRETRAINED_ARTIFACT_MODEL_NAME = "pretrained_artifact_model"
