"""permutect_amd: MI355X-native engine for the Permutect artifact-model hot path.

Scope: ArtifactModel forward / backward / clip+AdamW as used by
train_artifact_model and filter_variants.  The compute path is hand-written HIP for
gfx950 behind a C ABI (include/permutect_amd.h); this package is the thin Python host
that mirrors the reference's ArtifactModel / Batch interface.
"""

__version__ = "0.1.0"
