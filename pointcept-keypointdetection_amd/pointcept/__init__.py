"""MI355X-native `pointcept` overlay: only the PTv3 serialized-window attention path.

Registers "PT-v3m1", "OffsetKeypointPTv3" and "DefaultSegmentorV2" under pointcept.models.MODELS with
the reference's constructor keywords and state_dict keys (see INTEGRATION.md for grafting this
`models/` package into a reference checkout so tools/train.py / tools/infer_offset.py run unchanged).
"""
