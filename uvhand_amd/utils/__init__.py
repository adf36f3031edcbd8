from .transformer_inputs import (decoder_reference_points, encoder_reference_points, flatten_feature_levels,
                                 get_valid_ratio)

__all__ = ["flatten_feature_levels", "get_valid_ratio", "encoder_reference_points", "decoder_reference_points"]
