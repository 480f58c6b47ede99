"""The build's own caller harness: tiny-Llama model + budgets + generate loop (hipGraph decode)."""
from .generation import (GraphedDecoder, apply_pattern, apply_pyramid_pattern, decode_n_tokens, decode_one_token,  # noqa: F401
                         generate, greedy, negotiate_graphed_decoder, normalize_cache_length, prefill, setup_caches)
from .model import CONFIGS, ModelArgs, Transformer, find_multiple  # noqa: F401
