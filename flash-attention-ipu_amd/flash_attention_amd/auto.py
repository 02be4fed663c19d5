"""Importing this module patches torch.nn.functional.scaled_dot_product_attention
(reference: flash_attention_ipu/auto.py:3)."""

from ._impl.scaled_dot_product_attention import _scaled_dot_product_attention_amd  # noqa: F401
