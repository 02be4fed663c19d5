"""flash_attention_amd - MI355X-native drop-in for the forward path of flash_attention_ipu.

Mirrors the reference package surface (flash_attention_ipu/__init__.py:4-26):
importing the package loads the native library (ImportError if it cannot be found) and
re-exports `flash_attention_qkv_packed`; `import flash_attention_amd.auto` installs the
`torch.nn.functional.scaled_dot_product_attention` override.
"""

from ._impl import _native

_native.load_native_library()

from ._impl.flash_attention_qkv_packed import *  # noqa: E402,F401,F403
from ._impl.flash_attention_qkv_packed import __all__  # noqa: E402,F401


def init_graph_dropout_state(device=None, seed=None):
    """Create (or re-seed) the device-resident dropout generator state that graph-captured dropout launches read
    (`_impl/ops.py`; not part of the reference's surface: the reference rejects dropout, sdpa.py:25-28).  Call it - or run one
    eager dropout call - before `torch.cuda.graph` captures a model with attention dropout."""
    from ._impl import ops

    return ops.init_graph_dropout_state(device, seed)
