"""Function-patching helper (same contract as flash_attention_ipu/_impl/utils.py:10-39).

`patch_function(orig_fn, modules)` returns a decorator.  The decorated function receives the
original function as its first argument; a wrapper carrying the original's metadata
(`functools.wraps`, so `.__wrapped__ is orig_fn` - the reference's tests un-patch through it,
tests/python/test_scaled_dot_product_attention.py:125) is installed under the original's name on
every module in `modules`.
"""

import functools
import logging
from typing import Any, Callable, Sequence

logger = logging.getLogger(__name__)


def patch_function(orig_fn: Callable, modules_to_patch: Sequence[Any]):
    name = orig_fn.__name__

    def decorate(replacement: Callable):
        @functools.wraps(orig_fn)
        def patch_wrapper(*args, **kwargs):
            return replacement(orig_fn, *args, **kwargs)

        for module in modules_to_patch:
            logger.info("flash_attention_amd: patching %s!", name)
            setattr(module, name, patch_wrapper)
        return patch_wrapper

    return decorate
