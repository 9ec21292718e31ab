"""visinger_amd -- MI355X-native (gfx950) implementation of VISinger's variational-inference hot path.

`visinger_amd.modules.*` mirrors the reference's ``modules/visinger/*``, ``modules/rel_transformer.py`` and
``modules/commons/utils.py`` (same class names, constructor/forward signatures and state_dict keys); the arithmetic
runs in ``csrc/libvisinger_hip.so`` through the C ABI of ``include/visinger_hip.h``.  ``install_as_reference_modules``
makes the reference's own ``models/visinger.py`` import these classes unchanged.
"""
import sys

__version__ = "0.1.0"

_ALIASES = {
    "modules.visinger.encoder": "visinger_amd.modules.visinger.encoder",
    "modules.visinger.flow": "visinger_amd.modules.visinger.flow",
    "modules.visinger.decoder": "visinger_amd.modules.visinger.decoder",
    "modules.visinger.predictor": "visinger_amd.modules.visinger.predictor",
    "modules.rel_transformer": "visinger_amd.modules.rel_transformer",
    "modules.commons.utils": "visinger_amd.modules.commons.utils",
    "modules.discriminator": "visinger_amd.modules.discriminator",
    "models.commons.align_ops": "visinger_amd.models.commons.align_ops",
}


def install_as_reference_modules():
    """Register this package's modules under the reference's import paths, so that the reference's
    ``models/visinger.py`` (``from modules.visinger.encoder import TextEncoder ...``, models/visinger.py:6-12)
    builds a VISinger whose hot path runs on the MI355X.  Call before importing ``models.visinger``."""
    import importlib
    for ref_name, ours in _ALIASES.items():
        sys.modules[ref_name] = importlib.import_module(ours)
