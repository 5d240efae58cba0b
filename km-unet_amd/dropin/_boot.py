"""Makes `import km_unet_amd` work from the drop-in shims regardless of the caller's sys.path."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _root not in sys.path:
    sys.path.insert(0, _root)
import km_unet_amd  # noqa: E402,F401
