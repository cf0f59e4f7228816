"""Import overlay: `import midasmednet.unet.model` resolves to the MI355X implementation (mednet_hip), while every other
submodule of the reference package (`midasmednet.segmentation`, `.landmarks`, `.dataset`, `.utils`) keeps resolving from
the reference's own tree, so its callers (midasmednet/segmentation.py:16-18, landmarks.py:16-18,
examples/train_seg.py:18) run unchanged.

Put this directory's parent in front of the reference on `sys.path` / `PYTHONPATH`:

    PYTHONPATH=/path/to/torch-mednet_amd:/path/to/torch-mednet python examples/train_seg.py ...

`pkgutil.extend_path` appends the `midasmednet/` directories of all later path entries to this package's `__path__`;
this overlay only ships `unet/`, which therefore wins for `midasmednet.unet.*`, and nothing else is shadowed."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
