"""stcat_amd — MI355X-native hot path of STCAT (spatio-temporal video grounding).

Two ways in:

* drop-in:  with a jy0205/STCAT checkout on ``sys.path``::

      import stcat_amd; stcat_amd.install()      # before models.build_model(cfg)

  rebinds the three factory names that ``models/pipeline.py:6-8`` imports
  (``build_vis_encoder``, ``build_encoder``, ``build_decoder``) so ``STCATNet`` is built from the
  HIP-backed modules; every reference file stays byte-identical.

* standalone: ``stcat_amd.pipeline.build_model()`` — same module tree, loss and post-processor,
  no reference checkout needed (what tests and bench.py use on the GPU box).

All arithmetic runs in ``lib/libstcat_hip.so`` (C ABI: include/stcat_hip.h); there is no CPU fallback.
"""
from . import synth  # noqa: F401  (pure numpy/torch helpers; no GPU needed)

__version__ = "0.1.0"


def install():
    """Rebind the reference's factory seam.  Requires the reference package ``models``
    to be importable; must run before ``STCATNet`` is constructed."""
    import importlib

    from .backbone import build_vis_encoder
    from .grounding import build_decoder, build_encoder

    pipeline = importlib.import_module("models.pipeline")
    vision = importlib.import_module("models.vision_model")
    grounding = importlib.import_module("models.grounding_model")
    pipeline.build_vis_encoder = build_vis_encoder
    pipeline.build_encoder = build_encoder
    pipeline.build_decoder = build_decoder
    vision.build_vis_encoder = build_vis_encoder
    grounding.build_encoder = build_encoder
    grounding.build_decoder = build_decoder
    return pipeline
