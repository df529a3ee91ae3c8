"""Host-side mirror of the reference's ``vision_language/engine`` package for the
UML head fine-tune hot path: same module paths, names, argument meaning and
error behaviour, backed by the HIP kernels in ``umlh`` instead of PyTorch ops.

Put this directory (``unpaired-multimodal-learning_amd/``) on ``sys.path`` where
the reference puts ``vision_language/`` and ``from engine.models.head import UML``
etc. resolve to this build.  Only what the hot path needs is here (SURVEY.md
section 8); backbones, CLIP, dataset readers and prompt templates are out of scope.
"""
