"""`core.*` of the reference, served by rnd_semantic_segmentation_amd.host (see dropin.py for the mapping)."""
from rnd_semantic_segmentation_amd import dropin

dropin.install()
