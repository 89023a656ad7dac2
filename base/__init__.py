"""`base.*` of the reference (plugin API), served by rnd_semantic_segmentation_amd.host.plugin."""
from rnd_semantic_segmentation_amd import dropin

dropin.install()
