"""The synthetic-frame generators live in the package (uchirp/synth.py); the tests keep their old import name."""
from uchirp.synth import chirp_pair, make_frames  # noqa: F401
