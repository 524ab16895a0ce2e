"""castrec_amd -- MI355X-native SASRec/CAST training hot path.

Drop-in for the hot path of Spijkervet/Context-Aware-Sequential-Recommendation
(``modules.py``, ``models/*.py``, ``sampler.py``, ``main.py`` train/eval loop)
built on hand-written HIP kernels for gfx950 behind the C ABI declared in
``include/castrec.h``.  There is no CPU fallback: constructing a model or a
sampler without the built native libraries raises.
"""
__version__ = "0.1.0"
PKG_DIR = __import__("os").path.dirname(__import__("os").path.abspath(__file__))
