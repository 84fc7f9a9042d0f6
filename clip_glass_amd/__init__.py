"""clip_glass_amd — MI355X-native CLIP-GLaSS fitness-evaluation engine (host side).

The product path is the HIP engine behind the C ABI in include/glass.h; this
package is the Python mirror of the reference's own interface for that path
(problem.GenerationProblem / generator.Generator / latent.* / config / operators).
It never imports oracle/.
"""
__version__ = "0.1.0"
