"""quantization_analysis_amd — MI355X (gfx950) backend of the mixed-tile quantization-format search path
of johanna-rock/quantization_analysis.

Same user-facing surface as the reference for this path (compression_algorithms registry, Quantizer,
quantization_formats, the wq CLI) plus `--backend hip`, which routes the per-tile BFP quantize + metric
reductions through hand-written HIP kernels in libmtq_hip.so (include/mtq.h).
"""
__version__ = "0.1.0"
