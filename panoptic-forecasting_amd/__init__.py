"""MI355X-native background forecasting path (warp/splat + FC-HarDNet-70).

Drop-in for the reference's ``panoptic_forecasting.models.build_model`` registry for
``task: pc_transform`` and ``task: bg`` (+ the fused ``task: bg_forecast``); device work
is done by the C-ABI HIP library ``csrc/libpfhip.so`` (include/pfhip.h).
"""
__version__ = '0.1.0'
