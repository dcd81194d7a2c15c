"""MI355X-native engine for the image-restoration platform's queue-worker hot path.

classify -> restore -> optional <=3-view fusion, as hand-written HIP kernels for gfx950 behind
the C ABI of include/ire.h, with host-side mirrors of the reference's service seams
(ClassifierService.analyze, PromptEnhancerService.enhance, GeminiClient.restoreImage,
RestoratorService.restore/restoreBatch).
"""
__version__ = "0.1.0"
