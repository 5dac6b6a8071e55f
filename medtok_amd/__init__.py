"""medtok_amd -- MI355X-native vector-quantisation hot path of MedTok.

Host side: Python classes mirroring the reference's VQ interfaces
(vector_quantization_soft_one_new.py, norm_ema_quantizer.py, loss.py and the
quantize call sites of tokenizer.py / inference.py).  Device side: hand-written
gfx950 kernels behind the C ABI in include/medtok_vq.h, bound with ctypes.
"""
__version__ = "0.1.0"
