"""MI355X-native (gfx950) implementation of the DFCNN(+SE)+CTC / Transformer hot path of
786440445/ASR_DFCNN_Transformer.  Host code mirrors the reference's Python API surface
(hparams, data_loader, model classes); all arithmetic runs in hand-written HIP kernels
behind the C ABI declared in include/asr_hip.h (libasrhip.so, loaded via ctypes).
There is no CPU fallback: importing the compute modules without the built library fails."""

__version__ = '0.1.0'
