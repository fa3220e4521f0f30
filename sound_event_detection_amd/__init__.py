"""MI355X-native implementation of the sound-event-detection inference hot
path (mel frontend -> CRNN -> threshold / event decode) behind the reference's
Python call signatures.  Everything numerical runs in hand-written gfx950 HIP
kernels reached through the C ABI of libsedmi.so (include/sedmi.h)."""

__all__ = ["feature_extraction", "evaluation", "nnet", "synth"]
