"""Model side of the hot path: the baseline CRNN behind the reference's
constructor and forward signatures (nnet/CRNN.py, nnet/CNN.py, nnet/RNN.py)."""
from .CRNN import CRNN  # noqa: F401
