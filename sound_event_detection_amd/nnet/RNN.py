"""Parameter container for the recurrent part, named like the reference's
``BidirectionalGRU`` (nnet/RNN.py:7-30) so that ``rnn.rnn.weight_ih_l0`` ...
load unchanged.  The recurrence itself runs in the HIP kernels (default:
linear_split_kernel in csrc/conv_split.h + gru_split_kernel in csrc/crnn.hip;
exact-float32 mode: linear_kernel + gru_kernel) through ``CRNN.forward``."""
import torch.nn as nn


class BidirectionalGRU(nn.Module):
    def __init__(self, n_in, n_hidden, dropout=0, num_layers=1):
        super().__init__()
        self.rnn = nn.GRU(n_in, n_hidden, bidirectional=True, dropout=dropout, batch_first=True,
                          num_layers=num_layers)

    def forward(self, input_feat):
        raise RuntimeError("the GRU runs inside CRNN.forward (HIP); it has no standalone CPU path")
