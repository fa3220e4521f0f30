"""Parameter containers for the conv stack, named so that a state_dict saved
by the reference (nnet/CNN.py:33-100) loads unchanged: ``cnn.conv{i}``,
``cnn.batchnorm{i}``, ``cnn.glu{i}.linear`` — plus the parameter-free
``dropout{i}`` / ``pooling{i}`` entries the reference's Sequential also holds.

These modules only own tensors.  The arithmetic of the stack runs in
csrc/crnn.hip and csrc/conv_split.h (default: conv0_split_kernel and
conv_split_kernel, float32 operands as f16 hi/lo halves on the f16 matrix
cores; exact-float32 mode: conv0_kernel and conv_mfma_kernel) and is reached
through ``CRNN.forward``; calling a container directly is an error, not a CPU
path.
"""
import torch.nn as nn


class GLU(nn.Module):
    """Holds the C x C linear of the reference's GLU (nnet/CNN.py:5-16):
    out = linear(x over channels) * sigmoid(x)."""

    def __init__(self, input_num):
        super().__init__()
        self.sigmoid = nn.Sigmoid()
        self.linear = nn.Linear(input_num, input_num)

    def forward(self, x):
        raise RuntimeError("GLU is evaluated inside the fused HIP conv block; call CRNN.forward")


class CNN(nn.Module):
    def __init__(self, n_in_channel, activation="Relu", conv_dropout=0, kernel_size=(3, 3, 3),
                 padding=(1, 1, 1), stride=(1, 1, 1), nb_filters=(64, 64, 64),
                 pooling=((1, 4), (1, 4), (1, 4)), normalization="batch", **transformer_kwargs):
        super().__init__()
        if activation.lower() != "glu" or normalization != "batch":
            raise NotImplementedError("only activation='glu' with batch normalisation is implemented "
                                      "(the configuration of training.py:207-212)")
        self.nb_filters = list(nb_filters)
        self.pooling = [tuple(p) for p in pooling]
        self.kernel_size, self.padding, self.stride = list(kernel_size), list(padding), list(stride)
        stack = nn.Sequential()
        c_in = n_in_channel
        for i, c_out in enumerate(self.nb_filters):
            stack.add_module(f"conv{i}", nn.Conv2d(c_in, c_out, kernel_size[i], stride[i], padding[i]))
            stack.add_module(f"batchnorm{i}", nn.BatchNorm2d(c_out, eps=0.001, momentum=0.99))
            stack.add_module(f"glu{i}", GLU(c_out))
            if conv_dropout is not None:
                stack.add_module(f"dropout{i}", nn.Dropout(conv_dropout))
            stack.add_module(f"pooling{i}", nn.AvgPool2d(self.pooling[i]))
            c_in = c_out
        self.cnn = stack

    def forward(self, x):
        raise RuntimeError("the conv stack runs inside CRNN.forward (HIP); it has no standalone CPU path")
