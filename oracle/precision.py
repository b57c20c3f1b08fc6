"""ORACLE (test infrastructure only): the rounding points of the HIP path's bf16 throughput mode, as switchable hooks."""
import torch


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(Precision.dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(Precision.dtype).to(g.dtype)


class _RoundGradBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(Precision.dtype).to(g.dtype)


class Precision:
    """Where the bf16 throughput mode of the HIP path rounds (activations stored as bf16 with fp32 accumulation, parameters and
    statistics): every stored activation tensor -- a convolution's output, a BatchNorm+activation(+residual) output, a resampled map --
    and the packed weights of dense convolutions; gradients are rounded at the same tensors.  With `bf16 = False` (default) every hook
    is the identity and the oracle is the plain fp32 restatement.  Used to check the bf16 KERNELS against the same arithmetic: the
    precision loss of bf16 storage itself (BatchNorm in training mode amplifies it by |mean| / std per layer) is not a kernel property."""
    bf16 = False              # switch: emulate the 16-bit mode
    dtype = torch.bfloat16    # its storage type (torch.bfloat16, or torch.float16 for the fp16 mode)

    @staticmethod
    def r(x):
        return _RoundBF16.apply(x) if Precision.bf16 else x

    @staticmethod
    def g(x):
        """Identity forward, gradient rounded to bf16: a tensor that the HIP path never stores in forward (the nearest-upsampled view
        folded into a gather) but whose gradient it does store."""
        return _RoundGradBF16.apply(x) if Precision.bf16 else x

    @staticmethod
    def w(w):
        return w + (w.to(Precision.dtype).to(w.dtype) - w).detach() if Precision.bf16 else w
