"""MI355X counterparts of the reference's layer zoo (utils/net_utils.py), same class names, constructor
arguments, forward() signatures and state_dict keys; every forward is a sequence of libriders_hip.so launches.

Reference: utils/net_utils.py  activation_func :4, Conv2d :29, UpConv2d :156, FullyConnected :201,
ResNetBlock :253, DecoderBlock :473, OutlierRemoval :575.

torch.nn.Conv2d / BatchNorm2d / Linear instances are used purely as parameter containers (they give the
reference's default initialisation and key names); their own forward() is never called.
"""
import torch

from . import engine
from .engine import ACT_LRELU, ACT_NONE, ACT_RELU


def activation_func(activation_fn):
    """Same selector as utils/net_utils.py:4-27 (LeakyReLU slope 0.20)."""
    if 'linear' in activation_fn:
        return None
    elif 'leaky_relu' in activation_fn:
        return torch.nn.LeakyReLU(negative_slope=0.20, inplace=True)
    elif 'relu' in activation_fn:
        return torch.nn.ReLU()
    elif 'elu' in activation_fn:
        return torch.nn.ELU()
    elif 'sigmoid' in activation_fn:
        return torch.nn.Sigmoid()
    else:
        raise ValueError('Unsupported activation function: {}'.format(activation_fn))


def _act_code(fn):
    if fn is None:
        return ACT_NONE, 0.0
    if isinstance(fn, torch.nn.LeakyReLU):
        return ACT_LRELU, float(fn.negative_slope)
    if isinstance(fn, torch.nn.ReLU):
        return ACT_RELU, 0.0
    raise NotImplementedError('activation %s has no HIP kernel on this path' % type(fn).__name__)


def _init_weight(weight, weight_initializer):
    # utils/net_utils.py:72-77: only these three re-initialise; 'kaiming_uniform' keeps torch's default
    if weight_initializer == 'kaiming_normal':
        torch.nn.init.kaiming_normal_(weight)
    elif weight_initializer == 'xavier_normal':
        torch.nn.init.xavier_normal_(weight)
    elif weight_initializer == 'xavier_uniform':
        torch.nn.init.xavier_uniform_(weight)


class _BNCounter:
    """num_batches_tracked is bumped on the host and flushed into the buffer when a state_dict is taken."""

    def _bn_setup(self):
        self._nbt_pending = 0
        self.register_state_dict_pre_hook(lambda m, prefix, keep_vars: m._bn_flush())

    def _bn_flush(self):
        if getattr(self, '_nbt_pending', 0) and getattr(self, 'use_batch_norm', False):
            self.batch_norm.num_batches_tracked += self._nbt_pending
            self._nbt_pending = 0


class Conv2d(torch.nn.Module, _BNCounter):
    """conv (bias=False, pad=k//2) -> BatchNorm2d -> activation.  Reference: utils/net_utils.py:29-91."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, weight_initializer='kaiming_uniform',
                 activation_func=torch.nn.LeakyReLU(negative_slope=0.10, inplace=True), use_batch_norm=False):
        super(Conv2d, self).__init__()
        self.use_batch_norm = use_batch_norm
        padding = kernel_size // 2
        self.conv = torch.nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                                    bias=False)
        _init_weight(self.conv.weight, weight_initializer)
        self.activation_func = activation_func
        if self.use_batch_norm:
            self.batch_norm = torch.nn.BatchNorm2d(out_channels)
        self._bn_setup()

    def _fwd(self, x, x2=None, up=None, lazy=0):
        """lazy = 1 / 2 (callers whose consumer is another conv_block / add_act): at that engine.set_lazy_bn level the BatchNorm-ed output
        stays virtual (engine.LazyAct) and the consumer applies scale / shift / activation while it stages the raw convolution output."""
        act, slope = _act_code(self.activation_func)
        bn = self.batch_norm if self.use_batch_norm else None
        if bn is not None and self.training:
            self._nbt_pending += 1
        return engine.conv_block(x, self.conv.weight, x2=x2, stride=self.conv.stride[0], pad=self.conv.padding[0], up=up,
                                 bn=bn, act=act, slope=slope, training=self.training, lazy_out=lazy)

    def forward(self, x):
        def run(x):
            return engine.to_nchw_out(self._fwd(engine.from_nchw(x)), x.dtype)
        return engine.run_region(run, (x,), list(self.parameters()))


class UpConv2d(torch.nn.Module):
    """F.interpolate(x, size=shape) (nearest) -> Conv2d; the upsample is folded into the conv gather.
    Reference: utils/net_utils.py:156-198."""

    def __init__(self, in_channels, out_channels, kernel_size=3, weight_initializer='kaiming_uniform',
                 activation_func=torch.nn.LeakyReLU(negative_slope=0.10, inplace=True), use_batch_norm=False):
        super(UpConv2d, self).__init__()
        self.conv = Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=1,
                           weight_initializer=weight_initializer, activation_func=activation_func,
                           use_batch_norm=use_batch_norm)

    def _fwd(self, x, shape, lazy=0):
        return self.conv._fwd(x, up=(int(shape[0]), int(shape[1])), lazy=lazy)

    def forward(self, x, shape):
        def run(x):
            return engine.to_nchw_out(self._fwd(engine.from_nchw(x), shape), x.dtype)
        return engine.run_region(run, (x,), list(self.parameters()))


class FullyConnected(torch.nn.Module):
    """Linear(bias=True) -> activation (-> dropout).  Reference: utils/net_utils.py:201-247."""

    def __init__(self, in_features, out_features, weight_initializer='kaiming_uniform',
                 activation_func=torch.nn.LeakyReLU(negative_slope=0.10, inplace=True), dropout_rate=0.00):
        super(FullyConnected, self).__init__()
        self.fully_connected = torch.nn.Linear(in_features, out_features)
        _init_weight(self.fully_connected.weight, weight_initializer)
        self.activation_func = activation_func
        if dropout_rate > 0.00 and dropout_rate <= 1.00:
            raise NotImplementedError('dropout is not used on the RIDERS hot path (dropout_rate=0 everywhere)')
        self.dropout = None

    def _fwd(self, x):
        act, slope = _act_code(self.activation_func)
        return engine.linear(x, self.fully_connected.weight, bias=self.fully_connected.bias, act=act, slope=slope)

    def forward(self, x):
        def run(x):
            return self._fwd(x if x.is_contiguous() else x.contiguous())
        return engine.run_region(run, (x,), list(self.parameters()))


class ResNetBlock(torch.nn.Module):
    """act(conv2(conv1(x)) + X), X = projection(x) iff shapes differ.  Reference: utils/net_utils.py:253-323.
    `projection` is always constructed (and therefore present in the state_dict) as in the reference."""

    def __init__(self, in_channels, out_channels, stride=1, weight_initializer='kaiming_uniform',
                 activation_func=torch.nn.LeakyReLU(negative_slope=0.10, inplace=True), use_batch_norm=False):
        super(ResNetBlock, self).__init__()
        self.activation_func = activation_func
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=3, stride=stride,
                            weight_initializer=weight_initializer, activation_func=activation_func,
                            use_batch_norm=use_batch_norm)
        self.conv2 = Conv2d(out_channels, out_channels, kernel_size=3, stride=1,
                            weight_initializer=weight_initializer, activation_func=activation_func,
                            use_batch_norm=use_batch_norm)
        self.projection = Conv2d(in_channels, out_channels, kernel_size=1, stride=stride,
                                 weight_initializer=weight_initializer, activation_func=None, use_batch_norm=False)

    def _fwd(self, x):
        conv1 = self.conv1._fwd(x, lazy=1)       # consumed by conv2's staging
        conv2 = self.conv2._fwd(conv1, lazy=1)   # consumed by the fused apply + add + activation below
        if tuple(x.shape[1:3]) != tuple(conv2.shape[1:3]) or x.shape[3] != conv2.shape[3]:
            X = self.projection._fwd(x)
        else:
            X = x
        act, slope = _act_code(self.activation_func)
        return engine.add_act(conv2, X, act, slope)

    def forward(self, x):
        def run(x):
            return engine.to_nchw_out(self._fwd(engine.from_nchw(x)), x.dtype)
        return engine.run_region(run, (x,), list(self.parameters()))


class DecoderBlock(torch.nn.Module):
    """UpConv2d to the skip's size -> cat([deconv, skip]) -> Conv2d; upsample and concat are never materialised.
    Reference: utils/net_utils.py:473-569 (deconv_type 'up' only, as configured by RCNetModel)."""

    def __init__(self, in_channels, skip_channels, out_channels, weight_initializer='kaiming_uniform',
                 activation_func=torch.nn.LeakyReLU(negative_slope=0.10, inplace=True), use_batch_norm=False,
                 deconv_type='up'):
        super(DecoderBlock, self).__init__()
        self.skip_channels = skip_channels
        self.deconv_type = deconv_type
        if deconv_type != 'up':
            raise NotImplementedError("only deconv_type='up' is on the RIDERS hot path (rcnet_model.py:94)")
        self.deconv = UpConv2d(in_channels, out_channels, kernel_size=3, weight_initializer=weight_initializer,
                               activation_func=activation_func, use_batch_norm=use_batch_norm)
        concat_channels = skip_channels + out_channels
        self.conv = Conv2d(concat_channels, out_channels, kernel_size=3, stride=1,
                           weight_initializer=weight_initializer, activation_func=activation_func,
                           use_batch_norm=use_batch_norm)

    def _fwd(self, x, skip=None, shape=None, lazy=0):
        if skip is not None:
            shape = tuple(skip.shape[1:3])
        elif shape is None:
            shape = (int(2 * x.shape[1]), int(2 * x.shape[2]))
        deconv = self.deconv._fwd(x, shape, lazy=2)      # consumed by self.conv's staging (level 2: slower on MI355X, see engine.set_lazy_bn)
        if self.skip_channels > 0:
            return self.conv._fwd(deconv, x2=skip, lazy=lazy)
        return self.conv._fwd(deconv, lazy=lazy)

    def forward(self, x, skip=None, shape=None):
        ins = (x,) if skip is None else (x, skip)

        def run(x, skip=None):
            s = None if skip is None else engine.from_nchw(skip)
            return engine.to_nchw_out(self._fwd(engine.from_nchw(x), s, shape), x.dtype)
        return engine.run_region(run, ins, list(self.parameters()))


class OutlierRemoval(object):
    """Min-filter based outlier removal of sparse depth.  Reference: utils/net_utils.py:575-638."""

    def __init__(self, kernel_size=7, threshold=1.5):
        self.kernel_size = kernel_size
        self.threshold = threshold

    def remove_outliers(self, depth):
        d = depth if depth.is_contiguous() else depth.contiguous()
        return engine.outlier_removal(d.float(), self.kernel_size, self.threshold)
