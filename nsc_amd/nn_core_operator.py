"""Drop-in for the reference's nn_core_operator.py (file:line cited per function): same names, positional order,
keyword names and defaults, channels_last ``[B, T, C]`` tensors.  The arithmetic behind every function is a HIP
kernel of libnsc_hip.so (see include/nsc_hip.h); variables are created implicitly in the current
``nsc_amd.scope.variable_scope`` exactly where the reference's TF layers create theirs.

``activation`` accepts ``None``, the strings 'tanh' / 'lrelu', or the module-level ``tanh`` sentinel that stands in
for ``tf.nn.tanh`` (the reference's default argument, nn_core_operator.py:6).
"""
from __future__ import annotations

import torch

from . import ops
from .scope import current_store

tanh = "tanh"   # stands in for tf.nn.tanh in `activation=` arguments
FUSED_BLOCKS = True   # gated_bottleneck as one fused kernel per direction where the shape allows (False: composed op by op)


def _act_name(activation):
    if activation is None:
        return None
    if activation in ("tanh", "lrelu", "none"):
        return None if activation == "none" else activation
    raise ValueError(f"unsupported activation {activation!r} (None, 'tanh' or 'lrelu')")


def _conv1d_variables(cin, num_filters, filter_size):
    """The kernel / bias pair tf.compat.v1.layers.conv1d creates (glorot-uniform kernel, zero bias), under the next free 'conv1d_N'."""
    st = current_store()
    name = st.uniq("conv1d")
    w = st.get(name + "/kernel", (filter_size, cin, num_filters), st.glorot(filter_size * cin, filter_size * num_filters))
    b = st.get(name + "/bias", (num_filters,), lambda s: torch.zeros(s).numpy())
    return w, b


def conv1d(inputs, num_filters, filter_size, padding='SAME', dilation_rate=1, strides=1, activation=tanh):
    """nn_core_operator.py:6-14 (tf.compat.v1.layers.conv1d, channels_last, glorot-uniform kernel, zero bias)."""
    if padding != 'SAME':
        raise ValueError("only padding='SAME' is used by the reference and implemented")
    w, b = _conv1d_variables(int(inputs.shape[-1]), num_filters, filter_size)
    return ops.Conv1dFn.apply(inputs, w, b, int(dilation_rate), int(strides), _act_name(activation))


def conv1d_depth(inputs, num_filters, filter_size, padding='SAME', dilation_rate=1, strides=1, activation=tanh):
    """nn_core_operator.py:17-21 (tf.keras.layers.SeparableConv1D: depthwise [K,C,1] -> pointwise [1,C,F] -> +bias)."""
    if padding != 'SAME' or dilation_rate != 1 or strides != 1:
        raise ValueError("SeparableConv1D is used with SAME / dilation 1 / stride 1 by the reference; others unsupported")
    st = current_store()
    name = st.uniq("separable_conv1d")
    c = int(inputs.shape[-1])
    wd = st.get(name + "/depthwise_kernel", (filter_size, c, 1), st.glorot(filter_size * c, filter_size))
    wp = st.get(name + "/pointwise_kernel", (1, c, num_filters), st.glorot(c, num_filters))
    b = st.get(name + "/bias", (num_filters,), lambda s: torch.zeros(s).numpy())
    dw = ops.DepthwiseFn.apply(inputs, wd.reshape(filter_size, c))
    return ops.Conv1dFn.apply(dw, wp, b, 1, 1, _act_name(activation))


def conv1d_depth_shuffle(inputs, num_filters, filter_size, activation=tanh, stride=2):
    """conv1d_depth(...) followed by the sub-pixel shuffle of neural_speech_coding_module.py:158-167 - the body of the reference's
    _up_sampling_mod (:169-181) - creating conv1d_depth's variables.  The shipped shape (C -> C in {100, 50}, 9 taps, stride 2) is ONE
    autograd node on the engine's fused up-sampling kernels (ops.UpsampleFn); anything else: the two calls."""
    c = int(inputs.shape[-1])
    if FUSED_BLOCKS and stride == 2 and filter_size == 9 and num_filters == c and c in (100, 50):
        st = current_store()
        name = st.uniq("separable_conv1d")
        wd = st.get(name + "/depthwise_kernel", (filter_size, c, 1), st.glorot(filter_size * c, filter_size))
        wp = st.get(name + "/pointwise_kernel", (1, c, num_filters), st.glorot(c, num_filters))
        b = st.get(name + "/bias", (num_filters,), lambda s: torch.zeros(s).numpy())
        return ops.UpsampleFn.apply(inputs, wd, wp, b, _act_name(activation))
    assert stride == 2
    return ops.ShuffleFn.apply(conv1d_depth(inputs, num_filters, filter_size, activation=activation))


def activation_func(_x):
    """nn_core_operator.py:24-31: tf.nn.leaky_relu (alpha 0.2); the PReLU / ELU / ReLU variants are commented out there."""
    return ops.ActFn.apply(_x, "lrelu")


def batch_norm(_x, training):
    """nn_core_operator.py:34-40: identity in the reference."""
    return _x


def change_channel(the_input, wide_layer=30, the_channel=1, kernel_size=9, dilation_rate=1, strides=1, activation=None):
    """nn_core_operator.py:45-54 (dilation hard-coded to 1 there, :52)."""
    return conv1d(the_input, the_channel, filter_size=kernel_size, padding='SAME', dilation_rate=1, strides=strides,
                  activation=activation)


def the_bottleneck(the_input, wide_layer=30, narrow_layer=10, non_dilated_neck_kernel_size=9,
                   dilated_neck_kernel_size=9, dilation_rate=1, is_last_flat=False):
    """nn_core_operator.py:57-79 (dead under the shipped resnet_type='gln'; kept for surface parity)."""
    c = conv1d(the_input, narrow_layer, filter_size=non_dilated_neck_kernel_size, padding='SAME', dilation_rate=1, activation=None)
    c = activation_func(c)
    c = conv1d(c, narrow_layer, filter_size=dilated_neck_kernel_size, padding='SAME', dilation_rate=dilation_rate, activation=None)
    c = activation_func(c)
    c = conv1d(c, wide_layer, filter_size=non_dilated_neck_kernel_size, padding='SAME', dilation_rate=1, activation=None)
    y = ops.AddFn.apply(c, the_input)
    return y if is_last_flat else activation_func(y)


def gated_bottleneck(the_input, wide_layer=30, narrow_layer=10, non_dilated_neck_kernel_size=9, dilated_neck_kernel_size=9,
                     dilation_rate=1, is_last_flat=False, the_share=False):
    """nn_core_operator.py:82-112.  dilated kernel size is the hard-coded 15 of :92/:97 (dilated_neck_kernel_size is
    ignored there too); `the_share` is unused in the reference.

    The shipped shapes (narrow 20, 9-tap output conv, dilation 1 | 2, wide <= 112 with wide input channels, or one input channel into
    wide in {100, 50, 25}) run as ONE fused call per direction (ops.BlockFn: the persistent block kernels the engine uses); the four
    convs' variables are created in the same order and under the same names either way.  FUSED_BLOCKS = False, or any other shape:
    the composed form below, op by op."""
    cin = int(the_input.shape[-1])
    fusable = (FUSED_BLOCKS and narrow_layer == 20 and non_dilated_neck_kernel_size == 9 and int(dilation_rate) in (1, 2) and
               ((cin == wide_layer and 1 < wide_layer <= 112) or (cin == 1 and wide_layer in (100, 50, 25))))
    if fusable:
        return ops.BlockFn.apply(the_input, *_block_variables(cin, wide_layer, narrow_layer, non_dilated_neck_kernel_size),
                                 int(dilation_rate), bool(is_last_flat))
    c2 = conv1d(the_input, narrow_layer, filter_size=1, padding='SAME', dilation_rate=1, activation=None)
    c2 = activation_func(c2)
    left = conv1d(c2, narrow_layer, filter_size=15, padding='SAME', dilation_rate=dilation_rate, activation=None)
    right = conv1d(c2, narrow_layer, filter_size=15, padding='SAME', dilation_rate=dilation_rate, activation=tanh)
    c3 = ops.MulFn.apply(left, right)
    c2 = conv1d(c3, wide_layer, filter_size=non_dilated_neck_kernel_size, padding='SAME', dilation_rate=1, activation=None)
    y = ops.AddFn.apply(c2, the_input)
    return y if is_last_flat else activation_func(y)


def _block_variables(cin, wide_layer, narrow_layer, k9):
    w1, b1 = _conv1d_variables(cin, narrow_layer, 1)
    wl, bl = _conv1d_variables(narrow_layer, narrow_layer, 15)
    wr, br = _conv1d_variables(narrow_layer, narrow_layer, 15)
    w9, b9 = _conv1d_variables(narrow_layer, wide_layer, k9)
    return w1, b1, wl, bl, wr, br, w9, b9


def gated_bottleneck_stack(the_input, wide_layer, narrow_layer, non_dilated_neck_kernel_size, dilation_rates, is_last_flat=True,
                           the_share=False):
    """len(dilation_rates) gated_bottleneck calls in a row - what the reference's _stack_bottleneck_blocks loop builds
    (neural_speech_coding_module.py:209-216: every block but the last with its leaky-relu, the last one `is_last_flat`) - creating the
    same variables in the same order.  With the shipped shapes the whole stack is ONE autograd node (ops.BlockStackFn: the leaky-relu
    between two blocks is differentiated in the next block's data-gradient kernel); otherwise the calls are made one by one."""
    cin = int(the_input.shape[-1])
    dils = [int(d) for d in dilation_rates]
    fusable = (FUSED_BLOCKS and len(dils) >= 2 and narrow_layer == 20 and non_dilated_neck_kernel_size == 9 and all(d in (1, 2) for d in dils)
               and wide_layer in (100, 50, 25) and cin in (wide_layer, 1))
    if not fusable:
        c = the_input
        for i, d in enumerate(dils):
            c = gated_bottleneck(c, wide_layer=wide_layer, narrow_layer=narrow_layer, non_dilated_neck_kernel_size=non_dilated_neck_kernel_size,
                                 dilated_neck_kernel_size=15, dilation_rate=d, is_last_flat=(is_last_flat if i == len(dils) - 1 else False),
                                 the_share=the_share)
        return c
    params = []
    for i in range(len(dils)):
        params += _block_variables(cin if i == 0 else wide_layer, wide_layer, narrow_layer, non_dilated_neck_kernel_size)
    return ops.BlockStackFn.apply(the_input, *params, tuple(dils), bool(is_last_flat))


def gated_bottleneck_decoder(the_input, wide_layer=30, narrow_layer=10, non_dilated_neck_kernel_size=9,
                             dilated_neck_kernel_size=9, dilation_rate=1, is_last_flat=False, the_share=False):
    """nn_core_operator.py:115-137 (never called by the reference's live paths; kept for surface parity)."""
    c2 = conv1d(the_input, narrow_layer, filter_size=1, padding='SAME', dilation_rate=1, activation=None)
    c2 = activation_func(c2)
    left = conv1d(c2, narrow_layer, filter_size=dilated_neck_kernel_size, padding='SAME', dilation_rate=dilation_rate, activation=None)
    right = conv1d(c2, narrow_layer, filter_size=dilated_neck_kernel_size, padding='SAME', dilation_rate=dilation_rate, activation=tanh)
    c3 = ops.MulFn.apply(left, right)
    c2 = conv1d_depth(c3, wide_layer, filter_size=non_dilated_neck_kernel_size, padding='SAME', dilation_rate=1, activation=None)
    y = ops.AddFn.apply(c2, the_input)
    return y if is_last_flat else activation_func(y)


def scalar_softmax_quantization(floating_code, alpha, bins, is_quan_on, the_share, code_length, num_kmean_kernels):
    """nn_core_operator.py:140-164.  Returns (soft_assignment_3d, bit_code); the first output is ALWAYS the soft
    assignment (:147, :164); `the_share` truthy selects soft codes, falsy the one-hot argmax (tf.cond :154-158)."""
    if int(bins.numel()) != int(num_kmean_kernels):
        raise ValueError(f"bins has {bins.numel()} entries, num_kmean_kernels={num_kmean_kernels}")
    if int(floating_code.shape[1]) != int(code_length):
        raise ValueError(f"code_length {code_length} != floating_code.shape[1] {floating_code.shape[1]}")
    if not torch.is_tensor(alpha):
        alpha = torch.tensor(float(alpha), device=floating_code.device)
    return ops.QuantizeFn.apply(floating_code, alpha, bins, float(is_quan_on), bool(the_share))


def vector_softmax_quantization(floating_code, alpha, bins, is_quan_on, is_share, top_k, code_len):
    """nn_core_operator.py:167-195: dead code in the reference (0 call sites from main.py)."""
    raise NotImplementedError("vector_softmax_quantization is dead code in the reference and out of scope (SURVEY §2 #11)")
