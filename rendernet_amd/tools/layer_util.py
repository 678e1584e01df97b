"""Mirror of the reference's tools/layer_util.py on top of the HIP operators.

Same names, argument order and defaults as the reference builders; tensors are torch HIP
tensors in TF's channels-last layout; variables live in rendernet_amd.variables (TF names).
Differences forced by the move off TF graph mode are limited to:
  * `weight_initializer_type` takes the initialiser factories of rendernet_amd.variables;
  * every builder accepts the extra keyword arguments `activation_alpha` / `residual` /
    `sigmoid` (and `elu` on conv3d_transpose), used by the fused call sites (prelu / tf.add /
    tf.nn.sigmoid / tf.nn.elu folded into the conv epilogue).  Calling `prelu(conv3d(...))` unfused, as the reference spells it, also works.
"""
from .. import ops
from .. import variables as V
from ..variables import xavier_initializer, random_normal_initializer, constant_initializer  # noqa: F401


def _store():
    return V.get_default_store()


def prelu(x, trainable=True, alpha=None):
    """tools/layer_util.py:27-45.  Creates/loads `alpha` [C_last] (init 0) in the current scope."""
    a, _ = _store().get_variable('alpha', shape=[x.shape[-1]],
                                 initializer=alpha if alpha is not None else constant_initializer(0.0))
    return ops.prelu(x, a)


def _alpha_var(channels):
    a, _ = _store().get_variable('alpha', shape=[channels], initializer=constant_initializer(0.0))
    return a


def get_weight(weight_name, weight_dict):
    """tools/layer_util.py:47-58."""
    if weight_dict is None:
        return None
    return weight_dict.get(weight_name)


def bias_variable(shape, bias_initializer=None, trainable=True):
    """tools/layer_util.py:133-144: constant 0.001 unless an initialiser array is given."""
    b, _ = _store().get_variable('biases', shape=shape,
                                 initializer=bias_initializer if bias_initializer is not None
                                 else constant_initializer(0.001))
    return b


def keep_prob(dropout, train):
    """tools/layer_util.py:124-131."""
    return dropout if train else 1.0


def _conv_vars(scope, wshape, if_bias, weight_initializer, bias_initializer, weight_initializer_type,
               nout, default_bias=0.001):
    st = _store()
    with st.variable_scope(scope):
        w, wname = st.get_variable('weights', shape=wshape,
                                   initializer=weight_initializer if weight_initializer is not None
                                   else weight_initializer_type)
        b = None
        if if_bias:
            b, _ = st.get_variable('biases', shape=[nout],
                                   initializer=bias_initializer if bias_initializer is not None
                                   else constant_initializer(default_bias))
    return w, wname, b


def conv3d(input_, num_outputs, pad="SAME", reuse=False, kernel_size=[4, 4, 4], stride=[2, 2, 2], if_bias=True,
           trainable=True, scope="conv3d", weight_initializer=None, bias_initializer=None,
           weight_initializer_type=random_normal_initializer(stddev=0.02),
           activation_alpha=None, residual=None, sigmoid=False, _carry=False):
    """tools/layer_util.py:228-265.  (_carry: see ops._conv_apply -- res_block_3d's first conv)"""
    assert pad == "SAME", "only SAME padding is used by the reference nets"
    w, wname, b = _conv_vars(scope, list(kernel_size) + [input_.shape[-1], num_outputs], if_bias,
                             weight_initializer, bias_initializer, weight_initializer_type, num_outputs)
    pw = _store().packed(wname, lambda: ops.pack_conv(w))
    return ops.conv3d(input_, pw, b, activation_alpha, residual, tuple(stride), sigmoid, carry=_carry)


def conv2d(input_, num_outputs, kernel_size=[4, 4], stride=[1, 1], pad='SAME', if_bias=True, trainable=True,
           reuse=False, scope='conv2d', weight_initializer=None, bias_initializer=None,
           weight_initializer_type=random_normal_initializer(stddev=0.02),
           activation_alpha=None, residual=None, sigmoid=False, default_bias=0.001, _carry=False):
    """tools/layer_util.py:147-183.  (_carry: see ops._conv_apply -- res_block_2d's first conv)"""
    assert pad == "SAME"
    w, wname, b = _conv_vars(scope, list(kernel_size) + [input_.shape[-1], num_outputs], if_bias,
                             weight_initializer, bias_initializer, weight_initializer_type, num_outputs, default_bias)
    pw = _store().packed(wname, lambda: ops.pack_conv(w))
    return ops.conv2d(input_, pw, b, activation_alpha, residual, tuple(stride), sigmoid, carry=_carry)


def conv2d_transpose(x, num_outputs, kernel_size=(4, 4), stride=(1, 1), pad='SAME', if_bias=True, reuse=False,
                     scope="conv2d_transpose", trainable=True, weight_initializer=None, bias_initializer=None,
                     weight_initializer_type=random_normal_initializer(stddev=0.02),
                     activation_alpha=None, residual=None, sigmoid=False, default_bias=0.001):
    """tools/layer_util.py:186-226.  Filter layout [kh,kw,Cout,Cin] (:201)."""
    assert pad == "SAME"
    w, wname, b = _conv_vars(scope, list(kernel_size) + [num_outputs, x.shape[-1]], if_bias,
                             weight_initializer, bias_initializer, weight_initializer_type, num_outputs, default_bias)
    pw = _store().packed(wname, lambda: ops.pack_conv_transpose(w, stride[0]))
    return ops.conv2d_transpose(x, pw, b, activation_alpha, residual, tuple(stride), sigmoid)


def conv3d_transpose(x, num_output, kernel_size=(4, 4, 4), stride=(1, 1, 1), pad='SAME', if_bias=True, reuse=False,
                     scope="conv3d_transpose", trainable=True, weight_initializer=None, bias_initializer=None,
                     weight_initializer_type=random_normal_initializer(stddev=0.02),
                     activation_alpha=None, residual=None, sigmoid=False, elu=False):
    """tools/layer_util.py:269-309.  Filter layout [k1,k2,k3,Cout,Cin] (:284).  `elu` folds the tf.nn.elu of the
    shape decoder (Reconstruct_RenderNet_Face.py:49-68) into the epilogue."""
    assert pad == "SAME"
    w, wname, b = _conv_vars(scope, list(kernel_size) + [num_output, x.shape[-1]], if_bias,
                             weight_initializer, bias_initializer, weight_initializer_type, num_output)
    pw = _store().packed(wname, lambda: ops.pack_conv_transpose(w, stride[0]))
    return ops.conv3d_transpose(x, pw, b, activation_alpha, residual, tuple(stride), sigmoid, elu)


def fully_connected(input_, output_size, reuse=False, scope='fully_connected', if_bias=True, weight_initializer=None,
                    bias_initializer=None, trainable=True,
                    weight_initializer_type=random_normal_initializer(stddev=0.02), activation_alpha=None):
    """tools/layer_util.py:311-343."""
    w, _, b = _conv_vars(scope, [input_.shape[1], output_size], if_bias, weight_initializer, bias_initializer,
                         weight_initializer_type, output_size)
    return ops.fully_connected(input_, w, b, activation_alpha)


_RELU_SLOPE = {}


def _relu_slope(channels, device):
    """tf.nn.relu through the PReLU epilogue: a CONSTANT all-zero slope vector (max(0,x) + 0*min(0,x)), cached per
    (device, width) -- not a variable: the pretrained res blocks have no `alpha` (tools/layer_util.py:75-88, :107-121)."""
    import torch
    key = (str(device), int(channels))
    t = _RELU_SLOPE.get(key)
    if t is None:
        t = _RELU_SLOPE[key] = ops.mark_constant(torch.zeros(int(channels), dtype=torch.float32, device=device))
    return t


def res_block_3d(input, out_channels=64, scope='res_block', kernel=[3, 3, 3], stride=[1, 1, 1], weight_dict=None,
                 trainable=True):
    """tools/layer_util.py:60-88: input + conv3d(act(conv3d(input))).  The two branches of the reference are two different
    functions: WITHOUT a weight_dict (:66-73) the activation is prelu with an `alpha` variable in the block scope (:69 ->
    :35-40); WITH one (:75-88, the pretrained nets of Reconstruct_RenderNet_Face.py:150-159) it is tf.nn.relu and NO alpha
    variable exists.  Activation and residual add run in the epilogues of the two conv launches."""
    wd = weight_dict
    with _store().variable_scope(scope):
        alpha = _alpha_var(out_channels) if wd is None else _relu_slope(out_channels, input.device)
        # (the block's input comes back from the first conv as a second output: the skip path's gradient then joins dx inside that conv's
        # input-gradient launch instead of through a separate add)
        net, input = conv3d(input, out_channels, kernel_size=kernel, stride=stride, pad="SAME", scope="con1_3X3",
                            weight_initializer=get_weight(scope + '_con1_3X3_weights', wd),
                            bias_initializer=get_weight(scope + '_con1_3X3_biases', wd),
                            weight_initializer_type=xavier_initializer(), activation_alpha=alpha, _carry=True)
        net = conv3d(net, out_channels, kernel_size=kernel, stride=stride, pad="SAME", scope="conv2_3x3",
                     weight_initializer=get_weight(scope + '_conv2_3x3_weights', wd),
                     bias_initializer=get_weight(scope + '_conv2_3x3_biases', wd),
                     weight_initializer_type=xavier_initializer(), residual=input)
    return net


def res_block_2d(input, out_channels=64, scope='res_block', kernel=[3, 3], stride=[1, 1], weight_dict=None,
                 trainable=True):
    """tools/layer_util.py:91-121.  Without a weight_dict (:98-105): slim.conv2d (zero-initialised biases) + prelu with an
    `alpha` variable; with one (:107-121; Reconstruct_RenderNet_Face.py:183-192, :213-217): the hand-rolled conv2d (bias
    constant 0.001 unless the dict has it) + tf.nn.relu, no alpha variable."""
    wd = weight_dict
    with _store().variable_scope(scope):
        alpha = _alpha_var(out_channels) if wd is None else _relu_slope(out_channels, input.device)
        db = 0.0 if wd is None else 0.001
        net, input = conv2d(input, out_channels, kernel_size=kernel, stride=stride, scope="con1_3X3",
                            weight_initializer=get_weight(scope + '_con1_3X3_weights', wd),
                            bias_initializer=get_weight(scope + '_con1_3X3_biases', wd),
                            weight_initializer_type=xavier_initializer(), activation_alpha=alpha, default_bias=db, _carry=True)
        net = conv2d(net, out_channels, kernel_size=kernel, stride=stride, scope="conv2_3x3",
                     weight_initializer=get_weight(scope + '_conv2_3x3_weights', wd),
                     bias_initializer=get_weight(scope + '_conv2_3x3_biases', wd),
                     weight_initializer_type=xavier_initializer(), residual=input, default_bias=db)
    return net


def res_stack_2d(input, out_channels, n_blocks, scope_fmt='res_%d', kernel=[3, 3], skip_scope=None, skip_residual=None,
                 skip_default_bias=0.0, weight_dict=None):
    """`n_blocks` res_block_2d in a row (scopes scope_fmt % 1 .. n_blocks) and, with `skip_scope`, the conv
    `<skip_scope>/con1_3X3` + `skip_residual` behind them -- the loop of RenderNet_Shader.py:71-84 / :91-99 as one call, so that
    the whole stack can run as one Winograd chain (ops.res_stack_2d).  Variables, names, initialisers and creation order are
    exactly those of the loop over res_block_2d (tools/layer_util.py:91-121) followed by the skip conv."""
    wd = weight_dict
    st = _store()
    blocks = []
    for k in range(1, n_blocks + 1):
        scope = scope_fmt % k
        with st.variable_scope(scope):
            # the two branches of res_block_2d (tools/layer_util.py:98-105 | :107-121): prelu + alpha variable | relu, no alpha
            alpha = _alpha_var(out_channels) if wd is None else _relu_slope(out_channels, input.device)
            packs = []
            for cs in ("con1_3X3", "conv2_3x3"):
                w, wname, b = _conv_vars(cs, list(kernel) + [input.shape[-1], out_channels], True,
                                         get_weight(scope + '_' + cs + '_weights', wd), get_weight(scope + '_' + cs + '_biases', wd),
                                         xavier_initializer(), out_channels, 0.0 if wd is None else 0.001)
                packs.append((st.packed(wname, lambda w=w: ops.pack_conv(w)), b))
            blocks.append((packs[0][0], packs[0][1], alpha, packs[1][0], packs[1][1]))
    skip = None
    if skip_scope is not None:
        with st.variable_scope(skip_scope):
            w, wname, b = _conv_vars("con1_3X3", list(kernel) + [input.shape[-1], out_channels], True,
                                     get_weight(skip_scope + '_con1_3X3_weights', wd), get_weight(skip_scope + '_con1_3X3_biases', wd),
                                     xavier_initializer(), out_channels, skip_default_bias)
            skip = (st.packed(wname, lambda: ops.pack_conv(w)), b, skip_residual)
    return ops.res_stack_2d(input, blocks, skip)


def projection_unit(input, n_features=18, scope='projection_unit'):
    """tools/layer_util.py:8-22: depth-flatten + 1x1 conv + PReLU as ONE kernel reading the 3-D
    tensor in place (rn_projection_fwd).  Variables: <scope>/Conv/{weights,biases}, <scope>/alpha."""
    st = _store()
    n_features = int(input.shape[3] * input.shape[4])           # :19
    with st.variable_scope(scope):
        w, wname, b = _conv_vars('Conv', [1, 1, n_features, n_features], True, None, None,
                                 xavier_initializer(), n_features, default_bias=0.0)
        alpha = _alpha_var(n_features)
        pw = st.packed(wname, lambda: ops.pack_conv(w))
        return ops.projection(input, pw, b, alpha)
