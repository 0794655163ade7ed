"""Stand-ins for the reference's *unchanged* PyTorch networks, for stand-alone use.

The hot path calls three small coordinate MLPs that the reference keeps in
``model/networks`` (north_star: "unchanged"): the SDF field inside
``DMTetGeometry`` (``/root/reference/model/geometry/dmtet.py:187-212``), the
texture field and the DINO-feature field sampled per pixel in ``shade``
(``/root/reference/model/render/render.py:54,61``), plus the light MLP
(``/root/reference/model/render/light.py:169-193``).  When this package is
overlaid on the reference tree those classes are imported from there; when it
runs alone (tests, ``bench.py``) it needs networks of the same architecture and
with the same ``state_dict`` layout (``in_layer.*``, ``mlp.network.{0,2,..}.weight``,
``min_max``) so reference checkpoints load.  Plain PyTorch; rocBLAS / hipBLASLt GEMMs.

Same mathematics as the reference classes, evaluated faster for long point lists on the GPU (all equal to fp32 rounding, see
DESIGN.md "What is not a HIP kernel"): split-K weight gradients (``_LinearSplitK``), ReLU in the GEMM epilogue
(``_LinearReLUSplitK``), the per-image feature as a [B,C] GEMM plus a per-point add folded into the ReLU pass
(``CoordMLP._forward_indexed``), the input stage [x, sin, cos, 1] from one HIP kernel with the first bias as a weight column
(``CoordMLP._fused_input``), and the frequency table kept on the device.  Short lists (the SDF regulariser, which needs double
backward) take the plain torch path.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


# When True every class below evaluates exactly the way the reference's model/networks classes do -- per-point feature
# concatenation (MLPs.py:84-90), one plain Linear per layer, the frequency table copied host -> device in every forward
# (HarmonicEmbedding.py:41), no HIP input stage, no split-K, no fused ReLU epilogues, no per-image feature path: what a maintainer
# gets who overlays only model/geometry + model/render and keeps the reference's own networks.  bench.py --networks reference.
REFERENCE_FORMULATION = False


def reference_formulation(enable=True):
    global REFERENCE_FORMULATION
    REFERENCE_FORMULATION = bool(enable)
    return REFERENCE_FORMULATION


def _activation(name):
    table = {"tanh": nn.Tanh, "sigmoid": nn.Sigmoid, "relu": nn.ReLU}
    if name not in table:
        raise NotImplementedError(name)
    return table[name]()


class HarmonicEmbedding(nn.Module):
    """[sin(2^k s x), cos(2^k s x)], k<n  (reference networks/HarmonicEmbedding.py:33-44)."""

    def __init__(self, n_harmonic_functions=10, scalar=1.0):
        super().__init__()
        self.frequencies = scalar * (2.0 ** torch.arange(n_harmonic_functions))  # plain attribute, not in the state_dict
        self._on_device = {}

    def _frequencies(self, device):
        # the reference copies the table host->device in every forward (HarmonicEmbedding.py:41): a pageable copy that blocks the
        # host until the stream drains.  Same values, copied once per device.
        f = self._on_device.get(device)
        if f is None:
            f = self._on_device[device] = self.frequencies.to(device)
        return f

    def forward(self, x):
        freq = self.frequencies.to(x.device) if REFERENCE_FORMULATION else self._frequencies(x.device)
        ang = (x[..., None] * freq).reshape(*x.shape[:-1], -1)
        return torch.cat((ang.sin(), ang.cos()), dim=-1)


USE_FIELD_STACK = True  # the hidden stack of a field as one autograd node with MFMA input-gradient GEMMs (_FieldStack)
SPLITK_MIN_ROWS = 65536  # point lists at least this long take the split-K weight gradient below
SPLITK_PARTS = 16


class _LinearSplitK(torch.autograd.Function):
    """y = x @ W^T for a long point list x [M,K] (M ~ 2e5) and a small weight W [N,K].

    Forward and the input gradient are the GEMMs autograd would issue.  The weight gradient g^T x is a [N,M] x [M,K] product with
    a tiny output and a huge reduction: as ONE GEMM rocBLAS/hipBLASLt reach 65 TFLOP/s on gfx950 (409 us tuned, 606 us untuned at
    M=204800, N=K=256); cut into SPLITK_PARTS row blocks and issued as a batched GEMM + a sum of the partial products it takes
    222 us (a throw-away micro-benchmark, not kept).  Same sum, different association: equal to fp32 rounding (7e-6 relative).
    The backward is written with differentiable torch ops, so double backward (create_graph=True) still works.
    """

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return x.mm(weight.t())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        gx = g.mm(weight) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            gw = _split_k_weight_grad(g.contiguous(), x)
        return gx, gw


def _split_k_weight_grad(g, x):
    s, m = SPLITK_PARTS, x.shape[0]
    return torch.bmm(g.view(s, m // s, g.shape[1]).transpose(1, 2), x.view(s, m // s, x.shape[1])).sum(0)


_ZERO_BIAS = {}


def _zero_bias(x, weight):
    key = (x.device, weight.shape[0])
    b = _ZERO_BIAS.get(key)
    if b is None:
        b = _ZERO_BIAS[key] = torch.zeros(weight.shape[0], dtype=x.dtype, device=x.device)
    return b


class _LinearReLUSplitK(torch.autograd.Function):
    """relu(x @ W^T + b) for a long point list: the bias add and the ReLU ride in the GEMM epilogue (hipBLASLt through
    torch._addmm_activation: 235 us against 215 + 85 us for GEMM + a separate ReLU pass at M=204800, N=K=256; bit-identical
    values), the weight gradient is the split-K batched GEMM of _LinearSplitK."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        y = torch._addmm_activation(_zero_bias(x, weight) if bias is None else bias, x, weight.t(), use_gelu=False)
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        g = torch.ops.aten.threshold_backward(g.contiguous(), y, 0)
        gx = g.mm(weight) if ctx.needs_input_grad[0] else None
        gw = _split_k_weight_grad(g, x) if ctx.needs_input_grad[1] else None
        gb = g.sum(0) if ctx.needs_input_grad[2] else None
        return gx, gw, gb


class _FieldStack(torch.autograd.Function):
    """The hidden stack of a coordinate field over a long point list as ONE autograd node:

        y0 = relu(x_emb W_in^T)                                   in_layer (bias folded into W_in's last column)
        y1 = relu(y0 W_first^T + per_image[index])                only for fields with a per-image feature
        y_i = relu(y_{i-1} W_i^T)                                 the 256 -> 256 hidden layers

    Forward as before (ReLU in the GEMM epilogue).  Backward: every input-gradient GEMM is csrc/gemm.hip (hand-written fp32 MFMA)
    with the previous layer's ReLU adjoint in its epilogue, g_{i-1} = (g_i W_i) * (y_{i-1} > 0), so of the n+1 threshold_backward
    passes over [M,256] only the first (for the stack's own output) is left, and the per-image layer needs a plain segment sum instead
    of the masked one.  Weight gradients are the split-K batched GEMMs.  First-order only (the regulariser's short lists never get here).
    """

    @staticmethod
    def forward(ctx, x_emb, w_in, per_image, index, w_first, *hidden):
        from . import ops

        zero = _zero_bias(x_emb, w_in)
        ys = [torch._addmm_activation(zero, x_emb, w_in.t(), use_gelu=False)]
        if w_first is not None:
            y = ys[0].mm(w_first.t())
            ops.rows_add_relu_raw_(y, per_image, index)
            ys.append(y)
        for w in hidden:
            ys.append(torch._addmm_activation(zero, ys[-1], w.t(), use_gelu=False))
        ctx.save_for_backward(x_emb, w_in, index, w_first, *hidden, *ys)
        ctx.n_hidden, ctx.n_images = len(hidden), (per_image.shape[0] if per_image is not None else 0)
        return ys[-1]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from . import ops

        saved = ctx.saved_tensors
        x_emb, w_in, index, w_first = saved[:4]
        hidden, ys = saved[4:4 + ctx.n_hidden], saved[4 + ctx.n_hidden:]
        g = torch.ops.aten.threshold_backward(g.contiguous(), ys[-1], 0)
        g_hidden = []
        for i in range(ctx.n_hidden - 1, -1, -1):
            y_prev = ys[len(ys) - ctx.n_hidden - 1 + i]
            g_hidden.append(_split_k_weight_grad(g, y_prev))
            g = ops.gemm_nn_relumask(g, hidden[i], y_prev)
        g_hidden.reverse()
        g_rows = g_first = None
        if w_first is not None:
            g_rows = ops.rows_segsum_raw(g, index, ctx.n_images)
            g_first = _split_k_weight_grad(g, ys[0])
            g = ops.gemm_nn_relumask(g, w_first, ys[0])
        g_in = _split_k_weight_grad(g, x_emb)
        g_x = g.mm(w_in) if ctx.needs_input_grad[0] else None
        return (g_x, g_in, g_rows, None, g_first, *g_hidden)


def _long_list(x, weight):
    return (x.is_cuda and x.dim() == 2 and x.shape[0] >= SPLITK_MIN_ROWS and x.shape[0] % SPLITK_PARTS == 0 and x.is_contiguous()
            and x.dtype == torch.float32 and torch.is_grad_enabled() and weight.requires_grad)


def linear(x, weight, bias=None):
    """F.linear, with the split-K weight gradient for long point lists on the GPU."""
    if bias is None and _long_list(x, weight):
        return _LinearSplitK.apply(x, weight)
    return F.linear(x, weight, bias)


def linear_relu(x, weight, bias=None):
    """relu(F.linear(x, weight, bias)); epilogue-fused with the split-K weight gradient for long point lists on the GPU."""
    if _long_list(x, weight):
        return _LinearReLUSplitK.apply(x, weight, bias)
    if (x.is_cuda and x.dim() == 2 and x.shape[0] >= SPLITK_MIN_ROWS and x.dtype == torch.float32
            and not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)))):
        return torch._addmm_activation(_zero_bias(x, weight) if bias is None else bias, x, weight.t(), use_gelu=False)  # e.g. the SDF grid pass
    return torch.relu_(F.linear(x, weight, bias))


class MLP(nn.Module):
    """Bias-free Linear/ReLU stack (reference networks/MLPs.py:9-32)."""

    def __init__(self, cin, cout, num_layers, nf=256, dropout=0, activation=None):
        super().__init__()
        assert num_layers >= 1
        dims = [cin] + [nf] * (num_layers - 1) + [cout]
        layers = []
        for i in range(num_layers):
            if i > 0:
                layers.append(nn.ReLU(inplace=True))
            layers.append(nn.Linear(dims[i], dims[i + 1], bias=False))
            if dropout and 0 < i < num_layers - 1:
                layers.append(nn.Dropout(dropout))
        if activation is not None:
            layers.append(_activation(activation))
        self.network = nn.Sequential(*layers)

    def forward(self, x, first_weight=None, per_image=None, index=None):
        """Plain stack; or, with ``first_weight`` [nf,K], ``per_image`` [B,nf] and ``index`` [P] (CoordMLP's per-image feature
        path), the first Linear is evaluated as x @ first_weight^T + per_image[index], the addend folded into the ReLU pass."""
        if REFERENCE_FORMULATION and first_weight is None:
            return self.network(x)
        layers = list(self.network)
        i = 0
        if first_weight is not None:
            assert isinstance(layers[0], nn.Linear) and layers[0].bias is None
            x = linear(x, first_weight)
            if len(layers) > 1 and isinstance(layers[1], nn.ReLU) and x.is_cuda:
                from . import ops

                x = ops.rows_add_relu_(x, per_image, index)  # one in-place pass: + per_image[index], ReLU
                i = 2
            else:
                x = x + per_image[index]
                i = 1
        while i < len(layers):
            layer = layers[i]
            if isinstance(layer, nn.Linear) and i + 1 < len(layers) and isinstance(layers[i + 1], nn.ReLU):
                x = linear_relu(x, layer.weight, layer.bias)
                i += 2
                continue
            x = linear(x, layer.weight, layer.bias) if isinstance(layer, nn.Linear) else layer(x)
            i += 1
        return x


class CoordMLP(nn.Module):
    """Harmonic embedding -> in_layer (+feat concat) -> MLP (reference networks/MLPs.py:35-101)."""

    def __init__(self, cin, cout, num_layers, nf=256, dropout=0, activation=None, min_max=None, n_harmonic_functions=10,
                 embedder_scalar=1, embed_concat_pts=True, extra_feat_dim=0, symmetrize=False, in_layer_relu=False):
        super().__init__()
        self.extra_feat_dim = extra_feat_dim
        if n_harmonic_functions > 0:
            self.embedder = HarmonicEmbedding(n_harmonic_functions, embedder_scalar)
            dim_in = cin * 2 * n_harmonic_functions + (cin if embed_concat_pts else 0)
            self.embed_concat_pts = embed_concat_pts
        else:
            self.embedder = None
            dim_in = cin
        self.in_layer = nn.Linear(dim_in, nf)
        self.relu = nn.ReLU(inplace=True)
        self.mlp = MLP(nf + extra_feat_dim, cout, num_layers, nf, dropout, activation)
        self.symmetrize = symmetrize
        if min_max is not None:
            self.register_buffer("min_max", min_max)
        else:
            self.min_max = None
        self.bsdf = None
        self.in_layer_relu = in_layer_relu

    @property
    def indexed_feat(self):  # sample(x, feat=[B,C], feat_index=[P]) is understood (not in the reference formulation)
        return not REFERENCE_FORMULATION

    def forward(self, x, feat=None, feat_index=None):
        """``feat_index`` (int64 [P], optional, not in the reference signature): ``feat`` is then one row per IMAGE and point p uses
        row feat_index[p].  The reference concatenates the per-point copy of the feature to the hidden vector and multiplies by
        the [nf, nf+C] weight (MLPs.py:84-90); with W = [W_h | W_f] that product is relu(h) W_h^T + (relu(feat) W_f^T)[index]:
        the feature half becomes a [B,C] x [C,nf] GEMM and a row gather instead of a [P,C] x [C,nf] GEMM over 2e5 points
        (and its two backward GEMMs), and the [P, nf+C] concatenation disappears.  Same sum, equal to fp32 rounding."""
        assert (feat is None and self.extra_feat_dim == 0) or (feat.shape[-1] == self.extra_feat_dim)
        if REFERENCE_FORMULATION:
            return self._forward_reference(x, feat if feat_index is None or feat is None else feat[feat_index])
        if feat_index is not None and feat is not None and x.dim() == 2 and isinstance(self.mlp.network[0], nn.Linear) \
                and self.mlp.network[0].bias is None:
            return self._forward_indexed(x, feat, feat_index)
        if feat_index is not None and feat is not None:
            feat = feat[feat_index]
        if feat is None and self._stack_ok(x, False):
            return self._stacked(x, None, None)
        if feat is None and self._fused_input_ok(x):
            out = self.mlp(self._fused_input(x))
            if self.min_max is not None:
                out = out * (self.min_max[:, 1] - self.min_max[:, 0]) + self.min_max[:, 0]
            return out
        if self.symmetrize:
            x = torch.cat([x[..., :1].abs(), x[..., 1:]], -1)
        h = x
        if self.embedder is not None:
            h = self.embedder(x)
            if self.embed_concat_pts:
                h = torch.cat([x, h], -1)
        if feat is None:  # relu(in_layer(.)) feeds the stack directly: one fused Linear+ReLU
            out = self.mlp(linear_relu(h, self.in_layer.weight, self.in_layer.bias))
        else:
            h = self.in_layer(h)
            if self.in_layer_relu:
                h = self.relu(h)
            while feat.dim() < h.dim():
                feat = feat.unsqueeze(1)
            h = torch.cat([h, feat.expand(*h.shape[:-1], -1)], dim=-1)
            out = self.mlp(self.relu(h))
        if self.min_max is not None:
            out = out * (self.min_max[:, 1] - self.min_max[:, 0]) + self.min_max[:, 0]
        return out

    def _forward_reference(self, x, feat):
        """The reference's own evaluation order (MLPs.py:73-98)."""
        if self.symmetrize:
            xs, ys, zs = x.unbind(-1)
            x = torch.stack([xs.abs(), ys, zs], -1)
        x_in = x
        if self.embedder is not None:
            x_in = self.embedder(x)
            if self.embed_concat_pts:
                x_in = torch.cat([x, x_in], -1)
        x_in = self.in_layer(x_in)
        if self.in_layer_relu:
            x_in = self.relu(x_in)
        if feat is not None:
            for _ in range(x_in.dim() - feat.dim()):
                feat = feat.unsqueeze(1)
            x_in = torch.cat([x_in, feat.expand(*x_in.shape[:-1], -1)], dim=-1)
        out = self.mlp(self.relu(x_in))
        if self.min_max is not None:
            out = out * (self.min_max[:, 1] - self.min_max[:, 0]) + self.min_max[:, 0]
        return out

    def _fused_input_ok(self, x):
        """Long point list on the GPU with the standard input stage (embedding + concatenated points): one HIP kernel builds
        [x, sin, cos, 1] and the in_layer's bias rides as the weight column that multiplies the ones (K = 64 for 10 frequencies)."""
        return (x.is_cuda and x.dim() == 2 and x.shape[1] == 3 and x.shape[0] >= SPLITK_MIN_ROWS and x.dtype == torch.float32
                and self.embedder is not None and self.embed_concat_pts and self.in_layer.bias is not None)

    def _fused_input(self, x):
        """relu(in_layer([x, embed(x)])) for a long point list (first-order differentiable)."""
        from . import ops

        h = ops.harmonic_embed(x, self.embedder._frequencies(x.device), symmetrize=self.symmetrize, ones=True)
        weight = torch.cat([self.in_layer.weight, self.in_layer.bias[:, None]], dim=1)
        return linear_relu(h, weight, None)

    def _stack_ok(self, x, with_feat):
        """The whole hidden stack as one autograd node (_FieldStack): long list, training, 256-wide bias-free Linear/ReLU pairs."""
        if not (USE_FIELD_STACK and self._fused_input_ok(x) and torch.is_grad_enabled() and x.shape[0] % SPLITK_PARTS == 0
                and self.in_layer.weight.requires_grad and self.in_layer.weight.shape[0] == 256):
            return False
        layers = list(self.mlp.network)
        n_pairs = 0
        while 2 * n_pairs + 1 < len(layers) and isinstance(layers[2 * n_pairs], nn.Linear) and isinstance(layers[2 * n_pairs + 1], nn.ReLU):
            lin = layers[2 * n_pairs]
            want_in = 256 + (self.extra_feat_dim if (with_feat and n_pairs == 0) else 0)
            if lin.bias is not None or tuple(lin.weight.shape) != (256, want_in):
                return False
            n_pairs += 1
        return n_pairs >= 1

    def _stacked(self, x, feat, feat_index):
        from . import ops

        layers = list(self.mlp.network)
        pairs, i = [], 0
        while i + 1 < len(layers) and isinstance(layers[i], nn.Linear) and isinstance(layers[i + 1], nn.ReLU):
            pairs.append(layers[i])
            i += 2
        x_emb = ops.harmonic_embed(x, self.embedder._frequencies(x.device), symmetrize=self.symmetrize, ones=True)
        w_in = torch.cat([self.in_layer.weight, self.in_layer.bias[:, None]], dim=1)
        if feat is not None:
            first = pairs[0].weight
            per_image = F.linear(torch.relu(feat), first[:, 256:])  # [B, 256]
            h = _FieldStack.apply(x_emb, w_in, per_image, feat_index, first[:, :256].contiguous(), *[l.weight for l in pairs[1:]])
        else:
            h = _FieldStack.apply(x_emb, w_in, None, None, None, *[l.weight for l in pairs])
        for layer in layers[i:]:
            h = linear(h, layer.weight, layer.bias) if isinstance(layer, nn.Linear) else layer(h)
        if self.min_max is not None:
            h = h * (self.min_max[:, 1] - self.min_max[:, 0]) + self.min_max[:, 0]
        return h

    def _forward_indexed(self, x, feat, feat_index):
        if self._stack_ok(x, True):
            return self._stacked(x, feat, feat_index)
        if self._fused_input_ok(x):
            h = self._fused_input(x)
        else:
            if self.symmetrize:
                x = torch.cat([x[..., :1].abs(), x[..., 1:]], -1)
            h = x
            if self.embedder is not None:
                h = self.embedder(x)
                if self.embed_concat_pts:
                    h = torch.cat([x, h], -1)
            h = linear_relu(h, self.in_layer.weight, self.in_layer.bias)
        nf = h.shape[-1]
        weight = self.mlp.network[0].weight
        per_image = F.linear(torch.relu(feat), weight[:, nf:])  # [B, nf]
        out = self.mlp(h, first_weight=weight[:, :nf].contiguous(), per_image=per_image, index=feat_index)
        if self.min_max is not None:
            out = out * (self.min_max[:, 1] - self.min_max[:, 0]) + self.min_max[:, 0]
        return out

    def sample(self, x, feat=None, feat_index=None):
        return self.forward(x, feat, feat_index)


# ---------------------------------------------------------------------------------------------------------------------------------
# Weight-modulated field of the pan-category model (Fauna): same classes and state_dict layout as the reference's
# CoordMLP_Mod / MLP_Mod / Linear_Mod (networks/MLPs.py:104-247).  One 128-d embedding per batch modulates and demodulates every
# layer's weight (StyleGAN2 style); plain torch.
class Linear_Mod(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty((out_features, in_features)))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features))
        else:
            self.register_parameter("bias", None)
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / fan_in ** 0.5 if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, style):
        if style.dim() > 1:  # one style per batch: the first row (MLPs.py:233-235)
            style = style.reshape(-1, style.shape[-1])[0]
        weight = self.weight * style.unsqueeze(0)
        weight = weight / ((weight * weight).sum(dim=-1, keepdim=True) + 1e-5).sqrt()
        return F.linear(x, weight, self.bias)


class MLP_Mod(nn.Module):
    def __init__(self, cin, cout, num_layers, nf=256, dropout=0, activation=None):
        super().__init__()
        assert num_layers >= 1
        self.num_layers = num_layers
        if num_layers == 1:
            self.network = Linear_Mod(cin, cout, bias=False)
        else:
            self.relu = nn.ReLU(inplace=True)
            for i in range(num_layers):
                setattr(self, f"linear_{i}", Linear_Mod(cin if i == 0 else nf, cout if i == num_layers - 1 else nf, bias=False))

    def forward(self, x, style):
        if self.num_layers == 1:
            return self.network(x, style)
        for i in range(self.num_layers):
            x = getattr(self, f"linear_{i}")(x, style)
            if i < self.num_layers - 1:
                x = self.relu(x)
        return x


class CoordMLP_Mod(nn.Module):
    def __init__(self, cin, cout, num_layers, nf=256, dropout=0, activation=None, min_max=None, n_harmonic_functions=10, embedder_scalar=1,
                 embed_concat_pts=True, extra_feat_dim=0, symmetrize=False, condition_dim=128):
        super().__init__()
        self.extra_feat_dim, self.condition_dim = extra_feat_dim, condition_dim
        if n_harmonic_functions > 0:
            self.embedder = HarmonicEmbedding(n_harmonic_functions, embedder_scalar)
            dim_in = cin * 2 * n_harmonic_functions + (cin if embed_concat_pts else 0)
            self.embed_concat_pts = embed_concat_pts
        else:
            self.embedder = None
            dim_in = cin
        self.in_layer = nn.Linear(dim_in, nf)
        self.relu = nn.ReLU(inplace=True)
        self.mlp = MLP_Mod(nf, cout, num_layers, nf, dropout, activation)
        self.style_mlp = MLP(condition_dim, nf, 2, nf, dropout, None)
        self.symmetrize = symmetrize
        if min_max is not None:
            self.register_buffer("min_max", min_max)
        else:
            self.min_max = None
        self.bsdf = None

    def forward(self, x, feat=None):
        assert feat is not None and feat.shape[-1] == self.condition_dim
        if self.symmetrize:
            x = torch.cat([x[..., :1].abs(), x[..., 1:]], -1)
        x_in = x
        if self.embedder is not None:
            x_in = self.embedder(x)
            if self.embed_concat_pts:
                x_in = torch.cat([x, x_in], -1)
        x_in = self.relu(self.in_layer(x_in))
        out = self.mlp(x_in, self.style_mlp(feat))
        if self.min_max is not None:
            out = out * (self.min_max[:, 1] - self.min_max[:, 0]) + self.min_max[:, 0]
        return out

    def sample(self, x, feat=None):
        return self.forward(x, feat)
